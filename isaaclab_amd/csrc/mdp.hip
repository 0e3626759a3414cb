// Fused MDP step: termination / reward / reset-bookkeeping / command / observation managers in ONE launch.
//
// A kBlock = 256-thread block owns kE = 16 consecutive environments and works in three stages:
//   STAGE   every row the active terms read (joint arrays, action, contact timers, command, root state, episode
//           sums ...) is one contiguous chunk for 16 consecutive envs, so the block copies the chunks into LDS
//           with fully coalesced loads, all issued back to back: one memory round trip for the whole step.  The
//           peak contact force per body (max over the history of |F|) is reduced on the way in.
//   DECIDE  one LANE per environment (16 lanes of each of the 4 waves): per-env scalar math -- root-frame
//           velocities, terminations, rewards, reset bookkeeping, command update -- runs once per env instead of
//           once per lane of a wave.  The four waves split the reward terms round-robin.
//   EMIT    all 256 threads write the observation rows (16 threads x 4 consecutive columns per env and trip,
//           one Philox4x32 call per 4 noisy columns), episode sums, per-term rewards and termination flags from
//           LDS: coalesced stores.
// The chain termination -> reward -> reset -> command -> observation never round-trips through memory.
// Term tables are data (`kind` dispatch); term formulas cite the reference file:line; op order for decisions
// follows math_dev.h.
#include "mdp_shared.h"

#include <cstring>
#include <algorithm>
#include <vector>

namespace ilx {

constexpr int kE = 16;             // environments per block
constexpr int kBlock = 256;        // threads per block (measured on the ANYmal-C plan, envs x threads -> us:
                                   // 8x256 66, 16x256 46, 32x256 56, 16x512 77, 32x512 45)
static_assert(kWaves == kBlock / 64, "the reward terms are dealt to the block's wavefronts");
constexpr int kMaxLdsBytes = 64 * 1024;
constexpr int kMaxSegs = 160;      // STAGE copy instructions per block (a kernel-argument table of 16-byte records)

enum Need : uint32_t {
    kNeedJoint = 1u, kNeedForces = 2u, kNeedTimers = 4u, kNeedBodyVel = 8u, kNeedBodyAcc = 16u, kNeedAction = 32u,
    kNeedRays = 64u,
};

// Word (4-byte) offsets into the block's LDS image; rows are [env][column] unless stated.
struct Layout {
    int terms, ids;
    int q, qd, qdd, tau_a, tau_c, q0, qd0, vlim, lim;  // joints (lim: 2 per joint)
    int act, pact;                                     // action columns
    int air, last_air, contact_t;                      // sensor bodies
    int fraw;                                          // force history [env][history][body][3] as it lies in memory
    int bvx, bvy, bacc;                                // asset bodies
    int cmd, root;                                     // command row, 13-wide root state
    int eps, srew;                                     // episode sums [term][env], step reward [env][term]
    int ptotal;                                        // reward partial totals [wave][env]
    int derived;                                       // [env][kDerived]
    int bits;                                          // term bit mask per env, then done flag per env
    int logp;                                          // per-block partial sums of the logging vector
    int coldesc;                                       // observation columns: one 16-byte ColDesc each
    int zero;                                          // one word of 0.0f (the subtrahend of columns that subtract nothing)
    int total;
    int colrec;                                        // observation columns resolved for this launch: two 16-byte words each (ColRec)
};
constexpr int kDerived = 10;  // lin_b(3) ang_b(3) grav_b(3) sensor_z
static_assert(offsetof(Layout, total) < 32 * 4, "ColDesc names a Layout field (all those before `total`) with 5 bits");

// How the EMIT stage produces one observation column, prepared once per plan.  Almost every column is a word of the
// block's LDS image (minus another one for the *_rel terms): `row` names the image row by its field in Layout, `k` the
// column inside an env's part of it, `stride` the width of that part.
enum ColMode : uint32_t { kColAffine = 0, kColScan = 1, kColGeneric = 2 };
struct alignas(16) ColDesc {
    uint32_t mode_term;  // ColMode | observation term index << 8
    uint32_t src;        // affine: Layout field (word index in the struct) << 16 | k;  others: column inside the term
    uint32_t sub;        // the same for the subtrahend; columns that subtract nothing name the image's zero word
    uint32_t stride;     // stride of src | stride of sub << 16 (0 for the zero word)
};
// The words of an ilx_term the EMIT stage reads, as two aligned 16-byte halves (terms lie 48 bytes apart in LDS).
struct alignas(16) TermHead {
    int32_t kind, flags;
    float weight, p0, p1, p2, p3, aux;
};
static_assert(sizeof(ilx_term) == 48 && offsetof(ilx_term, aux) == 28, "TermHead mirrors the head of ilx_term");
// A column as the EMIT stage's env loop wants it, resolved once per block (by the wavefronts that wait while the first one
// runs the command term) from the column's descriptor, its term's parameters and the launch's layout.
struct alignas(16) ColRec {
    uint32_t base;     // LDS byte address of the value for the block's first env | that of the subtrahend << 16
    uint32_t stride;   // bytes per env of the value's row | of the subtrahend's << 16
    float n_off, n_rng;         // noise: x + u * n_rng + n_off (0, 0: none)
    float c_lo, c_hi;           // clip bounds (-inf, +inf: none)
    float scale;                // 1: none
    uint32_t meta;     // ColMode | draws noise << 2 | observation term << 3 | byte offset into the env's ray hits << 11
};
struct alignas(16) MdpDev {
    ilx_mdp d;                // by-value copy; the HOST pointers inside are nulled
    const ilx_term* terms;    // device: [terminations | rewards | observations]
    const int32_t* ids;       // device
    const uint32_t* cols;     // device: one ColDesc per observation column
    uint32_t need;
    int any_noise;            // some observation term draws noise
    uint32_t noisy_terms;     // bit i: observation term i draws noise (kMaxTerms = 32)
    Layout L;
    uint32_t inv_rewards;     // ceil(2^32 / n): i / n == umulhi(i, inv) for the i < 2^16 of a block
    int num_segs;             // STAGE work list, see stage_segments
    uint4 segs[kMaxSegs];
};

// The layout and the descriptor's most used scalars: a second, small kernel argument that the kernel reads in place (a
// few scalar loads at its start, straight into scalar registers).
struct Hot {
    Layout L;
    int num_joints, num_bodies, num_sensor_bodies, command_dim, action_dim, num_rays, command_kind;
    int num_terminations, num_rewards, obs_dim, history_len, num_segs, dense_scan;
    uint32_t phases, need, inv_rewards;
    float step_dt, gravity[3];
    int64_t num_envs;
    uint64_t seed, step_index;
    const ilx_term* terms;  // the plan's table in device memory: [terminations | rewards | observations]
    const uint64_t* step_index_device;
    uint64_t reward_owner;  // two bits per reward term: the wavefront that evaluates it
};

// One environment as seen by its lane in the DECIDE stage: the accessor type of mdp_shared.h over the block's LDS image.
struct Env {
    const MdpDev& m;
    const Hot& h;
    const float* S;      // LDS image
    const int32_t* ids;  // LDS copy of the index lists
    int e;               // env slot in the block
    int64_t env;
    f3 pos, lin_w, ang_w, lin_b, ang_b, grav_b;
    f4 quat;
    float heading;
    float time_left;
    int64_t command_counter;

    ILX_DEV const ilx_mdp& desc() const { return m.d; }
    ILX_DEV int id(const ilx_term& t, int k) const { return t.ids_cnt < 0 ? k : ids[t.ids_off + k]; }
    ILX_DEV int id2(const ilx_term& t, int k) const { return t.ids2_cnt < 0 ? k : ids[t.ids2_off + k]; }
    ILX_DEV int jcount(const ilx_term& t) const { return t.ids_cnt < 0 ? h.num_joints : t.ids_cnt; }
    ILX_DEV int bcount_sensor(const ilx_term& t) const { return t.ids_cnt < 0 ? h.num_sensor_bodies : t.ids_cnt; }
    ILX_DEV int bcount_asset(const ilx_term& t) const { return t.ids_cnt < 0 ? h.num_bodies : t.ids_cnt; }
    ILX_DEV int num_actions() const { return h.action_dim; }
    ILX_DEV int num_rays() const { return h.num_rays; }
    ILX_DEV int row(Fld f) const {  // the image row of a field (the field is a constant at every call site)
        const Layout& L = h.L;
        switch (f) {
            case Fld::q: return L.q;
            case Fld::qd: return L.qd;
            case Fld::qdd: return L.qdd;
            case Fld::tau_a: return L.tau_a;
            case Fld::tau_c: return L.tau_c;
            case Fld::q0: return L.q0;
            case Fld::qd0: return L.qd0;
            case Fld::vlim: return L.vlim;
            case Fld::air: return L.air;
            case Fld::last_air: return L.last_air;
            case Fld::contact_t: return L.contact_t;
            case Fld::bvx: return L.bvx;
            case Fld::bvy: return L.bvy;
            case Fld::bacc: return L.bacc;
        }
        return 0;
    }
    ILX_DEV float joint(Fld f, int j) const { return S[row(f) + e * h.num_joints + j]; }
    ILX_DEV float limit(int j, int side) const { return S[h.L.lim + 2 * (e * h.num_joints + j) + side]; }
    ILX_DEV float sbody(Fld f, int b) const { return S[row(f) + e * h.num_sensor_bodies + b]; }
    ILX_DEV float abody(Fld f, int b) const { return S[row(f) + e * h.num_bodies + b]; }
    ILX_DEV float action(int k) const { return S[h.L.act + e * h.action_dim + k]; }
    ILX_DEV float prev_action(int k) const { return S[h.L.pact + e * h.action_dim + k]; }
    ILX_DEV const float* body_row(int b) const { return m.d.body_state_w + 13 * (env * h.num_bodies + b); }
    ILX_DEV float ray_hit_z(int k) const { return m.d.ray_hits_w[3 * (env * h.num_rays + k) + 2]; }
    // max over the history of |F| of one sensor body (rewards.py:266 / terminations.py:157), from the force history as
    // it lies in memory: the largest sum of squares, then one square root (the root is monotonic and correctly
    // rounded, so this is the largest norm, bit for bit)
    // (the staged rows are in LDS: a body's peak is formed where it is asked for)
    static constexpr bool kReadsMemory = false;
    static constexpr int kBatch = 4;
    struct ForceRaw {
        float peak;
    };
    ILX_DEV ForceRaw force_raw(int b) const { return {force_peak(b)}; }
    ILX_DEV float force_peak_of(const ForceRaw& r) const { return r.peak; }
    ILX_DEV float force_peak(int b) const {
        const int H = h.history_len, CB = h.num_sensor_bodies;
        const float* f = S + h.L.fraw + 3 * (e * H * CB + b);
        float mx = -1.0f;
        for (int k = 0; k < H; ++k, f += 3 * CB) mx = fmaxf(mx, fmaf_(f[2], f[2], fmaf_(f[1], f[1], f[0] * f[0])));
        return mx < 0.0f ? -__builtin_inff() : sqrt_(mx);
    }
    ILX_DEV float cmd(int k) const { return S[h.L.cmd + e * h.command_dim + k]; }
};

// Column k of an observation term for env slot e (EMIT stage; everything except the height scan comes from LDS).
ILX_DEV float eval_obs(const MdpDev& m, const Hot& h, const float* S, const int32_t* ids, int e, int64_t env, const ilx_term& t, int k) {
    const ilx_mdp& d = m.d;
    const Layout& L = h.L;
    const int J = h.num_joints;
    const float* root = S + L.root + 13 * e;
    const float* der = S + L.derived + kDerived * e;
    auto jid = [&](int kk) { return e * J + (t.ids_cnt < 0 ? kk : ids[t.ids_off + kk]); };
    switch (t.kind) {
        case ILX_OBS_BASE_POS_Z: return root[2];                 // observations.py:33-37
        case ILX_OBS_BASE_LIN_VEL: return der[k];                 // :40-44
        case ILX_OBS_BASE_ANG_VEL: return der[3 + k];             // :47-51
        case ILX_OBS_PROJECTED_GRAVITY: return der[6 + k];        // :54-58
        case ILX_OBS_ROOT_POS_W: return root[k] - (d.env_origins ? d.env_origins[3 * env + k] : 0.0f);  // :61-65
        case ILX_OBS_ROOT_QUAT_W: {  // :68-82
            const float v = root[3 + k];
            return ((t.flags & ILX_OBS_FLAG_QUAT_UNIQUE) && root[3] < 0.0f) ? -v : v;
        }
        case ILX_OBS_ROOT_LIN_VEL_W: return root[7 + k];   // :85-89
        case ILX_OBS_ROOT_ANG_VEL_W: return root[10 + k];  // :92-96
        case ILX_OBS_JOINT_POS: return S[L.q + jid(k)];    // :104-111
        case ILX_OBS_JOINT_POS_REL: {                      // :114-121
            const int i = jid(k);
            return S[L.q + i] - S[L.q0 + i];
        }
        case ILX_OBS_JOINT_POS_LIMIT_NORMALIZED: {  // :124-137 (math.py:23-40)
            const int i = jid(k);
            const float lo = S[L.lim + 2 * i], hi = S[L.lim + 2 * i + 1];
            const float off = (lo + hi) * 0.5f;
            return div_(2.0f * (S[L.q + i] - off), hi - lo);
        }
        case ILX_OBS_JOINT_VEL: return S[L.qd + jid(k)];  // :140-147
        case ILX_OBS_JOINT_VEL_REL: {                     // :150-157
            const int i = jid(k);
            return S[L.qd + i] - S[L.qd0 + i];
        }
        case ILX_OBS_HEIGHT_SCAN:  // :165-173
            return der[9] - d.ray_hits_w[3 * (env * h.num_rays + k) + 2] - t.aux;
        case ILX_OBS_LAST_ACTION: return S[L.act + e * h.action_dim + k];          // :512-521
        case ILX_OBS_GENERATED_COMMANDS: return S[L.cmd + e * h.command_dim + k];  // :529-531
        case ILX_OBS_EXTERNAL: {  // a buffer another component keeps per env (:177-230: body wrench, IMU readings)
            const int slot = static_cast<int>(t.aux);
            return d.external_obs[slot][env * d.external_obs_dim[slot] + k];
        }
    }
    return 0.0f;
}

// A term record as wave-uniform values, read from the plan's table in memory with scalar loads (three of them, through
// the constant cache; no vector instruction).  A copy from the LDS image would land in vector registers and the
// compiler would then treat `kind`, the id ranges and the parameters as per-lane data: the dispatch on `kind` becomes
// thirty exec-masked compare-and-skip blocks and every loop bound a vector compare.
ILX_DEV ilx_term uniform_term(const ilx_term* table, int index) {
    ilx_term t;
    const auto* in = reinterpret_cast<const __attribute__((address_space(4))) int32_t*>(
                         (const __attribute__((address_space(4))) char*)table) + index * static_cast<int>(sizeof(ilx_term) / 4);
    int32_t* out = reinterpret_cast<int32_t*>(&t);
#pragma unroll
    for (int w = 0; w < static_cast<int>(sizeof(ilx_term) / 4); ++w) out[w] = in[w];
    return t;
}

// STAGE is a work list.  Every array a block stages is one contiguous chunk for its kE consecutive envs (or a table that
// is the same for all blocks); the host cuts each chunk into segments of one asynchronous global->LDS copy instruction
// (`global_load_lds`: the wavefront's 64 lanes write 64 consecutive 4- or 16-byte words straight into LDS, no register
// destination, so every copy of the block is in flight at once) and hands the list over as 16-byte records in the
// kernel arguments.  Wavefront w takes records w, w + 4, ...: one vector load brings them in (lane j holds record
// w + 4 j), a uniform loop reads them lane by lane.  The data is in LDS after the next vmcnt(0) + barrier.
//   x, y  address of the segment's first word in the array's row of env 0 (bytes)
//   z     words per env (0: a table) | lanes that copy in a block of kE envs << 16 | 16-byte copies << 23
//   w     LDS word of the segment's first word | first word of the segment within the chunk / 64 << 16
// Only the last block of a launch can hold fewer than kE envs; it recomputes the lane bounds and copies word by word
// (16-byte lanes could read past the end of the array).
typedef __attribute__((address_space(3))) void LdsPtr;
typedef const __attribute__((address_space(1))) void GlobalPtr;
typedef const __attribute__((address_space(4))) char KernargBytes;

ILX_DEV void stage_segments(float* S, const uint4 rec, int count, int64_t env0, int ne) {
    const uint32_t lane = threadIdx.x & 63;
    if (ne == kE) {
        // All lanes at once: the first byte of each lane's record for this block.  Per record the loop is then four lane
        // reads, the lane bound and the copy.
        const uint64_t mine = ((static_cast<uint64_t>(rec.y) << 32) | rec.x) +
                              static_cast<uint64_t>(static_cast<uint32_t>(env0)) * ((rec.z & 0xffffu) * 4u);
        const uint32_t mine_lo = static_cast<uint32_t>(mine), mine_hi = static_cast<uint32_t>(mine >> 32);
        const uint32_t lane4 = 4u * lane, lane16 = 16u * lane;
        for (int j = 0; j < count; ++j) {
            const uint32_t lo = __builtin_amdgcn_readlane(mine_lo, j), hi = __builtin_amdgcn_readlane(mine_hi, j),
                           a = __builtin_amdgcn_readlane(rec.z, j), b = __builtin_amdgcn_readlane(rec.w, j);
            const auto* base = scalar_base(reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo));
            uint32_t* q = reinterpret_cast<uint32_t*>(S) + (b & 0xffffu);
            if (lane < ((a >> 16) & 127u)) {
                if (a & (1u << 23)) __builtin_amdgcn_global_load_lds((GlobalPtr*)(base + lane16), (LdsPtr*)q, 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((GlobalPtr*)(base + lane4), (LdsPtr*)q, 4, 0, 0);
            }
        }
        return;
    }
    // the launch's last block, with fewer envs: tables as they are, chunks word by word up to the block's own end
    for (int j = 0; j < count; ++j) {
        const uint32_t lo = __builtin_amdgcn_readlane(rec.x, j), hi = __builtin_amdgcn_readlane(rec.y, j),
                       a = __builtin_amdgcn_readlane(rec.z, j), b = __builtin_amdgcn_readlane(rec.w, j);
        const uint32_t per_env = a & 0xffffu, lanes = (a >> 16) & 127u;
        const bool wide = (a >> 23) & 1u;
        const uint64_t base = ((static_cast<uint64_t>(hi) << 32) | lo) + static_cast<uint64_t>(env0) * (per_env * 4u);
        uint32_t* q = reinterpret_cast<uint32_t*>(S) + (b & 0xffffu);
        if (per_env == 0) {
            if (lane < lanes) {
                if (wide) __builtin_amdgcn_global_load_lds((GlobalPtr*)(base + 16u * lane), (LdsPtr*)q, 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((GlobalPtr*)(base + 4u * lane), (LdsPtr*)q, 4, 0, 0);
            }
        } else {
            const uint32_t first = (b >> 16) << 6, total = static_cast<uint32_t>(ne) * per_env;
            for (uint32_t r = 0; r < (wide ? 4u : 1u); ++r)
                if (first + 64 * r + lane < total)
                    __builtin_amdgcn_global_load_lds((GlobalPtr*)(base + 4u * (64 * r + lane)), (LdsPtr*)(q + 64 * r), 4, 0, 0);
        }
    }
}

#ifdef ILX_MDP_STAMPS  // diagnostics build only: wall-clock stamps (10 ns ticks) of thread 0 of the first blocks
__device__ unsigned long long g_mdp_stamps[16 * 2048];
__device__ unsigned int g_mdp_hwid[8 * 2048];  // per block: the four wavefronts' HW_ID, then their XCC_ID
#define ILX_STAMP(k) \
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_mdp_stamps[blockIdx.x * 16 + (k)] = wall_clock64();
#else
#define ILX_STAMP(k)
#endif

__global__ void __launch_bounds__(kBlock, 4) mdp_step_kernel(MdpDev args, const Hot hot) {
    extern __shared__ float S[];
    // The kernel arguments (descriptor, layout, work list: a few KB) go into LDS first, one 16-byte load per thread, and
    // are read from there.  Read in place they are scalar loads spread over the whole kernel, and each first touch of
    // one of their ~50 cache lines stalls its wavefront for a full memory latency.
    {
        const auto* src = reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(
            (KernargBytes*)__builtin_amdgcn_kernarg_segment_ptr());
        uint32_t* dst = reinterpret_cast<uint32_t*>(S);
        for (int i = threadIdx.x; i < static_cast<int>(sizeof(MdpDev) / 16); i += kBlock) {
            const uint32_t a = src[4 * i], b = src[4 * i + 1], c = src[4 * i + 2], e = src[4 * i + 3];
            dst[4 * i] = a, dst[4 * i + 1] = b, dst[4 * i + 2] = c, dst[4 * i + 3] = e;
        }
        __syncthreads();
    }
#ifdef ILX_MDP_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {
        g_mdp_hwid[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_mdp_hwid[blockIdx.x * 8 + 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    const MdpDev& m = *reinterpret_cast<const MdpDev*>(S);
    const ilx_mdp& d = m.d;
    const Layout& L = hot.L;
    const int NT = hot.num_terminations, NR = hot.num_rewards;
    const int64_t N = hot.num_envs;
    const int64_t env0 = static_cast<int64_t>(blockIdx.x) * kE;
    const int ne = static_cast<int>(N - env0 < kE ? N - env0 : kE);
    const int B = hot.num_bodies, CD = hot.command_dim;
    const int tid = threadIdx.x;
    const uint32_t phases = hot.phases, need = hot.need;
    const bool ph_term = phases & ILX_PHASE_TERMINATION, ph_rew = phases & ILX_PHASE_REWARD,
               ph_reset = phases & ILX_PHASE_RESET, ph_obs = phases & ILX_PHASE_OBSERVATION;
    const bool has_cmd_term = hot.command_kind != ILX_CMD_NONE;
    const bool ph_cmd = (phases & ILX_PHASE_COMMAND) && has_cmd_term;
    const bool cmd_state = has_cmd_term && (phases & (ILX_PHASE_COMMAND | ILX_PHASE_RESET));
    const int num_segs = hot.num_segs;
    const ilx_term* terms = reinterpret_cast<const ilx_term*>(S + L.terms);
    const int32_t* ids = reinterpret_cast<const int32_t*>(S + L.ids);
    // The generator's step word: the argument plus the caller's device-side count when there is one (a scalar load; the
    // caller advances the word between launches, so it is constant over this one).
    const uint64_t seed = hot.seed;
    uint64_t gen_step = hot.step_index;
    if (hot.step_index_device)
        gen_step += *reinterpret_cast<const __attribute__((address_space(4))) uint64_t*>(
            (const __attribute__((address_space(4))) char*)hot.step_index_device);

    ILX_STAMP(0)
    // ================= STAGE =================
    auto* p_action = uni(d.action);
    auto* p_prev_action = uni(d.prev_action);
    const bool sa = need & kNeedAction, spa = sa && p_prev_action;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    uint4 my_segs = {0u, 0u, 0u, 0u};
    if (wave + kWaves * lane < num_segs) my_segs = m.segs[wave + kWaves * lane];
    const bool decide = lane < ne;
    const int e = lane;
    const uint32_t ue = static_cast<uint32_t>(lane);
    const int64_t env = env0 + (decide ? e : 0);
    // per-env scalars of the DECIDE lanes (coalesced over the 16 lanes): their loads go out first and come back while
    // the row copies below are in flight
    int64_t ep_len = 0;
    CmdState cs;
    float sensor_z = 0.0f;
    bool terminated = false, time_out = false;
    uint32_t term_bits = 0;
    {
        // the pointers first (one batch of LDS reads), then the loads they guard
        const auto* p_len = uni(d.episode_length_buf);
        const auto* p_left = uni(d.time_left);
        const auto* p_count = uni(d.command_counter);
        const auto* p_metric = uni(d.command_metrics);
        const auto* p_head = uni(d.heading_target);
        const auto* p_is_head = uni(d.is_heading_env);
        const auto* p_is_stand = uni(d.is_standing_env);
        const auto* p_sensor = uni(d.sensor_pos_w);
        const auto* p_term = uni(d.terminated);
        const auto* p_tout = uni(d.time_outs);
        const auto* p_dones = uni(d.term_dones);
        const bool velocity = hot.command_kind == ILX_CMD_VELOCITY, rays = need & kNeedRays;
        if (decide) {
            if (p_len) ep_len = ld_at(p_len + env0, 8u * ue);
            if (p_left) cs.time_left = ld_at(p_left + env0, 4u * ue);
            if (p_count) cs.counter = ld_at(p_count + env0, 8u * ue);
            if (cmd_state && wave == 0) {
                cs.metric0 = ld_at(p_metric + env0, 4u * ue), cs.metric1 = ld_at(p_metric + N + env0, 4u * ue);
                if (velocity) {
                    cs.heading_target = ld_at(p_head + env0, 4u * ue);
                    cs.is_heading = ld_at(p_is_head + env0, ue) != 0;
                    cs.is_standing = ld_at(p_is_stand + env0, ue) != 0;
                }
            }
            if (rays && wave == 0) sensor_z = ld_at(p_sensor + 3 * env0, 12u * ue + 8u);
            if (!ph_term && p_term) {
                terminated = ld_at(p_term + env0, ue) != 0;
                time_out = ld_at(p_tout + env0, ue) != 0;
                for (int i = 0; i < NT; ++i) term_bits |= ld_at(p_dones + static_cast<int64_t>(i) * N + env0, ue) ? (1u << i) : 0u;
            }
        }
    }
    // strided rows (two columns of the 13-wide body states, 3-vectors of the body accelerations) come through registers
    float bv[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, ba[2] = {0.0f, 0.0f};
    const auto* p_body_state = uni(d.body_state_w);
    const auto* p_body_acc = uni(d.body_lin_acc_w);
    if (need & kNeedBodyVel) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kBlock;
            if (i < ne * B) {
                const auto* rows = p_body_state + 13 * (env0 * B) + 7;
                bv[k][0] = ld_at(rows, 52u * i), bv[k][1] = ld_at(rows, 52u * i + 4u);
            }
        }
    }
    if (need & kNeedBodyAcc) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kBlock;
            if (i < ne * B) {
                const auto* rows = p_body_acc + 3 * (env0 * B);
                ba[k] = norm3(ld_at(rows, 12u * i), ld_at(rows, 12u * i + 4u), ld_at(rows, 12u * i + 8u));
            }
        }
    }
    // every contiguous chunk and table: the work list (one memory round trip for all of it)
    ILX_STAMP(14)
    stage_segments(S, my_segs, (num_segs - wave + kWaves - 1) / kWaves, env0, ne);
    for (int i = tid; i < NR + NT + 3; i += kBlock) S[L.logp + i] = 0.0f;
    if (tid == 0) S[L.zero] = 0.0f;
    if (need & kNeedBodyVel) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kBlock;
            if (i < ne * B) S[L.bvx + i] = bv[k][0], S[L.bvy + i] = bv[k][1];
        }
        for (int i = tid + 2 * kBlock; i < ne * B; i += kBlock) {
            const auto* rows = p_body_state + 13 * (env0 * B) + 7;
            S[L.bvx + i] = ld_at(rows, 52u * i), S[L.bvy + i] = ld_at(rows, 52u * i + 4u);
        }
    }
    if (need & kNeedBodyAcc) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kBlock;
            if (i < ne * B) S[L.bacc + i] = ba[k];
        }
        for (int i = tid + 2 * kBlock; i < ne * B; i += kBlock) {
            const auto* rows = p_body_acc + 3 * (env0 * B);
            S[L.bacc + i] = norm3(ld_at(rows, 12u * i), ld_at(rows, 12u * i + 4u), ld_at(rows, 12u * i + 8u));
        }
    }
    ILX_STAMP(1)
    __syncthreads();  // waits for this wave's copies (vmcnt(0)), then for every wave's
    ILX_STAMP(2)
    // ================= DECIDE: one lane per env; the four waves split the reward terms (dealt by cost at plan creation) =================
    Env en{m, hot, S, ids, e, env};
    const float dt = hot.step_dt;
    const f3 gravity = {hot.gravity[0], hot.gravity[1], hot.gravity[2]};
    if (decide) {
        const float* rs = S + L.root + 13 * e;
        en.pos = {rs[0], rs[1], rs[2]};
        en.quat = {rs[3], rs[4], rs[5], rs[6]};
        en.lin_w = {rs[7], rs[8], rs[9]};
        en.ang_w = {rs[10], rs[11], rs[12]};
        // articulation_data.py:411-425, 502-518
        en.lin_b = quat_rotate_inverse(en.quat, en.lin_w);
        en.ang_b = quat_rotate_inverse(en.quat, en.ang_w);
        en.grav_b = quat_rotate_inverse(en.quat, gravity);
        en.time_left = cs.time_left, en.command_counter = cs.counter;
        // ---- terminations (manager_based_rl_env.py:199-202, termination_manager.py:151-174); every wave needs the result
        if (ph_term) {
            ep_len += 1;
            for (int i = 0; i < NT; ++i) {
                const ilx_term t = uniform_term(hot.terms, i);
                const bool v = eval_termination(en, t, ep_len);
                term_bits |= v ? (1u << i) : 0u;
                if (t.flags & 1) time_out = time_out || v; else terminated = terminated || v;
            }
        }
        ILX_STAMP(4)
        // ---- rewards (reward_manager.py:128-157): this wave's share of the terms
        if (ph_rew) {
            float partial = 0.0f;
            for (int i = 0; i < NR; ++i) {
                if (static_cast<int>((hot.reward_owner >> (2 * i)) & 3u) != wave) continue;
                const ilx_term t = uniform_term(hot.terms, NT + i);
                if (t.weight == 0.0f) continue;  // :145-146 (episode sums and step_reward keep their values)
                const float value = eval_reward(en, t, terminated, time_out, term_bits) * t.weight * dt;
                partial += value;
                S[L.eps + i * kE + e] += value;
                S[L.srew + e * NR + i] = div_(value, dt);
            }
            S[L.ptotal + wave * kE + e] = partial;
        }
        if (wave == 0) {  // the decisions, for the wavefronts that write them out
            reinterpret_cast<uint32_t*>(S + L.bits)[e] = term_bits;
            reinterpret_cast<uint32_t*>(S + L.bits)[kE + e] = (terminated || time_out) ? 1u : 0u;
        }
    }
    ILX_STAMP(5)
    __syncthreads();
    // The observation columns of this launch, resolved once per block by the wavefronts that have little to do until the
    // first one is through with the command term: descriptor -> LDS addresses under this launch's layout, the term's
    // pipeline parameters made neutral where a step is off (x + u * 0 + 0, clip to +-inf, x * 1: the env loop then runs
    // without branches or selects on them).  The records may lie over the force history (no reader is left).
    if (ph_obs && wave != 0) {
        const int O = hot.obs_dim;
        const ColDesc* cdesc = reinterpret_cast<const ColDesc*>(S + L.coldesc);
        const ilx_term* oterms = terms + NT + hot.num_rewards;
        const int32_t* ltab = reinterpret_cast<const int32_t*>(&m.L);   // the layout, as words (its LDS copy: indexed per lane)
        ColRec* recs = reinterpret_cast<ColRec*>(S + L.colrec);
        for (int col = (wave - 1) * 64 + lane; col < O; col += 64 * (kWaves - 1)) {
            const ColDesc cd = cdesc[col];
            const uint32_t term = cd.mode_term >> 8, mode = cd.mode_term & 255u;
            const TermHead th = *reinterpret_cast<const TermHead*>(oterms + term);
            const uint32_t flags = static_cast<uint32_t>(th.flags);
            const uint32_t a = static_cast<uint32_t>(ltab[(cd.src >> 16) & 31u]), b = static_cast<uint32_t>(ltab[(cd.sub >> 16) & 31u]);
            const bool affine = mode == kColAffine, noisy = flags & ILX_OBS_FLAG_NOISE, clip = flags & ILX_OBS_FLAG_CLIP;
            ColRec r;
            // value and subtrahend of the other kinds: the zero word
            r.base = affine ? ((4u * (a + (cd.src & 0xffffu))) | ((4u * (b + (cd.sub & 0xffffu))) << 16)) : 4u * static_cast<uint32_t>(L.zero) * 0x10001u;
            r.stride = affine ? ((4u * (cd.stride & 0xffffu)) | ((4u * (cd.stride >> 16)) << 16)) : 0u;
            r.n_off = noisy ? th.p0 : 0.0f, r.n_rng = noisy ? th.p1 : 0.0f;
            // clip as one median-of-three; bounds given the wrong way round clip to the upper one, as torch.clip does
            r.c_hi = clip ? th.p3 : __builtin_inff();
            r.c_lo = clip ? fminf(th.p2, th.p3) : -__builtin_inff();
            r.scale = (flags & ILX_OBS_FLAG_SCALE) ? th.weight : 1.0f;
            const uint32_t zoff = mode == kColScan ? (hot.dense_scan ? 4u * cd.src : 12u * cd.src + 8u) : 0u;  // bytes into the env's hits
            r.meta = mode | (noisy ? 4u : 0u) | (term << 3) | (zoff << 11);
            recs[col] = r;
        }
    }
    // What needs nothing but the decisions runs beside wave 0's command work.  Wave 1 adds the episode sums of the done
    // envs to the logging partials (reward_manager.py:115-121, termination_manager.py:142-144) and writes the sums out;
    // wave 2 restarts the done envs' action history (action_manager.py:309-311: the observations of this launch already
    // see zeros; the rewards above used the step's own actions) and writes the per-term rewards out; wave 3 writes the
    // termination terms' flags.
    static_assert(kWaves == 4, "the reset bookkeeping and the write-out of the decisions are dealt to wavefronts 1, 2 and 3");
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(S + L.bits);
    if (wave == 1) {
        if (decide && ph_reset && (terminated || time_out)) {
            float* logp = S + L.logp;
            for (int i = 0; i < NR; ++i) {
                atomicAdd(logp + i, S[L.eps + i * kE + e]);  // per-block partials in LDS (ds_add_f32), one global atomic per entry and block
                S[L.eps + i * kE + e] = 0.0f;
            }
            for (int i = 0; i < NT; ++i)
                if (term_bits & (1u << i)) atomicAdd(logp + NR + i, 1.0f);
            atomicAdd(logp + NR + NT + 2, 1.0f);
        }
        if (ph_rew || ph_reset) {  // the episode sums go out as they are after that (the wavefront's LDS operations keep their order)
            auto* o_sums = uni(d.episode_sums);
            for (int i = lane; i < NR * kE; i += 64) {
                const int term = i / kE, ee = i - term * kE;
                if (ee >= ne) continue;
                const bool active = ph_rew && terms[NT + term].weight != 0.0f;
                if (active || (ph_reset && bits[kE + ee])) o_sums[static_cast<int64_t>(term) * N + env0 + ee] = S[L.eps + i];
            }
        }
    }
    if (wave == 2) {
        if (decide && ph_reset && (terminated || time_out)) {
            auto* act = p_action;
            auto* pact = p_prev_action;
            const int A = hot.action_dim;
            for (int k = 0; k < A; ++k) {
                const uint32_t at = 4u * (static_cast<uint32_t>(e) * static_cast<uint32_t>(A) + k);
                if (act) st_at(act + env0 * A, at, 0.0f);
                if (pact) st_at(pact + env0 * A, at, 0.0f);
                if (sa) S[L.act + e * A + k] = 0.0f;
                if (spa) S[L.pact + e * A + k] = 0.0f;
            }
        }
        if (ph_rew) {
            auto* o_step_reward = uni(d.step_reward);
            const uint32_t inv = hot.inv_rewards;
            for (int i = lane; i < ne * NR; i += 64) {
                const int term = NR > 1 ? i - static_cast<int>(__umulhi(static_cast<uint32_t>(i), inv)) * NR : 0;  // i % NR
                if (terms[NT + term].weight != 0.0f) st_at(o_step_reward + env0 * NR, 4u * i, S[L.srew + i]);
            }
        }
    }
    if (wave == 3 && ph_term) {
        auto* o_dones = uni(d.term_dones);
        for (int i = lane; i < NT * kE; i += 64) {
            const int term = i / kE, ee = i - term * kE;
            if (ee < ne) o_dones[static_cast<int64_t>(term) * N + env0 + ee] = (bits[ee] >> term) & 1u;
        }
    }
    if (wave == 0) {
        ILX_STAMP(6)
        // the descriptor's words this part uses, in one batch of LDS reads
        auto* o_terminated = uni(d.terminated);
        auto* o_time_outs = uni(d.time_outs);
        auto* o_reset = uni(d.reset_buf);
        auto* o_reward = uni(d.reward_buf);
        auto* o_len = uni(d.episode_length_buf);
        auto* o_command = uni(d.command);
        auto* o_left = uni(d.time_left);
        auto* o_count = uni(d.command_counter);
        auto* o_metric = uni(d.command_metrics);
        auto* o_head = uni(d.heading_target);
        auto* o_is_head = uni(d.is_heading_env);
        auto* o_is_stand = uni(d.is_standing_env);
        const auto* in_reset_samples = uni(d.reset_command_samples);
        const auto* in_samples = uni(d.command_samples);
        const bool velocity = hot.command_kind == ILX_CMD_VELOCITY;
        const float max_command_step = uni(d.max_command_step), stiffness = uni(d.heading_control_stiffness),
                    yaw_lo = uni(d.cmd_range[3][0]), yaw_hi = uni(d.cmd_range[3][1]);
        const bool heading_command = uni(d.heading_command) != 0;
      if (decide) {
        const bool done = terminated || time_out;
        if (ph_term) {
            st_at(o_terminated + env0, ue, terminated);
            st_at(o_time_outs + env0, ue, time_out);
            if (o_reset) st_at(o_reset + env0, ue, done);
        }
        if (ph_rew)
        {
            float total = S[L.ptotal + e];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) total += S[L.ptotal + w * kE + e];
            st_at(o_reward + env0, 4u * ue, total);
        }
        if (has_cmd_term) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < CD) cs.c[k] = en.cmd(k);
        }
        // ---- reset bookkeeping of the managers for done envs (manager_based_rl_env.py:364-391)
        bool cmd_dirty = false;
        if (ph_reset && done) {
            if (has_cmd_term) {  // command_manager.py:120-149
                float* logp = S + L.logp;
                atomicAdd(logp + NR + NT, cs.metric0);
                atomicAdd(logp + NR + NT + 1, cs.metric1);
                cs.metric0 = cs.metric1 = 0.0f;
                cs.counter = 0;
                command_resample(d, seed, gen_step, env, in_reset_samples, 16u, cs);
                cmd_dirty = true;
            }
            ep_len = 0;  // manager_based_rl_env.py:391
        }
        if (o_len && (ph_term || (ph_reset && done))) st_at(o_len + env0, 8u * ue, ep_len);
        // ---- command term (command_manager.py:151-166)
        if (ph_cmd) {
            command_compute(d, en, velocity, heading_command, dt, max_command_step, stiffness, yaw_lo, yaw_hi, seed, gen_step, env,
                            in_samples, p_body_state + 13 * (env * B + d.pose_body_index), cs);
            cmd_dirty = true;
        }
        if (cmd_dirty) {
            const int width = velocity ? 3 : 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (k < width) {
                    S[L.cmd + e * CD + k] = cs.c[k];
                    st_at(o_command + env0 * CD, 4u * (ue * static_cast<uint32_t>(CD) + k), cs.c[k]);
                }
            }
            st_at(o_left + env0, 4u * ue, cs.time_left);
            st_at(o_count + env0, 8u * ue, cs.counter);
            st_at(o_metric + env0, 4u * ue, cs.metric0);
            st_at(o_metric + N + env0, 4u * ue, cs.metric1);
            if (velocity) {
                st_at(o_head + env0, 4u * ue, cs.heading_target);
                st_at(o_is_head + env0, ue, cs.is_heading);
                st_at(o_is_stand + env0, ue, cs.is_standing);
            }
        }
        ILX_STAMP(7)
        float* der = S + L.derived + kDerived * e;
        der[0] = en.lin_b.x, der[1] = en.lin_b.y, der[2] = en.lin_b.z;
        der[3] = en.ang_b.x, der[4] = en.ang_b.y, der[5] = en.ang_b.z;
        der[6] = en.grav_b.x, der[7] = en.grav_b.y, der[8] = en.grav_b.z;
        der[9] = sensor_z;
      }
    }
    __syncthreads();

    ILX_STAMP(8)
    // ================= EMIT =================
    if (ph_reset && tid < NR + NT + 3) {
        auto* o_log = uni(d.log_sums);
        const float v = S[L.logp + tid];
        if (o_log && v != 0.0f) unsafeAtomicAdd((float*)(o_log + tid), v);  // hardware float add at L2, no return value
    }
    ILX_STAMP(9)
    // ---- observations (observation_manager.py:260-335).  The concatenated row is cut into groups of 4 columns (a
    // group's noise is one Philox4x32 call).  A lane owns one group and walks down the block's envs with it, so the
    // columns' descriptors and their terms' noise / clip / scale parameters are fetched once per lane; a wavefront
    // covers 64 / (groups, rounded up to a power of two) envs per trip.  Per column and env what is left is the
    // value's LDS word (or its ray hit), the pipeline arithmetic and the store.
    if (ph_obs) {
        const int O = hot.obs_dim;
        const int G = (O + 3) >> 2;
        int gp = 1;
        while (gp < G && gp < 64) gp <<= 1;
        const int per = 64 / gp, g_in = lane & (gp - 1), esub = lane / gp;
        const ilx_term* oterms = terms + NT + NR;
        const ColDesc* cdesc = reinterpret_cast<const ColDesc*>(S + L.coldesc);
        const auto* p_scan = uni(d.height_scan);
        const auto* p_hits = uni(d.ray_hits_w);
        const bool dense_scan = p_scan != nullptr;  // the ray caster's own height scan: value as it is
        const auto* hits = dense_scan ? p_scan : p_hits;
        const auto* noise_in = uni(d.noise_u);
        auto* o_obs = uni(d.obs);
        const int R = hot.num_rays;
        const bool generate = !noise_in;
        // memory: the block's rows as uniform bases
        auto* rows = o_obs + env0 * O;
        const auto* noise_rows = noise_in + env0 * O;
        const auto* hit_rows = hits + (dense_scan ? 1 : 3) * (env0 * R);
        const uint32_t hit_env_bytes = 4u * static_cast<uint32_t>((dense_scan ? 1 : 3) * R);
        const ColRec* recs = reinterpret_cast<const ColRec*>(S + L.colrec);
        for (int g = g_in; g < G; g += 64) {
            // The lane's four columns: their records (independent reads, one LDS latency), then the lane's own start in
            // the rows.  Past the row's end: like the last column, not stored.
            ColRec rec[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) rec[c] = recs[4 * g + c < O ? 4 * g + c : O - 1];
            // LDS byte addresses of value (low half) and subtrahend (high half) for the lane's first env, and what one
            // trip of the env loop adds to them (the halves never carry: every address is below 64 KB)
            const uint32_t ee0 = static_cast<uint32_t>(wave * per + esub), estep = static_cast<uint32_t>(kWaves * per);
            uint32_t addr[4], astep[4], meta[4], zoff[4], all_meta = 0;
            float n_off[4], n_rng[4], c_lo[4], c_hi[4], scl[4], hgt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                addr[c] = rec[c].base + ee0 * rec[c].stride, astep[c] = estep * rec[c].stride;
                n_off[c] = rec[c].n_off, n_rng[c] = rec[c].n_rng, c_lo[c] = rec[c].c_lo, c_hi[c] = rec[c].c_hi, scl[c] = rec[c].scale;
                meta[c] = rec[c].meta, zoff[c] = rec[c].meta >> 11, all_meta |= rec[c].meta;
                hgt[c] = 0.0f;
            }
            const uint32_t all_modes = all_meta & 3u;
            const bool any_noise = all_meta & 4u, any_scan = all_modes & kColScan;
            if (any_scan && !dense_scan) {  // the term's offset, for hits read as records
#pragma unroll
                for (int c = 0; c < 4; ++c) hgt[c] = oterms[(meta[c] >> 3) & 255u].aux;
            }
            const char* lds = reinterpret_cast<const char*>(S);
            ILX_STAMP(11)
            // memory: the lane's place in the block's rows as running 32-bit byte offsets
            uint32_t row_at = 4u * (ee0 * static_cast<uint32_t>(O) + 4u * static_cast<uint32_t>(g));  // column 4 g of the lane's env
            uint32_t hit_at = ee0 * hit_env_bytes;
            const uint32_t row_step = 4u * estep * static_cast<uint32_t>(O), hit_step = estep * hit_env_bytes;
            for (int ee = wave * per + esub; ee < ne; ee += kWaves * per, row_at += row_step, hit_at += hit_step) {
                const int64_t oenv = env0 + ee;
                // values from memory (ray hits, injected noise) are requested first; the generator call hides their latency
                float zv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (any_scan) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) zv[c] = ld_at(hit_rows, hit_at + zoff[c]);  // columns of other kinds read the first ray (unused)
                }
                if (!generate && any_noise) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) u[c] = ld_at(noise_rows, 4 * g + c < O ? row_at + 4u * c : row_at - 16u * g + 4u * (O - 1));
                }
                // LDS: value and subtrahend, the sensor height
                float val[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    val[c] = *reinterpret_cast<const float*>(lds + (addr[c] & 0xffffu)) - *reinterpret_cast<const float*>(lds + (addr[c] >> 16));
                    addr[c] += astep[c];
                }
                const float sensor_height = S[L.derived + kDerived * ee + 9];
                if (ee < 4) { ILX_STAMP(12) }
                if (generate && any_noise) {
                    const u4 r = philox4(seed, gen_step, static_cast<uint64_t>(oenv), static_cast<uint32_t>(g));
                    u[0] = u01(r.x), u[1] = u01(r.y), u[2] = u01(r.z), u[3] = u01(r.w);
                }
                if (ee < 4) { ILX_STAMP(13) }
                if (any_scan && !dense_scan) {  // observations.py:165-173 (the dense scan holds this difference already)
#pragma unroll
                    for (int c = 0; c < 4; ++c) zv[c] = sensor_height - zv[c] - hgt[c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t mode = meta[c] & 3u;
                    float v = mode == kColScan ? zv[c] : val[c];
                    v = v + u[c] * n_rng[c] + n_off[c];                                          // noise_model.py:63-64
                    v = __builtin_amdgcn_fmed3f(v, c_lo[c], c_hi[c]);                            // observation_manager.py:315-316
                    v = v * scl[c];                                                              // :317-318
                    if (4 * g + c < O && mode != kColGeneric) st_at(rows, row_at + 4u * c, v);
                }
                // columns of the other kinds (root position with env origins, unique quaternions, normalised joint limits,
                // external buffers): one at a time through the term switch
                if (all_modes & kColGeneric) {
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t w = c == 0 ? meta[0] : c == 1 ? meta[1] : c == 2 ? meta[2] : meta[3];
                        if ((w & 3u) != kColGeneric || 4 * g + c >= O) continue;
                        const uint32_t k = cdesc[4 * g + c].src;
                        const float uc = c == 0 ? u[0] : c == 1 ? u[1] : c == 2 ? u[2] : u[3];
                        const ilx_term& t = oterms[(w >> 3) & 255u];
                        float v = eval_obs(m, hot, S, ids, ee, oenv, t, static_cast<int>(k));
                        if (t.flags & ILX_OBS_FLAG_NOISE) v = v + uc * t.p[1] + t.p[0];
                        if (t.flags & ILX_OBS_FLAG_CLIP) v = fminf(fmaxf(v, t.p[2]), t.p[3]);
                        if (t.flags & ILX_OBS_FLAG_SCALE) v = v * t.weight;
                        st_at(rows, row_at + 4u * c, v);
                    }
                }
            }
        }
    }
    ILX_STAMP(10)
}

}  // namespace ilx

using namespace ilx;

static int host_obs_dim(const ilx_mdp& d, const ilx_term& t) {
    const int nj = t.ids_cnt < 0 ? d.num_joints : t.ids_cnt;
    switch (t.kind) {
        case ILX_OBS_BASE_POS_Z: return 1;
        case ILX_OBS_ROOT_QUAT_W: return 4;
        case ILX_OBS_JOINT_POS: case ILX_OBS_JOINT_POS_REL: case ILX_OBS_JOINT_POS_LIMIT_NORMALIZED:
        case ILX_OBS_JOINT_VEL: case ILX_OBS_JOINT_VEL_REL: return nj;
        case ILX_OBS_HEIGHT_SCAN: return d.num_rays;
        case ILX_OBS_LAST_ACTION: return d.action_dim;
        case ILX_OBS_GENERATED_COMMANDS: return d.command_dim;
        case ILX_OBS_EXTERNAL: return d.external_obs_dim[static_cast<int>(t.aux)];
        default: return 3;
    }
}

static int validate_terms(const ilx_mdp& d) {
    ILX_REQUIRE(d.num_terminations >= 0 && d.num_terminations <= kMaxTerms && d.num_rewards >= 0 &&
                    d.num_rewards <= kMaxTerms && d.num_observations >= 0 && d.num_observations <= kMaxTerms,
                ILX_ESHAPE, "ilx_mdp: at most %d terms per manager", kMaxTerms);
    ILX_REQUIRE((d.num_terminations == 0 || d.terminations) && (d.num_rewards == 0 || d.rewards) &&
                    (d.num_observations == 0 || d.observations),
                ILX_EINVAL, "ilx_mdp: NULL term table");
    auto ids_ok = [&](const ilx_term& t, int limit, const char* what) {
        if (t.ids_cnt < 0) return ILX_OK;
        ILX_REQUIRE(t.ids_off >= 0 && t.ids_off + t.ids_cnt <= d.num_ids, ILX_EINVAL, "ilx_mdp: %s id list out of range", what);
        for (int k = 0; k < t.ids_cnt; ++k)
            ILX_REQUIRE(d.ids[t.ids_off + k] >= 0 && d.ids[t.ids_off + k] < limit, ILX_EINVAL,
                        "ilx_mdp: %s index %d outside [0, %d)", what, d.ids[t.ids_off + k], limit);
        return ILX_OK;
    };
    for (int i = 0; i < d.num_terminations; ++i) {
        const ilx_term& t = d.terminations[i];
        ILX_REQUIRE(t.kind >= 0 && t.kind < ILX_TERM_KIND_COUNT, ILX_EINVAL, "ilx_mdp: unknown termination kind %d", t.kind);
        const int rc = ids_ok(t, t.kind == ILX_TERM_ILLEGAL_CONTACT ? d.num_sensor_bodies : d.num_joints, "termination");
        if (rc != ILX_OK) return rc;
    }
    for (int i = 0; i < d.num_rewards; ++i) {
        const ilx_term& t = d.rewards[i];
        ILX_REQUIRE(t.kind >= 0 && t.kind < ILX_REW_KIND_COUNT, ILX_EINVAL, "ilx_mdp: unknown reward kind %d", t.kind);
        const bool sensor = t.kind == ILX_REW_UNDESIRED_CONTACTS || t.kind == ILX_REW_CONTACT_FORCES ||
                            t.kind == ILX_REW_FEET_AIR_TIME || t.kind == ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED ||
                            t.kind == ILX_REW_FEET_SLIDE;
        const bool body = t.kind == ILX_REW_BODY_LIN_ACC_L2 || t.kind == ILX_REW_POSITION_COMMAND_ERROR ||
                          t.kind == ILX_REW_POSITION_COMMAND_ERROR_TANH || t.kind == ILX_REW_ORIENTATION_COMMAND_ERROR;
        const int rc = ids_ok(t, sensor ? d.num_sensor_bodies : (body ? d.num_bodies : d.num_joints), "reward");
        if (rc != ILX_OK) return rc;
        if (body && t.kind != ILX_REW_BODY_LIN_ACC_L2)
            ILX_REQUIRE(t.ids_cnt >= 1 && d.command_dim >= 7, ILX_EINVAL,
                        "ilx_mdp: pose-command reward needs a body id and a 7-wide command");
        if (t.kind == ILX_REW_FEET_SLIDE) {
            ILX_REQUIRE((t.ids2_cnt < 0) == (t.ids_cnt < 0) && (t.ids2_cnt < 0 || t.ids2_cnt == t.ids_cnt), ILX_EINVAL,
                        "ilx_mdp: feet_slide needs sensor and asset body lists of equal length");
            if (t.ids2_cnt >= 0) {
                ILX_REQUIRE(t.ids2_off >= 0 && t.ids2_off + t.ids2_cnt <= d.num_ids, ILX_EINVAL, "ilx_mdp: ids2 out of range");
                for (int k = 0; k < t.ids2_cnt; ++k)
                    ILX_REQUIRE(d.ids[t.ids2_off + k] >= 0 && d.ids[t.ids2_off + k] < d.num_bodies, ILX_EINVAL,
                                "ilx_mdp: feet_slide asset body index out of range");
            }
        }
    }
    int dim = 0;
    for (int i = 0; i < d.num_observations; ++i) {
        const ilx_term& t = d.observations[i];
        ILX_REQUIRE(t.kind >= 0 && t.kind < ILX_OBS_KIND_COUNT, ILX_EINVAL, "ilx_mdp: unknown observation kind %d", t.kind);
        const int rc = ids_ok(t, d.num_joints, "observation");
        if (rc != ILX_OK) return rc;
        if (t.kind == ILX_OBS_EXTERNAL) {
            const int slot = static_cast<int>(t.aux);
            ILX_REQUIRE(t.aux >= 0.0f && slot < ILX_MAX_EXTERNAL_OBS && static_cast<float>(slot) == t.aux, ILX_EINVAL,
                        "ilx_mdp: an external observation names slot %g of %d", static_cast<double>(t.aux), ILX_MAX_EXTERNAL_OBS);
            ILX_REQUIRE(d.external_obs_dim[slot] > 0, ILX_ESHAPE, "ilx_mdp: external observation slot %d has no width", slot);
        }
        dim += host_obs_dim(d, t);
    }
    ILX_REQUIRE(dim == d.obs_dim, ILX_ESHAPE, "ilx_mdp: observation terms add up to %d columns, obs_dim is %d", dim, d.obs_dim);
    return ILX_OK;
}

extern "C" int ilx_mdp_plan_create(const ilx_mdp* desc, ilx_mdp_plan_t* out) {
    ILX_REQUIRE(desc && out, ILX_EINVAL, "ilx_mdp_plan_create: NULL argument");
    *out = nullptr;
    const ilx_mdp& d = *desc;
    ILX_REQUIRE(d.num_ids == 0 || d.ids, ILX_EINVAL, "ilx_mdp_plan_create: ids is NULL");
    ILX_REQUIRE(d.num_ids >= 0 && d.num_ids <= 4096, ILX_ESHAPE, "ilx_mdp_plan_create: %d index-list entries (max 4096)",
                d.num_ids);
    int rc = validate_terms(d);
    if (rc != ILX_OK) return rc;
    auto* p = new ilx_mdp_plan_s();
    p->nt = d.num_terminations, p->nr = d.num_rewards, p->no = d.num_observations, p->nids = d.num_ids;
    p->obs_dim = d.obs_dim;
    for (int i = 0; i < d.num_terminations; ++i) {
        const int k = d.terminations[i].kind;
        p->t_contact |= k == ILX_TERM_ILLEGAL_CONTACT;
        p->t_joint |= k >= ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT && k <= ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT;
        p->t_cmdstate |= k == ILX_TERM_COMMAND_RESAMPLE;
    }
    for (int i = 0; i < d.num_rewards; ++i) {
        const int k = d.rewards[i].kind;
        if (d.rewards[i].weight == 0.0f) continue;
        p->r_contact |= k == ILX_REW_UNDESIRED_CONTACTS || k == ILX_REW_CONTACT_FORCES || k == ILX_REW_FEET_SLIDE;
        p->r_timers |= k == ILX_REW_FEET_AIR_TIME || k == ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED;
        p->r_body_state |= k == ILX_REW_FEET_SLIDE || (k >= ILX_REW_POSITION_COMMAND_ERROR && k <= ILX_REW_ORIENTATION_COMMAND_ERROR);
        p->r_body_vel |= k == ILX_REW_FEET_SLIDE;
        p->r_body_acc |= k == ILX_REW_BODY_LIN_ACC_L2;
        p->r_joint |= (k >= ILX_REW_JOINT_TORQUES_L2 && k <= ILX_REW_APPLIED_TORQUE_LIMITS) || k == ILX_REW_JOINT_POS_TARGET_L2;
        p->r_action |= k == ILX_REW_ACTION_RATE_L2 || k == ILX_REW_ACTION_L2;
        p->r_rays |= k == ILX_REW_BASE_HEIGHT_L2 && (d.rewards[i].flags & 1);
        p->r_command |= (k >= ILX_REW_TRACK_LIN_VEL_XY_EXP && k <= ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED) ||
                        (k >= ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP && k <= ILX_REW_ORIENTATION_COMMAND_ERROR);
    }
    {   // the reward terms dealt to the block's four wavefronts (mdp_shared.h)
        TermSig sig[kMaxTerms];
        for (int i = 0; i < d.num_rewards; ++i)
            sig[i] = TermSig{d.rewards[i].kind, d.rewards[i].ids_cnt, d.rewards[i].ids2_cnt, d.rewards[i].flags, d.rewards[i].weight != 0.0f};
        p->reward_owner = deal_rewards(sig, d.num_rewards, MdpDims{d.num_joints, d.num_bodies, d.num_sensor_bodies, d.history_len,
                                                                     d.num_rays, d.action_dim, d.command_dim});
    }
    for (int i = 0; i < d.num_observations; ++i) {
        const int k = d.observations[i].kind;
        p->o_joint |= k >= ILX_OBS_JOINT_POS && k <= ILX_OBS_JOINT_VEL_REL;
        p->o_rays |= k == ILX_OBS_HEIGHT_SCAN;
        p->o_action |= k == ILX_OBS_LAST_ACTION;
        p->o_command |= k == ILX_OBS_GENERATED_COMMANDS;
        if (k == ILX_OBS_EXTERNAL) p->o_external |= 1u << static_cast<int>(d.observations[i].aux);
    }
    std::vector<ilx_term> all;
    all.insert(all.end(), d.terminations, d.terminations + d.num_terminations);
    all.insert(all.end(), d.rewards, d.rewards + d.num_rewards);
    all.insert(all.end(), d.observations, d.observations + d.num_observations);
    if (!all.empty()) {
        rc = check_hip(hipMalloc(&p->d_terms, all.size() * sizeof(ilx_term)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_terms, all.data(), all.size() * sizeof(ilx_term), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    if (rc == ILX_OK && d.num_ids > 0) {
        // padded to whole 16-byte words: the staging copies move four at a time
        rc = check_hip(hipMalloc(&p->d_ids, ((d.num_ids + 3) & ~3) * sizeof(int32_t)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_ids, d.ids, d.num_ids * sizeof(int32_t), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    if (rc == ILX_OK && d.num_observations > 0) {
        ILX_REQUIRE(d.obs_dim <= 8192, ILX_ESHAPE, "ilx_mdp_plan_create: %d observation columns (max 8192)", d.obs_dim);
        ILX_REQUIRE(d.num_joints <= 0xffff && d.action_dim <= 0xffff && d.command_dim <= 0xffff, ILX_ESHAPE,
                    "ilx_mdp_plan_create: rows wider than 65535 columns");
        auto field = [](int Layout::*f) {  // word index of a Layout field inside the struct
            Layout probe{};
            return static_cast<uint32_t>(reinterpret_cast<const char*>(&(probe.*f)) - reinterpret_cast<const char*>(&probe)) / 4u;
        };
        std::vector<ColDesc> cols(static_cast<size_t>(d.obs_dim));
        int col = 0;
        for (int i = 0; i < d.num_observations; ++i) {
            const ilx_term& t = d.observations[i];
            const int dim = host_obs_dim(d, t);
            const uint32_t J = static_cast<uint32_t>(d.num_joints);
            for (int k = 0; k < dim; ++k, ++col) {
                ColDesc c{kColGeneric | (static_cast<uint32_t>(i) << 8), static_cast<uint32_t>(k), field(&Layout::zero) << 16, 0u};
                auto affine = [&](int Layout::*row, uint32_t kk, uint32_t stride) {
                    c.mode_term = kColAffine | (static_cast<uint32_t>(i) << 8);
                    c.src = (field(row) << 16) | kk;
                    c.stride = stride;
                };
                auto minus = [&](int Layout::*row, uint32_t kk, uint32_t stride) {
                    c.sub = (field(row) << 16) | kk;
                    c.stride |= stride << 16;
                };
                const uint32_t jk = static_cast<uint32_t>(t.ids_cnt < 0 ? k : d.ids[t.ids_off + k]);
                switch (t.kind) {
                    case ILX_OBS_BASE_POS_Z: affine(&Layout::root, 2u, 13u); break;
                    case ILX_OBS_BASE_LIN_VEL: affine(&Layout::derived, static_cast<uint32_t>(k), kDerived); break;
                    case ILX_OBS_BASE_ANG_VEL: affine(&Layout::derived, 3u + k, kDerived); break;
                    case ILX_OBS_PROJECTED_GRAVITY: affine(&Layout::derived, 6u + k, kDerived); break;
                    case ILX_OBS_ROOT_QUAT_W:
                        if (!(t.flags & ILX_OBS_FLAG_QUAT_UNIQUE)) affine(&Layout::root, 3u + k, 13u);
                        break;
                    case ILX_OBS_ROOT_LIN_VEL_W: affine(&Layout::root, 7u + k, 13u); break;
                    case ILX_OBS_ROOT_ANG_VEL_W: affine(&Layout::root, 10u + k, 13u); break;
                    case ILX_OBS_JOINT_POS: affine(&Layout::q, jk, J); break;
                    case ILX_OBS_JOINT_POS_REL: affine(&Layout::q, jk, J), minus(&Layout::q0, jk, J); break;
                    case ILX_OBS_JOINT_VEL: affine(&Layout::qd, jk, J); break;
                    case ILX_OBS_JOINT_VEL_REL: affine(&Layout::qd, jk, J), minus(&Layout::qd0, jk, J); break;
                    case ILX_OBS_LAST_ACTION: affine(&Layout::act, static_cast<uint32_t>(k), static_cast<uint32_t>(d.action_dim)); break;
                    case ILX_OBS_GENERATED_COMMANDS:
                        affine(&Layout::cmd, static_cast<uint32_t>(k), static_cast<uint32_t>(d.command_dim));
                        break;
                    case ILX_OBS_HEIGHT_SCAN: c.mode_term = kColScan | (static_cast<uint32_t>(i) << 8); break;
                    default: break;  // root position (env origins), unique quaternions, normalised limits, external buffers
                }
                cols[col] = c;
            }
            p->any_noise |= (t.flags & ILX_OBS_FLAG_NOISE) != 0;
            if (t.flags & ILX_OBS_FLAG_NOISE) p->noisy_terms |= 1u << i;
        }
        p->dims[0] = d.num_joints, p->dims[1] = d.action_dim, p->dims[2] = d.command_dim;
        rc = check_hip(hipMalloc(&p->d_cols, cols.size() * sizeof(ColDesc)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_cols, cols.data(), cols.size() * sizeof(ColDesc), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    // host copies of the tables, the neutral post-processing record of every observation column, and the kernel
    // compiled for this term list, if there is one
    for (size_t i = 0; i < all.size(); ++i) p->h_terms[i] = all[i];
    for (int i = 0; i < d.num_ids && i < static_cast<int>(sizeof(p->h_ids) / sizeof(p->h_ids[0])); ++i) p->h_ids[i] = d.ids[i];
    if (rc == ILX_OK && d.num_observations > 0) {
        std::vector<ColParam> cp;
        for (int i = 0; i < d.num_observations; ++i)
            cp.insert(cp.end(), static_cast<size_t>(host_obs_dim(d, d.observations[i])), col_param_of(d.observations[i]));
        cp.resize((cp.size() + 3) & ~size_t(3), ColParam{0.0f, 0.0f, -__builtin_inff(), __builtin_inff(), 1.0f, {0.0f, 0.0f, 0.0f}});
        rc = check_hip(hipMalloc(&p->d_colparam, cp.size() * sizeof(ColParam)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_colparam, cp.data(), cp.size() * sizeof(ColParam), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    if (rc != ILX_OK) {
        ilx_mdp_plan_destroy(p);
        return rc;
    }
    p->fast_kernel = mdp_fast_match(*p, d);
    *out = p;
    return ILX_OK;
}

extern "C" const char* ilx_mdp_plan_kernel(ilx_mdp_plan_t plan) {
    if (!plan) return "";
    return (plan->fast_kernel >= 0 && plan->fast_enabled) ? mdp_fast_name(plan->fast_kernel) : "generic";
}

extern "C" int ilx_mdp_plan_select_kernel(ilx_mdp_plan_t plan, int compiled) {
    ILX_REQUIRE(plan, ILX_EINVAL, "ilx_mdp_plan_select_kernel: plan is NULL");
    plan->fast_enabled = compiled != 0;
    return ILX_OK;
}

extern "C" void ilx_mdp_plan_destroy(ilx_mdp_plan_t plan) {
    if (!plan) return;
    if (plan->d_colparam) (void)hipFree(plan->d_colparam);
    if (plan->d_terms) (void)hipFree(plan->d_terms);
    if (plan->d_ids) (void)hipFree(plan->d_ids);
    if (plan->d_cols) (void)hipFree(plan->d_cols);
    delete plan;
}

#ifdef ILX_MDP_STAMPS
extern "C" int ilx_debug_mdp_hwid(unsigned int* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_mdp_hwid), sizeof(unsigned int) * count) == hipSuccess ? 0 : 1;
}
extern "C" int ilx_debug_mdp_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_mdp_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif

__global__ void mdp_begin_step_kernel(float* log_sums, int count, uint64_t* step_index_device) {
    for (int i = threadIdx.x; i < count; i += blockDim.x) log_sums[i] = 0.0f;
    if (step_index_device && threadIdx.x == 0) *step_index_device += 1;
}

extern "C" int ilx_mdp_begin_step(float* log_sums, int32_t count, uint64_t* step_index_device, void* stream) {
    ILX_REQUIRE(count >= 0 && (count == 0 || log_sums), ILX_EINVAL, "ilx_mdp_begin_step: %d floats at a NULL pointer", count);
    if (count == 0 && !step_index_device) return ILX_OK;
    ILX_LAUNCH(mdp_begin_step_kernel, dim3(1), dim3(64), 0, as_stream(stream), log_sums, log_sums ? count : 0, step_index_device);
    ILX_LAUNCH_CHECK("ilx_mdp_begin_step");
    return ILX_OK;
}

// `scene` (or null): the scene update that precedes the step (ilx_mdp_step_scene_update).
static int mdp_step(ilx_mdp_plan_t plan, const ilx_mdp* desc, void* stream, const ilx_scene* scene);

extern "C" int ilx_mdp_step(ilx_mdp_plan_t plan, const ilx_mdp* desc, void* stream) { return mdp_step(plan, desc, stream, nullptr); }

extern "C" int ilx_mdp_step_scene_update(ilx_mdp_plan_t plan, const ilx_mdp* desc, const ilx_scene* scene, void* stream) {
    ILX_REQUIRE(scene, ILX_EINVAL, "ilx_mdp_step_scene_update: scene is NULL");
    return mdp_step(plan, desc, stream, scene);
}

static int mdp_step(ilx_mdp_plan_t plan, const ilx_mdp* desc, void* stream, const ilx_scene* scene) {
    ILX_REQUIRE(plan && desc, ILX_EINVAL, "ilx_mdp_step: NULL argument");
    const ilx_mdp& d = *desc;
    ILX_REQUIRE(d.num_envs >= 0, ILX_EINVAL, "ilx_mdp_step: negative env count");
    ILX_REQUIRE(d.num_envs < (1ll << 31), ILX_EINVAL, "ilx_mdp_step: more than 2^31 - 1 envs in one launch");
    ILX_REQUIRE(d.num_terminations == plan->nt && d.num_rewards == plan->nr && d.num_observations == plan->no &&
                    d.obs_dim == plan->obs_dim,
                ILX_EINVAL, "ilx_mdp_step: term counts differ from the plan");
    ILX_REQUIRE(plan->no == 0 || (d.num_joints == plan->dims[0] && d.action_dim == plan->dims[1] && d.command_dim == plan->dims[2]),
                ILX_EINVAL, "ilx_mdp_step: joint / action / command widths differ from the plan");
    if (d.num_envs == 0 || (d.phases & ILX_PHASE_ALL) == 0) return scene ? ilx_scene_update(scene, stream) : ILX_OK;
    ILX_REQUIRE(d.root_state_w, ILX_EINVAL, "ilx_mdp_step: root_state_w is NULL");
    ILX_REQUIRE(d.num_joints >= 0 && d.num_bodies >= 0 && d.num_sensor_bodies >= 0 && d.action_dim >= 0 &&
                    d.command_dim >= 0 && d.history_len >= 0 && d.num_rays >= 0,
                ILX_EINVAL, "ilx_mdp_step: negative size");
    if (d.command_kind != ILX_CMD_NONE)
        ILX_REQUIRE(d.command_dim >= (d.command_kind == ILX_CMD_VELOCITY ? 3 : 7), ILX_ESHAPE,
                    "ilx_mdp_step: command_dim %d too small for the command term", d.command_dim);
    if (d.phases & ILX_PHASE_TERMINATION)
        ILX_REQUIRE(d.term_dones && d.terminated && d.time_outs, ILX_EINVAL, "ilx_mdp_step: termination outputs are NULL");
    if (d.phases & ILX_PHASE_REWARD)
        ILX_REQUIRE(d.num_rewards == 0 || (d.reward_buf && d.episode_sums && d.step_reward), ILX_EINVAL,
                    "ilx_mdp_step: reward outputs are NULL");
    if ((d.phases & (ILX_PHASE_COMMAND | ILX_PHASE_RESET)) && d.command_kind != ILX_CMD_NONE)
        ILX_REQUIRE(d.command && d.time_left && d.command_counter && d.command_metrics &&
                        (d.command_kind != ILX_CMD_VELOCITY || (d.heading_target && d.is_heading_env && d.is_standing_env)),
                    ILX_EINVAL, "ilx_mdp_step: command buffers are NULL");
    if (d.phases & ILX_PHASE_OBSERVATION) ILX_REQUIRE(d.obs, ILX_EINVAL, "ilx_mdp_step: obs is NULL");
    // every array a term of an active phase dereferences must be there
    const bool joints = d.joint_pos && d.joint_vel && d.joint_acc && d.applied_torque && d.computed_torque &&
                        d.default_joint_pos && d.default_joint_vel && d.soft_joint_pos_limits && d.soft_joint_vel_limits;
    const bool timers = d.current_air_time && d.last_air_time && d.current_contact_time && d.last_contact_time;
    if (d.phases & ILX_PHASE_TERMINATION) {
        ILX_REQUIRE(!plan->t_contact || d.net_forces_w_history, ILX_EINVAL, "ilx_mdp_step: a termination term needs net_forces_w_history");
        ILX_REQUIRE(!plan->t_joint || joints, ILX_EINVAL, "ilx_mdp_step: a termination term needs the joint state arrays");
        ILX_REQUIRE(!plan->t_cmdstate || (d.time_left && d.command_counter), ILX_EINVAL, "ilx_mdp_step: command_resample needs the command state");
    }
    if (d.phases & ILX_PHASE_REWARD) {
        ILX_REQUIRE(!plan->r_contact || d.net_forces_w_history, ILX_EINVAL, "ilx_mdp_step: a reward term needs net_forces_w_history");
        ILX_REQUIRE(!plan->r_timers || timers, ILX_EINVAL, "ilx_mdp_step: a reward term needs the contact timers");
        ILX_REQUIRE(!plan->r_body_state || d.body_state_w, ILX_EINVAL, "ilx_mdp_step: a reward term needs body_state_w");
        ILX_REQUIRE(!plan->r_body_acc || d.body_lin_acc_w, ILX_EINVAL, "ilx_mdp_step: a reward term needs body_lin_acc_w");
        ILX_REQUIRE(!plan->r_joint || joints, ILX_EINVAL, "ilx_mdp_step: a reward term needs the joint state arrays");
        ILX_REQUIRE(!plan->r_action || (d.action && d.prev_action), ILX_EINVAL, "ilx_mdp_step: a reward term needs action / prev_action");
        ILX_REQUIRE(!plan->r_rays || d.ray_hits_w, ILX_EINVAL, "ilx_mdp_step: a reward term needs ray_hits_w");
        ILX_REQUIRE(!plan->r_command || d.command, ILX_EINVAL, "ilx_mdp_step: a reward term needs the command tensor");
    }
    if (d.phases & ILX_PHASE_OBSERVATION) {
        ILX_REQUIRE(!plan->o_joint || joints, ILX_EINVAL, "ilx_mdp_step: an observation term needs the joint state arrays");
        ILX_REQUIRE(!plan->o_rays || d.height_scan || (d.ray_hits_w && d.sensor_pos_w), ILX_EINVAL,
                    "ilx_mdp_step: height_scan needs ray_hits_w and sensor_pos_w, or the ray caster's height_scan");
        for (int slot = 0; slot < ILX_MAX_EXTERNAL_OBS; ++slot)
            ILX_REQUIRE(!(plan->o_external & (1u << slot)) || d.external_obs[slot], ILX_EINVAL,
                        "ilx_mdp_step: external observation buffer %d is NULL", slot);
        ILX_REQUIRE(!plan->o_action || d.action, ILX_EINVAL, "ilx_mdp_step: last_action needs the action tensor");
        ILX_REQUIRE(!plan->o_command || d.command, ILX_EINVAL, "ilx_mdp_step: generated_commands needs the command tensor");
    }
    if ((d.phases & ILX_PHASE_COMMAND) && d.command_kind == ILX_CMD_POSE)
        ILX_REQUIRE(d.body_state_w && d.pose_body_index >= 0 && d.pose_body_index < d.num_bodies, ILX_EINVAL,
                    "ilx_mdp_step: the pose command needs body_state_w and a valid body index");
    if (plan->fast_kernel >= 0 && plan->fast_enabled) {   // the kernel compiled for this term list takes the whole step
        const int status = mdp_fast_step(*plan, d, stream, scene);
        if (status <= 0) return status;
    }
    if (scene) {   // the plan-interpreting kernel: the update as a launch of its own ahead of it
        const int status = ilx_scene_update(scene, stream);
        if (status != ILX_OK) return status;
    }
    const bool pt = d.phases & ILX_PHASE_TERMINATION, pr = d.phases & ILX_PHASE_REWARD, po = d.phases & ILX_PHASE_OBSERVATION;
    MdpDev m;
    m.d = d;
    m.terms = plan->d_terms;
    m.ids = plan->d_ids;
    m.cols = plan->d_cols;
    m.any_noise = plan->any_noise;
    m.noisy_terms = plan->noisy_terms;
    auto inverse = [](int n) { return n > 1 ? static_cast<uint32_t>((0x100000000ull + n - 1) / n) : 0xffffffffu; };
    m.inv_rewards = inverse(d.num_rewards);
    m.need = 0;
    if ((pt && plan->t_joint) || (pr && plan->r_joint) || (po && plan->o_joint)) m.need |= kNeedJoint;
    if ((pt && plan->t_contact) || (pr && plan->r_contact)) m.need |= kNeedForces;
    if (pr && plan->r_timers) m.need |= kNeedTimers;
    if (pr && plan->r_body_vel) m.need |= kNeedBodyVel;
    if (pr && plan->r_body_acc) m.need |= kNeedBodyAcc;
    if ((pr && plan->r_action) || (po && plan->o_action)) m.need |= kNeedAction;
    if (po && plan->o_rays) m.need |= kNeedRays;
    // LDS image of one block (kE envs)
    Layout& L = m.L;
    int w = static_cast<int>(sizeof(MdpDev) / 4);  // the image starts behind the block's copy of the kernel arguments
    auto take = [&](int words) { const int at = w; w += (words + 3) & ~3; return at; };
    const int J = d.num_joints, B = d.num_bodies, CB = d.num_sensor_bodies, A = d.action_dim, CD = d.command_dim;
    L.terms = take((plan->nt + plan->nr + plan->no) * static_cast<int>(sizeof(ilx_term) / 4));
    L.ids = take(plan->nids);
    const bool nj = m.need & kNeedJoint;
    L.q = take(nj ? kE * J : 0), L.qd = take(nj ? kE * J : 0), L.qdd = take(nj ? kE * J : 0);
    L.tau_a = take(nj ? kE * J : 0), L.tau_c = take(nj ? kE * J : 0), L.q0 = take(nj ? kE * J : 0);
    L.qd0 = take(nj ? kE * J : 0), L.vlim = take(nj ? kE * J : 0), L.lim = take(nj ? 2 * kE * J : 0);
    const bool na = m.need & kNeedAction;
    L.act = take(na ? kE * A : 0), L.pact = take(na ? kE * A : 0);
    const bool ntm = m.need & kNeedTimers;
    L.air = take(ntm ? kE * CB : 0), L.last_air = take(ntm ? kE * CB : 0), L.contact_t = take(ntm ? kE * CB : 0);
    L.fraw = take((m.need & kNeedForces) ? kE * 3 * d.history_len * CB : 0);
    L.bvx = take((m.need & kNeedBodyVel) ? kE * B : 0), L.bvy = take((m.need & kNeedBodyVel) ? kE * B : 0);
    L.bacc = take((m.need & kNeedBodyAcc) ? kE * B : 0);
    L.cmd = take(kE * CD), L.root = take(kE * 13);
    L.eps = take(kE * plan->nr), L.srew = take(kE * plan->nr), L.ptotal = take(kWaves * kE);
    L.derived = take(kE * kDerived), L.bits = take(2 * kE);
    L.logp = take(plan->nr + plan->nt + 3);
    L.coldesc = take(po ? plan->obs_dim * static_cast<int>(sizeof(ColDesc) / 4) : 0);
    L.zero = take(1);
    {   // the resolved column records: over the force history when it is large enough (nothing reads it by then)
        const int rec_words = po ? plan->obs_dim * static_cast<int>(sizeof(ColRec) / 4) : 0;
        const int fraw_words = (m.need & kNeedForces) ? kE * 3 * d.history_len * CB : 0;
        L.colrec = (rec_words > 0 && fraw_words >= rec_words) ? L.fraw : take(rec_words);
    }
    L.total = w;
    const size_t lds_bytes = static_cast<size_t>(w) * 4;
    ILX_REQUIRE(lds_bytes <= kMaxLdsBytes, ILX_ESHAPE,
                "ilx_mdp_step: %zu bytes of per-block state (16 envs x row widths) exceed the %d-byte LDS budget", lds_bytes,
                kMaxLdsBytes);
    m.d.terminations = m.d.rewards = m.d.observations = nullptr;
    m.d.ids = nullptr;
    // STAGE work list (see stage_segments): one record per copy instruction of a full block
    m.num_segs = 0;
    bool fits = true;
    auto add = [&](const void* src, int per_env_words, int table_words, int dst) {
        const int total = per_env_words ? kE * per_env_words : table_words;
        if (!src || total <= 0) return;
        fits = fits && per_env_words <= 0xffff;
        // 16-byte copies: aligned source and destination; a table that is not a multiple of four words long is padded by
        // its owner (the plan's device tables are)
        const bool wide = reinterpret_cast<uintptr_t>(src) % 16 == 0 && dst % 4 == 0;
        const uint64_t addr = reinterpret_cast<uintptr_t>(src);
        const int step = wide ? 256 : 64, unit = wide ? 4 : 1;
        fits = fits && dst + total <= 0xffff && total <= (0xffff << 6);
        for (int first = 0; first < total; first += step) {
            if (m.num_segs == kMaxSegs) {
                fits = false;
                return;
            }
            const uint64_t at = addr + 4u * static_cast<uint64_t>(first);
            const int lanes = ((total - first < step ? total - first : step) + unit - 1) / unit;
            m.segs[m.num_segs++] = uint4{static_cast<uint32_t>(at), static_cast<uint32_t>(at >> 32),
                                         static_cast<uint32_t>(per_env_words) | (static_cast<uint32_t>(lanes) << 16) | (wide ? 1u << 23 : 0u),
                                         static_cast<uint32_t>(dst + first) | (static_cast<uint32_t>(first >> 6) << 16)};
        }
    };
    add(d.root_state_w, 13, 0, L.root);
    if (nj) {
        add(d.joint_pos, J, 0, L.q), add(d.joint_vel, J, 0, L.qd), add(d.joint_acc, J, 0, L.qdd);
        add(d.applied_torque, J, 0, L.tau_a), add(d.computed_torque, J, 0, L.tau_c);
        add(d.default_joint_pos, J, 0, L.q0), add(d.default_joint_vel, J, 0, L.qd0);
        add(d.soft_joint_vel_limits, J, 0, L.vlim), add(d.soft_joint_pos_limits, 2 * J, 0, L.lim);
    }
    if (na) add(d.action, A, 0, L.act), add(d.prev_action, A, 0, L.pact);
    if (ntm) {
        add(d.current_air_time, CB, 0, L.air), add(d.last_air_time, CB, 0, L.last_air);
        add(d.current_contact_time, CB, 0, L.contact_t);
    }
    add(d.command, CD, 0, L.cmd);
    if (m.need & kNeedForces) add(d.net_forces_w_history, 3 * d.history_len * CB, 0, L.fraw);
    if (pr || (d.phases & ILX_PHASE_RESET))  // episode sums: [term][env] in memory and in LDS, one short run per term
        for (int i = 0; i < plan->nr; ++i) add(d.episode_sums + static_cast<int64_t>(i) * d.num_envs, 1, 0, L.eps + i * kE);
    add(plan->d_terms, 0, (plan->nt + plan->nr + plan->no) * static_cast<int>(sizeof(ilx_term) / 4), L.terms);
    add(plan->d_ids, 0, plan->nids, L.ids);
    if (po) add(plan->d_cols, 0, plan->obs_dim * static_cast<int>(sizeof(ColDesc) / 4), L.coldesc);
    ILX_REQUIRE(fits, ILX_ESHAPE, "ilx_mdp_step: the plan's per-block state needs more than %d staging copies", kMaxSegs);
    const unsigned blocks = static_cast<unsigned>((d.num_envs + kE - 1) / kE);
    Hot hot;
    hot.L = m.L;
    hot.num_joints = J, hot.num_bodies = B, hot.num_sensor_bodies = CB, hot.command_dim = CD, hot.action_dim = A;
    hot.num_rays = d.num_rays, hot.command_kind = d.command_kind;
    hot.num_terminations = plan->nt, hot.num_rewards = plan->nr, hot.obs_dim = plan->obs_dim;
    hot.history_len = d.history_len, hot.num_segs = m.num_segs, hot.dense_scan = d.height_scan != nullptr;
    hot.phases = d.phases, hot.need = m.need;
    hot.inv_rewards = m.inv_rewards;
    hot.step_dt = d.step_dt;
    for (int k = 0; k < 3; ++k) hot.gravity[k] = d.gravity_dir[k];
    hot.num_envs = d.num_envs, hot.seed = d.seed, hot.step_index = d.step_index;
    hot.terms = plan->d_terms, hot.step_index_device = d.step_index_device, hot.reward_owner = plan->reward_owner;
    ILX_LAUNCH(mdp_step_kernel, dim3(blocks), dim3(kBlock), lds_bytes, as_stream(stream), m, hot);
    ILX_LAUNCH_CHECK("ilx_mdp_step");
    return ILX_OK;
}
