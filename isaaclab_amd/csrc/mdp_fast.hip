// Fused MDP step, compiled per term list.
//
// ilx_mdp_step's plan-interpreting kernel (mdp.hip) serves any term list; here the same step is compiled for the term
// lists of the reference's own task configurations, with every size and every term kind a compile-time constant: the
// term formulas of mdp_shared.h inline into straight-line code, no term records are fetched, no kind is dispatched on.
// The work is cut differently, too:
//   DECIDE blocks   320 threads own 64 consecutive envs.  Each of the five wavefronts has ONE LANE PER ENV (all 64 lanes
//                   busy).  Wavefronts 0..3 evaluate, for their env, the terminations plus their share of the reward
//                   terms -- the same dealing of terms to wavefronts as the plan-interpreting kernel, so the
//                   per-wavefront partial sums and with them every bit of the reward are the same.  The fifth wavefront
//                   runs the reset bookkeeping and the command term (the longest dependent chain of the step).  A lane
//                   reads its env's rows straight from memory (a wavefront's 64 rows are one contiguous range; the loads
//                   of one row are issued together).  The non-scan observation columns are put into LDS by the lanes
//                   that hold their values, then all 320 threads add noise / clip / scale and store them, one lane per
//                   group of four columns (one Philox4x32 call, keyed as in mdp.hip, drawn while the loads are in flight).
//   SCAN blocks     the height-scan columns depend on nothing the DECIDE blocks compute: separate blocks of the same
//                   grid stream them from the ray caster's dense height scan to the observation rows (noise, clip),
//                   one lane per (env, group of four columns), beside the DECIDE blocks.
// State that is updated in place and read by several wavefronts of a block (episode length, command, action history) is
// stored only after the block's barrier.
#include "mdp_shared.h"

#include <cstring>
#include <type_traits>
#include <utility>

namespace ilx {

constexpr int kFastEnvs = 64;    // envs per DECIDE block
constexpr int kScanEnvs = 34;    // envs per SCAN block
constexpr int kCmdWave = kWaves; // the DECIDE block's fifth wavefront: reset bookkeeping and the command term
constexpr int kDrawWave = 3;     // the reward wavefront that draws the new pose commands beside it
constexpr int kFastBlock = 64 * (kWaves + 1);
constexpr int kFastMaxIds = 64;
constexpr int kFastMaxTerms = 24;   // terminations + rewards of a compiled term list

typedef const __attribute__((address_space(1))) char GBytes;
typedef __attribute__((address_space(1))) char GBytesW;

// A uniform address pinned to scalar registers (so that the lane's 32-bit offset stays a separate operand of the memory
// instruction).  Not volatile: a volatile statement would fence the instruction scheduler, and a wavefront's loads are
// meant to travel together.
ILX_DEV GBytes* pinned(GBytes* p) {
    uint64_t a = reinterpret_cast<uint64_t>(p);
    asm("" : "+s"(a));
    return (GBytes*)a;
}
template <typename Tp>
ILX_DEV Tp ldg(const Tp* base, int64_t first_elem, uint32_t voff) {   // base[first_elem] + lane's byte offset
    return *reinterpret_cast<const __attribute__((address_space(1))) Tp*>(pinned((GBytes*)base + static_cast<int64_t>(sizeof(Tp)) * first_elem) + voff);
}
template <typename Tp, typename V>
ILX_DEV void stg(Tp* base, int64_t first_elem, uint32_t voff, V v) {
    *reinterpret_cast<__attribute__((address_space(1))) Tp*>((GBytesW*)pinned((GBytes*)base + static_cast<int64_t>(sizeof(Tp)) * first_elem) + voff) = static_cast<Tp>(v);
}

struct FastArgs {
    ilx_mdp d;                          // by value; the host pointers inside are nulled
    ilx_term terms[kFastMaxTerms];      // [terminations | rewards]: weights and parameters
    int32_t ids[kFastMaxIds];
    const ColParam* colparam;           // one per observation column
    uint32_t decide_blocks, scan_blocks;
    SceneArgs scene;                    // the scene update whose workgroups follow the step's in the grid (scene_blocks of them)
    uint32_t scene_blocks;
};

#ifdef ILX_FAST_STAMPS  // diagnostics build only: wall-clock stamps (10 ns ticks) of lane 0 of every wavefront of the first blocks
__device__ unsigned long long g_fast_stamps[512 * 5 * 8];
__device__ unsigned long long g_scan_stamps[2048 * 8];
#define SCAN_STAMP(k)                                                                                                   \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (threadIdx.x == 0 && block < 2048) g_scan_stamps[block * 8 + (k)] = wall_clock64();                         \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define FAST_STAMP(k)                                                                                                   \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) g_fast_stamps[(blockIdx.x * 5 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define FAST_STAMP(k)
#define SCAN_STAMP(k)
#endif

// ---- compile-time loops -----------------------------------------------------------------------------
template <int... I, class F>
ILX_DEV void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
ILX_DEV void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ---- term lists ---------------------------------------------------------------------------------------
// Sizes, the three term tables, and the command term's kind.  Parameters, weights (apart from "is zero") and index
// values are not part of a list: they come with the plan.
constexpr TermSig T(int kind, int ids_cnt = -1, int ids2_cnt = -1, int flags = 0, bool active = true) {
    return TermSig{kind, ids_cnt, ids2_cnt, flags, active};
}

// velocity_env_cfg.py:119-265 with config/anymal_c/rough_env_cfg.py (ANYmal-C, rough terrain)
struct AnymalRough {
    static constexpr const char* name = "anymal_c_rough";
    static constexpr int kWavesPerSimd = 5;
    static constexpr int kBatch = 2;
    static constexpr bool kContactsAhead = false;
    static constexpr bool kSceneBeside = false;   // (the scene update beside the managers: lists without sensors that do not read joint_acc)
    static constexpr int J = 12, B = 17, CB = 17, H = 3, A = 12, CD = 3, R = 187, cmd_kind = ILX_CMD_VELOCITY;
    static constexpr int NT = 2, NR = 11, NO = 8;
    static constexpr TermSig term[NT] = {T(ILX_TERM_TIME_OUT, -1, -1, 1), T(ILX_TERM_ILLEGAL_CONTACT, 1)};
    static constexpr TermSig rew[NR] = {
        T(ILX_REW_TRACK_LIN_VEL_XY_EXP), T(ILX_REW_TRACK_ANG_VEL_Z_EXP), T(ILX_REW_LIN_VEL_Z_L2), T(ILX_REW_ANG_VEL_XY_L2),
        T(ILX_REW_JOINT_TORQUES_L2), T(ILX_REW_JOINT_ACC_L2), T(ILX_REW_ACTION_RATE_L2), T(ILX_REW_FEET_AIR_TIME, 4),
        T(ILX_REW_UNDESIRED_CONTACTS, 4), T(ILX_REW_FLAT_ORIENTATION_L2, -1, -1, 0, false),
        T(ILX_REW_JOINT_POS_LIMITS, -1, -1, 0, false)};
    static constexpr TermSig obs[NO] = {T(ILX_OBS_BASE_LIN_VEL), T(ILX_OBS_BASE_ANG_VEL), T(ILX_OBS_PROJECTED_GRAVITY),
                                        T(ILX_OBS_GENERATED_COMMANDS), T(ILX_OBS_JOINT_POS_REL), T(ILX_OBS_JOINT_VEL_REL),
                                        T(ILX_OBS_LAST_ACTION), T(ILX_OBS_HEIGHT_SCAN)};
};

// config/h1/rough_env_cfg.py:19-115 over velocity_env_cfg.py (Unitree H1, rough terrain)
struct H1Rough {
    static constexpr const char* name = "h1_rough";
    static constexpr int kWavesPerSimd = 3;
    static constexpr int kBatch = 4;
    static constexpr bool kContactsAhead = true;
    static constexpr bool kSceneBeside = false;
    static constexpr int J = 19, B = 20, CB = 20, H = 3, A = 19, CD = 3, R = 187, cmd_kind = ILX_CMD_VELOCITY;
    static constexpr int NT = 2, NR = 14, NO = 8;
    static constexpr TermSig term[NT] = {T(ILX_TERM_TIME_OUT, -1, -1, 1), T(ILX_TERM_ILLEGAL_CONTACT, 1)};
    static constexpr TermSig rew[NR] = {
        T(ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP), T(ILX_REW_TRACK_ANG_VEL_Z_WORLD_EXP), T(ILX_REW_ANG_VEL_XY_L2),
        T(ILX_REW_JOINT_TORQUES_L2, -1, -1, 0, false), T(ILX_REW_JOINT_ACC_L2), T(ILX_REW_ACTION_RATE_L2),
        T(ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED, 2), T(ILX_REW_FLAT_ORIENTATION_L2), T(ILX_REW_JOINT_POS_LIMITS, 2),
        T(ILX_REW_IS_TERMINATED), T(ILX_REW_FEET_SLIDE, 2, 2), T(ILX_REW_JOINT_DEVIATION_L1, 4), T(ILX_REW_JOINT_DEVIATION_L1, 8),
        T(ILX_REW_JOINT_DEVIATION_L1, 1)};
    static constexpr TermSig obs[NO] = {T(ILX_OBS_BASE_LIN_VEL), T(ILX_OBS_BASE_ANG_VEL), T(ILX_OBS_PROJECTED_GRAVITY),
                                        T(ILX_OBS_GENERATED_COMMANDS), T(ILX_OBS_JOINT_POS_REL), T(ILX_OBS_JOINT_VEL_REL),
                                        T(ILX_OBS_LAST_ACTION), T(ILX_OBS_HEIGHT_SCAN)};
};

// classic/cartpole/cartpole_env_cfg.py:58-181
struct Cartpole {
    static constexpr const char* name = "cartpole";
    static constexpr int kWavesPerSimd = 4;
    static constexpr int kBatch = 4;
    static constexpr bool kContactsAhead = true;
    static constexpr bool kSceneBeside = true;
    static constexpr int J = 2, B = 3, CB = 0, H = 0, A = 1, CD = 0, R = 0, cmd_kind = ILX_CMD_NONE;
    static constexpr int NT = 2, NR = 5, NO = 2;
    static constexpr TermSig term[NT] = {T(ILX_TERM_TIME_OUT, -1, -1, 1), T(ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT, 1)};
    static constexpr TermSig rew[NR] = {T(ILX_REW_IS_ALIVE), T(ILX_REW_IS_TERMINATED), T(ILX_REW_JOINT_POS_TARGET_L2, 1),
                                        T(ILX_REW_JOINT_VEL_L1, 1), T(ILX_REW_JOINT_VEL_L1, 1)};
    static constexpr TermSig obs[NO] = {T(ILX_OBS_JOINT_POS_REL), T(ILX_OBS_JOINT_VEL_REL)};
};

// manipulation/reach/reach_env_cfg.py:65-208 for an arm of J_ joints, B_ bodies and an A_-wide arm action (Franka: 9 / 11
// with the joint-position or absolute-pose action 7 wide, the relative-pose action 6 wide; xArm6 with gripper: 8 / 10)
template <int J_, int B_, int A_>
struct Reach {
    static constexpr const char* name = J_ == 9 ? (A_ == 7 ? "reach_j9_a7" : "reach_j9_a6") : (A_ == 7 ? "reach_j8_a7" : "reach_j8_a6");
    static constexpr int kWavesPerSimd = 4;
    static constexpr int kBatch = 4;
    static constexpr bool kContactsAhead = true;
    static constexpr bool kSceneBeside = true;
    static constexpr int J = J_, B = B_, CB = 0, H = 0, A = A_, CD = 7, R = 0, cmd_kind = ILX_CMD_POSE;
    static constexpr int NT = 1, NR = 5, NO = 4;
    static constexpr TermSig term[NT] = {T(ILX_TERM_TIME_OUT, -1, -1, 1)};
    static constexpr TermSig rew[NR] = {T(ILX_REW_POSITION_COMMAND_ERROR, 1), T(ILX_REW_POSITION_COMMAND_ERROR_TANH, 1),
                                        T(ILX_REW_ORIENTATION_COMMAND_ERROR, 1), T(ILX_REW_ACTION_RATE_L2), T(ILX_REW_JOINT_VEL_L2)};
    static constexpr TermSig obs[NO] = {T(ILX_OBS_JOINT_POS_REL), T(ILX_OBS_JOINT_VEL_REL), T(ILX_OBS_GENERATED_COMMANDS),
                                        T(ILX_OBS_LAST_ACTION)};
};

// ---- what follows from a term list ------------------------------------------------------------------------
template <class S>
struct Derived {
    static constexpr MdpDims dims{S::J, S::B, S::CB, S::H, S::R, S::A, S::CD};
    static constexpr uint64_t owner = deal_rewards(S::rew, S::NR, dims);
    static constexpr int reward_owner(int i) { return static_cast<int>((owner >> (2 * i)) & 3u); }
    // the wavefront that carries a reward term's episode sum (active terms: their owner)
    static constexpr int sum_owner(int i) { return S::rew[i].active ? reward_owner(i) : i % kWaves; }
    // index lists are packed in table order: terminations, rewards, observations; per term ids, then ids2
    static constexpr int listed(const TermSig& t) { return (t.ids_cnt > 0 ? t.ids_cnt : 0) + (t.ids2_cnt > 0 ? t.ids2_cnt : 0); }
    static constexpr int term_ids_off(int i) {
        int off = 0;
        for (int k = 0; k < i; ++k) off += listed(S::term[k]);
        return off;
    }
    static constexpr int rew_ids_off(int i) {
        int off = term_ids_off(S::NT);
        for (int k = 0; k < i; ++k) off += listed(S::rew[k]);
        return off;
    }
    static constexpr int num_ids = rew_ids_off(S::NR);   // observation terms of a compiled list name every joint
    static constexpr int obs_width(int i) {
        switch (S::obs[i].kind) {
            case ILX_OBS_BASE_POS_Z: return 1;
            case ILX_OBS_ROOT_QUAT_W: return 4;
            case ILX_OBS_JOINT_POS: case ILX_OBS_JOINT_POS_REL: case ILX_OBS_JOINT_VEL: case ILX_OBS_JOINT_VEL_REL: return S::J;
            case ILX_OBS_HEIGHT_SCAN: return S::R;
            case ILX_OBS_LAST_ACTION: return S::A;
            case ILX_OBS_GENERATED_COMMANDS: return S::CD;
            default: return 3;
        }
    }
    static constexpr int obs_col(int i) {
        int c = 0;
        for (int k = 0; k < i; ++k) c += obs_width(k);
        return c;
    }
    static constexpr int O = obs_col(S::NO);
    static constexpr bool has_scan = S::obs[S::NO - 1].kind == ILX_OBS_HEIGHT_SCAN;   // a height scan is a list's last term
    static constexpr int scan_col = has_scan ? obs_col(S::NO - 1) : O;
    static constexpr int groups = (O + 3) / 4;
    static constexpr int p_groups = has_scan ? (scan_col + 3) / 4 : groups;   // groups of four columns that go through LDS
    static constexpr int scan_groups = groups - p_groups;                      // groups that hold scan columns only
    static constexpr int scan_head = has_scan ? 4 * p_groups - scan_col : 0;   // scan columns inside the last LDS group
    static constexpr int p_width = 4 * p_groups;
    static constexpr int p_stride = p_width | 1;   // odd: a wavefront's 64 rows fall into different banks
    static constexpr int obs_owner(int i) {  // the wavefront whose lanes put the term's values into LDS
        switch (S::obs[i].kind) {
            case ILX_OBS_JOINT_POS: case ILX_OBS_JOINT_POS_REL: return 1;
            case ILX_OBS_JOINT_VEL: case ILX_OBS_JOINT_VEL_REL: return 2;
            case ILX_OBS_LAST_ACTION: case ILX_OBS_HEIGHT_SCAN: return 3;
            case ILX_OBS_GENERATED_COMMANDS: return kCmdWave;   // the wavefront that has just updated the command
            default: return 0;   // root-frame quantities
        }
    }
    static constexpr bool reads(int kind) {
        for (int i = 0; i < S::NR; ++i)
            if (S::rew[i].active && S::rew[i].kind == kind) return true;
        return false;
    }
    static constexpr bool term_is(int kind) {
        for (int i = 0; i < S::NT; ++i)
            if (S::term[i].kind == kind) return true;
        return false;
    }
    static constexpr bool needs_action = S::A > 0;
};

// ---- one environment, read straight from memory -----------------------------------------------------------
// The accessor type of mdp_shared.h for a lane that owns env slot `le` of the block starting at env0.  Every array is
// addressed as a uniform base (advanced to the block's first row with scalar arithmetic) plus the lane's 32-bit byte
// offset into the block's rows; sizes are compile-time, so consecutive elements of a row share one address and the
// compiler fetches them with wide loads.
template <class S>
struct FastEnv {
    const FastArgs& a;
    int64_t env;     // the lane's env
    uint32_t le;     // its slot in the block
    GBytes *b_q, *b_qd, *b_qdd, *b_tau_a, *b_tau_c, *b_q0, *b_qd0, *b_vlim, *b_lim, *b_air, *b_last_air, *b_contact_t, *b_body,
        *b_bacc, *b_act, *b_pact, *b_cmd, *b_force, *b_hits;
    f3 pos, lin_w, ang_w, lin_b, ang_b, grav_b;
    f4 quat;
    float time_left = 0.0f;
    int64_t command_counter = 0;

    template <typename Tp>
    static ILX_DEV GBytes* block_base(const Tp* p, int64_t first_elem) {
        return pinned((GBytes*)p + static_cast<int64_t>(sizeof(Tp)) * first_elem);
    }
    static ILX_DEV float ldf(GBytes* base, uint32_t voff) { return *reinterpret_cast<const __attribute__((address_space(1))) float*>(base + voff); }

    ILX_DEV FastEnv(const FastArgs& args, int64_t env0, uint32_t slot) : a(args), env(env0 + slot), le(slot) {
        const ilx_mdp& d = a.d;
        b_q = block_base(d.joint_pos, env0 * S::J), b_qd = block_base(d.joint_vel, env0 * S::J);
        b_qdd = block_base(d.joint_acc, env0 * S::J), b_tau_a = block_base(d.applied_torque, env0 * S::J);
        b_tau_c = block_base(d.computed_torque, env0 * S::J), b_q0 = block_base(d.default_joint_pos, env0 * S::J);
        b_qd0 = block_base(d.default_joint_vel, env0 * S::J), b_vlim = block_base(d.soft_joint_vel_limits, env0 * S::J);
        b_lim = block_base(d.soft_joint_pos_limits, env0 * S::J * 2);
        b_air = block_base(d.current_air_time, env0 * S::CB), b_last_air = block_base(d.last_air_time, env0 * S::CB);
        b_contact_t = block_base(d.current_contact_time, env0 * S::CB);
        b_body = block_base(d.body_state_w, env0 * S::B * 13), b_bacc = block_base(d.body_lin_acc_w, env0 * S::B * 3);
        b_act = block_base(d.action, env0 * S::A), b_pact = block_base(d.prev_action, env0 * S::A);
        b_cmd = block_base(d.command, env0 * S::CD);
        b_force = block_base(d.net_forces_w_history, env0 * S::H * S::CB * 3);
        b_hits = block_base(d.ray_hits_w, env0 * S::R * 3);
        GBytes* root = block_base(d.root_state_w, env0 * 13);
        const uint32_t v = le * 52u;
        pos = {ldf(root, v), ldf(root + 4, v), ldf(root + 8, v)};
        quat = {ldf(root + 12, v), ldf(root + 16, v), ldf(root + 20, v), ldf(root + 24, v)};
        lin_w = {ldf(root + 28, v), ldf(root + 32, v), ldf(root + 36, v)};
        ang_w = {ldf(root + 40, v), ldf(root + 44, v), ldf(root + 48, v)};
    }
    // The root-frame quantities (articulation_data.py:411-425, 502-518): the first use of what the constructor asked for,
    // so whatever else the wavefront can ask for without them is asked for between the two.
    ILX_DEV void derive() {
        const ilx_mdp& d = a.d;
        lin_b = quat_rotate_inverse(quat, lin_w);
        ang_b = quat_rotate_inverse(quat, ang_w);
        grav_b = quat_rotate_inverse(quat, f3{d.gravity_dir[0], d.gravity_dir[1], d.gravity_dir[2]});
    }

    ILX_DEV const ilx_mdp& desc() const { return a.d; }
    ILX_DEV int id(const ilx_term& t, int k) const { return t.ids_cnt < 0 ? k : a.ids[t.ids_off + k]; }
    ILX_DEV int id2(const ilx_term& t, int k) const { return t.ids2_cnt < 0 ? k : a.ids[t.ids2_off + k]; }
    ILX_DEV int jcount(const ilx_term& t) const { return t.ids_cnt < 0 ? S::J : t.ids_cnt; }
    ILX_DEV int bcount_sensor(const ilx_term& t) const { return t.ids_cnt < 0 ? S::CB : t.ids_cnt; }
    ILX_DEV int bcount_asset(const ilx_term& t) const { return t.ids_cnt < 0 ? S::B : t.ids_cnt; }
    ILX_DEV int num_actions() const { return S::A; }
    ILX_DEV int num_rays() const { return S::R; }
    ILX_DEV GBytes* joint_base(Fld f) const {
        switch (f) {
            case Fld::q: return b_q;
            case Fld::qd: return b_qd;
            case Fld::qdd: return b_qdd;
            case Fld::tau_a: return b_tau_a;
            case Fld::tau_c: return b_tau_c;
            case Fld::q0: return b_q0;
            case Fld::qd0: return b_qd0;
            default: return b_vlim;
        }
    }
    ILX_DEV float joint(Fld f, int j) const { return ldf(joint_base(f) + 4 * j, le * (4u * S::J)); }
    ILX_DEV float limit(int j, int side) const { return ldf(b_lim + 4 * (2 * j + side), le * (8u * S::J)); }
    ILX_DEV float sbody(Fld f, int b) const {
        GBytes* base = f == Fld::air ? b_air : (f == Fld::last_air ? b_last_air : b_contact_t);
        return ldf(base + 4 * b, le * (4u * S::CB));
    }
    ILX_DEV float abody(Fld f, int b) const {
        if (f == Fld::bacc) {  // |a| of the body, from the 3-vector as it lies in memory
            const uint32_t v = le * (12u * S::B);
            return norm3(ldf(b_bacc + 12 * b, v), ldf(b_bacc + 12 * b + 4, v), ldf(b_bacc + 12 * b + 8, v));
        }
        return ldf(b_body + 52 * b + (f == Fld::bvx ? 28 : 32), le * (52u * S::B));  // linear velocity x / y of the 13-wide body state
    }
    ILX_DEV float action(int k) const { return ldf(b_act + 4 * k, le * (4u * S::A)); }
    ILX_DEV float prev_action(int k) const { return ldf(b_pact + 4 * k, le * (4u * S::A)); }
    ILX_DEV float cmd(int k) const { return ldf(b_cmd + 4 * k, le * (4u * S::CD)); }
    ILX_DEV const __attribute__((address_space(1))) float* body_row(int b) const {
        return reinterpret_cast<const __attribute__((address_space(1))) float*>(b_body + 52 * b + le * (52u * S::B));
    }
    ILX_DEV float ray_hit_z(int k) const { return ldf(b_hits + 12 * k + 8, le * (12u * S::R)); }
    // max over the history of |F| of one sensor body: as Env::force_peak of mdp.hip, the same operations on the same
    // values; the history's entries are requested (force_raw) apart from their use (force_peak_of)
    static constexpr bool kReadsMemory = true;
    static constexpr int kBatch = S::kBatch;   // bodies whose values are in flight together (registers: 3 H + 2 floats each)
    struct ForceRaw {
        float f[S::H > 0 ? S::H : 1][3];
    };
    ILX_DEV ForceRaw force_raw(int b) const {
        const uint32_t v = le * (12u * S::H * S::CB);
        ForceRaw r;
#pragma unroll
        for (int k = 0; k < S::H; ++k) {
            GBytes* f = b_force + 12 * (k * S::CB + b);
            r.f[k][0] = ldf(f, v), r.f[k][1] = ldf(f + 4, v), r.f[k][2] = ldf(f + 8, v);
        }
        return r;
    }
    ILX_DEV float force_peak_of(const ForceRaw& r) const {
        float mx = -1.0f;
#pragma unroll
        for (int k = 0; k < S::H; ++k) mx = fmaxf(mx, fmaf_(r.f[k][2], r.f[k][2], fmaf_(r.f[k][1], r.f[k][1], r.f[k][0] * r.f[k][0])));
        return mx < 0.0f ? -__builtin_inff() : sqrt_(mx);
    }
};

// A term of the compiled list: parameters from the kernel arguments, everything else a constant.
template <class S, int I>
ILX_DEV ilx_term termination_of(const FastArgs& a) {
    constexpr TermSig sig = S::term[I];
    constexpr int off = Derived<S>::term_ids_off(I), off2 = off + (sig.ids_cnt > 0 ? sig.ids_cnt : 0);
    ilx_term t = a.terms[I];
    t.kind = sig.kind, t.flags = sig.flags, t.ids_cnt = sig.ids_cnt, t.ids2_cnt = sig.ids2_cnt, t.ids_off = off, t.ids2_off = off2;
    return t;
}
template <class S, int I>
ILX_DEV ilx_term reward_of(const FastArgs& a) {
    constexpr TermSig sig = S::rew[I];
    constexpr int off = Derived<S>::rew_ids_off(I), off2 = off + (sig.ids_cnt > 0 ? sig.ids_cnt : 0);
    ilx_term t = a.terms[S::NT + I];
    t.kind = sig.kind, t.flags = sig.flags, t.ids_cnt = sig.ids_cnt, t.ids2_cnt = sig.ids2_cnt, t.ids_off = off, t.ids2_off = off2;
    return t;
}

constexpr int kFastMaxGroups = 32;   // groups of four observation columns that a DECIDE block passes through LDS, at most
struct FastShared {   // the LDS of a DECIDE block apart from the observation values
    alignas(16) float colp[5][4 * kFastMaxGroups];   // those columns' post-processing parameters: n_off, n_rng, c_lo, c_hi, scale
    float ptotal[kWaves][kFastEnvs];
    float logp[2 * kMaxTerms + 3];
    uint32_t done[kFastEnvs];
    float drawn[2][8][kFastEnvs];   // pose command: a lane's new command (time left, then the pose), first and second draw
};

// One wavefront of a DECIDE block: lane = env.  Everything up to the barrier only reads memory (and writes outputs that
// nothing in this launch reads); the in-place updates follow it.
// A thread's place in the block's last stage: one group of four observation columns, every kPerTrip-th env.
template <class S>
struct EmitLane {
    static constexpr int GP = Derived<S>::p_groups, O = Derived<S>::O;
    static constexpr int kPerTrip = kFastBlock / GP, kTrips = (kFastEnvs + kPerTrip - 1) / kPerTrip;
    int esub, g;
    bool active, noisy;
    float n_rng[4];       // the noise ranges of the group's columns (0: none)
    float u[kTrips][4];   // the group's noise draws per trip
};

template <class S, int W, class Emit>
ILX_DEV void decide(const FastArgs& a, int64_t env0, int ne, uint32_t lane, uint64_t seed, uint64_t steps_dev, float* P, FastShared& sh,
                    EmitLane<S>& em, Emit&& emit) {
    using D = Derived<S>;
    const ilx_mdp& d = a.d;
    const int64_t N = d.num_envs;
    const bool valid = static_cast<int>(lane) < ne;
    const uint32_t le = valid ? lane : static_cast<uint32_t>(ne - 1);   // lanes past the launch's end shadow its last env and store nothing
    // the command wavefront (the one with time to spare) fills the block's table of column parameters in LDS: a lane asks
    // for the whole records of columns lane, lane + 64, ...
    constexpr bool kFillsTable = W == kCmdWave;
    constexpr int kTableTrips = (4 * EmitLane<S>::GP + 63) / 64;
    ColParam column[kTableTrips];
    if constexpr (kFillsTable) {
#pragma unroll
        for (int k = 0; k < kTableTrips; ++k) {
            const int col = 64 * k + static_cast<int>(lane);
            column[k] = a.colparam[col < EmitLane<S>::O ? col : EmitLane<S>::O - 1];
        }
    }
    FastEnv<S> en(a, env0, le);
    // Everything the wavefront can ask for before it knows anything: requested here, in one trip to memory with the root
    // state (the code below is many basic blocks, and a load is not moved from one to another).
    const int64_t env = en.env;
    const float dt = d.step_dt;
    int64_t ep_len = ldg(d.episode_length_buf, env0, 8u * le);
    CmdState cs;
    constexpr bool has_cmd = S::cmd_kind != ILX_CMD_NONE;
    constexpr bool velocity = S::cmd_kind == ILX_CMD_VELOCITY;
    if constexpr (has_cmd && (W == kCmdWave || D::term_is(ILX_TERM_COMMAND_RESAMPLE))) {
        cs.time_left = ldg(d.time_left, env0, 4u * le);
        cs.counter = ldg(d.command_counter, env0, 8u * le);
        en.time_left = cs.time_left, en.command_counter = cs.counter;
    }
    if constexpr (has_cmd && W == kCmdWave) {
        cs.metric0 = ldg(d.command_metrics, env0, 4u * le);
        cs.metric1 = ldg(d.command_metrics, N + env0, 4u * le);
        if constexpr (velocity) {
            cs.heading_target = ldg(d.heading_target, env0, 4u * le);
            cs.is_heading = ldg(d.is_heading_env, env0, le) != 0;
            cs.is_standing = ldg(d.is_standing_env, env0, le) != 0;
        }
#pragma unroll
        for (int k = 0; k < S::CD; ++k) cs.c[k] = en.cmd(k);
    }
    // the pose command's tracked body (pose_command.py:92-108): requested with the wavefront's other loads; and the
    // generator's output for both of the step's resampling streams, made while those loads travel (which lanes will use
    // which stream is known only behind the terminations)
    // The pose command is dealt to two wavefronts: the command wavefront keeps the bookkeeping and the metrics, one of
    // the reward wavefronts draws the new commands of the lanes that resample (two generator calls, three sine / cosine
    // pairs) and hands them over through LDS at the block's barrier.
    constexpr bool pose = has_cmd && !velocity;
    float tracked[7] = {};
    CommandBits bits_reset{}, bits_step{};
    float draw_time_left = 0.0f;
    if constexpr (pose && W == kCmdWave) {
        const auto body = en.body_row(d.pose_body_index);
#pragma unroll
        for (int k = 0; k < 7; ++k) tracked[k] = body[k];
    }
    if constexpr (pose && W == kDrawWave) draw_time_left = ldg(d.time_left, env0, 4u * le);
    float sums[S::NR] = {};
    static_for<S::NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (D::sum_owner(i) == W) sums[i] = ldg(d.episode_sums, static_cast<int64_t>(i) * N + env0, 4u * le);
    });
    // ... and the contact histories the terminations look at (terminations.py:150-158): a few bodies, read by every wavefront
    constexpr int kContactBodies = [] {
        int n = 0;
        for (int i = 0; i < S::NT; ++i)
            if (S::term[i].kind == ILX_TERM_ILLEGAL_CONTACT) n += S::term[i].ids_cnt < 0 ? S::CB : S::term[i].ids_cnt;
        return n;
    }();
    constexpr bool kContactsAhead = S::kContactsAhead && kContactBodies > 0 && kContactBodies <= 2;   // (where the registers are there)
    typename FastEnv<S>::ForceRaw contact_raw[kContactsAhead ? kContactBodies : 1];
    if constexpr (kContactsAhead) {
        int at = 0;
        static_for<S::NT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (S::term[i].kind == ILX_TERM_ILLEGAL_CONTACT) {
                const ilx_term t = termination_of<S, i>(a);
                constexpr int n = S::term[i].ids_cnt < 0 ? S::CB : S::term[i].ids_cnt;
#pragma unroll
                for (int k = 0; k < n; ++k) contact_raw[at + k] = en.force_raw(en.id(t, k));
                at += n;
            }
        });
    }
    // The generator's step word: the argument plus the caller's device-side count, which the kernel asked for before any
    // other load (so it is here before them).
    const uint64_t gen_step = d.step_index + uni(steps_dev);
    if constexpr (pose && W == kDrawWave) {
        bits_reset = command_bits(seed, gen_step, env, 16u);
        bits_step = command_bits(seed, gen_step, env, 0u);
    }
    en.derive();
    // The columns' records go to the block's table (they were asked for before the env's state, so they are here): the
    // block's last stage reads them from LDS.  Kept in registers from the kernel's entry to that stage they would cost
    // twenty registers, and the wait for them there -- the compiler cannot see that they arrived long before -- would stand
    // behind the stores issued in between (loads and stores share one in-order counter): a wait for those stores' trip to
    // memory, once per pass over the block's envs.
    if constexpr (kFillsTable) {
#pragma unroll
        for (int k = 0; k < kTableTrips; ++k) {
            const uint32_t col = 64u * k + lane;
            if (col < 4u * EmitLane<S>::GP) {
                sh.colp[0][col] = column[k].n_off, sh.colp[1][col] = column[k].n_rng, sh.colp[2][col] = column[k].c_lo;
                sh.colp[3][col] = column[k].c_hi, sh.colp[4][col] = column[k].scale;
            }
        }
    }
    // The noise of the thread's observation columns needs nothing but (seed, step, env, group): drawn here, while the
    // wavefront's loads are in flight.  A trip past the block's last env shadows that env (and stores nothing later).
    {
        using E = EmitLane<S>;
        GBytes* noise = FastEnv<S>::block_base(d.noise_u, env0 * E::O);
        // (the columns' parameters were requested at the kernel's entry; this is their first use, behind the env's loads)
        em.noisy = em.active && (em.n_rng[0] != 0.0f || em.n_rng[1] != 0.0f || em.n_rng[2] != 0.0f || em.n_rng[3] != 0.0f);
#pragma unroll
        for (int trip = 0; trip < E::kTrips; ++trip) {
            const int raw = trip * E::kPerTrip + em.esub, ee = raw < ne ? raw : ne - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) em.u[trip][c] = 0.0f;
            if (em.noisy) {
                if (d.noise_u) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        em.u[trip][c] = FastEnv<S>::ldf(noise, 4u * static_cast<uint32_t>(ee * E::O + (4 * em.g + c < E::O ? 4 * em.g + c : E::O - 1)));
                    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): handed-in draws (tests) are waited for here, not behind the block's stores
                } else {
                    const u4 r = philox4(seed, gen_step, static_cast<uint64_t>(env0 + ee), static_cast<uint32_t>(em.g));
                    em.u[trip][0] = u01(r.x), em.u[trip][1] = u01(r.y), em.u[trip][2] = u01(r.z), em.u[trip][3] = u01(r.w);
                }
            }
        }
    }
    // ---- terminations (manager_based_rl_env.py:199-202, termination_manager.py:151-174): every wavefront needs the result
    ep_len += 1;
    bool terminated = false, time_out = false;
    uint32_t term_bits = 0;
    int contact_at = 0;
    static_for<S::NT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const ilx_term t = termination_of<S, i>(a);
        bool v;
        if constexpr (kContactsAhead && S::term[i].kind == ILX_TERM_ILLEGAL_CONTACT) {   // as eval_termination, on the values asked for above
            constexpr int n = S::term[i].ids_cnt < 0 ? S::CB : S::term[i].ids_cnt;
            v = false;
#pragma unroll
            for (int k = 0; k < n; ++k) v = v | (en.force_peak_of(contact_raw[contact_at + k]) > t.p[0]);
            contact_at += n;
        } else {
            v = eval_termination(en, t, ep_len);
        }
        term_bits |= v ? (1u << i) : 0u;
        if constexpr ((S::term[i].flags & 1) != 0) time_out = time_out || v; else terminated = terminated || v;
    });
    const bool done = terminated || time_out;
    const bool done_v = done && valid;
    if (done) { FAST_STAMP(1); } else { FAST_STAMP(1); }
    // ---- rewards (reward_manager.py:128-157): this wavefront's share of the terms, in table order
    float value[S::NR] = {};
    float partial = 0.0f;
    static_for<S::NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (S::rew[i].active && D::reward_owner(i) == W) {
            const ilx_term t = reward_of<S, i>(a);
            value[i] = eval_reward(en, t, terminated, time_out, term_bits) * t.weight * dt;
            partial += value[i];
        }
    });
    if constexpr (W < kWaves) sh.ptotal[W][lane] = partial;
    FAST_STAMP(2);
    // ---- the observation values this wavefront's lanes hold (observation_manager.py:260-335, before noise / clip / scale)
    float* row = P + lane * D::p_stride;
    static_for<S::NO>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int kind = S::obs[i].kind, c0 = D::obs_col(i);
        if constexpr (D::obs_owner(i) == W) {
            if constexpr (kind == ILX_OBS_BASE_POS_Z) row[c0] = en.pos.z;                                     // observations.py:33-37
            if constexpr (kind == ILX_OBS_BASE_LIN_VEL) row[c0] = en.lin_b.x, row[c0 + 1] = en.lin_b.y, row[c0 + 2] = en.lin_b.z;   // :40-44
            if constexpr (kind == ILX_OBS_BASE_ANG_VEL) row[c0] = en.ang_b.x, row[c0 + 1] = en.ang_b.y, row[c0 + 2] = en.ang_b.z;   // :47-51
            if constexpr (kind == ILX_OBS_PROJECTED_GRAVITY) row[c0] = en.grav_b.x, row[c0 + 1] = en.grav_b.y, row[c0 + 2] = en.grav_b.z;  // :54-58
            if constexpr (kind == ILX_OBS_ROOT_QUAT_W) row[c0] = en.quat.w, row[c0 + 1] = en.quat.x, row[c0 + 2] = en.quat.y, row[c0 + 3] = en.quat.z;
            if constexpr (kind == ILX_OBS_ROOT_LIN_VEL_W) row[c0] = en.lin_w.x, row[c0 + 1] = en.lin_w.y, row[c0 + 2] = en.lin_w.z;  // :85-89
            if constexpr (kind == ILX_OBS_ROOT_ANG_VEL_W) row[c0] = en.ang_w.x, row[c0 + 1] = en.ang_w.y, row[c0 + 2] = en.ang_w.z;  // :92-96
            if constexpr (kind == ILX_OBS_JOINT_POS) {                                                          // :104-111
#pragma unroll
                for (int k = 0; k < S::J; ++k) row[c0 + k] = en.joint(Fld::q, k);
            }
            if constexpr (kind == ILX_OBS_JOINT_POS_REL) {                                                      // :114-121
#pragma unroll
                for (int k = 0; k < S::J; ++k) row[c0 + k] = en.joint(Fld::q, k) - en.joint(Fld::q0, k);
            }
            if constexpr (kind == ILX_OBS_JOINT_VEL) {                                                          // :140-147
#pragma unroll
                for (int k = 0; k < S::J; ++k) row[c0 + k] = en.joint(Fld::qd, k);
            }
            if constexpr (kind == ILX_OBS_JOINT_VEL_REL) {                                                      // :150-157
#pragma unroll
                for (int k = 0; k < S::J; ++k) row[c0 + k] = en.joint(Fld::qd, k) - en.joint(Fld::qd0, k);
            }
            if constexpr (kind == ILX_OBS_LAST_ACTION) {   // :512-521; a done env's action history restarts first (action_manager.py:309-311)
#pragma unroll
                for (int k = 0; k < S::A; ++k) row[c0 + k] = done ? 0.0f : en.action(k);
            }
            if constexpr (kind == ILX_OBS_HEIGHT_SCAN && D::scan_head > 0) {   // the scan columns that share a group with other terms
                GBytes* scan = FastEnv<S>::block_base(d.height_scan, env0 * S::R);
#pragma unroll
                for (int k = 0; k < D::scan_head; ++k) row[c0 + k] = FastEnv<S>::ldf(scan + 4 * k, le * (4u * S::R));
            }
        }
    });
    // ---- reset bookkeeping of the managers for done envs (manager_based_rl_env.py:364-391) and the command term
    // (command_manager.py:151-166), in registers; the fifth wavefront
    int64_t ep_len_out = ep_len;
    if constexpr (W == kCmdWave) {
        if (done) ep_len_out = 0;  // manager_based_rl_env.py:391
        if constexpr (has_cmd) {
            const auto reset_samples = as_global(d.reset_command_samples);
            const auto step_samples = as_global(d.command_samples);
            if (done) {
                if (valid) {
                    atomicAdd(&sh.logp[S::NR + S::NT], cs.metric0);
                    atomicAdd(&sh.logp[S::NR + S::NT + 1], cs.metric1);
                }
                cs.metric0 = cs.metric1 = 0.0f;
                cs.counter = 0;
            }
            if constexpr (velocity) {   // a draw of the velocity term is a handful of instructions: the reference's order as it is
                if (done) command_resample(d, seed, gen_step, env, reset_samples, 16u, cs);
                command_compute(d, en, true, d.heading_command != 0, dt, d.max_command_step, d.heading_control_stiffness,
                                d.cmd_range[3][0], d.cmd_range[3][1], seed, gen_step, env, step_samples, en.body_row(d.pose_body_index), cs);
            } else {
                // metrics on the command as it stands, for the lanes that are not reset (a reset lane's are formed behind the
                // barrier, on its new command)
                CmdState standing = cs;
                command_metrics(en, false, d.max_command_step, tracked, standing);
                if (!done) cs.metric0 = standing.metric0, cs.metric1 = standing.metric1;
            }
            const bool drawn_elsewhere = pose && (done || (cs.time_left - dt) <= 0.0f);   // kDrawWave writes these lanes' columns
            static_for<S::NO>([&](auto ic) {   // generated_commands (observations.py:529-531) sees the updated command
                constexpr int i = decltype(ic)::value;
                if constexpr (S::obs[i].kind == ILX_OBS_GENERATED_COMMANDS) {
                    if (!drawn_elsewhere) {
#pragma unroll
                        for (int k = 0; k < S::CD; ++k) row[D::obs_col(i) + k] = cs.c[k];
                    }
                }
            });
        }
        sh.done[lane] = done_v ? 1u : 0u;
    }
    if constexpr (pose && W == kDrawWave) {
        // Which lanes draw is known here: a done lane with the reset's stream (command_manager.py:120-149), the others by
        // their clocks with compute's (:151-166); the wavefront draws once, every lane with its own stream.  A done lane
        // whose new interval is over within this very step draws again, as in the reference.
        const auto reset_samples = as_global(d.reset_command_samples);
        const auto step_samples = as_global(d.command_samples);
        if (done || (draw_time_left - dt) <= 0.0f) {
            CmdState fresh;
            const CommandBits bits = done ? bits_reset : bits_step;
            command_resample(d, seed, gen_step, env, done ? reset_samples : step_samples, done ? 16u : 0u, fresh, &bits);
            sh.drawn[0][0][lane] = fresh.time_left;
#pragma unroll
            for (int k = 0; k < 7; ++k) sh.drawn[0][1 + k][lane] = fresh.c[k];
            float second = __builtin_nanf("");
            if (done && (fresh.time_left - dt) <= 0.0f) {
                command_resample(d, seed, gen_step, env, step_samples, 0u, fresh);
                second = fresh.time_left;
#pragma unroll
                for (int k = 0; k < 7; ++k) sh.drawn[1][1 + k][lane] = fresh.c[k];
            }
            sh.drawn[1][0][lane] = second;
            static_for<S::NO>([&](auto ic) {   // generated_commands (observations.py:529-531) of these lanes
                constexpr int i = decltype(ic)::value;
                if constexpr (S::obs[i].kind == ILX_OBS_GENERATED_COMMANDS) {
#pragma unroll
                    for (int k = 0; k < S::CD; ++k) row[D::obs_col(i) + k] = fresh.c[k];
                }
            });
        }
    }
    FAST_STAMP(3);
    // ---- logging sums of the done envs (reward_manager.py:115-121, termination_manager.py:142-144)
    if (done_v) {
        static_for<S::NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (D::sum_owner(i) == W) atomicAdd(&sh.logp[i], sums[i] + value[i]);
        });
        if constexpr (W == 1) {
            static_for<S::NT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (term_bits & (1u << i)) atomicAdd(&sh.logp[S::NR + i], 1.0f);
            });
            atomicAdd(&sh.logp[S::NR + S::NT + 2], 1.0f);
        }
    }
    // ---- outputs nothing in this launch reads
    static_for<S::NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr bool active = S::rew[i].active;
        if constexpr (D::sum_owner(i) == W) {
            // the sum is the lane's own word: read above, written here
            const float s = done ? 0.0f : sums[i] + value[i];
            if (valid && (active || done)) stg(d.episode_sums, static_cast<int64_t>(i) * N + env0, 4u * le, s);
            if constexpr (active)
                if (valid) stg(d.step_reward, env0 * S::NR + i, le * (4u * S::NR), div_(value[i], dt));
        }
    });
    if constexpr (W == 3) {
        static_for<S::NT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (valid) stg(d.term_dones, static_cast<int64_t>(i) * N + env0, le, static_cast<uint8_t>((term_bits >> i) & 1u));
        });
    }
    if constexpr (W == 2) {
        if (valid) {
            stg(d.terminated, env0, le, terminated);
            stg(d.time_outs, env0, le, time_out);
            if (d.reset_buf) stg(d.reset_buf, env0, le, done);
        }
    }
    FAST_STAMP(4);
    __syncthreads();
    FAST_STAMP(5);
    // ---- the block's observation rows first: their stores then leave ahead of, not behind, the in-place stores below.  (The
    // compiler guards registers it believes to be still awaited with waits on the counter that loads and stores share, in
    // order; behind a batch of stores such a wait is a wait for the stores' trip to memory.)
    emit();
    // ---- in-place updates of what other wavefronts of the block have read by now
    if constexpr (pose && W == kCmdWave) {   // the drawn commands take their places (pose_command.py:92-120)
        if (done || (cs.time_left - dt) <= 0.0f) {
            cs.time_left = sh.drawn[0][0][lane];
#pragma unroll
            for (int k = 0; k < 7; ++k) cs.c[k] = sh.drawn[0][1 + k][lane];
            cs.counter += 1;
            if (done) {
                command_metrics(en, false, d.max_command_step, tracked, cs);
                cs.time_left = cs.time_left - dt;
                if (cs.time_left <= 0.0f) {
                    cs.time_left = sh.drawn[1][0][lane];
#pragma unroll
                    for (int k = 0; k < 7; ++k) cs.c[k] = sh.drawn[1][1 + k][lane];
                    cs.counter += 1;
                }
            }
        } else {
            cs.time_left = cs.time_left - dt;
        }
    }
    if constexpr (W == kCmdWave) {
        if (valid) {
            stg(d.episode_length_buf, env0, 8u * le, ep_len_out);
            if constexpr (has_cmd) {
                constexpr int width = velocity ? 3 : 7;
#pragma unroll
                for (int k = 0; k < width; ++k) stg(d.command, env0 * S::CD + k, le * (4u * S::CD), cs.c[k]);
                stg(d.time_left, env0, 4u * le, cs.time_left);
                stg(d.command_counter, env0, 8u * le, cs.counter);
                stg(d.command_metrics, env0, 4u * le, cs.metric0);
                stg(d.command_metrics, N + env0, 4u * le, cs.metric1);
                if constexpr (velocity) {
                    stg(d.heading_target, env0, 4u * le, cs.heading_target);
                    stg(d.is_heading_env, env0, le, cs.is_heading);
                    stg(d.is_standing_env, env0, le, cs.is_standing);
                }
            }
        }
    }
    if constexpr (W == 1) {   // the reward: the four partial sums in wavefront order
        float total = sh.ptotal[0][lane];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) total += sh.ptotal[w][lane];
        if (valid) stg(d.reward_buf, env0, 4u * le, total);
    }
}

// Four consecutive floats at any 4-byte boundary, moved by one 16-byte instruction (observation rows and scan rows are
// 4 O and 4 R bytes long: a group of four columns starts at a multiple of four bytes, not of sixteen).
typedef float Group4 __attribute__((ext_vector_type(4), aligned(4)));

// The height-scan columns that fill whole groups of four: dense scan -> noise -> clip -> scale -> observation row.
template <class S>
ILX_DEV void scan_emit(const FastArgs& a, uint32_t block, uint64_t seed, uint64_t gen_step) {
    using D = Derived<S>;
    if constexpr (D::scan_groups > 0) {
        const ilx_mdp& d = a.d;
        const int64_t env0 = static_cast<int64_t>(block) * kScanEnvs;
        const int64_t left = d.num_envs - env0;
        const int ne = static_cast<int>(left < kScanEnvs ? left : kScanEnvs);
        constexpr int GS = D::scan_groups, G0 = D::p_groups, O = D::O, R = S::R;
        constexpr int kTrips = (kScanEnvs * GS + kFastBlock - 1) / kFastBlock;
        // the term's post-processing: one record for all its columns, read as uniform values
        const auto* cp = reinterpret_cast<const __attribute__((address_space(4))) float*>(
            (const __attribute__((address_space(4))) char*)(a.colparam + 4 * G0));
        const float n_off = cp[0], n_rng = cp[1], c_lo = cp[2], c_hi = cp[3], scale = cp[4];
        const bool noisy = n_rng != 0.0f;
        GBytes* scan = FastEnv<S>::block_base(d.height_scan, env0 * R);
        GBytesW* rows = (GBytesW*)FastEnv<S>::block_base(d.obs, env0 * O);
        GBytes* noise = FastEnv<S>::block_base(d.noise_u, env0 * O);
        const bool inject = d.noise_u != nullptr;
        const int items = ne * GS;
        SCAN_STAMP(0);
        // Every trip's loads leave before the first trip's stores: loads and stores share one in-order counter, and a wait
        // for a load behind a store waits for the store to reach memory too (a trip to memory and back per trip).
        float z[kTrips][4], u[kTrips][4];
#pragma unroll
        for (int trip = 0; trip < kTrips; ++trip) {
            const int raw = static_cast<int>(threadIdx.x) + kFastBlock * trip;
            const int idx = raw < items ? raw : items - 1;   // lanes past the end shadow the last item and store nothing
            const int el = idx / GS, gs = idx - el * GS, g = G0 + gs;
            const int k0 = 4 * g - D::scan_col;              // the group's first ray
            const uint32_t in_at = 4u * static_cast<uint32_t>(el * R + k0), out_at = 4u * static_cast<uint32_t>(el * O + 4 * g);
            if (O % 4 == 0 || 4 * g + 3 < O) {   // a whole group: one 16-byte load
                const Group4 q = *reinterpret_cast<const __attribute__((address_space(1))) Group4*>(scan + in_at);
#pragma unroll
                for (int c = 0; c < 4; ++c) z[trip][c] = q[c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {   // the row's last group: a column past its end repeats the last one (not stored)
                    const uint32_t cc = 4 * g + c < O ? static_cast<uint32_t>(c) : static_cast<uint32_t>(O - 1 - 4 * g);
                    z[trip][c] = FastEnv<S>::ldf(scan, in_at + 4u * cc);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t cc = (O % 4 == 0 || 4 * g + c < O) ? static_cast<uint32_t>(c) : static_cast<uint32_t>(O - 1 - 4 * g);
                u[trip][c] = 0.0f;
                if (inject && noisy) u[trip][c] = FastEnv<S>::ldf(noise, out_at + 4u * cc);
            }
        }
        __builtin_amdgcn_sched_barrier(0x3cf);   // the loads stay above
#pragma unroll
        for (int trip = 0; trip < kTrips; ++trip) {
            const int raw = static_cast<int>(threadIdx.x) + kFastBlock * trip;
            const int idx = raw < items ? raw : items - 1;
            const int el = idx / GS, gs = idx - el * GS, g = G0 + gs;
            const uint32_t out_at = 4u * static_cast<uint32_t>(el * O + 4 * g);
            if (!inject && noisy) {
                const u4 r = philox4(seed, gen_step, static_cast<uint64_t>(env0 + el), static_cast<uint32_t>(g));
                u[trip][0] = u01(r.x), u[trip][1] = u01(r.y), u[trip][2] = u01(r.z), u[trip][3] = u01(r.w);
            }
            Group4 out;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = z[trip][c];
                v = v + u[trip][c] * n_rng + n_off;              // noise_model.py:63-64
                v = __builtin_amdgcn_fmed3f(v, c_lo, c_hi);      // observation_manager.py:315-316
                out[c] = v * scale;                            // :317-318
            }
            if (raw < items) {
                if (O % 4 == 0 || 4 * g + 3 < O) {
                    *reinterpret_cast<__attribute__((address_space(1))) Group4*>(rows + out_at) = out;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (4 * g + c < O) *reinterpret_cast<__attribute__((address_space(1))) float*>(rows + out_at + 4u * c) = out[c];
                }
            }
            if (trip < 6) SCAN_STAMP(2 + trip);
        }
        SCAN_STAMP(1);
    }
}

// S::kWavesPerSimd: the registers are held to what that many resident wavefronts per SIMD leave each of them.  The
// headline list runs 16 384 envs per GPU as 256 DECIDE + 482 SCAN blocks: at five wavefronts per SIMD three blocks of five
// wavefronts fit a CU however their wavefronts fall on its four SIMDs, so all of them are resident together; the lists
// that run a few thousand envs per GPU fill a fraction of the GPU and take the registers instead.
template <class S>
__global__ void __launch_bounds__(kFastBlock) __attribute__((amdgpu_waves_per_eu(S::kWavesPerSimd))) mdp_fast_kernel(const FastArgs a) {
    using D = Derived<S>;
    request_kernel_arguments<sizeof(FastArgs)>();
    const ilx_mdp& d = a.d;
    // The generator's step word is the argument plus the caller's device-side count when there is one (the caller
    // advances the count between launches, so it is constant over this one).
    const uint64_t seed = d.seed;
    // The scene update of the last physics substep beside the managers (lists without sensors, none of whose terms reads
    // what the update writes -- joint_acc and the previous joint velocities): its workgroups are the grid's last.
    if constexpr (S::kSceneBeside) {
        static_assert(!D::reads(ILX_REW_JOINT_ACC_L2) && S::CB == 0 && S::R == 0, "the update writes what this list reads");
        if (blockIdx.x >= a.decide_blocks + a.scan_blocks) {
            scene_update_thread(a.scene, blockIdx.x - a.decide_blocks - a.scan_blocks);
            return;
        }
    }
    if (blockIdx.x >= a.decide_blocks) {
        uint64_t gen_step = d.step_index;   // (a scalar load: these workgroups wait for their columns' parameters, scalar loads too, anyway)
        if (d.step_index_device)
            gen_step += *reinterpret_cast<const __attribute__((address_space(4))) uint64_t*>(
                (const __attribute__((address_space(4))) char*)d.step_index_device);
        scan_emit<S>(a, blockIdx.x - a.decide_blocks, seed, gen_step);
        return;
    }
    // Here the count is asked for first of all and through the vector memory path, and turned into a uniform value where
    // the step word is first needed: its trip to memory runs beside the wavefront's other loads.  (As a scalar load it
    // would be waited for by the next wait for any scalar load -- the scalar counter has no order -- that is, before the
    // first vector load could leave.)
    uint64_t steps_dev = 0;
    if (d.step_index_device) {
        uint32_t lane_zero = 0;
        asm volatile("" : "+v"(lane_zero));   // no instruction: the address is no longer known to be uniform
        steps_dev = *reinterpret_cast<const __attribute__((address_space(1))) uint64_t*>(
            (const __attribute__((address_space(1))) char*)d.step_index_device + lane_zero);
    }
    __shared__ float P[kFastEnvs * D::p_stride];
    __shared__ FastShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6));
    const int64_t env0 = static_cast<int64_t>(blockIdx.x) * kFastEnvs;
    const int64_t left = d.num_envs - env0;
    const int ne = static_cast<int>(left < kFastEnvs ? left : kFastEnvs);
    if (tid < S::NR + S::NT + 3) sh.logp[tid] = 0.0f;
    // The thread's place in the last stage (a group of four observation columns, a first env) and its columns'
    // post-processing parameters: requested now, so that they are there when the block comes out of its barrier.
    using E = EmitLane<S>;
    constexpr int GP = E::GP, O = E::O, kPerTrip = E::kPerTrip, kTrips = E::kTrips;
    E em;
    em.esub = static_cast<int>(tid) / GP, em.g = static_cast<int>(tid) - em.esub * GP;
    em.active = em.esub < kPerTrip;
    em.noisy = false;
    const int g = em.g, esub = em.esub;
    static_assert(GP <= kFastMaxGroups, "the block's table of column parameters");
#pragma unroll
    for (int c = 0; c < 4; ++c)   // a column past the row's end is treated like the row's last one (not stored)
        em.n_rng[c] = a.colparam[em.active ? (4 * g + c < O ? 4 * g + c : O - 1) : 0].n_rng;

    FAST_STAMP(0);
    // The block's first barrier orders the zeroed logging sums (LDS) ahead of the wavefronts' additions and nothing else:
    // a wait for the LDS writes and the barrier instruction, without the fence of __syncthreads(), which would also wait
    // for the column parameters' loads above -- a trip to memory before the env's loads could leave.
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    auto emit = [&]() {
        // ---- the observation columns that went through LDS: noise (drawn above), clip, scale, store
        // (observation_manager.py:313-318).  A lane owns one group of four columns and walks down the block's envs with it.
        if (em.active) {
            GBytesW* rows = (GBytesW*)FastEnv<S>::block_base(d.obs, env0 * O);
            float n_off[4], n_rng[4], c_lo[4], c_hi[4], scl[4];
    #pragma unroll
            for (int c = 0; c < 4; ++c) {
                n_off[c] = sh.colp[0][4 * g + c], n_rng[c] = sh.colp[1][4 * g + c], c_lo[c] = sh.colp[2][4 * g + c];
                c_hi[c] = sh.colp[3][4 * g + c], scl[c] = sh.colp[4][4 * g + c];
            }
    #pragma unroll
            for (int trip = 0; trip < kTrips; ++trip) {
                const int raw = trip * kPerTrip + esub, ee = raw < ne ? raw : ne - 1;
                const uint32_t out_at = 4u * static_cast<uint32_t>(raw * O + 4 * g);
                Group4 out;
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = P[ee * D::p_stride + 4 * g + c];
                    x = x + em.u[trip][c] * n_rng[c] + n_off[c];         // noise_model.py:63-64
                    x = __builtin_amdgcn_fmed3f(x, c_lo[c], c_hi[c]);    // observation_manager.py:315-316
                    out[c] = x * scl[c];                               // :317-318
                }
                if (raw < ne) {
                    if (4 * g + 3 < O) {   // a whole group: one 16-byte store
                        *reinterpret_cast<__attribute__((address_space(1))) Group4*>(rows + out_at) = out;
                    } else {
    #pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (4 * g + c < O) *reinterpret_cast<__attribute__((address_space(1))) float*>(rows + out_at + 4u * c) = out[c];
                    }
                }
            }
        }
    };
    static_assert(kWaves == 4, "one path per wavefront below");
    switch (wave) {
        case 0: decide<S, 0>(a, env0, ne, lane, seed, steps_dev, P, sh, em, emit); break;
        case 1: decide<S, 1>(a, env0, ne, lane, seed, steps_dev, P, sh, em, emit); break;
        case 2: decide<S, 2>(a, env0, ne, lane, seed, steps_dev, P, sh, em, emit); break;
        case 3: decide<S, 3>(a, env0, ne, lane, seed, steps_dev, P, sh, em, emit); break;
        default: decide<S, kCmdWave>(a, env0, ne, lane, seed, steps_dev, P, sh, em, emit); break;
    }
    // (each of the five paths above ends behind the block's barrier)
    FAST_STAMP(6);
    // ---- a done env's action history restarts (action_manager.py:309-311)
    if constexpr (D::needs_action) {
        for (int i = static_cast<int>(tid); i < ne * S::A; i += kFastBlock) {
            const int e = i / S::A;
            if (sh.done[e]) {
                stg(d.action, env0 * S::A, 4u * i, 0.0f);
                stg(d.prev_action, env0 * S::A, 4u * i, 0.0f);
            }
        }
    }
    // ---- the block's logging sums: one hardware float add per entry (no return value)
    if (d.log_sums && tid < S::NR + S::NT + 3) {
        const float v = sh.logp[tid];
        if (v != 0.0f) unsafeAtomicAdd(d.log_sums + tid, v);
    }
    FAST_STAMP(7);
}

#ifdef ILX_FAST_STAMPS
extern "C" int ilx_debug_scan_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_scan_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
extern "C" int ilx_debug_fast_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_fast_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif

// ---- the table of compiled kernels ---------------------------------------------------------------------------
template <class S>
static bool sig_equal(const TermSig& s, const ilx_term& t, bool reward) {
    return s.kind == t.kind && s.ids_cnt == t.ids_cnt && s.ids2_cnt == t.ids2_cnt && s.flags == t.flags &&
           (!reward || s.active == (t.weight != 0.0f));
}

template <class S>
static bool matches(const ilx_mdp_plan_s& p, const ilx_mdp& d) {
    using D = Derived<S>;
    static_assert(S::NT + S::NR <= kFastMaxTerms && D::num_ids <= kFastMaxIds, "term list too long for the kernel arguments");
    static_assert(S::NR + S::NT + 3 <= 2 * kMaxTerms + 3 && D::p_groups >= 1 && D::p_groups <= kFastBlock, "block geometry");
    if (p.nt != S::NT || p.nr != S::NR || p.no != S::NO || p.obs_dim != D::O || p.nids != D::num_ids) return false;
    if (d.num_joints != S::J || d.num_bodies != S::B || d.num_sensor_bodies != S::CB || d.history_len != S::H ||
        d.num_rays != S::R || d.action_dim != S::A || d.command_dim != S::CD || d.command_kind != S::cmd_kind)
        return false;
    const ilx_term* t = p.h_terms;
    for (int i = 0; i < S::NT; ++i)
        if (!sig_equal<S>(S::term[i], t[i], false) || (t[i].ids_cnt > 0 && t[i].ids_off != D::term_ids_off(i))) return false;
    t += S::NT;
    for (int i = 0; i < S::NR; ++i) {
        if (!sig_equal<S>(S::rew[i], t[i], true)) return false;
        if (t[i].ids_cnt > 0 && t[i].ids_off != D::rew_ids_off(i)) return false;
        if (t[i].ids2_cnt > 0 && t[i].ids2_off != D::rew_ids_off(i) + (t[i].ids_cnt > 0 ? t[i].ids_cnt : 0)) return false;
    }
    t += S::NR;
    for (int i = 0; i < S::NO; ++i)
        if (S::obs[i].kind != t[i].kind || t[i].ids_cnt >= 0 || (t[i].flags & ILX_OBS_FLAG_QUAT_UNIQUE)) return false;
    return p.reward_owner == D::owner;   // the partial sums of the reward are formed per wavefront
}

template <class S>
static int launch(const ilx_mdp_plan_s& p, const ilx_mdp& d, void* stream, const ilx_scene* scene) {
    using D = Derived<S>;
    // every buffer the compiled list touches must be there; anything else is the plan-interpreting kernel's business
    const bool joints = d.joint_pos && d.joint_vel && d.joint_acc && d.applied_torque && d.computed_torque && d.default_joint_pos &&
                        d.default_joint_vel && d.soft_joint_pos_limits && d.soft_joint_vel_limits;
    if (!joints || !d.episode_length_buf || !d.term_dones || !d.terminated || !d.time_outs || !d.reward_buf || !d.episode_sums ||
        !d.step_reward || !d.obs)
        return 1;
    if (D::needs_action && (!d.action || !d.prev_action)) return 1;
    if (S::cmd_kind != ILX_CMD_NONE && !(d.command && d.time_left && d.command_counter && d.command_metrics)) return 1;
    if (S::cmd_kind == ILX_CMD_VELOCITY && !(d.heading_target && d.is_heading_env && d.is_standing_env)) return 1;
    if (S::cmd_kind == ILX_CMD_POSE && !d.body_state_w) return 1;
    if (S::CB > 0 && !(d.net_forces_w_history && d.current_air_time && d.last_air_time && d.current_contact_time)) return 1;
    if (D::has_scan && !d.height_scan) return 1;
    if (D::reads(ILX_REW_FEET_SLIDE) && !d.body_state_w) return 1;
    FastArgs a;
    a.d = d;
    a.d.terminations = a.d.rewards = a.d.observations = nullptr;
    a.d.ids = nullptr;
    std::memset(a.terms, 0, sizeof(a.terms));
    std::memcpy(a.terms, p.h_terms, sizeof(ilx_term) * (S::NT + S::NR));
    std::memset(a.ids, 0, sizeof(a.ids));
    std::memcpy(a.ids, p.h_ids, sizeof(int32_t) * D::num_ids);
    a.colparam = p.d_colparam;
    a.decide_blocks = static_cast<uint32_t>((d.num_envs + kFastEnvs - 1) / kFastEnvs);
    a.scan_blocks = D::scan_groups > 0 ? static_cast<uint32_t>((d.num_envs + kScanEnvs - 1) / kScanEnvs) : 0u;
    a.scene = SceneArgs{};
    a.scene_blocks = 0;
    if (scene && scene->num_envs > 0) {
        // beside the managers: an articulation-only update (joint_acc), which writes nothing that this list reads
        const bool beside = S::kSceneBeside && !scene->net_forces_in && !scene->ray_timestamp;
        if (beside) {
            unsigned blocks = 0;
            const int rc = scene_args_from(*scene, a.scene, blocks);
            if (rc != ILX_OK) return rc;
            a.scene_blocks = blocks;
        } else {
            const int rc = ilx_scene_update(scene, stream);
            if (rc != ILX_OK) return rc;
        }
    }
    ILX_LAUNCH(mdp_fast_kernel<S>, dim3(a.decide_blocks + a.scan_blocks + a.scene_blocks), dim3(kFastBlock), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_mdp_step");
    return ILX_OK;
}

struct FastKernel {
    const char* name;
    bool (*matches)(const ilx_mdp_plan_s&, const ilx_mdp&);
    int (*launch)(const ilx_mdp_plan_s&, const ilx_mdp&, void*, const ilx_scene*);
};
template <class S>
static constexpr FastKernel entry() {
    return FastKernel{S::name, &matches<S>, &launch<S>};
}
static const FastKernel kKernels[] = {entry<AnymalRough>(), entry<H1Rough>(), entry<Cartpole>(), entry<Reach<9, 11, 7>>(),
                                      entry<Reach<9, 11, 6>>(), entry<Reach<8, 10, 7>>(), entry<Reach<8, 10, 6>>()};
constexpr int kNumKernels = static_cast<int>(sizeof(kKernels) / sizeof(kKernels[0]));

int mdp_fast_match(const ilx_mdp_plan_s& plan, const ilx_mdp& d) {
    for (int i = 0; i < kNumKernels; ++i)
        if (kKernels[i].matches(plan, d)) return i;
    return -1;
}

const char* mdp_fast_name(int index) { return (index >= 0 && index < kNumKernels) ? kKernels[index].name : "generic"; }

int mdp_fast_step(const ilx_mdp_plan_s& plan, const ilx_mdp& d, void* stream, const ilx_scene* scene) {
    if (plan.fast_kernel < 0 || plan.fast_kernel >= kNumKernels) return 1;
    if ((d.phases & ILX_PHASE_ALL) != ILX_PHASE_ALL) return 1;
    // the sizes are part of the match; a descriptor that changed them since the plan was made goes the general way
    if (!kKernels[plan.fast_kernel].matches(plan, d)) return 1;
    return kKernels[plan.fast_kernel].launch(plan, d, stream, scene);
}

}  // namespace ilx
