// Fused MDP step: termination / reward / reset-bookkeeping / command / observation managers in ONE launch.
//
// Mapping: one 64-lane wavefront per environment, four environments per 256-thread block.  Rows of the
// ArticulationData arrays ((N,J), (N,B,3), (N,CB) ...) are contiguous per env, so lane k reading element k of the
// env's row is a coalesced access without any AoS->SoA staging.
//
// Schedule per wave: (1) GATHER -- every load any active term needs is issued back to back into registers (lane j
// holds joint j's nine values, lane b body b's contact timers / peak force / velocity, lane i reward term i's
// episode sum, ...), so the wave pays ONE memory round trip instead of one per term; (2) the term tables, staged
// in LDS once per block, are interpreted from registers: sums over joints / bodies are shuffle butterflies,
// index lists are lane permutes, decisions are ballots; (3) SCATTER -- outputs leave in a few coalesced stores.
// The chain termination -> reward -> reset -> command -> observation never round-trips through memory.
//
// `kind` is uniform across the wave, so the dispatch is scalar control flow.  Term formulas cite the reference
// file:line; op order for decisions follows math_dev.h.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd_mdp.h"

#include <cstring>
#include <vector>

namespace ilx {

constexpr int kMaxTerms = 32;      // per manager
constexpr int kMaxIds = 1024;      // index-list entries staged in LDS
constexpr int kMaxRow = 64;        // joints / bodies / sensor bodies / action / command columns: one per lane
constexpr int kRayRegs = 4;        // ray z values prefetched per lane (num_rays <= 256; beyond that: direct loads)

enum Need : uint32_t {
    kNeedJoint = 1u, kNeedForces = 2u, kNeedTimers = 4u, kNeedBodyVel = 8u, kNeedBodyAcc = 16u, kNeedAction = 32u,
    kNeedRays = 64u,
};

struct MdpDev {
    ilx_mdp d;                // by-value copy; the HOST pointers inside are nulled
    const ilx_term* terms;    // device: [terminations | rewards | observations]
    const int32_t* ids;       // device
    uint32_t need;
};

// Cross-lane moves inside the VALU (data-parallel primitives): no LDS round trip.  Called with all 64 lanes active.
template <int CTRL>
ILX_DEV float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;
ILX_DEV float lane_const(float v, int src) {  // src uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// Sum / min over the 64 lanes, result uniform.  Four in-row steps, then the four row totals are combined.
ILX_DEV float wave_sum(float v) {
    v += dpp<kQuadXor1>(v);
    v += dpp<kQuadXor2>(v);
    v += dpp<kRowHalfMirror>(v);
    v += dpp<kRowMirror>(v);
    return (lane_const(v, 0) + lane_const(v, 16)) + (lane_const(v, 32) + lane_const(v, 48));
}
ILX_DEV float wave_min(float v) {
    v = fminf(v, dpp<kQuadXor1>(v));
    v = fminf(v, dpp<kQuadXor2>(v));
    v = fminf(v, dpp<kRowHalfMirror>(v));
    v = fminf(v, dpp<kRowMirror>(v));
    return fminf(fminf(lane_const(v, 0), lane_const(v, 16)), fminf(lane_const(v, 32), lane_const(v, 48)));
}
ILX_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }
// Column selector of a term entry: `identity` (uniform) short-cuts the permute when the column is the lane itself.
struct Sel {
    int c;
    bool identity;
};
ILX_DEV float lane_get(float v, Sel s) { return s.identity ? v : __shfl(v, s.c, 64); }
ILX_DEV float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
ILX_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Philox4x32-10 (counter based): reproducible per (seed, step, env, element) whatever the launch geometry.
ILX_DEV uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
ILX_DEV float philox_uniform(uint64_t seed, uint64_t step, uint64_t env, uint32_t elem) {
    uint32_t c0 = elem, c1 = static_cast<uint32_t>(env), c2 = static_cast<uint32_t>(env >> 32) ^ static_cast<uint32_t>(step),
             c3 = static_cast<uint32_t>(step >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return static_cast<float>(c0 >> 8) * (1.0f / 16777216.0f);  // 24 random mantissa bits, [0, 1)
}

// A term with its fields in scalar registers.
struct Term {
    int kind, flags;
    float weight, p0, p1, p2, p3, aux;
    int ids_off, ids_cnt, ids2_off, ids2_cnt;
};
ILX_DEV Term fetch_term(const ilx_term* lds) {
    const ilx_term t = *lds;
    return Term{uni(t.kind), uni(t.flags), uni(t.weight), uni(t.p[0]), uni(t.p[1]), uni(t.p[2]), uni(t.p[3]), uni(t.aux),
                uni(t.ids_off), uni(t.ids_cnt), uni(t.ids2_off), uni(t.ids2_cnt)};
}

// Per-lane slice of one environment's state (lane k owns column k of every row).
struct Row {
    float q = 0, qd = 0, qdd = 0, tau_a = 0, tau_c = 0, q0 = 0, qd0 = 0, lim_lo = 0, lim_hi = 0, vlim = 0;  // joint k
    float act = 0, pact = 0;                                                                           // action column k
    float air = 0, last_air = 0, contact_t = 0, fpeak = 0;  // sensor body k: timers, max_h |F_h| (rewards.py:266)
    float bvx = 0, bvy = 0, bacc = 0;                       // asset body k: lin vel xy, |lin acc|
    float cmd = 0;                                          // command column k
};

struct Env {
    const MdpDev& m;
    const int32_t* ids;  // LDS copy of the index lists
    int64_t env;
    int lane;
    Row r;
    f3 pos, lin_w, ang_w, lin_b, ang_b, grav_b;
    f4 quat;
    float heading;
    float time_left = 0.0f;
    int64_t command_counter = 0;

    // register-file column a term's k-th entry refers to (ids list, or identity for slice(None))
    ILX_DEV Sel col(const Term& t, int n) const {
        if (t.ids_cnt < 0) return Sel{lane, true};
        return Sel{lane < n ? ids[t.ids_off + lane] : 0, false};
    }
    ILX_DEV Sel col2(const Term& t, int n) const {
        if (t.ids2_cnt < 0) return Sel{lane, true};
        return Sel{lane < n ? ids[t.ids2_off + lane] : 0, false};
    }
    ILX_DEV int jcount(const Term& t) const { return t.ids_cnt < 0 ? m.d.num_joints : t.ids_cnt; }
    ILX_DEV int bcount_sensor(const Term& t) const { return t.ids_cnt < 0 ? m.d.num_sensor_bodies : t.ids_cnt; }
    ILX_DEV int bcount_asset(const Term& t) const { return t.ids_cnt < 0 ? m.d.num_bodies : t.ids_cnt; }
    ILX_DEV float cmd(int k) const { return lane_const(r.cmd, k); }
    // sum over the listed joints of f(column)
    template <typename F>
    ILX_DEV float joint_sum(const Term& t, F f) const {
        const int n = jcount(t);
        const float v = f(col(t, n));
        return wave_sum(lane < n ? v : 0.0f);
    }
    template <typename F>
    ILX_DEV bool joint_any(const Term& t, F f) const {
        const int n = jcount(t);
        const bool v = f(col(t, n));
        return wave_any(lane < n && v);
    }
};

// --------------------------------------------------------------------------------------------------
ILX_DEV bool eval_termination(const Env& e, const Term& t, int64_t ep_len) {
    const ilx_mdp& d = e.m.d;
    const Row& r = e.r;
    switch (t.kind) {
        case ILX_TERM_TIME_OUT:  // terminations.py:30-32
            return ep_len >= d.max_episode_length;
        case ILX_TERM_COMMAND_RESAMPLE:  // :35-42
            return (e.time_left <= t.p0) && (e.command_counter == static_cast<int64_t>(t.p1));
        case ILX_TERM_BAD_ORIENTATION:  // :50-59
            return fabsf(acosf(-e.grav_b.z)) > t.p0;
        case ILX_TERM_ROOT_HEIGHT_BELOW_MINIMUM:  // :62-72
            return e.pos.z < t.p0;
        case ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT:  // :90-105
            return e.joint_any(t, [&](Sel c) { const float q = lane_get(r.q, c); return (q > t.p1) || (q < t.p0); });
        case ILX_TERM_JOINT_VEL_OUT_OF_LIMIT:  // :108-114
            return e.joint_any(t, [&](Sel c) { return fabsf(lane_get(r.qd, c)) > lane_get(r.vlim, c); });
        case ILX_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT:  // :117-124
            return e.joint_any(t, [&](Sel c) { return fabsf(lane_get(r.qd, c)) > t.p0; });
        case ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT:  // :127-142, torch.isclose(rtol=1e-5, atol=1e-8)
            return e.joint_any(t, [&](Sel c) {
                const float a = lane_get(r.tau_c, c), b = lane_get(r.tau_a, c);
                return (a == b) || (fabsf(a - b) <= (1.0e-8f + 1.0e-5f * fabsf(b)));
            });
        case ILX_TERM_ILLEGAL_CONTACT: {  // :150-158
            const int n = e.bcount_sensor(t);
            const float f = lane_get(r.fpeak, e.col(t, n));
            return wave_any(e.lane < n && f > t.p0);
        }
    }
    return false;
}

ILX_DEV float eval_reward(const Env& e, const Term& t, bool terminated, bool time_out, uint32_t term_bits) {
    const ilx_mdp& d = e.m.d;
    const Row& r = e.r;
    const int B = d.num_bodies, A = d.action_dim;
    const int lane = e.lane;
    switch (t.kind) {
        case ILX_REW_IS_ALIVE: return terminated ? 0.0f : 1.0f;       // rewards.py:31-33
        case ILX_REW_IS_TERMINATED: return terminated ? 1.0f : 0.0f;  // :36-38
        case ILX_REW_IS_TERMINATED_TERM:                              // :41-68
            return static_cast<float>(__popc(term_bits & static_cast<uint32_t>(t.flags))) * (time_out ? 0.0f : 1.0f);
        case ILX_REW_LIN_VEL_Z_L2: return e.lin_b.z * e.lin_b.z;                              // :76-80
        case ILX_REW_ANG_VEL_XY_L2: return e.ang_b.x * e.ang_b.x + e.ang_b.y * e.ang_b.y;    // :83-87
        case ILX_REW_FLAT_ORIENTATION_L2: return e.grav_b.x * e.grav_b.x + e.grav_b.y * e.grav_b.y;  // :90-97
        case ILX_REW_BASE_HEIGHT_L2: {  // :100-122
            float target = t.p0;
            if (t.flags & 1) {
                float acc = 0.0f;
                for (int k = lane; k < d.num_rays; k += 64) acc += d.ray_hits_w[3 * (e.env * d.num_rays + k) + 2];
                target = t.p0 + div_(wave_sum(acc), static_cast<float>(d.num_rays));
            }
            const float dz = e.pos.z - target;
            return dz * dz;
        }
        case ILX_REW_BODY_LIN_ACC_L2: {  // :125-128
            const int n = e.bcount_asset(t);
            const float a = lane_get(r.bacc, e.col(t, n));
            return wave_sum(lane < n ? a : 0.0f);
        }
        case ILX_REW_JOINT_TORQUES_L2:  // :136-143
            return e.joint_sum(t, [&](Sel c) { const float v = lane_get(r.tau_a, c); return v * v; });
        case ILX_REW_JOINT_VEL_L1:  // :146-150
            return e.joint_sum(t, [&](Sel c) { return fabsf(lane_get(r.qd, c)); });
        case ILX_REW_JOINT_VEL_L2:  // :153-160
            return e.joint_sum(t, [&](Sel c) { const float v = lane_get(r.qd, c); return v * v; });
        case ILX_REW_JOINT_ACC_L2:  // :163-170
            return e.joint_sum(t, [&](Sel c) { const float v = lane_get(r.qdd, c); return v * v; });
        case ILX_REW_JOINT_DEVIATION_L1:  // :173-179
            return e.joint_sum(t, [&](Sel c) { return fabsf(lane_get(r.q, c) - lane_get(r.q0, c)); });
        case ILX_REW_JOINT_POS_LIMITS:  // :182-196
            return e.joint_sum(t, [&](Sel c) {
                const float q = lane_get(r.q, c);
                return -fminf(q - lane_get(r.lim_lo, c), 0.0f) + fmaxf(q - lane_get(r.lim_hi, c), 0.0f);
            });
        case ILX_REW_JOINT_VEL_LIMITS:  // :199-218
            return e.joint_sum(t, [&](Sel c) {
                return fminf(fmaxf(fabsf(lane_get(r.qd, c)) - lane_get(r.vlim, c) * t.p0, 0.0f), 1.0f);
            });
        case ILX_REW_APPLIED_TORQUE_LIMITS:  // :226-242
            return e.joint_sum(t, [&](Sel c) { return fabsf(lane_get(r.tau_a, c) - lane_get(r.tau_c, c)); });
        case ILX_REW_ACTION_RATE_L2: {  // :245-247
            const float v = r.act - r.pact;
            return wave_sum(lane < A ? v * v : 0.0f);
        }
        case ILX_REW_ACTION_L2:  // :250-252
            return wave_sum(lane < A ? r.act * r.act : 0.0f);
        case ILX_REW_UNDESIRED_CONTACTS: {  // :260-268 (count of bodies above the threshold)
            const int n = e.bcount_sensor(t);
            const float f = lane_get(r.fpeak, e.col(t, n));
            return wave_sum((lane < n && f > t.p0) ? 1.0f : 0.0f);
        }
        case ILX_REW_CONTACT_FORCES: {  // :271-279
            const int n = e.bcount_sensor(t);
            const float f = lane_get(r.fpeak, e.col(t, n));
            return wave_sum(lane < n ? fmaxf(f - t.p0, 0.0f) : 0.0f);
        }
        case ILX_REW_TRACK_LIN_VEL_XY_EXP: {  // :287-298
            const float dx = e.cmd(0) - e.lin_b.x, dy = e.cmd(1) - e.lin_b.y;
            return expf(div_(-(dx * dx + dy * dy), t.p0));
        }
        case ILX_REW_TRACK_ANG_VEL_Z_EXP: {  // :301-309
            const float dz = e.cmd(2) - e.ang_b.z;
            return expf(div_(-(dz * dz), t.p0));
        }
        case ILX_REW_FEET_AIR_TIME: {  // velocity/mdp/rewards.py:25-44
            const int n = e.bcount_sensor(t);
            const Sel c = e.col(t, n);
            const float cc = lane_get(r.contact_t, c), la = lane_get(r.last_air, c);
            const bool first = (cc > 0.0f) && (cc < t.p1);  // contact_sensor.py:200-203
            const float acc = wave_sum(lane < n ? (la - t.p0) * (first ? 1.0f : 0.0f) : 0.0f);
            return acc * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED: {  // velocity/mdp/rewards.py:47-66
            const int n = e.bcount_sensor(t);
            const Sel c = e.col(t, n);
            const float cc = lane_get(r.contact_t, c), ca = lane_get(r.air, c);
            const bool single = wave_sum((lane < n && cc > 0.0f) ? 1.0f : 0.0f) == 1.0f;
            const float mode = (cc > 0.0f) ? cc : ca;
            const float mn = wave_min(lane < n ? (single ? mode : 0.0f) : __builtin_inff());
            return fminf(mn, t.p0) * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_SLIDE: {  // velocity/mdp/rewards.py:69-83
            const int n = e.bcount_sensor(t);
            const bool contact = lane_get(r.fpeak, e.col(t, n)) > 1.0f;
            const Sel ab = e.col2(t, n);
            const float sp = norm2(lane_get(r.bvx, ab), lane_get(r.bvy, ab));
            return wave_sum(lane < n ? sp * (contact ? 1.0f : 0.0f) : 0.0f);
        }
        case ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {  // velocity/mdp/rewards.py:86-96
            const f3 v = quat_rotate_inverse(yaw_quat(e.quat), e.lin_w);
            const float dx = e.cmd(0) - v.x, dy = e.cmd(1) - v.y;
            return expf(div_(-(dx * dx + dy * dy), t.p0));
        }
        case ILX_REW_TRACK_ANG_VEL_Z_WORLD_EXP: {  // velocity/mdp/rewards.py:99-106
            const float dz = e.cmd(2) - e.ang_w.z;
            return expf(div_(-(dz * dz), t.p0));
        }
        case ILX_REW_POSITION_COMMAND_ERROR:
        case ILX_REW_POSITION_COMMAND_ERROR_TANH: {  // reach/mdp/rewards.py:19-52
            const f3 rr = quat_apply(e.quat, f3{e.cmd(0), e.cmd(1), e.cmd(2)});
            const f3 des = {e.pos.x + rr.x, e.pos.y + rr.y, e.pos.z + rr.z};
            const float* b = d.body_state_w + 13 * (e.env * B + e.ids[t.ids_off]);
            const float dist = norm3(b[0] - des.x, b[1] - des.y, b[2] - des.z);
            return t.kind == ILX_REW_POSITION_COMMAND_ERROR ? dist : (1.0f - tanhf(div_(dist, t.p0)));
        }
        case ILX_REW_ORIENTATION_COMMAND_ERROR: {  // reach/mdp/rewards.py:55-69
            const f4 des = quat_mul(e.quat, f4{e.cmd(3), e.cmd(4), e.cmd(5), e.cmd(6)});
            const float* b = d.body_state_w + 13 * (e.env * B + e.ids[t.ids_off]) + 3;
            return quat_error_magnitude(f4{b[0], b[1], b[2], b[3]}, des);
        }
        case ILX_REW_JOINT_POS_TARGET_L2:  // cartpole/mdp/rewards.py:19-26
            return e.joint_sum(t, [&](Sel c) { const float v = wrap_to_pi(lane_get(r.q, c)) - t.p0; return v * v; });
    }
    return 0.0f;
}

ILX_DEV int obs_dim_of(const ilx_mdp& d, const Term& t) {
    const int nj = t.ids_cnt < 0 ? d.num_joints : t.ids_cnt;
    switch (t.kind) {
        case ILX_OBS_BASE_POS_Z: return 1;
        case ILX_OBS_BASE_LIN_VEL:
        case ILX_OBS_BASE_ANG_VEL:
        case ILX_OBS_PROJECTED_GRAVITY:
        case ILX_OBS_ROOT_POS_W:
        case ILX_OBS_ROOT_LIN_VEL_W:
        case ILX_OBS_ROOT_ANG_VEL_W: return 3;
        case ILX_OBS_ROOT_QUAT_W: return 4;
        case ILX_OBS_JOINT_POS:
        case ILX_OBS_JOINT_POS_REL:
        case ILX_OBS_JOINT_POS_LIMIT_NORMALIZED:
        case ILX_OBS_JOINT_VEL:
        case ILX_OBS_JOINT_VEL_REL: return nj;
        case ILX_OBS_HEIGHT_SCAN: return d.num_rays;
        case ILX_OBS_LAST_ACTION: return d.action_dim;
        case ILX_OBS_GENERATED_COMMANDS: return d.command_dim;
    }
    return 0;
}

// Column k = lane of a term that is at most 64 wide (everything except the height scan).
ILX_DEV float eval_obs_row(const Env& e, const Term& t, int dim) {
    const ilx_mdp& d = e.m.d;
    const Row& r = e.r;
    const int k = e.lane;
    auto pick3 = [](f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); };
    switch (t.kind) {
        case ILX_OBS_BASE_POS_Z: return e.pos.z;                 // observations.py:33-37
        case ILX_OBS_BASE_LIN_VEL: return pick3(e.lin_b, k);      // :40-44
        case ILX_OBS_BASE_ANG_VEL: return pick3(e.ang_b, k);      // :47-51
        case ILX_OBS_PROJECTED_GRAVITY: return pick3(e.grav_b, k);  // :54-58
        case ILX_OBS_ROOT_POS_W:                                  // :61-65
            return pick3(e.pos, k) - ((d.env_origins && k < 3) ? d.env_origins[3 * e.env + k] : 0.0f);
        case ILX_OBS_ROOT_QUAT_W: {  // :68-82
            const f4 q = (t.flags & ILX_OBS_FLAG_QUAT_UNIQUE) ? quat_unique(e.quat) : e.quat;
            return k == 0 ? q.w : (k == 1 ? q.x : (k == 2 ? q.y : q.z));
        }
        case ILX_OBS_ROOT_LIN_VEL_W: return pick3(e.lin_w, k);  // :85-89
        case ILX_OBS_ROOT_ANG_VEL_W: return pick3(e.ang_w, k);  // :92-96
        case ILX_OBS_JOINT_POS: return lane_get(r.q, e.col(t, dim));  // :104-111
        case ILX_OBS_JOINT_POS_REL: {                                   // :114-121
            const Sel c = e.col(t, dim);
            return lane_get(r.q, c) - lane_get(r.q0, c);
        }
        case ILX_OBS_JOINT_POS_LIMIT_NORMALIZED: {  // :124-137 (math.py:23-40)
            const Sel c = e.col(t, dim);
            const float lo = lane_get(r.lim_lo, c), hi = lane_get(r.lim_hi, c);
            const float off = (lo + hi) * 0.5f;
            return div_(2.0f * (lane_get(r.q, c) - off), hi - lo);
        }
        case ILX_OBS_JOINT_VEL: return lane_get(r.qd, e.col(t, dim));  // :140-147
        case ILX_OBS_JOINT_VEL_REL: {                                    // :150-157
            const Sel c = e.col(t, dim);
            return lane_get(r.qd, c) - lane_get(r.qd0, c);
        }
        case ILX_OBS_LAST_ACTION: return r.act;          // :512-521
        case ILX_OBS_GENERATED_COMMANDS: return r.cmd;   // :529-531
    }
    return 0.0f;
}

// Command term state of one env, uniform across the wave (lane 0 stores it).
struct CmdState {
    float time_left = 0.0f, metric0 = 0.0f, metric1 = 0.0f, heading_target = 0.0f;
    int64_t counter = 0;
    bool is_heading = false, is_standing = false;
    float c[7] = {0, 0, 0, 0, 0, 0, 0};
};

// sample s of the command term: injected value, or generated in the stated range
ILX_DEV float cmd_sample(const ilx_mdp& d, const float* injected, int64_t env, int s, bool unit, uint32_t salt) {
    if (injected) return injected[env * 7 + s];
    const float u = philox_uniform(d.seed, d.step_index, static_cast<uint64_t>(env), 0x40000000u + salt + s);
    if (unit) return u;
    return u * (d.cmd_range[s][1] - d.cmd_range[s][0]) + d.cmd_range[s][0];
}

// CommandTerm._resample for one env: command_manager.py:172-187 + the term's _resample_command
ILX_DEV void command_resample(const ilx_mdp& d, int64_t env, const float* injected, uint32_t salt, CmdState& s) {
    s.time_left = cmd_sample(d, injected, env, 0, false, salt);
    s.counter += 1;
    if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:123-138
        s.c[0] = cmd_sample(d, injected, env, 1, false, salt);
        s.c[1] = cmd_sample(d, injected, env, 2, false, salt);
        s.c[2] = cmd_sample(d, injected, env, 3, false, salt);
        if (d.heading_command) {
            s.heading_target = cmd_sample(d, injected, env, 4, false, salt);
            s.is_heading = cmd_sample(d, injected, env, 5, true, salt) <= d.rel_heading_envs;
        }
        s.is_standing = cmd_sample(d, injected, env, 6, true, salt) <= d.rel_standing_envs;
    } else if (d.command_kind == ILX_CMD_POSE) {  // pose_command.py:107-120
        s.c[0] = cmd_sample(d, injected, env, 1, false, salt);
        s.c[1] = cmd_sample(d, injected, env, 2, false, salt);
        s.c[2] = cmd_sample(d, injected, env, 3, false, salt);
        f4 q = quat_from_euler_xyz(cmd_sample(d, injected, env, 4, false, salt), cmd_sample(d, injected, env, 5, false, salt),
                                   cmd_sample(d, injected, env, 6, false, salt));
        if (d.make_quat_unique) q = quat_unique(q);
        s.c[3] = q.w, s.c[4] = q.x, s.c[5] = q.y, s.c[6] = q.z;
    }
}

__global__ void __launch_bounds__(256) mdp_step_kernel(MdpDev m) {
    __shared__ ilx_term s_terms[3 * kMaxTerms];
    __shared__ int32_t s_ids[kMaxIds];
    const ilx_mdp& d = m.d;
    const int NT = d.num_terminations, NR = d.num_rewards, NO = d.num_observations;
    const int lane = threadIdx.x & 63;
    const int64_t N = d.num_envs;
    const int64_t wave_env = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t env = wave_env < N ? wave_env : N - 1;  // a surplus wave gathers a valid row, then leaves
    const int J = d.num_joints, B = d.num_bodies, CB = d.num_sensor_bodies, A = d.action_dim, CD = d.command_dim;
    const int R = d.num_rays;
    const bool ph_term = d.phases & ILX_PHASE_TERMINATION, ph_rew = d.phases & ILX_PHASE_REWARD,
               ph_reset = d.phases & ILX_PHASE_RESET, ph_obs = d.phases & ILX_PHASE_OBSERVATION;
    const bool has_cmd_term = d.command_kind != ILX_CMD_NONE;
    const bool ph_cmd = (d.phases & ILX_PHASE_COMMAND) && has_cmd_term;
    const bool cmd_state = has_cmd_term && (d.phases & (ILX_PHASE_COMMAND | ILX_PHASE_RESET));

    // ================= GATHER: every load is issued before the first use =================
    Env e{m, s_ids, env, lane};
    Row& r = e.r;
    const float* rs = d.root_state_w + 13 * env;
    const float rs_lane = lane < 13 ? rs[lane] : 0.0f;
    if ((m.need & kNeedJoint) && lane < J) {
        const int64_t i = env * J + lane;
        r.q = d.joint_pos[i], r.qd = d.joint_vel[i], r.qdd = d.joint_acc[i];
        r.tau_a = d.applied_torque[i], r.tau_c = d.computed_torque[i];
        r.q0 = d.default_joint_pos[i], r.qd0 = d.default_joint_vel[i];
        const float2 lim = *reinterpret_cast<const float2*>(d.soft_joint_pos_limits + 2 * i);
        r.lim_lo = lim.x, r.lim_hi = lim.y;
        r.vlim = d.soft_joint_vel_limits[i];
    }
    if ((m.need & kNeedAction) && lane < A) {
        r.act = d.action[env * A + lane];
        if (d.prev_action) r.pact = d.prev_action[env * A + lane];
    }
    if ((m.need & kNeedTimers) && lane < CB) {
        const int64_t i = env * CB + lane;
        r.air = d.current_air_time[i], r.last_air = d.last_air_time[i], r.contact_t = d.current_contact_time[i];
    }
    if ((m.need & kNeedBodyVel) && lane < B) {
        const float* v = d.body_state_w + 13 * (env * B + lane) + 7;
        r.bvx = v[0], r.bvy = v[1];
    }
    f3 bacc = {0, 0, 0};
    if ((m.need & kNeedBodyAcc) && lane < B) bacc = load3(d.body_lin_acc_w + 3 * (env * B + lane));
    if (d.command && lane < CD) r.cmd = d.command[env * CD + lane];
    float ep_sum = 0.0f;  // lane i: episode sum of reward term i
    if ((ph_rew || ph_reset) && lane < NR) ep_sum = d.episode_sums[static_cast<int64_t>(lane) * N + env];
    float rz[kRayRegs] = {0, 0, 0, 0};
    float sensor_z = 0.0f;
    if (m.need & kNeedRays) {
        sensor_z = d.sensor_pos_w[3 * env + 2];
#pragma unroll
        for (int it = 0; it < kRayRegs; ++it) {
            const int k = it * 64 + lane;
            if (k < R) rz[it] = d.ray_hits_w[3 * (env * R + k) + 2];
        }
    }
    int64_t ep_len = d.episode_length_buf ? d.episode_length_buf[env] : 0;
    CmdState cs;
    if (d.time_left) cs.time_left = d.time_left[env];
    if (d.command_counter) cs.counter = d.command_counter[env];
    if (cmd_state) {
        cs.metric0 = d.command_metrics[env], cs.metric1 = d.command_metrics[N + env];
        if (d.command_kind == ILX_CMD_VELOCITY) {
            cs.heading_target = d.heading_target[env];
            cs.is_heading = d.is_heading_env[env] != 0;
            cs.is_standing = d.is_standing_env[env] != 0;
        }
    }
    // peak contact force over the history per sensor body (last: the longest dependent chain of the gather)
    if ((m.need & kNeedForces) && lane < CB) {
        const int H = d.history_len;
        float mx = -__builtin_inff();
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const float* f = d.net_forces_w_history + 3 * ((env * H + h) * CB + lane);
            mx = fmaxf(mx, norm3(f[0], f[1], f[2]));
        }
        r.fpeak = mx;
    }
    r.bacc = norm3(bacc);

    // term tables and index lists -> LDS; their loads are in flight together with the gather above
    {
        const int32_t* src = reinterpret_cast<const int32_t*>(m.terms);
        int32_t* dst = reinterpret_cast<int32_t*>(s_terms);
        const int words = (NT + NR + NO) * static_cast<int>(sizeof(ilx_term) / 4);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < d.num_ids; i += 256) s_ids[i] = m.ids[i];
    }
    __syncthreads();
    if (wave_env >= N) return;  // wave-uniform

    // root frame quantities (articulation_data.py:411-425, 502-518), uniform across the wave
    e.pos = {lane_const(rs_lane, 0), lane_const(rs_lane, 1), lane_const(rs_lane, 2)};
    e.quat = {lane_const(rs_lane, 3), lane_const(rs_lane, 4), lane_const(rs_lane, 5), lane_const(rs_lane, 6)};
    e.lin_w = {lane_const(rs_lane, 7), lane_const(rs_lane, 8), lane_const(rs_lane, 9)};
    e.ang_w = {lane_const(rs_lane, 10), lane_const(rs_lane, 11), lane_const(rs_lane, 12)};
    e.lin_b = quat_rotate_inverse(e.quat, e.lin_w);
    e.ang_b = quat_rotate_inverse(e.quat, e.ang_w);
    e.grav_b = quat_rotate_inverse(e.quat, f3{d.gravity_dir[0], d.gravity_dir[1], d.gravity_dir[2]});
    {
        const f3 fwd = quat_apply(e.quat, f3{1.0f, 0.0f, 0.0f});
        e.heading = atan2f(fwd.y, fwd.x);
    }
    e.time_left = cs.time_left, e.command_counter = cs.counter;
    if (has_cmd_term) {
#pragma unroll
        for (int k = 0; k < 7; ++k) cs.c[k] = lane_const(r.cmd, k);
    }
    const float dt = d.step_dt;

    // ================= terminations (manager_based_rl_env.py:199-202, termination_manager.py:151-174)
    bool terminated = false, time_out = false;
    uint32_t term_bits = 0;
    if (ph_term) {
        ep_len += 1;
        for (int i = 0; i < NT; ++i) {
            const Term t = fetch_term(s_terms + i);
            const bool v = eval_termination(e, t, ep_len);
            term_bits |= v ? (1u << i) : 0u;
            if (t.flags & 1) time_out = time_out || v; else terminated = terminated || v;
        }
        if (lane < NT) d.term_dones[static_cast<int64_t>(lane) * N + env] = (term_bits >> lane) & 1u;
        if (lane == 0) {
            d.terminated[env] = terminated;
            d.time_outs[env] = time_out;
            if (d.reset_buf) d.reset_buf[env] = terminated || time_out;
        }
    } else if (d.terminated) {
        terminated = d.terminated[env] != 0;
        time_out = d.time_outs[env] != 0;
        const bool mine = lane < NT && d.term_dones[static_cast<int64_t>(lane) * N + env] != 0;
        term_bits = static_cast<uint32_t>(__ballot(mine));
    }
    const bool done = terminated || time_out;

    // ================= rewards (reward_manager.py:128-157)
    bool sum_dirty = false;
    if (ph_rew) {
        float total = 0.0f, my_step = 0.0f;
        for (int i = 0; i < NR; ++i) {
            const Term t = fetch_term(s_terms + NT + i);
            if (t.weight == 0.0f) continue;  // :145-146 (episode sums and step_reward keep their values)
            const float value = eval_reward(e, t, terminated, time_out, term_bits) * t.weight * dt;
            total += value;
            if (lane == i) {
                ep_sum += value;
                my_step = div_(value, dt);
                sum_dirty = true;
            }
        }
        if (sum_dirty) d.step_reward[env * NR + lane] = my_step;
        if (lane == 0) d.reward_buf[env] = total;
    }

    // ================= reset bookkeeping of the managers for done envs (manager_based_rl_env.py:364-391)
    bool cmd_dirty = false;
    if (ph_reset && done) {
        if (lane < NR) {  // reward_manager.py:115-121
            if (d.log_sums) atomicAdd(d.log_sums + lane, ep_sum);
            ep_sum = 0.0f;
            sum_dirty = true;
        }
        if (d.log_sums) {
            if (lane < NT && ((term_bits >> lane) & 1u)) atomicAdd(d.log_sums + NR + lane, 1.0f);  // termination_manager.py:142-144
            if (lane == 0) atomicAdd(d.log_sums + NR + NT + 2, 1.0f);
        }
        if (has_cmd_term) {  // command_manager.py:120-149
            if (d.log_sums && lane < 2) atomicAdd(d.log_sums + NR + NT + lane, lane == 0 ? cs.metric0 : cs.metric1);
            cs.metric0 = cs.metric1 = 0.0f;
            cs.counter = 0;
            command_resample(d, env, d.reset_command_samples, 16u, cs);
            cmd_dirty = true;
        }
        ep_len = 0;  // manager_based_rl_env.py:391
    }
    if (sum_dirty) d.episode_sums[static_cast<int64_t>(lane) * N + env] = ep_sum;
    if (lane == 0 && d.episode_length_buf && (ph_term || (ph_reset && done))) d.episode_length_buf[env] = ep_len;

    // ================= command term (command_manager.py:151-166)
    if (ph_cmd) {
        if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:111-121
            cs.metric0 = cs.metric0 + div_(norm2(cs.c[0] - e.lin_b.x, cs.c[1] - e.lin_b.y), d.max_command_step);
            cs.metric1 = cs.metric1 + div_(fabsf(cs.c[2] - e.ang_b.z), d.max_command_step);
        } else {  // pose_command.py:92-108 (metrics are overwritten each step)
            f3 pw;
            f4 qw;
            combine_frame_transforms(e.pos, e.quat, f3{cs.c[0], cs.c[1], cs.c[2]}, f4{cs.c[3], cs.c[4], cs.c[5], cs.c[6]}, pw, qw);
            const float* b = d.body_state_w + 13 * (env * B + d.pose_body_index);
            const f3 aa = axis_angle_from_quat(quat_error(qw, f4{b[3], b[4], b[5], b[6]}));
            cs.metric0 = norm3(b[0] - pw.x, b[1] - pw.y, b[2] - pw.z);
            cs.metric1 = norm3(aa);
        }
        cs.time_left = cs.time_left - dt;
        if (cs.time_left <= 0.0f) command_resample(d, env, d.command_samples, 0u, cs);
        if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:140-160
            if (d.heading_command && cs.is_heading) {
                const float err = wrap_to_pi(cs.heading_target - e.heading);
                cs.c[2] = fminf(fmaxf(d.heading_control_stiffness * err, d.cmd_range[3][0]), d.cmd_range[3][1]);
            }
            if (cs.is_standing) cs.c[0] = cs.c[1] = cs.c[2] = 0.0f;
        }
        cmd_dirty = true;
    }
    if (cmd_dirty) {
        const int width = d.command_kind == ILX_CMD_VELOCITY ? 3 : 7;
        float mine = r.cmd;
#pragma unroll
        for (int k = 0; k < 7; ++k) mine = (lane == k && k < width) ? cs.c[k] : mine;
        r.cmd = mine;
        if (lane < width) d.command[env * CD + lane] = mine;
        if (lane == 0) {
            d.time_left[env] = cs.time_left;
            d.command_counter[env] = cs.counter;
            d.command_metrics[env] = cs.metric0;
            d.command_metrics[N + env] = cs.metric1;
            if (d.command_kind == ILX_CMD_VELOCITY) {
                d.heading_target[env] = cs.heading_target;
                d.is_heading_env[env] = cs.is_heading;
                d.is_standing_env[env] = cs.is_standing;
            }
        }
    }

    // ================= observations (observation_manager.py:260-335)
    if (ph_obs) {
        int col = 0;
        float* row = d.obs + env * d.obs_dim;
        for (int i = 0; i < NO; ++i) {
            const Term t = fetch_term(s_terms + NT + NR + i);
            const int dim = obs_dim_of(d, t);
            auto finish = [&](float v, int k) {
                if (t.flags & ILX_OBS_FLAG_NOISE) {  // noise_model.py:63-64: data + rand * (max - min) + min
                    const float u = d.noise_u ? d.noise_u[env * d.obs_dim + col + k]
                                              : philox_uniform(d.seed, d.step_index, static_cast<uint64_t>(env), col + k);
                    v = v + u * t.p1 + t.p0;
                }
                if (t.flags & ILX_OBS_FLAG_CLIP) v = fminf(fmaxf(v, t.p2), t.p3);  // :315-316
                if (t.flags & ILX_OBS_FLAG_SCALE) v = v * t.weight;                // :317-318
                row[col + k] = v;
            };
            if (t.kind == ILX_OBS_HEIGHT_SCAN) {  // observations.py:165-173
#pragma unroll
                for (int it = 0; it < kRayRegs; ++it) {
                    const int k = it * 64 + lane;
                    if (k < dim) finish(sensor_z - rz[it] - t.aux, k);
                }
                for (int k = kRayRegs * 64 + lane; k < dim; k += 64)
                    finish(sensor_z - d.ray_hits_w[3 * (env * R + k) + 2] - t.aux, k);
            } else {
                const float v = eval_obs_row(e, t, dim);
                if (lane < dim) finish(v, lane);
            }
            col += dim;
        }
    }
}

}  // namespace ilx

using namespace ilx;

struct ilx_mdp_plan_s {
    ilx_term* d_terms = nullptr;  // [terminations | rewards | observations]
    int32_t* d_ids = nullptr;
    int nt = 0, nr = 0, no = 0, nids = 0;
    int obs_dim = 0;
    // data the term tables read, per phase group (checked against NULL at every step)
    bool t_contact = false, t_joint = false, t_cmdstate = false;
    bool r_contact = false, r_timers = false, r_body_state = false, r_body_vel = false, r_body_acc = false, r_joint = false,
         r_action = false, r_rays = false, r_command = false;
    bool o_joint = false, o_rays = false, o_action = false, o_command = false;
};

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
    ILX_REQUIRE(d.num_ids >= 0 && d.num_ids <= kMaxIds, ILX_ESHAPE, "ilx_mdp_plan_create: %d index-list entries (max %d)",
                d.num_ids, kMaxIds);
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
    for (int i = 0; i < d.num_observations; ++i) {
        const int k = d.observations[i].kind;
        p->o_joint |= k >= ILX_OBS_JOINT_POS && k <= ILX_OBS_JOINT_VEL_REL;
        p->o_rays |= k == ILX_OBS_HEIGHT_SCAN;
        p->o_action |= k == ILX_OBS_LAST_ACTION;
        p->o_command |= k == ILX_OBS_GENERATED_COMMANDS;
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
        rc = check_hip(hipMalloc(&p->d_ids, d.num_ids * sizeof(int32_t)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_ids, d.ids, d.num_ids * sizeof(int32_t), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    if (rc != ILX_OK) {
        ilx_mdp_plan_destroy(p);
        return rc;
    }
    *out = p;
    return ILX_OK;
}

extern "C" void ilx_mdp_plan_destroy(ilx_mdp_plan_t plan) {
    if (!plan) return;
    if (plan->d_terms) (void)hipFree(plan->d_terms);
    if (plan->d_ids) (void)hipFree(plan->d_ids);
    delete plan;
}

extern "C" int ilx_mdp_step(ilx_mdp_plan_t plan, const ilx_mdp* desc, void* stream) {
    ILX_REQUIRE(plan && desc, ILX_EINVAL, "ilx_mdp_step: NULL argument");
    const ilx_mdp& d = *desc;
    ILX_REQUIRE(d.num_envs >= 0, ILX_EINVAL, "ilx_mdp_step: negative env count");
    ILX_REQUIRE(d.num_terminations == plan->nt && d.num_rewards == plan->nr && d.num_observations == plan->no &&
                    d.obs_dim == plan->obs_dim,
                ILX_EINVAL, "ilx_mdp_step: term counts differ from the plan");
    if (d.num_envs == 0 || d.phases == 0) return ILX_OK;
    ILX_REQUIRE(d.root_state_w, ILX_EINVAL, "ilx_mdp_step: root_state_w is NULL");
    // one register-file column per joint / body / action / command entry
    ILX_REQUIRE(d.num_joints >= 0 && d.num_joints <= kMaxRow && d.num_bodies >= 0 && d.num_bodies <= kMaxRow &&
                    d.num_sensor_bodies >= 0 && d.num_sensor_bodies <= kMaxRow && d.action_dim >= 0 &&
                    d.action_dim <= kMaxRow && d.command_dim >= 0 && d.command_dim <= kMaxRow,
                ILX_ESHAPE, "ilx_mdp_step: joints / bodies / sensor bodies / action / command widths above %d are not supported",
                kMaxRow);
    ILX_REQUIRE(d.history_len >= 0 && d.num_rays >= 0, ILX_EINVAL, "ilx_mdp_step: negative history length or ray count");
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
        ILX_REQUIRE(!plan->o_rays || (d.ray_hits_w && d.sensor_pos_w), ILX_EINVAL, "ilx_mdp_step: height_scan needs ray_hits_w and sensor_pos_w");
        ILX_REQUIRE(!plan->o_action || d.action, ILX_EINVAL, "ilx_mdp_step: last_action needs the action tensor");
        ILX_REQUIRE(!plan->o_command || d.command, ILX_EINVAL, "ilx_mdp_step: generated_commands needs the command tensor");
    }
    if ((d.phases & ILX_PHASE_COMMAND) && d.command_kind == ILX_CMD_POSE)
        ILX_REQUIRE(d.body_state_w && d.pose_body_index >= 0 && d.pose_body_index < d.num_bodies, ILX_EINVAL,
                    "ilx_mdp_step: the pose command needs body_state_w and a valid body index");
    const bool pt = d.phases & ILX_PHASE_TERMINATION, pr = d.phases & ILX_PHASE_REWARD, po = d.phases & ILX_PHASE_OBSERVATION;
    MdpDev m;
    m.d = d;
    m.terms = plan->d_terms;
    m.ids = plan->d_ids;
    m.need = 0;
    if ((pt && plan->t_joint) || (pr && plan->r_joint) || (po && plan->o_joint)) m.need |= kNeedJoint;
    if ((pt && plan->t_contact) || (pr && plan->r_contact)) m.need |= kNeedForces;
    if (pr && plan->r_timers) m.need |= kNeedTimers;
    if (pr && plan->r_body_vel) m.need |= kNeedBodyVel;
    if (pr && plan->r_body_acc) m.need |= kNeedBodyAcc;
    if ((pr && plan->r_action) || (po && plan->o_action)) m.need |= kNeedAction;
    if (po && plan->o_rays) m.need |= kNeedRays;
    m.d.terminations = m.d.rewards = m.d.observations = nullptr;
    m.d.ids = nullptr;
    const unsigned blocks = static_cast<unsigned>((d.num_envs + 3) / 4);
    hipLaunchKernelGGL(mdp_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), m);
    ILX_LAUNCH_CHECK("ilx_mdp_step");
    return ILX_OK;
}
