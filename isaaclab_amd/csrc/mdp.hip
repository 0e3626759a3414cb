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
#include "common.h"
#include "math_dev.h"
#include "philox.h"

#include "../../include/isaaclab_amd_mdp.h"

#include <cstring>
#include <vector>

namespace ilx {

constexpr int kMaxTerms = 32;      // per manager
constexpr int kE = 16;             // environments per block
constexpr int kBlock = 256;        // threads per block (measured on the ANYmal-C plan, envs x threads -> us:
                                   // 8x256 66, 16x256 46, 32x256 56, 16x512 77, 32x512 45)
constexpr int kWaves = kBlock / 64;
constexpr int kEmitLanes = kBlock / kE;  // threads that write one env's observation row
constexpr int kMaxLdsBytes = 64 * 1024;

enum Need : uint32_t {
    kNeedJoint = 1u, kNeedForces = 2u, kNeedTimers = 4u, kNeedBodyVel = 8u, kNeedBodyAcc = 16u, kNeedAction = 32u,
    kNeedRays = 64u,
};

// Word (4-byte) offsets into the block's LDS image; rows are [env][column] unless stated.
struct Layout {
    int terms, ids;
    int q, qd, qdd, tau_a, tau_c, q0, qd0, vlim, lim;  // joints (lim: 2 per joint)
    int act, pact;                                     // action columns
    int air, last_air, contact_t, fpeak;               // sensor bodies
    int bvx, bvy, bacc;                                // asset bodies
    int cmd, root;                                     // command row, 13-wide root state
    int eps, srew;                                     // episode sums [term][env], step reward [env][term]
    int ptotal;                                        // reward partial totals [wave][env]
    int derived;                                       // [env][kDerived]
    int bits;                                          // term bit mask per env, then done flag per env
    int logp;                                          // per-block partial sums of the logging vector
    int colterm, tstart;                               // observation columns: term index per column (u8), first column per term
    int total;
};
constexpr int kDerived = 10;  // lin_b(3) ang_b(3) grav_b(3) sensor_z

struct MdpDev {
    ilx_mdp d;                // by-value copy; the HOST pointers inside are nulled
    const ilx_term* terms;    // device: [terminations | rewards | observations]
    const int32_t* ids;       // device
    const uint32_t* cols;     // device: [ceil(obs_dim / 4) packed u8 term indices | first column per obs term]
    uint32_t need;
    int any_noise;            // some observation term draws noise
    Layout L;
};

// One environment as seen by its lane in the DECIDE stage.
struct Env {
    const MdpDev& m;
    const float* S;      // LDS image
    const int32_t* ids;  // LDS copy of the index lists
    int e;               // env slot in the block
    int64_t env;
    f3 pos, lin_w, ang_w, lin_b, ang_b, grav_b;
    f4 quat;
    float heading;
    float time_left;
    int64_t command_counter;

    ILX_DEV int id(const ilx_term& t, int k) const { return t.ids_cnt < 0 ? k : ids[t.ids_off + k]; }
    ILX_DEV int id2(const ilx_term& t, int k) const { return t.ids2_cnt < 0 ? k : ids[t.ids2_off + k]; }
    ILX_DEV int jcount(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_joints : t.ids_cnt; }
    ILX_DEV int bcount_sensor(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_sensor_bodies : t.ids_cnt; }
    ILX_DEV int bcount_asset(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_bodies : t.ids_cnt; }
    ILX_DEV float joint(int off, int j) const { return S[off + e * m.d.num_joints + j]; }
    ILX_DEV float sbody(int off, int b) const { return S[off + e * m.d.num_sensor_bodies + b]; }
    ILX_DEV float abody(int off, int b) const { return S[off + e * m.d.num_bodies + b]; }
    ILX_DEV float cmd(int k) const { return S[m.L.cmd + e * m.d.command_dim + k]; }
    template <typename F>
    ILX_DEV float joint_sum(const ilx_term& t, F f) const {
        const int n = jcount(t);
        float acc = 0.0f;
        // four elements' reads in flight together; the additions keep their order (the sum is compared bit for bit)
        int k = 0;
        for (; k + 4 <= n; k += 4) {
            const float f0 = f(id(t, k)), f1 = f(id(t, k + 1)), f2 = f(id(t, k + 2)), f3 = f(id(t, k + 3));
            acc += f0;
            acc += f1;
            acc += f2;
            acc += f3;
        }
        for (; k < n; ++k) acc += f(id(t, k));
        return acc;
    }
    template <typename F>
    ILX_DEV bool joint_any(const ilx_term& t, F f) const {
        const int n = jcount(t);
        bool hit = false;
        for (int k = 0; k < n; ++k) hit = hit | f(id(t, k));  // no short circuit: the reads do not wait for each other
        return hit;
    }
};

// --------------------------------------------------------------------------------------------------
ILX_DEV bool eval_termination(const Env& e, const ilx_term& t, int64_t ep_len) {
    const ilx_mdp& d = e.m.d;
    const Layout& L = e.m.L;
    switch (t.kind) {
        case ILX_TERM_TIME_OUT:  // terminations.py:30-32
            return ep_len >= d.max_episode_length;
        case ILX_TERM_COMMAND_RESAMPLE:  // :35-42
            return (e.time_left <= t.p[0]) && (e.command_counter == static_cast<int64_t>(t.p[1]));
        case ILX_TERM_BAD_ORIENTATION:  // :50-59
            return fabsf(acosf(-e.grav_b.z)) > t.p[0];
        case ILX_TERM_ROOT_HEIGHT_BELOW_MINIMUM:  // :62-72
            return e.pos.z < t.p[0];
        case ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT:  // :90-105
            return e.joint_any(t, [&](int j) { const float q = e.joint(L.q, j); return (q > t.p[1]) || (q < t.p[0]); });
        case ILX_TERM_JOINT_VEL_OUT_OF_LIMIT:  // :108-114
            return e.joint_any(t, [&](int j) { return fabsf(e.joint(L.qd, j)) > e.joint(L.vlim, j); });
        case ILX_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT:  // :117-124
            return e.joint_any(t, [&](int j) { return fabsf(e.joint(L.qd, j)) > t.p[0]; });
        case ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT:  // :127-142, torch.isclose(rtol=1e-5, atol=1e-8)
            return e.joint_any(t, [&](int j) {
                const float a = e.joint(L.tau_c, j), b = e.joint(L.tau_a, j);
                return (a == b) || (fabsf(a - b) <= (1.0e-8f + 1.0e-5f * fabsf(b)));
            });
        case ILX_TERM_ILLEGAL_CONTACT: {  // :150-158
            const int n = e.bcount_sensor(t);
            bool hit = false;
            for (int k = 0; k < n; ++k) hit = hit | (e.sbody(L.fpeak, e.id(t, k)) > t.p[0]);
            return hit;
        }
    }
    return false;
}

ILX_DEV float eval_reward(const Env& e, const ilx_term& t, bool terminated, bool time_out, uint32_t term_bits) {
    const ilx_mdp& d = e.m.d;
    const Layout& L = e.m.L;
    const int B = d.num_bodies, A = d.action_dim;
    switch (t.kind) {
        case ILX_REW_IS_ALIVE: return terminated ? 0.0f : 1.0f;       // rewards.py:31-33
        case ILX_REW_IS_TERMINATED: return terminated ? 1.0f : 0.0f;  // :36-38
        case ILX_REW_IS_TERMINATED_TERM:                              // :41-68
            return static_cast<float>(__popc(term_bits & static_cast<uint32_t>(t.flags))) * (time_out ? 0.0f : 1.0f);
        case ILX_REW_LIN_VEL_Z_L2: return e.lin_b.z * e.lin_b.z;                              // :76-80
        case ILX_REW_ANG_VEL_XY_L2: return e.ang_b.x * e.ang_b.x + e.ang_b.y * e.ang_b.y;    // :83-87
        case ILX_REW_FLAT_ORIENTATION_L2: return e.grav_b.x * e.grav_b.x + e.grav_b.y * e.grav_b.y;  // :90-97
        case ILX_REW_BASE_HEIGHT_L2: {  // :100-122
            float target = t.p[0];
            if (t.flags & 1) {
                float acc = 0.0f;
                for (int k = 0; k < d.num_rays; ++k) acc += d.ray_hits_w[3 * (e.env * d.num_rays + k) + 2];
                target = t.p[0] + div_(acc, static_cast<float>(d.num_rays));
            }
            const float dz = e.pos.z - target;
            return dz * dz;
        }
        case ILX_REW_BODY_LIN_ACC_L2: {  // :125-128
            const int n = e.bcount_asset(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc += e.abody(L.bacc, e.id(t, k));
            return acc;
        }
        case ILX_REW_JOINT_TORQUES_L2:  // :136-143
            return e.joint_sum(t, [&](int j) { const float v = e.joint(L.tau_a, j); return v * v; });
        case ILX_REW_JOINT_VEL_L1:  // :146-150
            return e.joint_sum(t, [&](int j) { return fabsf(e.joint(L.qd, j)); });
        case ILX_REW_JOINT_VEL_L2:  // :153-160
            return e.joint_sum(t, [&](int j) { const float v = e.joint(L.qd, j); return v * v; });
        case ILX_REW_JOINT_ACC_L2:  // :163-170
            return e.joint_sum(t, [&](int j) { const float v = e.joint(L.qdd, j); return v * v; });
        case ILX_REW_JOINT_DEVIATION_L1:  // :173-179
            return e.joint_sum(t, [&](int j) { return fabsf(e.joint(L.q, j) - e.joint(L.q0, j)); });
        case ILX_REW_JOINT_POS_LIMITS:  // :182-196
            return e.joint_sum(t, [&](int j) {
                const float q = e.joint(L.q, j);
                const float lo = e.S[L.lim + 2 * (e.e * d.num_joints + j)], hi = e.S[L.lim + 2 * (e.e * d.num_joints + j) + 1];
                return -fminf(q - lo, 0.0f) + fmaxf(q - hi, 0.0f);
            });
        case ILX_REW_JOINT_VEL_LIMITS:  // :199-218
            return e.joint_sum(t, [&](int j) {
                return fminf(fmaxf(fabsf(e.joint(L.qd, j)) - e.joint(L.vlim, j) * t.p[0], 0.0f), 1.0f);
            });
        case ILX_REW_APPLIED_TORQUE_LIMITS:  // :226-242
            return e.joint_sum(t, [&](int j) { return fabsf(e.joint(L.tau_a, j) - e.joint(L.tau_c, j)); });
        case ILX_REW_ACTION_RATE_L2: {  // :245-247
            float acc = 0.0f;
            for (int k = 0; k < A; ++k) {
                const float v = e.S[L.act + e.e * A + k] - e.S[L.pact + e.e * A + k];
                acc += v * v;
            }
            return acc;
        }
        case ILX_REW_ACTION_L2: {  // :250-252
            float acc = 0.0f;
            for (int k = 0; k < A; ++k) {
                const float v = e.S[L.act + e.e * A + k];
                acc += v * v;
            }
            return acc;
        }
        case ILX_REW_UNDESIRED_CONTACTS: {  // :260-268 (count of bodies above the threshold)
            const int n = e.bcount_sensor(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc += (e.sbody(L.fpeak, e.id(t, k)) > t.p[0]) ? 1.0f : 0.0f;
            return acc;
        }
        case ILX_REW_CONTACT_FORCES: {  // :271-279
            const int n = e.bcount_sensor(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc += fmaxf(e.sbody(L.fpeak, e.id(t, k)) - t.p[0], 0.0f);
            return acc;
        }
        case ILX_REW_TRACK_LIN_VEL_XY_EXP: {  // :287-298
            const float dx = e.cmd(0) - e.lin_b.x, dy = e.cmd(1) - e.lin_b.y;
            return expf(div_(-(dx * dx + dy * dy), t.p[0]));
        }
        case ILX_REW_TRACK_ANG_VEL_Z_EXP: {  // :301-309
            const float dz = e.cmd(2) - e.ang_b.z;
            return expf(div_(-(dz * dz), t.p[0]));
        }
        case ILX_REW_FEET_AIR_TIME: {  // velocity/mdp/rewards.py:25-44
            const int n = e.bcount_sensor(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) {
                const int b = e.id(t, k);
                const float cc = e.sbody(L.contact_t, b);
                const bool first = (cc > 0.0f) && (cc < t.p[1]);  // contact_sensor.py:200-203
                acc += (e.sbody(L.last_air, b) - t.p[0]) * (first ? 1.0f : 0.0f);
            }
            return acc * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED: {  // velocity/mdp/rewards.py:47-66
            const int n = e.bcount_sensor(t);
            float cnt = 0.0f;
            for (int k = 0; k < n; ++k) cnt += (e.sbody(L.contact_t, e.id(t, k)) > 0.0f) ? 1.0f : 0.0f;
            const bool single = cnt == 1.0f;
            float mn = __builtin_inff();
            for (int k = 0; k < n; ++k) {
                const int b = e.id(t, k);
                const float cc = e.sbody(L.contact_t, b);
                const float mode = (cc > 0.0f) ? cc : e.sbody(L.air, b);
                mn = fminf(mn, single ? mode : 0.0f);
            }
            return fminf(mn, t.p[0]) * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_SLIDE: {  // velocity/mdp/rewards.py:69-83
            const int n = e.bcount_sensor(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) {
                const bool contact = e.sbody(L.fpeak, e.id(t, k)) > 1.0f;
                const int ab = e.id2(t, k);
                acc += norm2(e.abody(L.bvx, ab), e.abody(L.bvy, ab)) * (contact ? 1.0f : 0.0f);
            }
            return acc;
        }
        case ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {  // velocity/mdp/rewards.py:86-96
            const f3 v = quat_rotate_inverse(yaw_quat(e.quat), e.lin_w);
            const float dx = e.cmd(0) - v.x, dy = e.cmd(1) - v.y;
            return expf(div_(-(dx * dx + dy * dy), t.p[0]));
        }
        case ILX_REW_TRACK_ANG_VEL_Z_WORLD_EXP: {  // velocity/mdp/rewards.py:99-106
            const float dz = e.cmd(2) - e.ang_w.z;
            return expf(div_(-(dz * dz), t.p[0]));
        }
        case ILX_REW_POSITION_COMMAND_ERROR:
        case ILX_REW_POSITION_COMMAND_ERROR_TANH: {  // reach/mdp/rewards.py:19-52
            const f3 rr = quat_apply(e.quat, f3{e.cmd(0), e.cmd(1), e.cmd(2)});
            const f3 des = {e.pos.x + rr.x, e.pos.y + rr.y, e.pos.z + rr.z};
            const float* b = d.body_state_w + 13 * (e.env * B + e.ids[t.ids_off]);
            const float dist = norm3(b[0] - des.x, b[1] - des.y, b[2] - des.z);
            return t.kind == ILX_REW_POSITION_COMMAND_ERROR ? dist : (1.0f - tanhf(div_(dist, t.p[0])));
        }
        case ILX_REW_ORIENTATION_COMMAND_ERROR: {  // reach/mdp/rewards.py:55-69
            const f4 des = quat_mul(e.quat, f4{e.cmd(3), e.cmd(4), e.cmd(5), e.cmd(6)});
            const float* b = d.body_state_w + 13 * (e.env * B + e.ids[t.ids_off]) + 3;
            return quat_error_magnitude(f4{b[0], b[1], b[2], b[3]}, des);
        }
        case ILX_REW_JOINT_POS_TARGET_L2:  // cartpole/mdp/rewards.py:19-26
            return e.joint_sum(t, [&](int j) { const float v = wrap_to_pi(e.joint(L.q, j)) - t.p[0]; return v * v; });
    }
    return 0.0f;
}

ILX_DEV int obs_dim_of(const ilx_mdp& d, const ilx_term& t) {
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
        case ILX_OBS_EXTERNAL: return d.external_obs_dim[static_cast<int>(t.aux)];
    }
    return 0;
}

// Column k of an observation term for env slot e (EMIT stage; everything except the height scan comes from LDS).
ILX_DEV float eval_obs(const MdpDev& m, const float* S, const int32_t* ids, int e, int64_t env, const ilx_term& t, int k) {
    const ilx_mdp& d = m.d;
    const Layout& L = m.L;
    const int J = d.num_joints;
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
            return der[9] - d.ray_hits_w[3 * (env * d.num_rays + k) + 2] - t.aux;
        case ILX_OBS_LAST_ACTION: return S[L.act + e * d.action_dim + k];          // :512-521
        case ILX_OBS_GENERATED_COMMANDS: return S[L.cmd + e * d.command_dim + k];  // :529-531
        case ILX_OBS_EXTERNAL: {  // a buffer another component keeps per env (:177-230: body wrench, IMU readings)
            const int slot = static_cast<int>(t.aux);
            return d.external_obs[slot][env * d.external_obs_dim[slot] + k];
        }
    }
    return 0.0f;
}

// Command term state of one env (held by the env's lane).
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

// coalesced copy of the block's chunk of an [env][width] array into LDS
ILX_DEV void stage_rows(float* dst, const float* src, int64_t env0, int ne, int width) {
    const float* p = src + env0 * width;
    for (int i = threadIdx.x; i < ne * width; i += kBlock) dst[i] = p[i];
}

// A block's chunk of an [env][width] array, taken in two steps: issue() puts every thread's loads in flight (two floats
// per thread cover rows up to 32 wide; indices past the chunk read its last element instead of branching), commit()
// writes them to LDS.  With all issues of the STAGE section ahead of all commits the global round trips overlap; one
// copy loop per array instead costs a full memory latency per array.
struct Staged {
    float a, b;
};
ILX_DEV Staged issue_rows(const float* src, int64_t env0, int ne, int width) {
    const float* p = src + env0 * width;
    const int n = ne * width, t = threadIdx.x;
    return {p[t < n ? t : n - 1], p[t + kBlock < n ? t + kBlock : n - 1]};
}
ILX_DEV void commit_rows(float* dst, Staged r, const float* src, int64_t env0, int ne, int width) {
    const int n = ne * width, t = threadIdx.x;
    if (t < n) dst[t] = r.a;
    if (t + kBlock < n) dst[t + kBlock] = r.b;
    const float* p = src + env0 * width;
    for (int i = t + 2 * kBlock; i < n; i += kBlock) dst[i] = p[i];
}

__global__ void __launch_bounds__(kBlock) mdp_step_kernel(MdpDev m) {
    extern __shared__ float S[];
    const ilx_mdp& d = m.d;
    const Layout& L = m.L;
    const int NT = d.num_terminations, NR = d.num_rewards, NO = d.num_observations;
    const int64_t N = d.num_envs;
    const int64_t env0 = static_cast<int64_t>(blockIdx.x) * kE;
    const int ne = static_cast<int>(N - env0 < kE ? N - env0 : kE);
    const int J = d.num_joints, B = d.num_bodies, CB = d.num_sensor_bodies, A = d.action_dim, CD = d.command_dim;
    const int tid = threadIdx.x;
    const bool ph_term = d.phases & ILX_PHASE_TERMINATION, ph_rew = d.phases & ILX_PHASE_REWARD,
               ph_reset = d.phases & ILX_PHASE_RESET, ph_obs = d.phases & ILX_PHASE_OBSERVATION;
    const bool has_cmd_term = d.command_kind != ILX_CMD_NONE;
    const bool ph_cmd = (d.phases & ILX_PHASE_COMMAND) && has_cmd_term;
    const bool cmd_state = has_cmd_term && (d.phases & (ILX_PHASE_COMMAND | ILX_PHASE_RESET));
    const ilx_term* terms = reinterpret_cast<const ilx_term*>(S + L.terms);
    const int32_t* ids = reinterpret_cast<const int32_t*>(S + L.ids);

    // ================= STAGE =================
    const bool sj = m.need & kNeedJoint, sa = m.need & kNeedAction, spa = sa && d.prev_action, st = m.need & kNeedTimers,
               sc = d.command != nullptr;
    const Staged zero{0.0f, 0.0f};
    const Staged r_root = issue_rows(d.root_state_w, env0, ne, 13);
    const Staged r_q = sj ? issue_rows(d.joint_pos, env0, ne, J) : zero, r_qd = sj ? issue_rows(d.joint_vel, env0, ne, J) : zero,
                 r_qdd = sj ? issue_rows(d.joint_acc, env0, ne, J) : zero, r_ta = sj ? issue_rows(d.applied_torque, env0, ne, J) : zero,
                 r_tc = sj ? issue_rows(d.computed_torque, env0, ne, J) : zero, r_q0 = sj ? issue_rows(d.default_joint_pos, env0, ne, J) : zero,
                 r_qd0 = sj ? issue_rows(d.default_joint_vel, env0, ne, J) : zero,
                 r_vl = sj ? issue_rows(d.soft_joint_vel_limits, env0, ne, J) : zero,
                 r_lim = sj ? issue_rows(d.soft_joint_pos_limits, env0, ne, 2 * J) : zero;
    const Staged r_act = sa ? issue_rows(d.action, env0, ne, A) : zero, r_pact = spa ? issue_rows(d.prev_action, env0, ne, A) : zero;
    const Staged r_air = st ? issue_rows(d.current_air_time, env0, ne, CB) : zero, r_lair = st ? issue_rows(d.last_air_time, env0, ne, CB) : zero,
                 r_ct = st ? issue_rows(d.current_contact_time, env0, ne, CB) : zero;
    const Staged r_cmd = sc ? issue_rows(d.command, env0, ne, CD) : zero;
    // the term tables (cold in L2 at the start of a launch) are fetched while the row loads above are in flight
    {
        const int32_t* src = reinterpret_cast<const int32_t*>(m.terms);
        int32_t* dst = reinterpret_cast<int32_t*>(S + L.terms);
        const int words = (NT + NR + NO) * static_cast<int>(sizeof(ilx_term) / 4);
        for (int i = tid; i < words; i += kBlock) dst[i] = src[i];
        int32_t* idst = reinterpret_cast<int32_t*>(S + L.ids);
        for (int i = tid; i < d.num_ids; i += kBlock) idst[i] = m.ids[i];
        uint32_t* cdst = reinterpret_cast<uint32_t*>(S + L.colterm);  // colterm and tstart are adjacent
        const int cwords = ((d.obs_dim + 3) >> 2) + NO;
        if (ph_obs)
            for (int i = tid; i < cwords; i += kBlock) cdst[i] = m.cols[i];
        for (int i = tid; i < NR + NT + 3; i += kBlock) S[L.logp + i] = 0.0f;
    }
    commit_rows(S + L.root, r_root, d.root_state_w, env0, ne, 13);
    if (sj) {
        commit_rows(S + L.q, r_q, d.joint_pos, env0, ne, J);
        commit_rows(S + L.qd, r_qd, d.joint_vel, env0, ne, J);
        commit_rows(S + L.qdd, r_qdd, d.joint_acc, env0, ne, J);
        commit_rows(S + L.tau_a, r_ta, d.applied_torque, env0, ne, J);
        commit_rows(S + L.tau_c, r_tc, d.computed_torque, env0, ne, J);
        commit_rows(S + L.q0, r_q0, d.default_joint_pos, env0, ne, J);
        commit_rows(S + L.qd0, r_qd0, d.default_joint_vel, env0, ne, J);
        commit_rows(S + L.vlim, r_vl, d.soft_joint_vel_limits, env0, ne, J);
        commit_rows(S + L.lim, r_lim, d.soft_joint_pos_limits, env0, ne, 2 * J);
    }
    if (sa) commit_rows(S + L.act, r_act, d.action, env0, ne, A);
    if (spa) commit_rows(S + L.pact, r_pact, d.prev_action, env0, ne, A);
    if (st) {
        commit_rows(S + L.air, r_air, d.current_air_time, env0, ne, CB);
        commit_rows(S + L.last_air, r_lair, d.last_air_time, env0, ne, CB);
        commit_rows(S + L.contact_t, r_ct, d.current_contact_time, env0, ne, CB);
    }
    if (sc) commit_rows(S + L.cmd, r_cmd, d.command, env0, ne, CD);
    if (ph_rew || ph_reset) {  // episode sums: [term][env] in memory and in LDS
        for (int i = tid; i < NR * kE; i += kBlock) {
            const int term = i / kE, e = i - term * kE;
            if (e < ne) S[L.eps + i] = d.episode_sums[static_cast<int64_t>(term) * N + env0 + e];
        }
    }
    if (m.need & kNeedBodyVel) {
        for (int i = tid; i < ne * B; i += kBlock) {
            const float* v = d.body_state_w + 13 * (env0 * B + i) + 7;
            S[L.bvx + i] = v[0], S[L.bvy + i] = v[1];
        }
    }
    if (m.need & kNeedBodyAcc) {
        for (int i = tid; i < ne * B; i += kBlock) S[L.bacc + i] = norm3(load3(d.body_lin_acc_w + 3 * (env0 * B + i)));
    }
    if (m.need & kNeedForces) {  // max over the history of |F| per (env, body): rewards.py:266 / terminations.py:157
        const int H = d.history_len;
        for (int i = tid; i < ne * CB; i += kBlock) {
            const int e = i / CB, b = i - e * CB;
            float mx = -__builtin_inff();
#pragma unroll 4
            for (int h = 0; h < H; ++h) mx = fmaxf(mx, norm3(load3(d.net_forces_w_history + 3 * (((env0 + e) * H + h) * CB + b))));
            S[L.fpeak + i] = mx;
        }
    }
    // per-env scalars of the DECIDE lanes (coalesced over the 16 lanes), loaded before the barrier
    const int wave = tid >> 6, lane = tid & 63;
    const bool decide = lane < ne;
    const int e = lane;
    const int64_t env = env0 + (decide ? e : 0);
    int64_t ep_len = 0;
    CmdState cs;
    float sensor_z = 0.0f;
    bool terminated = false, time_out = false;
    uint32_t term_bits = 0;
    if (decide) {
        if (d.episode_length_buf) ep_len = d.episode_length_buf[env];
        if (d.time_left) cs.time_left = d.time_left[env];
        if (d.command_counter) cs.counter = d.command_counter[env];
        if (cmd_state && wave == 0) {
            cs.metric0 = d.command_metrics[env], cs.metric1 = d.command_metrics[N + env];
            if (d.command_kind == ILX_CMD_VELOCITY) {
                cs.heading_target = d.heading_target[env];
                cs.is_heading = d.is_heading_env[env] != 0;
                cs.is_standing = d.is_standing_env[env] != 0;
            }
        }
        if ((m.need & kNeedRays) && wave == 0) sensor_z = d.sensor_pos_w[3 * env + 2];
        if (!ph_term && d.terminated) {
            terminated = d.terminated[env] != 0;
            time_out = d.time_outs[env] != 0;
            for (int i = 0; i < NT; ++i) term_bits |= d.term_dones[static_cast<int64_t>(i) * N + env] ? (1u << i) : 0u;
        }
    }
    __syncthreads();

    // ================= DECIDE: one lane per env; the four waves split the reward terms =================
    Env en{m, S, ids, e, env};
    const float dt = d.step_dt;
    if (decide) {
        const float* rs = S + L.root + 13 * e;
        en.pos = {rs[0], rs[1], rs[2]};
        en.quat = {rs[3], rs[4], rs[5], rs[6]};
        en.lin_w = {rs[7], rs[8], rs[9]};
        en.ang_w = {rs[10], rs[11], rs[12]};
        // articulation_data.py:411-425, 502-518
        en.lin_b = quat_rotate_inverse(en.quat, en.lin_w);
        en.ang_b = quat_rotate_inverse(en.quat, en.ang_w);
        en.grav_b = quat_rotate_inverse(en.quat, f3{d.gravity_dir[0], d.gravity_dir[1], d.gravity_dir[2]});
        en.time_left = cs.time_left, en.command_counter = cs.counter;
        // ---- terminations (manager_based_rl_env.py:199-202, termination_manager.py:151-174); every wave needs the result
        if (ph_term) {
            ep_len += 1;
            for (int i = 0; i < NT; ++i) {
                const ilx_term t = terms[i];
                const bool v = eval_termination(en, t, ep_len);
                term_bits |= v ? (1u << i) : 0u;
                if (t.flags & 1) time_out = time_out || v; else terminated = terminated || v;
            }
        }
        // ---- rewards (reward_manager.py:128-157): this wave's share of the terms
        if (ph_rew) {
            float partial = 0.0f;
            for (int i = wave; i < NR; i += kWaves) {
                const ilx_term t = terms[NT + i];
                if (t.weight == 0.0f) continue;  // :145-146 (episode sums and step_reward keep their values)
                const float value = eval_reward(en, t, terminated, time_out, term_bits) * t.weight * dt;
                partial += value;
                S[L.eps + i * kE + e] += value;
                S[L.srew + e * NR + i] = div_(value, dt);
            }
            S[L.ptotal + wave * kE + e] = partial;
        }
    }
    __syncthreads();
    if (decide && wave == 0) {
        const bool done = terminated || time_out;
        if (ph_term) {
            d.terminated[env] = terminated;
            d.time_outs[env] = time_out;
            if (d.reset_buf) d.reset_buf[env] = done;
        }
        if (ph_rew)
        {
            float total = S[L.ptotal + e];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) total += S[L.ptotal + w * kE + e];
            d.reward_buf[env] = total;
        }
        reinterpret_cast<uint32_t*>(S + L.bits)[e] = term_bits;
        reinterpret_cast<uint32_t*>(S + L.bits)[kE + e] = done ? 1u : 0u;
        if (has_cmd_term) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < CD) cs.c[k] = en.cmd(k);
        }
        // ---- reset bookkeeping of the managers for done envs (manager_based_rl_env.py:364-391)
        bool cmd_dirty = false;
        if (ph_reset && done) {
            // logging sums go through per-block partials in LDS (ds_add_f32), one global atomic per entry and block
            float* logp = S + L.logp;
            for (int i = 0; i < NR; ++i) {  // reward_manager.py:115-121
                atomicAdd(logp + i, S[L.eps + i * kE + e]);
                S[L.eps + i * kE + e] = 0.0f;
            }
            for (int i = 0; i < NT; ++i)  // termination_manager.py:142-144
                if (term_bits & (1u << i)) atomicAdd(logp + NR + i, 1.0f);
            atomicAdd(logp + NR + NT + 2, 1.0f);
            if (has_cmd_term) {  // command_manager.py:120-149
                atomicAdd(logp + NR + NT, cs.metric0);
                atomicAdd(logp + NR + NT + 1, cs.metric1);
                cs.metric0 = cs.metric1 = 0.0f;
                cs.counter = 0;
                command_resample(d, env, d.reset_command_samples, 16u, cs);
                cmd_dirty = true;
            }
            // action_manager.py:309-311: the action history of a done env restarts at zero (the observations of this
            // launch already see it; the rewards above used the step's own actions)
            for (int k = 0; k < d.action_dim; ++k) {
                if (d.action) d.action[env * d.action_dim + k] = 0.0f;
                if (d.prev_action) d.prev_action[env * d.action_dim + k] = 0.0f;
                if (sa) S[L.act + e * d.action_dim + k] = 0.0f;
                if (spa) S[L.pact + e * d.action_dim + k] = 0.0f;
            }
            ep_len = 0;  // manager_based_rl_env.py:391
        }
        if (d.episode_length_buf && (ph_term || (ph_reset && done))) d.episode_length_buf[env] = ep_len;
        // ---- command term (command_manager.py:151-166)
        if (ph_cmd) {
            if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:111-121
                cs.metric0 = cs.metric0 + div_(norm2(cs.c[0] - en.lin_b.x, cs.c[1] - en.lin_b.y), d.max_command_step);
                cs.metric1 = cs.metric1 + div_(fabsf(cs.c[2] - en.ang_b.z), d.max_command_step);
            } else {  // pose_command.py:92-108 (metrics are overwritten each step)
                f3 pw;
                f4 qw;
                combine_frame_transforms(en.pos, en.quat, f3{cs.c[0], cs.c[1], cs.c[2]}, f4{cs.c[3], cs.c[4], cs.c[5], cs.c[6]}, pw, qw);
                const float* b = d.body_state_w + 13 * (env * B + d.pose_body_index);
                const f3 aa = axis_angle_from_quat(quat_error(qw, f4{b[3], b[4], b[5], b[6]}));
                cs.metric0 = norm3(b[0] - pw.x, b[1] - pw.y, b[2] - pw.z);
                cs.metric1 = norm3(aa);
            }
            cs.time_left = cs.time_left - dt;
            if (cs.time_left <= 0.0f) command_resample(d, env, d.command_samples, 0u, cs);
            if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:140-160
                if (d.heading_command && cs.is_heading) {
                    const f3 fwd = quat_apply(en.quat, f3{1.0f, 0.0f, 0.0f});  // articulation_data.py:416-425
                    const float err = wrap_to_pi(cs.heading_target - atan2f(fwd.y, fwd.x));
                    cs.c[2] = fminf(fmaxf(d.heading_control_stiffness * err, d.cmd_range[3][0]), d.cmd_range[3][1]);
                }
                if (cs.is_standing) cs.c[0] = cs.c[1] = cs.c[2] = 0.0f;
            }
            cmd_dirty = true;
        }
        if (cmd_dirty) {
            const int width = d.command_kind == ILX_CMD_VELOCITY ? 3 : 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (k < width) {
                    S[L.cmd + e * CD + k] = cs.c[k];
                    d.command[env * CD + k] = cs.c[k];
                }
            }
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
        float* der = S + L.derived + kDerived * e;
        der[0] = en.lin_b.x, der[1] = en.lin_b.y, der[2] = en.lin_b.z;
        der[3] = en.ang_b.x, der[4] = en.ang_b.y, der[5] = en.ang_b.z;
        der[6] = en.grav_b.x, der[7] = en.grav_b.y, der[8] = en.grav_b.z;
        der[9] = sensor_z;
    }
    __syncthreads();

    // ================= EMIT =================
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(S + L.bits);
    if (ph_term) {
        for (int i = tid; i < NT * kE; i += kBlock) {
            const int term = i / kE, ee = i - term * kE;
            if (ee < ne) d.term_dones[static_cast<int64_t>(term) * N + env0 + ee] = (bits[ee] >> term) & 1u;
        }
    }
    if (ph_rew || ph_reset) {
        for (int i = tid; i < NR * kE; i += kBlock) {
            const int term = i / kE, ee = i - term * kE;
            if (ee >= ne) continue;
            const bool active = ph_rew && terms[NT + term].weight != 0.0f;
            if (active || (ph_reset && bits[kE + ee])) d.episode_sums[static_cast<int64_t>(term) * N + env0 + ee] = S[L.eps + i];
        }
        if (ph_rew) {
            for (int i = tid; i < ne * NR; i += kBlock) {
                const int term = i % NR;
                if (terms[NT + term].weight != 0.0f) d.step_reward[env0 * NR + i] = S[L.srew + i];
            }
        }
    }
    if (ph_reset && d.log_sums && tid < NR + NT + 3) {
        const float v = S[L.logp + tid];
        if (v != 0.0f) unsafeAtomicAdd(d.log_sums + tid, v);  // hardware float add at L2, no return value
    }
    // ---- observations (observation_manager.py:260-335).  The concatenated row is cut into groups of 4 columns;
    // 16 threads per env take group g, g + 16, ...; a group's noise is one Philox4x32 call.
    if (ph_obs) {
        const int ee = tid / kEmitLanes, g = tid % kEmitLanes;
        if (ee < ne) {
            const int64_t oenv = env0 + ee;
            const int O = d.obs_dim;
            float* row = d.obs + oenv * O;
            const uint32_t* colterm = reinterpret_cast<const uint32_t*>(S + L.colterm);
            const int32_t* tstart = reinterpret_cast<const int32_t*>(S + L.tstart);
            const ilx_term* oterms = terms + NT + NR;
            for (int cg = g; 4 * cg < O; cg += kEmitLanes) {
                const uint32_t packed = colterm[cg];
                int cur = static_cast<int>(packed & 255u);
                ilx_term t = oterms[cur];
                float u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (!d.noise_u && m.any_noise) {
                    const u4 r = philox4(d.seed, d.step_index, static_cast<uint64_t>(oenv), static_cast<uint32_t>(cg));
                    u[0] = u01(r.x), u[1] = u01(r.y), u[2] = u01(r.z), u[3] = u01(r.w);
                }
                if (t.kind == ILX_OBS_HEIGHT_SCAN && packed == static_cast<uint32_t>(cur) * 0x01010101u && 4 * cg + 3 < O) {
                    // Four columns of the height scan (most of a locomotion observation) with their hit heights read
                    // back to back; through the per-column path below each load sits in its own branch of the term switch
                    // and waits for the one before it.
                    const float* z = d.ray_hits_w + 3 * (oenv * d.num_rays + (4 * cg - tstart[cur])) + 2;
                    const float z0 = z[0], z1 = z[3], z2 = z[6], z3 = z[9];
                    float un[4] = {u[0], u[1], u[2], u[3]};
                    if ((t.flags & ILX_OBS_FLAG_NOISE) && d.noise_u) {
                        const float* nu = d.noise_u + oenv * O + 4 * cg;
                        un[0] = nu[0], un[1] = nu[1], un[2] = nu[2], un[3] = nu[3];
                    }
                    const float sensor_height = S[L.derived + kDerived * ee + 9];
                    float v[4] = {sensor_height - z0 - t.aux, sensor_height - z1 - t.aux, sensor_height - z2 - t.aux,
                                  sensor_height - z3 - t.aux};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (t.flags & ILX_OBS_FLAG_NOISE) v[c] = v[c] + un[c] * t.p[1] + t.p[0];
                        if (t.flags & ILX_OBS_FLAG_CLIP) v[c] = fminf(fmaxf(v[c], t.p[2]), t.p[3]);
                        if (t.flags & ILX_OBS_FLAG_SCALE) v[c] = v[c] * t.weight;
                        row[4 * cg + c] = v[c];
                    }
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = 4 * cg + c;
                    if (col >= O) break;
                    const int ti = static_cast<int>((packed >> (8 * c)) & 255u);
                    if (ti != cur) {
                        cur = ti;
                        t = oterms[cur];
                    }
                    float v = eval_obs(m, S, ids, ee, oenv, t, col - tstart[cur]);
                    if (t.flags & ILX_OBS_FLAG_NOISE) {  // noise_model.py:63-64: data + rand * (max - min) + min
                        const float uu = d.noise_u ? d.noise_u[oenv * O + col] : u[c];
                        v = v + uu * t.p[1] + t.p[0];
                    }
                    if (t.flags & ILX_OBS_FLAG_CLIP) v = fminf(fmaxf(v, t.p[2]), t.p[3]);  // :315-316
                    if (t.flags & ILX_OBS_FLAG_SCALE) v = v * t.weight;                    // :317-318
                    row[col] = v;
                }
            }
        }
    }
}

}  // namespace ilx

using namespace ilx;

struct ilx_mdp_plan_s {
    ilx_term* d_terms = nullptr;  // [terminations | rewards | observations]
    int32_t* d_ids = nullptr;
    uint32_t* d_cols = nullptr;  // packed term index per observation column, then the first column of every term
    bool any_noise = false;
    int nt = 0, nr = 0, no = 0, nids = 0;
    int obs_dim = 0;
    // data the term tables read, per phase group (checked against NULL at every step)
    bool t_contact = false, t_joint = false, t_cmdstate = false;
    bool r_contact = false, r_timers = false, r_body_state = false, r_body_vel = false, r_body_acc = false, r_joint = false,
         r_action = false, r_rays = false, r_command = false;
    bool o_joint = false, o_rays = false, o_action = false, o_command = false;
    uint32_t o_external = 0;  // slots of the external observation buffers the plan reads
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
        rc = check_hip(hipMalloc(&p->d_ids, d.num_ids * sizeof(int32_t)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_ids, d.ids, d.num_ids * sizeof(int32_t), hipMemcpyHostToDevice),
                           "ilx_mdp_plan_create: hipMemcpy");
    }
    if (rc == ILX_OK && d.num_observations > 0) {
        ILX_REQUIRE(d.obs_dim <= 16384, ILX_ESHAPE, "ilx_mdp_plan_create: %d observation columns (max 16384)", d.obs_dim);
        const int groups = (d.obs_dim + 3) / 4;
        std::vector<uint32_t> cols(static_cast<size_t>(groups) + d.num_observations, 0u);
        int col = 0;
        for (int i = 0; i < d.num_observations; ++i) {
            cols[groups + i] = static_cast<uint32_t>(col);
            const int dim = host_obs_dim(d, d.observations[i]);
            for (int k = 0; k < dim; ++k, ++col) cols[col >> 2] |= static_cast<uint32_t>(i) << (8 * (col & 3));
            p->any_noise |= (d.observations[i].flags & ILX_OBS_FLAG_NOISE) != 0;
        }
        rc = check_hip(hipMalloc(&p->d_cols, cols.size() * sizeof(uint32_t)), "ilx_mdp_plan_create: hipMalloc");
        if (rc == ILX_OK)
            rc = check_hip(hipMemcpy(p->d_cols, cols.data(), cols.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
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
    if (plan->d_cols) (void)hipFree(plan->d_cols);
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
        ILX_REQUIRE(!plan->o_rays || (d.ray_hits_w && d.sensor_pos_w), ILX_EINVAL, "ilx_mdp_step: height_scan needs ray_hits_w and sensor_pos_w");
        for (int slot = 0; slot < ILX_MAX_EXTERNAL_OBS; ++slot)
            ILX_REQUIRE(!(plan->o_external & (1u << slot)) || d.external_obs[slot], ILX_EINVAL,
                        "ilx_mdp_step: external observation buffer %d is NULL", slot);
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
    m.cols = plan->d_cols;
    m.any_noise = plan->any_noise;
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
    int w = 0;
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
    L.fpeak = take((m.need & kNeedForces) ? kE * CB : 0);
    L.bvx = take((m.need & kNeedBodyVel) ? kE * B : 0), L.bvy = take((m.need & kNeedBodyVel) ? kE * B : 0);
    L.bacc = take((m.need & kNeedBodyAcc) ? kE * B : 0);
    L.cmd = take(kE * CD), L.root = take(kE * 13);
    L.eps = take(kE * plan->nr), L.srew = take(kE * plan->nr), L.ptotal = take(kWaves * kE);
    L.derived = take(kE * kDerived), L.bits = take(2 * kE);
    L.logp = take(plan->nr + plan->nt + 3);
    L.colterm = take((plan->obs_dim + 3) / 4 + plan->no);  // one block: tstart follows the packed term indices directly
    L.tstart = L.colterm + (plan->obs_dim + 3) / 4;
    L.total = w;
    const size_t lds_bytes = static_cast<size_t>(w) * 4;
    ILX_REQUIRE(lds_bytes <= kMaxLdsBytes, ILX_ESHAPE,
                "ilx_mdp_step: %zu bytes of per-block state (16 envs x row widths) exceed the %d-byte LDS budget", lds_bytes,
                kMaxLdsBytes);
    m.d.terminations = m.d.rewards = m.d.observations = nullptr;
    m.d.ids = nullptr;
    const unsigned blocks = static_cast<unsigned>((d.num_envs + kE - 1) / kE);
    hipLaunchKernelGGL(mdp_step_kernel, dim3(blocks), dim3(kBlock), lds_bytes, as_stream(stream), m);
    ILX_LAUNCH_CHECK("ilx_mdp_step");
    return ILX_OK;
}
