// Fused MDP step: termination / reward / reset-bookkeeping / command / observation managers in ONE launch.
//
// Mapping: one 64-lane wavefront per environment, four environments per 256-thread block.  Rows of the
// ArticulationData arrays ((N,J), (N,B,3), (N,R,3) ...) are contiguous per env, so lane k reading element k of
// the env's row is a fully coalesced access without any AoS->SoA staging; sums over joints / bodies / rays are
// wave reductions (shuffle butterflies), decisions are ballots.  Every env's state is read once per launch and
// the whole chain termination -> reward -> command -> observation stays in registers.
//
// The term tables (ilx_term arrays) are interpreted: `kind` is uniform across the wave, so the dispatch is
// scalar control flow.  Term formulas cite the reference file:line; op order for decisions follows math_dev.h.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd_mdp.h"

#include <cstring>
#include <vector>

namespace ilx {

constexpr int kMaxTerms = 32;

struct MdpDev {
    ilx_mdp d;               // by-value copy; the HOST pointers inside are replaced below
    const ilx_term* terms_t;  // device
    const ilx_term* terms_r;
    const ilx_term* terms_o;
    const int32_t* ids;
};

ILX_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
ILX_DEV float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
ILX_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }

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

struct Env {
    const MdpDev& m;
    int64_t env;
    int lane;
    f3 pos, lin_w, ang_w, lin_b, ang_b, grav_b;
    f4 quat;
    float heading;

    ILX_DEV int jid(const ilx_term& t, int k) const { return t.ids_cnt < 0 ? k : m.ids[t.ids_off + k]; }
    ILX_DEV int jcount(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_joints : t.ids_cnt; }
    ILX_DEV int bcount_sensor(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_sensor_bodies : t.ids_cnt; }
    ILX_DEV int bcount_asset(const ilx_term& t) const { return t.ids_cnt < 0 ? m.d.num_bodies : t.ids_cnt; }

    // max over the history axis of |F| for sensor body b: rewards.py:266 / terminations.py:157
    ILX_DEV float hist_max_norm(int b) const {
        const int H = m.d.history_len, CB = m.d.num_sensor_bodies;
        float mx = -__builtin_inff();
        for (int h = 0; h < H; ++h) {
            const float* f = m.d.net_forces_w_history + 3 * ((env * H + h) * CB + b);
            mx = fmaxf(mx, norm3(f[0], f[1], f[2]));
        }
        return mx;
    }
    template <typename F>
    ILX_DEV float joint_sum(const ilx_term& t, F f) const {
        const int n = jcount(t);
        float acc = 0.0f;
        for (int k = lane; k < n; k += 64) acc += f(env * m.d.num_joints + jid(t, k));
        return wave_sum(acc);
    }
    ILX_DEV float cmd(int k) const { return m.d.command[env * m.d.command_dim + k]; }
};

// --------------------------------------------------------------------------------------------------
ILX_DEV bool eval_termination(const Env& e, const ilx_term& t, int64_t ep_len) {
    const ilx_mdp& d = e.m.d;
    const int J = d.num_joints;
    switch (t.kind) {
        case ILX_TERM_TIME_OUT:  // terminations.py:30-32
            return ep_len >= d.max_episode_length;
        case ILX_TERM_COMMAND_RESAMPLE:  // :35-42
            return (d.time_left[e.env] <= t.p[0]) && (d.command_counter[e.env] == static_cast<int64_t>(t.p[1]));
        case ILX_TERM_BAD_ORIENTATION:  // :50-59
            return fabsf(acosf(-e.grav_b.z)) > t.p[0];
        case ILX_TERM_ROOT_HEIGHT_BELOW_MINIMUM:  // :62-72
            return e.pos.z < t.p[0];
        case ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT: {  // :90-105
            bool hit = false;
            for (int k = e.lane; k < e.jcount(t); k += 64) {
                const float q = d.joint_pos[e.env * J + e.jid(t, k)];
                hit = hit || (q > t.p[1]) || (q < t.p[0]);
            }
            return wave_any(hit);
        }
        case ILX_TERM_JOINT_VEL_OUT_OF_LIMIT: {  // :108-114
            bool hit = false;
            for (int k = e.lane; k < e.jcount(t); k += 64) {
                const int64_t i = e.env * J + e.jid(t, k);
                hit = hit || (fabsf(d.joint_vel[i]) > d.soft_joint_vel_limits[i]);
            }
            return wave_any(hit);
        }
        case ILX_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT: {  // :117-124
            bool hit = false;
            for (int k = e.lane; k < e.jcount(t); k += 64)
                hit = hit || (fabsf(d.joint_vel[e.env * J + e.jid(t, k)]) > t.p[0]);
            return wave_any(hit);
        }
        case ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT: {  // :127-142, torch.isclose(rtol=1e-5, atol=1e-8)
            bool hit = false;
            for (int k = e.lane; k < e.jcount(t); k += 64) {
                const int64_t i = e.env * J + e.jid(t, k);
                const float a = d.computed_torque[i], b = d.applied_torque[i];
                hit = hit || (a == b) || (fabsf(a - b) <= (1.0e-8f + 1.0e-5f * fabsf(b)));
            }
            return wave_any(hit);
        }
        case ILX_TERM_ILLEGAL_CONTACT: {  // :150-158
            bool hit = false;
            for (int k = e.lane; k < e.bcount_sensor(t); k += 64) hit = hit || (e.hist_max_norm(e.jid(t, k)) > t.p[0]);
            return wave_any(hit);
        }
    }
    return false;
}

ILX_DEV float eval_reward(const Env& e, const ilx_term& t, bool terminated, bool time_out, uint32_t term_bits) {
    const ilx_mdp& d = e.m.d;
    const int J = d.num_joints, B = d.num_bodies, CB = d.num_sensor_bodies, A = d.action_dim;
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
                for (int k = e.lane; k < d.num_rays; k += 64) acc += d.ray_hits_w[3 * (e.env * d.num_rays + k) + 2];
                target = t.p[0] + div_(wave_sum(acc), static_cast<float>(d.num_rays));
            }
            const float dz = e.pos.z - target;
            return dz * dz;
        }
        case ILX_REW_BODY_LIN_ACC_L2: {  // :125-128
            float acc = 0.0f;
            for (int k = e.lane; k < e.bcount_asset(t); k += 64) {
                const float* a = d.body_lin_acc_w + 3 * (e.env * B + e.jid(t, k));
                acc += norm3(a[0], a[1], a[2]);
            }
            return wave_sum(acc);
        }
        case ILX_REW_JOINT_TORQUES_L2:  // :136-143
            return e.joint_sum(t, [&](int64_t i) { const float v = d.applied_torque[i]; return v * v; });
        case ILX_REW_JOINT_VEL_L1:  // :146-150
            return e.joint_sum(t, [&](int64_t i) { return fabsf(d.joint_vel[i]); });
        case ILX_REW_JOINT_VEL_L2:  // :153-160
            return e.joint_sum(t, [&](int64_t i) { const float v = d.joint_vel[i]; return v * v; });
        case ILX_REW_JOINT_ACC_L2:  // :163-170
            return e.joint_sum(t, [&](int64_t i) { const float v = d.joint_acc[i]; return v * v; });
        case ILX_REW_JOINT_DEVIATION_L1:  // :173-179
            return e.joint_sum(t, [&](int64_t i) { return fabsf(d.joint_pos[i] - d.default_joint_pos[i]); });
        case ILX_REW_JOINT_POS_LIMITS:  // :182-196
            return e.joint_sum(t, [&](int64_t i) {
                const float q = d.joint_pos[i];
                return -fminf(q - d.soft_joint_pos_limits[2 * i], 0.0f) + fmaxf(q - d.soft_joint_pos_limits[2 * i + 1], 0.0f);
            });
        case ILX_REW_JOINT_VEL_LIMITS:  // :199-218
            return e.joint_sum(t, [&](int64_t i) {
                return fminf(fmaxf(fabsf(d.joint_vel[i]) - d.soft_joint_vel_limits[i] * t.p[0], 0.0f), 1.0f);
            });
        case ILX_REW_APPLIED_TORQUE_LIMITS:  // :226-242
            return e.joint_sum(t, [&](int64_t i) { return fabsf(d.applied_torque[i] - d.computed_torque[i]); });
        case ILX_REW_ACTION_RATE_L2: {  // :245-247
            float acc = 0.0f;
            for (int k = e.lane; k < A; k += 64) {
                const float v = d.action[e.env * A + k] - d.prev_action[e.env * A + k];
                acc += v * v;
            }
            return wave_sum(acc);
        }
        case ILX_REW_ACTION_L2: {  // :250-252
            float acc = 0.0f;
            for (int k = e.lane; k < A; k += 64) {
                const float v = d.action[e.env * A + k];
                acc += v * v;
            }
            return wave_sum(acc);
        }
        case ILX_REW_UNDESIRED_CONTACTS: {  // :260-268 (count of bodies above the threshold)
            float acc = 0.0f;
            for (int k = e.lane; k < e.bcount_sensor(t); k += 64) acc += (e.hist_max_norm(e.jid(t, k)) > t.p[0]) ? 1.0f : 0.0f;
            return wave_sum(acc);
        }
        case ILX_REW_CONTACT_FORCES: {  // :271-279
            float acc = 0.0f;
            for (int k = e.lane; k < e.bcount_sensor(t); k += 64) acc += fmaxf(e.hist_max_norm(e.jid(t, k)) - t.p[0], 0.0f);
            return wave_sum(acc);
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
            float acc = 0.0f;
            for (int k = e.lane; k < e.bcount_sensor(t); k += 64) {
                const int64_t i = e.env * CB + e.jid(t, k);
                const float cc = d.current_contact_time[i];
                const bool first = (cc > 0.0f) && (cc < t.p[1]);  // contact_sensor.py:200-203
                acc += (d.last_air_time[i] - t.p[0]) * (first ? 1.0f : 0.0f);
            }
            return wave_sum(acc) * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED: {  // velocity/mdp/rewards.py:47-66
            const int n = e.bcount_sensor(t);
            float cnt = 0.0f;
            for (int k = e.lane; k < n; k += 64) cnt += (d.current_contact_time[e.env * CB + e.jid(t, k)] > 0.0f) ? 1.0f : 0.0f;
            const bool single = wave_sum(cnt) == 1.0f;
            float mn = __builtin_inff();
            for (int k = e.lane; k < n; k += 64) {
                const int64_t i = e.env * CB + e.jid(t, k);
                const float cc = d.current_contact_time[i];
                const float mode = (cc > 0.0f) ? cc : d.current_air_time[i];
                mn = fminf(mn, single ? mode : 0.0f);
            }
            return fminf(wave_min(mn), t.p[0]) * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_SLIDE: {  // velocity/mdp/rewards.py:69-83
            float acc = 0.0f;
            for (int k = e.lane; k < e.bcount_sensor(t); k += 64) {
                const bool contact = e.hist_max_norm(e.jid(t, k)) > 1.0f;
                const int ab = t.ids2_cnt < 0 ? k : e.m.ids[t.ids2_off + k];
                const float* v = d.body_state_w + 13 * (e.env * B + ab) + 7;
                acc += norm2(v[0], v[1]) * (contact ? 1.0f : 0.0f);
            }
            return wave_sum(acc);
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
            const f3 r = quat_apply(e.quat, f3{e.cmd(0), e.cmd(1), e.cmd(2)});
            const f3 des = {e.pos.x + r.x, e.pos.y + r.y, e.pos.z + r.z};
            const float* b = d.body_state_w + 13 * (e.env * B + e.m.ids[t.ids_off]);
            const float dist = norm3(b[0] - des.x, b[1] - des.y, b[2] - des.z);
            return t.kind == ILX_REW_POSITION_COMMAND_ERROR ? dist : (1.0f - tanhf(div_(dist, t.p[0])));
        }
        case ILX_REW_ORIENTATION_COMMAND_ERROR: {  // reach/mdp/rewards.py:55-69
            const f4 des = quat_mul(e.quat, f4{e.cmd(3), e.cmd(4), e.cmd(5), e.cmd(6)});
            const float* b = d.body_state_w + 13 * (e.env * B + e.m.ids[t.ids_off]) + 3;
            return quat_error_magnitude(f4{b[0], b[1], b[2], b[3]}, des);
        }
        case ILX_REW_JOINT_POS_TARGET_L2:  // cartpole/mdp/rewards.py:19-26
            return e.joint_sum(t, [&](int64_t i) { const float v = wrap_to_pi(d.joint_pos[i]) - t.p[0]; return v * v; });
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
    }
    return 0;
}

ILX_DEV float eval_obs(const Env& e, const ilx_term& t, int k) {
    const ilx_mdp& d = e.m.d;
    const int J = d.num_joints;
    auto pick3 = [](f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); };
    switch (t.kind) {
        case ILX_OBS_BASE_POS_Z: return e.pos.z;                 // observations.py:33-37
        case ILX_OBS_BASE_LIN_VEL: return pick3(e.lin_b, k);      // :40-44
        case ILX_OBS_BASE_ANG_VEL: return pick3(e.ang_b, k);      // :47-51
        case ILX_OBS_PROJECTED_GRAVITY: return pick3(e.grav_b, k);  // :54-58
        case ILX_OBS_ROOT_POS_W: return pick3(e.pos, k) - (d.env_origins ? d.env_origins[3 * e.env + k] : 0.0f);  // :61-65
        case ILX_OBS_ROOT_QUAT_W: {  // :68-82
            const f4 q = (t.flags & ILX_OBS_FLAG_QUAT_UNIQUE) ? quat_unique(e.quat) : e.quat;
            return k == 0 ? q.w : (k == 1 ? q.x : (k == 2 ? q.y : q.z));
        }
        case ILX_OBS_ROOT_LIN_VEL_W: return pick3(e.lin_w, k);  // :85-89
        case ILX_OBS_ROOT_ANG_VEL_W: return pick3(e.ang_w, k);  // :92-96
        case ILX_OBS_JOINT_POS: return d.joint_pos[e.env * J + e.jid(t, k)];  // :104-111
        case ILX_OBS_JOINT_POS_REL: {                                           // :114-121
            const int64_t i = e.env * J + e.jid(t, k);
            return d.joint_pos[i] - d.default_joint_pos[i];
        }
        case ILX_OBS_JOINT_POS_LIMIT_NORMALIZED: {  // :124-137 (math.py:23-40)
            const int64_t i = e.env * J + e.jid(t, k);
            const float lo = d.soft_joint_pos_limits[2 * i], hi = d.soft_joint_pos_limits[2 * i + 1];
            const float off = (lo + hi) * 0.5f;
            return div_(2.0f * (d.joint_pos[i] - off), hi - lo);
        }
        case ILX_OBS_JOINT_VEL: return d.joint_vel[e.env * J + e.jid(t, k)];  // :140-147
        case ILX_OBS_JOINT_VEL_REL: {                                           // :150-157
            const int64_t i = e.env * J + e.jid(t, k);
            return d.joint_vel[i] - d.default_joint_vel[i];
        }
        case ILX_OBS_HEIGHT_SCAN:  // :165-173
            return d.sensor_pos_w[3 * e.env + 2] - d.ray_hits_w[3 * (e.env * d.num_rays + k) + 2] - t.aux;
        case ILX_OBS_LAST_ACTION: return d.action[e.env * d.action_dim + k];     // :512-521
        case ILX_OBS_GENERATED_COMMANDS: return d.command[e.env * d.command_dim + k];  // :529-531
    }
    return 0.0f;
}

// sample s of the command term: injected value, or generated in the stated range
ILX_DEV float cmd_sample(const ilx_mdp& d, const float* injected, int64_t env, int s, bool unit, uint32_t salt) {
    if (injected) return injected[env * 7 + s];
    const float u = philox_uniform(d.seed, d.step_index, static_cast<uint64_t>(env), 0x40000000u + salt + s);
    if (unit) return u;
    return u * (d.cmd_range[s][1] - d.cmd_range[s][0]) + d.cmd_range[s][0];
}

// CommandTerm._resample for one env (lane 0 only): command_manager.py:172-187 + the term's _resample_command
ILX_DEV void command_resample(const ilx_mdp& d, int64_t env, const float* injected, uint32_t salt) {
    d.time_left[env] = cmd_sample(d, injected, env, 0, false, salt);
    d.command_counter[env] += 1;
    float* c = d.command + env * d.command_dim;
    if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:123-138
        c[0] = cmd_sample(d, injected, env, 1, false, salt);
        c[1] = cmd_sample(d, injected, env, 2, false, salt);
        c[2] = cmd_sample(d, injected, env, 3, false, salt);
        if (d.heading_command) {
            d.heading_target[env] = cmd_sample(d, injected, env, 4, false, salt);
            d.is_heading_env[env] = cmd_sample(d, injected, env, 5, true, salt) <= d.rel_heading_envs;
        }
        d.is_standing_env[env] = cmd_sample(d, injected, env, 6, true, salt) <= d.rel_standing_envs;
    } else if (d.command_kind == ILX_CMD_POSE) {  // pose_command.py:107-120
        c[0] = cmd_sample(d, injected, env, 1, false, salt);
        c[1] = cmd_sample(d, injected, env, 2, false, salt);
        c[2] = cmd_sample(d, injected, env, 3, false, salt);
        f4 q = quat_from_euler_xyz(cmd_sample(d, injected, env, 4, false, salt), cmd_sample(d, injected, env, 5, false, salt),
                                   cmd_sample(d, injected, env, 6, false, salt));
        if (d.make_quat_unique) q = quat_unique(q);
        c[3] = q.w, c[4] = q.x, c[5] = q.y, c[6] = q.z;
    }
}

__global__ void __launch_bounds__(256) mdp_step_kernel(MdpDev m) {
    const ilx_mdp& d = m.d;
    const int lane = threadIdx.x & 63;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (env >= d.num_envs) return;  // wave-uniform

    Env e{m, env, lane};
    {
        const float* s = d.root_state_w + 13 * env;
        e.pos = load3(s);
        e.quat = load4u(s + 3);
        e.lin_w = load3(s + 7);
        e.ang_w = load3(s + 10);
        e.lin_b = quat_rotate_inverse(e.quat, e.lin_w);                                   // articulation_data.py:502-509
        e.ang_b = quat_rotate_inverse(e.quat, e.ang_w);                                   // :511-518
        e.grav_b = quat_rotate_inverse(e.quat, f3{d.gravity_dir[0], d.gravity_dir[1], d.gravity_dir[2]});  // :411-414
        const f3 fwd = quat_apply(e.quat, f3{1.0f, 0.0f, 0.0f});                          // :416-425
        e.heading = atan2f(fwd.y, fwd.x);
    }
    const int NT = d.num_terminations, NR = d.num_rewards;
    const float dt = d.step_dt;

    // ---- terminations (manager_based_rl_env.py:199-202, termination_manager.py:151-174)
    bool terminated = false, time_out = false;
    uint32_t term_bits = 0;
    int64_t ep_len = d.episode_length_buf ? d.episode_length_buf[env] : 0;
    if (d.phases & ILX_PHASE_TERMINATION) {
        ep_len += 1;
        for (int i = 0; i < NT; ++i) {
            const ilx_term& t = m.terms_t[i];
            const bool v = eval_termination(e, t, ep_len);
            term_bits |= v ? (1u << i) : 0u;
            if (t.flags & 1) time_out = time_out || v; else terminated = terminated || v;
            if (lane == 0) d.term_dones[static_cast<int64_t>(i) * d.num_envs + env] = v;
        }
        if (lane == 0) {
            if (d.episode_length_buf) d.episode_length_buf[env] = ep_len;
            d.terminated[env] = terminated;
            d.time_outs[env] = time_out;
            if (d.reset_buf) d.reset_buf[env] = terminated || time_out;
        }
    } else if (d.terminated) {
        terminated = d.terminated[env] != 0;
        time_out = d.time_outs[env] != 0;
        for (int i = 0; i < NT; ++i) term_bits |= d.term_dones[static_cast<int64_t>(i) * d.num_envs + env] ? (1u << i) : 0u;
    }
    const bool done = terminated || time_out;

    // ---- rewards (reward_manager.py:128-157)
    if (d.phases & ILX_PHASE_REWARD) {
        float total = 0.0f;
        for (int i = 0; i < NR; ++i) {
            const ilx_term& t = m.terms_r[i];
            if (t.weight == 0.0f) continue;  // :145-146 (episode sums and step_reward keep their values)
            const float value = eval_reward(e, t, terminated, time_out, term_bits) * t.weight * dt;
            total += value;
            if (lane == 0) {
                d.episode_sums[static_cast<int64_t>(i) * d.num_envs + env] += value;
                d.step_reward[env * NR + i] = div_(value, dt);
            }
        }
        if (lane == 0) d.reward_buf[env] = total;
    }

    // ---- reset bookkeeping of the managers for done envs (manager_based_rl_env.py:364-391)
    if ((d.phases & ILX_PHASE_RESET) && done && lane == 0) {
        for (int i = 0; i < NR; ++i) {  // reward_manager.py:115-121
            float* s = d.episode_sums + static_cast<int64_t>(i) * d.num_envs + env;
            if (d.log_sums) atomicAdd(d.log_sums + i, *s);
            *s = 0.0f;
        }
        if (d.log_sums) {
            for (int i = 0; i < NT; ++i)  // termination_manager.py:142-144
                if (term_bits & (1u << i)) atomicAdd(d.log_sums + NR + i, 1.0f);
            atomicAdd(d.log_sums + NR + NT + 2, 1.0f);
        }
        if (d.command_kind != ILX_CMD_NONE) {  // command_manager.py:120-149
            for (int k = 0; k < 2; ++k) {
                float* mt = d.command_metrics + static_cast<int64_t>(k) * d.num_envs + env;
                if (d.log_sums) atomicAdd(d.log_sums + NR + NT + k, *mt);
                *mt = 0.0f;
            }
            d.command_counter[env] = 0;
            command_resample(d, env, d.reset_command_samples, 16u);
        }
        if (d.episode_length_buf) d.episode_length_buf[env] = 0;  // manager_based_rl_env.py:391
    }

    // ---- command term (command_manager.py:151-166)
    if ((d.phases & ILX_PHASE_COMMAND) && d.command_kind != ILX_CMD_NONE) {
        if (lane == 0) {
            if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:111-121
                const float* c = d.command + env * d.command_dim;
                float* mxy = d.command_metrics + env;
                float* myaw = d.command_metrics + d.num_envs + env;
                *mxy = *mxy + div_(norm2(c[0] - e.lin_b.x, c[1] - e.lin_b.y), d.max_command_step);
                *myaw = *myaw + div_(fabsf(c[2] - e.ang_b.z), d.max_command_step);
            } else {  // pose_command.py:92-108 (metrics are overwritten each step)
                const float* c = d.command + env * d.command_dim;
                f3 pw;
                f4 qw;
                combine_frame_transforms(e.pos, e.quat, f3{c[0], c[1], c[2]}, f4{c[3], c[4], c[5], c[6]}, pw, qw);
                const float* b = d.body_state_w + 13 * (env * d.num_bodies + d.pose_body_index);
                const f3 aa = axis_angle_from_quat(quat_error(qw, f4{b[3], b[4], b[5], b[6]}));
                d.command_metrics[env] = norm3(b[0] - pw.x, b[1] - pw.y, b[2] - pw.z);
                d.command_metrics[d.num_envs + env] = norm3(aa);
            }
            const float left = d.time_left[env] - dt;
            d.time_left[env] = left;
            if (left <= 0.0f) command_resample(d, env, d.command_samples, 0u);
            if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:140-160
                float* c = d.command + env * d.command_dim;
                if (d.heading_command && d.is_heading_env[env]) {
                    const float err = wrap_to_pi(d.heading_target[env] - e.heading);
                    c[2] = fminf(fmaxf(d.heading_control_stiffness * err, d.cmd_range[3][0]), d.cmd_range[3][1]);
                }
                if (d.is_standing_env[env]) c[0] = c[1] = c[2] = 0.0f;
            }
        }
        // the observation phase of this wave reads the command written by lane 0
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- observations (observation_manager.py:260-335)
    if (d.phases & ILX_PHASE_OBSERVATION) {
        int col = 0;
        float* row = d.obs + env * d.obs_dim;
        for (int i = 0; i < d.num_observations; ++i) {
            const ilx_term& t = m.terms_o[i];
            const int dim = obs_dim_of(d, t);
            for (int k = lane; k < dim; k += 64) {
                float v = eval_obs(e, t, k);
                if (t.flags & ILX_OBS_FLAG_NOISE) {  // noise_model.py:63-64: data + rand * (max - min) + min
                    const float u = d.noise_u ? d.noise_u[env * d.obs_dim + col + k]
                                              : philox_uniform(d.seed, d.step_index, static_cast<uint64_t>(env), col + k);
                    v = v + u * t.p[1] + t.p[0];
                }
                if (t.flags & ILX_OBS_FLAG_CLIP) v = fminf(fmaxf(v, t.p[2]), t.p[3]);  // :315-316
                if (t.flags & ILX_OBS_FLAG_SCALE) v = v * t.weight;                    // :317-318
                row[col + k] = v;
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
    bool r_contact = false, r_timers = false, r_body_state = false, r_body_acc = false, r_joint = false, r_action = false,
         r_rays = false, r_command = false;
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
    MdpDev m;
    m.d = d;
    m.terms_t = plan->d_terms;
    m.terms_r = plan->d_terms + plan->nt;
    m.terms_o = plan->d_terms + plan->nt + plan->nr;
    m.ids = plan->d_ids;
    m.d.terminations = m.d.rewards = m.d.observations = nullptr;
    m.d.ids = nullptr;
    const unsigned blocks = static_cast<unsigned>((d.num_envs + 3) / 4);
    hipLaunchKernelGGL(mdp_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), m);
    ILX_LAUNCH_CHECK("ilx_mdp_step");
    return ILX_OK;
}
