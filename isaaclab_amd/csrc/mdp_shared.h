// What the two fused MDP kernels share: the term formulas (written once, against an environment accessor type, so the
// plan-interpreting kernel and the kernels compiled for one term list produce the same bits), the command term, the
// plan object and the dealing of reward terms to wavefronts.
//
// An accessor type E gives one environment's view of the step's inputs to the lane that evaluates it:
//   e.desc()                       the ilx_mdp descriptor (uniform scalars, pointers for the rare direct reads)
//   e.pos, quat, lin_w, ang_w, lin_b, ang_b, grav_b, time_left, command_counter     per-env values in registers
//   e.joint(Fld, j), e.limit(j, side), e.sbody(Fld, b), e.abody(Fld, b), e.action(k), e.prev_action(k), e.cmd(k)
//   e.force_raw(b) -> E::ForceRaw, e.force_peak_of(raw), e.body_row(b), e.ray_hit_z(k), E::kReadsMemory, E::kBatch
//   e.id(t, k), e.id2(t, k), e.jcount(t), e.bcount_sensor(t), e.bcount_asset(t), e.num_actions(), e.num_rays()
// The plan-interpreting kernel reads them from its LDS image (mdp.hip), the compiled kernels from memory with
// compile-time shapes (mdp_fast.hip).
#pragma once

#include "common.h"
#include "math_dev.h"
#include "philox.h"
#include "uniform_access.h"
#include "scene_update_dev.h"

#include "../../include/isaaclab_amd_mdp.h"

namespace ilx {

constexpr int kMaxTerms = 32;  // per manager
constexpr int kWaves = 4;      // wavefronts of a block; the reward terms are dealt to them

// The per-env arrays a term can name.
enum class Fld : int { q, qd, qdd, tau_a, tau_c, q0, qd0, vlim, air, last_air, contact_t, bvx, bvy, bacc };

// Sums and any-tests over a term's joints.  Four elements' reads in flight together; the additions keep their order (the
// sums are compared bit for bit).
template <class E, typename F>
ILX_DEV float joint_sum(const E& e, const ilx_term& t, F f) {
    const int n = e.jcount(t);
    float acc = 0.0f;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        const float f0 = f(e.id(t, k)), f1 = f(e.id(t, k + 1)), f2 = f(e.id(t, k + 2)), f3 = f(e.id(t, k + 3));
        acc += f0;
        acc += f1;
        acc += f2;
        acc += f3;
    }
    for (; k < n; ++k) acc += f(e.id(t, k));
    return acc;
}
template <class E, typename F>
ILX_DEV bool joint_any(const E& e, const ilx_term& t, F f) {
    const int n = e.jcount(t);
    bool hit = false;
    for (int k = 0; k < n; ++k) hit = hit | f(e.id(t, k));  // no short circuit: the reads do not wait for each other
    return hit;
}

// Per-body values of a term, requested E::kBatch bodies at a time and folded in order.  An accessor that reads memory
// (E::kReadsMemory) gets all of a batch's loads out before the first value is used: left alone, the compiler reuses one
// set of registers and waits for every load before it issues the next -- a trip to memory per value.
template <class E, class G, class F>
ILX_DEV void gather_fold(const E&, int n, G get, F fold) {
    using V = decltype(get(0));
    constexpr int B = E::kBatch;
    for (int k0 = 0; k0 < n; k0 += B) {
        V v[B];
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (k0 + i < n) v[i] = get(k0 + i);
        if constexpr (E::kReadsMemory) __builtin_amdgcn_sched_barrier(0x3cf);   // only the loads are held in place
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (k0 + i < n) fold(k0 + i, v[i]);
    }
}

// --------------------------------------------------------------------------------------------------
template <class E>
ILX_DEV bool eval_termination(const E& e, const ilx_term& t, int64_t ep_len) {
    const ilx_mdp& d = e.desc();
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
            return joint_any(e, t, [&](int j) { const float q = e.joint(Fld::q, j); return (q > t.p[1]) || (q < t.p[0]); });
        case ILX_TERM_JOINT_VEL_OUT_OF_LIMIT:  // :108-114
            return joint_any(e, t, [&](int j) { return fabsf(e.joint(Fld::qd, j)) > e.joint(Fld::vlim, j); });
        case ILX_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT:  // :117-124
            return joint_any(e, t, [&](int j) { return fabsf(e.joint(Fld::qd, j)) > t.p[0]; });
        case ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT:  // :127-142, torch.isclose(rtol=1e-5, atol=1e-8)
            return joint_any(e, t, [&](int j) {
                const float a = e.joint(Fld::tau_c, j), b = e.joint(Fld::tau_a, j);
                return (a == b) || (fabsf(a - b) <= (1.0e-8f + 1.0e-5f * fabsf(b)));
            });
        case ILX_TERM_ILLEGAL_CONTACT: {  // :150-158
            bool hit = false;
            gather_fold(e, e.bcount_sensor(t), [&](int k) { return e.force_raw(e.id(t, k)); },
                        [&](int, const typename E::ForceRaw& r) { hit = hit | (e.force_peak_of(r) > t.p[0]); });
            return hit;
        }
    }
    return false;
}

template <class E>
ILX_DEV float eval_reward(const E& e, const ilx_term& t, bool terminated, bool time_out, uint32_t term_bits) {
    const int A = e.num_actions(), R = e.num_rays();
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
                for (int k = 0; k < R; ++k) acc += e.ray_hit_z(k);
                target = t.p[0] + div_(acc, static_cast<float>(R));
            }
            const float dz = e.pos.z - target;
            return dz * dz;
        }
        case ILX_REW_BODY_LIN_ACC_L2: {  // :125-128
            const int n = e.bcount_asset(t);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc += e.abody(Fld::bacc, e.id(t, k));
            return acc;
        }
        case ILX_REW_JOINT_TORQUES_L2:  // :136-143
            return joint_sum(e, t, [&](int j) { const float v = e.joint(Fld::tau_a, j); return v * v; });
        case ILX_REW_JOINT_VEL_L1:  // :146-150
            return joint_sum(e, t, [&](int j) { return fabsf(e.joint(Fld::qd, j)); });
        case ILX_REW_JOINT_VEL_L2:  // :153-160
            return joint_sum(e, t, [&](int j) { const float v = e.joint(Fld::qd, j); return v * v; });
        case ILX_REW_JOINT_ACC_L2:  // :163-170
            return joint_sum(e, t, [&](int j) { const float v = e.joint(Fld::qdd, j); return v * v; });
        case ILX_REW_JOINT_DEVIATION_L1:  // :173-179
            return joint_sum(e, t, [&](int j) { return fabsf(e.joint(Fld::q, j) - e.joint(Fld::q0, j)); });
        case ILX_REW_JOINT_POS_LIMITS:  // :182-196
            return joint_sum(e, t, [&](int j) {
                const float q = e.joint(Fld::q, j);
                const float lo = e.limit(j, 0), hi = e.limit(j, 1);
                return -fminf(q - lo, 0.0f) + fmaxf(q - hi, 0.0f);
            });
        case ILX_REW_JOINT_VEL_LIMITS:  // :199-218
            return joint_sum(e, t, [&](int j) {
                return fminf(fmaxf(fabsf(e.joint(Fld::qd, j)) - e.joint(Fld::vlim, j) * t.p[0], 0.0f), 1.0f);
            });
        case ILX_REW_APPLIED_TORQUE_LIMITS:  // :226-242
            return joint_sum(e, t, [&](int j) { return fabsf(e.joint(Fld::tau_a, j) - e.joint(Fld::tau_c, j)); });
        case ILX_REW_ACTION_RATE_L2: {  // :245-247
            float acc = 0.0f;
            for (int k = 0; k < A; ++k) {
                const float v = e.action(k) - e.prev_action(k);
                acc += v * v;
            }
            return acc;
        }
        case ILX_REW_ACTION_L2: {  // :250-252
            float acc = 0.0f;
            for (int k = 0; k < A; ++k) {
                const float v = e.action(k);
                acc += v * v;
            }
            return acc;
        }
        case ILX_REW_UNDESIRED_CONTACTS: {  // :260-268 (count of bodies above the threshold)
            float acc = 0.0f;
            gather_fold(e, e.bcount_sensor(t), [&](int k) { return e.force_raw(e.id(t, k)); },
                        [&](int, const typename E::ForceRaw& r) { acc += (e.force_peak_of(r) > t.p[0]) ? 1.0f : 0.0f; });
            return acc;
        }
        case ILX_REW_CONTACT_FORCES: {  // :271-279
            float acc = 0.0f;
            gather_fold(e, e.bcount_sensor(t), [&](int k) { return e.force_raw(e.id(t, k)); },
                        [&](int, const typename E::ForceRaw& r) { acc += fmaxf(e.force_peak_of(r) - t.p[0], 0.0f); });
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
            struct Foot {
                float contact_t, last_air;
            };
            float acc = 0.0f;
            gather_fold(e, e.bcount_sensor(t),
                        [&](int k) {
                            const int b = e.id(t, k);
                            return Foot{e.sbody(Fld::contact_t, b), e.sbody(Fld::last_air, b)};
                        },
                        [&](int, const Foot& f) {
                            const bool first = (f.contact_t > 0.0f) && (f.contact_t < t.p[1]);  // contact_sensor.py:200-203
                            acc += (f.last_air - t.p[0]) * (first ? 1.0f : 0.0f);
                        });
            return acc * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED: {  // velocity/mdp/rewards.py:47-66
            struct Foot {
                float contact_t, air;
            };
            const int n = e.bcount_sensor(t);
            const auto foot = [&](int k) {
                const int b = e.id(t, k);
                return Foot{e.sbody(Fld::contact_t, b), e.sbody(Fld::air, b)};
            };
            float cnt = 0.0f;
            gather_fold(e, n, foot, [&](int, const Foot& f) { cnt += (f.contact_t > 0.0f) ? 1.0f : 0.0f; });
            const bool single = cnt == 1.0f;
            float mn = __builtin_inff();
            gather_fold(e, n, foot, [&](int, const Foot& f) {
                const float mode = (f.contact_t > 0.0f) ? f.contact_t : f.air;
                mn = fminf(mn, single ? mode : 0.0f);
            });
            return fminf(mn, t.p[0]) * ((norm2(e.cmd(0), e.cmd(1)) > 0.1f) ? 1.0f : 0.0f);
        }
        case ILX_REW_FEET_SLIDE: {  // velocity/mdp/rewards.py:69-83
            struct Foot {
                typename E::ForceRaw force;
                float vx, vy;
            };
            float acc = 0.0f;
            gather_fold(e, e.bcount_sensor(t),
                        [&](int k) {
                            const int ab = e.id2(t, k);
                            return Foot{e.force_raw(e.id(t, k)), e.abody(Fld::bvx, ab), e.abody(Fld::bvy, ab)};
                        },
                        [&](int, const Foot& f) { acc += norm2(f.vx, f.vy) * ((e.force_peak_of(f.force) > 1.0f) ? 1.0f : 0.0f); });
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
            const auto* b = e.body_row(e.id(t, 0));
            const float dist = norm3(b[0] - des.x, b[1] - des.y, b[2] - des.z);
            return t.kind == ILX_REW_POSITION_COMMAND_ERROR ? dist : (1.0f - tanhf(div_(dist, t.p[0])));
        }
        case ILX_REW_ORIENTATION_COMMAND_ERROR: {  // reach/mdp/rewards.py:55-69
            const f4 des = quat_mul(e.quat, f4{e.cmd(3), e.cmd(4), e.cmd(5), e.cmd(6)});
            const auto* b = e.body_row(e.id(t, 0)) + 3;
            return quat_error_magnitude(f4{b[0], b[1], b[2], b[3]}, des);
        }
        case ILX_REW_JOINT_POS_TARGET_L2:  // cartpole/mdp/rewards.py:19-26
            return joint_sum(e, t, [&](int j) { const float v = wrap_to_pi(e.joint(Fld::q, j)) - t.p[0]; return v * v; });
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

// The seven samples a resampling env consumes: injected values, or two generator calls mapped to the term's ranges
// (the two coin flips of the velocity term stay in [0, 1)).
// The two generator calls behind a resampling env's seven samples (stream `salt`: 0 in compute, 16 in the reset).
struct CommandBits {
    u4 a, b;
};
ILX_DEV CommandBits command_bits(uint64_t seed, uint64_t step, int64_t env, uint32_t salt) {
    return {philox4(seed, step, static_cast<uint64_t>(env), 0x40000000u + salt), philox4(seed, step, static_cast<uint64_t>(env), 0x40000001u + salt)};
}
// `drawn` (or null): the generator's output for this env and stream, made ahead of time.
template <typename P>
ILX_DEV void command_draws(const ilx_mdp& d, uint64_t seed, uint64_t step, P injected, int64_t env, uint32_t salt, float (&smp)[7],
                           const CommandBits* drawn = nullptr) {
    if (injected) {
#pragma unroll
        for (int k = 0; k < 7; ++k) smp[k] = injected[env * 7 + k];
        return;
    }
    const CommandBits bits = drawn ? *drawn : command_bits(seed, step, env, salt);
    const u4 a = bits.a, b = bits.b;
    const float u[7] = {u01(a.x), u01(a.y), u01(a.z), u01(a.w), u01(b.x), u01(b.y), u01(b.z)};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const bool coin = d.command_kind == ILX_CMD_VELOCITY && k >= 5;
        smp[k] = coin ? u[k] : u[k] * (d.cmd_range[k][1] - d.cmd_range[k][0]) + d.cmd_range[k][0];
    }
}

// CommandTerm._resample for one env: command_manager.py:172-187 + the term's _resample_command
template <typename P>
ILX_DEV void command_resample(const ilx_mdp& d, uint64_t seed, uint64_t step, int64_t env, P injected, uint32_t salt, CmdState& s,
                              const CommandBits* drawn = nullptr) {
    float v[7];
    command_draws(d, seed, step, injected, env, salt, v, drawn);
    s.time_left = v[0];
    s.counter += 1;
    if (d.command_kind == ILX_CMD_VELOCITY) {  // velocity_command.py:123-138
        s.c[0] = v[1], s.c[1] = v[2], s.c[2] = v[3];
        if (d.heading_command) {
            s.heading_target = v[4];
            s.is_heading = v[5] <= d.rel_heading_envs;
        }
        s.is_standing = v[6] <= d.rel_standing_envs;
    } else if (d.command_kind == ILX_CMD_POSE) {  // pose_command.py:107-120
        s.c[0] = v[1], s.c[1] = v[2], s.c[2] = v[3];
        f4 q = quat_from_euler_xyz(v[4], v[5], v[6]);
        if (d.make_quat_unique) q = quat_unique(q);
        s.c[3] = q.w, s.c[4] = q.x, s.c[5] = q.y, s.c[6] = q.z;
    }
}

// CommandTerm.compute for one env (command_manager.py:151-166) in its three parts: metrics, time left + resampling, the
// term's _update_command.  `velocity` / `heading_command` and the scalars are the descriptor's, handed in as uniform
// values; `body` is the tracked body's 13-wide state row (pose command only).
template <class E, typename BP>
ILX_DEV void command_metrics(const E& en, bool velocity, float max_command_step, BP body, CmdState& cs) {
    if (velocity) {  // velocity_command.py:111-121
        cs.metric0 = cs.metric0 + div_(norm2(cs.c[0] - en.lin_b.x, cs.c[1] - en.lin_b.y), max_command_step);
        cs.metric1 = cs.metric1 + div_(fabsf(cs.c[2] - en.ang_b.z), max_command_step);
    } else {  // pose_command.py:92-108 (metrics are overwritten each step)
        f3 pw;
        f4 qw;
        combine_frame_transforms(en.pos, en.quat, f3{cs.c[0], cs.c[1], cs.c[2]}, f4{cs.c[3], cs.c[4], cs.c[5], cs.c[6]}, pw, qw);
        const f3 aa = axis_angle_from_quat(quat_error(qw, f4{body[3], body[4], body[5], body[6]}));
        cs.metric0 = norm3(body[0] - pw.x, body[1] - pw.y, body[2] - pw.z);
        cs.metric1 = norm3(aa);
    }
}
template <class E>
ILX_DEV void command_update(const E& en, bool velocity, bool heading_command, float stiffness, float yaw_lo, float yaw_hi, CmdState& cs) {
    if (velocity) {  // velocity_command.py:140-160
        if (heading_command && cs.is_heading) {
            const f3 fwd = quat_apply(en.quat, f3{1.0f, 0.0f, 0.0f});  // articulation_data.py:416-425
            const float err = wrap_to_pi(cs.heading_target - atan2f(fwd.y, fwd.x));
            cs.c[2] = fminf(fmaxf(stiffness * err, yaw_lo), yaw_hi);
        }
        if (cs.is_standing) cs.c[0] = cs.c[1] = cs.c[2] = 0.0f;
    }
}
template <class E, typename P, typename BP>
ILX_DEV void command_compute(const ilx_mdp& d, const E& en, bool velocity, bool heading_command, float dt, float max_command_step,
                             float stiffness, float yaw_lo, float yaw_hi, uint64_t seed, uint64_t step, int64_t env, P in_samples,
                             BP body, CmdState& cs) {
    command_metrics(en, velocity, max_command_step, body, cs);
    cs.time_left = cs.time_left - dt;
    if (cs.time_left <= 0.0f) command_resample(d, seed, step, env, in_samples, 0u, cs);
    command_update(en, velocity, heading_command, stiffness, yaw_lo, yaw_hi, cs);
}

// What a term list looks like to the code that is compiled for it: everything about a term except its parameters.
struct TermSig {
    int kind, ids_cnt, ids2_cnt, flags;
    bool active;  // rewards: weight != 0 (reward_manager.py:145-146 skips the others)
};
struct MdpDims {
    int num_joints, num_bodies, num_sensor_bodies, history_len, num_rays, action_dim, command_dim;
};

// The block's four wavefronts split the reward terms; the stage lasts as long as its most loaded wavefront, so the terms
// are dealt by a rough count of the reads each makes, heaviest first (ties: lower index first), to the least loaded
// wavefront (ties: the lower one).  Two bits per term.  constexpr: the compiled kernels deal their term list at compile
// time and must agree with the plan (the per-wavefront partial sums fix the reward's rounding).
constexpr int reward_cost(const TermSig& t, const MdpDims& d) {
    if (!t.active) return 0;
    const int k = t.kind;
    const int nj = t.ids_cnt < 0 ? d.num_joints : t.ids_cnt, nb = t.ids_cnt < 0 ? d.num_sensor_bodies : t.ids_cnt;
    if ((k >= ILX_REW_JOINT_TORQUES_L2 && k <= ILX_REW_APPLIED_TORQUE_LIMITS) || k == ILX_REW_JOINT_POS_TARGET_L2) return 10 + 3 * nj;
    if (k == ILX_REW_ACTION_RATE_L2 || k == ILX_REW_ACTION_L2) return 10 + 3 * d.action_dim;
    if (k == ILX_REW_UNDESIRED_CONTACTS || k == ILX_REW_CONTACT_FORCES || k == ILX_REW_FEET_SLIDE)
        return 10 + (4 + 4 * d.history_len) * nb;
    if (k == ILX_REW_FEET_AIR_TIME || k == ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED) return 15 + 5 * nb;
    if (k == ILX_REW_BODY_LIN_ACC_L2) return 10 + 2 * (t.ids_cnt < 0 ? d.num_bodies : t.ids_cnt);
    if (k == ILX_REW_BASE_HEIGHT_L2 && (t.flags & 1)) return 10 + 4 * d.num_rays;
    if (k >= ILX_REW_TRACK_LIN_VEL_XY_EXP && k <= ILX_REW_TRACK_ANG_VEL_Z_EXP) return 25;
    if (k >= ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP) return 40;
    return 8;
}
constexpr uint64_t deal_rewards(const TermSig* sig, int count, const MdpDims& d) {
    int order[kMaxTerms] = {}, cost[kMaxTerms] = {};
    for (int i = 0; i < count; ++i) order[i] = i, cost[i] = reward_cost(sig[i], d);
    for (int i = 1; i < count; ++i) {  // stable insertion sort, heaviest first
        const int x = order[i];
        int j = i;
        for (; j > 0 && cost[order[j - 1]] < cost[x]; --j) order[j] = order[j - 1];
        order[j] = x;
    }
    int load[kWaves] = {};
    uint64_t owner = 0;
    for (int n = 0; n < count; ++n) {
        const int i = order[n];
        int w = 0;
        for (int k = 1; k < kWaves; ++k)
            if (load[k] < load[w]) w = k;
        load[w] += cost[i] + 12;  // + the fixed part of a term: its record, the dispatch, the sums it updates
        owner |= static_cast<uint64_t>(w) << (2 * i);
    }
    return owner;
}

// The neutral form of an observation column's post-processing (observation_manager.py:313-318), so that the column
// loops run without branches: x + u * n_rng + n_off, clip to [c_lo, c_hi] as one median-of-three, x * scale.
struct alignas(16) ColParam {
    float n_off, n_rng;  // 0, 0: no noise
    float c_lo, c_hi;    // -inf, +inf: no clip; bounds given the wrong way round clip to the upper one, as torch.clip does
    float scale;         // 1: none
    float pad[3];
};
inline ColParam col_param_of(const ilx_term& t) {
    const bool noisy = t.flags & ILX_OBS_FLAG_NOISE, clip = t.flags & ILX_OBS_FLAG_CLIP;
    ColParam c{};
    c.n_off = noisy ? t.p[0] : 0.0f, c.n_rng = noisy ? t.p[1] : 0.0f;
    c.c_hi = clip ? t.p[3] : __builtin_inff();
    c.c_lo = clip ? (t.p[2] < t.p[3] ? t.p[2] : t.p[3]) : -__builtin_inff();
    c.scale = (t.flags & ILX_OBS_FLAG_SCALE) ? t.weight : 1.0f;
    return c;
}

}  // namespace ilx

// The plan: term tables and id lists on the device and on the host, and what was learnt about the term list at creation.
struct ilx_mdp_plan_s {
    ilx_term* d_terms = nullptr;  // [terminations | rewards | observations]
    int32_t* d_ids = nullptr;
    uint32_t* d_cols = nullptr;  // one ColDesc per observation column (plan-interpreting kernel)
    int dims[3] = {0, 0, 0};     // num_joints, action_dim, command_dim the column descriptors were made for
    bool any_noise = false;
    uint32_t noisy_terms = 0;
    uint64_t reward_owner = 0;   // two bits per reward term: the wavefront that evaluates it
    int nt = 0, nr = 0, no = 0, nids = 0;
    int obs_dim = 0;
    // data the term tables read, per phase group (checked against NULL at every step)
    bool t_contact = false, t_joint = false, t_cmdstate = false;
    bool r_contact = false, r_timers = false, r_body_state = false, r_body_vel = false, r_body_acc = false, r_joint = false,
         r_action = false, r_rays = false, r_command = false;
    bool o_joint = false, o_rays = false, o_action = false, o_command = false;
    uint32_t o_external = 0;  // slots of the external observation buffers the plan reads
    // host copies of the tables, and the kernel compiled for this term list (mdp_fast.hip), if there is one
    ilx_term h_terms[3 * ilx::kMaxTerms];
    int32_t h_ids[256];   // the first ids (compiled kernels take theirs as kernel arguments)
    ilx::ColParam* d_colparam = nullptr;   // one per observation column
    int fast_kernel = -1;         // index into the table of compiled kernels, -1: none matches
    bool fast_enabled = true;     // ilx_mdp_plan_select_kernel
};

namespace ilx {
// mdp_fast.hip: the compiled kernel that matches the plan's term list and the descriptor's sizes (-1: none), its name,
// and its launch -- ILX_OK when launched, 1 when this step is not for it (other phases, missing optional buffers), <0 on error.
int mdp_fast_match(const ilx_mdp_plan_s& plan, const ilx_mdp& d);
const char* mdp_fast_name(int index);
// `scene` (or null): a scene update that is to precede the step (ilx_mdp_step_scene_update).  A kernel that launches takes
// care of it -- as workgroups of its own grid where the term list and the scene allow, else as a launch ahead of its own --;
// a kernel that declines (1) has launched nothing.
int mdp_fast_step(const ilx_mdp_plan_s& plan, const ilx_mdp& d, void* stream, const ilx_scene* scene = nullptr);
}  // namespace ilx
