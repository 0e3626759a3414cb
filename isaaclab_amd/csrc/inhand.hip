// In-hand manipulation direct env (reference: isaaclab_tasks/direct/inhand_manipulation/inhand_manipulation_env.py).
//
// ilx_inhand_step: a 256-thread block owns 16 environments, 16 threads each.  The first thread of an env takes the
// decisions (fall / time-out, reward, goal reached, new goal orientation) and leaves what the observation needs in
// LDS; then the 16 threads write the env's observation row (157 columns for the Shadow hand) with coalesced stores.
// The two global sums of compute_rewards are reduced per block in LDS and leave as one hardware float add each.
#include "common.h"
#include "math_dev.h"
#include "uniform_access.h"

#include "../../include/isaaclab_amd_inhand.h"

namespace ilx {

constexpr int kEnvsPerBlock = 16;

// Philox4x32-10, as in the fused MDP kernel: (seed, step, env, element) -> uniform [0, 1)
ILX_DEV float inhand_uniform(uint64_t seed, uint64_t step, uint64_t env, uint32_t elem) {
    uint32_t c0 = elem, c1 = static_cast<uint32_t>(env), c2 = static_cast<uint32_t>(env >> 32) ^ static_cast<uint32_t>(step),
             c3 = static_cast<uint32_t>(step >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0, c1 = lo1, c2 = hi0 ^ c3 ^ k1, c3 = lo0;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return static_cast<float>(c0 >> 8) * (1.0f / 16777216.0f);
}

// :342-346
ILX_DEV float rotation_distance(f4 object_rot, f4 target_rot) {
    const f4 d = quat_mul(object_rot, quat_conjugate(target_rot));
    return 2.0f * asinf(fminf(norm3(d.x, d.y, d.z), 1.0f));
}

// :335-339 with the x / y unit axes
ILX_DEV f4 randomize_rotation(float rand0, float rand1) {
    return quat_mul(quat_from_angle_axis(rand0 * kPi, f3{1.0f, 0.0f, 0.0f}), quat_from_angle_axis(rand1 * kPi, f3{0.0f, 1.0f, 0.0f}));
}

// _apply_action (:97-117): one thread per (env, actuated joint)
__global__ void __launch_bounds__(256) inhand_apply_action_kernel(ilx_inhand d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= d.num_envs * d.num_actuated) return;
    const int64_t env = i / d.num_actuated;
    const int64_t j = env * d.num_joints + d.actuated_ids[i - env * d.num_actuated];
    const float lo = d.lower_limits[j], hi = d.upper_limits[j];
    float t = 0.5f * (d.actions[i] + 1.0f) * (hi - lo) + lo;                                 // scale :325-327
    t = d.act_moving_average * t + d.one_minus_act_moving_average * d.prev_targets[j];
    t = fminf(fmaxf(t, lo), hi);                                                             // saturate
    d.cur_targets[j] = t;
    d.prev_targets[j] = t;
}

struct EnvShared {
    float goal[4];   // goal orientation after the resets
    float diff[4];   // object_rot * conj(goal)
    float obj_pos[3];
};

__global__ void __launch_bounds__(256) inhand_step_kernel(ilx_inhand d) {
    request_kernel_arguments<kernel_argument_bytes<ilx_inhand>()>();
    __shared__ EnvShared sh[kEnvsPerBlock];
    __shared__ float block_stats[2];
    const int tid = threadIdx.x;
    const int e = tid >> 4, g = tid & 15;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * kEnvsPerBlock + e;
    const bool live = env < d.num_envs;
    if (tid < 2) block_stats[tid] = 0.0f;
    __syncthreads();
    const int J = d.num_joints, A = d.num_actuated, F = d.num_fingertips;
    if (live && g == 0) {
        const float* os = d.object_state_w + 13 * env;
        const f3 origin = load3(d.env_origins + 3 * env), hand = load3(d.in_hand_pos + 3 * env);
        const f3 obj = {os[0] - origin.x, os[1] - origin.y, os[2] - origin.z};   // :276
        const f4 obj_rot = load4u(os + 3);
        f4 goal = load4u(d.goal_rot + 4 * env);
        float successes = d.successes[env];
        int64_t ep = d.episode_length_buf[env] + 1;  // direct_rl_env.py:352
        // ---- _get_dones :179-199
        const float goal_dist = norm3(obj.x - hand.x, obj.y - hand.y, obj.z - hand.z);
        const bool out_of_reach = goal_dist >= d.fall_dist;
        const float rot_dist = rotation_distance(obj_rot, goal);
        const bool at_goal = fabsf(rot_dist) <= d.success_tolerance;
        bool time_out;
        if (d.max_consecutive_success > 0) {
            if (at_goal) ep = 0;
            time_out = (ep >= d.max_episode_length - 1) || (successes >= static_cast<float>(d.max_consecutive_success));
        } else {
            time_out = ep >= d.max_episode_length - 1;
        }
        const bool reset = out_of_reach || time_out;
        d.terminated[env] = out_of_reach;
        d.time_outs[env] = time_out;
        if (d.reset_buf) d.reset_buf[env] = reset;
        // ---- compute_rewards :349-420
        float act2 = 0.0f;
        for (int k = 0; k < A; ++k) {
            const float a = d.actions[env * A + k];
            act2 += a * a;
        }
        float reward = goal_dist * d.dist_reward_scale + div_(1.0f, fabsf(rot_dist) + d.rot_eps) * d.rot_reward_scale +
                       act2 * d.action_penalty_scale;
        const bool goal_reset = at_goal || d.reset_goal_buf[env] != 0;
        successes = successes + (goal_reset ? 1.0f : 0.0f);
        if (goal_reset) reward = reward + d.reach_goal_bonus;
        if (out_of_reach) reward = reward + d.fall_penalty;
        d.reward[env] = reward;
        if (reset) {  // resets = where(goal_dist >= fall_dist, 1, reset_buf): reset_buf already contains the fallen envs
            atomicAdd(&block_stats[0], 1.0f);
            atomicAdd(&block_stats[1], successes);
        }
        // ---- goal resets (:168-171) then env resets (direct_rl_env.py:369-371, :200-247), simulator writes left out
        const uint64_t gen_step = d.step_index + (d.step_index_device ? *d.step_index_device : 0ull);
        if (goal_reset) {
            const float r0 = d.goal_samples ? d.goal_samples[2 * env] : 2.0f * inhand_uniform(d.seed, gen_step, env, 0) - 1.0f;
            const float r1 = d.goal_samples ? d.goal_samples[2 * env + 1] : 2.0f * inhand_uniform(d.seed, gen_step, env, 1) - 1.0f;
            goal = randomize_rotation(r0, r1);
        }
        if (reset) {
            const float r0 = d.reset_goal_samples ? d.reset_goal_samples[2 * env] : 2.0f * inhand_uniform(d.seed, gen_step, env, 2) - 1.0f;
            const float r1 = d.reset_goal_samples ? d.reset_goal_samples[2 * env + 1] : 2.0f * inhand_uniform(d.seed, gen_step, env, 3) - 1.0f;
            goal = randomize_rotation(r0, r1);
            successes = 0.0f;
            ep = 0;
        }
        if (goal_reset || reset) store4u(d.goal_rot + 4 * env, goal);
        d.reset_goal_buf[env] = 0;  // set by compute_rewards, cleared again by _reset_target_pose for exactly those envs
        d.successes[env] = successes;
        d.episode_length_buf[env] = ep;
        const f4 diff = quat_mul(obj_rot, quat_conjugate(goal));
        EnvShared& s = sh[e];
        s.goal[0] = goal.w, s.goal[1] = goal.x, s.goal[2] = goal.y, s.goal[3] = goal.z;
        s.diff[0] = diff.w, s.diff[1] = diff.x, s.diff[2] = diff.y, s.diff[3] = diff.z;
        s.obj_pos[0] = obj.x, s.obj_pos[1] = obj.y, s.obj_pos[2] = obj.z;
    }
    __syncthreads();
    if (tid < 2 && block_stats[tid] != 0.0f) unsafeAtomicAdd(d.stats + tid, block_stats[tid]);
    if (!live || !d.obs) return;
    // ---- compute_full_observations :298-322
    const int O = 2 * J + 24 + 13 * F + A;
    const EnvShared& s = sh[e];
    const float* os = d.object_state_w + 13 * env;
    float* row = d.obs + env * O;
    for (int c = g; c < O; c += 16) {
        int k = c;
        float v;
        if (k < J) {  // unscale :330-332
            const float lo = d.lower_limits[env * J + k], hi = d.upper_limits[env * J + k];
            v = div_(2.0f * d.joint_pos[env * J + k] - hi - lo, hi - lo);
        } else if ((k -= J) < J) {
            v = d.vel_obs_scale * d.joint_vel[env * J + k];
        } else if ((k -= J) < 3) {
            v = s.obj_pos[k];
        } else if ((k -= 3) < 4) {
            v = os[3 + k];
        } else if ((k -= 4) < 3) {
            v = os[7 + k];
        } else if ((k -= 3) < 3) {
            v = d.vel_obs_scale * os[10 + k];
        } else if ((k -= 3) < 3) {
            v = d.in_hand_pos[3 * env + k];
        } else if ((k -= 3) < 4) {
            v = s.goal[k];
        } else if ((k -= 4) < 4) {
            v = s.diff[k];
        } else if ((k -= 4) < 3 * F) {  // fingertip positions relative to the env origin (:267-270)
            const int f = k / 3, a = k - 3 * f;
            v = d.body_state_w[13 * (env * d.num_bodies + d.fingertip_ids[f]) + a] - d.env_origins[3 * env + a];
        } else if ((k -= 3 * F) < 4 * F) {
            const int f = k >> 2, a = k & 3;
            v = d.body_state_w[13 * (env * d.num_bodies + d.fingertip_ids[f]) + 3 + a];
        } else if ((k -= 4 * F) < 6 * F) {
            const int f = k / 6, a = k - 6 * f;
            v = d.body_state_w[13 * (env * d.num_bodies + d.fingertip_ids[f]) + 7 + a];
        } else {
            k -= 6 * F;
            v = d.actions[env * A + k];
        }
        row[c] = v;
    }
}

// :422-431 from the (possibly all-reduced) sums; one thread
__global__ void inhand_finish_kernel(ilx_inhand d) {
    const float num_resets = d.stats[0], finished = d.stats[1];
    const float prev = d.consecutive_successes[0];
    if (num_resets > 0.0f) d.consecutive_successes[0] = div_(d.av_factor * finished, num_resets) + d.one_minus_av_factor * prev;
    d.stats[0] = 0.0f;
    d.stats[1] = 0.0f;
    if (d.step_index_device) *d.step_index_device += 1;   // the step's launches are behind this one on the stream
}

static int check_common(const ilx_inhand& d, const char* who) {
    ILX_REQUIRE(d.num_envs >= 0 && d.num_joints > 0 && d.num_actuated > 0 && d.num_actuated <= d.num_joints, ILX_EINVAL,
                "%s: bad shape (n=%lld, joints=%d, actuated=%d)", who, (long long)d.num_envs, d.num_joints, d.num_actuated);
    return ILX_OK;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_inhand_apply_action(const ilx_inhand* env, void* stream) {
    ILX_REQUIRE(env, ILX_EINVAL, "ilx_inhand_apply_action: descriptor is NULL");
    const ilx_inhand& d = *env;
    const int rc = check_common(d, "ilx_inhand_apply_action");
    if (rc != ILX_OK) return rc;
    if (d.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(d.actuated_ids && d.lower_limits && d.upper_limits && d.actions && d.cur_targets && d.prev_targets, ILX_EINVAL,
                "ilx_inhand_apply_action: NULL operand");
    hipLaunchKernelGGL(inhand_apply_action_kernel, dim3(blocks_for(d.num_envs * d.num_actuated)), dim3(256), 0, as_stream(stream), d);
    ILX_LAUNCH_CHECK("ilx_inhand_apply_action");
    return ILX_OK;
}

extern "C" int ilx_inhand_step(const ilx_inhand* env, void* stream) {
    ILX_REQUIRE(env, ILX_EINVAL, "ilx_inhand_step: descriptor is NULL");
    const ilx_inhand& d = *env;
    const int rc = check_common(d, "ilx_inhand_step");
    if (rc != ILX_OK) return rc;
    if (d.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(d.num_bodies > 0 && d.num_fingertips >= 0 && d.num_fingertips <= d.num_bodies, ILX_EINVAL,
                "ilx_inhand_step: bad body counts (%d bodies, %d fingertips)", d.num_bodies, d.num_fingertips);
    ILX_REQUIRE(d.object_state_w && d.env_origins && d.in_hand_pos && d.actions && d.goal_rot && d.successes &&
                    d.reset_goal_buf && d.episode_length_buf && d.stats && d.reward && d.terminated && d.time_outs,
                ILX_EINVAL, "ilx_inhand_step: NULL operand");
    if (d.obs)
        ILX_REQUIRE(d.joint_pos && d.joint_vel && d.lower_limits && d.upper_limits && d.body_state_w &&
                        (d.num_fingertips == 0 || d.fingertip_ids),
                    ILX_EINVAL, "ilx_inhand_step: the observation needs joint state, limits, body_state_w and fingertip_ids");
    const unsigned blocks = static_cast<unsigned>((d.num_envs + kEnvsPerBlock - 1) / kEnvsPerBlock);
    hipLaunchKernelGGL(inhand_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d);
    ILX_LAUNCH_CHECK("ilx_inhand_step");
    return ILX_OK;
}

extern "C" int ilx_inhand_finish(const ilx_inhand* env, void* stream) {
    ILX_REQUIRE(env, ILX_EINVAL, "ilx_inhand_finish: descriptor is NULL");
    ILX_REQUIRE(env->stats && env->consecutive_successes, ILX_EINVAL, "ilx_inhand_finish: NULL operand");
    hipLaunchKernelGGL(inhand_finish_kernel, dim3(1), dim3(1), 0, as_stream(stream), *env);
    ILX_LAUNCH_CHECK("ilx_inhand_finish");
    return ILX_OK;
}
