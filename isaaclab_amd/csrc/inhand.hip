// In-hand manipulation direct env (reference: isaaclab_tasks/direct/inhand_manipulation/inhand_manipulation_env.py).
//
// ilx_inhand_step: a 256-thread block owns 16 environments, 16 threads each.  For the shapes of the reference's hands
// (inhand_step_kernel) every thread issues all of its loads -- the decisions' inputs, its columns of the observation
// row -- before anything else; the env's 16 threads then each take the env's decisions (fall / time-out, reward, goal
// reached, new goal orientation: the same values in all of them, so nothing is handed over through LDS), the first one
// stores them, and the 16 write the observation row (157 columns for the Shadow hand) with coalesced stores.  Other
// shapes take inhand_step_general_kernel: the first thread of an env decides and leaves what the row needs in LDS.
// The two global sums of compute_rewards are reduced per wavefront / block and leave as one hardware float add each.
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

__global__ void __launch_bounds__(256) inhand_step_general_kernel(ilx_inhand d) {
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
    if (tid < 2) {
        if (d.block_sums) d.block_sums[2 * blockIdx.x + tid] = block_stats[tid];
        else if (block_stats[tid] != 0.0f) unsafeAtomicAdd(d.stats + tid, block_stats[tid]);
    }
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

#ifdef ILX_INHAND_STAMPS
__device__ unsigned long long g_inhand_stamps[8 * 1024];
#define IH_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 1024) g_inhand_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IH_STAMP(k)
#endif

// Shapes of the unrolled kernel: 2 J <= 64, 13 F <= 80, A <= 32 (Shadow hand: J = 24, F = 5, A = 20; Allegro: 16, 4, 16)
constexpr int kJointRounds = 4, kTipRounds = 5, kActionRounds = 2;
static bool inhand_unrolled_shape(int J, int F, int A) { return 2 * J <= 16 * kJointRounds && 13 * F <= 16 * kTipRounds && A <= 16 * kActionRounds; }

// The launch is a chain of trips to memory, not a stream of bytes (one block per CU at 4096 envs), so the kernel is laid
// out around the trips: one for every input (addresses clamped instead of branched around, since a load inside a branch
// is waited for where the branch ends), a second one only for the fingertips' body states behind their body indices,
// and no load after the first store (loads and stores share one in-order counter).
__global__ void __launch_bounds__(256) inhand_step_kernel(ilx_inhand d) {
    request_kernel_arguments<kernel_argument_bytes<ilx_inhand>()>();
    IH_STAMP(0);
    __shared__ float act_s[kEnvsPerBlock][16 * kActionRounds];
    __shared__ float block_stats[2];
    __shared__ uint32_t waves_done;
    typedef __attribute__((address_space(1))) char* Base;
    const uint32_t tid = threadIdx.x, e = tid >> 4, g = tid & 15;
    const int64_t env0 = static_cast<int64_t>(blockIdx.x) * kEnvsPerBlock;
    const int64_t env = env0 + e;
    const bool live = env < d.num_envs;
    const uint32_t ec = live ? e : static_cast<uint32_t>(d.num_envs - 1 - env0);   // the env whose values a thread past the end reads
    const uint32_t J = d.num_joints, A = d.num_actuated, F = d.num_fingertips, B = d.num_bodies, O = 2 * J + 24 + 13 * F + A;
    const Base b_act = block_base(d.actions, env0 * A), b_fid = block_base(d.fingertip_ids, 0);
    const Base b_os = block_base(d.object_state_w, env0 * 13), b_origin = block_base(d.env_origins, env0 * 3);
    const Base b_hand = block_base(d.in_hand_pos, env0 * 3), b_goal = block_base(d.goal_rot, env0 * 4);
    const Base b_succ = block_base(d.successes, env0), b_ep = block_base(d.episode_length_buf, env0);
    const Base b_rgb = block_base(d.reset_goal_buf, env0);
    const Base b_q = block_base(d.joint_pos, env0 * J), b_qd = block_base(d.joint_vel, env0 * J);
    const Base b_lo = block_base(d.lower_limits, env0 * J), b_hi = block_base(d.upper_limits, env0 * J);
    const Base b_body = block_base(d.body_state_w, env0 * B * 13), b_row = block_base(d.obs, env0 * O);

    // ---- first trip: the actions, the fingertips' body indices (both have something waiting for them), then the rest
    float act[kActionRounds];
#pragma unroll
    for (int t = 0; t < kActionRounds; ++t)
        act[t] = ld_b<float>(b_act, 4u * (__umul24(ec, A) + min(g + 16u * t, A - 1)));
    uint32_t tip_a[kTipRounds];
    int32_t tip_body[kTipRounds];
#pragma unroll
    for (int t = 0; t < kTipRounds; ++t) {   // fingertip positions (relative to the env origin, :267-270), then rotations, then velocities
        const uint32_t k = min(g + 16u * t, 13 * F - 1), k1 = k - 3 * F, k2 = k - 7 * F;
        const bool pos = k < 3 * F, rot = k < 7 * F;
        // the three readings of k are all computed and one is selected: left to the compiler they become branches, and a
        // 64-bit multiply-add in one of them takes a register with a load in flight as its unused upper half and waits for it
        uint32_t f_pos = k / 3, f_vel = k2 / 6;
        asm volatile("" : "+v"(f_pos), "+v"(f_vel));
        uint32_t a_pos = k - __umul24(f_pos, 3u), a_vel = 7 + k2 - __umul24(f_vel, 6u);
        asm volatile("" : "+v"(a_pos), "+v"(a_vel));
        const uint32_t f = pos ? f_pos : rot ? k1 >> 2 : f_vel;
        tip_a[t] = pos ? a_pos : rot ? 3 + (k1 & 3) : a_vel;
        tip_body[t] = ld_b<int32_t>(b_fid, 4u * f);
    }
    float os[13];
#pragma unroll
    for (int k = 0; k < 7; ++k) os[k] = ld_b<float>(b_os, 4u * (13u * ec + k));
    float origin[3], hand[3], goal_in[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) origin[k] = ld_b<float>(b_origin, 4u * (3u * ec + k)), hand[k] = ld_b<float>(b_hand, 4u * (3u * ec + k));
#pragma unroll
    for (int k = 0; k < 4; ++k) goal_in[k] = ld_b<float>(b_goal, 4u * (4u * ec + k));
    float successes = ld_b<float>(b_succ, 4u * ec);
    int64_t ep = ld_b<int64_t>(b_ep, 8u * ec) + 1;   // direct_rl_env.py:352
    const bool goal_flag = ld_b<uint8_t>(b_rgb, ec) != 0;
    // optional inputs: read through a pointer that is valid either way (a value loaded inside a branch is waited for where
    // the branch ends, and with it every load before it); what was read in place of an absent one is not used
    const float* gs = d.goal_samples ? d.goal_samples + 2 * (env0 + ec) : d.goal_rot + 4 * (env0 + ec);
    const float* rs = d.reset_goal_samples ? d.reset_goal_samples + 2 * (env0 + ec) : d.goal_rot + 4 * (env0 + ec);
    const float sample[4] = {gs[0], gs[1], rs[0], rs[1]};
    const uint64_t steps_before = *(d.step_index_device ? d.step_index_device : reinterpret_cast<const uint64_t*>(d.episode_length_buf));
    float joint[kJointRounds], lo[kJointRounds], hi[kJointRounds];
#pragma unroll
    for (int t = 0; t < kJointRounds; ++t) {
        const uint32_t c = g + 16u * t;
        const uint32_t at = 4u * (__umul24(ec, J) + min(c < J ? c : c - J, J - 1));
        joint[t] = ld_b<float>(c < J ? b_q : b_qd, at);
        lo[t] = ld_b<float>(b_lo, at), hi[t] = ld_b<float>(b_hi, at);
    }
    // columns 3-12 of the 24 between the joints and the fingertips are the object's rotation and velocities, 13-15 the hand's position
    const float misc_os = ld_b<float>(b_os, 4u * (13u * ec + min(max(g, 3u), 12u)));
    const float misc_hand = ld_b<float>(b_hand, 4u * (3u * ec + min(max(g, 13u), 15u) - 13u));
    // ---- second trip: the fingertips' states
    float tip[kTipRounds];
#pragma unroll
    for (int t = 0; t < kTipRounds; ++t)
        tip[t] = ld_b<float>(b_body, 4u * (__umul24(__umul24(ec, B) + static_cast<uint32_t>(tip_body[t]), 13u) + tip_a[t]));
#pragma unroll
    for (int t = 0; t < kActionRounds; ++t)
        if (g + 16u * t < A) act_s[e][g + 16 * t] = act[t];
    if (tid < 2) block_stats[tid] = 0.0f;
    if (tid == 2) waves_done = 0;
    __syncthreads();
    IH_STAMP(1);

    // ---- _get_dones :179-199, in every thread of the env
    const f3 obj = {os[0] - origin[0], os[1] - origin[1], os[2] - origin[2]};   // :276
    const f4 obj_rot = {os[3], os[4], os[5], os[6]};
    f4 goal = {goal_in[0], goal_in[1], goal_in[2], goal_in[3]};
    const float goal_dist = norm3(obj.x - hand[0], obj.y - hand[1], obj.z - hand[2]);
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
    IH_STAMP(2);
    // ---- compute_rewards :349-420
    float act2 = 0.0f;
    {
        uint32_t k = 0;
        for (; k + 4 <= A; k += 4) {   // four reads in flight, the additions in the order of the plain loop
            const float a0 = act_s[e][k], a1 = act_s[e][k + 1], a2 = act_s[e][k + 2], a3 = act_s[e][k + 3];
            act2 += a0 * a0;
            act2 += a1 * a1;
            act2 += a2 * a2;
            act2 += a3 * a3;
        }
        for (; k < A; ++k) act2 += act_s[e][k] * act_s[e][k];
    }
    float reward = goal_dist * d.dist_reward_scale + div_(1.0f, fabsf(rot_dist) + d.rot_eps) * d.rot_reward_scale +
                   act2 * d.action_penalty_scale;
    const bool goal_reset = at_goal || goal_flag;
    successes = successes + (goal_reset ? 1.0f : 0.0f);
    if (goal_reset) reward = reward + d.reach_goal_bonus;
    if (out_of_reach) reward = reward + d.fall_penalty;
    // resets = where(goal_dist >= fall_dist, 1, reset_buf): reset_buf already contains the fallen envs.  A wavefront's four
    // envs are summed in its first lane, in lane order, and added to the block's sums in LDS; the wavefront that finishes
    // last sends the block's sums on (below, behind the row's stores: an add to an address that every block adds to is
    // slow to leave the CU, and issued here it held up every store behind it)
    {
        const bool counts = live && reset;
        const float one = counts ? 1.0f : 0.0f, fin = counts ? successes : 0.0f;
        float n_reset = 0.0f, n_fin = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            n_reset += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, one), 16 * w));
            n_fin += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), 16 * w));
        }
        if ((tid & 63) == 0 && n_reset != 0.0f) {
            atomicAdd(&block_stats[0], n_reset);
            atomicAdd(&block_stats[1], n_fin);
        }
    }
    IH_STAMP(3);
    // ---- goal resets (:168-171) then env resets (direct_rl_env.py:369-371, :200-247), simulator writes left out
    const uint64_t uenv = static_cast<uint64_t>(env0 + ec), gen_step = d.step_index + (d.step_index_device ? steps_before : 0ull);
    if (goal_reset) {
        const float r0 = d.goal_samples ? sample[0] : 2.0f * inhand_uniform(d.seed, gen_step, uenv, 0) - 1.0f;
        const float r1 = d.goal_samples ? sample[1] : 2.0f * inhand_uniform(d.seed, gen_step, uenv, 1) - 1.0f;
        goal = randomize_rotation(r0, r1);
    }
    if (reset) {
        const float r0 = d.reset_goal_samples ? sample[2] : 2.0f * inhand_uniform(d.seed, gen_step, uenv, 2) - 1.0f;
        const float r1 = d.reset_goal_samples ? sample[3] : 2.0f * inhand_uniform(d.seed, gen_step, uenv, 3) - 1.0f;
        goal = randomize_rotation(r0, r1);
        successes = 0.0f;
        ep = 0;
    }
    const f4 diff = quat_mul(obj_rot, quat_conjugate(goal));
    if (live && g == 0) {
        d.terminated[env] = out_of_reach;
        d.time_outs[env] = time_out;
        if (d.reset_buf) d.reset_buf[env] = reset;
        d.reward[env] = reward;
        if (goal_reset || reset) store4u(d.goal_rot + 4 * env, goal);
        d.reset_goal_buf[env] = 0;  // set by compute_rewards, cleared again by _reset_target_pose for exactly those envs
        d.successes[env] = successes;
        d.episode_length_buf[env] = ep;
    }
    IH_STAMP(4);
    // ---- compute_full_observations :298-322: joint positions and velocities; the object, hand and goal values; the
    // fingertips' states; the actions -- each run dealt over the env's sixteen threads
    const uint32_t row_at = __umul24(e, O);
#pragma unroll
    for (int t = 0; t < kJointRounds; ++t) {
        const uint32_t c = g + 16u * t;
        if (live && c < J) st_b<float>(b_row, 4u * (row_at + c), div_(2.0f * joint[t] - hi[t] - lo[t], hi[t] - lo[t]));   // unscale :330-332
        else if (live && c < 2 * J) st_b<float>(b_row, 4u * (row_at + c), d.vel_obs_scale * joint[t]);
    }
    {
        const float first = g < 3 ? (g == 0 ? obj.x : g == 1 ? obj.y : obj.z)
                          : g < 10 ? misc_os : g < 13 ? d.vel_obs_scale * misc_os : misc_hand;
        const uint32_t h = g & 3;
        const float goal_h = h == 0 ? goal.w : h == 1 ? goal.x : h == 2 ? goal.y : goal.z;
        const float diff_h = h == 0 ? diff.w : h == 1 ? diff.x : h == 2 ? diff.y : diff.z;
        if (live) st_b<float>(b_row, 4u * (row_at + 2 * J + g), first);
        if (live && g < 8) st_b<float>(b_row, 4u * (row_at + 2 * J + 16 + g), g < 4 ? goal_h : diff_h);
    }
#pragma unroll
    for (int t = 0; t < kTipRounds; ++t) {
        const uint32_t k = g + 16u * t;
        if (live && k < 13 * F) st_b<float>(b_row, 4u * (row_at + 2 * J + 24 + k), k < 3 * F ? tip[t] - (tip_a[t] == 0 ? origin[0] : tip_a[t] == 1 ? origin[1] : origin[2]) : tip[t]);
    }
#pragma unroll
    for (int t = 0; t < kActionRounds; ++t) {
        const uint32_t k = g + 16u * t;
        if (live && k < A) st_b<float>(b_row, 4u * (row_at + 2 * J + 24 + 13 * F + k), act[t]);
    }
    if ((tid & 63) == 0 && atomicAdd(&waves_done, 1u) == 3u) {   // LDS takes a wavefront's operations in order: the sums are complete
        const float n_reset = block_stats[0], n_fin = block_stats[1];
        if (d.block_sums) {   // a plain store per block: ilx_inhand_finish adds them up
            d.block_sums[2 * blockIdx.x] = n_reset, d.block_sums[2 * blockIdx.x + 1] = n_fin;
        } else {
            if (n_reset != 0.0f) unsafeAtomicAdd(d.stats, n_reset);
            if (n_fin != 0.0f) unsafeAtomicAdd(d.stats + 1, n_fin);
        }
    }
    IH_STAMP(5);
}
#ifdef ILX_INHAND_STAMPS
}  // namespace ilx
extern "C" int ilx_debug_inhand_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_inhand_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
namespace ilx {
#endif

// :422-431 from the (possibly all-reduced) sums.  One thread -- or, with per-workgroup sums to add up first, one block:
// thread t adds the pairs t, t + 256, ..., the block adds the threads' sums as a tree; the order is fixed, so the result
// does not change from run to run.
__global__ void __launch_bounds__(256) inhand_finish_kernel(ilx_inhand d, int pairs) {
    __shared__ float part[2][256];
    const int tid = threadIdx.x;
    float n_reset = 0.0f, n_fin = 0.0f;
    if (d.block_sums) {
        for (int i = tid; i < pairs; i += 256) n_reset += d.block_sums[2 * i], n_fin += d.block_sums[2 * i + 1];
        part[0][tid] = n_reset, part[1][tid] = n_fin;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) part[0][tid] += part[0][tid + w], part[1][tid] += part[1][tid + w];
            __syncthreads();
        }
        n_reset = part[0][0], n_fin = part[1][0];
    }
    if (tid != 0) return;
    const float num_resets = d.stats[0] + n_reset, finished = d.stats[1] + n_fin;
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
    ILX_LAUNCH(inhand_apply_action_kernel, dim3(blocks_for(d.num_envs * d.num_actuated)), dim3(256), 0, as_stream(stream), d);
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
    if (d.obs && inhand_unrolled_shape(d.num_joints, d.num_fingertips, d.num_actuated) && d.num_fingertips > 0)
        ILX_LAUNCH(inhand_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d);
    else
        ILX_LAUNCH(inhand_step_general_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d);
    ILX_LAUNCH_CHECK("ilx_inhand_step");
    return ILX_OK;
}

extern "C" int ilx_inhand_finish(const ilx_inhand* env, void* stream) {
    ILX_REQUIRE(env, ILX_EINVAL, "ilx_inhand_finish: descriptor is NULL");
    ILX_REQUIRE(env->stats && env->consecutive_successes, ILX_EINVAL, "ilx_inhand_finish: NULL operand");
    const int pairs = static_cast<int>((env->num_envs + kEnvsPerBlock - 1) / kEnvsPerBlock);
    ILX_LAUNCH(inhand_finish_kernel, dim3(1), dim3(env->block_sums ? 256 : 1), 0, as_stream(stream), *env, pairs);
    ILX_LAUNCH_CHECK("ilx_inhand_finish");
    return ILX_OK;
}
