// Command terms that run as their own launch.
// 2-D pose command terms (reference: isaaclab/envs/mdp/commands/pose_2d_command.py on top of
// isaaclab/managers/command_manager.py:151-187): a target position in the world and a heading, re-expressed every step in
// the robot's base frame.  One thread per env does the whole CommandTerm.compute / reset for that env.
#include "common.h"
#include "math_dev.h"
#include "philox.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

struct Pose2dArgs {
    ilx_pose2d_command c;
    int reset;  // 0: compute(dt); 1: reset of the envs flagged in reset_mask
    float dt;
    const uint8_t* reset_mask;
    float* log_sums;  // reset: [sum error_pos, sum error_heading, sum error_pos_2d, count]
};

ILX_DEV float pose2d_draw(const ilx_pose2d_command& c, const float* injected, int64_t env, int col, float lo, float hi, uint32_t salt) {
    if (injected) return injected[4 * env + col];  // the value itself, as torch's uniform_ produced it
    return philox_uniform(c.seed, c.draw, static_cast<uint64_t>(env), salt + static_cast<uint32_t>(col)) * (hi - lo) + lo;
}

// command_manager.py:172-187 + pose_2d_command.py:93-122 (uniform) / :177-206 (terrain based)
ILX_DEV void pose2d_resample(const ilx_pose2d_command& c, const float* injected, const int64_t* injected_ids, int64_t env, f3 root_pos,
                             float heading_w, uint32_t salt, float& time_left, int64_t& counter, f3& pos_w, float& head_w) {
    time_left = pose2d_draw(c, injected, env, 0, c.resampling_time_range[0], c.resampling_time_range[1], salt);
    counter += 1;
    if (c.valid_targets) {
        int64_t pick;
        if (injected_ids) pick = injected_ids[env];
        else pick = static_cast<int64_t>(philox4(c.seed, c.draw, static_cast<uint64_t>(env), salt + 8u).x % static_cast<uint32_t>(c.num_patches));
        const int64_t at = ((c.terrain_levels[env] * c.num_terrain_cols + c.terrain_types[env]) * c.num_patches + pick) * 3;
        pos_w = load3(c.valid_targets + at);
    } else {
        const f3 o = load3(c.env_origins + 3 * env);
        pos_w = {o.x + pose2d_draw(c, injected, env, 1, c.pos_x[0], c.pos_x[1], salt),
                 o.y + pose2d_draw(c, injected, env, 2, c.pos_y[0], c.pos_y[1], salt), o.z};
    }
    pos_w.z += c.default_root_state[13 * env + 2];
    if (c.simple_heading) {  // :103-119: towards the target or away from it, whichever is the smaller turn
        const float dir = atan2f(pos_w.y - root_pos.y, pos_w.x - root_pos.x);
        const float flipped = wrap_to_pi(dir + kPi);
        const float to_dir = fabsf(wrap_to_pi(dir - heading_w)), to_flipped = fabsf(wrap_to_pi(flipped - heading_w));
        head_w = to_dir < to_flipped ? dir : flipped;
    } else {
        head_w = pose2d_draw(c, injected, env, 3, c.heading[0], c.heading[1], salt);
    }
}

__global__ void __launch_bounds__(256) pose2d_command_kernel(Pose2dArgs a) {
    const ilx_pose2d_command& c = a.c;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (env >= c.num_envs) return;
    const float* rs = c.root_state_w + 13 * env;
    const f3 root_pos = {rs[0], rs[1], rs[2]};
    const f4 q = {rs[3], rs[4], rs[5], rs[6]};
    const f3 fwd = quat_apply(q, f3{1.0f, 0.0f, 0.0f});  // articulation_data.py:416-425
    const float heading_w = atan2f(fwd.y, fwd.x);
    f3 pos_w = load3(c.pos_command_w + 3 * env);
    float head_w = c.heading_command_w[env];
    float time_left = c.time_left[env];
    int64_t counter = c.command_counter[env];
    if (a.reset) {  // command_manager.py:120-149
        if (!a.reset_mask[env]) return;
        if (a.log_sums) {
            atomicAdd(a.log_sums + 0, c.error_pos[env]);
            atomicAdd(a.log_sums + 1, c.error_heading[env]);
            atomicAdd(a.log_sums + 2, c.error_pos_2d[env]);
            atomicAdd(a.log_sums + 3, 1.0f);
        }
        c.error_pos[env] = c.error_heading[env] = c.error_pos_2d[env] = 0.0f;
        counter = 0;
        pose2d_resample(c, c.reset_samples, c.reset_patch_ids, env, root_pos, heading_w, 16u, time_left, counter, pos_w, head_w);
    } else {  // command_manager.py:151-166
        c.error_pos_2d[env] = norm2(pos_w.x - root_pos.x, pos_w.y - root_pos.y);   // pose_2d_command.py:88-91
        c.error_heading[env] = fabsf(wrap_to_pi(head_w - heading_w));
        time_left = time_left - a.dt;
        if (time_left <= 0.0f)
            pose2d_resample(c, c.samples, c.patch_ids, env, root_pos, heading_w, 0u, time_left, counter, pos_w, head_w);
        // pose_2d_command.py:124-128
        const f3 rel = quat_rotate_inverse(yaw_quat(q), f3{pos_w.x - root_pos.x, pos_w.y - root_pos.y, pos_w.z - root_pos.z});
        store3(c.pos_command_b + 3 * env, rel);
        c.heading_command_b[env] = wrap_to_pi(head_w - heading_w);
    }
    store3(c.pos_command_w + 3 * env, pos_w);
    c.heading_command_w[env] = head_w;
    c.time_left[env] = time_left;
    c.command_counter[env] = counter;
}

static int check_pose2d(const ilx_pose2d_command& c, const char* who) {
    ILX_REQUIRE(c.num_envs >= 0, ILX_EINVAL, "%s: negative env count", who);
    if (c.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(c.root_state_w && c.default_root_state && c.pos_command_w && c.heading_command_w && c.pos_command_b &&
                    c.heading_command_b && c.time_left && c.command_counter && c.error_pos && c.error_heading && c.error_pos_2d,
                ILX_EINVAL, "%s: NULL operand", who);
    if (c.valid_targets)
        ILX_REQUIRE(c.terrain_levels && c.terrain_types && c.num_patches > 0 && c.num_terrain_cols > 0, ILX_EINVAL,
                    "%s: terrain-based targets need terrain_levels, terrain_types and the patch table's shape", who);
    else
        ILX_REQUIRE(c.env_origins, ILX_EINVAL, "%s: env_origins is NULL", who);
    return ILX_OK;
}

// NormalVelocityCommand (velocity_command.py:218-287 under command_manager.py:120-187): a base-frame velocity command drawn
// from a normal distribution with random sign; per-component and whole-command zeroing by coin flips.
struct NormalVelArgs {
    ilx_normal_velocity_command c;
    int reset;
    float dt;
    const uint8_t* reset_mask;
    float* log_sums;  // reset: [sum error_vel_xy, sum error_vel_yaw, count]
};

// velocity_command.py:252-272; ``s`` is the env's row of 11 injected values or null
ILX_DEV void normal_vel_resample(const ilx_normal_velocity_command& c, const float* s, int64_t env, uint32_t salt, float& time_left,
                                 int64_t& counter, float* v, bool* zero, bool& standing) {
    float g[3], coin[8];
    if (s) {
        time_left = s[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = s[1 + k];
#pragma unroll
        for (int k = 0; k < 7; ++k) coin[k] = s[4 + k];
    } else {
        const u4 a = philox4(c.seed, c.draw, static_cast<uint64_t>(env), salt), b = philox4(c.seed, c.draw, static_cast<uint64_t>(env), salt + 1u),
                 e = philox4(c.seed, c.draw, static_cast<uint64_t>(env), salt + 2u);
        time_left = u01(a.x) * (c.resampling_time_range[1] - c.resampling_time_range[0]) + c.resampling_time_range[0];
        const float u1 = 1.0f - u01(a.y), u2 = u01(a.z), u3 = 1.0f - u01(a.w), u4v = u01(b.x);  // (0, 1] for the logarithm
        const float ra = sqrt_(-2.0f * logf(u1)), rb = sqrt_(-2.0f * logf(u3));
        const float n0 = ra * cosf(6.2831853f * u2), n1 = ra * sinf(6.2831853f * u2), n2 = rb * cosf(6.2831853f * u4v);
        g[0] = n0 * c.std_vel[0] + c.mean_vel[0], g[1] = n1 * c.std_vel[1] + c.mean_vel[1], g[2] = n2 * c.std_vel[2] + c.mean_vel[2];
        coin[0] = u01(b.y), coin[1] = u01(b.z), coin[2] = u01(b.w), coin[3] = u01(e.x), coin[4] = u01(e.y), coin[5] = u01(e.z), coin[6] = u01(e.w);
    }
    counter += 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = g[k] * (coin[k] <= 0.5f ? 1.0f : -1.0f);
        zero[k] = coin[3 + k] <= c.zero_prob[k];
    }
    standing = coin[6] <= c.rel_standing_envs;
}

__global__ void __launch_bounds__(256) normal_velocity_command_kernel(NormalVelArgs a) {
    const ilx_normal_velocity_command& c = a.c;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (env >= c.num_envs) return;
    float v[3] = {c.vel_command_b[3 * env], c.vel_command_b[3 * env + 1], c.vel_command_b[3 * env + 2]};
    bool zero[3] = {c.is_zero_vel_env[3 * env] != 0, c.is_zero_vel_env[3 * env + 1] != 0, c.is_zero_vel_env[3 * env + 2] != 0};
    bool standing = c.is_standing_env[env] != 0;
    float time_left = c.time_left[env];
    int64_t counter = c.command_counter[env];
    if (a.reset) {  // command_manager.py:120-149
        if (!a.reset_mask[env]) return;
        if (a.log_sums) {
            atomicAdd(a.log_sums + 0, c.error_vel_xy[env]);
            atomicAdd(a.log_sums + 1, c.error_vel_yaw[env]);
            atomicAdd(a.log_sums + 2, 1.0f);
        }
        c.error_vel_xy[env] = c.error_vel_yaw[env] = 0.0f;
        counter = 0;
        normal_vel_resample(c, c.reset_samples ? c.reset_samples + 11 * env : nullptr, env, 16u, time_left, counter, v, zero, standing);
    } else {  // command_manager.py:151-166
        const float* rs = c.root_state_w + 13 * env;
        const f4 q = {rs[3], rs[4], rs[5], rs[6]};
        const f3 lin_b = quat_rotate_inverse(q, f3{rs[7], rs[8], rs[9]}), ang_b = quat_rotate_inverse(q, f3{rs[10], rs[11], rs[12]});
        c.error_vel_xy[env] += div_(norm2(v[0] - lin_b.x, v[1] - lin_b.y), c.max_command_step);  // velocity_command.py:111-121
        c.error_vel_yaw[env] += div_(fabsf(v[2] - ang_b.z), c.max_command_step);
        time_left = time_left - a.dt;
        if (time_left <= 0.0f)
            normal_vel_resample(c, c.samples ? c.samples + 11 * env : nullptr, env, 0u, time_left, counter, v, zero, standing);
        if (standing) v[0] = v[1] = v[2] = 0.0f;  // velocity_command.py:274-287
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (zero[k]) v[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c.vel_command_b[3 * env + k] = v[k];
        c.is_zero_vel_env[3 * env + k] = zero[k];
    }
    c.is_standing_env[env] = standing;
    c.time_left[env] = time_left;
    c.command_counter[env] = counter;
}

static int check_normal_vel(const ilx_normal_velocity_command& c, const char* who) {
    ILX_REQUIRE(c.num_envs >= 0, ILX_EINVAL, "%s: negative env count", who);
    if (c.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(c.root_state_w && c.vel_command_b && c.is_zero_vel_env && c.is_standing_env && c.time_left && c.command_counter &&
                    c.error_vel_xy && c.error_vel_yaw,
                ILX_EINVAL, "%s: NULL operand", who);
    ILX_REQUIRE(c.max_command_step > 0.0f, ILX_EINVAL, "%s: max_command_step must be positive", who);
    return ILX_OK;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_pose2d_command_compute(const ilx_pose2d_command* cmd, float dt, void* stream) {
    ILX_REQUIRE(cmd, ILX_EINVAL, "ilx_pose2d_command_compute: descriptor is NULL");
    if (const int s = check_pose2d(*cmd, "ilx_pose2d_command_compute")) return s;
    if (cmd->num_envs == 0) return ILX_OK;
    Pose2dArgs a{*cmd, 0, dt, nullptr, nullptr};
    ILX_LAUNCH(pose2d_command_kernel, dim3(blocks_for(cmd->num_envs)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_pose2d_command_compute");
    return ILX_OK;
}

extern "C" int ilx_pose2d_command_reset(const ilx_pose2d_command* cmd, const uint8_t* reset_mask, float* log_sums, void* stream) {
    ILX_REQUIRE(cmd, ILX_EINVAL, "ilx_pose2d_command_reset: descriptor is NULL");
    if (const int s = check_pose2d(*cmd, "ilx_pose2d_command_reset")) return s;
    if (cmd->num_envs == 0) return ILX_OK;
    ILX_REQUIRE(reset_mask, ILX_EINVAL, "ilx_pose2d_command_reset: reset_mask is NULL");
    Pose2dArgs a{*cmd, 1, 0.0f, reset_mask, log_sums};
    ILX_LAUNCH(pose2d_command_kernel, dim3(blocks_for(cmd->num_envs)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_pose2d_command_reset");
    return ILX_OK;
}

extern "C" int ilx_normal_velocity_command_compute(const ilx_normal_velocity_command* cmd, float dt, void* stream) {
    ILX_REQUIRE(cmd, ILX_EINVAL, "ilx_normal_velocity_command_compute: descriptor is NULL");
    if (const int s = check_normal_vel(*cmd, "ilx_normal_velocity_command_compute")) return s;
    if (cmd->num_envs == 0) return ILX_OK;
    NormalVelArgs a{*cmd, 0, dt, nullptr, nullptr};
    ILX_LAUNCH(normal_velocity_command_kernel, dim3(blocks_for(cmd->num_envs)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_normal_velocity_command_compute");
    return ILX_OK;
}

extern "C" int ilx_normal_velocity_command_reset(const ilx_normal_velocity_command* cmd, const uint8_t* reset_mask, float* log_sums,
                                                 void* stream) {
    ILX_REQUIRE(cmd, ILX_EINVAL, "ilx_normal_velocity_command_reset: descriptor is NULL");
    if (const int s = check_normal_vel(*cmd, "ilx_normal_velocity_command_reset")) return s;
    if (cmd->num_envs == 0) return ILX_OK;
    ILX_REQUIRE(reset_mask, ILX_EINVAL, "ilx_normal_velocity_command_reset: reset_mask is NULL");
    NormalVelArgs a{*cmd, 1, 0.0f, reset_mask, log_sums};
    ILX_LAUNCH(normal_velocity_command_kernel, dim3(blocks_for(cmd->num_envs)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_normal_velocity_command_reset");
    return ILX_OK;
}
