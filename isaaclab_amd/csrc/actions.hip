// The remaining joint-space action terms of isaaclab/envs/mdp/actions: position targets scaled to the joint limits with an
// optional moving average (joint_actions_to_limits.py), open / close commands (binary_joint_actions.py) and the planar
// base driven by forward speed and yaw rate (non_holonomic_actions.py).  One thread per (env, joint) or per env.
#include "common.h"
#include "math_dev.h"

namespace ilx {

struct ToLimitsArgs {
    const float* raw;
    const float* scale;       // (n,a) or null
    float scale_scalar;
    const float* clip;        // (n,a,2) or null
    int rescale;
    const float* limits;      // soft_joint_pos_limits (n,j,2)
    const int32_t* joint_ids; // (a) or null: joint k is column k
    int j;
    const float* alpha;       // (n,a) or null
    float alpha_scalar;
    float* prev_applied;      // (n,a) or null: no moving average
    float* processed;         // (n,a)
    int64_t n;
    int a;
};

// joint_actions_to_limits.py:107-129 and, with prev_applied, :213-229.
__global__ void __launch_bounds__(256) joint_action_to_limits_kernel(ToLimitsArgs p) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= p.n * p.a) return;
    const int64_t env = i / p.a;
    const int k = static_cast<int>(i - env * p.a);
    const int joint = p.joint_ids ? p.joint_ids[k] : k;
    float v = p.raw[i] * (p.scale ? p.scale[i] : p.scale_scalar);
    if (p.clip) v = fminf(fmaxf(v, p.clip[2 * i]), p.clip[2 * i + 1]);
    float lo = 0.0f, hi = 0.0f;
    if (p.rescale || p.prev_applied) {
        const float* l = p.limits + 2 * (env * p.j + joint);
        lo = l[0], hi = l[1];
    }
    if (p.rescale) {  // math.py:61-78 unscale_transform of the action clamped to [-1, 1]
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        const float offset = (lo + hi) * 0.5f;
        v = v * (hi - lo) * 0.5f + offset;
    }
    if (p.prev_applied) {
        const float alpha = p.alpha ? p.alpha[i] : p.alpha_scalar;
        float ema = alpha * v;
        ema = ema + (1.0f - alpha) * p.prev_applied[i];
        v = fminf(fmaxf(ema, lo), hi);
        p.prev_applied[i] = v;
    }
    p.processed[i] = v;
}

// binary_joint_actions.py:118-131: a negative (or, for boolean actions, false) action closes, anything else opens.
__global__ void __launch_bounds__(256)
binary_joint_action_kernel(const float* __restrict__ raw, const uint8_t* __restrict__ raw_bool, const float* __restrict__ open_cmd,
                           const float* __restrict__ close_cmd, const float* __restrict__ clip, float* __restrict__ processed,
                           int64_t n, int j) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n * j) return;
    const int64_t env = i / j;
    const int k = static_cast<int>(i - env * j);
    const bool close = raw_bool ? raw_bool[env] == 0 : raw[env] < 0.0f;
    float v = close ? close_cmd[k] : open_cmd[k];
    if (clip) v = fminf(fmaxf(v, clip[2 * i]), clip[2 * i + 1]);
    processed[i] = v;
}

// non_holonomic_actions.py:147-155: (forward speed, yaw rate) of the body -> velocities of the dummy x, y, yaw joints.
__global__ void __launch_bounds__(256)
non_holonomic_kernel(const float* __restrict__ body_quat, int64_t quat_stride, const float* __restrict__ processed,
                     float* __restrict__ joint_vel_command, int64_t n) {
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (env >= n) return;
    const float yaw = euler_xyz_from_quat(load4u(body_quat + quat_stride * env)).z;
    const float v = processed[2 * env];
    joint_vel_command[3 * env] = cosf(yaw) * v;
    joint_vel_command[3 * env + 1] = sinf(yaw) * v;
    joint_vel_command[3 * env + 2] = processed[2 * env + 1];
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_joint_action_to_limits(const float* raw_action, const float* scale, float scale_scalar, const float* clip,
                                          int rescale_to_limits, const float* soft_joint_pos_limits, const int32_t* joint_ids,
                                          int num_joints, const float* alpha, float alpha_scalar, float* prev_applied_actions,
                                          float* processed, int64_t n, int a, void* stream) {
    ILX_REQUIRE(n >= 0 && a > 0 && num_joints >= a, ILX_EINVAL, "ilx_joint_action_to_limits: bad shape (%lld, %d of %d joints)",
                (long long)n, a, num_joints);
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(raw_action && processed, ILX_EINVAL, "ilx_joint_action_to_limits: NULL operand");
    ILX_REQUIRE(!(rescale_to_limits || prev_applied_actions) || soft_joint_pos_limits, ILX_EINVAL,
                "ilx_joint_action_to_limits: soft_joint_pos_limits is needed to rescale or to average");
    ILX_REQUIRE(!prev_applied_actions || alpha || (alpha_scalar >= 0.0f && alpha_scalar <= 1.0f), ILX_EINVAL,
                "ilx_joint_action_to_limits: Moving average weight must be in the range [0, 1]. Got %g.", alpha_scalar);
    ToLimitsArgs p{raw_action, scale, scale_scalar, clip, rescale_to_limits, soft_joint_pos_limits, joint_ids, num_joints,
                   alpha, alpha_scalar, prev_applied_actions, processed, n, a};
    ILX_LAUNCH(joint_action_to_limits_kernel, dim3(blocks_for(n * a)), dim3(256), 0, as_stream(stream), p);
    ILX_LAUNCH_CHECK("ilx_joint_action_to_limits");
    return ILX_OK;
}

extern "C" int ilx_binary_joint_action(const float* raw_action, const uint8_t* raw_action_bool, const float* open_command,
                                       const float* close_command, const float* clip, float* processed, int64_t n, int j,
                                       void* stream) {
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_binary_joint_action: bad shape (%lld, %d)", (long long)n, j);
    if (n == 0) return ILX_OK;
    ILX_REQUIRE((raw_action != nullptr) != (raw_action_bool != nullptr), ILX_EINVAL,
                "ilx_binary_joint_action: exactly one of the float and the boolean action is given");
    ILX_REQUIRE(open_command && close_command && processed, ILX_EINVAL, "ilx_binary_joint_action: NULL operand");
    ILX_LAUNCH(binary_joint_action_kernel, dim3(blocks_for(n * j)), dim3(256), 0, as_stream(stream), raw_action,
                       raw_action_bool, open_command, close_command, clip, processed, n, j);
    ILX_LAUNCH_CHECK("ilx_binary_joint_action");
    return ILX_OK;
}

extern "C" int ilx_non_holonomic_action_apply(const float* body_quat_w, int64_t body_quat_stride, const float* processed,
                                              float* joint_vel_command, int64_t n, void* stream) {
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_non_holonomic_action_apply: negative env count");
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(body_quat_w && processed && joint_vel_command, ILX_EINVAL, "ilx_non_holonomic_action_apply: NULL operand");
    ILX_REQUIRE(body_quat_stride >= 4, ILX_EINVAL, "ilx_non_holonomic_action_apply: bad quaternion stride");
    ILX_LAUNCH(non_holonomic_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), body_quat_w, body_quat_stride,
                       processed, joint_vel_command, n);
    ILX_LAUNCH_CHECK("ilx_non_holonomic_action_apply");
    return ILX_OK;
}
