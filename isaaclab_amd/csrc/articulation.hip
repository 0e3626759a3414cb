// Derived ArticulationData quantities (reference: isaaclab/assets/articulation/articulation_data.py).
#include "common.h"
#include "math_dev.h"

namespace ilx {

// One thread per env: body-frame velocities, projected gravity and heading from the (N,13) root state.
// articulation_data.py:411-425, 502-518.
__global__ void __launch_bounds__(256)
root_frame_derive_kernel(const float* __restrict__ root_state, float gx, float gy, float gz, float* __restrict__ lin_b,
                         float* __restrict__ ang_b, float* __restrict__ grav_b, float* __restrict__ heading, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* s = root_state + 13 * i;
    const f4 q = load4u(s + 3);
    if (lin_b) store3(lin_b + 3 * i, quat_rotate_inverse(q, load3(s + 7)));
    if (ang_b) store3(ang_b + 3 * i, quat_rotate_inverse(q, load3(s + 10)));
    if (grav_b) store3(grav_b + 3 * i, quat_rotate_inverse(q, f3{gx, gy, gz}));
    if (heading) {
        const f3 f = quat_apply(q, f3{1.0f, 0.0f, 0.0f});
        heading[i] = atan2f(f.y, f.x);
    }
}

// articulation_data.py:263-279: (N,7) xyzw transforms + (N,6) velocities -> (N,13) wxyz root state.
__global__ void __launch_bounds__(256)
root_state_pack_kernel(const float* __restrict__ tf, const float* __restrict__ vel, float* __restrict__ out, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tf + 7 * i;
    float* o = out + 13 * i;
    o[0] = t[0], o[1] = t[1], o[2] = t[2];
    o[3] = t[6], o[4] = t[3], o[5] = t[4], o[6] = t[5];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[7 + k] = vel[6 * i + k];
}

// articulation_data.py:445-455: acc = (vel - prev) / elapsed; prev <- vel.  One thread per element.
__global__ void __launch_bounds__(256)
joint_acc_kernel(const float* __restrict__ vel, float* __restrict__ prev, float* __restrict__ acc, float elapsed,
                 int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float v = vel[i];
    acc[i] = div_(v - prev[i], elapsed);
    prev[i] = v;
}

// ActionManager.process_action + JointAction.process_actions (action_manager.py:331-332, joint_actions.py:130-139):
// prev_action <- action; action <- raw; processed = raw * scale + offset [clamped].  One thread per element.
__global__ void __launch_bounds__(256)
joint_action_kernel(const float* __restrict__ raw, float* __restrict__ action, float* __restrict__ prev_action,
                    const float* __restrict__ scale, float scale_scalar, const float* __restrict__ offset,
                    float offset_scalar, const float* __restrict__ clip, float* __restrict__ processed, int64_t total,
                    float* __restrict__ log_sums, int log_count, uint64_t* __restrict__ step_index_device) {
    // the opening of an env step rides on its first launch (ilx_joint_action_process_begin): the managers' statistics
    // vector cleared and their device-side step count advanced, by the first workgroup
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < log_count; k += blockDim.x) log_sums[k] = 0.0f;
        if (step_index_device && threadIdx.x == 0) *step_index_device += 1;
    }
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float r = raw[i];
    if (prev_action) prev_action[i] = action[i];
    if (action) action[i] = r;
    float v = r * (scale ? scale[i] : scale_scalar) + (offset ? offset[i] : offset_scalar);
    if (clip) v = fminf(fmaxf(v, clip[2 * i]), clip[2 * i + 1]);
    processed[i] = v;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_joint_action_process(const float* raw_action, float* action, float* prev_action, const float* scale,
                                        float scale_scalar, const float* offset, float offset_scalar, const float* clip,
                                        float* processed, int64_t n, int a, void* stream) {
    ILX_REQUIRE(n >= 0 && a > 0, ILX_EINVAL, "ilx_joint_action_process: bad shape (%lld, %d)", (long long)n, a);
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(raw_action && processed, ILX_EINVAL, "ilx_joint_action_process: NULL operand");
    ILX_REQUIRE(!prev_action || action, ILX_EINVAL, "ilx_joint_action_process: prev_action needs action");
    const int64_t total = n * a;
    ILX_LAUNCH(joint_action_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), raw_action, action,
                       prev_action, scale, scale_scalar, offset, offset_scalar, clip, processed, total, nullptr, 0, nullptr);
    ILX_LAUNCH_CHECK("ilx_joint_action_process");
    return ILX_OK;
}

extern "C" int ilx_joint_action_process_begin(const float* raw_action, float* action, float* prev_action, const float* scale,
                                              float scale_scalar, const float* offset, float offset_scalar, const float* clip,
                                              float* processed, int64_t n, int a, float* log_sums, int32_t log_count,
                                              uint64_t* step_index_device, void* stream) {
    ILX_REQUIRE(n > 0 && a > 0, ILX_EINVAL, "ilx_joint_action_process_begin: bad shape (%lld, %d)", (long long)n, a);
    ILX_REQUIRE(raw_action && processed, ILX_EINVAL, "ilx_joint_action_process_begin: NULL operand");
    ILX_REQUIRE(!prev_action || action, ILX_EINVAL, "ilx_joint_action_process_begin: prev_action needs action");
    ILX_REQUIRE(log_count >= 0 && (log_count == 0 || log_sums), ILX_EINVAL,
                "ilx_joint_action_process_begin: %d floats at a NULL pointer", log_count);
    const int64_t total = n * a;
    ILX_LAUNCH(joint_action_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), raw_action, action,
                       prev_action, scale, scale_scalar, offset, offset_scalar, clip, processed, total, log_sums,
                       log_sums ? log_count : 0, step_index_device);
    ILX_LAUNCH_CHECK("ilx_joint_action_process_begin");
    return ILX_OK;
}

extern "C" int ilx_root_frame_derive(const float* root_state, float gx, float gy, float gz, float* lin_vel_b,
                                     float* ang_vel_b, float* projected_gravity_b, float* heading_w, int64_t n,
                                     void* stream) {
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_root_frame_derive: negative env count");
    ILX_REQUIRE(root_state, ILX_EINVAL, "ilx_root_frame_derive: root_state is NULL");
    if (n == 0) return ILX_OK;
    ILX_LAUNCH(root_frame_derive_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), root_state, gx, gy,
                       gz, lin_vel_b, ang_vel_b, projected_gravity_b, heading_w, n);
    ILX_LAUNCH_CHECK("ilx_root_frame_derive");
    return ILX_OK;
}

extern "C" int ilx_root_state_pack(const float* root_transforms_xyzw, const float* root_velocities, float* root_state,
                                   int64_t n, void* stream) {
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_root_state_pack: negative env count");
    ILX_REQUIRE(root_transforms_xyzw && root_velocities && root_state, ILX_EINVAL, "ilx_root_state_pack: NULL operand");
    if (n == 0) return ILX_OK;
    ILX_LAUNCH(root_state_pack_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), root_transforms_xyzw,
                       root_velocities, root_state, n);
    ILX_LAUNCH_CHECK("ilx_root_state_pack");
    return ILX_OK;
}

extern "C" int ilx_joint_acc(const float* joint_vel, float* previous_joint_vel, float* joint_acc, float time_elapsed,
                             int64_t count, void* stream) {
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_joint_acc: negative element count");
    ILX_REQUIRE(joint_vel && previous_joint_vel && joint_acc, ILX_EINVAL, "ilx_joint_acc: NULL operand");
    if (count == 0) return ILX_OK;
    ILX_LAUNCH(joint_acc_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), joint_vel,
                       previous_joint_vel, joint_acc, time_elapsed, count);
    ILX_LAUNCH_CHECK("ilx_joint_acc");
    return ILX_OK;
}
