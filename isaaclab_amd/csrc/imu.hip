// IMU sensor (reference: isaaclab/sensors/imu/imu.py:142-182 under sensors/sensor_base.py:197-205, 284-293): pose of the
// sensor frame, body-frame velocities and finite-difference accelerations.  One thread per sensor does the clock and the
// refresh.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

__global__ void __launch_bounds__(256) imu_update_kernel(ilx_imu s, int advance_clock, float dt_clock, float dt) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= s.num_sensors) return;
    float t = s.timestamp[i];
    bool outdated = s.is_outdated[i] != 0;
    if (advance_clock) {  // sensor_base.py:197-205
        t += dt_clock;
        s.timestamp[i] = t;
        outdated = outdated || (t - s.timestamp_last_update[i] + 1.0e-6f >= s.update_period);
    }
    if (!outdated) {
        if (advance_clock) s.is_outdated[i] = 0;
        return;
    }
    const float* tf = s.transforms + 7 * i;
    const f3 pos = {tf[0], tf[1], tf[2]};
    const f4 q = {tf[6], tf[3], tf[4], tf[5]};  // the physics view hands out x, y, z, w
    const f3 off = {s.offset_pos[0], s.offset_pos[1], s.offset_pos[2]};
    const f4 offq = {s.offset_quat[0], s.offset_quat[1], s.offset_quat[2], s.offset_quat[3]};
    const f3 r = quat_rotate(q, off);
    const f4 quat_w = quat_mul(q, offq);
    store3(s.pos_w + 3 * i, f3{pos.x + r.x, pos.y + r.y, pos.z + r.z});
    store4u(s.quat_w + 4 * i, quat_w);
    // velocity of the sensor origin from the velocity of the link's centre of mass (imu.py:160-169)
    const f3 com = load3(s.coms + 7 * i);
    const float* v = s.velocities + 6 * i;
    const f3 ang = {v[3], v[4], v[5]};
    const f3 arm = quat_rotate(q, f3{off.x - com.x, off.y - com.y, off.z - com.z});
    const f3 c = cross3(ang, arm);
    const f3 lin = {v[0] + c.x, v[1] + c.y, v[2] + c.z};
    const f3 pl = load3(s.prev_lin_vel_w + 3 * i), pa = load3(s.prev_ang_vel_w + 3 * i);
    const f3 lin_acc = {div_(lin.x - pl.x, dt) + s.gravity_bias[0], div_(lin.y - pl.y, dt) + s.gravity_bias[1],
                        div_(lin.z - pl.z, dt) + s.gravity_bias[2]};
    const f3 ang_acc = {div_(ang.x - pa.x, dt), div_(ang.y - pa.y, dt), div_(ang.z - pa.z, dt)};
    store3(s.lin_vel_b + 3 * i, quat_rotate_inverse(quat_w, lin));
    store3(s.ang_vel_b + 3 * i, quat_rotate_inverse(quat_w, ang));
    store3(s.lin_acc_b + 3 * i, quat_rotate_inverse(quat_w, lin_acc));
    store3(s.ang_acc_b + 3 * i, quat_rotate_inverse(quat_w, ang_acc));
    store3(s.prev_lin_vel_w + 3 * i, lin);
    store3(s.prev_ang_vel_w + 3 * i, ang);
    s.timestamp_last_update[i] = t;  // sensor_base.py:289-293
    s.is_outdated[i] = 0;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_imu_update(const ilx_imu* imu, int32_t advance_clock, float dt_clock, float dt, void* stream) {
    ILX_REQUIRE(imu, ILX_EINVAL, "ilx_imu_update: descriptor is NULL");
    const ilx_imu& s = *imu;
    ILX_REQUIRE(s.num_sensors >= 0, ILX_EINVAL, "ilx_imu_update: negative sensor count");
    if (s.num_sensors == 0) return ILX_OK;
    ILX_REQUIRE(s.transforms && s.velocities && s.coms, ILX_EINVAL, "ilx_imu_update: the physics views' buffers are NULL");
    ILX_REQUIRE(s.pos_w && s.quat_w && s.lin_vel_b && s.ang_vel_b && s.lin_acc_b && s.ang_acc_b && s.prev_lin_vel_w && s.prev_ang_vel_w,
                ILX_EINVAL, "ilx_imu_update: NULL data buffer");
    ILX_REQUIRE(s.timestamp && s.timestamp_last_update && s.is_outdated, ILX_EINVAL, "ilx_imu_update: NULL clock buffer");
    ILX_REQUIRE(dt > 0.0f, ILX_EINVAL, "ilx_imu_update: the derivative's time step must be positive (update() comes before the first read)");
    ILX_LAUNCH(imu_update_kernel, dim3(blocks_for(s.num_sensors)), dim3(256), 0, as_stream(stream), s, advance_clock, dt_clock, dt);
    ILX_LAUNCH_CHECK("ilx_imu_update");
    return ILX_OK;
}
