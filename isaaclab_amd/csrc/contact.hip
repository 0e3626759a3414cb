// Contact-sensor bookkeeping (reference: isaaclab/sensors/contact_sensor/contact_sensor.py:314-373 and the
// per-sensor update clocks of isaaclab/sensors/sensor_base.py:197-205, 284-293).
//
// One launch = one ContactSensor.update(dt).  A 256-thread block owns whole environments (floor(256 / B)
// of them) with one thread per (env, body), so the per-env clock words are read by every thread of the env
// before the barrier and written by the body-0 thread after it.  All traffic is contiguous per env:
// forces/history rows are B*3 floats, timer rows B floats.  The device code is in scene_update_dev.h.
#include "scene_update_dev.h"

namespace ilx {

__global__ void __launch_bounds__(256) contact_update_kernel(ContactArgs a) {
    const int t = threadIdx.x;
    const bool wide = a.b > 256;  // more bodies than threads: one env per block, threads stride over bodies
    const int le = wide ? 0 : t / a.b;
    const int b0 = wide ? t : t - le * a.b;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * a.envs_per_block + le;
    const bool active = le < a.envs_per_block && env < a.n;

    Clock c{0.0f, 0.0f, false};
    int slot = 0;
    if (active) c = read_clock(a.ts, a.ts_last, a.outdated, env, a.tick, a.dt, a.period);
    if (active && a.head) slot = next_slot(a.head[env], a.h);
    __syncthreads();
    if (active && b0 == 0) write_clock(a.ts, a.ts_last, a.outdated, env, c, false);
    if (!active || !c.refresh) return;
    if (a.head && b0 == 0) a.head[env] = slot;
    const float elapsed = c.ts_new - c.last;  // contact_sensor.py:349
    for (int body = b0; body < a.b; body += 256) {
        contact_body(a, env, body, elapsed, slot);
        if (!wide) break;
    }
}

__global__ void __launch_bounds__(256) scene_update_kernel(SceneArgs s) {
    request_kernel_arguments<kernel_argument_bytes<SceneArgs>()>();
    scene_update_thread(s, blockIdx.x);
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_contact_update(const float* net_forces_in, float* net_forces_w, float* history, int32_t* history_head,
                                  float* current_air_time, float* last_air_time, float* current_contact_time,
                                  float* last_contact_time, float* timestamp, float* timestamp_last_update,
                                  uint8_t* is_outdated, int advance_clock, float dt, float update_period,
                                  float force_threshold, int64_t n, int b, int h, void* stream) {
    ILX_REQUIRE(n >= 0 && b > 0 && h >= 0, ILX_EINVAL, "ilx_contact_update: bad shape (n=%lld, b=%d, h=%d)", (long long)n,
                b, h);
    ILX_REQUIRE(h <= kMaxContactHistory, ILX_ESHAPE, "ilx_contact_update: history length %d (max %d)", h,
                kMaxContactHistory);
    ILX_REQUIRE(net_forces_in && timestamp && timestamp_last_update && is_outdated, ILX_EINVAL,
                "ilx_contact_update: NULL operand");
    ILX_REQUIRE(net_forces_w || history, ILX_EINVAL, "ilx_contact_update: net_forces_w may be NULL only with a history");
    ILX_REQUIRE((h == 0) == (history == nullptr), ILX_EINVAL, "ilx_contact_update: history must be NULL iff h == 0");
    ILX_REQUIRE(history || !history_head, ILX_EINVAL, "ilx_contact_update: history_head without a history");
    const int timers = (current_air_time != nullptr) + (last_air_time != nullptr) + (current_contact_time != nullptr) +
                       (last_contact_time != nullptr);
    ILX_REQUIRE(timers == 0 || timers == 4, ILX_EINVAL, "ilx_contact_update: pass all four timers or none");
    if (n == 0) return ILX_OK;
    ContactArgs a{net_forces_in, net_forces_w, history, history_head, current_air_time, last_air_time, current_contact_time,
                  last_contact_time, timestamp, timestamp_last_update, is_outdated, dt, update_period, force_threshold,
                  advance_clock, n, b, h, b > 256 ? 1 : 256 / b};
    const unsigned blocks = static_cast<unsigned>((n + a.envs_per_block - 1) / a.envs_per_block);
    ILX_LAUNCH(contact_update_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_contact_update");
    return ILX_OK;
}

extern "C" int ilx_scene_update(const ilx_scene* scene, void* stream) {
    ILX_REQUIRE(scene, ILX_EINVAL, "ilx_scene_update: scene is NULL");
    const ilx_scene& d = *scene;
    ILX_REQUIRE(d.num_envs >= 0, ILX_EINVAL, "ilx_scene_update: negative env count");
    if (d.num_envs == 0) return ILX_OK;
    SceneArgs s{};
    unsigned blocks = 0;
    const int rc = scene_args_from(d, s, blocks);
    if (rc != ILX_OK) return rc;
    ILX_LAUNCH(scene_update_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_scene_update");
    return ILX_OK;
}
