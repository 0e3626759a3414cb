// Contact-sensor bookkeeping (reference: isaaclab/sensors/contact_sensor/contact_sensor.py:314-373 and the
// per-sensor update clocks of isaaclab/sensors/sensor_base.py:197-205, 284-293).
//
// One launch = one ContactSensor.update(dt).  A 256-thread block owns whole environments (floor(256 / B)
// of them) with one thread per (env, body), so the per-env clock words are read by every thread of the env
// before the barrier and written by the body-0 thread after it.  All traffic is contiguous per env:
// forces/history rows are B*3 floats, timer rows B floats.
#include "common.h"
#include "math_dev.h"

namespace ilx {

constexpr int kMaxContactHistory = 8;

struct ContactArgs {
    const float* in;
    float* net;
    float* hist;
    float* cur_air;
    float* last_air;
    float* cur_contact;
    float* last_contact;
    float* ts;
    float* ts_last;
    uint8_t* outdated;
    float dt, period, thr;
    int tick;
    int64_t n;
    int b, h;
    int envs_per_block;
};

__global__ void __launch_bounds__(256) contact_update_kernel(ContactArgs a) {
    const int t = threadIdx.x;
    const bool wide = a.b > 256;  // more bodies than threads: one env per block, threads stride over bodies
    const int le = wide ? 0 : t / a.b;
    const int b0 = wide ? t : t - le * a.b;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * a.envs_per_block + le;
    const bool active = le < a.envs_per_block && env < a.n;

    float ts_new = 0.0f, last = 0.0f;
    bool refresh = false;
    if (active) {
        ts_new = a.ts[env];
        last = a.ts_last[env];
        refresh = a.outdated[env] != 0;
        if (a.tick) {
            ts_new = ts_new + a.dt;                                          // sensor_base.py:199
            refresh = refresh || ((ts_new - last + 1e-6f) >= a.period);      // sensor_base.py:200
        }
    }
    __syncthreads();
    if (active && b0 == 0) {
        a.ts[env] = ts_new;
        if (refresh) a.ts_last[env] = ts_new;  // sensor_base.py:291
        a.outdated[env] = 0;                   // refreshed now, or it was not outdated to begin with
    }
    if (!active || !refresh) return;

    const float elapsed = ts_new - last;  // contact_sensor.py:349
    for (int body = b0; body < a.b; body += 256) {
        const int64_t eb = env * a.b + body;
        const f3 f = load3(a.in + 3 * eb);
        store3(a.net + 3 * eb, f);  // contact_sensor.py:324
        if (a.hist) {               // contact_sensor.py:326-328
            const int64_t hb = (env * a.h) * a.b + body;
            f3 old[kMaxContactHistory - 1];
#pragma unroll
            for (int k = 0; k < kMaxContactHistory - 1; ++k)
                if (k < a.h - 1) old[k] = load3(a.hist + 3 * (hb + static_cast<int64_t>(k) * a.b));
            store3(a.hist + 3 * hb, f);
#pragma unroll
            for (int k = 0; k < kMaxContactHistory - 1; ++k)
                if (k < a.h - 1) store3(a.hist + 3 * (hb + static_cast<int64_t>(k + 1) * a.b), old[k]);
        }
        if (a.cur_air) {  // contact_sensor.py:346-373
            const bool contact = norm3(f) > a.thr;
            const float ca = a.cur_air[eb], cc = a.cur_contact[eb];
            const bool first_contact = (ca > 0.0f) && contact;
            const bool first_detached = (cc > 0.0f) && !contact;
            if (first_contact) a.last_air[eb] = ca + elapsed;
            a.cur_air[eb] = contact ? 0.0f : ca + elapsed;
            if (first_detached) a.last_contact[eb] = cc + elapsed;
            a.cur_contact[eb] = contact ? cc + elapsed : 0.0f;
        }
        if (!wide) break;
    }
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_contact_update(const float* net_forces_in, float* net_forces_w, float* history,
                                  float* current_air_time, float* last_air_time, float* current_contact_time,
                                  float* last_contact_time, float* timestamp, float* timestamp_last_update,
                                  uint8_t* is_outdated, int advance_clock, float dt, float update_period,
                                  float force_threshold, int64_t n, int b, int h, void* stream) {
    ILX_REQUIRE(n >= 0 && b > 0 && h >= 0, ILX_EINVAL, "ilx_contact_update: bad shape (n=%lld, b=%d, h=%d)", (long long)n,
                b, h);
    ILX_REQUIRE(h <= kMaxContactHistory, ILX_ESHAPE, "ilx_contact_update: history length %d (max %d)", h,
                kMaxContactHistory);
    ILX_REQUIRE(net_forces_in && net_forces_w && timestamp && timestamp_last_update && is_outdated, ILX_EINVAL,
                "ilx_contact_update: NULL operand");
    ILX_REQUIRE((h == 0) == (history == nullptr), ILX_EINVAL, "ilx_contact_update: history must be NULL iff h == 0");
    const int timers = (current_air_time != nullptr) + (last_air_time != nullptr) + (current_contact_time != nullptr) +
                       (last_contact_time != nullptr);
    ILX_REQUIRE(timers == 0 || timers == 4, ILX_EINVAL, "ilx_contact_update: pass all four timers or none");
    if (n == 0) return ILX_OK;
    ContactArgs a{net_forces_in, net_forces_w, history, current_air_time, last_air_time, current_contact_time,
                  last_contact_time, timestamp, timestamp_last_update, is_outdated, dt, update_period, force_threshold,
                  advance_clock, n, b, h, b > 256 ? 1 : 256 / b};
    const unsigned blocks = static_cast<unsigned>((n + a.envs_per_block - 1) / a.envs_per_block);
    hipLaunchKernelGGL(contact_update_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_contact_update");
    return ILX_OK;
}
