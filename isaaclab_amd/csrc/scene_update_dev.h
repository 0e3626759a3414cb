// Contact-sensor bookkeeping and the one-launch scene update as device code, shared by contact.hip (the launches of
// their own) and raycast.hip (the scene update that also prepares the height scanner's refresh).
// Reference: isaaclab/sensors/contact_sensor/contact_sensor.py:314-373, sensors/sensor_base.py:197-205, 284-293,
// scene/interactive_scene.py:496-513, assets/articulation/articulation_data.py:77-82, 445-455.
#pragma once
#include "common.h"
#include "math_dev.h"
#include "uniform_access.h"

namespace ilx {

constexpr int kMaxContactHistory = 8;

struct ContactArgs {
    const float* in;
    float* net;
    float* hist;
    int32_t* head;  // (n) ring mode: slot of the newest history entry per env; null: slot 0 is always the newest
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

// Sensor clock of one env (sensor_base.py:197-205): returns whether the env is refreshed in this launch.
// tick: advance the clock by dt; lazy: only flag outdated envs, refresh nothing.
struct Clock {
    float ts_new, last;
    bool refresh;
};
ILX_DEV Clock read_clock(const float* ts, const float* ts_last, const uint8_t* outdated, int64_t env, int tick, float dt,
                         float period) {
    Clock c{ts[env], ts_last[env], outdated[env] != 0};
    if (tick) {
        c.ts_new = c.ts_new + dt;                                              // sensor_base.py:199
        c.refresh = c.refresh || ((c.ts_new - c.last + 1e-6f) >= period);      // sensor_base.py:200
    }
    return c;
}
// Ring mode: the newest entry moves one slot down (mod h), so that age k sits at slot (head + k) % h.
ILX_DEV int next_slot(int head, int h) { return (head <= 0 || head >= h ? h : head) - 1; }

ILX_DEV void write_clock(float* ts, float* ts_last, uint8_t* outdated, int64_t env, const Clock& c, bool lazy) {
    ts[env] = c.ts_new;
    if (lazy) {
        outdated[env] = c.refresh;
    } else {
        if (c.refresh) ts_last[env] = c.ts_new;  // sensor_base.py:291
        outdated[env] = 0;                       // refreshed now, or it was not outdated to begin with
    }
}

// What a refresh of one (env, body) reads besides the history: loaded ahead of the decision whether the env refreshes,
// so that these loads and the clock's travel together.
struct BodyIn {
    f3 f;
    float cur_air, cur_contact;
};
ILX_DEV BodyIn load_body(const ContactArgs& a, int64_t env, int body) {
    const int64_t eb = env * a.b + body;
    BodyIn in{load3(a.in + 3 * eb), 0.0f, 0.0f};
    if (a.cur_air) in.cur_air = a.cur_air[eb], in.cur_contact = a.cur_contact[eb];
    return in;
}

// contact_sensor.py:324-373 for one (env, body).
// `slot`: ring mode, the history slot this refresh writes.
ILX_DEV void contact_body(const ContactArgs& a, int64_t env, int body, float elapsed, int slot, const BodyIn& in) {
    const int64_t eb = env * a.b + body;
    const f3 f = in.f;
    if (a.net) store3(a.net + 3 * eb, f);  // contact_sensor.py:324 (null: the newest history entry stands for it)
    if (a.hist && a.head) {     // the same history as below, kept as a ring: one row written instead of h rows moved
        store3(a.hist + 3 * ((env * a.h + slot) * a.b + body), f);
    } else if (a.hist) {        // contact_sensor.py:326-328
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
        const float ca = in.cur_air, cc = in.cur_contact;
        const bool first_contact = (ca > 0.0f) && contact;
        const bool first_detached = (cc > 0.0f) && !contact;
        if (first_contact) a.last_air[eb] = ca + elapsed;
        a.cur_air[eb] = contact ? 0.0f : ca + elapsed;
        if (first_detached) a.last_contact[eb] = cc + elapsed;
        a.cur_contact[eb] = contact ? cc + elapsed : 0.0f;
    }
}
ILX_DEV void contact_body(const ContactArgs& a, int64_t env, int body, float elapsed, int slot) {
    contact_body(a, env, body, elapsed, slot, load_body(a, env, body));
}

// InteractiveScene.update(dt) for the usual locomotion scene in ONE launch (scene/interactive_scene.py:496-513):
// articulation -> joint_acc (articulation_data.py:77-82, 445-455); contact sensor -> clock + bookkeeping;
// ray caster -> clock only (its buffers refresh lazily, on the next read).  Thread layout as above: a block owns
// whole envs, `width` threads per env.
struct SceneArgs {
    ContactArgs c;          // c.in == nullptr: no contact sensor
    int contact_lazy;       // history_length == 0 and lazy updates: flag, do not refresh
    const float* joint_vel; // nullptr: no articulation
    float* prev_joint_vel;
    float* joint_acc;
    int j;
    float joint_elapsed;
    float* ray_ts;          // nullptr: no ray caster
    float* ray_ts_last;
    uint8_t* ray_outdated;
    float ray_period;
    int width;
    uint32_t inv_width;     // ceil(2^32 / width): t / width == umulhi(t, inv_width) for the t < 256 of a block
};

// What one thread of a scene update knows afterwards (the fused scene-update + scanner-preparation launch goes on from it).
struct SceneThread {
    int64_t env;
    bool first;   // the env's first thread: the one that owns the env's clocks
    Clock ray;    // first thread, scene with a ray caster: the caster's clock as this update left it (refresh = flagged)
};

// The whole update for the calling thread; contains one __syncthreads(): every thread of the block calls it.
// `block`: the block's index among the update's blocks; `ray_clock_too` = false leaves the ray caster's clock to
// somebody else (the launch that also prepares the scanner's refresh gives it to the workgroups that do that).
// Addresses: a block owns a run of envs, so every array is reached as the run's first element (a scalar base, formed once
// with scalar arithmetic) plus a 32-bit offset per thread; formed per access as 64-bit element indices the addresses were
// two thirds of the kernel's vector instructions, a good part of them quarter-rate integer multiplies.
ILX_DEV SceneThread scene_update_thread(const SceneArgs& s, unsigned block, bool ray_clock_too = true) {
    typedef __attribute__((address_space(1))) char* Base;
    const ContactArgs& a = s.c;
    const uint32_t t = threadIdx.x;
    const uint32_t le = s.width > 1 ? __umulhi(t, s.inv_width) : t;  // t / width
    const uint32_t k0 = t - __umul24(le, static_cast<uint32_t>(s.width));
    const int64_t env0 = static_cast<int64_t>(block) * a.envs_per_block;
    const int64_t env = env0 + le;
    const bool active = le < static_cast<uint32_t>(a.envs_per_block) && env < a.n;
    const bool has_contact = a.in != nullptr;
    const uint32_t nb = static_cast<uint32_t>(a.b), nj = static_cast<uint32_t>(s.j), nh = static_cast<uint32_t>(a.h);
    const Base b_ts = block_base(a.ts, env0), b_ts_last = block_base(a.ts_last, env0), b_outdated = block_base(a.outdated, env0);
    const Base b_head = block_base(a.head, env0);
    const Base b_in = block_base(a.in, env0 * a.b * 3), b_net = block_base(a.net, env0 * a.b * 3);
    const Base b_hist = block_base(a.hist, env0 * a.h * a.b * 3);
    const Base b_cur_air = block_base(a.cur_air, env0 * a.b), b_last_air = block_base(a.last_air, env0 * a.b);
    const Base b_cur_contact = block_base(a.cur_contact, env0 * a.b), b_last_contact = block_base(a.last_contact, env0 * a.b);
    const Base b_vel = block_base(s.joint_vel, env0 * s.j), b_prev = block_base(s.prev_joint_vel, env0 * s.j);
    const Base b_acc = block_base(s.joint_acc, env0 * s.j);
    const Base b_ray_ts = block_base(s.ray_ts, env0), b_ray_last = block_base(s.ray_ts_last, env0);
    const Base b_ray_outdated = block_base(s.ray_outdated, env0);
    auto clock_to = [&](Base ts, Base ts_last, Base outdated, const Clock& c, bool lazy) {   // write_clock
        st_b<float>(ts, 4u * le, c.ts_new);
        if (lazy) {
            st_b<uint8_t>(outdated, le, c.refresh);
        } else {
            if (c.refresh) st_b<float>(ts_last, 4u * le, c.ts_new);  // sensor_base.py:291
            st_b<uint8_t>(outdated, le, 0);                          // refreshed now, or it was not outdated to begin with
        }
    };
    auto body_at = [&](uint32_t body) {   // load_body
        const uint32_t eb = __umul24(le, nb) + body, eb12 = __umul24(eb, 12u);
        BodyIn in{f3{ld_b<float>(b_in, eb12), ld_b<float>(b_in, eb12 + 4u), ld_b<float>(b_in, eb12 + 8u)}, 0.0f, 0.0f};
        if (a.cur_air) in.cur_air = ld_b<float>(b_cur_air, 4u * eb), in.cur_contact = ld_b<float>(b_cur_contact, 4u * eb);
        return in;
    };

    // Everything the thread may read goes out before the barrier, in ONE trip to memory: the clocks, the ring's head, and
    // the inputs of the first body / joint the thread owns (read whether or not the env turns out to refresh).  The loads
    // are unconditional -- a thread without an env reads the block's first one, a thread past the bodies or joints the
    // last of them, an array the scene does not have is replaced by one it has, at offset 0 -- and nothing is computed
    // from them until all are out: a load inside a branch is waited for where the branch ends, and with the clock's
    // arithmetic next to its loads each group of loads was a trip of its own (three before the barrier).
    if (!has_contact && !s.joint_vel && !s.ray_ts) {
        __syncthreads();
        return SceneThread{env, false, Clock{0.0f, 0.0f, false}};
    }
    const Base b_any = has_contact ? b_ts : (s.joint_vel ? b_vel : b_ray_ts);
    auto base_or_any = [&](Base b, const void* p) { return p ? b : b_any; };
    auto at_or_0 = [&](uint32_t off, const void* p) { return p ? off : 0u; };
    const uint32_t le_c = active ? le : 0u;
    const uint32_t eb_c = __umul24(le_c, nb) + min(k0, nb - 1u);
    const uint32_t jt_c = __umul24(le_c, nj) + min(k0, nj - 1u);
    const Base l_ts = base_or_any(b_ts, a.ts), l_ts_last = base_or_any(b_ts_last, a.ts), l_outdated = base_or_any(b_outdated, a.ts);
    const float c_ts = ld_b<float>(l_ts, at_or_0(4u * le_c, a.ts)), c_last = ld_b<float>(l_ts_last, at_or_0(4u * le_c, a.ts));
    const uint32_t c_flag = ld_b<uint8_t>(l_outdated, at_or_0(le_c, a.ts));
    const int32_t head_in = ld_b<int32_t>(base_or_any(b_head, a.head), at_or_0(4u * le_c, a.head));
    const Base l_in = base_or_any(b_in, a.in);
    const uint32_t in_at = at_or_0(__umul24(eb_c, 12u), a.in);
    BodyIn first_body{f3{ld_b<float>(l_in, in_at), ld_b<float>(l_in, in_at + (a.in ? 4u : 0u)), ld_b<float>(l_in, in_at + (a.in ? 8u : 0u))},
                      ld_b<float>(base_or_any(b_cur_air, a.cur_air), at_or_0(4u * eb_c, a.cur_air)),
                      ld_b<float>(base_or_any(b_cur_contact, a.cur_air), at_or_0(4u * eb_c, a.cur_air))};
    float first_v = ld_b<float>(base_or_any(b_vel, s.joint_vel), at_or_0(4u * jt_c, s.joint_vel));
    float first_prev = ld_b<float>(base_or_any(b_prev, s.joint_vel), at_or_0(4u * jt_c, s.joint_vel));
    const float r_ts = ld_b<float>(base_or_any(b_ray_ts, s.ray_ts), at_or_0(4u * le_c, s.ray_ts));
    const float r_last = ld_b<float>(base_or_any(b_ray_last, s.ray_ts), at_or_0(4u * le_c, s.ray_ts));
    const uint32_t r_flag = ld_b<uint8_t>(base_or_any(b_ray_outdated, s.ray_ts), at_or_0(le_c, s.ray_ts));
    // no instruction: all of them are in registers here -- one wait, with no store pending yet.  Left to the compiler each
    // value is waited for at its first use, behind the stores of the values used before it, and since loads and stores
    // share one in-order counter each such wait is then a wait for those stores to be acknowledged as well
    asm volatile("" ::"v"(c_ts), "v"(c_last), "v"(c_flag), "v"(head_in), "v"(first_body.f.x), "v"(first_body.f.y), "v"(first_body.f.z),
                 "v"(first_body.cur_air), "v"(first_body.cur_contact), "v"(first_v), "v"(first_prev), "v"(r_ts), "v"(r_last), "v"(r_flag));
    auto ticked = [&](float ts, float last, uint32_t flag, float period, bool mine) {   // read_clock with tick = 1
        Clock k{ts + a.dt, last, flag != 0};                                    // sensor_base.py:199
        k.refresh = k.refresh || ((k.ts_new - k.last + 1e-6f) >= period);        // sensor_base.py:200
        return mine ? k : Clock{0.0f, 0.0f, false};
    };
    const Clock c = ticked(c_ts, c_last, c_flag, a.period, active && has_contact);
    const int slot = (active && has_contact && a.head) ? next_slot(head_in, a.h) : 0;
    // the ray caster's clock belongs to the env's first thread alone
    const bool ray_clock = active && k0 == 0 && s.ray_ts && ray_clock_too;
    const Clock r = ticked(r_ts, r_last, r_flag, s.ray_period, ray_clock);
    __syncthreads();
    SceneThread out{env, active && k0 == 0, r};
    if (!active) return out;
    if (k0 == 0) {
        if (has_contact) clock_to(b_ts, b_ts_last, b_outdated, c, s.contact_lazy != 0);
        if (has_contact && a.head && c.refresh && !s.contact_lazy) st_b<int32_t>(b_head, 4u * le, slot);
        if (ray_clock) clock_to(b_ray_ts, b_ray_last, b_ray_outdated, r, true);
    }
    // (the thread's first joint and body, whose inputs are in registers, ahead of the loops over further ones: a loop
    // with loads in it makes the compiler wait at its top for everything in flight, the stores issued so far included)
    auto joint_store = [&](uint32_t j, float v, float prev) {
        const uint32_t at = 4u * (__umul24(le, nj) + j);
        st_b<float>(b_acc, at, div_(v - prev, s.joint_elapsed));
        st_b<float>(b_prev, at, v);
    };
    if (s.joint_vel && k0 < nj) joint_store(k0, first_v, first_prev);
    const bool refresh = has_contact && c.refresh && !s.contact_lazy;
    const float elapsed = c.ts_new - c.last;
    auto body_store = [&](uint32_t body, const BodyIn& in) {   // contact_body with the block's addressing
        const uint32_t eb = __umul24(le, nb) + body, eb12 = __umul24(eb, 12u);
        const f3 f = in.f;
        if (a.net) st_b<float>(b_net, eb12, f.x), st_b<float>(b_net, eb12 + 4u, f.y), st_b<float>(b_net, eb12 + 8u, f.z);
        if (a.hist && a.head) {   // the ring: one row written
            const uint32_t at = __umul24(__umul24(__umul24(le, nh) + static_cast<uint32_t>(slot), nb) + body, 12u);
            st_b<float>(b_hist, at, f.x), st_b<float>(b_hist, at + 4u, f.y), st_b<float>(b_hist, at + 8u, f.z);
        } else if (a.hist) {      // contact_sensor.py:326-328: every row moves one slot up
            const uint32_t hb = __umul24(__umul24(__umul24(le, nh), nb) + body, 12u), row = __umul24(nb, 12u);
            f3 old[kMaxContactHistory - 1];
#pragma unroll
            for (uint32_t k = 0; k < kMaxContactHistory - 1; ++k) {   // (a row past the last but one: that one again, not stored)
                const uint32_t at = hb + __umul24(min(k, max(nh, 2u) - 2u), row);
                old[k] = f3{ld_b<float>(b_hist, at), ld_b<float>(b_hist, at + 4u), ld_b<float>(b_hist, at + 8u)};
            }
#pragma unroll
            for (uint32_t k = 0; k < kMaxContactHistory - 1; ++k)   // no instruction: all rows are here before the first store
                asm volatile("" ::"v"(old[k].x), "v"(old[k].y), "v"(old[k].z));
            st_b<float>(b_hist, hb, f.x), st_b<float>(b_hist, hb + 4u, f.y), st_b<float>(b_hist, hb + 8u, f.z);
#pragma unroll
            for (uint32_t k = 0; k < kMaxContactHistory - 1; ++k)
                if (k + 1 < nh) {
                    const uint32_t at = hb + __umul24(k + 1, row);
                    st_b<float>(b_hist, at, old[k].x), st_b<float>(b_hist, at + 4u, old[k].y), st_b<float>(b_hist, at + 8u, old[k].z);
                }
        }
        if (a.cur_air) {  // contact_sensor.py:346-373
            const bool contact = norm3(f) > a.thr;
            const float ca = in.cur_air, cc = in.cur_contact;
            const bool first_contact = (ca > 0.0f) && contact;
            const bool first_detached = (cc > 0.0f) && !contact;
            if (first_contact) st_b<float>(b_last_air, 4u * eb, ca + elapsed);
            st_b<float>(b_cur_air, 4u * eb, contact ? 0.0f : ca + elapsed);
            if (first_detached) st_b<float>(b_last_contact, 4u * eb, cc + elapsed);
            st_b<float>(b_cur_contact, 4u * eb, contact ? cc + elapsed : 0.0f);
        }
    };
    if (refresh && k0 < nb) body_store(k0, first_body);
    if (s.joint_vel) {
        for (uint32_t j = k0 + static_cast<uint32_t>(s.width); j < nj; j += static_cast<uint32_t>(s.width)) {
            const uint32_t at = 4u * (__umul24(le, nj) + j);
            joint_store(j, ld_b<float>(b_vel, at), ld_b<float>(b_prev, at));
        }
    }
    if (refresh)
        for (uint32_t body = k0 + static_cast<uint32_t>(s.width); body < nb; body += static_cast<uint32_t>(s.width)) body_store(body, body_at(body));
    return out;
}

// ilx_scene -> SceneArgs and the launch's grid (argument checks of ilx_scene_update).
inline int scene_args_from(const ilx_scene& d, SceneArgs& s, unsigned& blocks) {
    s = SceneArgs{};
    int width = 1;
    if (d.joint_vel) {
        ILX_REQUIRE(d.previous_joint_vel && d.joint_acc && d.num_joints > 0, ILX_EINVAL,
                    "ilx_scene_update: articulation needs previous_joint_vel, joint_acc and num_joints > 0");
        ILX_REQUIRE(d.joint_acc_elapsed > 0.0f, ILX_EINVAL, "ilx_scene_update: joint_acc_elapsed must be positive");
        s.joint_vel = d.joint_vel, s.prev_joint_vel = d.previous_joint_vel, s.joint_acc = d.joint_acc;
        s.j = d.num_joints, s.joint_elapsed = d.joint_acc_elapsed;
        width = d.num_joints;
    }
    ContactArgs& a = s.c;
    a.n = d.num_envs, a.dt = d.dt;
    if (d.net_forces_in) {
        const int b = d.num_bodies, h = d.history_length;
        ILX_REQUIRE(b > 0 && h >= 0, ILX_EINVAL, "ilx_scene_update: bad contact shape (b=%d, h=%d)", b, h);
        ILX_REQUIRE(h <= kMaxContactHistory, ILX_ESHAPE, "ilx_scene_update: history length %d (max %d)", h, kMaxContactHistory);
        ILX_REQUIRE(d.contact_timestamp && d.contact_timestamp_last_update && d.contact_is_outdated,
                    ILX_EINVAL, "ilx_scene_update: NULL contact operand");
        ILX_REQUIRE(d.net_forces_w || d.net_forces_w_history, ILX_EINVAL,
                    "ilx_scene_update: net_forces_w may be NULL only with a history");
        ILX_REQUIRE((h == 0) == (d.net_forces_w_history == nullptr), ILX_EINVAL,
                    "ilx_scene_update: history must be NULL iff history_length == 0");
        const int timers = (d.current_air_time != nullptr) + (d.last_air_time != nullptr) +
                           (d.current_contact_time != nullptr) + (d.last_contact_time != nullptr);
        ILX_REQUIRE(timers == 0 || timers == 4, ILX_EINVAL, "ilx_scene_update: pass all four timers or none");
        ILX_REQUIRE(d.net_forces_w_history || !d.contact_history_head, ILX_EINVAL, "ilx_scene_update: history head without a history");
        a.in = d.net_forces_in, a.net = d.net_forces_w, a.hist = d.net_forces_w_history, a.head = d.contact_history_head;
        a.cur_air = d.current_air_time, a.last_air = d.last_air_time, a.cur_contact = d.current_contact_time;
        a.last_contact = d.last_contact_time;
        a.ts = d.contact_timestamp, a.ts_last = d.contact_timestamp_last_update, a.outdated = d.contact_is_outdated;
        a.period = d.contact_update_period, a.thr = d.force_threshold, a.tick = 1, a.b = b, a.h = h;
        s.contact_lazy = d.contact_lazy;
        width = b;
    }
    if (d.ray_timestamp) {
        ILX_REQUIRE(d.ray_timestamp_last_update && d.ray_is_outdated, ILX_EINVAL, "ilx_scene_update: NULL ray caster clock");
        s.ray_ts = d.ray_timestamp, s.ray_ts_last = d.ray_timestamp_last_update, s.ray_outdated = d.ray_is_outdated;
        s.ray_period = d.ray_update_period;
    }
    if (width > 256) width = 256;
    s.width = width;
    s.inv_width = static_cast<uint32_t>((0x100000000ull + width - 1) / width);
    a.envs_per_block = 256 / width;
    blocks = static_cast<unsigned>((d.num_envs + a.envs_per_block - 1) / a.envs_per_block);
    return ILX_OK;
}

}  // namespace ilx
