// Reset / push event terms (reference: isaaclab/envs/mdp/events.py:610-866): new root and joint states for a list of
// envs, drawn uniformly from per-axis ranges.  The reference hands the result to the simulator
// (write_root_pose_to_sim / write_root_velocity_to_sim / write_joint_state_to_sim); here it lands in the state buffers
// the rest of the step reads.  The unit-interval draws are either given (u, one row per listed env: parity tests) or
// come from Philox keyed by (seed, draw, env id, column), so a reset is reproducible whatever the launch geometry.
#include "common.h"
#include "math_dev.h"
#include "philox.h"

namespace ilx {

struct Range6 {
    float lo[6], hi[6];
};

struct EventRng {
    const float* u;  // (count, width) or null
    int width;
    uint64_t seed, draw;
};
ILX_DEV float draw01(const EventRng& r, int64_t k, int64_t env, int col) {
    return r.u ? r.u[k * r.width + col] : philox_uniform(r.seed, r.draw, static_cast<uint64_t>(env), static_cast<uint32_t>(col));
}
// isaaclab.utils.math.sample_uniform (math.py:1313-1331): rand * (upper - lower) + lower
ILX_DEV float ranged(float u, float lo, float hi) { return u * (hi - lo) + lo; }

// events.py:638-680, one thread per listed env.
__global__ void __launch_bounds__(256)
reset_root_kernel(const int64_t* __restrict__ ids, int64_t count, const float* __restrict__ def, const float* __restrict__ origins,
                  Range6 pose, Range6 vel, EventRng rng, float* __restrict__ root) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t env = ids[k];
    const float* d = def + 13 * env;
    float rp[6], rv[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        rp[c] = ranged(draw01(rng, k, env, c), pose.lo[c], pose.hi[c]);
        rv[c] = ranged(draw01(rng, k, env, 6 + c), vel.lo[c], vel.hi[c]);
    }
    float* out = root + 13 * env;
    const f3 o = origins ? load3(origins + 3 * env) : f3{0.0f, 0.0f, 0.0f};
    out[0] = d[0] + o.x + rp[0], out[1] = d[1] + o.y + rp[1], out[2] = d[2] + o.z + rp[2];  // events.py:669
    const f4 q = quat_mul(f4{d[3], d[4], d[5], d[6]}, quat_from_euler_xyz(rp[3], rp[4], rp[5]));  // :670-671
    out[3] = q.w, out[4] = q.x, out[5] = q.y, out[6] = q.z;
#pragma unroll
    for (int c = 0; c < 6; ++c) out[7 + c] = d[7 + c] + rv[c];  // :677
}

// events.py:735-799, one thread per listed env: position = a flat patch of the env's own sub-terrain + default root
// position, orientation from sampled euler angles, velocity = default + sample.  As in the reference the velocity takes
// the default of row k (the position in the list), not of env_ids[k]: it reads default_root_state[:, 7:13] unindexed.
struct TerrainPatches {
    const float* valid_positions;  // (rows, cols, patches, 3)
    const int64_t* levels;         // (n)
    const int64_t* types;          // (n)
    const int64_t* picks;          // (count) patch index per listed env, or null: Philox
    int32_t cols, patches;
};
__global__ void __launch_bounds__(256)
reset_root_from_terrain_kernel(const int64_t* __restrict__ ids, int64_t count, const float* __restrict__ def, TerrainPatches tp,
                               Range6 pose, Range6 vel, EventRng rng, float* __restrict__ root) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t env = ids[k];
    const int64_t pick = tp.picks ? tp.picks[k]
                                  : static_cast<int64_t>(philox4(rng.seed, rng.draw, static_cast<uint64_t>(env), 0x20000000u).x %
                                                         static_cast<uint32_t>(tp.patches));
    const float* patch = tp.valid_positions + ((tp.levels[env] * tp.cols + tp.types[env]) * tp.patches + pick) * 3;
    const float* d = def + 13 * env;
    float* out = root + 13 * env;
    out[0] = patch[0] + d[0], out[1] = patch[1] + d[1], out[2] = patch[2] + d[2];  // :779-780
    float e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) e[c] = ranged(draw01(rng, k, env, c), pose.lo[3 + c], pose.hi[3 + c]);  // :783-785
    const f4 q = quat_from_euler_xyz(e[0], e[1], e[2]);                                                 // :788
    out[3] = q.w, out[4] = q.x, out[5] = q.y, out[6] = q.z;
    const float* dv = def + 13 * k + 7;
#pragma unroll
    for (int c = 0; c < 6; ++c) out[7 + c] = dv[c] + ranged(draw01(rng, k, env, 3 + c), vel.lo[c], vel.hi[c]);  // :791-795
}

// events.py:802-866, one thread per (listed env, joint).
__global__ void __launch_bounds__(256)
reset_joints_kernel(int scale, const int64_t* __restrict__ ids, int64_t count, int j, const float* __restrict__ def_pos,
                    const float* __restrict__ def_vel, const float* __restrict__ pos_limits, const float* __restrict__ vel_limits,
                    float pos_lo, float pos_hi, float vel_lo, float vel_hi, EventRng rng, float* __restrict__ joint_pos,
                    float* __restrict__ joint_vel) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= count * j) return;
    const int64_t k = t / j;
    const int c = static_cast<int>(t - k * j);
    const int64_t env = ids[k], at = env * j + c;
    const float sp = ranged(draw01(rng, k, env, c), pos_lo, pos_hi), sv = ranged(draw01(rng, k, env, j + c), vel_lo, vel_hi);
    float p = scale ? def_pos[at] * sp : def_pos[at] + sp;
    float v = scale ? def_vel[at] * sv : def_vel[at] + sv;
    p = fminf(fmaxf(p, pos_limits[2 * at]), pos_limits[2 * at + 1]);  // torch.clamp: max with the lower bound first
    const float vl = vel_limits[at];
    v = fminf(fmaxf(v, -vl), vl);
    joint_pos[at] = p;
    joint_vel[at] = v;
}

// events.py:610-635, one thread per (listed env, velocity component).
__global__ void __launch_bounds__(256)
push_velocity_kernel(const int64_t* __restrict__ ids, int64_t count, Range6 vel, EventRng rng, float* __restrict__ root) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= count * 6) return;
    const int64_t k = t / 6;
    const int c = static_cast<int>(t - k * 6);
    const int64_t env = ids[k];
    root[13 * env + 7 + c] += ranged(draw01(rng, k, env, c), vel.lo[c], vel.hi[c]);
}

// events.py:683-732: position = default + origin + sample, orientation uniform on SO(3) as a normalised 4-vector of
// standard normal draws (math.py:1251-1267), velocity = default + sample.  One thread per listed env.
struct NormalRng {
    const float* z;  // (count, 4) standard normal draws, or null: Box-Muller on Philox
    uint64_t seed, draw;
};
__global__ void __launch_bounds__(256)
reset_root_random_orientation_kernel(const int64_t* __restrict__ ids, int64_t count, const float* __restrict__ def,
                                     const float* __restrict__ origins, Range6 pose, Range6 vel, EventRng rng, NormalRng nrm,
                                     float* __restrict__ root) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t env = ids[k];
    const float* d = def + 13 * env;
    float* out = root + 13 * env;
    const f3 o = origins ? load3(origins + 3 * env) : f3{0.0f, 0.0f, 0.0f};
    const float og[3] = {o.x, o.y, o.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = d[c] + og[c] + ranged(draw01(rng, k, env, c), pose.lo[c], pose.hi[c]);
    float g[4];
    if (nrm.z) {
        for (int c = 0; c < 4; ++c) g[c] = nrm.z[4 * k + c];
    } else {  // two Box-Muller pairs from one Philox call
        const u4 r = philox4(nrm.seed, nrm.draw, static_cast<uint64_t>(env), 0x40000000u);
        const float u1 = 1.0f - u01(r.x), u2 = u01(r.y), u3 = 1.0f - u01(r.z), u4v = u01(r.w);  // (0, 1] for the logarithm
        const float ra = sqrt_(-2.0f * logf(u1)), rb = sqrt_(-2.0f * logf(u3));
        g[0] = ra * cosf(6.2831853f * u2), g[1] = ra * sinf(6.2831853f * u2);
        g[2] = rb * cosf(6.2831853f * u4v), g[3] = rb * sinf(6.2831853f * u4v);
    }
    const float nq = fmaxf(norm4(f4{g[0], g[1], g[2], g[3]}), 1.0e-12f);  // torch.nn.functional.normalize(eps=1e-12)
#pragma unroll
    for (int c = 0; c < 4; ++c) out[3 + c] = div_(g[c], nq);
#pragma unroll
    for (int c = 0; c < 6; ++c) out[7 + c] = d[7 + c] + ranged(draw01(rng, k, env, 3 + c), vel.lo[c], vel.hi[c]);
}

// events.py:579-607 with Articulation.set_external_force_and_torque (articulation.py:778-805): uniform forces and torques
// into the wrench buffers of the listed (env, body) pairs.  One thread per (listed env, listed body, component).
__global__ void __launch_bounds__(256)
external_wrench_kernel(const int64_t* __restrict__ ids, int64_t count, const int32_t* __restrict__ body_ids, int nb, int num_bodies,
                       float f_lo, float f_hi, float t_lo, float t_hi, const float* __restrict__ u, uint64_t seed, uint64_t draw,
                       float* __restrict__ force, float* __restrict__ torque) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t per = static_cast<int64_t>(nb) * 3;
    if (t >= count * per) return;
    const int64_t k = t / per;
    const int r = static_cast<int>(t - k * per), b = r / 3, c = r - 3 * b;
    const int64_t env = ids[k];
    const int body = body_ids ? body_ids[b] : b;
    const float uf = u ? u[t] : philox_uniform(seed, draw, static_cast<uint64_t>(env), static_cast<uint32_t>(r));
    const float ut = u ? u[count * per + t] : philox_uniform(seed, draw, static_cast<uint64_t>(env), static_cast<uint32_t>(per + r));
    const int64_t at = (env * num_bodies + body) * 3 + c;
    force[at] = ranged(uf, f_lo, f_hi);
    torque[at] = ranged(ut, t_lo, t_hi);
}

// events.py:911-933 for one articulation: the default root state moved to the env origin and the default joint state,
// copied as they are (no sampling, no clamping).  One thread per (listed env, column) over 13 root + j joint columns.
__global__ void __launch_bounds__(256)
scene_to_default_kernel(const int64_t* __restrict__ ids, int64_t count, int j, const float* __restrict__ def_root,
                        const float* __restrict__ origins, float* __restrict__ root, const float* __restrict__ def_pos,
                        const float* __restrict__ def_vel, float* __restrict__ joint_pos, float* __restrict__ joint_vel) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int width = 13 + j;
    if (t >= count * width) return;
    const int64_t k = t / width;
    const int c = static_cast<int>(t - k * width);
    const int64_t env = ids[k];
    if (c < 13) {
        if (!root) return;
        float v = def_root[13 * env + c];
        if (c < 3 && origins) v += origins[3 * env + c];
        root[13 * env + c] = v;
    } else if (joint_pos) {
        const int64_t at = env * j + (c - 13);
        joint_pos[at] = def_pos[at];
        joint_vel[at] = def_vel[at];
    }
}

// Terrain curriculum: the term of isaaclab_tasks/manager_based/locomotion/velocity/mdp/curriculums.py:26-55 fused with
// TerrainImporter.update_env_origins (isaaclab/terrains/terrain_importer.py:307-322), one thread per listed env.
struct CurriculumArgs {
    const int64_t* ids;
    int64_t count;
    const float* root;        // (n,13)
    const float* command;     // (n,command_dim) or null when the move flags are given
    int command_dim;
    float half_size, max_episode_length_s;
    const uint8_t* move_up;   // (count) or null: decided here from the distance walked
    const uint8_t* move_down; // (count) or null
    int64_t* levels;          // (n)
    const int64_t* types;     // (n)
    const float* terrain_origins;  // (rows, cols, 3)
    int rows, cols;
    const int64_t* random_levels;  // (count) or null: Philox
    uint64_t seed, draw;
    float* origins;           // (n,3), read (distance) then written
};
__global__ void __launch_bounds__(256) terrain_levels_kernel(CurriculumArgs a) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= a.count) return;
    const int64_t env = a.ids[k];
    bool up, down;
    if (a.move_up) {
        up = a.move_up[k] != 0, down = a.move_down[k] != 0;
    } else {
        const float* r = a.root + 13 * env;
        const float* o = a.origins + 3 * env;
        const float distance = norm2(r[0] - o[0], r[1] - o[1]);
        const float* c = a.command + static_cast<int64_t>(a.command_dim) * env;
        up = distance > a.half_size;
        down = distance < norm2(c[0], c[1]) * a.max_episode_length_s * 0.5f;
        down = down && !up;
    }
    int64_t level = a.levels[env] + (up ? 1 : 0) - (down ? 1 : 0);
    if (level >= a.rows) {  // solved the last level: sent to a random one
        level = a.random_levels ? a.random_levels[k]
                                : static_cast<int64_t>(philox4(a.seed, a.draw, static_cast<uint64_t>(env), 0u).x % static_cast<uint32_t>(a.rows));
    } else if (level < 0) {
        level = 0;
    }
    a.levels[env] = level;
    store3(a.origins + 3 * env, load3(a.terrain_origins + 3 * (level * a.cols + a.types[env])));
}

static Range6 range6(const float* host12) {
    Range6 r;
    for (int c = 0; c < 6; ++c) r.lo[c] = host12[c], r.hi[c] = host12[6 + c];
    return r;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_event_reset_root_state_uniform(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                                  const float* env_origins, const float* pose_range_host,
                                                  const float* velocity_range_host, const float* u, uint64_t seed,
                                                  uint64_t draw, float* root_state_w, void* stream) {
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_event_reset_root_state_uniform: negative env count");
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && default_root_state && pose_range_host && velocity_range_host && root_state_w, ILX_EINVAL,
                "ilx_event_reset_root_state_uniform: NULL operand");
    ILX_LAUNCH(reset_root_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), env_ids, count,
                       default_root_state, env_origins, range6(pose_range_host), range6(velocity_range_host),
                       EventRng{u, 12, seed, draw}, root_state_w);
    ILX_LAUNCH_CHECK("ilx_event_reset_root_state_uniform");
    return ILX_OK;
}

extern "C" int ilx_event_reset_root_state_from_terrain(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                                       const float* valid_positions, const int64_t* terrain_levels,
                                                       const int64_t* terrain_types, int32_t num_terrain_cols, int32_t num_patches,
                                                       const int64_t* patch_ids, const float* pose_range_host,
                                                       const float* velocity_range_host, const float* u, uint64_t seed, uint64_t draw,
                                                       float* root_state_w, void* stream) {
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_event_reset_root_state_from_terrain: negative env count");
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && default_root_state && pose_range_host && velocity_range_host && root_state_w, ILX_EINVAL,
                "ilx_event_reset_root_state_from_terrain: NULL operand");
    ILX_REQUIRE(valid_positions && terrain_levels && terrain_types && num_terrain_cols > 0 && num_patches > 0, ILX_EINVAL,
                "ilx_event_reset_root_state_from_terrain: the flat patches need their table, the envs' levels and types, and the table's shape");
    ILX_LAUNCH(reset_root_from_terrain_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), env_ids, count,
                       default_root_state, TerrainPatches{valid_positions, terrain_levels, terrain_types, patch_ids, num_terrain_cols, num_patches},
                       range6(pose_range_host), range6(velocity_range_host), EventRng{u, 9, seed, draw}, root_state_w);
    ILX_LAUNCH_CHECK("ilx_event_reset_root_state_from_terrain");
    return ILX_OK;
}

extern "C" int ilx_event_reset_joints(int by_scale, const int64_t* env_ids, int64_t count, const float* default_joint_pos,
                                      const float* default_joint_vel, const float* soft_joint_pos_limits,
                                      const float* soft_joint_vel_limits, float position_lo, float position_hi,
                                      float velocity_lo, float velocity_hi, const float* u, uint64_t seed, uint64_t draw,
                                      float* joint_pos, float* joint_vel, int j, void* stream) {
    ILX_REQUIRE(count >= 0 && j > 0, ILX_EINVAL, "ilx_event_reset_joints: bad shape (%lld, %d)", (long long)count, j);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && default_joint_pos && default_joint_vel && soft_joint_pos_limits && soft_joint_vel_limits && joint_pos &&
                    joint_vel,
                ILX_EINVAL, "ilx_event_reset_joints: NULL operand");
    ILX_LAUNCH(reset_joints_kernel, dim3(blocks_for(count * j)), dim3(256), 0, as_stream(stream), by_scale, env_ids, count, j,
                       default_joint_pos, default_joint_vel, soft_joint_pos_limits, soft_joint_vel_limits, position_lo, position_hi,
                       velocity_lo, velocity_hi, EventRng{u, 2 * j, seed, draw}, joint_pos, joint_vel);
    ILX_LAUNCH_CHECK("ilx_event_reset_joints");
    return ILX_OK;
}

extern "C" int ilx_event_push_by_setting_velocity(const int64_t* env_ids, int64_t count, const float* velocity_range_host,
                                                  const float* u, uint64_t seed, uint64_t draw, float* root_state_w,
                                                  void* stream) {
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_event_push_by_setting_velocity: negative env count");
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && velocity_range_host && root_state_w, ILX_EINVAL, "ilx_event_push_by_setting_velocity: NULL operand");
    ILX_LAUNCH(push_velocity_kernel, dim3(blocks_for(count * 6)), dim3(256), 0, as_stream(stream), env_ids, count,
                       range6(velocity_range_host), EventRng{u, 6, seed, draw}, root_state_w);
    ILX_LAUNCH_CHECK("ilx_event_push_by_setting_velocity");
    return ILX_OK;
}

extern "C" int ilx_curriculum_terrain_levels(const int64_t* env_ids, int64_t count, const float* root_state_w,
                                             const float* command, int command_dim, float half_terrain_size,
                                             float max_episode_length_s, const uint8_t* move_up, const uint8_t* move_down,
                                             int64_t* terrain_levels, const int64_t* terrain_types,
                                             const float* terrain_origins, int num_rows, int num_cols,
                                             const int64_t* random_levels, uint64_t seed, uint64_t draw, float* env_origins,
                                             void* stream) {
    ILX_REQUIRE(count >= 0 && num_rows > 0 && num_cols > 0, ILX_EINVAL, "ilx_curriculum_terrain_levels: bad shape (%lld envs, %d x %d)",
                (long long)count, num_rows, num_cols);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && terrain_levels && terrain_types && terrain_origins && env_origins, ILX_EINVAL,
                "ilx_curriculum_terrain_levels: NULL operand");
    ILX_REQUIRE((move_up != nullptr) == (move_down != nullptr), ILX_EINVAL,
                "ilx_curriculum_terrain_levels: move_up and move_down come together");
    ILX_REQUIRE(move_up || (root_state_w && command && command_dim >= 2), ILX_EINVAL,
                "ilx_curriculum_terrain_levels: without move flags the root state and a command of width >= 2 are needed");
    CurriculumArgs a{env_ids, count, root_state_w, command, command_dim, half_terrain_size, max_episode_length_s, move_up, move_down,
                     terrain_levels, terrain_types, terrain_origins, num_rows, num_cols, random_levels, seed, draw, env_origins};
    ILX_LAUNCH(terrain_levels_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_curriculum_terrain_levels");
    return ILX_OK;
}

extern "C" int ilx_event_reset_root_state_random_orientation(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                                             const float* env_origins, const float* pose_range_host,
                                                             const float* velocity_range_host, const float* u,
                                                             const float* normal_draws, uint64_t seed, uint64_t draw,
                                                             float* root_state_w, void* stream) {
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_event_reset_root_state_random_orientation: negative env count");
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && default_root_state && pose_range_host && velocity_range_host && root_state_w, ILX_EINVAL,
                "ilx_event_reset_root_state_random_orientation: NULL operand");
    ILX_REQUIRE((u != nullptr) == (normal_draws != nullptr), ILX_EINVAL,
                "ilx_event_reset_root_state_random_orientation: uniform and normal draws are injected together");
    ILX_LAUNCH(reset_root_random_orientation_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), env_ids, count,
                       default_root_state, env_origins, range6(pose_range_host), range6(velocity_range_host),
                       EventRng{u, 9, seed, draw}, NormalRng{normal_draws, seed, draw}, root_state_w);
    ILX_LAUNCH_CHECK("ilx_event_reset_root_state_random_orientation");
    return ILX_OK;
}

extern "C" int ilx_event_apply_external_force_torque(const int64_t* env_ids, int64_t count, const int32_t* body_ids,
                                                     int num_selected_bodies, int num_bodies, float force_lo, float force_hi,
                                                     float torque_lo, float torque_hi, const float* u, uint64_t seed,
                                                     uint64_t draw, float* external_force_b, float* external_torque_b,
                                                     void* stream) {
    ILX_REQUIRE(count >= 0 && num_selected_bodies > 0 && num_selected_bodies <= num_bodies, ILX_EINVAL,
                "ilx_event_apply_external_force_torque: bad shape (%lld envs, %d of %d bodies)", (long long)count,
                num_selected_bodies, num_bodies);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && external_force_b && external_torque_b, ILX_EINVAL, "ilx_event_apply_external_force_torque: NULL operand");
    ILX_LAUNCH(external_wrench_kernel, dim3(blocks_for(count * num_selected_bodies * 3)), dim3(256), 0, as_stream(stream),
                       env_ids, count, body_ids, num_selected_bodies, num_bodies, force_lo, force_hi, torque_lo, torque_hi, u, seed, draw,
                       external_force_b, external_torque_b);
    ILX_LAUNCH_CHECK("ilx_event_apply_external_force_torque");
    return ILX_OK;
}

extern "C" int ilx_event_reset_scene_to_default(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                               const float* env_origins, float* root_state_w, const float* default_joint_pos,
                                               const float* default_joint_vel, float* joint_pos, float* joint_vel, int j,
                                               void* stream) {
    ILX_REQUIRE(count >= 0 && j >= 0, ILX_EINVAL, "ilx_event_reset_scene_to_default: bad shape (%lld, %d)", (long long)count, j);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(env_ids && (root_state_w || joint_pos), ILX_EINVAL, "ilx_event_reset_scene_to_default: NULL operand");
    ILX_REQUIRE(!root_state_w || default_root_state, ILX_EINVAL, "ilx_event_reset_scene_to_default: default_root_state is NULL");
    ILX_REQUIRE(!joint_pos || (j > 0 && default_joint_pos && default_joint_vel && joint_vel), ILX_EINVAL,
                "ilx_event_reset_scene_to_default: the joint state comes with its defaults");
    if (!joint_pos) j = 0;
    ILX_LAUNCH(scene_to_default_kernel, dim3(blocks_for(count * (13 + j))), dim3(256), 0, as_stream(stream), env_ids, count, j,
                       default_root_state, env_origins, root_state_w, default_joint_pos, default_joint_vel, joint_pos, joint_vel);
    ILX_LAUNCH_CHECK("ilx_event_reset_scene_to_default");
    return ILX_OK;
}
