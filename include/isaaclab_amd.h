/*
 * isaaclab_amd — C ABI of the MI355X (gfx950) post-physics rollout kernels.
 *
 * The reference (Isaac Lab) has no FFI for this path: its seams are Python call signatures
 * (manager term functions, ActuatorBase.compute, raycast_mesh, DifferentialIKController,
 * isaaclab.utils.math).  Each entry point below names the reference interface it stands behind
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless the name ends in _host;
 *   - the library never allocates or frees caller buffers and never synchronises the stream;
 *     work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *   - all tensors are dense row-major float32 unless stated; quaternions are (w, x, y, z);
 *   - every function returns ILX_OK (0) or a negative error code and leaves a message retrievable
 *     with ilx_last_error() (thread-local);
 *   - functions are re-entrant; the only library-owned state is inside ilx_mesh handles, which are
 *     immutable after creation.
 */
#ifndef ISAACLAB_AMD_H
#define ISAACLAB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILX_OK 0
#define ILX_EINVAL (-1) /* bad argument value (null pointer, unknown op, negative size) */
#define ILX_ESHAPE (-2) /* shape outside what the kernel supports */
#define ILX_EHIP (-3)   /* HIP runtime error; message holds hipGetErrorString */

#define ILX_ABI_VERSION 1

int ilx_version(void);
const char* ilx_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Command lists.  No counterpart in the reference (its launches are torch's).  A step of a few short
 * launches is bound by what the host pays to issue it: a captured HIP graph costs 14 us per replay
 * however few nodes it has, a launch issued from C 2.9 us.  Between ilx_commands_begin and
 * ilx_commands_end every call of this library made by the calling thread keeps its kernel launches
 * -- kernel, grid and a copy of the arguments, device pointers included -- in the list instead of
 * issuing them (nothing runs; argument checks and host-side set-up happen as usual).
 * ilx_commands_replay issues the kept launches in order on `stream`, one hipLaunchKernel each, with
 * no allocation and no other host work.  Like a captured graph the list holds the pointers it was
 * recorded with: the buffers must stay where they are, and whatever changes from step to step must
 * live in device memory (e.g. ilx_mdp::step_index_device).  The recorded calls must all name one
 * stream (a list is one chain; ilx_commands_end fails otherwise).  A list may be replayed from any
 * thread, by one thread at a time.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ilx_commands_s* ilx_commands_t;
int ilx_commands_begin(ilx_commands_t* out);                 /* start recording on the calling thread */
int ilx_commands_end(void);                                  /* stop recording */
int ilx_commands_count(ilx_commands_t commands, int* count); /* number of launches kept */
int ilx_commands_replay(ilx_commands_t commands, void* stream);
int ilx_commands_destroy(ilx_commands_t commands);

/* ------------------------------------------------------------------------------------------------
 * isaaclab.utils.math — source/isaaclab/isaaclab/utils/math.py
 * One launch evaluates one function over n rows.  Operand meaning per op:
 *   a, b, c, d : inputs (row widths in the table), out0/out1 : outputs.
 * ------------------------------------------------------------------------------------------------ */
enum ilx_math_op {
    ILX_QUAT_ROTATE = 0,         /* math.py:583  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_ROTATE_INVERSE = 1, /* math.py:606  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_APPLY = 2,          /* math.py:546  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_APPLY_YAW = 3,      /* math.py:568  a=q(4) b=v(3)            -> out0(3) */
    ILX_YAW_QUAT = 4,            /* math.py:522  a=q(4)                   -> out0(4) */
    ILX_QUAT_MUL = 5,            /* math.py:464  a=q1(4) b=q2(4)          -> out0(4) */
    ILX_QUAT_CONJUGATE = 6,      /* math.py:224  a=q(4)                   -> out0(4) */
    ILX_QUAT_INV = 7,            /* math.py:239  a=q(4)                   -> out0(4) */
    ILX_QUAT_UNIQUE = 8,         /* math.py:448  a=q(4)                   -> out0(4) */
    ILX_NORMALIZE3 = 9,          /* math.py:82   a=v(3)                   -> out0(3) */
    ILX_NORMALIZE4 = 10,         /* math.py:82   a=q(4)                   -> out0(4) */
    ILX_MATRIX_FROM_QUAT = 11,   /* math.py:144  a=q(4)                   -> out0(9) */
    ILX_QUAT_BOX_MINUS = 12,     /* math.py:503  a=q1(4) b=q2(4)          -> out0(3) */
    ILX_AXIS_ANGLE_FROM_QUAT = 13, /* math.py:646 a=q(4)                  -> out0(3) */
    ILX_QUAT_ERROR_MAGNITUDE = 14, /* math.py:678 a=q1(4) b=q2(4)         -> out0(1) */
    ILX_QUAT_FROM_ANGLE_AXIS = 15, /* math.py:629 a=angle(1) b=axis(3)    -> out0(4) */
    ILX_QUAT_FROM_EULER_XYZ = 16,  /* math.py:252 a=roll(1) b=pitch(1) c=yaw(1) -> out0(4) */
    ILX_EULER_XYZ_FROM_QUAT = 17,  /* math.py:414 a=q(4)                  -> out0(3) = (roll,pitch,yaw) */
    ILX_COMBINE_FRAME = 18,      /* math.py:750  a=t01(3) b=q01(4) c=t12(3) d=q12(4) -> out0=t02(3) out1=q02(4) */
    ILX_SUBTRACT_FRAME = 19,     /* math.py:785  a=t01(3) b=q01(4) c=t02(3) d=q02(4) -> out0=t12(3) out1=q12(4) */
    ILX_POSE_ERROR_AA = 20,      /* math.py:820  a=t01(3) b=q01(4) c=t02(3) d=q02(4) -> out0=dp(3) out1=axis-angle(3) */
    ILX_POSE_ERROR_QUAT = 21,    /* math.py:820  same inputs                         -> out0=dp(3) out1=quat(4) */
    ILX_APPLY_DELTA_POSE = 22,   /* math.py:873  a=pos(3) b=quat(4) c=delta(6)       -> out0=pos(3) out1=quat(4) */
    ILX_WRAP_TO_PI = 23,         /* math.py:96   a=angle(1)               -> out0(1) */
    ILX_SKEW = 24,               /* math.py:693  a=v(3)                   -> out0(9) */
    ILX_MATH_OP_COUNT = 25
};
int ilx_math_op(int op, const float* a, const float* b, const float* c, const float* d, float* out0,
                float* out1, int64_t n, void* stream);

/* math.py:23 / :44 / :65 — scale_transform (mode 0), unscale_transform (1), saturate (2) over n x w
 * elements; lower/upper are (w,) when bounds_per_row == 0, else (n, w). */
int ilx_math_range_op(int mode, const float* x, const float* lower, const float* upper, int bounds_per_row,
                      float* out, int64_t n, int w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ArticulationData derived quantities — source/isaaclab/isaaclab/assets/articulation/articulation_data.py
 * ------------------------------------------------------------------------------------------------ */

/* articulation_data.py:263-279  root_state_w (n,13) = [pos, quat wxyz, lin vel, ang vel] from the physics
 * view's root transforms (n,7; quaternion xyzw) and root velocities (n,6). */
int ilx_root_state_pack(const float* root_transforms_xyzw, const float* root_velocities, float* root_state,
                        int64_t n, void* stream);

/* articulation_data.py:502-518 (root_lin_vel_b, root_ang_vel_b), :411-414 (projected_gravity_b, gravity
 * direction (gx,gy,gz) already normalised), :416-425 (heading_w).  Any output may be NULL. */
int ilx_root_frame_derive(const float* root_state, float gx, float gy, float gz, float* lin_vel_b,
                          float* ang_vel_b, float* projected_gravity_b, float* heading_w, int64_t n,
                          void* stream);

/* articulation_data.py:445-455  joint_acc = (joint_vel - previous_joint_vel) / time_elapsed, then
 * previous_joint_vel <- joint_vel; `count` = n * num_joints elements. */
int ilx_joint_acc(const float* joint_vel, float* previous_joint_vel, float* joint_acc, float time_elapsed,
                  int64_t count, void* stream);

/* Samplers of isaaclab.utils.math (source/isaaclab/isaaclab/utils/math.py:1225-1390), out (rows, width):
 *   UNIFORM      rand * (upper - lower) + lower            (:1313-1331)   a = lower, b = upper
 *   LOG_UNIFORM  exp(uniform(log lower, log upper))        (:1334-1368)
 *   TRIANGLE     triangular distribution on [lower, upper] (:1288-1310)
 *   GAUSSIAN     mean + std * N(0, 1)                      (:1371-1390)   a = mean, b = std
 *   RANDOM_ORIENTATION      normalised 4-vector of N(0, 1) (:1251-1267)   width 4, a / b unused
 *   RANDOM_YAW_ORIENTATION  quat_from_euler_xyz(0, 0, 2 pi rand) (:1270-1285), width 4
 * a / b are device tensors addressed as a[row * a_row_stride + col * a_col_stride] (0, 0 = one value; 0, 1 = per column;
 * width, 1 = per element) or NULL (then the scalar).  u / z (rows, width; rows for the yaw case) replay given unit-interval
 * / standard-normal draws, NULL = Philox keyed by (seed, draw, row, column). */
enum { ILX_SAMPLE_UNIFORM = 0, ILX_SAMPLE_LOG_UNIFORM = 1, ILX_SAMPLE_TRIANGLE = 2, ILX_SAMPLE_GAUSSIAN = 3,
       ILX_SAMPLE_RANDOM_ORIENTATION = 4, ILX_SAMPLE_RANDOM_YAW_ORIENTATION = 5 };
int ilx_sample(int kind, const float* a, int64_t a_row_stride, int64_t a_col_stride, float a_scalar, const float* b,
               int64_t b_row_stride, int64_t b_col_stride, float b_scalar, const float* u, const float* z, uint64_t seed,
               uint64_t draw, float* out, int64_t rows, int width, void* stream);

/* ActionManager.process_action (managers/action_manager.py:318-339) for one JointAction term
 * (envs/mdp/actions/joint_actions.py:130-139): prev_action <- action, action <- raw_action,
 * processed = raw * scale + offset, clamped to clip (n,a,2) when given.  scale / offset are (n,a) tensors or
 * NULL (then the scalar is used).  action / prev_action may be NULL (term-level call without the manager). */
int ilx_joint_action_process(const float* raw_action, float* action, float* prev_action, const float* scale,
                             float scale_scalar, const float* offset, float offset_scalar, const float* clip,
                             float* processed, int64_t n, int a, void* stream);
/* The same launch, which also opens the env step for the fused managers (what ilx_mdp_begin_step of isaaclab_amd_mdp.h
 * does as a launch of its own): clears `log_sums` (`log_count` floats, or NULL) and adds one to `step_index_device` (or
 * NULL).  process_action is the first call of ManagerBasedRLEnv.step (envs/manager_based_rl_env.py:172), so a step
 * recorded in a HIP graph is one node shorter.  n > 0. */
int ilx_joint_action_process_begin(const float* raw_action, float* action, float* prev_action, const float* scale,
                                   float scale_scalar, const float* offset, float offset_scalar, const float* clip,
                                   float* processed, int64_t n, int a, float* log_sums, int32_t log_count,
                                   uint64_t* step_index_device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Actuators — source/isaaclab/isaaclab/actuators/
 * Per-joint parameters are (j,) when params_per_env == 0, else (n, j) like the reference's buffers.
 * ------------------------------------------------------------------------------------------------ */
#define ILX_ACTUATOR_IDEAL_PD 0 /* IdealPDActuator.compute / ImplicitActuator torque estimate: actuator_pd.py:62-87,131-145 */
#define ILX_ACTUATOR_DC_MOTOR 1 /* DCMotor.compute + _clip_effort: actuator_pd.py:211-233 */

/* computed = kp*(q_des-q) + kd*(qd_des-qd) + tau_ff ; applied = clip (symmetric, or the DC-motor envelope).
 * qd_des / tau_ff may be NULL (treated as zeros).  All state tensors are (n, j). */
int ilx_actuator_pd(int model, const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                    const float* qd, const float* stiffness, const float* damping, const float* effort_limit,
                    const float* velocity_limit, int params_per_env, float saturation_effort, float* computed,
                    float* applied, int64_t n, int j, void* stream);

/* DelayedPDActuator.compute (actuator_pd.py:283-291) and, with a lookup table, RemotizedPDActuator.compute
 * (:347-363), in one launch.  The three set-points are appended to their DelayBuffer rings (delay_buffer.py:158-177
 * over circular_buffer.py:121-175: rings of shape (ring_length, n, j) with ONE write pointer for all envs; `pointer` is
 * the slot this call writes, i.e. (previous + 1) % ring_length, starting at 0; the first push after a reset fills every
 * slot of the env) and read back min(time_lags[env], pushes since reset) steps late; then the ideal PD law and the
 * effort clip; then, if lookup_angles is given, the clamp to the torque limit interpolated at the joint position
 * (utils/interpolation/linear_interpolation.py:55-88, ascending angles, held constant outside).  num_pushes (n) int64
 * is advanced by one; time_lags (n) int32 is read.  qd_des / tau_ff may be NULL together with their rings (zeros).
 * A reset (actuator_pd.py:262-281) is the caller's: num_pushes <- 0, new time_lags, ring rows <- 0. */
int ilx_actuator_delayed_pd(const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                            const float* qd, const float* stiffness, const float* damping,
                            const float* effort_limit, int params_per_env, float* positions_ring,
                            float* velocities_ring, float* efforts_ring, int ring_length, int pointer,
                            int64_t* num_pushes, const int32_t* time_lags, const float* lookup_angles,
                            const float* lookup_max_torque, int lookup_size, float* computed, float* applied,
                            int64_t n, int j, void* stream);

#define ILX_ACT_SOFTSIGN 0
#define ILX_ACT_ELU 1
#define ILX_ACT_TANH 2
#define ILX_ACT_RELU 3
#define ILX_ACT_IDENTITY 4

/* Actuator network: Linear(dims[0],dims[1]) act ... Linear(dims[L-1],dims[L]=1); stands in for the
 * TorchScript module the reference loads at actuator_net.py:128-129.  weights_host[l] is (dims[l+1], dims[l])
 * row-major (torch.nn.Linear layout), biases_host[l] is (dims[l+1],); HOST pointers, copied to the current
 * device.  Widths <= 64, input width <= 16. */
typedef struct ilx_mlp_s* ilx_mlp_t;
int ilx_mlp_create(int num_layers, const int32_t* dims, int activation, const float* const* weights_host,
                   const float* const* biases_host, ilx_mlp_t* out);
void ilx_mlp_destroy(ilx_mlp_t net);

#define ILX_MLP_FORCE_GENERIC 1 /* flags: skip the MFMA path (testing) */
#define ILX_MLP_EXACT_F32 2     /* flags: every product in f32 (an f32 fmaf chain per unit).  Without it the hidden layers of
                                 * a 32-wide net with a bounded activation (softsign, tanh) and hidden weights below 6e4 run
                                 * on the f16 matrix cores with both factors split into an f16 head and an f16 remainder:
                                 * relative error of a product ~1e-6 instead of 6e-8, outputs within 1e-5 |y| + 2e-5 of the
                                 * float64 evaluation.  Networks with ELU / ReLU / identity activations (unbounded hidden
                                 * values) always take the f32 kernels. */

/* ActuatorNetMLP.compute — actuator_net.py:147-188.  In one launch: roll pos_error_history / velocity_history
 * (n, history_length, j; slot 0 newest) and write q_des - q / qd into slot 0, gather input_idx, scale,
 * concatenate (input_order 0 = "pos_vel", 1 = "vel_pos"), run the network per (env, joint) row,
 * computed = out * torque_scale, applied = DC-motor clip. */
int ilx_actuator_net_mlp(ilx_mlp_t net, const float* q_des, const float* q, const float* qd,
                         float* pos_error_history, float* velocity_history, int history_length,
                         const int32_t* input_idx_host, int num_input_idx, int input_order, float pos_scale,
                         float vel_scale, float torque_scale, const float* effort_limit,
                         const float* velocity_limit, int params_per_env, float saturation_effort,
                         float* computed, float* applied, int64_t n, int j, int flags, void* stream);

/* Recurrent actuator network: what ActuatorNetLSTM drives (actuator_net.py:48-54,86-89), i.e. a
 * torch.nn.LSTM(input_size, hidden_size, num_layers, batch_first=True) read from `network.lstm`, followed by the rest
 * of the TorchScript module, here a Linear stack ("head") from hidden_size to 1 with one activation between layers.
 * Per LSTM layer l, HOST pointers in torch's layout and gate order (input, forget, cell, output): weight_ih (4H, in_l),
 * weight_hh (4H, H), bias_ih (4H), bias_hh (4H), with in_0 = input_size and in_l = H above.  Head: head_dims[0] = H,
 * head_dims[head_layers] = 1, weights (dims[l+1], dims[l]) as torch.nn.Linear.  num_layers <= 4, hidden_size <= 16,
 * input_size 2..4, head widths <= 32. */
typedef struct ilx_lstm_s* ilx_lstm_t;
int ilx_lstm_create(int num_layers, int input_size, int hidden_size, const float* const* weight_ih_host,
                    const float* const* weight_hh_host, const float* const* bias_ih_host,
                    const float* const* bias_hh_host, int head_layers, const int32_t* head_dims, int head_activation,
                    const float* const* head_weights_host, const float* const* head_biases_host, ilx_lstm_t* out);
void ilx_lstm_destroy(ilx_lstm_t net);

/* ActuatorNetLSTM.compute — actuator_net.py:76-99.  In one launch, per (env, joint) row: network input
 * (q_des - q, qd), one time step of the LSTM on hidden_state / cell_state (num_layers, n*j, hidden_size; updated in
 * place, 16-byte aligned), the head on the last layer's output -> computed, applied = DC-motor clip
 * (actuator_pd.py:223-233).  Resetting envs (actuator_net.py:70-74) is a plain zero fill of their state rows by the
 * caller. */
int ilx_actuator_net_lstm(ilx_lstm_t net, const float* q_des, const float* q, const float* qd, float* hidden_state,
                          float* cell_state, const float* effort_limit, const float* velocity_limit,
                          int params_per_env, float saturation_effort, float* computed, float* applied, int64_t n,
                          int j, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reset / push event terms — source/isaaclab/isaaclab/envs/mdp/events.py:610-866.  The reference passes the new state to
 * the simulator (write_root_pose_to_sim, write_root_velocity_to_sim, write_joint_state_to_sim); these entry points write
 * it into the state buffers.  env_ids (count) int64 lists the envs, as the reference's env_ids tensor does.  Random
 * numbers: `u` non-NULL = one row of unit-interval draws per listed env, in the order the reference draws them
 * (torch.rand through math.sample_uniform, math.py:1313-1331); `u` NULL = Philox keyed by (seed, draw, env id, column).
 * Ranges are HOST arrays of 12 floats: lower bounds of x, y, z, roll, pitch, yaw, then the upper bounds.
 * ------------------------------------------------------------------------------------------------ */

/* reset_root_state_uniform (:638-680): position = default + env origin + sample, orientation = default * quat_from_euler
 * of the sampled angles, velocity = default + sample; rows env_ids of root_state_w (n,13).  u: (count, 12) = pose draws
 * then velocity draws.  env_origins (n,3) may be NULL. */
int ilx_event_reset_root_state_uniform(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                       const float* env_origins, const float* pose_range_host,
                                       const float* velocity_range_host, const float* u, uint64_t seed, uint64_t draw,
                                       float* root_state_w, void* stream);
/* reset_root_state_from_terrain (:735-799): position = valid_positions[level[env], type[env], pick] + default root position
 * (valid_positions (rows, cols, patches, 3) = terrain.flat_patches["init_pos"]), orientation = quat_from_euler_xyz of the
 * sampled roll, pitch, yaw, velocity = default + sample.  patch_ids (count): the picks (torch.randint in the reference) or
 * NULL: Philox.  u: (count, 9) = three angle draws then six velocity draws.  The reference adds the velocity samples to
 * default_root_state[:, 7:13] without indexing it by env_ids, so it only runs when every env is listed and row k of the
 * list takes the default velocity of env k; the same is done here. */
int ilx_event_reset_root_state_from_terrain(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                            const float* valid_positions, const int64_t* terrain_levels, const int64_t* terrain_types,
                                            int32_t num_terrain_cols, int32_t num_patches, const int64_t* patch_ids,
                                            const float* pose_range_host, const float* velocity_range_host, const float* u,
                                            uint64_t seed, uint64_t draw, float* root_state_w, void* stream);
/* reset_joints_by_scale (:802-832, by_scale = 1) / reset_joints_by_offset (:835-866, by_scale = 0): default joint state
 * times / plus a sample, clamped to the soft position limits (n,j,2) and +-soft velocity limits (n,j); rows env_ids of
 * joint_pos / joint_vel (n,j).  u: (count, 2j) = position draws then velocity draws. */
int ilx_event_reset_joints(int by_scale, const int64_t* env_ids, int64_t count, const float* default_joint_pos,
                           const float* default_joint_vel, const float* soft_joint_pos_limits,
                           const float* soft_joint_vel_limits, float position_lo, float position_hi, float velocity_lo,
                           float velocity_hi, const float* u, uint64_t seed, uint64_t draw, float* joint_pos,
                           float* joint_vel, int j, void* stream);
/* push_by_setting_velocity (:610-635): root velocity (columns 7..12 of root_state_w) += sample.  u: (count, 6). */
int ilx_event_push_by_setting_velocity(const int64_t* env_ids, int64_t count, const float* velocity_range_host,
                                       const float* u, uint64_t seed, uint64_t draw, float* root_state_w, void* stream);

/* reset_root_state_with_random_orientation (:683-732): position = default + env origin + sample (x, y, z entries of
 * pose_range_host, laid out as for ilx_event_reset_root_state_uniform), orientation uniform on SO(3) = a normalised
 * 4-vector of standard normal draws (utils/math.py:1251-1267), velocity = default + sample.  u (count, 9) = 3 position
 * then 6 velocity draws and normal_draws (count, 4) are injected together, or both NULL (Philox, Box-Muller). */
int ilx_event_reset_root_state_random_orientation(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                                  const float* env_origins, const float* pose_range_host,
                                                  const float* velocity_range_host, const float* u,
                                                  const float* normal_draws, uint64_t seed, uint64_t draw,
                                                  float* root_state_w, void* stream);
/* apply_external_force_torque (:579-607) with Articulation.set_external_force_and_torque
 * (assets/articulation/articulation.py:778-805): uniform forces / torques for the listed envs and bodies (body_ids device
 * int32 (num_selected_bodies) or NULL = the first num_selected_bodies bodies) written into external_force_b /
 * external_torque_b (n,num_bodies,3).  u (2, count, num_selected_bodies, 3): force draws, then torque draws. */
int ilx_event_apply_external_force_torque(const int64_t* env_ids, int64_t count, const int32_t* body_ids,
                                          int num_selected_bodies, int num_bodies, float force_lo, float force_hi,
                                          float torque_lo, float torque_hi, const float* u, uint64_t seed, uint64_t draw,
                                          float* external_force_b, float* external_torque_b, void* stream);
/* reset_scene_to_default (:911-933) for one articulation, rows env_ids: root_state_w <- default_root_state with the env
 * origin added to the position, joint_pos / joint_vel <- the default joint state (no sampling, no clamping).  Either part
 * may be left out (root_state_w NULL, or joint_pos NULL). */
int ilx_event_reset_scene_to_default(const int64_t* env_ids, int64_t count, const float* default_root_state,
                                     const float* env_origins, float* root_state_w, const float* default_joint_pos,
                                     const float* default_joint_vel, float* joint_pos, float* joint_vel, int j,
                                     void* stream);

/* Terrain curriculum of the velocity tasks: terrain_levels_vel
 * (source/isaaclab_tasks/isaaclab_tasks/manager_based/locomotion/velocity/mdp/curriculums.py:26-55) fused with
 * TerrainImporter.update_env_origins (source/isaaclab/isaaclab/terrains/terrain_importer.py:307-322), for the listed envs:
 *   distance  = |root_pos_w.xy - env_origins.xy|
 *   move_up   = distance > half_terrain_size                     (terrain_generator.size[0] / 2)
 *   move_down = distance < |command.xy| * max_episode_length_s * 0.5, and not move_up
 *   level += move_up - move_down; level >= num_rows -> a random level in [0, num_rows); level < 0 -> 0
 *   env_origins[env] = terrain_origins[level, terrain_types[env]]
 * move_up / move_down (count bytes each, both or neither): given flags replace the first three lines, which makes the
 * call TerrainImporter.update_env_origins alone.  random_levels (count) holds the replacement levels
 * (torch.randint_like draws one per listed env) or is NULL: Philox keyed by (seed, draw, env).  terrain_levels /
 * terrain_types are int64 (n); terrain_origins (num_rows, num_cols, 3); env_origins (n,3). */
int ilx_curriculum_terrain_levels(const int64_t* env_ids, int64_t count, const float* root_state_w, const float* command,
                                  int command_dim, float half_terrain_size, float max_episode_length_s,
                                  const uint8_t* move_up, const uint8_t* move_down, int64_t* terrain_levels,
                                  const int64_t* terrain_types, const float* terrain_origins, int num_rows, int num_cols,
                                  const int64_t* random_levels, uint64_t seed, uint64_t draw, float* env_origins,
                                  void* stream);

/* JointPositionToLimitsAction.process_actions (envs/mdp/actions/joint_actions_to_limits.py:107-129):
 * processed = raw * scale [clamped to clip (n,a,2)]; with rescale_to_limits, clamped to [-1, 1] and mapped onto the soft
 * joint position limits (n,num_joints,2) of the term's joints (joint_ids (a) device int32, NULL = the first a joints).
 * With prev_applied_actions (n,a) the call is EMAJointPositionToLimitsAction.process_actions (:213-229):
 * processed = clamp(alpha * processed + (1 - alpha) * prev_applied, limits), prev_applied <- processed
 * (alpha (n,a) tensor or NULL + scalar). */
int ilx_joint_action_to_limits(const float* raw_action, const float* scale, float scale_scalar, const float* clip,
                               int rescale_to_limits, const float* soft_joint_pos_limits, const int32_t* joint_ids,
                               int num_joints, const float* alpha, float alpha_scalar, float* prev_applied_actions,
                               float* processed, int64_t n, int a, void* stream);
/* BinaryJointAction.process_actions (envs/mdp/actions/binary_joint_actions.py:118-131): one action per env, float
 * (n,) (negative closes) or boolean bytes (n,) (false closes); processed (n,j) = close_command (j) or open_command (j),
 * clamped to clip (n,j,2) when given. */
int ilx_binary_joint_action(const float* raw_action, const uint8_t* raw_action_bool, const float* open_command,
                            const float* close_command, const float* clip, float* processed, int64_t n, int j,
                            void* stream);
/* NonHolonomicAction.apply_actions (envs/mdp/actions/non_holonomic_actions.py:147-155): yaw of the body from its
 * quaternion (strided rows, w,x,y,z; euler_xyz_from_quat), joint_vel_command (n,3) = (cos yaw * v, sin yaw * v, yaw rate)
 * from processed (n,2) = (v, yaw rate). */
int ilx_non_holonomic_action_apply(const float* body_quat_w, int64_t body_quat_stride, const float* processed,
                                   float* joint_vel_command, int64_t n, void* stream);

/* UniformPose2dCommand / TerrainBasedPose2dCommand (envs/mdp/commands/pose_2d_command.py) under CommandTerm
 * (managers/command_manager.py:120-187).
 *   compute(dt): metrics error_pos_2d = |pos_command_w.xy - root_pos.xy|, error_heading = |wrap(heading_command_w - heading_w)|
 *                (:88-91); time_left -= dt; envs at <= 0 resample: time_left, command_counter += 1, a new target = env origin
 *                + uniform (x, y) (:93-101), or one of the terrain's flat patches valid_targets[level, type, pick] (:177-185),
 *                z += default root z; heading towards / away from the target (simple_heading, :103-119) or uniform; then
 *                pos_command_b = quat_rotate_inverse(yaw_quat(root_quat), pos_command_w - root_pos),
 *                heading_command_b = wrap(heading_command_w - heading_w) (:124-128).
 *   reset(mask): for flagged envs the three metrics are added to log_sums[0..2] (+1 to log_sums[3]) and zeroed, the
 *                counter is zeroed, and the command is resampled.
 * Draws: samples / reset_samples (n,4) hold the VALUES to use (time_left, x offset, y offset, heading) and patch_ids /
 * reset_patch_ids (n) the patch picks, or NULL: Philox keyed by (seed, draw, env). */
typedef struct ilx_pose2d_command {
    int64_t num_envs;
    const float* root_state_w;         /* (n,13) */
    const float* default_root_state;   /* (n,13) */
    const float* env_origins;          /* (n,3), uniform variant */
    const float* valid_targets;        /* (rows, cols, patches, 3) or NULL: terrain-based variant */
    const int64_t* terrain_levels;     /* (n) */
    const int64_t* terrain_types;      /* (n) */
    int32_t num_terrain_cols, num_patches;
    int32_t simple_heading;
    float resampling_time_range[2], pos_x[2], pos_y[2], heading[2];
    float* pos_command_w;              /* (n,3) */
    float* heading_command_w;          /* (n) */
    float* pos_command_b;              /* (n,3) */
    float* heading_command_b;          /* (n) */
    float* time_left;                  /* (n) */
    int64_t* command_counter;          /* (n) */
    float* error_pos;                  /* (n) metrics["error_pos"]: never updated by the reference, only logged and zeroed */
    float* error_heading;              /* (n) */
    float* error_pos_2d;               /* (n) */
    const float* samples;              /* (n,4) or NULL */
    const float* reset_samples;
    const int64_t* patch_ids;          /* (n) or NULL */
    const int64_t* reset_patch_ids;
    uint64_t seed, draw;
} ilx_pose2d_command;
int ilx_pose2d_command_compute(const ilx_pose2d_command* cmd, float dt, void* stream);
int ilx_pose2d_command_reset(const ilx_pose2d_command* cmd, const uint8_t* reset_mask, float* log_sums, void* stream);

/* NormalVelocityCommand (envs/mdp/commands/velocity_command.py:218-287) under CommandTerm (managers/command_manager.py:120-187).
 *   compute(dt): metrics error_vel_xy += |cmd.xy - root_lin_vel_b.xy| / max_command_step, error_vel_yaw += |cmd.z -
 *                root_ang_vel_b.z| / max_command_step (:111-121; max_command_step = resampling_time_range[1] / step_dt);
 *                time_left -= dt; envs at <= 0 resample (:252-272): time_left, command_counter += 1, each component ~
 *                N(mean, std) times a +-1 coin (u <= 0.5 keeps the sign), per-component zero flags (u <= zero_prob) and the
 *                standing flag (u <= rel_standing_envs); then standing envs and flagged components are zeroed (:274-287).
 *   reset(mask): for flagged envs the two metrics are added to log_sums[0..1] (+1 to log_sums[2]) and zeroed, the counter
 *                is zeroed, and the command is resampled (no zeroing: that happens at the next compute).
 * Draws: samples / reset_samples (n,11) hold, per env, time_left, the three normal VALUES (mean and std applied), the three
 * sign coins, the three zero coins and the standing coin, in the order the reference draws them; NULL: Philox keyed by
 * (seed, draw, env) with Box-Muller normals. */
typedef struct ilx_normal_velocity_command {
    int64_t num_envs;
    const float* root_state_w;   /* (n,13) */
    float mean_vel[3], std_vel[3], zero_prob[3];
    float rel_standing_envs, max_command_step;
    float resampling_time_range[2];
    float* vel_command_b;        /* (n,3) */
    uint8_t* is_zero_vel_env;    /* (n,3): x, y, yaw */
    uint8_t* is_standing_env;    /* (n) */
    float* time_left;            /* (n) */
    int64_t* command_counter;    /* (n) */
    float* error_vel_xy;         /* (n) */
    float* error_vel_yaw;        /* (n) */
    const float* samples;        /* (n,11) or NULL */
    const float* reset_samples;
    uint64_t seed, draw;
} ilx_normal_velocity_command;
int ilx_normal_velocity_command_compute(const ilx_normal_velocity_command* cmd, float dt, void* stream);
int ilx_normal_velocity_command_reset(const ilx_normal_velocity_command* cmd, const uint8_t* reset_mask, float* log_sums, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Observation modifiers — source/isaaclab/isaaclab/utils/modifiers/modifier.py, applied per term by
 * managers/observation_manager.py:308-313 before noise, clip and scale.
 * ------------------------------------------------------------------------------------------------ */
#define ILX_MOD_COPY 0            /* out = in (strided copy of a column block) */
#define ILX_MOD_SCALE 1           /* modifier.py:22-32   out = in * p0 */
#define ILX_MOD_CLIP 2            /* modifier.py:35-46   out = clip(in, p0 if has_lo, p1 if has_hi) */
#define ILX_MOD_BIAS 3            /* modifier.py:49-59   out = in + p0 */
#define ILX_MOD_DIGITAL_FILTER 4  /* modifier.py:67-191  y = sum_j B[j] x[i-j] - sum_j A[j] y[i-1-j]; histories shift by one */
#define ILX_MOD_INTEGRATOR 5      /* modifier.py:194-259 integral += (in + previous) / 2 * p0; previous = in; out = integral */
#define ILX_MOD_MAX_TAPS 16
typedef struct ilx_modifier {
    const float* in;       /* (n, d), rows in_stride floats apart */
    float* out;            /* (n, d), rows out_stride floats apart; may alias in */
    int64_t in_stride, out_stride, n;
    int32_t d;
    float p0, p1;          /* multiplier / bias / lower bound / integrator dt; upper bound */
    int32_t has_lo, has_hi;
    int32_t na, nb;        /* digital filter: number of A and B coefficients */
    float A[ILX_MOD_MAX_TAPS], B[ILX_MOD_MAX_TAPS];
    float* x_n;            /* digital filter: (n, d, nb) input history, slot 0 newest; integrator: (n, d) integral */
    float* y_n;            /* digital filter: (n, d, na) output history, slot 0 newest; integrator: (n, d) previous input */
} ilx_modifier;
int ilx_modifier_apply(int32_t kind, const ilx_modifier* mod, void* stream);
/* DigitalFilter.reset / Integrator.reset (modifier.py:157-168, 232-243): zero the rows (len_a / len_b floats per env) of
 * the two state buffers for the listed envs (env_ids NULL: the first count envs). */
int ilx_modifier_reset(float* state_a, int64_t len_a, float* state_b, int64_t len_b, const int64_t* env_ids, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-patch search — source/isaaclab/isaaclab/terrains/utils.py:135-273 (find_flat_patches); one round of its rejection
 * sampling is ilx_flat_patch_rays, ilx_raycast_mesh (z-parallel rays), ilx_flat_patch_check.
 * ------------------------------------------------------------------------------------------------ */

/* utils.py:236-244: for the count candidate patches (rows patch_ids of flat_patches (num_patches,3)) write the sampled
 * centre (x, y) and build the rays: starts (count, ring_points, 3) = centre + ring (ring_points,3) at z = 100, directions -z. */
int ilx_flat_patch_rays(const float* centre_x, const float* centre_y, const float* ring, const int64_t* patch_ids, int64_t count,
                        int32_t ring_points, float* flat_patches, float* ray_starts, float* ray_directions, void* stream);
/* utils.py:248-258: patch height = height of the last ring point; not_valid[k] = any height outside [z_lo, z_hi] or
 * max - min > max_height_diff. */
int ilx_flat_patch_check(const float* ray_hits, const int64_t* patch_ids, int64_t count, int32_t ring_points, float z_lo, float z_hi,
                         float max_height_diff, float* flat_patches, uint8_t* not_valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Homogeneous poses and camera geometry — source/isaaclab/isaaclab/utils/math.py
 * ------------------------------------------------------------------------------------------------ */
#define ILX_POSE_MAKE 0         /* math.py:1564-1583  a (n,3) positions, b (n,3,3) rotations -> out (n,4,4) */
#define ILX_POSE_INV 1          /* math.py:1601-1626  a (n,4,4) -> out (n,4,4) = [R^T, -R^T t; 0 1] */
#define ILX_POSE_MUL 2          /* math.py:1629-1643  out (n,4,4) = a @ b (pose_in_A_to_pose_in_B(pose_in_A = b, pose_A_in_B = a)) */
#define ILX_POSE_COPYSIGN 3     /* math.py:121-135    out (n) = |scalar| * sign(a) */
#define ILX_POSE_IS_IDENTITY 4  /* math.py:728-746    a (n,3), b (n,4): *flag = 1 when any pose is not allclose to identity */
/* a_batch / b_batch: n, or 1 when the operand is one matrix shared by all n. */
int ilx_pose_op(int32_t op, const float* a, int64_t a_batch, const float* b, int64_t b_batch, float scalar, float* out, int32_t* flag,
                int64_t n, void* stream);

#define ILX_CAMERA_ORTHOGONALIZE 0  /* math.py:991-1079   depth (n,h,w) -> out (n,h,w) = depth / sqrt(1 + ((u-cx)/fx)^2 + ((v-cy)/fy)^2) */
#define ILX_CAMERA_UNPROJECT 1      /* math.py:1083-1166  depth (n,h,w) -> out (n, w*h, 3), point u*h + v = K^-1 [u v 1]^T / last * depth */
#define ILX_CAMERA_PROJECT 2        /* math.py:1170-1225  points (n,h,3) (h points per batch, w unused) -> out (n,h,3) = (x/z, y/z of K p, Z) */
/* intrinsics (intrinsics_batch,3,3) with intrinsics_batch = n, or 1 for one calibration matrix shared by the batch. */
int ilx_camera_op(int32_t op, const float* data, const float* intrinsics, int64_t intrinsics_batch, float* out, int64_t n, int32_t h,
                  int32_t w, void* stream);

/* sample_cylinder (math.py:1387-1420): out (n,3) = (radius cos a, radius sin a, z), a = (u0*2-1)*pi, z = u1*(h_max-h_min)+h_min.
 * u (2,n): the unit-interval draws (angles first, then heights), or NULL: Philox keyed by (seed, draw, point). */
int ilx_sample_cylinder(float radius, float h_min, float h_max, const float* u, uint64_t seed, uint64_t draw, float* out, int64_t n,
                        void* stream);

/* LinearInterpolation.compute (utils/interpolation/linear_interpolation.py:55-88): piecewise-linear function through the
 * samples (x ascending, num_samples of them), held constant outside them; out[i] = f(q[i]). */
int ilx_linear_interpolation(const float* x, const float* y, int32_t num_samples, const float* q, float* out, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batched ring buffers — source/isaaclab/isaaclab/utils/buffers/circular_buffer.py, delay_buffer.py (observation
 * histories, observation_manager.py:319-327; delayed set-points).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ilx_circular_buffer {
    float* buffer;         /* (max_len, n, d) */
    int64_t* num_pushes;   /* (n) pushes since the env's last reset */
    int32_t max_len;
    int32_t pointer;       /* the slot of the newest entry; for an append the caller has already advanced it (:126) */
    int64_t n;
    int32_t d;
} ilx_circular_buffer;
/* CircularBuffer.append (circular_buffer.py:107-135) when data != NULL: write the slot `pointer`; envs with no push since
 * their reset get every slot filled; num_pushes += 1.  Then, from the state after the append:
 *   ordered_out (n, max_len*d) = CircularBuffer.buffer flattened (:78-86), oldest entry first;
 *   delayed_out (n, d) = CircularBuffer[time_lags] (:137-166; DelayBuffer.compute, delay_buffer.py:158-177): the entry
 *   min(time_lags[env], num_pushes[env] - 1) pushes back (time_lags NULL: the newest).
 * Either output may be NULL.  data, ordered_out and delayed_out are rows with the given strides (in floats). */
int ilx_circular_buffer_append(const ilx_circular_buffer* ring, const float* data, int64_t data_stride, float* ordered_out,
                               int64_t ordered_stride, const int32_t* time_lags, float* delayed_out, int64_t delayed_stride,
                               void* stream);
/* CircularBuffer.reset (circular_buffer.py:92-105): num_pushes = 0 and zeroed slots for the listed envs (NULL: the first count). */
int ilx_circular_buffer_reset(const ilx_circular_buffer* ring, const int64_t* env_ids, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Noise — source/isaaclab/isaaclab/utils/noise/noise_model.py:20-98 (constant_noise, uniform_noise, gaussian_noise) and the
 * per-env bias of NoiseModelWithAdditiveBias (:145-182).
 * ------------------------------------------------------------------------------------------------ */
#define ILX_NOISE_CONSTANT 0   /* noise = a (the bias) */
#define ILX_NOISE_UNIFORM 1    /* noise = U[0,1) * (b - a) + a      (a = n_min, b = n_max) */
#define ILX_NOISE_GAUSSIAN 2   /* noise = a + b * N(0,1)            (a = mean, b = std) */
#define ILX_NOISE_ADD 0        /* out = data + noise */
#define ILX_NOISE_SCALE 1      /* out = data * noise */
#define ILX_NOISE_ABS 2        /* out = noise */
typedef struct ilx_noise {
    const float* data;        /* (rows, d), rows data_stride floats apart */
    float* out;               /* (rows, d), rows out_stride floats apart; may alias data */
    int64_t data_stride, out_stride, n;
    int32_t d;
    const int64_t* row_ids;   /* (n) rows of data / out to work on, or NULL: rows 0..n-1 */
    float a, b, range;        /* scalar parameters; range = n_max - n_min as the caller's arithmetic rounds it */
    const float* a_cols;      /* (d) per-column parameters or NULL (then the scalars are used) */
    const float* b_cols;
    const float* draws;       /* (n, d) dense U[0,1) / N(0,1) draws to use, or NULL: Philox keyed by (seed, draw, row, column) */
    const float* row_bias;    /* (rows) added after the noise (NoiseModelWithAdditiveBias.apply), or NULL */
    uint64_t seed, draw;
} ilx_noise;
int ilx_noise_apply(int32_t kind, int32_t operation, const ilx_noise* noise, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IMU sensor — source/isaaclab/isaaclab/sensors/imu/imu.py:142-182 with the update timing of
 * sensors/sensor_base.py:197-205, 284-293.
 * ------------------------------------------------------------------------------------------------ */

/* advance_clock = 1: SensorBase.update(dt_clock) (timestamp += dt_clock; flag sensors whose period elapsed) followed by the
 * refresh of the flagged sensors; advance_clock = 0: only the refresh of sensors already flagged.  A refresh computes
 *   pos_w = p + quat_rotate(q, offset_pos), quat_w = q * offset_quat                      (q: the view's x,y,z,w as w,x,y,z)
 *   lin_vel_w = v + w x quat_rotate(q, offset_pos - com_pos_b), ang_vel_w = w
 *   lin_acc_w = (lin_vel_w - prev_lin_vel_w) / dt + gravity_bias, ang_acc_w = (ang_vel_w - prev_ang_vel_w) / dt
 *   the four body-frame outputs = quat_rotate_inverse(quat_w, .), then prev_* = the world velocities. */
typedef struct ilx_imu {
    int64_t num_sensors;
    const float* transforms;   /* (n,7) RigidBodyView.get_transforms(): position, quaternion x,y,z,w */
    const float* velocities;   /* (n,6) RigidBodyView.get_velocities(): linear, angular, of the centre of mass */
    const float* coms;         /* (n,7) RigidBodyView.get_coms(): the first three are the centre of mass in the link frame */
    float offset_pos[3], offset_quat[4], gravity_bias[3];   /* ImuCfg.offset.pos / .rot (w,x,y,z), ImuCfg.gravity_bias */
    float update_period;
    float* pos_w;              /* (n,3) */
    float* quat_w;             /* (n,4) w,x,y,z */
    float* lin_vel_b;          /* (n,3) */
    float* ang_vel_b;
    float* lin_acc_b;
    float* ang_acc_b;
    float* prev_lin_vel_w;     /* (n,3) state of the finite difference */
    float* prev_ang_vel_w;
    float* timestamp;          /* (n) SensorBase clocks */
    float* timestamp_last_update;
    uint8_t* is_outdated;      /* (n) */
} ilx_imu;
int ilx_imu_update(const ilx_imu* imu, int32_t advance_clock, float dt_clock, float dt, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FrameTransformer sensor — source/isaaclab/isaaclab/sensors/frame_transformer/frame_transformer.py:322-385
 * ------------------------------------------------------------------------------------------------ */

/* One refresh of all envs: source pose = view row source_rows[env] (+ offset, combine_frame_transforms), target pose of
 * frame f = view row target_rows[env * F + f] (+ offset), target in source = subtract_frame_transforms(source, target).
 * The row indices are the reference's chain of gathers composed once:
 *   source_rows[env]          = _per_env_indices[_source_frame_body_ids[env]]
 *   target_rows[env * F + f]  = _per_env_indices[_target_frame_body_ids[_duplicate_frame_indices[env * F + f]]]
 * Offsets are NULL when the reference skips them (_apply_source_frame_offset / _apply_target_frame_offset False). */
typedef struct ilx_frame_transformer {
    int64_t num_envs;
    int32_t num_frames;                /* F: target frames per env */
    const float* transforms;           /* (rows,7) RigidBodyView.get_transforms(): position, quaternion x,y,z,w */
    const int64_t* source_rows;        /* (n) */
    const int64_t* target_rows;        /* (n*F) */
    const float* source_offset_pos;    /* (n,3) or NULL */
    const float* source_offset_quat;   /* (n,4) w,x,y,z */
    const float* target_offset_pos;    /* (n*F,3) or NULL */
    const float* target_offset_quat;   /* (n*F,4) */
    float* source_pos_w;               /* (n,3) */
    float* source_quat_w;              /* (n,4) */
    float* target_pos_w;               /* (n,F,3) */
    float* target_quat_w;              /* (n,F,4) */
    float* target_pos_source;          /* (n,F,3) */
    float* target_quat_source;         /* (n,F,4) */
} ilx_frame_transformer;
int ilx_frame_transformer_update(const ilx_frame_transformer* sensor, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Contact sensor bookkeeping — source/isaaclab/isaaclab/sensors/contact_sensor/contact_sensor.py:314-373
 * plus the update timing of sensors/sensor_base.py:197-205,284-293.
 * ------------------------------------------------------------------------------------------------ */

/* advance_clock = 1: one SensorBase.update(dt) (timestamp += dt; flag sensors whose period elapsed) followed by
 * the refresh of the flagged sensors; advance_clock = 0: only SensorBase._update_outdated_buffers (refresh the
 * sensors already flagged; dt is ignored).  A refresh copies net_forces_in (n,b,3) into net_forces_w, pushes it
 * into the history (n,h,b,3; NULL when h == 0) and advances the air / contact timers (n,b; all four NULL when air
 * time is not tracked).  timestamp, timestamp_last_update (n,) float and is_outdated (n,) uint8 are the SensorBase
 * clocks.
 * History layout: with history_head == NULL it is the reference's (contact_sensor.py:326-328): slot 0 is the newest
 * entry and a push moves every row one slot up.  With history_head (n,) int32 the same entries are kept as a ring per
 * env: the entry of age k sits at slot (history_head[env] + k) % h and a push writes one row and moves the head (the
 * terms that read the history take a maximum over it, rewards.py:266 / terminations.py:157, so they need no order;
 * a reader that does reorders by the head).  Zero-filled heads and history are a valid initial / reset state.
 * net_forces_w may be NULL when there is a history: its newest entry holds the same values at all times (both are
 * written by the same refresh and zeroed by the same reset), so a caller whose step reads only the history can leave the
 * copy out and hand out the newest entry as net_forces_w when somebody asks. */
int ilx_contact_update(const float* net_forces_in, float* net_forces_w, float* history, int32_t* history_head,
                       float* current_air_time,
                       float* last_air_time, float* current_contact_time, float* last_contact_time,
                       float* timestamp, float* timestamp_last_update, uint8_t* is_outdated, int advance_clock,
                       float dt, float update_period, float force_threshold, int64_t n, int b, int h, void* stream);

/* InteractiveScene.update(dt) in ONE launch (scene/interactive_scene.py:496-513) for a scene with up to one
 * articulation, one contact sensor and one ray caster:
 *   articulation : joint_acc = (joint_vel - previous_joint_vel) / joint_acc_elapsed; previous <- joint_vel
 *                  (articulation_data.py:77-82, 445-455)
 *   contact      : clock += dt, then the bookkeeping of ilx_contact_update for the envs that are due
 *                  (contact_lazy != 0: only flag them, as a sensor without history does under lazy_sensor_update)
 *   ray caster   : clock += dt and flag the envs that are due (sensor_base.py:197-205); the rays themselves are cast
 *                  by ilx_raycaster_update when the data is read.
 * A group whose first pointer (joint_vel / net_forces_in / ray_timestamp) is NULL is skipped. */
typedef struct ilx_scene {
    int64_t num_envs;
    float dt;
    /* articulation */
    const float* joint_vel;          /* (n,j) */
    float* previous_joint_vel;       /* (n,j) */
    float* joint_acc;                /* (n,j) */
    int32_t num_joints;
    float joint_acc_elapsed;
    /* contact sensor: same meaning as the arguments of ilx_contact_update */
    const float* net_forces_in;      /* (n,b,3) */
    float* net_forces_w;             /* (n,b,3), or NULL with a history */
    float* net_forces_w_history;     /* (n,h,b,3) or NULL */
    float* current_air_time;
    float* last_air_time;
    float* current_contact_time;
    float* last_contact_time;
    float* contact_timestamp;
    float* contact_timestamp_last_update;
    uint8_t* contact_is_outdated;
    float contact_update_period, force_threshold;
    int32_t num_bodies, history_length, contact_lazy;
    /* ray caster clock */
    float* ray_timestamp;
    float* ray_timestamp_last_update;
    uint8_t* ray_is_outdated;
    float ray_update_period;
    /* contact sensor, ring mode of the history (see ilx_contact_update), or NULL */
    int32_t* contact_history_head;
} ilx_scene;
int ilx_scene_update(const ilx_scene* scene, void* stream);
/* ilx_actuator_net_mlp (declared above) for physics substep k + 1 together with ilx_scene_update for substep k.  The two
 * have no physics step between them (manager_based_rl_env.py:180-195: scene.update(dt) ends one pass of the decimation
 * loop, action_manager.apply_action + write_data_to_sim open the next) and share no buffer: the scene update reads joint
 * velocities and contact forces and writes joint_acc, the contact buffers and the sensor clocks; the actuator model writes
 * torques and its histories.  For the three-sample 6-32-32-32-1-shaped networks served by the split-f16 kernel they are
 * workgroups of one grid (one launch and its fixed cost fewer per substep, the memory-bound update beside the
 * matrix-bound network); every other network is served by the two launches, the scene update first.  Results are those
 * of the two calls. */
/* ilx_actuator_net_mlp for the FIRST physics substep of an env step, which also does what the launch just before it
 * would do -- ilx_joint_action_process_begin (declared above): prev_action <- action <- raw_action,
 * q_des <- raw_action * action_scale + action_offset (joint_actions.py:130-160 with a scalar scale and an (n,j) offset,
 * e.g. the default joint positions), the statistics vector cleared and the step count advanced -- and runs the network on
 * that q_des.  Here q_des is an OUTPUT (the articulation's joint position target; the later substeps read it).  For the
 * three-sample networks of the split-f16 kernel this is one launch; every other network gets the two launches. */
int ilx_actuator_net_mlp_joint_action(ilx_mlp_t net, const float* raw_action, float* action, float* prev_action,
                                      float action_scale, const float* action_offset, float* q_des, const float* q,
                                      const float* qd, float* pos_error_history, float* velocity_history,
                                      int history_length, const int32_t* input_idx_host, int num_input_idx,
                                      int input_order, float pos_scale, float vel_scale, float torque_scale,
                                      const float* effort_limit, const float* velocity_limit, int params_per_env,
                                      float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                      int flags, float* log_sums, int32_t log_count, uint64_t* step_index_device,
                                      void* stream);
/* ... and for the PD models (ilx_actuator_pd, declared above), with a joint action term whose joints are consecutive
 * (JointPositionAction / JointEffortAction, joint_actions.py:130-160, 214-225; action_manager.py:331-332): prev_action <-
 * action <- raw_action (n, width); the target of joints first_joint .. first_joint + width - 1 <- raw_action *
 * action_scale + action_offset ((n, width) or NULL: zeros) -- `target` 0: their position target in q_des, 1: their effort
 * target in tau_ff, both OUTPUTS for those joints (the later substeps read them) --; the statistics vector cleared and the
 * step count advanced as by ilx_joint_action_process_begin; then the model on all j joints.  One launch.  n > 0. */
int ilx_actuator_pd_joint_action(int model, const float* raw_action, float* action, float* prev_action, float action_scale,
                                 const float* action_offset, int32_t target, int32_t first_joint, int32_t width, float* q_des,
                                 const float* qd_des, float* tau_ff, const float* q, const float* qd, const float* stiffness,
                                 const float* damping, const float* effort_limit, const float* velocity_limit,
                                 int params_per_env, float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                 float* log_sums, int32_t log_count, uint64_t* step_index_device, void* stream);
/* The same pairing for the PD models (ilx_actuator_pd, declared above): always one launch. */
int ilx_actuator_pd_scene_update(int model, const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                                 const float* qd, const float* stiffness, const float* damping, const float* effort_limit,
                                 const float* velocity_limit, int params_per_env, float saturation_effort, float* computed,
                                 float* applied, int64_t n, int j, const ilx_scene* scene, void* stream);
int ilx_actuator_net_mlp_scene_update(ilx_mlp_t net, const float* q_des, const float* q, const float* qd,
                                      float* pos_error_history, float* velocity_history, int history_length,
                                      const int32_t* input_idx_host, int num_input_idx, int input_order, float pos_scale,
                                      float vel_scale, float torque_scale, const float* effort_limit,
                                      const float* velocity_limit, int params_per_env, float saturation_effort,
                                      float* computed, float* applied, int64_t n, int j, int flags, const ilx_scene* scene,
                                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mesh ray casting — source/isaaclab/isaaclab/utils/warp/ops.py, utils/warp/kernels.py,
 * sensors/ray_caster/ray_caster.py
 * ------------------------------------------------------------------------------------------------ */

/* convert_to_warp_mesh (ops.py:130-145): builds the acceleration structure (4-wide BVH) for a triangle mesh
 * given as HOST arrays: vertices (num_verts, 3) float32 and triangles (num_tris, 3) int32.  The handle owns
 * device memory on the current device and is immutable. */
typedef struct ilx_mesh_s* ilx_mesh_t;
int ilx_mesh_create(const float* verts_host, int64_t num_verts, const int32_t* tris_host, int64_t num_tris,
                    ilx_mesh_t* out);
void ilx_mesh_destroy(ilx_mesh_t mesh);
int ilx_mesh_info(ilx_mesh_t mesh, int64_t* num_tris, int64_t* num_nodes, int32_t* depth);
/* Diagnostics of the column grid that serves ILX_RAYS_Z_PARALLEL casts (no counterpart in the reference, where the
 * structure is hidden inside wp.Mesh, ops.py:130-145).  out[0..11] = cells, empty cells, cells naming a run of 1 / 2 /
 * 3 or more neighbouring records, cells with a candidate list, distinct lists, 32-bit words of list storage, candidates
 * summed over all cells, cells per row, cells naming two / three unrelated records.  All zero when the mesh has no grid. */
int ilx_mesh_grid_stats(ilx_mesh_t mesh, int64_t out[12]);
/* Diagnostics of the height-field patches found in the mesh (no counterpart in the reference).  A patch is a run of
 * faces in the vertex / face order of convert_height_field_to_mesh (terrains/height_field/utils.py:119-173); the sensor
 * refresh serves ILX_RAYS_Z_PARALLEL rays over its regular cells from the height grid alone.  out[0..3] = patches,
 * faces inside patches, cells inside patches, cells whose rays take the height-grid path. */
int ilx_mesh_lattice_stats(ilx_mesh_t mesh, int64_t out[4]);
/* Diagnostics: with enable != 0 the sensor refreshes that follow count the rays they serve from the height grid in a
 * device counter owned by the mesh, starting from zero (enable == 0 stops the counting); served (or NULL) receives
 * the count before the call.  The
 * call synchronises the device when it reads the counter; it is for tests, not for the step path. */
int ilx_mesh_lattice_counter(ilx_mesh_t mesh, int enable, int64_t* served);
/* The same four numbers for host arrays, computed on the host alone (no device is touched, nothing is kept). */
int ilx_mesh_lattice_probe(const float* verts_host, int64_t num_verts, const int32_t* tris_host, int64_t num_tris,
                           int64_t out[4]);

/* raycast_mesh (ops.py:24-127) + raycast_mesh_kernel (kernels.py:13-75): closest hit per ray with
 * 0 <= t < max_dist; ray_hits (n,3) = start + t*dir, +inf on a miss; optional outputs (NULL to skip):
 * ray_distance (n) [+inf], ray_normal (n,3) unit face normal [+inf], ray_face_id (n) int32 [-1].
 * Ties in t go to the lowest face index.
 * hints: bit mask of ilx_ray_hint; a hint changes which kernel runs, never the result. */
enum ilx_ray_hint {
    ILX_RAYS_Z_PARALLEL = 1, /* the caller expects every direction to be (0, 0, +-z), e.g. a height-scan pattern */
    /* With ILX_RAYS_Z_PARALLEL, for the sensor entry points with attach_yaw_only (world directions = the pattern's
     * directions, ray_caster.py:247): a PROMISE that the caller has checked every pattern direction to be (0, 0, +-z).
     * The launch that would pick up other rays is then not queued; a ray that breaks the promise is left with a NaN in
     * ray_hits_w[..., 0] instead of a result. */
    ILX_RAYS_Z_PARALLEL_CHECKED = 2,
    /* For ilx_raycaster_refresh: a PROMISE that ray_starts_local / ray_directions_local hold what they held at the
     * previous refresh issued for this mesh on this stream (a sensor's pattern is fixed once it is initialised,
     * ray_caster.py:190-212).  The refresh then reuses the packed pattern and its bounding box that that refresh left in
     * the stream's workspace instead of reading and reducing the pattern again.  It has no effect when there was no
     * such refresh (first call, other arrays or ray count, workspace regrown). */
    ILX_RAYS_PATTERN_STATIC = 4,
    /* For ilx_raycaster_refresh: the refresh's preparation launch was issued by ilx_scene_update_prepare_raycaster for
     * this descriptor, on this stream, with nothing that touches the sensor in between; only the cast is launched.
     * Where a refresh has no preparation launch (no height-field patches, a pattern above 1 024 rays, no vertical-ray
     * hint) the hint changes nothing. */
    ILX_RAYS_PREPARED = 8,
    /* For ilx_raycaster_refresh with `height_scan` given: the caller reads only the height scan for now.  Where the
     * refresh goes over the height-field patches under ILX_RAYS_Z_PARALLEL_CHECKED (its cast is then the last launch and
     * nothing reads the hit records as a work list) the cast writes `height_scan` alone and leaves `ray_hits_w` as it
     * is: a quarter of the bytes.  RayCasterData.pos_w / quat_w are written as always, so the hit records of the same
     * refresh can be formed later by ilx_raycaster_update on those poses -- bit for bit what this refresh would have
     * stored (ray_caster.py:220-260 evaluates eagerly; the sensor is lazy anyway, sensor_base.py:197-205).  Every other
     * refresh ignores the hint and writes the hit records.  A ray that breaks the ILX_RAYS_Z_PARALLEL_CHECKED promise
     * is then left with a NaN in `height_scan`. */
    ILX_RAYS_HEIGHT_ONLY = 16
};
int ilx_raycast_mesh(ilx_mesh_t mesh, const float* ray_starts, const float* ray_directions, int64_t num_rays,
                     float max_dist, float* ray_hits, float* ray_distance, float* ray_normal, int32_t* ray_face_id,
                     int hints, void* stream);

/* Diagnostics (no counterpart in the reference): same cast, plus per ray the number of BVH nodes visited and
 * triangles tested in counters (n,2) int32. */
int ilx_raycast_mesh_stats(ilx_mesh_t mesh, const float* ray_starts, const float* ray_directions, int64_t num_rays,
                           float max_dist, float* ray_hits, int32_t* counters, void* stream);

/* RayCaster._update_buffers_impl (ray_caster.py:242-260) in one launch: for every env flagged in env_mask
 * (n) uint8 (NULL = all), rotate the local ray starts (num_rays,3) by the sensor orientation quat_w (n,4)
 * (yaw only when attach_yaw_only, directions then stay unrotated), add pos_w (n,3), cast, and write
 * ray_hits_w (n,num_rays,3).  If height_scan (n,num_rays) is not NULL it receives
 * pos_w.z - hit.z - height_offset (observations.py:165-173).  hints as in ilx_raycast_mesh. */
int ilx_raycaster_update(ilx_mesh_t mesh, const float* pos_w, const float* quat_w, const float* ray_starts_local,
                         const float* ray_directions_local, const uint8_t* env_mask, int attach_yaw_only,
                         float max_dist, float* ray_hits_w, float* height_scan, float height_offset, int64_t n,
                         int num_rays, int hints, void* stream);

/* SensorBase._update_outdated_buffers + RayCaster._update_buffers_impl (sensor_base.py:284-293,
 * ray_caster.py:220-260) for the envs flagged in is_outdated, in two launches and without temporaries:
 *   pos_w <- pose_pos + drift, quat_w <- pose_quat (flagged envs only), world rays, cast, ray_hits_w [, height_scan];
 *   then timestamp_last_update <- timestamp and is_outdated <- 0 for those envs.
 * pose_pos / pose_quat are strided views (row stride in floats), e.g. root_state_w and root_state_w + 3 with
 * stride 13; quaternion (w,x,y,z). */
typedef struct ilx_raycaster {
    ilx_mesh_t mesh;
    int64_t num_envs;
    int32_t num_rays, attach_yaw_only, hints;
    float max_distance;
    const float* pose_pos;
    const float* pose_quat;
    int64_t pose_pos_stride, pose_quat_stride;
    const float* drift;                 /* (n,3) or NULL */
    const float* ray_starts_local;      /* (r,3) */
    const float* ray_directions_local;  /* (r,3) */
    float* pos_w;                       /* RayCasterData (ray_caster_data.py:10-30) */
    float* quat_w;
    float* ray_hits_w;
    float* height_scan;                 /* (n,r) or NULL: pos_w.z - hit.z - height_offset */
    float height_offset;
    float* timestamp;                   /* sensor clock (sensor_base.py) */
    float* timestamp_last_update;
    uint8_t* is_outdated;
} ilx_raycaster;
int ilx_raycaster_refresh(const ilx_raycaster* rc, void* stream);
/* ilx_scene_update (declared above) for the last scene update before the scanner is read, which also issues the
 * preparation of the refresh that follows -- ilx_raycaster_refresh(rc with hints | ILX_RAYS_PREPARED) on the same stream
 * -- as part of the same launch: the thread that advances an env's sensor clocks (sensor_base.py:197-205) also turns the
 * sensor pose into the env's record once it knows whether the scanner is due.  One launch and its latency fewer on the
 * path to the ray hits.  `scene`'s ray-caster clock must be `rc`'s clock (ILX_EINVAL otherwise); the pose in `rc` must be
 * the one the refresh will see.  The first call for a pattern on a stream, and sensors whose refresh has no preparation
 * launch, are served by separate launches with the same results.  The two calls together leave every buffer as
 * ilx_scene_update followed by ilx_raycaster_refresh does. */
int ilx_scene_update_prepare_raycaster(const ilx_scene* scene, const ilx_raycaster* rc, void* stream);
/* ilx_raycaster_refresh for callers that cannot allocate (no counterpart in the reference): code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same results.  Where
 * ilx_raycaster_refresh keeps the scan pattern and one record per env between its launches in a per-stream workspace that
 * it grows with hipMalloc, this entry point uses a 3,162,144-byte buffer built into the library's code object (one per
 * GPU, held until the process ends) and calls no allocating routine.  Limit: 32,768 envs per call (at up to 1,024 rays
 * each); more is refused with ILX_ESHAPE before anything is queued.  All calls on one GPU share that buffer: calls that
 * overlap on one GPU (different streams or threads of the process) race -- keep them on one stream.  Nothing in the buffer has to be cleared between calls.
 * The limit applies where the refresh keeps something between its launches (a mesh with height-field patches, a
 * vertical pattern of up to 1,024 rays, attach_yaw_only); other refreshes keep nothing and have no such limit.
 * ILX_RAYS_PREPARED is refused with ILX_EINVAL (ilx_scene_update_prepare_raycaster prepares in the per-stream workspace,
 * which this entry point never reads) and ILX_RAYS_PATTERN_STATIC has no effect (the buffer keeps nothing from call to
 * call).  The buffer's device address is looked up when ilx_mesh_create runs on the device, not here, so the first call
 * may already be under capture. */
int ilx_raycaster_refresh_static(const ilx_raycaster* rc, void* stream);
/* No counterpart in the reference (Warp allocates as it goes): sets aside, now, the per-stream workspaces that
 * ilx_raycaster_refresh / ilx_scene_update_prepare_raycaster would otherwise allocate with hipMalloc on their first call
 * on a stream -- `streams` buffers sized for num_envs x num_rays (buffers set aside earlier and still unused count).  A
 * stream's first refresh of up to that size then takes one of them and allocates nothing; call it where allocation is
 * allowed (sensor set-up), so that no step under graph capture allocates. */
int ilx_raycaster_reserve(ilx_mesh_t mesh, int64_t num_envs, int32_t num_rays, int32_t streams);
/* No counterpart in the reference: drops what ILX_RAYS_PATTERN_STATIC lets a stream's workspace keep (the packed scan
 * pattern of the previous refresh), so the next refresh on every stream packs its pattern again.  For owners of pattern
 * arrays: call it when a pattern array is created or released -- an allocator may hand a released array's address to the
 * next pattern, which the promise behind ILX_RAYS_PATTERN_STATIC could not tell apart. */
int ilx_mesh_forget_patterns(ilx_mesh_t mesh);

/* RayCasterCamera._update_buffers_impl (sensors/ray_caster/ray_caster_camera.py:257-318) for the cameras flagged in
 * is_outdated (NULL = all), without temporaries:
 *   camera pose  = view pose composed with the offset: pos_w = view_pos + quat_apply(view_quat, offset_pos),
 *                  quat_w_world = view_quat * offset_quat (:415-430); frame += 1
 *   world rays   = quat_apply(quat_w_world, ray_starts) + pos_w, quat_apply(quat_w_world, ray_directions) (:269-272)
 *   cast         with max_dist 1e6 (:282) -> ray_hits_w [, normals]
 *   distance_to_image_plane = x of quat_apply(quat_inv(quat_w_world), depth * direction) (:294-301)
 *   distance_to_camera      = depth
 *   depth_clipping: MAX  -> both images clipped to max_distance, NaN (a miss times a zero direction component) -> max_distance
 *                   ZERO -> values > max_distance (and NaN) -> 0;  NONE -> left as they are (misses are inf / NaN)
 * then, when is_outdated is given, timestamp_last_update <- timestamp and is_outdated <- 0 for those cameras
 * (sensor_base.py:284-293).  ray_starts / ray_directions are per camera (n,r,3): intrinsics may differ between cameras
 * (set_intrinsic_matrices, :121-140).  Any of the three images may be NULL (data type not requested). */
enum { ILX_DEPTH_CLIP_NONE = 0, ILX_DEPTH_CLIP_MAX = 1, ILX_DEPTH_CLIP_ZERO = 2 };
typedef struct ilx_raycaster_camera {
    ilx_mesh_t mesh;
    int64_t num_envs;
    int32_t num_rays, depth_clipping;
    float max_distance;
    const float* view_pos;              /* strided rows: pose of the prim the camera is attached to */
    const float* view_quat;             /* (w,x,y,z) */
    int64_t view_pos_stride, view_quat_stride;
    const float* offset_pos;            /* (n,3) in the view frame */
    const float* offset_quat;           /* (n,4) in the "world" camera convention (+x forward, +z up) */
    const float* ray_starts;            /* (n,r,3) */
    const float* ray_directions;        /* (n,r,3) */
    float* pos_w;                       /* CameraData.pos_w (n,3) */
    float* quat_w_world;                /* CameraData.quat_w_world (n,4) */
    int64_t* frame;                     /* (n) or NULL */
    float* ray_hits_w;                  /* (n,r,3) */
    float* distance_to_image_plane;     /* (n,r) = (n,height,width,1) or NULL */
    float* distance_to_camera;          /* (n,r) or NULL */
    float* normals;                     /* (n,r,3) or NULL */
    float* timestamp;                   /* sensor clock, with is_outdated */
    float* timestamp_last_update;
    uint8_t* is_outdated;               /* (n) or NULL */
} ilx_raycaster_camera;
int ilx_raycaster_camera_refresh(const ilx_raycaster_camera* rc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Differential inverse kinematics — source/isaaclab/isaaclab/controllers/differential_ik.py and the action term
 * source/isaaclab/isaaclab/envs/mdp/actions/task_space_actions.py
 * ------------------------------------------------------------------------------------------------ */
#define ILX_IK_POSITION 0 /* command_type "position" (differential_ik_cfg.py:22-27) */
#define ILX_IK_POSE 1     /* command_type "pose" */
enum ilx_ik_method {      /* ik_method (differential_ik_cfg.py:37-38) */
    ILX_IK_PINV = 0,      /* differential_ik.py:197-203  k_val */
    ILX_IK_SVD = 1,       /* :204-219  k_val, min_singular_value */
    ILX_IK_TRANS = 2,     /* :220-226  k_val */
    ILX_IK_DLS = 3        /* :227-236  lambda_val */
};

/* Where the controlled frame is.  root_pos == NULL: ee_pos / ee_quat already are the frame pose the controller works
 * in (the plain DifferentialIKController interface, differential_ik.py:97-174).  root_pos != NULL: the action term's
 * _compute_frame_pose (task_space_actions.py:190-209): ee_* are the WORLD pose of the body, root_* the world pose of
 * the articulation root, and the optional fixed offset (body_offset cfg) is applied in the body frame; the Jacobian
 * handed to ilx_diff_ik_compute is then the WORLD Jacobian of that body and goes through jacobian_b (:140-149) and
 * _compute_frame_jacobian (:211-229) inside the kernel.  All pointers are strided row views (stride in floats),
 * quaternions (w,x,y,z). */
typedef struct ilx_ik_frame {
    const float* ee_pos;
    const float* ee_quat;
    int64_t ee_pos_stride, ee_quat_stride;
    const float* root_pos;
    const float* root_quat;
    int64_t root_pos_stride, root_quat_stride;
    int32_t has_offset;
    float offset_pos[3];
    float offset_rot[4];
} ilx_ik_frame;

/* DifferentialInverseKinematicsAction.process_actions (task_space_actions.py:155-167) fused with
 * DifferentialIKController.set_command (differential_ik.py:97-147):
 *   raw_actions != NULL: raw_store <- raw; processed <- clamp(raw * scale_host[k], clip) ; command = processed
 *   command != NULL    : the controller interface alone (command already in task space)
 *   -> ee_pos_des (n,3), ee_quat_des (n,4).  The action width follows action_dim (:79-86): 3, 6 (relative pose)
 *   or 7.  scale_host: HOST array of that width.  clip: (n,width,2) device or NULL.  frame may be NULL for an
 *   absolute pose command. */
int ilx_diff_ik_set_command(const ilx_ik_frame* frame, int command_type, int use_relative_mode, const float* raw_actions,
                            float* raw_store, float* processed, const float* scale_host, const float* clip,
                            const float* command, float* ee_pos_des, float* ee_quat_des, int64_t n, void* stream);

/* DifferentialIKController.compute (differential_ik.py:148-240), and with a world frame description the whole of
 * DifferentialInverseKinematicsAction.apply_actions (task_space_actions.py:169-181), in one launch:
 *   joint_pos_des[:, ids] = joint_pos[:, ids] + dq,  dq from the pose error and the 6 x num_joints Jacobian.
 * Jacobian element (env, row, col) is read at env * jacobian_env_stride + row * jacobian_row_stride +
 * (jacobian_cols ? jacobian_cols[col] : col), so a slice [:, body, :, joint_ids] of the simulator's Jacobian tensor
 * needs no copy.  joint_ids (device int32, or NULL for 0..num_joints-1) index joint_pos and joint_pos_des rows.
 * The 6x6 solves run in float64. */
int ilx_diff_ik_compute(const ilx_ik_frame* frame, int command_type, int ik_method, float k_val, float lambda_val,
                        float min_singular_value, const float* jacobian, int64_t jacobian_env_stride,
                        int64_t jacobian_row_stride, const int32_t* jacobian_cols, const float* joint_pos,
                        int64_t joint_pos_stride, const int32_t* joint_ids, const float* ee_pos_des,
                        const float* ee_quat_des, float* joint_pos_des, int64_t joint_pos_des_stride, int64_t n,
                        int num_joints, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISAACLAB_AMD_H */
