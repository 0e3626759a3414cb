/*
 * isaaclab_amd — fused MDP step: the managers' term loops as ONE kernel launch.
 *
 * Stands behind the post-physics half of ManagerBasedRLEnv.step
 * (source/isaaclab/isaaclab/envs/manager_based_rl_env.py:197-241):
 *   episode_length_buf += 1                                   (:199)
 *   TerminationManager.compute()                              (managers/termination_manager.py:151-174)
 *   RewardManager.compute(dt)                                 (managers/reward_manager.py:128-157)
 *   reset bookkeeping of the managers for the done envs       (manager_based_rl_env.py:346-391; the parts that
 *                                                              do not write into the simulator)
 *   CommandManager.compute(dt)                                (managers/command_manager.py:151-166)
 *   ObservationManager.compute()                              (managers/observation_manager.py:260-335)
 * with the term functions of envs/mdp/{rewards,terminations,observations}.py and of the velocity / reach /
 * cartpole task packages evaluated in registers, one wavefront per environment.
 *
 * The term lists are data (ilx_term arrays), exactly like the reference's term cfgs: the same kernel serves
 * every manager-based task built from the supported term kinds.
 */
#ifndef ISAACLAB_AMD_MDP_H
#define ISAACLAB_AMD_MDP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reward term kinds: envs/mdp/rewards.py (R), velocity/mdp/rewards.py (V), reach/mdp/rewards.py (RE),
 *      cartpole/mdp/rewards.py (CP).  p[] = parameters, ids = joint or body index list (count -1 = all). */
enum ilx_reward_kind {
    ILX_REW_IS_ALIVE = 0,                  /* R:31 */
    ILX_REW_IS_TERMINATED = 1,             /* R:36 */
    ILX_REW_IS_TERMINATED_TERM = 2,        /* R:41   flags = bit mask of termination term indices */
    ILX_REW_LIN_VEL_Z_L2 = 3,              /* R:76 */
    ILX_REW_ANG_VEL_XY_L2 = 4,             /* R:83 */
    ILX_REW_FLAT_ORIENTATION_L2 = 5,       /* R:90 */
    ILX_REW_BASE_HEIGHT_L2 = 6,            /* R:100  p0 = target height; flags&1: add mean ray-hit z */
    ILX_REW_BODY_LIN_ACC_L2 = 7,           /* R:125  ids = bodies */
    ILX_REW_JOINT_TORQUES_L2 = 8,          /* R:136  ids = joints */
    ILX_REW_JOINT_VEL_L1 = 9,              /* R:146 */
    ILX_REW_JOINT_VEL_L2 = 10,             /* R:153 */
    ILX_REW_JOINT_ACC_L2 = 11,             /* R:163 */
    ILX_REW_JOINT_DEVIATION_L1 = 12,       /* R:173 */
    ILX_REW_JOINT_POS_LIMITS = 13,         /* R:182 */
    ILX_REW_JOINT_VEL_LIMITS = 14,         /* R:199  p0 = soft_ratio */
    ILX_REW_APPLIED_TORQUE_LIMITS = 15,    /* R:226 */
    ILX_REW_ACTION_RATE_L2 = 16,           /* R:245 */
    ILX_REW_ACTION_L2 = 17,                /* R:250 */
    ILX_REW_UNDESIRED_CONTACTS = 18,       /* R:260  p0 = threshold; ids = sensor bodies */
    ILX_REW_CONTACT_FORCES = 19,           /* R:271  p0 = threshold */
    ILX_REW_TRACK_LIN_VEL_XY_EXP = 20,     /* R:287  p0 = std^2 */
    ILX_REW_TRACK_ANG_VEL_Z_EXP = 21,      /* R:301  p0 = std^2 */
    ILX_REW_FEET_AIR_TIME = 22,            /* V:25   p0 = threshold, p1 = step_dt + 1e-8 (host double -> float) */
    ILX_REW_FEET_AIR_TIME_POSITIVE_BIPED = 23, /* V:47 p0 = threshold */
    ILX_REW_FEET_SLIDE = 24,               /* V:69   ids = sensor bodies, ids2 = asset bodies */
    ILX_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP = 25, /* V:86  p0 = std^2 */
    ILX_REW_TRACK_ANG_VEL_Z_WORLD_EXP = 26,      /* V:99  p0 = std^2 */
    ILX_REW_POSITION_COMMAND_ERROR = 27,         /* RE:19 ids[0] = body */
    ILX_REW_POSITION_COMMAND_ERROR_TANH = 28,    /* RE:36 p0 = std */
    ILX_REW_ORIENTATION_COMMAND_ERROR = 29,      /* RE:55 */
    ILX_REW_JOINT_POS_TARGET_L2 = 30,            /* CP:19 p0 = target */
    ILX_REW_KIND_COUNT = 31
};

/* ---- termination term kinds: envs/mdp/terminations.py.  flags & 1 = time_out term (truncation). */
enum ilx_termination_kind {
    ILX_TERM_TIME_OUT = 0,                       /* :30 */
    ILX_TERM_COMMAND_RESAMPLE = 1,               /* :35  p0 = step_dt, p1 = num_resamples */
    ILX_TERM_BAD_ORIENTATION = 2,                /* :50  p0 = limit angle */
    ILX_TERM_ROOT_HEIGHT_BELOW_MINIMUM = 3,      /* :62  p0 = minimum height */
    ILX_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT = 4,  /* :90  p0, p1 = bounds */
    ILX_TERM_JOINT_VEL_OUT_OF_LIMIT = 5,         /* :108 */
    ILX_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT = 6,  /* :117 p0 = max velocity */
    ILX_TERM_JOINT_EFFORT_OUT_OF_LIMIT = 7,      /* :127 (as written: any(isclose(computed, applied))) */
    ILX_TERM_ILLEGAL_CONTACT = 8,                /* :150 p0 = threshold; ids = sensor bodies */
    ILX_TERM_KIND_COUNT = 9
};

/* ---- observation term kinds: envs/mdp/observations.py.  Post-processing per observation_manager.py:308-318:
 *      flags&1 uniform noise (p0 = n_min, p1 = n_max - n_min), flags&2 clip (p2, p3), flags&4 scale (weight). */
enum ilx_observation_kind {
    ILX_OBS_BASE_POS_Z = 0,                /* :33 */
    ILX_OBS_BASE_LIN_VEL = 1,              /* :40 */
    ILX_OBS_BASE_ANG_VEL = 2,              /* :47 */
    ILX_OBS_PROJECTED_GRAVITY = 3,         /* :54 */
    ILX_OBS_ROOT_POS_W = 4,                /* :61  (minus env origins) */
    ILX_OBS_ROOT_QUAT_W = 5,               /* :68  flags&8 = make_quat_unique */
    ILX_OBS_ROOT_LIN_VEL_W = 6,            /* :85 */
    ILX_OBS_ROOT_ANG_VEL_W = 7,            /* :92 */
    ILX_OBS_JOINT_POS = 8,                 /* :104 */
    ILX_OBS_JOINT_POS_REL = 9,             /* :114 */
    ILX_OBS_JOINT_POS_LIMIT_NORMALIZED = 10, /* :124 */
    ILX_OBS_JOINT_VEL = 11,                /* :140 */
    ILX_OBS_JOINT_VEL_REL = 12,            /* :150 */
    ILX_OBS_HEIGHT_SCAN = 13,              /* :165  offset in the term's `aux` field */
    ILX_OBS_LAST_ACTION = 14,              /* :512 */
    ILX_OBS_GENERATED_COMMANDS = 15,       /* :529 */
    ILX_OBS_EXTERNAL = 16,                 /* :177 body_incoming_wrench, :188 imu_orientation, :204 imu_ang_vel, :220 imu_lin_acc:
                                              a per-env buffer kept by another component, copied as it is; `aux` = slot in
                                              ilx_mdp.external_obs */
    ILX_OBS_KIND_COUNT = 17
};

#define ILX_OBS_FLAG_NOISE 1
#define ILX_OBS_FLAG_CLIP 2
#define ILX_OBS_FLAG_SCALE 4
#define ILX_OBS_FLAG_QUAT_UNIQUE 8
#define ILX_MAX_EXTERNAL_OBS 4

typedef struct ilx_term {
    int32_t kind;
    int32_t flags;
    float weight;       /* reward weight (reward_manager.py:148) or observation scale */
    float p[4];
    float aux;
    int32_t ids_off, ids_cnt;     /* into ilx_mdp.ids; cnt = -1 selects every joint / body (slice(None)) */
    int32_t ids2_off, ids2_cnt;
} ilx_term;

#define ILX_CMD_NONE 0      /* command tensor is maintained by the caller */
#define ILX_CMD_VELOCITY 1  /* UniformVelocityCommand: envs/mdp/commands/velocity_command.py */
#define ILX_CMD_POSE 2      /* UniformPoseCommand:     envs/mdp/commands/pose_command.py */

/* phases (bit mask) — one launch may run any contiguous subset, e.g. TERMINATION|REWARD before the simulator
 * resets the done envs and COMMAND|OBSERVATION after it */
#define ILX_PHASE_TERMINATION 1
#define ILX_PHASE_REWARD 2
#define ILX_PHASE_RESET 4
#define ILX_PHASE_COMMAND 8
#define ILX_PHASE_OBSERVATION 16
#define ILX_PHASE_ALL 31
typedef struct ilx_mdp {
    /* sizes */
    int64_t num_envs;
    int32_t num_joints, num_bodies;          /* articulation */
    int32_t num_sensor_bodies, history_len;  /* contact sensor */
    int32_t num_rays, action_dim, command_dim, obs_dim;
    /* articulation state (ArticulationData field layout, all (n, ...) float32) */
    const float* root_state_w;               /* (n,13) */
    const float* joint_pos;                  /* (n,j) */
    const float* joint_vel;
    const float* joint_acc;
    const float* applied_torque;
    const float* computed_torque;
    const float* default_joint_pos;
    const float* default_joint_vel;
    const float* soft_joint_pos_limits;      /* (n,j,2) */
    const float* soft_joint_vel_limits;      /* (n,j) */
    const float* body_state_w;               /* (n,b,13) or NULL */
    const float* body_lin_acc_w;             /* (n,b,3) or NULL */
    const float* env_origins;                /* (n,3) or NULL */
    float gravity_dir[3];
    /* contact sensor (ContactSensorData) */
    const float* net_forces_w_history;       /* (n,h,cb,3) */
    const float* current_air_time;           /* (n,cb) */
    const float* last_air_time;
    const float* current_contact_time;
    const float* last_contact_time;
    /* ray caster (RayCasterData) */
    const float* ray_hits_w;                 /* (n,r,3) */
    const float* sensor_pos_w;               /* (n,3) */
    /* action manager */
    float* action;                           /* (n,a); the RESET phase zeroes the rows of done envs (action_manager.py:309-311) */
    float* prev_action;
    /* episode bookkeeping */
    int64_t* episode_length_buf;             /* (n) */
    int64_t max_episode_length;
    float step_dt;                           /* dt handed to RewardManager.compute / CommandManager.compute */
    float max_episode_length_s;
    /* termination manager outputs */
    int32_t num_terminations;
    const ilx_term* terminations;            /* HOST pointer, num_terminations entries */
    uint8_t* term_dones;                     /* (num_terminations, n) */
    uint8_t* terminated;                     /* (n) */
    uint8_t* time_outs;                      /* (n) */
    uint8_t* reset_buf;                      /* (n) terminated | time_outs */
    /* reward manager */
    int32_t num_rewards;
    const ilx_term* rewards;                 /* HOST pointer */
    float* reward_buf;                       /* (n) */
    float* episode_sums;                     /* (num_rewards, n) */
    float* step_reward;                      /* (n, num_rewards) */
    /* command term */
    int32_t command_kind;
    float* command;                          /* (n, command_dim): vel_command_b or pose_command_b */
    float* heading_target;                   /* (n) velocity */
    uint8_t* is_heading_env;                 /* (n) */
    uint8_t* is_standing_env;                /* (n) */
    float* time_left;                        /* (n) */
    int64_t* command_counter;                /* (n) */
    float* command_metrics;                  /* (2, n): error_vel_xy / error_vel_yaw, or position / orientation error */
    const float* command_samples;            /* (n,7) pre-drawn values used by resampling envs, or NULL -> in-kernel
                                                counter-based generator keyed by (seed, step, env) */
    const float* reset_command_samples;      /* (n,7) same, consumed by CommandTerm.reset of the done envs */
    float cmd_range[7][2];                   /* time; velocity: vx, vy, wz, heading | pose: x, y, z, roll, pitch, yaw */
    float rel_heading_envs, rel_standing_envs, heading_control_stiffness;
    int32_t heading_command;
    int32_t pose_body_index;                 /* pose: body whose pose is tracked */
    int32_t make_quat_unique;
    float max_command_step;                  /* resampling_time_range[1] / step_dt (velocity metrics) */
    /* observation manager (one concatenated group) */
    int32_t num_observations;
    const ilx_term* observations;            /* HOST pointer */
    float* obs;                              /* (n, obs_dim) */
    const float* noise_u;                    /* (n, obs_dim) U[0,1) draws for the noisy terms, or NULL -> generator */
    /* id lists */
    const int32_t* ids;                      /* HOST pointer */
    int32_t num_ids;
    /* logging partial sums, accumulated over the done envs of this launch (device, zeroed by the caller):
     * [0, R)      sum of episode_sums per reward term        (reward_manager.py:115-121)
     * [R, R+T)    count of each termination term             (termination_manager.py:142-144)
     * [R+T, +2)   sum of the two command metrics             (command_manager.py:136-142)
     * [R+T+2]     number of done envs */
    float* log_sums;
    /* generator */
    uint64_t seed;
    uint64_t step_index;
    int32_t phases;
    /* buffers read by ILX_OBS_EXTERNAL terms: slot s is (n, external_obs_dim[s]) float32, dense */
    const float* external_obs[ILX_MAX_EXTERNAL_OBS];
    int32_t external_obs_dim[ILX_MAX_EXTERNAL_OBS];
    /* generator, continued: when not NULL, one uint64 in device memory that the launch adds to step_index.  The caller
     * advances it between launches (a one-word add on the stream).  A launch recorded in a HIP graph replays its
     * arguments; with the count in device memory and its advance recorded in the same graph, every replay still draws
     * fresh numbers.  The launches of one env step (phases issued separately) read the same value. */
    const uint64_t* step_index_device;
    /* (n, num_rays) or NULL: the height scan as ilx_raycaster_refresh writes it (`height_scan` there, computed with
     * `height_offset` = the observation term's offset): sensor height - hit height - offset.  When given, the
     * height_scan observation columns are read from it instead of being formed from ray_hits_w / sensor_pos_w -- the
     * same float operations, a third of the bytes (the hits are 12-byte records of which the term needs one word). */
    const float* height_scan;
} ilx_mdp;


/* Opaque prepared plan: term tables and id lists copied to the device once. */
typedef struct ilx_mdp_plan_s* ilx_mdp_plan_t;
int ilx_mdp_plan_create(const ilx_mdp* desc, ilx_mdp_plan_t* out);
void ilx_mdp_plan_destroy(ilx_mdp_plan_t plan);
/* Re-read the pointer / scalar fields of `desc` (term tables are fixed at creation) and launch `phases`.
 *
 * Two kinds of kernel stand behind this call.  The plan-interpreting kernel serves every term list.  For the term lists
 * of the reference's own task configurations -- velocity_env_cfg.py:119-265 (ANYmal-C rough), config/h1/rough_env_cfg.py:19-115,
 * classic/cartpole/cartpole_env_cfg.py:58-181, manipulation/reach/reach_env_cfg.py:65-208 -- the library also holds a
 * kernel compiled for exactly that list (every term straight-line code, sizes known at compile time).  A plan whose
 * term kinds, index-list lengths, zero weights and sizes equal one of those lists (parameters and weights are free) is
 * served by that kernel whenever the call asks for all phases, every optional output is given and, for the height
 * scan, `height_scan` is given; every other call goes to the plan-interpreting kernel.  Both kernels write the same
 * bits into every output (log_sums: the same addends in another order). */
int ilx_mdp_step(ilx_mdp_plan_t plan, const ilx_mdp* desc, void* stream);
/* ilx_scene_update(scene) (isaaclab_amd.h: InteractiveScene.update of the last physics substep, scene/interactive_scene.py:
 * 496-513) followed by ilx_mdp_step, as ManagerBasedRLEnv.step issues them (manager_based_rl_env.py:190-199): the results
 * are those of the two calls.  Where a compiled kernel serves the plan, the scene holds an articulation only (no contact
 * sensor, no ray caster clock) and no term of the list reads what the update writes (joint_acc), the update's workgroups
 * are part of the managers' grid -- one launch instead of two (the Cartpole and Reach lists); every other case is the two
 * launches. */
struct ilx_scene;
int ilx_mdp_step_scene_update(ilx_mdp_plan_t plan, const ilx_mdp* desc, const struct ilx_scene* scene, void* stream);
/* No counterpart in the reference: the name of the kernel that serves full steps of this plan -- "generic" or the
 * compiled kernel's ("anymal_c_rough", "h1_rough", "cartpole", "reach_j9_a7", ...).  The string is static. */
const char* ilx_mdp_plan_kernel(ilx_mdp_plan_t plan);
/* No counterpart in the reference: compiled == 0 sends every step of the plan to the plan-interpreting kernel (for
 * checking one kernel against the other); non-zero restores the default. */
int ilx_mdp_plan_select_kernel(ilx_mdp_plan_t plan, int compiled);
/* What a caller does between the launches of two env steps, as one small launch: clears `log_sums` (`count` floats, or
 * NULL: the reference's managers rebuild their `extras["log"]` entries at every reset, reward_manager.py:100-126) and
 * adds one to `step_index_device` (or NULL).  For steps recorded in a HIP graph, where each extra node costs. */
int ilx_mdp_begin_step(float* log_sums, int32_t count, uint64_t* step_index_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISAACLAB_AMD_MDP_H */
