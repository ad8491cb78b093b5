/* mjsim.h — C ABI of libmjsim.so: the MI355X-native batched replacement for the
 * reference's per-step hot path.
 *
 * The reference has no native boundary (it is Python over the third-party
 * `mujoco`/`dm_control` wheels), so each entry point below cites the Python
 * interface it replaces (paths relative to /root/reference/mujoco_sim/). A
 * handle steps N independent environments of ONE task on ONE GPU; all pointers
 * named *_dev are DEVICE pointers owned by the caller (PyTorch-ROCm tensors);
 * the engine owns only its persistent struct-of-arrays state. Every call
 * enqueues work on the caller's HIP stream (`stream` = hipStream_t, NULL = the
 * default stream) and never synchronises. Return value: 0 = ok, negative =
 * MJS_ERR_*; nothing throws across the ABI; per-env faults are data
 * (`mjs_outputs.fault`), not errors. A handle is not thread-safe.
 * Binding stubs for the reference side: INTEGRATION.md.
 */
#ifndef MJSIM_H
#define MJSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): mjs_config starts with struct_size (validated by mjs_create: a caller built against another header is refused
 * instead of being read past its end); Robot-Reach / Button-Push state blocks grew 18 rows (qacc_warmstart, carried cos / sin):
 * checkpoints of abi 1 do not fit mjs_set_state any more (mjs_state_dim reports the new widths). */
#define MJS_ABI_VERSION 3

/* tasks (environments/tasks/*.py) */
enum {
  MJS_TASK_POINTMASS_REACH = 0, /* tasks/point_reach.py */
  MJS_TASK_ROBOT_REACH = 1,     /* tasks/robot_reach.py */
  MJS_TASK_PLANAR_PUSH = 2,     /* tasks/robot_planar_push.py (intended semantics; the reference's mesh blocks by default, DESIGN D-9) */
  MJS_TASK_BUTTON_PUSH = 3      /* tasks/robot_push_button.py */
};
/* Button-Push action spaces (robot_push_button.py:35-36,143-157): absolute joints + gripper (7-D, the
 * registered default) or absolute TCP position + gripper (4-D). The last component is the commanded finger opening in
 * metres, [0, 0.085] (Robotiq2f85.move, gripper.py:77-84): it drives the reduced 2F-85 of this build (driver angle and
 * velocity = rows 16, 17 of the Button-Push state). Other tasks ignore the field. */
enum { MJS_ACTION_ABS_JOINT = 0, MJS_ACTION_ABS_EEF = 1 };
/* reward types: point_reach.py:11-14, robot_reach.py:37-38 */
enum { MJS_REW_SPARSE = 0, MJS_REW_DENSE_POTENTIAL = 1, MJS_REW_DENSE_NEG_DISTANCE = 2, MJS_REW_DENSE_BIASED_NEG_DISTANCE = 3 };
/* dm_env StepType as produced by composer.Environment (dmc2gym.py:144-145 reads .last()) */
enum { MJS_STEP_FIRST = 0, MJS_STEP_MID = 1, MJS_STEP_LAST = 2 };
/* auto-reset: NEXT_STEP is composer.Environment's behaviour (the step after LAST ignores the
 * action and returns the reset observation); SAME_STEP is what SB3 VecEnv expects
 * (scripts/sb3/reach_sac.py:93-96): reset immediately, expose `terminal_obs`. */
enum { MJS_AUTORESET_NEXT_STEP = 0, MJS_AUTORESET_SAME_STEP = 1, MJS_AUTORESET_DISABLED = 2 };
/* kernel variants (results identical up to rounding; used for A/B profiles). Variant 1 = the first-generation kernels:
 * the single-wavefront step kernels of Robot-Reach / Button-Push (default: two role-specialised wavefronts) and the
 * 8x8-tile camera kernel for every image (default: the rectangle walk for images up to 64x64) */
enum { MJS_VARIANT_DEFAULT = 0, MJS_VARIANT_SINGLE_WAVE = 1, MJS_VARIANT_TWO_ROLES = 2 /* Robot-Reach: round 1's two-wavefront kernel */,
       MJS_VARIANT_RESET_GROUPS = 3 /* Robot-Reach (Button-Push accepts it too): the default kernel with the next-step auto-resets on workgroups
                                       of their own (other CUs, same launch). For episodes that END AT DIFFERENT TIMES
                                       (terminate_on_success): 38 instead of 54 us per launch at 4096 envs with 1 % of the envs ending
                                       in every step; with synchronous episodes it costs 0.9 us per launch. Bitwise the default's
                                       results. (The Python host picks it when terminate_on_success is set.) */ };
/* Shard invariance (env i of a sharded job == env i of the whole job, global seeds via env_index_offset) is BITWISE as long as
 * every handle of the comparison launches the same kernel. Robot-Reach with MJS_VARIANT_DEFAULT switches kernels at 16384 envs
 * per handle (three-wavefront kernel up to there, the two-role kernel above: same results to rounding only), so shards of at
 * most 16384 envs are bit-identical to each other and to any whole job of at most 16384; pin kernel_variant to compare across
 * that size. All other tasks use one kernel at every size. */
/* mjs_outputs.fault bits */
enum {
  MJS_FAULT_BAD_STATE = 1,            /* NaN / huge qpos, qvel or qacc: dm_control's PhysicsError path (episode ends, reward 0, discount 0) */
  MJS_FAULT_IK_FAILED = 2,            /* servoL found no IK solution (the reference raises ValueError): the env holds its joints */
  MJS_FAULT_LIMIT_COLDSTART = 4,      /* constraint rows (joint limits / contacts) were active in some substep (informational; the solver is
                                         warm-started like mj_fwdConstraint: qacc_warmstart = the previous Physics.step()'s solution) */
  MJS_FAULT_UNSUPPORTED_CONTACT = 8,  /* a lane had more simultaneously ACTIVE contacts than the constraint stage holds (24 on the robot
                                         scenes' robust path: arm links on the floor are solved since abi 2; Planar-Push with the arm on
                                         the floor WHILE it is coupled to a block: 5 arm-floor contacts; articulated gripper: 16 contacts);
                                         the surplus made no rows */
  MJS_FAULT_FASTPATH_VIOLATED = 16    /* the row-free fast path's a-posteriori check failed (a joint left its range, or an arm geom / the
                                         gripper stand-in touches something, at the end of a step taken without constraint rows): this env's
                                         step is unreliable */
};

enum {
  MJS_OK = 0,
  MJS_ERR_INVALID_ARG = -1,
  MJS_ERR_NO_DEVICE = -2,
  MJS_ERR_HIP = -3,
  MJS_ERR_ALLOC = -4,
  MJS_ERR_UNSUPPORTED = -5
};

typedef struct mjs_handle mjs_handle;

typedef struct {
  uint32_t struct_size;         /* sizeof(mjs_config) of the caller's header: mjs_create refuses any other value */
  int32_t task;                 /* MJS_TASK_* */
  int32_t num_envs;             /* N on this GPU */
  int32_t device;               /* HIP device ordinal */
  int32_t reward_type;          /* MJS_REW_*; <0 = task default */
  int32_t autoreset;            /* MJS_AUTORESET_* */
  int32_t terminate_on_success; /* Robot-Reach opt-in; ignored by Pointmass */
  int32_t env_index_offset;     /* global index of local env 0 (multi-GPU shards keep global seeds) */
  int32_t kernel_variant;       /* MJS_VARIANT_*: 0 = default (tuned); others for A/B profiling */
  double time_limit;            /* composer.Environment(time_limit=...) (__init__.py:21); <=0 = task default */
  int32_t action_type;          /* MJS_ACTION_* (Button-Push only) */
  int32_t button_disturbances;  /* Button-Push only: after each control step an active, released switch is
                                 * deactivated with probability 0.01 from the env's stream (robot_push_button.py:159-165) */
  int32_t n_objects;            /* Planar-Push only: number of blocks, 1..5 (robot_planar_push.py:61; <= 0: 2 = the registered env, :315,
                                 * and BASELINE config 4). 1..2 run the 2-slot kernel, 3..5 the 5-slot kernel */
  int32_t max_episode_steps;    /* Planar-Push only: RobotTask step limit (tasks/base.py:47-51); <= 0: 500 */
  int32_t block_shape;          /* Planar-Push only: MJS_BLOCKS_MESH (0, the reference: GoogleBlockProp.sample_random_object per episode,
                                 * google_block.py:55-68, category / colour / scale from the env's seeded stream) or MJS_BLOCKS_BOX (round 1's box
                                 * stand-in of the cube mesh's bounding box, scale 1: a documented fast variant) */
  int32_t gripper_model;        /* Button-Push only (abi 3): MJS_GRIPPER_REDUCED (0: the one-coordinate 2F-85 of DESIGN.md D-1b on the 6-dof arm, the
                                 * fast default) or MJS_GRIPPER_ARTICULATED (1: the Robotiq 2F-85 as entities/eef/gripper.py:36-98 attaches it -
                                 * eight hinges, two connect equalities, the driver coupling, the fixed-tendon fingers_actuator, pad boxes,
                                 * elliptic cones with impratio 10: nv = 14; mjs_env_state_dim() reports its wider state) */
  int32_t reserved0;            /* 0 */
} mjs_config;
enum { MJS_BLOCKS_MESH = 0, MJS_BLOCKS_BOX = 1 };
enum { MJS_GRIPPER_REDUCED = 0, MJS_GRIPPER_ARTICULATED = 1 };

/* Per-step outputs. Device pointers, caller-owned, any may be NULL.
 * Replaces the (obs, reward, terminated, truncated, info) tuple of
 * DMCEnvironmentAdapter.step (environments/dmc2gym.py:133-155). */
typedef struct {
  double* obs;           /* [N, obs_dim] row-major, layout: mjs_obs_dim() */
  double* terminal_obs;  /* [N, obs_dim]; written only under SAME_STEP for envs that ended */
  double* reward;        /* [N]; 0 on FIRST (dm_env: None) */
  double* discount;      /* [N]; info["discount"] (dmc2gym.py:153); 1 on FIRST (dm_env: None) */
  uint8_t* terminated;   /* [N]; last && discount == 0 (dmc2gym.py:145) */
  uint8_t* truncated;    /* [N]; last && discount > 0  (dmc2gym.py:144) */
  uint8_t* is_success;   /* [N]; info["is_success"] (dmc2gym.py:149-150) */
  uint8_t* step_type;    /* [N]; MJS_STEP_* */
  uint8_t* fault;        /* [N]; MJS_FAULT_* bits */
  int32_t* ncon;         /* [N]; detected contacts after the step (MuJoCo's d->ncon) */
} mjs_outputs;

/* Rollout outputs: same fields with a leading time axis [T, N, ...]. */

const char* mjs_version(void);
/* flat observation width: Pointmass {pointmass/position(2), goal_position(2)} (point_reach.py:115-118);
 * Robot-Reach {ur5e/tcp_position(3), ur5e/joint_configuration(6), target_position(3)}
 * (robot_reach.py:134-137, robot.py:292-298); Button-Push {ur5e/joint_configuration(6),
 * ur5e/tcp_position(3), switch/position(3), switch/active(1)} (robot_push_button.py:113-119, switch.py:99-108);
 * Planar-Push {ur5e/tcp_position(3), target_position(2), block_positions(2 per block slot; 2 slots here,
 * 5 for handles with n_objects 3..5: mjs_env_obs_dim)}
 * (robot_planar_push.py:120-126,134-138,178-179) */
int mjs_obs_dim(int task);
/* action width: 2 (point_reach.py:204-209) / 3 (robot_reach.py:187-201) / Button-Push default 7
 * (robot_push_button.py:181-203) / Planar-Push 2 (absolute TCP xy, robot_planar_push.py:197-201,222-228) */
int mjs_action_dim(int task);
/* action width for a task + MJS_ACTION_* pair (Button-Push: 7 or 4); equals mjs_action_dim otherwise */
int mjs_action_dim_for(int task, int action_type);
/* number of float64 per env in mjs_get_state / mjs_set_state: the task's state rows + 1 (the flag byte as a double, last row).
 * Robot-Reach: q6, v6, time, target3, qacc_warmstart6, cos6, sin6; Button-Push: q6, v6, time, switch position3, gripper driver
 * angle and velocity, qacc_warmstart6, cos6, sin6. cos / sin are the kernels' carried cache of the joint angles: mjs_set_state
 * keeps rows that belong to the given q (a checkpoint resumes bit for bit) and rewrites them with exact values otherwise (a
 * caller that edits q need not touch them); it also recomputes the flags that are functions of the configuration. */
int mjs_state_dim(int task);
/* the same two widths for a created handle: Planar-Push with n_objects 3..5 uses 5 block slots
 * (obs 5 + 2*5 = 15, state 1 + 17 + 13*5 = 83); the per-task queries above describe the 2-slot layout */
int mjs_env_obs_dim(const mjs_handle* h);
int mjs_env_state_dim(const mjs_handle* h);
/* algorithmic HBM bytes one env-step moves (state R+W, action, outputs), from the real layout */
int mjs_algorithmic_bytes_per_env_step(int task);
/* physics substeps per control step: 5 (point_reach.py:24-25) / 20 (robot_reach.py:62-63, robot_planar_push.py:51-52,
 * robot_push_button.py:38-39) */
int mjs_substeps(int task);

/* Replaces task + composer.Environment + DMCEnvironmentAdapter construction
 * (mujoco_sim/__init__.py:19-23). Allocates the SoA state for N envs. */
int mjs_create(const mjs_config* cfg, mjs_handle** out);
void mjs_destroy(mjs_handle* h);
const char* mjs_last_error(const mjs_handle* h);

/* Replaces DMCEnvironmentAdapter.seed (dmc2gym.py:126-131): env i gets the numpy-legacy
 * MT19937 stream RandomState(base_seed + env_index_offset + i) (reach_sac.py:84 seeds
 * sub-env `rank` with seed+rank). The stream lives on the device. */
int mjs_seed(mjs_handle* h, uint32_t base_seed, void* stream);

/* Replaces DMCEnvironmentAdapter.reset -> composer.Environment.reset (dmc2gym.py:157-163):
 * starts a new episode for every env whose mask byte is non-zero (mask_dev NULL = all). */
int mjs_reset(mjs_handle* h, const uint8_t* mask_dev, const mjs_outputs* out, void* stream);

/* Replaces DMCEnvironmentAdapter.step -> composer.Environment.step -> n_sub x Physics.step
 * (dmc2gym.py:133-155): one control step of all N envs, one kernel launch.
 * actions_dev: float64 [N, action_dim]. */
int mjs_step(mjs_handle* h, const double* actions_dev, const mjs_outputs* out, void* stream);

/* T consecutive mjs_step calls with precomputed actions [T, N, action_dim]; outputs carry a
 * leading T axis. Open-loop rollouts (random / scripted policies, point_reach.py:218-240). */
int mjs_rollout(mjs_handle* h, const double* actions_dev, int32_t T, const mjs_outputs* out, void* stream);

/* Replaces Camera.get_rgb_image -> physics.render(height, width, camera_id) (entities/camera.py:94-103)
 * and DMCEnvironmentAdapter.render (dmc2gym.py:165-168) for the task's scene camera
 * (camera = MJS_CAMERA_SCENE) and, for Button-Push, the camera mounted on the flange
 * (MJS_CAMERA_WRIST, robot_push_button.py:90-96): rgb_dev uint8 [N, height, width, 3]. Own ray caster,
 * cannot match OpenGL pixels. */
enum { MJS_CAMERA_SCENE = 0, MJS_CAMERA_WRIST = 1 };
int mjs_render(mjs_handle* h, int32_t camera, int32_t height, int32_t width, uint8_t* rgb_dev, void* stream);

/* Test hook: the device implementation of ur5e.inverse_kinematics_closest (entities/robots/robot.py:33-37)
 * on n independent inputs: flange poses T_dev [n, 12] (row-major 3x3 rotation then translation),
 * guesses [n, 6] -> q_dev [n, 6], ok_dev [n]. No handle needed; device = current HIP device. */
int mjs_debug_ur5e_ik(const double* T_dev, const double* guess_dev, double* q_dev, uint8_t* ok_dev, int32_t n, void* stream);

/* Replaces Robot.get_joint_positions_from_tcp_pose (entities/robots/robot.py:33-37,140-151) for the top-down
 * TCP orientation every task uses: n TCP positions [n, 3] + current joints [n, 6] -> closest IK solution
 * q_dev [n, 6] (the guess itself where none exists), ok_dev [n]. What a host policy needs to turn a
 * Cartesian target into a Button-Push joint action (robot_push_button.py:287-291). */
int mjs_ur5e_tcp_to_joints(const double* tcp_pos_dev, const double* guess_dev, double* q_dev, uint8_t* ok_dev, int32_t n, void* stream);

/* Replaces the Robot entity's control API on a stand-alone UR5e (entities/robots/robot.py:198-272): Robot.moveJ :211-216,
 * Robot.movej_IK :198-209, Robot.servoL :218-225, Robot.servoJ :227-259, then n_substeps x (Robot.before_substep :261-272 +
 * JointTrajectory.get_target_joint_positions joint_trajectory.py:41-47; Physics.step), then Robot.get_tcp_pose :153-168.
 * This is the component the reference's own tests drive (test/test_ur_control_api.py:7-82: UR5e() + raw mjcf.Physics at
 * the XML's timestep); n independent robots, one lane each.
 *   state_dev   float64 [n, MJS_UR_STATE], in/out: q[6], v[6], ctrl[6], time, trajectory active, q0[6], q1[6], t0, t1
 *   target_dev  float64 [n, 7]: 6 joint angles (MOVEJ / SERVOJ) or a TCP pose xyz + scalar-LAST quaternion
 *               (MOVEJ_IK / SERVOL, type_aliases.py:6-10); unused for MJS_UR_CMD_NONE
 *   param       speed [rad/s] for MOVEJ / MOVEJ_IK, duration [s] for SERVOL / SERVOJ
 *   eef         MJS_UR_EEF_NONE (bare flange, TCP = flange) or MJS_UR_EEF_GRIPPER (lumped 2F-85, TCP offset 0.174)
 *   dt          physics timestep (the menagerie XML sets none: MuJoCo's default 0.002)
 *   tcp_pose_out_dev float64 [n, 7] or NULL; status_dev uint8 [n] or NULL: bit 0 = the command's IK found a solution
 *               (movej_IK prints and returns, servoL raises), bit 1 = a joint left its range (limit rows are not modelled by
 *               this entry point), bit 2 = non-finite acceleration. No handle; device = current HIP device. */
#define MJS_UR_STATE 34
enum { MJS_UR_CMD_NONE = 0, MJS_UR_CMD_MOVEJ = 1, MJS_UR_CMD_MOVEJ_IK = 2, MJS_UR_CMD_SERVOL = 3, MJS_UR_CMD_SERVOJ = 4 };
enum { MJS_UR_EEF_NONE = 0, MJS_UR_EEF_GRIPPER = 1 };
int mjs_ur5e_robot_run(double* state_dev, const double* target_dev, int32_t command, double param, int32_t n_substeps, int32_t eef, double dt,
                       double* tcp_pose_out_dev, uint8_t* status_dev, int32_t n, void* stream);

/* checkpoint / resume of the physics+task state: float64 [state_dim, N] ... */
int mjs_get_state(mjs_handle* h, double* state_dev, void* stream);
int mjs_set_state(mjs_handle* h, const double* state_dev, void* stream);
/* ... and of the per-env MT19937 streams: mt_dev uint32 [624, N], pos_dev int32 [N]
 * (what pickling env._random_state would capture in the reference, dmc2gym.py:129) */
int mjs_get_rng_state(mjs_handle* h, uint32_t* mt_dev, int32_t* pos_dev, void* stream);
int mjs_set_rng_state(mjs_handle* h, const uint32_t* mt_dev, const int32_t* pos_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MJSIM_H */
