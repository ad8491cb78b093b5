/* mjs_oracle.h — CPU float64 ORACLE for the env-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under mujoco_sim_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker.
 *
 * PARITY UNPINNED (physics): the arithmetic of the reference's hot path lives in
 * third-party packages (mujoco, dm_control, ur_analytic_ik, mujoco_menagerie
 * assets) that are neither vendored nor version-pinned in /root/reference
 * (setup.py:11-20) and are not installed here, so this restatement of their
 * published algorithms cannot be checked against outputs of the reference.
 * What IS pinned (tests/test_oracle_known_answers.py): numpy RandomState reset
 * draws (dmc2gym.py:126-131 seeding), float64 time-limit crossings, the
 * reference's own test tolerances (test/test_ur_control_api.py,
 * test/test_ur_frame_matches_real.py) and closed-form answers.
 *
 * Assumed engine semantics = MuJoCo 3.x defaults.
 */
#ifndef MJS_ORACLE_H
#define MJS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OM_MAXBODY 24
#define OM_MAXJNT 24
#define OM_MAXQ 48
#define OM_MAXV 36
#define OM_MAXU 8
#define OM_MAXGEOM 40
#define OM_MAXSITE 8
#define OM_MAXEQ 4
#define OM_MAXTENDON 1
#define OM_MAXCON 48
#define OM_MAXEFC 320
#define OM_MAXMOCAP 1

/* MuJoCo enums (values as in mjmodel.h) */
enum { OM_JNT_FREE = 0, OM_JNT_BALL = 1, OM_JNT_SLIDE = 2, OM_JNT_HINGE = 3 };
enum { OM_GEOM_PLANE = 0, OM_GEOM_SPHERE = 2, OM_GEOM_CAPSULE = 3, OM_GEOM_CYLINDER = 5, OM_GEOM_BOX = 6, OM_GEOM_MESH = 7 };
enum { OM_INT_EULER = 0, OM_INT_IMPLICITFAST = 3 };
enum { OM_CNSTR_EQUALITY = 0, OM_CNSTR_LIMIT_JOINT = 3, OM_CNSTR_CONTACT_FRICTIONLESS = 5, OM_CNSTR_CONTACT_PYRAMIDAL = 6, OM_CNSTR_CONTACT_ELLIPTIC = 7 };
enum { OM_EQ_CONNECT = 0, OM_EQ_WELD = 1, OM_EQ_JOINT = 2 };   /* mjtEq */
enum { OM_CONE_PYRAMIDAL = 0, OM_CONE_ELLIPTIC = 1 };            /* mjtCone */
enum { OM_TRN_JOINT = 0, OM_TRN_TENDON = 3 };                    /* mjtTrn */

typedef struct {
  int nbody, njnt, nq, nv, nu, ngeom, nsite, neq, nmocap;
  /* bodies */
  int body_parent[OM_MAXBODY], body_jntadr[OM_MAXBODY], body_jntnum[OM_MAXBODY];
  int body_dofadr[OM_MAXBODY], body_dofnum[OM_MAXBODY], body_mocapid[OM_MAXBODY], body_weldid[OM_MAXBODY];
  double body_pos[OM_MAXBODY][3], body_quat[OM_MAXBODY][4], body_ipos[OM_MAXBODY][3], body_iquat[OM_MAXBODY][4];
  double body_mass[OM_MAXBODY], body_inertia[OM_MAXBODY][3], body_gravcomp[OM_MAXBODY];
  double body_invweight0[OM_MAXBODY][2];
  /* joints / dofs */
  int jnt_type[OM_MAXJNT], jnt_body[OM_MAXJNT], jnt_qposadr[OM_MAXJNT], jnt_dofadr[OM_MAXJNT], jnt_limited[OM_MAXJNT];
  double jnt_pos[OM_MAXJNT][3], jnt_axis[OM_MAXJNT][3], jnt_range[OM_MAXJNT][2], jnt_margin[OM_MAXJNT];
  /* per-joint limit solver parameters (solreflimit / solimplimit; the scene builder fills the global defaults), joint spring
   * (stiffness, qpos_spring = springref) */
  double jnt_solref[OM_MAXJNT][2], jnt_solimp[OM_MAXJNT][5], jnt_stiffness[OM_MAXJNT], qpos_spring[OM_MAXQ];
  int dof_body[OM_MAXV], dof_jnt[OM_MAXV], dof_parent[OM_MAXV];
  double dof_armature[OM_MAXV], dof_damping[OM_MAXV], dof_invweight0[OM_MAXV];
  double qpos0[OM_MAXQ];
  /* geoms */
  int geom_type[OM_MAXGEOM], geom_body[OM_MAXGEOM], geom_contype[OM_MAXGEOM], geom_conaffinity[OM_MAXGEOM], geom_condim[OM_MAXGEOM];
  /* contact parameter mixing (mj_contactParam): the geom of higher priority gives condim / friction / solref / solimp, equal
   * priorities take the maxima and (solmix 1 : 1) the mean of solref / solimp */
  int geom_priority[OM_MAXGEOM];
  double geom_solref[OM_MAXGEOM][2], geom_solimp[OM_MAXGEOM][5];
  double geom_pos[OM_MAXGEOM][3], geom_quat[OM_MAXGEOM][4], geom_size[OM_MAXGEOM][3], geom_friction[OM_MAXGEOM][3];
  /* OM_GEOM_MESH: category of include/mjs_block_hulls.h (collided by its convex hull, as MuJoCo does) and mesh scale; the geom
   * frame sits at the mesh's centre of mass (MuJoCo re-centres a mesh geom there) */
  int geom_mesh[OM_MAXGEOM];
  double geom_mesh_scale[OM_MAXGEOM];
  /* sites */
  int site_body[OM_MAXSITE];
  double site_pos[OM_MAXSITE][3], site_quat[OM_MAXSITE][4];
  /* actuators: joint or fixed-tendon transmission (act_trntype, act_jnt = joint / tendon id), fixed gain, affine bias */
  int act_jnt[OM_MAXU], act_trntype[OM_MAXU], act_ctrllimited[OM_MAXU], act_forcelimited[OM_MAXU];
  /* fixed tendons: length = sum_j coef[j] qpos[dof j] (hinge / slide joints only) */
  int ntendon;
  double tendon_coef[OM_MAXTENDON][OM_MAXV];
  double act_gain[OM_MAXU], act_bias[OM_MAXU][3], act_ctrlrange[OM_MAXU][2], act_forcerange[OM_MAXU][2];
  /* equality: weld / connect (eq_body1/2 = bodies; connect data: anchor in body1, anchor in body2) and joint coupling
   * (eq_body1/2 = JOINT ids; data[0..4] = polycoef) */
  int eq_type[OM_MAXEQ], eq_body1[OM_MAXEQ], eq_body2[OM_MAXEQ];
  double eq_data[OM_MAXEQ][11], eq_solref[OM_MAXEQ][2], eq_solimp[OM_MAXEQ][5];
  /* reduced 2F-85 (Button-Push): the two finger-tip sphere geoms, re-placed from the driver angle in every substep (-1 none) */
  int gr_geom[2];
  /* touch sensor: site index (-1 none) and its cylinder size (radius, half-height) */
  int touch_site;
  double touch_size[2];
  /* options + statistics */
  double dt, gravity[3], tolerance, impratio, meaninertia;
  double solref[2], solimp[5]; /* global default contact/limit parameters */
  int iterations, integrator, cone;
} om_model;

typedef struct {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, exclude, efc_address;
  int zone; /* elliptic cone: 0 top (no force), 1 middle (cone surface), 2 bottom (fully quadratic): PrimalUpdateConstraint's state */
} om_contact;

typedef struct {
  /* state */
  double time, qpos[OM_MAXQ], qvel[OM_MAXV], ctrl[OM_MAXU], qacc_warmstart[OM_MAXV];
  double mocap_pos[OM_MAXMOCAP][3], mocap_quat[OM_MAXMOCAP][4];
  /* position-dependent */
  double xpos[OM_MAXBODY][3], xquat[OM_MAXBODY][4], xmat[OM_MAXBODY][9], xipos[OM_MAXBODY][3], ximat[OM_MAXBODY][9];
  double xanchor[OM_MAXJNT][3], xaxis[OM_MAXJNT][3];
  double geom_xpos[OM_MAXGEOM][3], geom_xmat[OM_MAXGEOM][9], site_xpos[OM_MAXSITE][3], site_xmat[OM_MAXSITE][9];
  double S[OM_MAXV][6];            /* motion subspace about the world origin: [ang, lin] */
  double M[OM_MAXV][OM_MAXV];      /* joint-space inertia (dense) */
  double Lm[OM_MAXV][OM_MAXV];     /* Cholesky factor of M */
  int ncon, nefc, ne, nl;
  om_contact contact[OM_MAXCON];
  double efc_J[OM_MAXEFC][OM_MAXV], efc_pos[OM_MAXEFC], efc_margin[OM_MAXEFC], efc_diagApprox[OM_MAXEFC];
  double efc_R[OM_MAXEFC], efc_D[OM_MAXEFC], efc_KBIP[OM_MAXEFC][4], efc_vel[OM_MAXEFC], efc_aref[OM_MAXEFC], efc_force[OM_MAXEFC];
  int efc_type[OM_MAXEFC], efc_id[OM_MAXEFC];
  /* velocity / force */
  double cvel[OM_MAXBODY][6];
  double qfrc_bias[OM_MAXV], qfrc_passive[OM_MAXV], qfrc_actuator[OM_MAXV], actuator_force[OM_MAXU];
  double qfrc_smooth[OM_MAXV], qacc_smooth[OM_MAXV], qacc[OM_MAXV], qfrc_constraint[OM_MAXV];
  int solver_niter, warning_bad;
  double touch_force; /* touch sensordata (mj_sensorAcc) */
} om_data;

/* numpy-compatible MT19937 (legacy RandomState(int seed)) */
typedef struct { uint32_t mt[624]; int pos; } om_rng;
void om_rng_seed(om_rng* r, uint32_t seed);
uint32_t om_rng_u32(om_rng* r);
double om_rng_double(om_rng* r);                       /* RandomState.random_sample() */
double om_rng_uniform(om_rng* r, double lo, double hi); /* RandomState.uniform(lo,hi) */

/* engine */
void om_set_const(om_model* m);                 /* mj_setConst: invweight0, meaninertia */
void om_reset_data(const om_model* m, om_data* d);
void om_forward(const om_model* m, om_data* d); /* mj_forward */
void om_step1(const om_model* m, om_data* d);   /* mj_step1 */
void om_step2(const om_model* m, om_data* d);   /* mj_step2 */
void om_physics_step(const om_model* m, om_data* d); /* dm_control legacy step: step2; step1 */

/* UR analytic IK (restating third-party ur_analytic_ik; call site robot.py:33-37) */
void om_ur5e_fk_dh(const double q[6], double T[16]);
int om_ur5e_ik_all(const double T[16], double sols[8][6]);
int om_ur5e_ik_closest(const double T[16], const double q_guess[6], double q_out[6]);

/* tasks */
enum { OM_TASK_POINTMASS = 0, OM_TASK_ROBOT_REACH = 1, OM_TASK_PLANAR_PUSH = 2, OM_TASK_BUTTON_PUSH = 3 };
enum { OM_STEP_FIRST = 0, OM_STEP_MID = 1, OM_STEP_LAST = 2 };
/* Pointmass reward types (point_reach.py:11-14) / Robot-Reach (robot_reach.py:37-38) */
enum { OM_REW_SPARSE = 0, OM_REW_DENSE_POTENTIAL = 1, OM_REW_DENSE_NEG_DISTANCE = 2, OM_REW_DENSE_BIASED_NEG_DISTANCE = 3 };
enum { OM_AUTORESET_NEXT_STEP = 0, OM_AUTORESET_SAME_STEP = 1, OM_AUTORESET_DISABLED = 2 };

enum { OM_ACTION_ABS_JOINT = 0, OM_ACTION_ABS_EEF = 1 }; /* robot_push_button.py:28-29 */
typedef struct {
  int task, reward_type, autoreset;
  double time_limit;         /* composer.Environment(time_limit=...) */
  int terminate_on_success;  /* Robot-Reach only: opt-in (deviation D-2) */
  int action_type;           /* Button-Push only */
  int button_disturbances;   /* Button-Push only: robot_push_button.py:159-165 */
  int n_objects;             /* Planar-Push only: 1..MJS_PP_MAX_OBJECTS blocks (<= 0: 2) */
  int max_episode_steps;     /* Planar-Push only: RobotTask step limit (base.py:47-51), default 500 */
  int block_shape;           /* Planar-Push only: OM_BLOCKS_MESH (reference: google_block.py, category / colour / scale drawn per
                              * episode from the env's seeded stream, deviation D-5) or OM_BLOCKS_BOX (round 1's stand-in) */
  int gripper_model;         /* Button-Push only: OM_GRIPPER_REDUCED (D-1b: one driver coordinate + two tip spheres, nv = 6) or
                              * OM_GRIPPER_ARTICULATED (the 2F-85's eight hinges, two connects, the driver coupling, the
                              * fixed-tendon actuator, pad boxes, elliptic cones with impratio 10; nv = 14) */
} om_task_config;
enum { OM_BLOCKS_MESH = 0, OM_BLOCKS_BOX = 1 };
enum { OM_GRIPPER_REDUCED = 0, OM_GRIPPER_ARTICULATED = 1 };

#define OM_MAXOBS 16
typedef struct {
  double obs[OM_MAXOBS];          /* task-specific flat layout, see om_obs_dim() */
  double terminal_obs[OM_MAXOBS]; /* same-step autoreset: observation of the LAST step */
  double reward, discount;
  int step_type, terminated, truncated, is_success, ncon, fault, ik_failed;
} om_step_out;

typedef struct {
  om_task_config cfg;
  om_model m;
  om_data d;
  om_rng rng;
  int reset_pending, n_sub;
  /* Pointmass bookkeeping (point_reach.py:112-113,165-170): NOT reset per episode */
  double distance_to_target, previous_distance_to_target;
  double target_pos[3];
  /* servo trajectory (robot.py:227-259, joint_trajectory.py:41-47) */
  int traj_active;
  double traj_q0[6], traj_q1[6], traj_t0, traj_t1;
  int ik_failed;
  /* Switch entity state (entities/props/switch.py:10-16,51-60) */
  int switch_active, switch_pressed, switch_num_pressed;
  double switch_pos[3];
  /* reduced Robotiq 2F-85 (gripper.py:36-98): driver angle of the two equality-coupled fingers, its
   * velocity, and the fingers_actuator ctrl that Robotiq2f85.move set in before_step */
  double gr_theta, gr_vel, gr_ctrl;
  /* Planar-Push: RobotTask.episode_step (base.py:29-32) */
  int episode_step;
  /* Planar-Push: the blocks of this episode (GoogleBlockProp.sample_random_object, google_block.py:55-68) */
  int block_cat[5], block_color[5];
  double block_scale[5];
  int dbg_arm_floor_seen; /* test knob, om_debug_arm_floor_seen() */
} om_env;

void om_default_config(int task, om_task_config* cfg);
int om_obs_dim(int task);
int om_obs_dim_for(const om_task_config* cfg); /* Planar-Push: 5 + 2 * block slots of cfg->n_objects */
int om_action_dim(int task);
void om_env_init(om_env* e, const om_task_config* cfg, uint32_t seed);
void om_env_seed(om_env* e, uint32_t seed);
void om_env_reset(om_env* e, om_step_out* out);
void om_env_step(om_env* e, const double* action, om_step_out* out);

/* scene-camera image (own ray caster, om_render.c): out uint8 [H, W, 3] */
void om_render_pointmass(const om_env* e, int H, int W, uint8_t* out);
void om_render_robot(const om_env* e, int H, int W, uint8_t* out);
/* camera 0 = the task's scene camera, 1 = Button-Push wrist camera */
void om_render_camera(const om_env* e, int camera, int H, int W, uint8_t* out);
void om_debug_button_dynamics(const double* q, const double* v, double* M_out, double* bias_out, double* invw_out);
void om_debug_set_robot_state(om_env* e, const double* q, const double* v);
void om_debug_link_invweights(int task, double* out7);
int om_debug_get_state(const om_env* e, double* qpos, double* qvel, double* time);
int om_debug_arm_floor_seen(om_env* e);
void om_debug_geom_shape(const om_env* e, int g, int* type_body /*2*/, double* size3);
void om_debug_set_block_shape(om_env* e, int i, int cat, int color, double scale);
void om_debug_get_block_shape(const om_env* e, int* cat, int* color, double* scale);
void om_debug_set_state(om_env* e, const double* qpos, const double* qvel);
/* reduced gripper of Button-Push: driver angle and velocity (set: re-places the finger tips, then mj_forward) */
void om_debug_get_gripper(const om_env* e, double* theta_vel);
void om_debug_set_gripper(om_env* e, double theta, double vel);
void om_debug_substeps(om_env* e, int n);
void om_debug_model_dims(const om_env* e, int* out8);
int om_debug_efc(const om_env* e, int maxrows, double* pos, double* J, int* type, double* force, double* aref, double* D);
void om_debug_geom_pose(const om_env* e, int g, double* pos3, double* mat9);
void om_debug_set_ctrl(om_env* e, int u, double value);
void om_debug_get_dynamics(const om_env* e, double* M, double* qfrc_smooth, double* qacc);
int om_debug_convex(int type1, const double* size1, const double* pos1, const double* mat1, int type2, const double* size2, const double* pos2,
                    const double* mat2, double* out);
void om_debug_reach_dynamics(const double* q, const double* v, double* M_out, double* bias_out);

/* ---- the Robot entity's control API on a stand-alone UR5e (entities/robots/robot.py:113-272), the component the
 * reference's own tests exercise (test/test_ur_control_api.py:7-82). State block (OM_UR_STATE doubles per robot, the same
 * layout as the device entry point mjs_ur5e_robot_run): q[6], v[6], ctrl[6], time, traj_active, traj_q0[6], traj_q1[6],
 * traj_t0, traj_t1. */
#define OM_UR_STATE 34
enum { OM_UR_CMD_NONE = 0, OM_UR_CMD_MOVEJ = 1, OM_UR_CMD_MOVEJ_IK = 2, OM_UR_CMD_SERVOL = 3, OM_UR_CMD_SERVOJ = 4 };
enum { OM_UR_EEF_NONE = 0, OM_UR_EEF_GRIPPER = 1 };
void om_build_ur5e_alone(om_model* m, int eef_gripper, double dt);
/* applies `command` (target: 6 joints or a TCP pose xyz + scalar-last quaternion; param: speed [rad/s] for moveJ /
 * movej_IK, duration [s] for servoL / servoJ), then n_substeps x (Robot.before_substep; Physics.step), and reports
 * Robot.get_tcp_pose (7 numbers). Returns 0 when the command's IK found no solution (movej_IK prints and returns,
 * servoL raises: robot.py:206-209,221-224), else 1. */
int om_ur_robot_run(double* state, const double* target, int command, double param, int n_substeps, int eef, double dt, double* tcp_pose_out);
void om_rotation_to_quat_xyzw(const double* R /*row-major 3x3*/, double* quat_xyzw); /* SE3Container.orientation_as_quaternion */

/* batch helpers for the CPU baseline / parity tests (OpenMP over envs) */
typedef struct om_batch om_batch;
om_batch* om_batch_create(const om_task_config* cfg, int n, uint32_t base_seed);
void om_batch_destroy(om_batch* b);
om_env* om_batch_env(om_batch* b, int i);
void om_batch_reset(om_batch* b, om_step_out* outs);
void om_batch_step(om_batch* b, const double* actions /*[n,A]*/, om_step_out* outs, int nthreads);
int om_sizeof_step_out(void);

#ifdef __cplusplus
}
#endif
#endif
