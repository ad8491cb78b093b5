/* mjs_scene_spec.h — scene constants (DATA ONLY) for the four task scenes.
 *
 * Own-authored. Two provenances, marked per block:
 *   [REF]  derived from the reference's scene-building code, cited file:line
 *          (paths relative to /root/reference/mujoco_sim/).
 *   [MEN]  re-authored from the public mujoco_menagerie UR5e / Robotiq 2F-85
 *          descriptions, which the reference downloads at run time
 *          (entities/robots/robot.py:311-313, entities/eef/gripper.py:5,37) and
 *          which are NOT present under /root/reference. Unverified recollection:
 *          "parity unpinned" for every robot scene.
 *   [MJ]   MuJoCo engine defaults (third-party, recalled).
 *
 * Consumed by oracle/ (CPU restatement, test infrastructure) and by
 * mujoco_sim_amd/csrc (HIP kernels). Quaternions here are MuJoCo order (w,x,y,z).
 */
#ifndef MJS_SCENE_SPEC_H
#define MJS_SCENE_SPEC_H

#ifdef __cplusplus
#define MJS_K constexpr
#else
#define MJS_K static const
#endif

/* ------------------------------------------------------------------ [MJ] */
MJS_K double MJS_GRAVITY_Z = -9.81;
MJS_K double MJS_MINVAL = 1e-15;               /* mjMINVAL */
MJS_K double MJS_MAXVAL = 1e10;                /* mjMAXVAL: bad-state threshold */
MJS_K double MJS_SOLREF_TIMECONST = 0.02;      /* default solref[0] */
MJS_K double MJS_SOLREF_DAMPRATIO = 1.0;       /* default solref[1] */
MJS_K double MJS_SOLIMP_D0 = 0.9;              /* default solimp */
MJS_K double MJS_SOLIMP_DWIDTH = 0.95;
MJS_K double MJS_SOLIMP_WIDTH = 0.001;
MJS_K double MJS_SOLIMP_MIDPOINT = 0.5;
MJS_K double MJS_SOLIMP_POWER = 2.0;
MJS_K double MJS_SOLVER_TOLERANCE = 1e-8;      /* opt.tolerance */
MJS_K int    MJS_SOLVER_ITERATIONS = 100;      /* opt.iterations */
MJS_K double MJS_GEOM_FRICTION_SLIDE = 1.0;    /* default geom friction */
MJS_K double MJS_GEOM_FRICTION_SPIN = 0.005;
MJS_K double MJS_GEOM_FRICTION_ROLL = 0.0001;
MJS_K double MJS_GEOM_DENSITY = 1000.0;

/* ------------------------------------------------- Pointmass-Reach [REF] */
/* environments/tasks/point_reach.py:24-28 */
MJS_K double MJS_PM_PHYSICS_DT = 0.02;
MJS_K double MJS_PM_CONTROL_DT = 0.1;
MJS_K int    MJS_PM_NSUB = 5;                  /* round(0.1/0.02) */
MJS_K int    MJS_PM_MAX_CONTROL_STEPS = 50;
MJS_K double MJS_PM_GOAL_THRESHOLD = 0.02;
MJS_K double MJS_PM_MAX_STEP_SIZE = 0.05;
/* entities/pointmass.py:52-55, point_reach.py:80 */
MJS_K double MJS_PM_RADIUS = 0.05;
MJS_K double MJS_PM_MASS = 0.1;
/* entities/arenas/walled_pointmass_arena.py:9-10, mjcf/walled_pointmass_arena.xml:15-19 */
MJS_K double MJS_PM_ARENA_LO = -0.5;
MJS_K double MJS_PM_ARENA_HI = 0.5;
MJS_K double MJS_PM_WALL_Z = 0.02;
/* point_reach.py:91-93 (target site z set to radius/2 in initialize_episode :136) */
MJS_K double MJS_PM_TARGET_DEFAULT_POS[3] = {0.25, 0.25, 0.01};
/* camera geoms point_reach.py:22, entities/camera.py:78-88 (static, never in reach) */
MJS_K double MJS_PM_CAMERA_POS[3] = {0.0, 0.0, 2.4};

/* ------------------------------------------------------ UR5e arm [MEN] */
#define MJS_UR_NJ 6
/* body frame offsets (parent frame), MuJoCo quat order (w,x,y,z), un-normalised
 * as in the XML; bodies: base, shoulder, upper_arm, forearm, wrist_1..3 */
#define MJS_UR_NBODY 7
MJS_K double MJS_UR_BODY_POS[MJS_UR_NBODY][3] = {
    {0.0, 0.0, 0.0},      /* base (robot_site at world origin, empty_robot_arena.py:22) */
    {0.0, 0.0, 0.163},    /* shoulder_link  */
    {0.0, 0.138, 0.0},    /* upper_arm_link */
    {0.0, -0.131, 0.425}, /* forearm_link   */
    {0.0, 0.0, 0.392},    /* wrist_1_link   */
    {0.0, 0.127, 0.0},    /* wrist_2_link   */
    {0.0, 0.0, 0.1},      /* wrist_3_link   */
};
MJS_K double MJS_UR_BODY_QUAT[MJS_UR_NBODY][4] = {
    {0.0, 0.0, 0.0, -1.0}, /* [REF] entities/robots/robot.py:320 */
    {1.0, 0.0, 0.0, 0.0},
    {1.0, 0.0, 1.0, 0.0},
    {1.0, 0.0, 0.0, 0.0},
    {1.0, 0.0, 1.0, 0.0},
    {1.0, 0.0, 0.0, 0.0},
    {1.0, 0.0, 0.0, 0.0},
};
MJS_K double MJS_UR_BODY_MASS[MJS_UR_NBODY] = {4.0, 3.7, 8.393, 2.275, 1.219, 1.219, 0.1889};
MJS_K double MJS_UR_BODY_IPOS[MJS_UR_NBODY][3] = {
    {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.2125}, {0.0, 0.0, 0.196},
    {0.0, 0.127, 0.0}, {0.0, 0.0, 0.1}, {0.0, 0.0771683, 0.0},
};
MJS_K double MJS_UR_BODY_IQUAT[MJS_UR_NBODY][4] = {
    {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0},
    {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 1},
};
MJS_K double MJS_UR_BODY_DIAGINERTIA[MJS_UR_NBODY][3] = {
    {0.00443333156, 0.00443333156, 0.0072},
    {0.0102675, 0.0102675, 0.00666},
    {0.133886, 0.133886, 0.0151074},
    {0.0311796, 0.0311796, 0.004095},
    {0.0025599, 0.0025599, 0.0021942},
    {0.0025599, 0.0025599, 0.0021942},
    {0.000132134, 9.90863e-05, 9.90863e-05},
};
/* joint j lives on body j+1; hinge axes in the body frame; joint pos = body origin */
MJS_K double MJS_UR_JNT_AXIS[MJS_UR_NJ][3] = {
    {0, 0, 1}, {0, 1, 0}, {0, 1, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 0},
};
MJS_K double MJS_UR_JNT_RANGE[MJS_UR_NJ][2] = {
    {-6.28319, 6.28319}, {-6.28319, 6.28319}, {-3.1415, 3.1415},
    {-6.28319, 6.28319}, {-6.28319, 6.28319}, {-6.28319, 6.28319},
};
MJS_K double MJS_UR_ARMATURE = 0.1;
/* position servos: force = kp*ctrl - kp*q - kd*qdot, clamped to +-frc
 * (general actuator, gaintype fixed, biastype affine) */
MJS_K double MJS_UR_ACT_KP[MJS_UR_NJ] = {2000, 2000, 2000, 500, 500, 500};
MJS_K double MJS_UR_ACT_KD[MJS_UR_NJ] = {400, 400, 400, 100, 100, 100};
MJS_K double MJS_UR_ACT_FRC[MJS_UR_NJ] = {150, 150, 150, 28, 28, 28};
MJS_K double MJS_UR_ACT_CTRLRANGE[MJS_UR_NJ][2] = {
    {-6.2831, 6.2831}, {-6.2831, 6.2831}, {-3.1415, 3.1415},
    {-6.2831, 6.2831}, {-6.2831, 6.2831}, {-6.2831, 6.2831},
};
/* flange site on wrist_3 ("attachment_site", [REF] robot.py:304-305 names it) */
MJS_K double MJS_UR_FLANGE_POS[3] = {0.0, 0.1, 0.0};
MJS_K double MJS_UR_FLANGE_QUAT[4] = {-1.0, 1.0, 0.0, 0.0};
/* [REF] robot.py:307 */
MJS_K double MJS_UR_HOME_Q[MJS_UR_NJ] = {-1.5707963267948966, -1.5707963267948966, 1.5707963267948966,
                                         -1.5707963267948966, -1.5707963267948966, -1.5707963267948966};
/* arm collision proxies (capsules; last one a cylinder): body index, local pos,
 * local quat, radius, half-length. Used for contact DETECTION vs the floor. */
#define MJS_UR_NCOLGEOM 10
MJS_K int MJS_UR_COL_BODY[MJS_UR_NCOLGEOM] = {1, 2, 2, 3, 3, 4, 5, 5, 6, 6};
MJS_K int MJS_UR_COL_TYPE[MJS_UR_NCOLGEOM] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 5}; /* 3 capsule, 5 cylinder */
MJS_K double MJS_UR_COL_POS[MJS_UR_NCOLGEOM][3] = {
    {0, 0, -0.04}, {0, -0.04, 0}, {0, 0, 0.2}, {0, 0.08, 0}, {0, 0, 0.2},
    {0, 0.05, 0},  {0, 0, 0.04},  {0, 0.02, 0.1}, {0, 0.08, 0}, {0, 0.08, 0},
};
MJS_K double MJS_UR_COL_QUAT[MJS_UR_NCOLGEOM][4] = {
    {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 0, 0, 0},
    {1, 1, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {1, 1, 0, 0}, {1, 1, 0, 0},
};
MJS_K double MJS_UR_COL_SIZE[MJS_UR_NCOLGEOM][2] = {
    {0.06, 0.06}, {0.06, 0.06}, {0.05, 0.2}, {0.055, 0.06}, {0.038, 0.19},
    {0.04, 0.07}, {0.04, 0.06}, {0.04, 0.04}, {0.04, 0.02}, {0.04, 0.02},
};

/* analytic-IK kinematic constants: real UR5e DH (inside third-party
 * ur_analytic_ik, call site [REF] robot.py:33-37). Differ from the model's own
 * rounded offsets by ~1 mm; the reference tolerates 1e-2
 * (test/test_ur_frame_matches_real.py:29). */
MJS_K double MJS_UR_DH_D1 = 0.1625;
MJS_K double MJS_UR_DH_A2 = -0.425;
MJS_K double MJS_UR_DH_A3 = -0.3922;
MJS_K double MJS_UR_DH_D4 = 0.1333;
MJS_K double MJS_UR_DH_D5 = 0.0997;
MJS_K double MJS_UR_DH_D6 = 0.0996;

/* ---------------------------------- end-effectors as attached bodies */
/* Robotiq 2F-85 lumped into ONE rigid payload body at the flange (deviation D-1,
 * SURVEY.md §8): the reference attaches the articulated 8-DoF gripper
 * (robot_reach.py:92-94). Mass/inertia [MEN, approximate]; gravcomp 0 because
 * gravcomp is set on the arm bodies before the EEF is attached (robot.py:80-82). */
MJS_K double MJS_G2F85_MASS = 0.925;
MJS_K double MJS_G2F85_IPOS[3] = {0.0, 0.0, 0.045};
MJS_K double MJS_G2F85_DIAGINERTIA[3] = {0.0011, 0.0009, 0.0005};
MJS_K double MJS_G2F85_TCP_Z = 0.174;          /* [REF] gripper.py:46-48 */
MJS_K double MJS_G2F85_OPEN = 0.085;           /* [REF] gripper.py:50-52 */
MJS_K double MJS_G2F85_MAX_DRIVER = 0.8;       /* [REF] gripper.py:38 */
/* [MEN] robotiq 2f85.xml (absent package robot_descriptions): `fingers_actuator` = general actuator on the fixed tendon
 * "split" (0.5 right_driver_joint + 0.5 left_driver_joint): gainprm 0.3137255 (= 80 / 255), biasprm 0 -100 -10, ctrlrange
 * 0 255, forcerange -5 5; driver joints: armature 0.005, damping 0.1, range 0 0.8. The reduced gripper keeps
 * the driver angle of the two equality-coupled fingers as its one coordinate: inertia 2 x armature, damping 2 x damping. */
MJS_K double MJS_G2F85_ACT_GAIN = 0.3137255;
MJS_K double MJS_G2F85_ACT_KP = 100.0;
MJS_K double MJS_G2F85_ACT_KV = 10.0;
MJS_K double MJS_G2F85_ACT_FORCE = 5.0;
MJS_K double MJS_G2F85_CTRL_MAX = 255.0;
MJS_K double MJS_G2F85_DRIVER_ARMATURE = 0.005;
MJS_K double MJS_G2F85_DRIVER_DAMPING = 0.1;
/* ---- Robotiq 2F-85, ARTICULATED [MEN]: robotiq_2f85/2f85.xml of mujoco_menagerie as recalled (the package
 * robot_descriptions that the reference loads it from, gripper.py:5,37, is absent: unverified, parity unpinned). What the
 * reference itself holds agrees with it: 8 hinge joints in the order of gripper.py:40 (right driver, coupler, spring_link,
 * follower, then left), the driver range 0..0.8 (gripper.py:38), an 85 mm stroke (gripper.py:50-52: the pads' inner faces are
 * 2 x 42.7 mm apart at q = 0 and 0.2 mm at 0.8 rad in this geometry), the actuator `fingers_actuator` with ctrl 0..255
 * (gripper.py:58-60,81-84), TCP = the finger tips when CLOSED (gripper.py:46-48: 0.174 m; pads end 0.160 m from the flange
 * when open, 0.174 m when closed).
 * Bodies in MJCF (depth-first) order; parent -1 = the attachment frame (dm_control `attach` puts the gripper's worldbody in
 * a body at the flange site). Every joint is a hinge about the body's local x (default class "2f85": axis 1 0 0). The
 * mesh collision geoms of base_mount / base / driver / coupler / spring_link / follower are NOT modelled (the meshes are not
 * in the reference): the pads' boxes are the gripper's only collision geoms. base_mount has no <inertial>
 * in the MJCF (MuJoCo derives it from the mesh): the value here makes the gripper weigh 0.925 kg like the lump of D-1. */
#define MJS_G85_NBODY 12
#define MJS_G85_NJ 8
MJS_K int MJS_G85_PARENT[MJS_G85_NBODY] = {-1, 0, 1, 2, 1, 4, 5, 1, 7, 1, 9, 10};
MJS_K double MJS_G85_POS[MJS_G85_NBODY][3] = {
    {0, 0, 0.007}, {0, 0, 0.0038},
    {0, 0.0306011, 0.054904}, {0, 0.0315, -0.0041}, {0, 0.0132, 0.0609}, {0, 0.055, 0.0375}, {0, -0.0189, 0.01352},
    {0, -0.0306011, 0.054904}, {0, 0.0315, -0.0041}, {0, -0.0132, 0.0609}, {0, 0.055, 0.0375}, {0, -0.0189, 0.01352}};
MJS_K double MJS_G85_QUAT[MJS_G85_NBODY][4] = {
    {1, 0, 0, 0}, {1, 0, 0, -1},
    {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0},
    {0, 0, 0, 1}, {1, 0, 0, 0}, {0, 0, 0, 1}, {1, 0, 0, 0}, {1, 0, 0, 0}};
MJS_K double MJS_G85_MASS[MJS_G85_NBODY] = {0.02500014, 0.777441, 0.00899563, 0.0140974, 0.0221642, 0.0125222, 0.0035,
                                            0.00899563, 0.0140974, 0.0221642, 0.0125222, 0.0035};
MJS_K double MJS_G85_IPOS[MJS_G85_NBODY][3] = {
    {0, 0, 0.002}, {0, -2.70394e-05, 0.0354675},
    {2.96931e-12, 0.0177547, 0.00107314}, {0, 0.00301209, 0.0232175}, {-8.65005e-09, 0.0181624, 0.0212658}, {0, -0.011046, 0.0124786}, {0, -0.0025, 0.0185},
    {2.96931e-12, 0.0177547, 0.00107314}, {0, 0.00301209, 0.0232175}, {-8.65005e-09, 0.0181624, 0.0212658}, {0, -0.011046, 0.0124786}, {0, -0.0025, 0.0185}};
MJS_K double MJS_G85_IQUAT[MJS_G85_NBODY][4] = {
    {1, 0, 0, 0}, {1, -0.00152849, 0, 0},
    {0.681301, 0.732003, 0, 0}, {0.705636, -0.0455904, 0.0455904, 0.705636}, {0.663403, -0.244737, 0.244737, 0.663403}, {1, 0.1664, 0, 0}, {0.707107, 0, 0, 0.707107},
    {0.681301, 0.732003, 0, 0}, {0.705636, -0.0455904, 0.0455904, 0.705636}, {0.663403, -0.244737, 0.244737, 0.663403}, {1, 0.1664, 0, 0}, {0.707107, 0, 0, 0.707107}};
MJS_K double MJS_G85_DIAGINERTIA[MJS_G85_NBODY][3] = {
    {6.0e-06, 6.0e-06, 1.1e-05}, {0.000260285, 0.000225381, 0.000152708},
    {1.72352e-06, 1.60906e-06, 3.22006e-07}, {4.16206e-06, 3.52216e-06, 8.88131e-07}, {8.96853e-06, 6.71733e-06, 2.63931e-06}, {2.67415e-06, 2.4559e-06, 6.02031e-07}, {4.73958e-07, 3.64583e-07, 1.23958e-07},
    {1.72352e-06, 1.60906e-06, 3.22006e-07}, {4.16206e-06, 3.52216e-06, 8.88131e-07}, {8.96853e-06, 6.71733e-06, 2.63931e-06}, {2.67415e-06, 2.4559e-06, 6.02031e-07}, {4.73958e-07, 3.64583e-07, 1.23958e-07}};
/* joint of a body: -1 none, else the class 0 driver, 1 coupler, 2 spring_link, 3 follower */
MJS_K int MJS_G85_JCLASS[MJS_G85_NBODY] = {-1, -1, 0, 1, 2, 3, -1, 0, 1, 2, 3, -1};
MJS_K double MJS_G85_JRANGE[4][2] = {{0, 0.8}, {-1.57, 0}, {-0.29670597283, 0.8}, {-0.872664, 0.872664}};
MJS_K double MJS_G85_JARMATURE[4] = {0.005, 0.001, 0.001, 0.001};
MJS_K double MJS_G85_JDAMPING[4] = {0.1, 0, 0.00125, 0};
MJS_K double MJS_G85_JSTIFFNESS[4] = {0, 0, 0.05, 0};
MJS_K double MJS_G85_JSPRINGREF[4] = {0, 0, 2.62, 0};
MJS_K double MJS_G85_JPOS[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, -0.018, 0.0065}};
/* solreflimit / solimplimit of the driver, coupler and follower classes; the spring_link class keeps MuJoCo's defaults. The
 * two connects and the driver coupling use the same pair. */
MJS_K int    MJS_G85_JSTIFFLIMIT[4] = {1, 1, 0, 1};
MJS_K double MJS_G85_SOLREF[2] = {0.005, 1};
MJS_K double MJS_G85_SOLIMP[5] = {0.95, 0.99, 0.001, 0.5, 2};
/* pad boxes (classes pad_box1 / pad_box2) on the two pad bodies: mass 0, priority 1, condim 3 */
MJS_K double MJS_G85_PAD_SIZE[3] = {0.011, 0.004, 0.009375};
MJS_K double MJS_G85_PAD_POS[2][3] = {{0, -0.0026, 0.028125}, {0, -0.0026, 0.009375}};
MJS_K double MJS_G85_PAD_FRICTION[2] = {0.7, 0.6};
MJS_K double MJS_G85_PAD_SOLREF[2] = {0.004, 1};
MJS_K double MJS_G85_PAD_SOLIMP[5] = {0.95, 0.99, 0.001, 0.5, 2};
/* table indices */
MJS_K int MJS_G85_B_RIGHT_COUPLER = 3, MJS_G85_B_RIGHT_FOLLOWER = 5, MJS_G85_B_RIGHT_PAD = 6;
MJS_K int MJS_G85_B_LEFT_COUPLER = 8, MJS_G85_B_LEFT_FOLLOWER = 10, MJS_G85_B_LEFT_PAD = 11;
MJS_K int MJS_G85_J_RIGHT_DRIVER = 0, MJS_G85_J_LEFT_DRIVER = 4;
/* <option cone="elliptic" impratio="10"/> of the gripper's MJCF merges into the scene's options on attach */
MJS_K double MJS_G85_IMPRATIO = 10.0;
MJS_K double MJS_G85_TENDON_COEF = 0.5;        /* fixed tendon "split": 0.5 right_driver_joint + 0.5 left_driver_joint */
/* cylinder EEF [REF] entities/eef/cylinder.py:17-41 */
MJS_K double MJS_CYL_RADIUS = 0.02;
MJS_K double MJS_CYL_HALFLEN = 0.05;
MJS_K double MJS_CYL_MASS = 0.1;
MJS_K double MJS_CYL_POS_Z = 0.051;
MJS_K double MJS_CYL_TCP_Z = 0.1;

/* ------------------------------------------------ Robot-Reach task [REF] */
/* environments/tasks/robot_reach.py:30,59-68,104,108-110 */
MJS_K double MJS_RR_PHYSICS_DT = 0.005;
MJS_K double MJS_RR_CONTROL_DT = 0.1;
MJS_K int    MJS_RR_NSUB = 20;
MJS_K int    MJS_RR_MAX_CONTROL_STEPS = 100;
MJS_K double MJS_RR_GOAL_THRESHOLD = 0.02;
MJS_K double MJS_RR_SPACE_LO[3] = {-0.1, -0.6, 0.02};
MJS_K double MJS_RR_SPACE_HI[3] = {0.1, -0.4, 0.2};
MJS_K double MJS_RR_TARGET_DEFAULT_POS[3] = {0.0, -0.5, 0.001};
/* TOP_DOWN_QUATERNION, scalar-LAST (x,y,z,w) per type_aliases.py:6-10 */
MJS_K double MJS_TOP_DOWN_QUAT_XYZW[4] = {1.0, 0.0, 0.0, 0.0};
MJS_K double MJS_ROBOT_ARENA_HALF = 1.5;       /* EmptyRobotArena(3), empty_robot_arena.py:18-20 */

/* ------------------------------------------------ Robot Button-Push [REF] */
/* environments/tasks/robot_push_button.py:35-43,76-79,108 */
MJS_K double MJS_BP_GOAL_THRESHOLD = 0.05;
MJS_K int    MJS_BP_MAX_CONTROL_STEPS = 100;
MJS_K double MJS_BP_ROBOT_SPACE_LO[3] = {-0.2, -0.6, 0.02};
MJS_K double MJS_BP_ROBOT_SPACE_HI[3] = {0.2, -0.3, 0.3};
MJS_K double MJS_BP_SWITCH_SPACE_LO[3] = {-0.2, -0.6, 0.0};
MJS_K double MJS_BP_SWITCH_SPACE_HI[3] = {0.2, -0.3, 0.1};
MJS_K double MJS_BP_ROBOT_END_POS[3] = {-0.3, -0.2, 0.3};
/* entities/props/switch.py:10-41,51-60,86-87 */
MJS_K double MJS_SW_BOX_HALF = 0.025;          /* box 0.05^3 centred at z = 0.025 */
MJS_K double MJS_SW_BUTTON_RADIUS = 0.02;      /* cylinder size [0.02, 0.02, 0.01]: radius, half-height */
MJS_K double MJS_SW_BUTTON_HALF = 0.02;
MJS_K double MJS_SW_BUTTON_Z = 0.05;
MJS_K double MJS_SW_SITE_SCALE = 1.01;
MJS_K double MJS_SW_MIN_FORCE = 5.0;
MJS_K double MJS_SW_MAX_FORCE = 200.0;
MJS_K double MJS_SW_POSITION_OFFSET = 0.01;    /* get_position: xpos + 0.5*size[1], added to all 3 coordinates */
/* wrist camera entity attached at the flange (robot_push_button.py:53-54,90-96; entities/camera.py:78-88):
 * a box (half 0.045,0.0125,0.0125) and a sphere (r 0.0125) at `pos` in the flange frame, default density */
MJS_K double MJS_WCAM_POS[3] = {0.0, 0.05, 0.0};
MJS_K double MJS_WCAM_QUAT[4] = {0.0, 0.0, 0.999, 0.04};
MJS_K double MJS_WCAM_FOVY = 42.0;
MJS_K double MJS_CAM_BOX_HALF[3] = {0.045, 0.0125, 0.0125};
MJS_K double MJS_CAM_SPHERE_RADIUS = 0.0125;
MJS_K double MJS_BP_CAM_POS[3] = {0.0, -1.7, 0.7};             /* scene camera, robot_push_button.py:51-52 */
MJS_K double MJS_BP_CAM_QUAT[4] = {-0.7, -0.35, 0.0, 0.0};
MJS_K double MJS_BP_CAM_FOVY = 70.0;
/* ------------------------------------------------------------ Planar-Push (a13) [REF] tasks/robot_planar_push.py
 * :29-73 config, :81-117 scene, :149-176 reset; entities/eef/cylinder.py:23-33; entities/props/google_block.py:37-49;
 * mjcf/google_language_table_blocks/cube.{xml,obj}. The block is a bevelled-cube MESH in the reference (bounding box
 * x,z in +-0.019826, y in [0, 0.0381], geom quat (1,1,0,0): mesh y -> body z, body origin = centre of the bottom face);
 * here it is a BOX of that bounding box (deviation D-9). */
MJS_K int    MJS_PP_MAX_OBJECTS = 5;              /* :61 reference default n_objects = 5 */
MJS_K int    MJS_PP_FAST_OBJECTS = 2;             /* :315 registered env / BASELINE config 4: 2 objects. Engines keep two
                                                   * layouts: block slots = 2 for n_objects <= 2, 5 otherwise */
#define MJS_PP_OBJECT_SLOTS(n) ((n) <= MJS_PP_FAST_OBJECTS ? MJS_PP_FAST_OBJECTS : MJS_PP_MAX_OBJECTS)
MJS_K int    MJS_PP_MAX_CONTROL_STEPS = 500;      /* :53 */
MJS_K double MJS_PP_TARGET_RADIUS = 0.05;         /* :60 */
MJS_K double MJS_PP_NEAREST_COEF = 0.1;           /* :59 */
MJS_K double MJS_PP_REWARD_SCALE = 0.1;           /* :220 */
MJS_K double MJS_PP_ACTION_Z = 0.02;              /* :199 */
MJS_K int    MJS_PP_SETTLE_STEPS = 150;           /* :160-161 */
MJS_K double MJS_PP_ROBOT_SPACE_LO[3] = {-0.2, -0.6, 0.02};   /* :103 */
MJS_K double MJS_PP_ROBOT_SPACE_HI[3] = {0.2, -0.3, 0.02};
MJS_K double MJS_PP_OBJECT_SPACE_LO[3] = {-0.15, -0.55, 0.05}; /* :104 */
MJS_K double MJS_PP_OBJECT_SPACE_HI[3] = {0.15, -0.35, 0.2};
MJS_K double MJS_PP_TARGET_SPACE_LO[3] = {-0.15, -0.55, 0.001}; /* :105 */
MJS_K double MJS_PP_TARGET_SPACE_HI[3] = {0.15, -0.35, 0.005};
MJS_K double MJS_PP_TARGET_DEFAULT_POS[3] = {0.0, -0.5, 0.001};
/* CylinderEEF: MJS_CYL_* above */
/* block stand-in: box half extents, mass (google_block.py:33), geom centre above the body origin, contact parameters
 * (google_block.py:47-49: condim 4, friction (1, 0.05, 0)) */
MJS_K double MJS_BLOCK_HALF[3] = {0.019826, 0.019826, 0.01905};
MJS_K double MJS_BLOCK_MASS = 0.1;                 /* google_block.py:30 */
MJS_K double MJS_BLOCK_SCALE_LO = 0.8;             /* google_block.py:59: scale_range */
MJS_K double MJS_BLOCK_SCALE_HI = 1.2;
MJS_K double MJS_BLOCK_GEOM_Z = 0.01905;
MJS_K int    MJS_BLOCK_CONDIM = 4;
MJS_K double MJS_BLOCK_FRICTION[3] = {1.0, 0.05, 0.0};
/* own MPR (Minkowski portal refinement) parameters for convex-convex pairs (cylinder-box, box-box) */
MJS_K int    MJS_MPR_MAX_ITER = 48;
MJS_K double MJS_MPR_TOLERANCE = 1e-6;

/* collision stand-in for the CLOSED 2F-85 finger tips (deviation D-1): a sphere whose lowest point is
 * the TCP, on the lumped gripper body. Own choice, not in the reference. */
MJS_K double MJS_G2F85_PROXY_RADIUS = 0.012;

/* --------------------------------------------------- rendering (a15) */
/* Fixed cameras [REF]: MuJoCo camera convention = looks along its local -z, +x right, +y up;
 * quaternions (w,x,y,z) as written in the task code; fovy in degrees. */
MJS_K double MJS_PM_CAM_POS[3] = {0.0, 0.0, 2.4};            /* point_reach.py:22 TOP_DOWN_CAMERA_CONFIG */
MJS_K double MJS_PM_CAM_QUAT[4] = {1.0, 0.0, 0.0, 0.0};
MJS_K double MJS_PM_CAM_FOVY = 30.0;
MJS_K double MJS_RR_CAM_POS[3] = {0.0, -1.1, 0.5};           /* robot_reach.py:52 FRONT_TILTED_CAMERA_CONFIG */
MJS_K double MJS_RR_CAM_QUAT[4] = {-0.7, -0.35, 0.0, 0.0};
MJS_K double MJS_RR_CAM_FOVY = 70.0;
/* Pointmass scene appearance [REF]: mjcf/walled_pointmass_arena.xml:4-9,12-19 (checker texture,
 * decoration material, two lights), pointmass.py:55 (sphere rgba, clamped to [0,1]),
 * point_reach.py:91-93 (target site box), entities/utils.py:40 (mocap site). */
MJS_K float MJS_PM_GRID_RGB1[3] = {0.1f, 0.2f, 0.3f};
MJS_K float MJS_PM_GRID_RGB2[3] = {0.2f, 0.3f, 0.4f};
MJS_K float MJS_PM_WALL_RGB[3] = {0.3f, 0.5f, 0.7f};
MJS_K float MJS_PM_SPHERE_RGBA[4] = {1.0f, 0.0f, 0.0f, 0.5f};
MJS_K float MJS_PM_TARGET_RGB[3] = {0.0f, 1.0f, 0.0f};
MJS_K float MJS_PM_TARGET_HALF = 0.04f;
MJS_K float MJS_PM_MOCAP_SITE_RADIUS = 0.005f;
MJS_K float MJS_SITE_DEFAULT_RGB[3] = {0.5f, 0.5f, 0.5f};
MJS_K float MJS_PM_LIGHT_POS[2][3] = {{0.25f, 0.25f, 1.0f}, {-0.25f, -0.25f, 1.0f}};
/* [MJ] default light / headlight / material parameters */
MJS_K float MJS_LIGHT_DIFFUSE = 0.7f;
MJS_K float MJS_LIGHT_SPECULAR = 0.3f;
MJS_K float MJS_LIGHT_CUTOFF_COS = 0.70710678f;  /* cutoff 45 deg */
MJS_K float MJS_LIGHT_CUTOFF_COS2 = 0.5f;
MJS_K int   MJS_LIGHT_EXPONENT = 10;
MJS_K float MJS_HEADLIGHT_AMBIENT = 0.1f;
MJS_K float MJS_HEADLIGHT_DIFFUSE = 0.4f;
MJS_K float MJS_HEADLIGHT_SPECULAR = 0.5f;
MJS_K float MJS_MATERIAL_SPECULAR = 0.5f;
MJS_K int   MJS_MATERIAL_SHININESS_POW2 = 6;     /* shininess 0.5 -> GL exponent 64 = 2^6 */

/* Robot scenes appearance. The reference shows the menagerie visual MESHES (absent); the arm is
 * drawn with its collision proxies (MJS_UR_COL_*) plus two stand-ins (deviation D-6). Colours follow
 * the menagerie material names (linkgray / urblue / jointgray / black) [MEN]. */
MJS_K float MJS_RR_FLOOR_RGB[3] = {0.3f, 0.3f, 0.3f};          /* [REF] empty_robot_arena.py:19 */
MJS_K float MJS_RR_TARGET_RGB[3] = {1.0f, 1.0f, 1.0f};         /* [REF] robot_reach.py:98-105 */
MJS_K float MJS_RR_TARGET_RADIUS = 0.03f;
MJS_K float MJS_UR_LINKGRAY[3] = {0.82f, 0.82f, 0.82f};
MJS_K float MJS_UR_URBLUE[3] = {0.49f, 0.678f, 0.8f};
MJS_K float MJS_UR_JOINTGRAY[3] = {0.278f, 0.278f, 0.278f};
MJS_K float MJS_UR_BLACK[3] = {0.033f, 0.033f, 0.033f};
MJS_K int   MJS_UR_COL_IS_JOINT[MJS_UR_NCOLGEOM] = {1, 1, 0, 1, 0, 1, 1, 1, 0, 0};  /* urblue vs linkgray */
MJS_K float MJS_UR_BASE_STANDIN[2] = {0.075f, 0.05f};           /* cylinder radius, half-height at z = 0.05 */
MJS_K float MJS_G2F85_STANDIN_HALF[3] = {0.04f, 0.02f, 0.07f};   /* box in the flange frame, centred at z = 0.07 */
/* Button-Push scene [REF]: switch.py:25-37 (white box, button red / green when active, switch.py:59),
 * entities/camera.py:78-88 (camera bodies: black box + lens sphere) */
MJS_K float MJS_SW_BOX_RGB[3] = {1.0f, 1.0f, 1.0f};
MJS_K float MJS_SW_BUTTON_RGB_OFF[3] = {1.0f, 0.0f, 0.0f};
MJS_K float MJS_SW_BUTTON_RGB_ON[3] = {0.0f, 1.0f, 0.0f};
MJS_K float MJS_CAM_BODY_RGB[3] = {0.0f, 0.0f, 0.0f};
/* Planar-Push scene [REF]: cylinder.py:30 (EEF rgba), robot_planar_push.py:91-98 (target site: white cylinder r 0.05,
 * half-height 0.001), google_block.py:11-18 (block colours; the reference draws one at random from the unseeded
 * global `random`: here block i takes COLORS[i], deviation D-9) */
MJS_K float MJS_CYL_RGB[3] = {0.2f, 0.2f, 0.2f};
MJS_K float MJS_PP_TARGET_RGB[3] = {1.0f, 1.0f, 1.0f};
MJS_K float MJS_PP_TARGET_HALF_HEIGHT = 0.001f;
MJS_K float MJS_BLOCK_RGB[5][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 1.0f, 0.0f}, {1.0f, 1.0f, 0.0f}, {1.0f, 0.5f, 0.0f}}; /* box stand-in: block i */
/* [REF] google_block.py:12-19 COLORS: red, blue, green, yellow, orange, purple (a mesh block's sampled colour) */
MJS_K float MJS_BLOCK_COLORS[6][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 1.0f, 0.0f}, {1.0f, 1.0f, 0.0f}, {1.0f, 0.5f, 0.0f}, {1.0f, 0.0f, 1.0f}};
/* [REF] empty_robot_arena.py:24-26: six positional lights at (x, +-x, 3), x in {-3, 3, 0.5} */
MJS_K float MJS_RR_LIGHT_POS[6][3] = {{-3.0f, -3.0f, 3.0f}, {-3.0f, 3.0f, 3.0f}, {3.0f, 3.0f, 3.0f},
                                      {3.0f, -3.0f, 3.0f},  {0.5f, 0.5f, 3.0f},  {0.5f, -0.5f, 3.0f}};

#endif /* MJS_SCENE_SPEC_H */
