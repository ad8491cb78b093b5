/* om_tasks.c — ORACLE (test infrastructure): scene assembly, the
 * composer.Environment step/reset loop and the task logic, restated from
 *   environments/dmc2gym.py:133-163          (term/trunc split, reset)
 *   environments/tasks/point_reach.py:125-212 (Pointmass-Reach task)
 *   entities/pointmass.py:44-66,87-148        (PointMass2D entity)
 *   mjcf/walled_pointmass_arena.xml:11-20     (arena geometry)
 *   environments/tasks/robot_reach.py:85-207  (Robot-Reach task)
 *   entities/robots/robot.py:113-272,301-321  (servoL/servoJ, gravcomp, base quat)
 *   entities/robots/joint_trajectory.py:33-47 (linear joint interpolation)
 *   environments/tasks/spaces.py:24-31        (reset sampling order)
 * and dm_control's composer loop semantics (third-party).
 * PARITY UNPINNED — see mjs_oracle.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mjs_scene_spec.h"
#include "mjs_oracle.h"
#include "../include/mjs_block_hulls.h"

/* ------------------------------------------------------- model building */
static void quat_z2vec(double* q, const double* vec) {
  /* mju_quatZ2Vec: rotation taking (0,0,1) to vec */
  double z[3] = {0, 0, 1}, v[3] = {vec[0], vec[1], vec[2]};
  double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  q[0] = 1; q[1] = q[2] = q[3] = 0;
  if (n < MJS_MINVAL) return;
  for (int k = 0; k < 3; k++) v[k] /= n;
  double ax[3] = {z[1] * v[2] - z[2] * v[1], z[2] * v[0] - z[0] * v[2], z[0] * v[1] - z[1] * v[0]};
  double s = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (s < 1e-10) { ax[0] = 1; ax[1] = ax[2] = 0; } else { for (int k = 0; k < 3; k++) ax[k] /= s; }
  double ang = atan2(s, v[2]);
  q[0] = cos(ang / 2);
  for (int k = 0; k < 3; k++) q[1 + k] = ax[k] * sin(ang / 2);
}
static void quat_norm(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; k++) q[k] /= n;
}

static void model_defaults(om_model* m) {
  memset(m, 0, sizeof *m);
  m->gravity[2] = MJS_GRAVITY_Z;
  m->tolerance = MJS_SOLVER_TOLERANCE;
  m->iterations = MJS_SOLVER_ITERATIONS;
  m->impratio = 1.0;
  m->solref[0] = MJS_SOLREF_TIMECONST; m->solref[1] = MJS_SOLREF_DAMPRATIO;
  m->solimp[0] = MJS_SOLIMP_D0; m->solimp[1] = MJS_SOLIMP_DWIDTH; m->solimp[2] = MJS_SOLIMP_WIDTH;
  m->solimp[3] = MJS_SOLIMP_MIDPOINT; m->solimp[4] = MJS_SOLIMP_POWER;
  m->nbody = 1; /* world */
  m->touch_site = -1;
  m->gr_geom[0] = m->gr_geom[1] = -1;
  m->body_mocapid[0] = -1;
  m->body_quat[0][0] = m->body_iquat[0][0] = 1;
}
static int add_body(om_model* m, int parent, const double* pos, const double* quat, double mass, const double* ipos,
                    const double* iquat, const double* inertia, double gravcomp) {
  int b = m->nbody++;
  m->body_parent[b] = parent;
  memcpy(m->body_pos[b], pos, sizeof(double) * 3);
  memcpy(m->body_quat[b], quat, sizeof(double) * 4);
  quat_norm(m->body_quat[b]);
  m->body_mass[b] = mass;
  if (ipos) memcpy(m->body_ipos[b], ipos, sizeof(double) * 3);
  m->body_iquat[b][0] = 1;
  if (iquat) { memcpy(m->body_iquat[b], iquat, sizeof(double) * 4); quat_norm(m->body_iquat[b]); }
  if (inertia) memcpy(m->body_inertia[b], inertia, sizeof(double) * 3);
  m->body_gravcomp[b] = gravcomp;
  m->body_mocapid[b] = -1;
  m->body_jntadr[b] = m->njnt;
  m->body_dofadr[b] = m->nv;
  return b;
}
static int add_joint(om_model* m, int body, int type, const double* axis, int limited, double lo, double hi, double armature) {
  int j = m->njnt++;
  m->jnt_type[j] = type; m->jnt_body[j] = body;
  m->jnt_qposadr[j] = m->nq; m->jnt_dofadr[j] = m->nv;
  if (axis) memcpy(m->jnt_axis[j], axis, sizeof(double) * 3);
  m->jnt_limited[j] = limited; m->jnt_range[j][0] = lo; m->jnt_range[j][1] = hi;
  memcpy(m->jnt_solref[j], m->solref, sizeof m->solref); /* solreflimit / solimplimit default to the global defaults */
  memcpy(m->jnt_solimp[j], m->solimp, sizeof m->solimp);
  int nd = type == OM_JNT_FREE ? 6 : 1, nqj = type == OM_JNT_FREE ? 7 : 1;
  for (int k = 0; k < nd; k++) {
    int dof = m->nv + k;
    m->dof_body[dof] = body; m->dof_jnt[dof] = j; m->dof_armature[dof] = armature;
    if (k > 0 || m->body_dofnum[body] > 0) m->dof_parent[dof] = dof - 1;
    else {
      /* last dof of the nearest ancestor that has dofs */
      int p = m->body_parent[body];
      while (p > 0 && m->body_dofnum[p] == 0) p = m->body_parent[p];
      m->dof_parent[dof] = (p > 0) ? m->body_dofadr[p] + m->body_dofnum[p] - 1 : -1;
    }
  }
  m->body_jntnum[body]++;
  m->body_dofnum[body] += nd;
  m->nv += nd; m->nq += nqj;
  return j;
}
static int add_geom(om_model* m, int body, int type, const double* pos, const double* quat, double s0, double s1, double s2) {
  int g = m->ngeom++;
  m->geom_type[g] = type; m->geom_body[g] = body;
  if (pos) memcpy(m->geom_pos[g], pos, sizeof(double) * 3);
  m->geom_quat[g][0] = 1;
  if (quat) { memcpy(m->geom_quat[g], quat, sizeof(double) * 4); quat_norm(m->geom_quat[g]); }
  m->geom_size[g][0] = s0; m->geom_size[g][1] = s1; m->geom_size[g][2] = s2;
  m->geom_contype[g] = 1; m->geom_conaffinity[g] = 1; m->geom_condim[g] = 3;
  m->geom_friction[g][0] = MJS_GEOM_FRICTION_SLIDE; m->geom_friction[g][1] = MJS_GEOM_FRICTION_SPIN; m->geom_friction[g][2] = MJS_GEOM_FRICTION_ROLL;
  memcpy(m->geom_solref[g], m->solref, sizeof m->solref);
  memcpy(m->geom_solimp[g], m->solimp, sizeof m->solimp);
  return g;
}

/* Pointmass-Reach scene (point_reach.py:75-102) */
static void build_pointmass(om_model* m) {
  model_defaults(m);
  m->dt = MJS_PM_PHYSICS_DT;
  m->integrator = OM_INT_EULER;
  const double zero3[3] = {0, 0, 0}, ident[4] = {1, 0, 0, 0};
  /* arena planes on the world body, XML order */
  add_geom(m, 0, OM_GEOM_PLANE, zero3, ident, 0.5, 0.5, 0.1);
  const double wall_pos[4][3] = {{MJS_PM_ARENA_LO, 0, MJS_PM_WALL_Z}, {0, MJS_PM_ARENA_LO, MJS_PM_WALL_Z}, {MJS_PM_ARENA_HI, 0, MJS_PM_WALL_Z}, {0, MJS_PM_ARENA_HI, MJS_PM_WALL_Z}};
  const double wall_z[4][3] = {{1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}};
  for (int w = 0; w < 4; w++) {
    double q[4];
    quat_z2vec(q, wall_z[w]);
    add_geom(m, 0, OM_GEOM_PLANE, wall_pos[w], q, 0.5, 0.5, 0.02);
  }
  /* mocap body (entities/utils.py:32-41) */
  int mocap = add_body(m, 0, zero3, ident, 0, NULL, NULL, NULL, 0);
  m->body_mocapid[mocap] = 0; m->nmocap = 1;
  /* pointmass body: sphere, two slide joints (pointmass.py:52-66) */
  const double bpos[3] = {0, 0, MJS_PM_RADIUS};
  double I = 0.4 * MJS_PM_MASS * MJS_PM_RADIUS * MJS_PM_RADIUS;
  const double inertia[3] = {I, I, I};
  int pm = add_body(m, 0, bpos, ident, MJS_PM_MASS, zero3, ident, inertia, 0);
  const double ax[3] = {1, 0, 0}, ay[3] = {0, 1, 0};
  add_joint(m, pm, OM_JNT_SLIDE, ax, 0, 0, 0, 0);
  add_joint(m, pm, OM_JNT_SLIDE, ay, 0, 0, 0, 0);
  add_geom(m, pm, OM_GEOM_SPHERE, zero3, ident, MJS_PM_RADIUS, 0, 0);
  /* weld mocap <-> pointmass (point_reach.py:84-89); relpose from qpos0 */
  m->neq = 1;
  m->eq_type[0] = OM_EQ_WELD;
  m->eq_body1[0] = mocap; m->eq_body2[0] = pm;
  double* data = m->eq_data[0];
  memset(data, 0, sizeof(double) * 11);
  data[3] = 0; data[4] = 0; data[5] = MJS_PM_RADIUS; /* body2 anchor seen from body1 at qpos0 */
  data[6] = 1; data[10] = 1;
  memcpy(m->eq_solref[0], m->solref, sizeof m->solref);
  memcpy(m->eq_solimp[0], m->solimp, sizeof m->solimp);
  om_set_const(m);
}

/* UR5e + lumped 2F-85 payload on the empty arena (robot_reach.py:90-119) */
static void build_robot(om_model* m, int eef_gripper) {
  model_defaults(m);
  m->dt = MJS_RR_PHYSICS_DT;
  m->integrator = OM_INT_IMPLICITFAST;
  const double zero3[3] = {0, 0, 0}, ident[4] = {1, 0, 0, 0};
  add_geom(m, 0, OM_GEOM_PLANE, zero3, ident, MJS_ROBOT_ARENA_HALF, MJS_ROBOT_ARENA_HALF, 0.1);
  int parent = 0, bodies[MJS_UR_NBODY];
  for (int b = 0; b < MJS_UR_NBODY; b++) {
    bodies[b] = add_body(m, parent, MJS_UR_BODY_POS[b], MJS_UR_BODY_QUAT[b], MJS_UR_BODY_MASS[b], MJS_UR_BODY_IPOS[b],
                         MJS_UR_BODY_IQUAT[b], MJS_UR_BODY_DIAGINERTIA[b], 1.0 /* robot.py:80-82 */);
    if (b > 0) {
      int j = b - 1;
      add_joint(m, bodies[b], OM_JNT_HINGE, MJS_UR_JNT_AXIS[j], 1, MJS_UR_JNT_RANGE[j][0], MJS_UR_JNT_RANGE[j][1], MJS_UR_ARMATURE);
    }
    parent = bodies[b];
  }
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++)
    add_geom(m, bodies[MJS_UR_COL_BODY[g]], MJS_UR_COL_TYPE[g], MJS_UR_COL_POS[g], MJS_UR_COL_QUAT[g], MJS_UR_COL_SIZE[g][0], MJS_UR_COL_SIZE[g][1], 0);
  /* flange site */
  m->nsite = 1;
  m->site_body[0] = bodies[6];
  memcpy(m->site_pos[0], MJS_UR_FLANGE_POS, sizeof(double) * 3);
  memcpy(m->site_quat[0], MJS_UR_FLANGE_QUAT, sizeof(double) * 4);
  quat_norm(m->site_quat[0]);
  /* end-effector body attached at the flange site, gravcomp 0 */
  if (eef_gripper)
    add_body(m, bodies[6], MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, MJS_G2F85_MASS, MJS_G2F85_IPOS, ident, MJS_G2F85_DIAGINERTIA, 0.0);
  /* position servos */
  m->nu = MJS_UR_NJ;
  for (int u = 0; u < MJS_UR_NJ; u++) {
    m->act_jnt[u] = u;
    m->act_gain[u] = MJS_UR_ACT_KP[u];
    m->act_bias[u][0] = 0; m->act_bias[u][1] = -MJS_UR_ACT_KP[u]; m->act_bias[u][2] = -MJS_UR_ACT_KD[u];
    m->act_ctrllimited[u] = 1; m->act_forcelimited[u] = 1;
    m->act_ctrlrange[u][0] = MJS_UR_ACT_CTRLRANGE[u][0]; m->act_ctrlrange[u][1] = MJS_UR_ACT_CTRLRANGE[u][1];
    m->act_forcerange[u][0] = -MJS_UR_ACT_FRC[u]; m->act_forcerange[u][1] = MJS_UR_ACT_FRC[u];
  }
  om_set_const(m);
}

/* the UR5e alone, as the reference's component tests build it (test/test_ur_control_api.py:8-9: UR5e() compiled without an
 * arena or end effector) or with the lumped gripper; no floor, no collision geoms: the test scenes have nothing to touch */
void om_build_ur5e_alone(om_model* m, int eef_gripper, double dt) {
  build_robot(m, eef_gripper);
  m->ngeom = 0;
  m->dt = dt;
  om_set_const(m);
}

/* ---- reduced Robotiq 2F-85 ------------------------------------------------------------------------
 * One coordinate: the driver angle theta of the two fingers (the reference model couples them by a joint equality and
 * closes each four-bar by a connect constraint; here the closure holds by construction). Its ctrl is what
 * Robotiq2f85.move computes (gripper.py:80-84), its force the menagerie actuator's (MJS_G2F85_ACT_*), integrated like
 * mj_implicit (fast) integrates a dof with joint damping and an affine actuator. The finger opening follows the
 * reference's own map (gripper.py:73-75) and places the two finger-tip spheres. The fingers' inertia stays lumped in the
 * gripper body; their motion does not react on the arm. */
static double gripper_ctrl_of_opening(double finger_distance) {
  /* gripper.py:77-84; arcsin argument and ctrl clamped to their ranges (MuJoCo clamps ctrl to ctrlrange) */
  double a = (1 - finger_distance / MJS_G2F85_OPEN) * sin(MJS_G2F85_MAX_DRIVER);
  a = a < -1 ? -1 : a > 1 ? 1 : a;
  double c = asin(a) / MJS_G2F85_MAX_DRIVER * MJS_G2F85_CTRL_MAX;
  return c < 0 ? 0 : c > MJS_G2F85_CTRL_MAX ? MJS_G2F85_CTRL_MAX : c;
}
static double gripper_opening(double theta) { return MJS_G2F85_OPEN * (1 - sin(theta) / sin(MJS_G2F85_MAX_DRIVER)); } /* gripper.py:73-75 */
static void gripper_place_tips(om_env* e) {
  const double y = 0.5 * gripper_opening(e->gr_theta) + MJS_G2F85_PROXY_RADIUS;
  e->m.geom_pos[e->m.gr_geom[0]][1] = y;
  e->m.geom_pos[e->m.gr_geom[1]][1] = -y;
}
static void gripper_integrate(om_env* e) {
  const double h = e->m.dt, inertia = 2 * MJS_G2F85_DRIVER_ARMATURE, damping = 2 * MJS_G2F85_DRIVER_DAMPING;
  double F = MJS_G2F85_ACT_GAIN * e->gr_ctrl - MJS_G2F85_ACT_KP * e->gr_theta - MJS_G2F85_ACT_KV * e->gr_vel;
  int clamped = 0;
  if (F > MJS_G2F85_ACT_FORCE) { F = MJS_G2F85_ACT_FORCE; clamped = 1; }
  if (F < -MJS_G2F85_ACT_FORCE) { F = -MJS_G2F85_ACT_FORCE; clamped = 1; }
  const double f = F - damping * e->gr_vel;
  e->gr_vel += h * f / (inertia + h * (damping + (clamped ? 0.0 : MJS_G2F85_ACT_KV)));
  e->gr_theta += h * e->gr_vel;
}
/* one Physics.step() of the Button-Push scene: mj_step2 (arm + gripper), finger tips re-placed, mj_step1 */
static void button_physics_step(om_env* e) {
  if (e->cfg.gripper_model == OM_GRIPPER_ARTICULATED) { om_physics_step(&e->m, &e->d); return; } /* the fingers are part of the model */
  om_step2(&e->m, &e->d);
  gripper_integrate(e);
  gripper_place_tips(e);
  om_step1(&e->m, &e->d);
}

/* Button-Push scene (robot_push_button.py:66-108): UR5e + lumped gripper (+ the reduced 2F-85's two finger-tip
 * spheres) + wrist-camera geoms' mass + static switch (box, button cylinder, touch
 * site). The switch body position is a MODEL field rewritten at every reset (Entity.set_pose). */
/* The articulated Robotiq 2F-85 on the flange (gripper.py:36-98 loads the menagerie MJCF; robot.py:84-107 attaches it; SURVEY 8
 * f-1): 12 bodies / 8 hinges from include/mjs_scene_spec.h (MJS_G85_*), the two `connect` equalities that close the finger
 * linkages, the `joint` equality that couples the drivers, the fixed tendon "split" with `fingers_actuator` on it, the pad
 * boxes, and the MJCF's <option cone="elliptic" impratio="10"/>, which dm_control's attach merges into the scene's options.
 * Returns the index of the first gripper joint. */
static int add_articulated_gripper(om_model* m, int wrist3) {
  const double ident[4] = {1, 0, 0, 0};
  int frame = add_body(m, wrist3, MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, 0, NULL, NULL, NULL, 0.0);
  int body[MJS_G85_NBODY], jnt0 = m->njnt;
  for (int b = 0; b < MJS_G85_NBODY; b++) {
    int parent = MJS_G85_PARENT[b] < 0 ? frame : body[MJS_G85_PARENT[b]];
    body[b] = add_body(m, parent, MJS_G85_POS[b], MJS_G85_QUAT[b], MJS_G85_MASS[b], MJS_G85_IPOS[b], MJS_G85_IQUAT[b], MJS_G85_DIAGINERTIA[b], 0.0);
    int c = MJS_G85_JCLASS[b];
    if (c < 0) continue;
    const double ax[3] = {1, 0, 0};
    int j = add_joint(m, body[b], OM_JNT_HINGE, ax, 1, MJS_G85_JRANGE[c][0], MJS_G85_JRANGE[c][1], MJS_G85_JARMATURE[c]);
    memcpy(m->jnt_pos[j], MJS_G85_JPOS[c], sizeof(double) * 3);
    m->dof_damping[m->jnt_dofadr[j]] = MJS_G85_JDAMPING[c];
    m->jnt_stiffness[j] = MJS_G85_JSTIFFNESS[c];
    m->qpos_spring[m->jnt_qposadr[j]] = MJS_G85_JSPRINGREF[c];
    if (MJS_G85_JSTIFFLIMIT[c]) {
      memcpy(m->jnt_solref[j], MJS_G85_SOLREF, sizeof MJS_G85_SOLREF);
      memcpy(m->jnt_solimp[j], MJS_G85_SOLIMP, sizeof MJS_G85_SOLIMP);
    }
  }
  /* pad boxes: right pad first (MJCF order), box1 then box2 */
  const int pads[2] = {body[MJS_G85_B_RIGHT_PAD], body[MJS_G85_B_LEFT_PAD]};
  for (int s = 0; s < 2; s++)
    for (int k = 0; k < 2; k++) {
      int g = add_geom(m, pads[s], OM_GEOM_BOX, MJS_G85_PAD_POS[k], ident, MJS_G85_PAD_SIZE[0], MJS_G85_PAD_SIZE[1], MJS_G85_PAD_SIZE[2]);
      m->geom_priority[g] = 1;
      m->geom_friction[g][0] = MJS_G85_PAD_FRICTION[k];
      memcpy(m->geom_solref[g], MJS_G85_PAD_SOLREF, sizeof MJS_G85_PAD_SOLREF);
      memcpy(m->geom_solimp[g], MJS_G85_PAD_SOLIMP, sizeof MJS_G85_PAD_SOLIMP);
    }
  /* equality: connect(right follower, right coupler), connect(left follower, left coupler), anchor = the follower's origin;
   * joint(right driver = left driver, polycoef 0 1 0 0 0) */
  const int fol[2] = {body[MJS_G85_B_RIGHT_FOLLOWER], body[MJS_G85_B_LEFT_FOLLOWER]}, cou[2] = {body[MJS_G85_B_RIGHT_COUPLER], body[MJS_G85_B_LEFT_COUPLER]};
  for (int s = 0; s < 2; s++) {
    int e = m->neq++;
    m->eq_type[e] = OM_EQ_CONNECT; m->eq_body1[e] = fol[s]; m->eq_body2[e] = cou[s];
    memset(m->eq_data[e], 0, sizeof m->eq_data[e]); /* anchor 0 0 0 in body1; body2's anchor from qpos0 (om_set_const) */
    memcpy(m->eq_solref[e], MJS_G85_SOLREF, sizeof MJS_G85_SOLREF);
    memcpy(m->eq_solimp[e], MJS_G85_SOLIMP, sizeof MJS_G85_SOLIMP);
  }
  {
    int e = m->neq++;
    m->eq_type[e] = OM_EQ_JOINT; m->eq_body1[e] = jnt0 + MJS_G85_J_RIGHT_DRIVER; m->eq_body2[e] = jnt0 + MJS_G85_J_LEFT_DRIVER;
    memset(m->eq_data[e], 0, sizeof m->eq_data[e]);
    m->eq_data[e][1] = 1;
    memcpy(m->eq_solref[e], MJS_G85_SOLREF, sizeof MJS_G85_SOLREF);
    memcpy(m->eq_solimp[e], MJS_G85_SOLIMP, sizeof MJS_G85_SOLIMP);
  }
  /* tendon "split" + fingers_actuator (general, gaintype fixed, biastype affine) */
  m->ntendon = 1;
  memset(m->tendon_coef[0], 0, sizeof m->tendon_coef[0]);
  m->tendon_coef[0][m->jnt_dofadr[jnt0 + MJS_G85_J_RIGHT_DRIVER]] = MJS_G85_TENDON_COEF;
  m->tendon_coef[0][m->jnt_dofadr[jnt0 + MJS_G85_J_LEFT_DRIVER]] = MJS_G85_TENDON_COEF;
  int u = m->nu++;
  m->act_trntype[u] = OM_TRN_TENDON; m->act_jnt[u] = 0;
  m->act_gain[u] = MJS_G2F85_ACT_GAIN;
  m->act_bias[u][0] = 0; m->act_bias[u][1] = -MJS_G2F85_ACT_KP; m->act_bias[u][2] = -MJS_G2F85_ACT_KV;
  m->act_ctrllimited[u] = 1; m->act_forcelimited[u] = 1;
  m->act_ctrlrange[u][0] = 0; m->act_ctrlrange[u][1] = MJS_G2F85_CTRL_MAX;
  m->act_forcerange[u][0] = -MJS_G2F85_ACT_FORCE; m->act_forcerange[u][1] = MJS_G2F85_ACT_FORCE;
  m->cone = OM_CONE_ELLIPTIC;
  m->impratio = MJS_G85_IMPRATIO;
  return jnt0;
}

static void build_button(om_model* m, int gripper_model) {
  build_robot(m, gripper_model == OM_GRIPPER_ARTICULATED ? 0 : 1);
  const double zero3[3] = {0, 0, 0}, ident[4] = {1, 0, 0, 0};
  int payload = m->nbody - 1, wrist3 = gripper_model == OM_GRIPPER_ARTICULATED ? payload : m->body_parent[payload];
  if (gripper_model == OM_GRIPPER_ARTICULATED) add_articulated_gripper(m, wrist3);
  else {
  /* finger-tip spheres on the gripper body (tips at the TCP plane), placed from the driver angle by gripper_place_tips */
  const double ppos[3] = {0, 0, MJS_G2F85_TCP_Z - MJS_G2F85_PROXY_RADIUS};
  m->gr_geom[0] = add_geom(m, payload, OM_GEOM_SPHERE, ppos, ident, MJS_G2F85_PROXY_RADIUS, 0, 0);
  m->gr_geom[1] = add_geom(m, payload, OM_GEOM_SPHERE, ppos, ident, MJS_G2F85_PROXY_RADIUS, 0, 0);
  }
  /* wrist camera body: box + sphere of default density, concentric at MJS_WCAM_POS (mass only) */
  double bx = MJS_CAM_BOX_HALF[0], by = MJS_CAM_BOX_HALF[1], bz = MJS_CAM_BOX_HALF[2], rs = MJS_CAM_SPHERE_RADIUS;
  double mb = MJS_GEOM_DENSITY * 8 * bx * by * bz, ms = MJS_GEOM_DENSITY * 4.0 / 3.0 * 3.14159265358979323846 * rs * rs * rs;
  double Is = 0.4 * ms * rs * rs;
  double inertia[3] = {mb * (by * by + bz * bz) / 3 + Is, mb * (bx * bx + bz * bz) / 3 + Is, mb * (bx * bx + by * by) / 3 + Is};
  add_body(m, wrist3, MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, mb + ms, MJS_WCAM_POS, MJS_WCAM_QUAT, inertia, 0.0);
  /* switch: static body */
  int sw = add_body(m, 0, zero3, ident, 0, NULL, NULL, NULL, 0);
  const double boxpos[3] = {0, 0, MJS_SW_BOX_HALF}, butpos[3] = {0, 0, MJS_SW_BUTTON_Z};
  add_geom(m, sw, OM_GEOM_BOX, boxpos, ident, MJS_SW_BOX_HALF, MJS_SW_BOX_HALF, MJS_SW_BOX_HALF);
  add_geom(m, sw, OM_GEOM_CYLINDER, butpos, ident, MJS_SW_BUTTON_RADIUS, MJS_SW_BUTTON_HALF, 0);
  int s = m->nsite++;
  m->site_body[s] = sw;
  memcpy(m->site_pos[s], butpos, sizeof butpos);
  m->site_quat[s][0] = 1;
  m->touch_site = s;
  m->touch_size[0] = MJS_SW_BUTTON_RADIUS * MJS_SW_SITE_SCALE;
  m->touch_size[1] = MJS_SW_BUTTON_HALF * MJS_SW_SITE_SCALE;
  om_set_const(m);
}

/* Planar-Push scene (robot_planar_push.py:81-117): UR5e + CylinderEEF (cylinder.py:23-33) + n free blocks
 * (box stand-in for the cube mesh of google_block.py, deviation D-9); the target is a site (no collision). */
double om_dbg_perturb = 0; /* test knob (tests only): offset added to every block's x AND drop height z after the reset draws (the
                            * floor is translation-invariant in x: a pure x offset does not probe a lone block's tumbling) */
static void build_push(om_model* m, int n_objects) {
  build_robot(m, 0);
  const double zero3[3] = {0, 0, 0}, ident[4] = {1, 0, 0, 0};
  int wrist3 = m->nbody - 1;
  double r = MJS_CYL_RADIUS, h = MJS_CYL_HALFLEN, mc = MJS_CYL_MASS;
  double ixx = mc * (3 * r * r + 4 * h * h) / 12;
  const double cyl_inertia[3] = {ixx, ixx, 0.5 * mc * r * r}, cyl_pos[3] = {0, 0, MJS_CYL_POS_Z};
  int eef = add_body(m, wrist3, MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, mc, cyl_pos, ident, cyl_inertia, 0.0);
  add_geom(m, eef, OM_GEOM_CYLINDER, cyl_pos, ident, r, h, 0);
  for (int i = 0; i < n_objects; i++) {
    double a = MJS_BLOCK_HALF[0], b = MJS_BLOCK_HALF[1], c = MJS_BLOCK_HALF[2], mb = MJS_BLOCK_MASS;
    const double inertia[3] = {mb * (b * b + c * c) / 3, mb * (a * a + c * c) / 3, mb * (a * a + b * b) / 3};
    const double gpos[3] = {0, 0, MJS_BLOCK_GEOM_Z};
    int blk = add_body(m, 0, zero3, ident, mb, gpos, ident, inertia, 0.0);
    int j = add_joint(m, blk, OM_JNT_FREE, NULL, 0, 0, 0, 0);
    m->qpos0[m->jnt_qposadr[j] + 3] = 1.0;
    int g = add_geom(m, blk, OM_GEOM_BOX, gpos, ident, a, b, c);
    m->geom_condim[g] = MJS_BLOCK_CONDIM;
    for (int k = 0; k < 3; k++) m->geom_friction[g][k] = MJS_BLOCK_FRICTION[k];
  }
  om_set_const(m);
}

/* GoogleBlockProp (google_block.py:37-49): body i's geom becomes the mesh of `cat` at `scale` (collided by its convex hull),
 * mass 0.1 kg set on the geom, inertia of the closed mesh (include/mjs_block_hulls.h), condim 4, friction (1, 0.05, 0).
 * The reference rebuilds the MJCF every episode (initialize_episode_mjcf, robot_planar_push.py:144-147,163-167): a model edit. */
static void set_block_shape(om_model* m, int block, int cat, double scale) {
  int body = -1, seen = 0;
  for (int b = 1; b < m->nbody; b++)
    if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == OM_JNT_FREE && seen++ == block) { body = b; break; }
  if (body < 0) return;
  int g = -1;
  for (int k = 0; k < m->ngeom; k++) if (m->geom_body[k] == body) g = k;
  m->geom_type[g] = OM_GEOM_MESH;
  m->geom_mesh[g] = cat;
  m->geom_mesh_scale[g] = scale;
  for (int k = 0; k < 3; k++) {
    m->geom_pos[g][k] = MJS_HULL_COM[cat][k] * scale;
    m->body_ipos[body][k] = MJS_HULL_COM[cat][k] * scale;
    m->body_inertia[body][k] = MJS_BLOCK_MASS * MJS_HULL_INERTIA_PER_MASS[cat][k] * scale * scale;
  }
  m->body_mass[body] = MJS_BLOCK_MASS;
}

/* ------------------------------------------------------------ task API */
void om_default_config(int task, om_task_config* cfg) {
  memset(cfg, 0, sizeof *cfg);
  cfg->task = task;
  cfg->autoreset = OM_AUTORESET_NEXT_STEP;
  if (task == OM_TASK_POINTMASS) {
    cfg->reward_type = OM_REW_DENSE_BIASED_NEG_DISTANCE;                  /* point_reach.py:63 */
    cfg->time_limit = MJS_PM_MAX_CONTROL_STEPS * MJS_PM_CONTROL_DT;      /* __init__.py:21,28 */
  } else if (task == OM_TASK_PLANAR_PUSH) {
    cfg->reward_type = OM_REW_DENSE_NEG_DISTANCE;                         /* robot_planar_push.py:69 */
    cfg->time_limit = 1e300;                                              /* scripts/sb3/planar_push.py:72: no Environment time limit */
    cfg->n_objects = MJS_PP_FAST_OBJECTS;                                 /* BASELINE config 4 / robot_planar_push.py:315 */
    cfg->block_shape = OM_BLOCKS_MESH;                                    /* robot_planar_push.py:164: GoogleBlockProp.sample_random_object() */
    cfg->max_episode_steps = MJS_PP_MAX_CONTROL_STEPS;                    /* robot_planar_push.py:53 */
  } else if (task == OM_TASK_BUTTON_PUSH) {
    cfg->reward_type = OM_REW_SPARSE;                                     /* robot_push_button.py:47 */
    cfg->time_limit = MJS_BP_MAX_CONTROL_STEPS * MJS_RR_CONTROL_DT;      /* __init__.py:32-34 */
    cfg->action_type = OM_ACTION_ABS_JOINT;                               /* robot_push_button.py:49 */
  } else {
    cfg->reward_type = OM_REW_DENSE_NEG_DISTANCE;                         /* robot_reach.py:75 */
    cfg->time_limit = MJS_RR_MAX_CONTROL_STEPS * MJS_RR_CONTROL_DT;      /* BASELINE cfg 3 */
  }
}
/* Button-Push flat obs: ur5e/joint_configuration(6), ur5e/tcp_position(3), switch position(3), active(1) */
/* Planar-Push flat obs: ur5e/tcp_position(3), target_position(2), block_positions(2 per block slot: 2 slots for
 * n_objects <= 2, 5 otherwise; om_obs_dim() is the 2-slot layout, om_obs_dim_for() the configured one) */
int om_obs_dim(int task) { return task == OM_TASK_POINTMASS ? 4 : task == OM_TASK_BUTTON_PUSH ? 13 : task == OM_TASK_PLANAR_PUSH ? 5 + 2 * MJS_PP_FAST_OBJECTS : 12; }
int om_obs_dim_for(const om_task_config* cfg) {
  return cfg->task == OM_TASK_PLANAR_PUSH ? 5 + 2 * MJS_PP_OBJECT_SLOTS(cfg->n_objects > 0 ? cfg->n_objects : MJS_PP_FAST_OBJECTS) : om_obs_dim(cfg->task);
}
int om_action_dim(int task) { return task == OM_TASK_POINTMASS ? 2 : task == OM_TASK_BUTTON_PUSH ? 7 : task == OM_TASK_PLANAR_PUSH ? 2 : 3; }
int om_sizeof_step_out(void) { return (int)sizeof(om_step_out); }

void om_env_seed(om_env* e, uint32_t seed) { om_rng_seed(&e->rng, seed); }

void om_env_init(om_env* e, const om_task_config* cfg, uint32_t seed) {
  memset(e, 0, sizeof *e);
  e->cfg = *cfg;
  if (cfg->task == OM_TASK_POINTMASS) { build_pointmass(&e->m); e->n_sub = (int)round(MJS_PM_CONTROL_DT / MJS_PM_PHYSICS_DT); }
  else if (cfg->task == OM_TASK_BUTTON_PUSH) { build_button(&e->m, cfg->gripper_model); e->n_sub = (int)round(MJS_RR_CONTROL_DT / MJS_RR_PHYSICS_DT); }
  else if (cfg->task == OM_TASK_PLANAR_PUSH) { build_push(&e->m, cfg->n_objects); e->n_sub = (int)round(MJS_RR_CONTROL_DT / MJS_RR_PHYSICS_DT); }
  else { build_robot(&e->m, 1); e->n_sub = (int)round(MJS_RR_CONTROL_DT / MJS_RR_PHYSICS_DT); }
  e->distance_to_target = 1.0;          /* point_reach.py:112-113 */
  e->previous_distance_to_target = 1.0;
  om_env_seed(e, seed);
  om_reset_data(&e->m, &e->d);
  e->reset_pending = 1;
}

/* TCP pose -> flange pose -> IK (robot.py:113-121,138-151); pose quaternion is scalar-last */
static int tcp_pose_to_joints(const double* pos, const double* quat_xyzw, double tcp_z, const double* q_guess, double* q_out) {
  double x = quat_xyzw[0], y = quat_xyzw[1], z = quat_xyzw[2], w = quat_xyzw[3];
  double n = sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  double T[16] = {R[0], R[1], R[2], pos[0] - R[2] * tcp_z, R[3], R[4], R[5], pos[1] - R[5] * tcp_z,
                  R[6], R[7], R[8], pos[2] - R[8] * tcp_z, 0, 0, 0, 1};
  return om_ur5e_ik_closest(T, q_guess, q_out);
}

static void get_tcp_position(const om_env* e, double* tcp) {
  /* robot.py:153-168: flange site pose, then the TCP offset along the flange z axis */
  const double* p = e->d.site_xpos[0];
  const double* R = e->d.site_xmat[0];
  const double tcp_z = e->cfg.task == OM_TASK_PLANAR_PUSH ? MJS_CYL_TCP_Z : MJS_G2F85_TCP_Z;
  for (int k = 0; k < 3; k++) tcp[k] = p[k] + R[3 * k + 2] * tcp_z;
}

static void write_obs(const om_env* e, double* obs) {
  if (e->cfg.task == OM_TASK_POINTMASS) {
    obs[0] = e->d.xpos[2][0]; obs[1] = e->d.xpos[2][1];     /* pointmass/position (pointmass.py:87-98) */
    obs[2] = e->target_pos[0]; obs[3] = e->target_pos[1];   /* goal_position (point_reach.py:211-212) */
  } else if (e->cfg.task == OM_TASK_PLANAR_PUSH) {
    get_tcp_position(e, obs);                                /* ur5e/tcp_position */
    obs[3] = e->target_pos[0]; obs[4] = e->target_pos[1];   /* target_position = site.pos[:2] (robot_planar_push.py:120) */
    for (int i = 0; i < MJS_PP_OBJECT_SLOTS(e->cfg.n_objects); i++) /* block_positions = body xpos[:2] per block (:178-179) */
      for (int k = 0; k < 2; k++) obs[5 + 2 * i + k] = i < e->cfg.n_objects ? e->d.qpos[6 + 7 * i + k] : 0.0;
  } else if (e->cfg.task == OM_TASK_BUTTON_PUSH) {
    for (int j = 0; j < 6; j++) obs[j] = e->d.qpos[j];       /* ur5e/joint_configuration (robot.py:296-298) */
    get_tcp_position(e, obs + 6);                            /* ur5e/tcp_position */
    /* Switch.get_position: button geom xpos + 0.5*size[1] on ALL coordinates (switch.py:86-87) */
    for (int k = 0; k < 3; k++) obs[9 + k] = e->switch_pos[k] + (k == 2 ? MJS_SW_BUTTON_Z : 0.0) + MJS_SW_POSITION_OFFSET;
    obs[12] = e->switch_active;                              /* SwitchObservables.active (intended, App. D-6) */
  } else {
    get_tcp_position(e, obs);                                /* ur5e/tcp_position */
    for (int j = 0; j < 6; j++) obs[3 + j] = e->d.qpos[j];   /* ur5e/joint_configuration */
    for (int k = 0; k < 3; k++) obs[9 + k] = e->target_pos[k]; /* target_position */
  }
}

/* Switch._update_activation (switch.py:51-60), called from initialize_episode and after every substep */
static void switch_update(om_env* e) {
  double f = e->d.touch_force;
  int was = e->switch_pressed;
  e->switch_pressed = (f >= MJS_SW_MIN_FORCE && f <= MJS_SW_MAX_FORCE);
  if (e->switch_pressed && !was) e->switch_active = !e->switch_active; /* flip on the rising edge */
  e->switch_num_pressed += e->switch_pressed;
}

static void episode_init(om_env* e) {
  om_model* m = &e->m;
  om_data* d = &e->d;
  om_reset_data(m, d);
  e->traj_active = 0;
  e->ik_failed = 0;
  if (e->cfg.task == OM_TASK_POINTMASS) {
    /* point_reach.py:125-144: goal_x, goal_y, point_x, point_y */
    double lo = MJS_PM_ARENA_LO + MJS_PM_RADIUS, hi = MJS_PM_ARENA_HI - MJS_PM_RADIUS;
    double gx = om_rng_uniform(&e->rng, lo, hi), gy = om_rng_uniform(&e->rng, lo, hi);
    e->target_pos[0] = gx; e->target_pos[1] = gy; e->target_pos[2] = MJS_PM_RADIUS / 2;
    double px = om_rng_uniform(&e->rng, lo, hi), py = om_rng_uniform(&e->rng, lo, hi);
    d->qpos[0] = px; d->qpos[1] = py; d->qvel[0] = d->qvel[1] = 0;
    d->mocap_pos[0][0] = px; d->mocap_pos[0][1] = py;
  } else if (e->cfg.task == OM_TASK_PLANAR_PUSH) {
    /* robot_planar_push.py:149-176 (intended semantics): robot xyz -> IK; target xyz; then the
     * blocks' xyz (identity orientation), ALL blocks re-drawn until mj_forward reports no contact; 150 settle steps */
    double rp[3], q[6], zeros[6] = {0, 0, 0, 0, 0, 0};
    e->episode_step = 0; /* base.py:29 */
    if (e->cfg.block_shape == OM_BLOCKS_MESH) {
      /* initialize_episode_mjcf (robot_planar_push.py:144-147,163-167) runs BEFORE initialize_episode: every block is replaced by
       * GoogleBlockProp.sample_random_object() (google_block.py:55-68): category, colour, scale in [0.8, 1.2]. The reference draws
       * them from Python's unseeded global `random`; here they come from the env's seeded stream (deviation D-5), three uniforms
       * per block in that order. */
      for (int i = 0; i < e->cfg.n_objects; i++) {
        int cat = (int)om_rng_uniform(&e->rng, 0.0, (double)MJS_HULL_NCAT), col = (int)om_rng_uniform(&e->rng, 0.0, 6.0);
        e->block_cat[i] = cat < MJS_HULL_NCAT ? cat : MJS_HULL_NCAT - 1;
        e->block_color[i] = col < 6 ? col : 5;
        e->block_scale[i] = om_rng_uniform(&e->rng, MJS_BLOCK_SCALE_LO, MJS_BLOCK_SCALE_HI);
        set_block_shape(m, i, e->block_cat[i], e->block_scale[i]);
      }
      om_set_const(m); /* the recompiled model's invweight0 / meaninertia */
      om_reset_data(m, d);
    }
    for (int k = 0; k < 3; k++) rp[k] = om_rng_uniform(&e->rng, MJS_PP_ROBOT_SPACE_LO[k], MJS_PP_ROBOT_SPACE_HI[k]);
    if (tcp_pose_to_joints(rp, MJS_TOP_DOWN_QUAT_XYZW, MJS_CYL_TCP_Z, zeros, q))
      for (int j = 0; j < 6; j++) { d->qpos[j] = q[j]; d->qvel[j] = 0; d->ctrl[j] = q[j]; }
    for (int k = 0; k < 3; k++) e->target_pos[k] = om_rng_uniform(&e->rng, MJS_PP_TARGET_SPACE_LO[k], MJS_PP_TARGET_SPACE_HI[k]);
    for (;;) {
      for (int i = 0; i < e->cfg.n_objects; i++) {
        double* qp = d->qpos + 6 + 7 * i;
        for (int k = 0; k < 3; k++) qp[k] = om_rng_uniform(&e->rng, MJS_PP_OBJECT_SPACE_LO[k], MJS_PP_OBJECT_SPACE_HI[k]);
        qp[3] = 1; qp[4] = qp[5] = qp[6] = 0;
      }
      om_forward(m, d);
      if (d->ncon == 0) break;
    }
    if (om_dbg_perturb != 0) { /* test knob: sensitivity of the settle phase (every block, offsets of different size) */
      for (int i = 0; i < e->cfg.n_objects; i++) { d->qpos[6 + 7 * i] += om_dbg_perturb * (1 + i); d->qpos[6 + 7 * i + 2] += om_dbg_perturb * (1 + 0.5 * i); }
      om_forward(m, d);
    }
    for (int s = 0; s < MJS_PP_SETTLE_STEPS; s++) om_physics_step(m, d);
    return;
  } else if (e->cfg.task == OM_TASK_BUTTON_PUSH) {
    /* robot_push_button.py:126-134: robot xyz -> IK -> joints; switch xyz -> switch.set_pose (model edit) */
    double rp[3], q[6], zeros[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 3; k++) rp[k] = om_rng_uniform(&e->rng, MJS_BP_ROBOT_SPACE_LO[k], MJS_BP_ROBOT_SPACE_HI[k]);
    if (tcp_pose_to_joints(rp, MJS_TOP_DOWN_QUAT_XYZW, MJS_G2F85_TCP_Z, zeros, q))
      for (int j = 0; j < 6; j++) { d->qpos[j] = q[j]; d->qvel[j] = 0; d->ctrl[j] = q[j]; }
    for (int k = 0; k < 3; k++) e->switch_pos[k] = om_rng_uniform(&e->rng, MJS_BP_SWITCH_SPACE_LO[k], MJS_BP_SWITCH_SPACE_HI[k]);
    memcpy(m->body_pos[m->site_body[m->touch_site]], e->switch_pos, sizeof e->switch_pos);
    e->gr_theta = e->gr_vel = e->gr_ctrl = 0; /* mj_resetData: gripper joints at qpos0 (open), ctrl 0 */
    if (e->cfg.gripper_model != OM_GRIPPER_ARTICULATED) gripper_place_tips(e);
    om_forward(m, d);
    /* Switch.initialize_episode (switch.py:62-65) */
    e->switch_num_pressed = 0;
    e->switch_active = 0;
    switch_update(e);
    return;
  } else {
    /* robot_reach.py:143-150 with spaces.py:24-31 */
    double rp[3], tp[3], q[6], zeros[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 3; k++) rp[k] = om_rng_uniform(&e->rng, MJS_RR_SPACE_LO[k], MJS_RR_SPACE_HI[k]);
    if (tcp_pose_to_joints(rp, MJS_TOP_DOWN_QUAT_XYZW, MJS_G2F85_TCP_Z, zeros, q)) {
      for (int j = 0; j < 6; j++) { d->qpos[j] = q[j]; d->qvel[j] = 0; d->ctrl[j] = q[j]; } /* robot.py:185-189 */
    } /* else: silently keep qpos0 (robot.py:179-183) */
    for (int k = 0; k < 3; k++) tp[k] = om_rng_uniform(&e->rng, MJS_RR_SPACE_LO[k], MJS_RR_SPACE_HI[k]);
    memcpy(e->target_pos, tp, sizeof tp);
  }
  om_forward(m, d);
}

static int count_contacts(const om_data* d) { return d->ncon; }

void om_env_reset(om_env* e, om_step_out* out) {
  memset(out, 0, sizeof *out);
  episode_init(e);
  e->reset_pending = 0;
  write_obs(e, out->obs);
  out->step_type = OM_STEP_FIRST;
  out->reward = 0.0; out->discount = 1.0; /* dm_env: None / None on FIRST */
  out->ncon = count_contacts(&e->d);
}

static int task_accomplished(const om_env* e, double* dist_out) {
  double tcp[3], dd = 0;
  get_tcp_position(e, tcp);
  for (int k = 0; k < 3; k++) dd += (tcp[k] - e->target_pos[k]) * (tcp[k] - e->target_pos[k]);
  dd = sqrt(dd);
  if (dist_out) *dist_out = dd;
  return dd < MJS_RR_GOAL_THRESHOLD;
}

void om_env_step(om_env* e, const double* action, om_step_out* out) {
  om_model* m = &e->m;
  om_data* d = &e->d;
  if (e->reset_pending && e->cfg.autoreset == OM_AUTORESET_NEXT_STEP) { om_env_reset(e, out); return; }
  memset(out, 0, sizeof *out);
  d->warning_bad = 0;
  /* before_step hooks */
  if (e->cfg.task == OM_TASK_POINTMASS) {
    /* point_reach.py:150-163 */
    for (int k = 0; k < 2; k++) {
      double t = d->xpos[2][k] + action[k];
      t = fmin(fmax(t, MJS_PM_ARENA_LO), MJS_PM_ARENA_HI);
      d->mocap_pos[0][k] = t;
    }
  } else if (e->cfg.task == OM_TASK_PLANAR_PUSH) {
    /* base.py:31-32 + robot_planar_push.py:185-201: the 2-D action is the absolute TCP xy in metres, z = 0.02 */
    e->episode_step++;
    double q_now[6], q_ik[6], tp[3] = {action[0], action[1], MJS_PP_ACTION_Z};
    memcpy(q_now, d->qpos, sizeof q_now);
    if (!tcp_pose_to_joints(tp, MJS_TOP_DOWN_QUAT_XYZW, MJS_CYL_TCP_Z, q_now, q_ik)) {
      e->ik_failed = 1;
      memcpy(q_ik, q_now, sizeof q_ik);
    }
    memcpy(e->traj_q0, q_now, sizeof q_now);
    memcpy(e->traj_q1, q_ik, sizeof q_ik);
    e->traj_t0 = d->time;
    e->traj_t1 = d->time + MJS_RR_CONTROL_DT;
    e->traj_active = 1;
  } else if (e->cfg.task == OM_TASK_BUTTON_PUSH && e->cfg.action_type == OM_ACTION_ABS_JOINT) {
    /* robot_push_button.py:151-157: gripper.move(a[6]) sets the finger actuator's ctrl; servoJ(a[:6]) -> robot.py:227-259 */
    e->gr_ctrl = gripper_ctrl_of_opening(action[6]);
    if (e->cfg.gripper_model == OM_GRIPPER_ARTICULATED) d->ctrl[6] = e->gr_ctrl; /* physics.bind(self.actuator).ctrl (gripper.py:84) */
    memcpy(e->traj_q0, d->qpos, sizeof e->traj_q0);
    memcpy(e->traj_q1, action, sizeof e->traj_q1);
    e->traj_t0 = d->time;
    e->traj_t1 = d->time + MJS_RR_CONTROL_DT;
    e->traj_active = 1;
  } else {
    /* robot_reach.py:159-169 / robot_push_button.py:143-149 -> robot.py:218-259 */
    if (e->cfg.task == OM_TASK_BUTTON_PUSH) e->gr_ctrl = gripper_ctrl_of_opening(action[3]); /* gripper.move(a[3]), :147 */
    if (e->cfg.task == OM_TASK_BUTTON_PUSH && e->cfg.gripper_model == OM_GRIPPER_ARTICULATED) d->ctrl[6] = e->gr_ctrl;
    double q_now[6], q_ik[6];
    memcpy(q_now, d->qpos, sizeof q_now);
    if (!tcp_pose_to_joints(action, MJS_TOP_DOWN_QUAT_XYZW, MJS_G2F85_TCP_Z, q_now, q_ik)) {
      e->ik_failed = 1; /* reference raises ValueError (robot.py:221-224); batched: flag + hold (D-4) */
      memcpy(q_ik, q_now, sizeof q_ik);
    }
    memcpy(e->traj_q0, q_now, sizeof q_now);
    memcpy(e->traj_q1, q_ik, sizeof q_ik);
    e->traj_t0 = d->time;
    e->traj_t1 = d->time + MJS_RR_CONTROL_DT;
    e->traj_active = 1;
  }
  /* substeps */
  for (int s = 0; s < e->n_sub; s++) {
    if (e->traj_active) {
      /* robot.py:261-263, joint_trajectory.py:33-47 */
      double t = fmin(fmax(d->time, e->traj_t0), e->traj_t1);
      for (int j = 0; j < 6; j++) d->ctrl[j] = e->traj_q0[j] + (e->traj_q1[j] - e->traj_q0[j]) * (t - e->traj_t0) / (e->traj_t1 - e->traj_t0);
    }
    if (e->cfg.task == OM_TASK_BUTTON_PUSH) button_physics_step(e);
    else om_physics_step(m, d);
    if (e->cfg.task == OM_TASK_BUTTON_PUSH) switch_update(e); /* Switch.after_substep (switch.py:71-72) */
    if (e->cfg.task != OM_TASK_POINTMASS) { /* test knob: a floor contact of one of the arm's own collision geoms (ids 1..10) or, in Planar-Push, of the CylinderEEF (id 11) */
      const int last = MJS_UR_NCOLGEOM + (e->cfg.task == OM_TASK_PLANAR_PUSH ? 1 : 0);
      for (int c = 0; c < d->ncon; c++)
        if (d->contact[c].geom1 == 0 && d->contact[c].geom2 >= 1 && d->contact[c].geom2 <= last) e->dbg_arm_floor_seen = 1;
    }
  }
  /* after_step + reward/discount/termination */
  if (e->cfg.task == OM_TASK_BUTTON_PUSH && e->cfg.button_disturbances) {
    /* robot_push_button.py:159-165: rand() is only drawn when the switch is active and not pressed */
    if (e->switch_active && !e->switch_pressed && om_rng_uniform(&e->rng, 0.0, 1.0) < 0.01) e->switch_active = 0;
  }
  int terminate = 0, success = 0;
  double reward = 0, discount = 1;
  if (e->cfg.task == OM_TASK_POINTMASS) {
    /* point_reach.py:165-202 */
    e->previous_distance_to_target = e->distance_to_target;
    double dx = d->xpos[2][0] - e->target_pos[0], dy = d->xpos[2][1] - e->target_pos[1];
    e->distance_to_target = sqrt(dx * dx + dy * dy);
    success = e->distance_to_target < MJS_PM_GOAL_THRESHOLD;
    switch (e->cfg.reward_type) {
      case OM_REW_SPARSE: reward = success ? 1.0 : 0.0; break; /* intended; reference branch broken (App. D-4) */
      case OM_REW_DENSE_NEG_DISTANCE: reward = -e->distance_to_target; break;
      case OM_REW_DENSE_POTENTIAL: reward = e->previous_distance_to_target - e->distance_to_target; break;
      default: reward = -e->distance_to_target + 0.5; break;
    }
    terminate = success;
    discount = success ? 0.0 : 1.0;
  } else if (e->cfg.task == OM_TASK_PLANAR_PUSH) {
    /* robot_planar_push.py:203-241 + base.py:47-57 */
    double tcp[3], sum = 0, nearest = INFINITY;
    int n = e->cfg.n_objects, inside = 0;
    get_tcp_position(e, tcp);
    for (int i = 0; i < n; i++) {
      const double* bp = d->qpos + 6 + 7 * i; /* body xpos of a free body = its qpos */
      double dt = sqrt((bp[0] - e->target_pos[0]) * (bp[0] - e->target_pos[0]) + (bp[1] - e->target_pos[1]) * (bp[1] - e->target_pos[1]));
      double dr = sqrt((tcp[0] - bp[0]) * (tcp[0] - bp[0]) + (tcp[1] - bp[1]) * (tcp[1] - bp[1]));
      sum += dt;
      inside += dt < MJS_PP_TARGET_RADIUS;
      if (dr < nearest) nearest = dr;
    }
    success = inside == n;
    if (e->cfg.reward_type == OM_REW_SPARSE) reward = inside;
    else reward = (-sum / n - MJS_PP_NEAREST_COEF * nearest) * MJS_PP_REWARD_SCALE;
    terminate = success || e->episode_step >= e->cfg.max_episode_steps;
    discount = success ? 0.0 : 1.0;
  } else if (e->cfg.task == OM_TASK_BUTTON_PUSH) {
    /* robot_push_button.py:167-170,205-219: goal = switch active AND tcp within 0.05 of the end position */
    double tcp[3], dd = 0;
    get_tcp_position(e, tcp);
    for (int k = 0; k < 3; k++) dd += (tcp[k] - MJS_BP_ROBOT_END_POS[k]) * (tcp[k] - MJS_BP_ROBOT_END_POS[k]);
    success = e->switch_active && sqrt(dd) < MJS_BP_GOAL_THRESHOLD;
    reward = success ? 1.0 : 0.0;
    terminate = success;
    discount = success ? 0.0 : 1.0;
  } else {
    double dist;
    success = task_accomplished(e, &dist);
    reward = (e->cfg.reward_type == OM_REW_SPARSE) ? (success ? 1.0 : 0.0) : -dist; /* robot_reach.py:174-181 */
    if (e->cfg.terminate_on_success && success) { terminate = 1; discount = 0.0; } /* opt-in, D-2 */
  }
  if (d->warning_bad) { reward = 0; discount = 0; terminate = 1; out->fault = 1; } /* PhysicsError path */
  if (d->time >= e->cfg.time_limit) terminate = 1;
  write_obs(e, out->obs);
  out->reward = reward; out->discount = discount;
  out->step_type = terminate ? OM_STEP_LAST : OM_STEP_MID;
  out->terminated = terminate && discount == 0.0; /* dmc2gym.py:144-145 */
  out->truncated = terminate && discount > 0.0;
  out->is_success = success;
  out->ncon = count_contacts(d);
  out->ik_failed = e->ik_failed;
  e->reset_pending = terminate;
  if (terminate && e->cfg.autoreset == OM_AUTORESET_SAME_STEP) {
    memcpy(out->terminal_obs, out->obs, sizeof out->obs);
    episode_init(e);
    e->reset_pending = 0;
    write_obs(e, out->obs);
    out->ncon = count_contacts(d); /* d->ncon as the caller would read it after the reset's mj_forward */
  }
}

/* debug hook for tests: joint-space inertia (incl. armature) and qfrc_bias - qfrc_passive of the
 * Robot-Reach model at (q, v) */
void om_debug_reach_dynamics(const double* q, const double* v, double* M_out /*36*/, double* bias_out /*6*/) {
  static __thread om_model m;
  static __thread om_data d;
  static __thread int built = 0;
  if (!built) { build_robot(&m, 1); built = 1; }
  om_reset_data(&m, &d);
  memcpy(d.qpos, q, sizeof(double) * 6);
  memcpy(d.qvel, v, sizeof(double) * 6);
  om_step1(&m, &d);
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) M_out[6 * i + j] = d.M[i][j];
    bias_out[i] = d.qfrc_bias[i] - d.qfrc_passive[i];
  }
}

/* debug hook for tests: same as om_debug_reach_dynamics for the Button-Push model, plus the
 * body_invweight0 (translation, rotation) of the gripper body (the contact rows' diagApprox) */
void om_debug_button_dynamics(const double* q, const double* v, double* M_out, double* bias_out, double* invw_out) {
  static __thread om_model m;
  static __thread om_data d;
  static __thread int built = 0;
  if (!built) { build_button(&m, OM_GRIPPER_REDUCED); built = 1; }
  om_reset_data(&m, &d);
  memcpy(d.qpos, q, sizeof(double) * 6);
  memcpy(d.qvel, v, sizeof(double) * 6);
  om_step1(&m, &d);
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) M_out[6 * i + j] = d.M[i][j];
    bias_out[i] = d.qfrc_bias[i] - d.qfrc_passive[i];
  }
  invw_out[0] = m.body_invweight0[8][0];
  invw_out[1] = m.body_invweight0[8][1];
  invw_out[2] = m.meaninertia;
  for (int i = 0; i < 6; i++) invw_out[3 + i] = m.dof_invweight0[i];
}

/* debug hook for tests: body_invweight0 (translation) of the UR5e's own bodies (base, links 1..6) in the model of `task`
 * (what mj_makeImpedance uses as diagApprox for a contact of one of the arm's collision geoms with the static floor) */
void om_debug_link_invweights(int task, double* out7) {
  static __thread om_model m;
  if (task == OM_TASK_BUTTON_PUSH) build_button(&m, OM_GRIPPER_REDUCED);
  else if (task == OM_TASK_PLANAR_PUSH) build_push(&m, 2);
  else build_robot(&m, 1);
  for (int k = 0; k < 7; k++) out7[k] = m.body_invweight0[1 + k][0];
}

/* debug hook for tests: give block `i` of a Planar-Push env the mesh `cat` at `scale` (as initialize_episode_mjcf would) */
void om_debug_set_block_shape(om_env* e, int i, int cat, int color, double scale) {
  e->block_cat[i] = cat; e->block_color[i] = color; e->block_scale[i] = scale;
  set_block_shape(&e->m, i, cat, scale);
  om_set_const(&e->m);
}

void om_debug_get_block_shape(const om_env* e, int* cat, int* color, double* scale) {
  for (int i = 0; i < 5; i++) { cat[i] = e->block_cat[i]; color[i] = e->block_color[i]; scale[i] = e->block_scale[i]; }
}

/* debug hook for tests: did an arm link touch the floor in any substep since the last call? (the HIP kernels detect and count
 * those contacts but do not solve them, DESIGN.md D-8: tests drop such envs from the comparison) */
int om_debug_arm_floor_seen(om_env* e) {
  int r = e->dbg_arm_floor_seen;
  e->dbg_arm_floor_seen = 0;
  return r;
}

/* debug hook for tests: copy out qpos[nq], qvel[nv] and time of an env; returns nq */
int om_debug_get_state(const om_env* e, double* qpos, double* qvel, double* time) {
  memcpy(qpos, e->d.qpos, sizeof(double) * e->m.nq);
  memcpy(qvel, e->d.qvel, sizeof(double) * e->m.nv);
  *time = e->d.time;
  return e->m.nq;
}

/* debug hooks for tests: overwrite the whole state (then mj_forward); run n raw Physics.step() with the current ctrl */
void om_debug_set_state(om_env* e, const double* qpos, const double* qvel) {
  memcpy(e->d.qpos, qpos, sizeof(double) * e->m.nq);
  memcpy(e->d.qvel, qvel, sizeof(double) * e->m.nv);
  memset(e->d.qacc_warmstart, 0, sizeof e->d.qacc_warmstart);
  for (int j = 0; j < e->m.nu && j < 6; j++) e->d.ctrl[j] = qpos[j]; /* the arm's servos hold the given joints; a finger actuator keeps its ctrl */
  e->traj_active = 0;
  e->reset_pending = 0;
  om_forward(&e->m, &e->d);
}
void om_debug_get_gripper(const om_env* e, double* theta_vel) {
  if (e->cfg.task == OM_TASK_BUTTON_PUSH && e->cfg.gripper_model == OM_GRIPPER_ARTICULATED) { theta_vel[0] = e->d.qpos[6]; theta_vel[1] = e->d.qvel[6]; return; } /* right driver */
  theta_vel[0] = e->gr_theta; theta_vel[1] = e->gr_vel;
}
/* debug hooks for the articulated-gripper tests: model sizes; the constraint rows of the current state (mj_forward's); a geom's
 * world pose; an actuator's ctrl */
void om_debug_model_dims(const om_env* e, int* out /*8*/) {
  out[0] = e->m.nq; out[1] = e->m.nv; out[2] = e->m.nbody; out[3] = e->m.ngeom; out[4] = e->m.neq; out[5] = e->m.nu; out[6] = e->m.njnt; out[7] = e->m.cone;
}
int om_debug_efc(const om_env* e, int maxrows, double* pos, double* J /*[maxrows][nv]*/, int* type, double* force, double* aref, double* D) {
  const int n = e->d.nefc < maxrows ? e->d.nefc : maxrows;
  for (int r = 0; r < n; r++) {
    pos[r] = e->d.efc_pos[r]; type[r] = e->d.efc_type[r]; force[r] = e->d.efc_force[r]; aref[r] = e->d.efc_aref[r]; D[r] = e->d.efc_D[r];
    memcpy(J + (size_t)r * e->m.nv, e->d.efc_J[r], sizeof(double) * e->m.nv);
  }
  return e->d.nefc;
}
void om_debug_geom_pose(const om_env* e, int g, double* pos3, double* mat9) {
  memcpy(pos3, e->d.geom_xpos[g], sizeof(double) * 3);
  memcpy(mat9, e->d.geom_xmat[g], sizeof(double) * 9);
}
void om_debug_geom_shape(const om_env* e, int g, int* type_body /*2*/, double* size3) {
  type_body[0] = e->m.geom_type[g]; type_body[1] = e->m.geom_body[g];
  memcpy(size3, e->m.geom_size[g], sizeof(double) * 3);
}
void om_debug_set_ctrl(om_env* e, int u, double value) { e->d.ctrl[u] = value; }
void om_debug_get_dynamics(const om_env* e, double* M /*[nv][nv]*/, double* qfrc_smooth, double* qacc) {
  for (int i = 0; i < e->m.nv; i++) {
    for (int j = 0; j < e->m.nv; j++) M[(size_t)i * e->m.nv + j] = e->d.M[i][j];
    qfrc_smooth[i] = e->d.qfrc_smooth[i]; qacc[i] = e->d.qacc[i];
  }
}
void om_debug_set_gripper(om_env* e, double theta, double vel) {
  e->gr_theta = theta; e->gr_vel = vel;
  if (e->cfg.task == OM_TASK_BUTTON_PUSH) { gripper_place_tips(e); om_forward(&e->m, &e->d); }
}
void om_debug_substeps(om_env* e, int n) {
  for (int s = 0; s < n; s++) om_physics_step(&e->m, &e->d);
}

/* debug hook for tests: overwrite joint positions / velocities of a robot env (then mj_forward) */
void om_debug_set_robot_state(om_env* e, const double* q, const double* v) {
  memcpy(e->d.qpos, q, sizeof(double) * 6);
  memcpy(e->d.qvel, v, sizeof(double) * 6);
  memset(e->d.qacc_warmstart, 0, sizeof e->d.qacc_warmstart);
  e->reset_pending = 0;
  om_forward(&e->m, &e->d);
}

/* ---------------------------------------------------------------- batch */
struct om_batch { int n; om_env* envs; };

om_batch* om_batch_create(const om_task_config* cfg, int n, uint32_t base_seed) {
  om_batch* b = (om_batch*)malloc(sizeof *b);
  b->n = n;
  b->envs = (om_env*)malloc(sizeof(om_env) * (size_t)n);
  for (int i = 0; i < n; i++) om_env_init(&b->envs[i], cfg, base_seed + (uint32_t)i);
  return b;
}
void om_batch_destroy(om_batch* b) { free(b->envs); free(b); }
om_env* om_batch_env(om_batch* b, int i) { return &b->envs[i]; }
void om_batch_reset(om_batch* b, om_step_out* outs) {
  for (int i = 0; i < b->n; i++) om_env_reset(&b->envs[i], &outs[i]);
}
void om_batch_step(om_batch* b, const double* actions, om_step_out* outs, int nthreads) {
  int A = om_action_dim(b->envs[0].cfg.task);
  if (b->envs[0].cfg.task == OM_TASK_BUTTON_PUSH && b->envs[0].cfg.action_type == OM_ACTION_ABS_EEF) A = 4; /* robot_push_button.py:177 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int i = 0; i < b->n; i++) om_env_step(&b->envs[i], actions + (size_t)i * A, &outs[i]);
}
