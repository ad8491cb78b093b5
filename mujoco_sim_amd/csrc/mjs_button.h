// mjs_button.h — Robot Button-Push fused control-step kernel (BASELINE config 5, physics + task
// logic; the cameras are in mjs_render.h).
//
// Path replaced (reference, paths under /root/reference/mujoco_sim/):
//   environments/tasks/robot_push_button.py:143-157 before_step: gripper.move + servoJ (7-D absolute
//       joint action, the registered default) or servoL (4-D absolute EEF action)
//   per substep: entities/robots/robot.py:261-263 servo interpolation, Physics.step() on the scene of
//       robot_push_button.py:66-108 (UR5e + gripper + wrist camera + static switch), then
//       entities/props/switch.py:51-60,71-72 Switch._update_activation on the touch sensor
//   robot_push_button.py:167-170,205-219 reward / goal / termination / discount,
//   robot_push_button.py:126-134 + switch.py:62-65 initialize_episode (6 uniforms + IK + switch pose)
// Deviation D-1b applies: the 2F-85 is reduced to its driver angle (state rows S_GRIP: angle, velocity; the action's last
// component goes through Robotiq2f85.move's map to the fingers_actuator ctrl); its finger pads are two collision spheres
// (include/mjs_scene_spec.h MJS_G2F85_PROXY_RADIUS, lowest point = the TCP plane) at +-(opening / 2 + r) along the
// gripper's y, so the contacts that exist are sphere-floor, sphere-switch box and sphere-button cylinder per tip,
// pyramidal condim 3. One wavefront per 64 envs (lane per env); the contact / joint-limit rows go through a
// generic in-lane primal Newton solver that is only entered by lanes that have active rows.
#pragma once
#include "mjs_kernel_common.h"
#include "mjs_reach.h"
#include <type_traits>
#include "mjs_push.h"  // the convex-pair machinery (MPR) for the one convex pair of this scene: wrist cylinder - switch box

namespace bp {

using rr::NJ;
constexpr int S_Q = 0, S_V = 6, S_TIME = 12, S_SWITCH = 13, S_GRIP = 16;  // S_GRIP: driver angle, velocity of the reduced 2F-85
constexpr int S_WARM = 18;  // rows 18-23: qacc_warmstart, robust path only (see rr::S_WARM)
constexpr int S_CS = 24, S_SN = 30, STATE_DIM = 36;  // rows 24-35: the carried cos / sin of the joint angles (see rr::S_CS)
constexpr int HOT_ROWS_READ = 18 + 12, HOT_ROWS_WRITTEN = 15 + 12;
constexpr int OBS_DIM = 13, ACT_DIM_JOINT = 7, ACT_DIM_EEF = 4;
enum { FLAG_SWITCH_ACTIVE = 4, FLAG_SWITCH_PRESSED = 8 };

using rr::make_frame;

constexpr int NCS = 6;  // contact slots: finger tip t (0: +y, 1: -y of the gripper frame) x (floor, switch box, button cylinder) = 3 t + k
struct ContactSet {
  int n;             // number of detected contacts (dist <= 0) of the two finger-tip spheres
  bool hit[NCS];     // fixed slots, per tip in MuJoCo's detection order: floor, switch box, button cylinder
  double dist[NCS];
  V3 pos[NCS], nrm[NCS];
  double sgn[NCS];   // +1: the sphere is geom2 (plane-sphere), -1: the sphere is geom1
  bool on_switch[NCS];
};

// collision of one finger-tip sphere (centre c) with floor, switch box, button cylinder (mjc_PlaneSphere, mjc_SphereBox,
// mjc_SphereCylinder; normals geom1 -> geom2) into the slots base .. base + 2; adds to cs.n
MJS_DEV void detect_tip(V3 c, V3 sw, ContactSet& cs, int base) {
  const double rp = MJS_G2F85_PROXY_RADIUS;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    cs.hit[base + k] = false; cs.dist[base + k] = 0; cs.nrm[base + k] = v3(0, 0, 1); cs.pos[base + k] = v3(0, 0, 0);
    cs.sgn[base + k] = k == 0 ? 1.0 : -1.0; cs.on_switch[base + k] = k != 0;
  }
  auto add = [&](int k_, double dist, V3 n) {
    const int k = base + k_;
    cs.n++;
    cs.hit[k] = true; cs.dist[k] = dist; cs.nrm[k] = n;
    cs.pos[k] = cs.sgn[k] > 0 ? madd(c, -(rp + 0.5 * dist), n) : madd(c, rp + 0.5 * dist, n);
  };
  if (!(c.z > rp)) add(0, c.z - rp, v3(0, 0, 1));
  {  // box: clamp the centre into the box
    const double h = MJS_SW_BOX_HALF;
    V3 bc = v3(sw.x, sw.y, sw.z + MJS_SW_BOX_HALF);
    V3 loc = c - bc;
    V3 cl = v3(clampd(loc.x, -h, h), clampd(loc.y, -h, h), clampd(loc.z, -h, h));
    bool inside = cl.x == loc.x && cl.y == loc.y && cl.z == loc.z;
    if (!inside) {
      V3 v = (bc + cl) - c;
      double len = sqrt(dot(v, v)), dist = len - rp;
      if (!(dist > 0.0)) add(1, dist, len < MJS_MINVAL ? v3(1, 0, 0) : (1.0 / len) * v);
    } else {
      double best = INFINITY, sg = 1;
      int ax = 0;
      const double l[3] = {loc.x, loc.y, loc.z};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (h - l[k] < best) { best = h - l[k]; ax = k; sg = 1; }
        if (h + l[k] < best) { best = h + l[k]; ax = k; sg = -1; }
      }
      V3 n = ax == 0 ? v3(-sg, 0, 0) : ax == 1 ? v3(0, -sg, 0) : v3(0, 0, -sg);
      add(1, -(best + rp), n);
    }
  }
  {  // button cylinder (axis z): side / cap / rim
    const double r2 = MJS_SW_BUTTON_RADIUS, h2 = MJS_SW_BUTTON_HALF;
    V3 cc = v3(sw.x, sw.y, sw.z + MJS_SW_BUTTON_Z);
    V3 vec = c - cc;
    double x = vec.z;
    V3 a = v3(vec.x, vec.y, vec.z - x);
    double a2 = dot(a, a);
    if (fabs(x) <= h2) {
      V3 p = v3(cc.x, cc.y, cc.z + x);
      V3 v = p - c;
      double len = sqrt(dot(v, v)), dist = len - rp - r2;
      if (!(dist > 0.0)) add(2, dist, len < MJS_MINVAL ? v3(1, 0, 0) : (1.0 / len) * v);
    } else if (a2 <= r2 * r2) {
      double sg = x > 0 ? 1.0 : -1.0, dist = fabs(x) - h2 - rp;
      if (!(dist > 0.0)) add(2, dist, v3(0, 0, -sg));
    } else {
      double sg = x > 0 ? 1.0 : -1.0, la = sqrt(a2);
      V3 p = v3(cc.x + a.x / la * r2, cc.y + a.y / la * r2, cc.z + h2 * sg + a.z / la * r2);
      V3 v = p - c;
      double len = sqrt(dot(v, v)), dist = len - rp;
      if (!(dist > 0.0)) add(2, dist, len < MJS_MINVAL ? v3(1, 0, 0) : (1.0 / len) * v);
    }
  }
}

// point of the tool axis at the height of the finger-tip centres (the tips sit at +-(opening / 2 + r) along the gripper's y)
MJS_DEV V3 proxy_centre(const rr::Chain& c) {
  return madd(c.p[6], MJS_UR_FLANGE_POS[1] + MJS_G2F85_TCP_Z - MJS_G2F85_PROXY_RADIUS, c.R[6].cy);
}

// ---- reduced Robotiq 2F-85 (gripper.py:36-98; oracle/om_tasks.c gripper_*) ------------------------------
// One coordinate per gripper: the driver angle of the two equality-coupled fingers. ctrl = Robotiq2f85.move's map of the
// commanded opening, force = the menagerie fingers_actuator, integrated as mj_implicit (fast) integrates a dof with joint
// damping and an affine actuator; opening = the reference's own sine map; the two finger-tip spheres follow it.
struct Grip { double th, vel; };
MJS_DEV double grip_ctrl_of_opening(double finger_distance) {  // gripper.py:77-84, arcsin argument and ctrl clamped
  const double a = clampd((1 - finger_distance / MJS_G2F85_OPEN) * sin(MJS_G2F85_MAX_DRIVER), -1.0, 1.0);
  return clampd(asin(a) / MJS_G2F85_MAX_DRIVER * MJS_G2F85_CTRL_MAX, 0.0, MJS_G2F85_CTRL_MAX);
}
MJS_DEV double grip_opening(double th) { return MJS_G2F85_OPEN * (1 - sin(th) / sin(MJS_G2F85_MAX_DRIVER)); }  // gripper.py:73-75
MJS_DEV void grip_integrate(Grip& g, double ctrl) {
  const double h = MJS_RR_PHYSICS_DT, inertia = 2 * MJS_G2F85_DRIVER_ARMATURE, damping = 2 * MJS_G2F85_DRIVER_DAMPING;
  double F = MJS_G2F85_ACT_GAIN * ctrl - MJS_G2F85_ACT_KP * g.th - MJS_G2F85_ACT_KV * g.vel;
  bool clamped = false;
  if (F > MJS_G2F85_ACT_FORCE) { F = MJS_G2F85_ACT_FORCE; clamped = true; }
  if (F < -MJS_G2F85_ACT_FORCE) { F = -MJS_G2F85_ACT_FORCE; clamped = true; }
  const double f = F - damping * g.vel;
  g.vel += h * f / (inertia + h * (damping + (clamped ? 0.0 : MJS_G2F85_ACT_KV)));
  g.th += h * g.vel;
}
// both finger tips against the scene: slots 0-2 = tip +y, 3-5 = tip -y (gripper frame y = -wrist_3 z)
MJS_DEV void detect_contacts(const rr::Chain& ch, double th, V3 sw, ContactSet& cs) {
  const V3 mid = proxy_centre(ch);
  const double off = 0.5 * grip_opening(th) + MJS_G2F85_PROXY_RADIUS;
  cs.n = 0;
  detect_tip(madd(mid, -off, ch.R[6].cz), sw, cs, 0);
  detect_tip(madd(mid, off, ch.R[6].cz), sw, cs, 3);
}

// The arm's last collision geom is a CYLINDER (MJS_UR_COL_* index 9, on wrist_3) and the switch's base a BOX: the one
// convex-convex pair MuJoCo's collision table evaluates between the arm and the switch besides the finger tips (the capsule
// pairs are not evaluated). Own MPR, shared with the Planar-Push kernel (one contact: normal cylinder -> box,
// depth = portal distance), behind the same bounding-sphere filter as the oracle's collide_convex. With joint actions the
// tool is not top-down and the wrist can reach the switch while the finger tips are 0.19 m away.
struct WristBox {
  bool hit;
  double dist;
  V3 pos, nrm;
};
MJS_DEV WristBox detect_wrist_box(const rr::Chain& ch, V3 sw) {
  const pp::Geom cyl = pp::wrist3_proxy_geom(ch);
  pp::Geom box;
  box.c = v3(sw.x, sw.y, sw.z + MJS_SW_BOX_HALF);
  box.R = M3{v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
  box.s = v3(MJS_SW_BOX_HALF, MJS_SW_BOX_HALF, MJS_SW_BOX_HALF);
  box.box = true;
  box.cat = -1;
  pp::Contact c;
  WristBox w;
  // Exact reject ahead of the MPR run: the switch box is axis-aligned in the world, and the cylinder's extent along world axis k is
  // half |a_k| + radius sqrt(1 - a_k^2) (a = its axis). Disjoint boxes mean disjoint geoms, for which the MPR finds no portal and reports
  // no hit - the same result without its iterations. The bounding-sphere test inside collide_convex alone lets the MPR run whenever
  // the wrist is within 9 cm of the box, and ONE such lane costs its whole wavefront the run in every substep of the robust path
  // (profiles/r04_i_*: 150 -> 7x us per row-free control step there).
  {
    const V3 a = cyl.R.cz, d = cyl.c - box.c;
    const double ex = cyl.s.y * fabs(a.x) + cyl.s.x * sqrt(fmax(0.0, 1.0 - a.x * a.x));
    const double ey = cyl.s.y * fabs(a.y) + cyl.s.x * sqrt(fmax(0.0, 1.0 - a.y * a.y));
    const double ez = cyl.s.y * fabs(a.z) + cyl.s.x * sqrt(fmax(0.0, 1.0 - a.z * a.z));
    constexpr double gap = 1e-9;
    if (fabs(d.x) > ex + MJS_SW_BOX_HALF + gap || fabs(d.y) > ey + MJS_SW_BOX_HALF + gap || fabs(d.z) > ez + MJS_SW_BOX_HALF + gap) {
      w.hit = false; w.dist = 0.0; w.pos = v3(0, 0, 0); w.nrm = v3(0, 0, 1);
      return w;
    }
  }
  w.hit = pp::collide_convex(cyl, box, 1, 0, 0.0, c);
  w.dist = w.hit ? c.dist : 0.0;
  w.pos = w.hit ? c.pos : v3(0, 0, 0);
  w.nrm = w.hit ? c.n : v3(0, 0, 1);
  return w;
}

// Constraint stage of one physics step for a lane that has rows: joint limits + finger-tip sphere contacts.
// STATIC row slots, everything unrolled (no indexed memory): 12 limit slots (2j = lower side of joint j, 2j+1 =
// upper; J = +-e_j) and 6 contact slots (per finger tip: floor, switch box, button: the only pairs of this scene)
// with 4 pyramid edges each, stored as the contact-frame Jacobian (normal, tangent 1, tangent 2)
// so that edge e = Jn +- mu * Jt: the oracle's row order. Primal Newton of mj_solPrimal (cost, gradient, exact
// Hessian, Cholesky, 1-D Newton line search with MuJoCo's stopping rules), cold-started at qacc_smooth.
// Cold path (noinline, works on copies): `Mf` = full symmetric M + armature, `qs` = qfrc_smooth in, += qfrc_constraint out.
// The two finger tips have 6 possible contacts but rarely more than two at once: the stage works on NACT compact slots,
// filled per lane with its ACTIVE contacts in detection order (branch-free selects), so that its unrolled static-slot code
// and register footprint stay those of a 4-contact scene. A lane with more active contacts than slots drops the rest and
// reports MJS_FAULT_UNSUPPORTED_CONTACT (both tips wedged between floor, box and button at once).
constexpr int NACT = 4;
struct ContactRows {
  bool on[NACT];
  double Jc[NACT][3][NJ];  // rows: normal, tangent 1, tangent 2 (already signed: geom2 - geom1)
  double D[NACT], aref[NACT][4];
};
// By-value interface (the arguments travel in registers): through pointers every use of M, q, v was a FLAT load from the
// caller's scratch frame that could not be kept in a register across the stage's own scratch stores.
struct StageIn {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], sw[3], grip_th, M[21] /* lower triangle of M + armature */, qs[NJ];
  double warm[NJ];  // qacc_warmstart: the previous substep's solution, when that substep had rows too
  bool has_warm;
};
struct StageOut { double qs[NJ], qacc[NJ], touch; bool overflow; };
__device__ __noinline__ StageOut constraint_stage(StageIn in) {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], qs[NJ], Mf[NJ][NJ], touch;
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    q[i] = in.q[i]; v[i] = in.v[i]; cs[i] = in.cs[i]; sn[i] = in.sn[i]; qs[i] = in.qs[i];
#pragma unroll
    for (int j = 0; j < NJ; j++) Mf[i][j] = i >= j ? in.M[i * (i + 1) / 2 + j] : in.M[j * (j + 1) / 2 + i];
  }
  const V3 sw = v3(in.sw[0], in.sw[1], in.sw[2]);
  const double grip_th = in.grip_th;
  const double mu = MJS_GEOM_FRICTION_SLIDE;
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  // ---- rows
  bool lon[rr::NLIM], any_lim = false;
  double lD[rr::NLIM], laref[rr::NLIM];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
#pragma unroll
    for (int side = 0; side < 2; side++) {  // mj_instantiateLimit, jnt_margin = 0
      const int k = 2 * j + side;
      const double sgn = side == 0 ? 1.0 : -1.0;
      const double dist = side == 0 ? q[j] - MJS_UR_JNT_RANGE[j][0] : MJS_UR_JNT_RANGE[j][1] - q[j];
      const double imp = impedance_default(dist);
      lon[k] = dist < 0.0;
      lD[k] = 1 / fmax(MJS_MINVAL, (1 - imp) * UR5E_BP_DOF_INVWEIGHT0[j] / imp);
      laref[k] = -B * (sgn * v[j]) - K * imp * dist;
      any_lim = any_lim || lon[k];
    }
  }
  // wave-uniform slot masks: a slot no lane of the wavefront uses costs nothing below (scalar branches)
  const bool use_lim = __any(any_lim);
  bool use_c[NACT], overflow;
  ContactRows cr;
  V3 cpos[NACT];
  bool on_switch[NACT];
  {
    rr::Chain ch;
    rr::fk_cs(cs, sn, ch);
    ContactSet all;
    detect_contacts(ch, grip_th, sw, all);
    // compact the active contacts (detected with dist < 0; dist == margin creates no rows) into the NACT slots
    struct { bool hit[NACT]; double dist[NACT], sgn[NACT]; V3 pos[NACT], nrm[NACT]; bool on_switch[NACT]; } con;
#pragma unroll
    for (int j = 0; j < NACT; j++) { con.hit[j] = false; con.dist[j] = 0; con.sgn[j] = 1; con.pos[j] = v3(0, 0, 0); con.nrm[j] = v3(0, 0, 1); con.on_switch[j] = false; }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < NCS; k++) {
      const bool act = all.hit[k] && all.dist[k] < 0.0;
#pragma unroll
      for (int j = 0; j < NACT; j++) {
        const bool take = act && cnt == j;
        con.hit[j] = take ? true : con.hit[j];
        con.dist[j] = take ? all.dist[k] : con.dist[j];
        con.sgn[j] = take ? all.sgn[k] : con.sgn[j];
        con.pos[j] = take ? all.pos[k] : con.pos[j];
        con.nrm[j] = take ? all.nrm[k] : con.nrm[j];
        con.on_switch[j] = take ? all.on_switch[k] : con.on_switch[j];
      }
      cnt += act ? 1 : 0;
    }
    overflow = cnt > NACT;
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      cr.on[c] = con.hit[c];
      use_c[c] = __any(cr.on[c]);
      cpos[c] = con.pos[c];
      on_switch[c] = con.on_switch[c];
      if (!use_c[c]) continue;
      V3 t1, t2;
      make_frame(con.nrm[c], t1, t2);
      double vel[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        V3 col = con.sgn[c] * cross(rr::joint_axis(ch, j), con.pos[c] - ch.p[j + 1]);  // (jac2 - jac1) column
        cr.Jc[c][0][j] = dot(con.nrm[c], col); cr.Jc[c][1][j] = dot(t1, col); cr.Jc[c][2][j] = dot(t2, col);
#pragma unroll
        for (int r = 0; r < 3; r++) vel[r] += cr.Jc[c][r][j] * v[j];
      }
      const double imp = impedance_default(con.dist[c]);
      const double dA = UR5E_BP_EEF_BODY_INVWEIGHT0[0] + mu * mu * UR5E_BP_EEF_BODY_INVWEIGHT0[0];
      cr.D[c] = 1 / (2 * mu * mu * fmax(MJS_MINVAL, (1 - imp) * dA / imp));
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double ve = vel[0] + ((e & 1) ? -mu : mu) * vel[1 + (e >> 1)];
        cr.aref[c][e] = -B * ve - K * imp * con.dist[c];
      }
    }
  }
  // ---- Newton
  double L[NJ][NJ], a[NJ], a_s[NJ], Ma[NJ], ljar[rr::NLIM], lforce[rr::NLIM], cjar[NACT][4], cforce[NACT][4];
  bool lact[rr::NLIM], cact[NACT][4];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j < NJ; j++) L[i][j] = Mf[i][j];
    a_s[i] = qs[i];
  }
  rr::chol6(L);
  rr::chol6_solve(L, a_s);  // qacc_smooth
#pragma unroll
  for (int i = 0; i < NJ; i++) a[i] = a_s[i];
  auto edge_values = [&](const double* x, int c, double* out4) {  // J_edge . x for the 4 edges of slot c
    double u[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int j = 0; j < NJ; j++) u[r] += cr.Jc[c][r][j] * x[j];
    }
    out4[0] = u[0] + mu * u[1]; out4[1] = u[0] - mu * u[1]; out4[2] = u[0] + mu * u[2]; out4[3] = u[0] - mu * u[2];
  };
  auto refresh = [&]() {
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m += Mf[i][k] * a[k];
      Ma[i] = m;
    }
    if (use_lim) {
#pragma unroll
      for (int k = 0; k < rr::NLIM; k++) ljar[k] = -laref[k] + ((k & 1) ? -a[k >> 1] : a[k >> 1]);
    }
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (!use_c[c]) continue;
      double ja[4];
      edge_values(a, c, ja);
#pragma unroll
      for (int e = 0; e < 4; e++) cjar[c][e] = ja[e] - cr.aref[c][e];
    }
  };
  auto update = [&]() {
    double cost = 0;
    if (use_lim) {
#pragma unroll
      for (int k = 0; k < rr::NLIM; k++) {
        const bool act = lon[k] && ljar[k] < 0;
        lact[k] = act;
        lforce[k] = act ? -lD[k] * ljar[k] : 0.0;
        if (act) cost += 0.5 * lD[k] * ljar[k] * ljar[k];
      }
    }
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (!use_c[c]) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool act = cr.on[c] && cjar[c][e] < 0;
        cact[c][e] = act;
        cforce[c][e] = act ? -cr.D[c] * cjar[c][e] : 0.0;
        if (act) cost += 0.5 * cr.D[c] * cjar[c][e] * cjar[c][e];
      }
    }
    double gauss = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) gauss += (Ma[i] - qs[i]) * (a[i] - a_s[i]);
    return cost + 0.5 * gauss;
  };
  auto constraint_force = [&](double* f) {  // J^T force
#pragma unroll
    for (int i = 0; i < NJ; i++) f[i] = use_lim ? lforce[2 * i] - lforce[2 * i + 1] : 0.0;
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (!use_c[c]) continue;
      const double fn = cforce[c][0] + cforce[c][1] + cforce[c][2] + cforce[c][3];
      const double f1 = mu * (cforce[c][0] - cforce[c][1]), f2 = mu * (cforce[c][2] - cforce[c][3]);
#pragma unroll
      for (int i = 0; i < NJ; i++) f[i] += fn * cr.Jc[c][0][i] + f1 * cr.Jc[c][1][i] + f2 * cr.Jc[c][2][i];
    }
  };
  refresh();
  double cost = update();
  // mj_fwdConstraint's warm start: begin at qacc_warmstart when its cost is lower than qacc_smooth's (a contact that lasts
  // starts the Newton iteration at the previous substep's solution and usually needs a single iteration)
  if (__any(in.has_warm)) {
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = in.has_warm ? in.warm[i] : a_s[i];
    refresh();
    const double cost_w = update();
    const bool keep = in.has_warm && !(cost < cost_w);  // ties keep the warm start, as the oracle's trial loop does
    if (__any(!keep)) {  // some lane goes back to qacc_smooth (the evaluation state belongs to the chosen start)
#pragma unroll
      for (int i = 0; i < NJ; i++) a[i] = keep ? a[i] : a_s[i];
      refresh();
      cost = update();
    } else
      cost = cost_w;
  }
  const double scale = 1 / (UR5E_BP_MEANINERTIA * NJ);
#pragma unroll 1
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    double grad[NJ], search[NJ], Mv[NJ], H[NJ][NJ], fc[NJ];
    constraint_force(fc);
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      grad[i] = Ma[i] - qs[i] - fc[i];
      search[i] = -grad[i];
#pragma unroll
      for (int j = 0; j <= i; j++) H[i][j] = Mf[i][j];
      if (use_lim && lact[2 * i]) H[i][i] += lD[2 * i];
      if (use_lim && lact[2 * i + 1]) H[i][i] += lD[2 * i + 1];
    }
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (!use_c[c]) continue;
      // sum over active edges of D (Jn + s mu Jt_k)(Jn + s mu Jt_k)^T
      const double n0 = cact[c][0], n1 = cact[c][1], n2 = cact[c][2], n3 = cact[c][3];
      const double wn = cr.D[c] * (n0 + n1 + n2 + n3), w1 = cr.D[c] * mu * (n0 - n1), w2 = cr.D[c] * mu * (n2 - n3);
      const double w11 = cr.D[c] * mu * mu * (n0 + n1), w22 = cr.D[c] * mu * mu * (n2 + n3);
      if (wn != 0.0) {
#pragma unroll
        for (int i = 0; i < NJ; i++) {
          const double jn = cr.Jc[c][0][i], j1 = cr.Jc[c][1][i], j2 = cr.Jc[c][2][i];
          const double rn = wn * jn + w1 * j1 + w2 * j2, r1 = w1 * jn + w11 * j1, r2 = w2 * jn + w22 * j2;
#pragma unroll
          for (int j = 0; j <= i; j++) H[i][j] += rn * cr.Jc[c][0][j] + r1 * cr.Jc[c][1][j] + r2 * cr.Jc[c][2][j];
        }
      }
    }
    if (!rr::chol6(H)) break;
    rr::chol6_solve(H, search);
    double g1 = 0, g2 = 0, snorm = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m += Mf[i][k] * search[k];
      Mv[i] = m;
    }
#pragma unroll
    for (int i = 0; i < NJ; i++) { g1 += search[i] * (Ma[i] - qs[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
    double cjv[NACT][4];
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (use_c[c]) edge_values(search, c, cjv[c]);
    }
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale;
    double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 50; it++) {
      double d1 = g1 + alpha * g2, d2 = g2;
      if (use_lim) {
#pragma unroll
        for (int k = 0; k < rr::NLIM; k++) {
          const double jv = (k & 1) ? -search[k >> 1] : search[k >> 1];
          const double x = ljar[k] + alpha * jv;
          if (lon[k] && x < 0) { d1 += lD[k] * x * jv; d2 += lD[k] * jv * jv; }
        }
      }
#pragma unroll
      for (int c = 0; c < NACT; c++) {
        if (!use_c[c]) continue;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double x = cjar[c][e] + alpha * cjv[c][e];
          if (cr.on[c] && x < 0) { d1 += cr.D[c] * x * cjv[c][e]; d2 += cr.D[c] * cjv[c][e] * cjv[c][e]; }
        }
      }
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      if (d2 <= 0) break;
      double next = alpha + (-d1 / d2);
      if (!(next > lo && next < hi)) next = isfinite(hi) ? 0.5 * (lo + hi) : (alpha > 0 ? 2 * alpha : 1.0);
      if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
      alpha = next;
    }
    if (alpha == 0) break;
#pragma unroll
    for (int i = 0; i < NJ; i++) { a[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
    if (use_lim) {
#pragma unroll
      for (int k = 0; k < rr::NLIM; k++) ljar[k] += alpha * ((k & 1) ? -search[k >> 1] : search[k >> 1]);
    }
#pragma unroll
    for (int c = 0; c < NACT; c++) {
      if (!use_c[c]) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) cjar[c][e] += alpha * cjv[c][e];
    }
    const double oldcost = cost;
    cost = update();
    constraint_force(fc);
    double gn = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      const double g = Ma[i] - qs[i] - fc[i];
      gn += g * g;
    }
    if (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE) break;
  }
  double fc[NJ];
  constraint_force(fc);
#pragma unroll
  for (int j = 0; j < NJ; j++) qs[j] += fc[j];
  // touch sensor (mj_sensorAcc): normal forces of contacts with the switch whose point is inside the site
  const double sr = MJS_SW_BUTTON_RADIUS * MJS_SW_SITE_SCALE, sh = MJS_SW_BUTTON_HALF * MJS_SW_SITE_SCALE;
  touch = 0;
#pragma unroll
  for (int c = 0; c < NACT; c++) {
    const V3 loc = cpos[c] - v3(sw.x, sw.y, sw.z + MJS_SW_BUTTON_Z);
    const bool in_site = !(loc.x * loc.x + loc.y * loc.y > sr * sr || fabs(loc.z) > sh);
    if (use_c[c] && cr.on[c] && on_switch[c] && in_site) touch += cforce[c][0] + cforce[c][1] + cforce[c][2] + cforce[c][3];
  }
  StageOut out;
#pragma unroll
  for (int j = 0; j < NJ; j++) { out.qs[j] = qs[j]; out.qacc[j] = a[j]; }
  out.touch = touch;
  out.overflow = overflow;
  return out;
}

// ---- the LEAN stage: at most two active finger-tip contacts, no joint-limit rows ------------------------------------------
// What almost every constraint solve of this task is (a tip or both on the floor / the switch after a reset, the button being
// pressed): the static-slot stage above is sized for 12 limit slots + 4 contacts and executes ~10 k instructions per call with
// a quarter of them register spills (14 k static instructions, 2.4 k v_accvgpr moves, 1.2 k scratch accesses: the post-reset
// tail of the launch is 20 such calls in a row on the slowest wavefront, profiles/r03_e_*). This one keeps the contact FRAME
// formulation of the general stage (per contact: W = Jc a + B Jc v, Jc search, D, K imp dist; edges formed on the fly), takes
// the two contacts from the caller's detection instead of detecting again, has no slot compaction, no limit arrays, and uses
// reciprocal / reciprocal-square-root refinements instead of IEEE division sequences. Same problem, same warm start, same
// stopping rules: it returns the same minimiser (to the solver's own rounding), so which of the three stages runs is invisible
// beyond the last bits.
template <int NC>
struct SmallIn {
  double v[NJ], cs[NJ], sn[NJ], M[21], qs[NJ], warm[NJ], sw[3];
  double pos[NC][3], nrm[NC][3], sgn[NC], dist[NC];
  bool on[NC], on_switch[NC], has_warm;
};
template <int NC>  // NC = 2 (the usual case) or 4 (both tips on two surfaces each)
__device__ __noinline__ StageOut constraint_stage_small(SmallIn<NC> in) {
  const double mu = MJS_GEOM_FRICTION_SLIDE;
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  double Jc[NC][3][NJ], D[NC], kid[NC], bv[NC][3], W[NC][3], us[NC][3];
  bool use[NC];  // wave-uniform slot masks: a slot no lane of the wavefront fills costs a scalar branch
#pragma unroll
  for (int c = 0; c < NC; c++) use[c] = c == 0 || __any(in.on[c]);
  {
    rr::Chain ch;
    rr::fk_cs(in.cs, in.sn, ch);
#pragma unroll
    for (int c = 0; c < NC; c++) {
      if (!use[c]) continue;
      const V3 n = v3(in.nrm[c][0], in.nrm[c][1], in.nrm[c][2]), pos = v3(in.pos[c][0], in.pos[c][1], in.pos[c][2]);
      // mju_makeFrame
      V3 y = (n.y > -0.5 && n.y < 0.5) ? v3(0, 1, 0) : v3(0, 0, 1);
      y = madd(y, -dot(n, y), n);
      const V3 t1 = rr::rsqrt_fast(dot(y, y)) * y, t2 = cross(n, t1);
      double vel[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const V3 col = in.sgn[c] * cross(rr::joint_axis(ch, j), pos - ch.p[j + 1]);  // (jac2 - jac1) column
        Jc[c][0][j] = dot(n, col); Jc[c][1][j] = dot(t1, col); Jc[c][2][j] = dot(t2, col);
#pragma unroll
        for (int r = 0; r < 3; r++) vel[r] = fma(Jc[c][r][j], in.v[j], vel[r]);
      }
      const double imp = impedance_default(in.dist[c]);
      const double dA = UR5E_BP_EEF_BODY_INVWEIGHT0[0] + mu * mu * UR5E_BP_EEF_BODY_INVWEIGHT0[0];
      D[c] = imp * rr::rcp_fast(2 * mu * mu * fmax(MJS_MINVAL * imp, (1 - imp) * dA));  // 1 / (2 mu^2 max(MINVAL, (1 - imp) dA / imp))
      kid[c] = K * imp * in.dist[c];
#pragma unroll
      for (int r = 0; r < 3; r++) bv[c][r] = B * vel[r];
    }
  }
  auto Mat = [&](int i, int j) { return i >= j ? in.M[i * (i + 1) / 2 + j] : in.M[j * (j + 1) / 2 + i]; };
  double a[NJ], a_s[NJ], Ma[NJ], fc[NJ], H[NJ][NJ], touch = 0;
  {
    double L[NJ][NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) L[i][j] = in.M[i * (i + 1) / 2 + j];
      a_s[i] = in.qs[i];
    }
    rr::chol6(L);
    rr::chol6_solve(L, a_s);  // qacc_smooth
  }
  const double sr = MJS_SW_BUTTON_RADIUS * MJS_SW_SITE_SCALE, sh = MJS_SW_BUTTON_HALF * MJS_SW_SITE_SCALE;
  // cost at `a`, J^T force, the frame residuals W and (need_H) the Hessian's lower triangle
  auto eval = [&](bool need_H) -> double {
    double cost = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m = fma(Mat(i, k), a[k], m);
      Ma[i] = m;
      fc[i] = 0;
#pragma unroll
      for (int j = 0; j <= i; j++) H[i][j] = in.M[i * (i + 1) / 2 + j];
    }
    touch = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      if (!use[c]) continue;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double u = bv[c][r];
#pragma unroll
        for (int j = 0; j < NJ; j++) u = fma(Jc[c][r][j], a[j], u);
        W[c][r] = u;
      }
      double f[4], nact[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double x = W[c][0] + ((e & 1) ? -mu : mu) * W[c][1 + (e >> 1)] + kid[c];
        const bool act = in.on[c] && x < 0;
        nact[e] = act ? 1.0 : 0.0;
        f[e] = act ? -D[c] * x : 0.0;
        if (act) cost += 0.5 * D[c] * x * x;
      }
      const double fn = f[0] + f[1] + f[2] + f[3], f1 = mu * (f[0] - f[1]), f2 = mu * (f[2] - f[3]);
#pragma unroll
      for (int i = 0; i < NJ; i++) fc[i] += fn * Jc[c][0][i] + f1 * Jc[c][1][i] + f2 * Jc[c][2][i];
      {  // touch sensor (mj_sensorAcc): normal force of a contact with the switch whose point lies inside the site
        const double lx = in.pos[c][0] - in.sw[0], ly = in.pos[c][1] - in.sw[1], lz = in.pos[c][2] - (in.sw[2] + MJS_SW_BUTTON_Z);
        if (in.on[c] && in.on_switch[c] && !(lx * lx + ly * ly > sr * sr || fabs(lz) > sh)) touch += fn;
      }
      if (need_H) {
        const double wn = D[c] * (nact[0] + nact[1] + nact[2] + nact[3]), w1 = D[c] * mu * (nact[0] - nact[1]), w2 = D[c] * mu * (nact[2] - nact[3]);
        const double w11 = D[c] * mu * mu * (nact[0] + nact[1]), w22 = D[c] * mu * mu * (nact[2] + nact[3]);
#pragma unroll
        for (int i = 0; i < NJ; i++) {
          const double jn = Jc[c][0][i], j1 = Jc[c][1][i], j2 = Jc[c][2][i];
          const double rn = wn * jn + w1 * j1 + w2 * j2, r1 = w1 * jn + w11 * j1, r2 = w2 * jn + w22 * j2;
#pragma unroll
          for (int j = 0; j <= i; j++) H[i][j] += rn * Jc[c][0][j] + r1 * Jc[c][1][j] + r2 * Jc[c][2][j];
        }
      }
    }
    double gauss = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) gauss += (Ma[i] - in.qs[i]) * (a[i] - a_s[i]);
    return cost + 0.5 * gauss;
  };
#pragma unroll
  for (int i = 0; i < NJ; i++) a[i] = a_s[i];
  if (__any(in.has_warm)) {  // mj_fwdConstraint: the cheaper of qacc_warmstart and qacc_smooth, ties to the warm start
    const double cost_s = eval(false);
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = in.has_warm ? in.warm[i] : a_s[i];
    const double cost_w = eval(false);
    const bool keep = in.has_warm && !(cost_s < cost_w);
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = keep ? a[i] : a_s[i];
  }
  const double scale = 1 / (UR5E_BP_MEANINERTIA * NJ);
  double oldcost = 0;
#pragma unroll 1
  for (int iter = 0; iter <= MJS_SOLVER_ITERATIONS; iter++) {
    const double cost = eval(true);
    double grad[NJ], gn = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) { grad[i] = Ma[i] - in.qs[i] - fc[i]; gn = fma(grad[i], grad[i], gn); }
    if (iter > 0 && (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE)) break;
    if (iter == MJS_SOLVER_ITERATIONS) break;
    oldcost = cost;
    double search[NJ], Mv[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) search[i] = -grad[i];
    if (!rr::chol6(H)) break;
    rr::chol6_solve(H, search);
    double g1 = 0, g2 = 0, snorm = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m = fma(Mat(i, k), search[k], m);
      Mv[i] = m;
    }
#pragma unroll
    for (int i = 0; i < NJ; i++) { g1 += search[i] * (Ma[i] - in.qs[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      if (!use[c]) continue;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double u = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++) u = fma(Jc[c][r][j], search[j], u);
        us[c][r] = u;
      }
    }
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale;
    double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 50; it++) {
      double d1 = g1 + alpha * g2, d2 = g2;
#pragma unroll
      for (int c = 0; c < NC; c++) {
        if (!use[c]) continue;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double sm = (e & 1) ? -mu : mu;
          const double jar = W[c][0] + sm * W[c][1 + (e >> 1)] + kid[c], jv = us[c][0] + sm * us[c][1 + (e >> 1)];
          const double x = jar + alpha * jv;
          if (in.on[c] && x < 0) { d1 += D[c] * x * jv; d2 += D[c] * jv * jv; }
        }
      }
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      if (d2 <= 0) break;
      double next = alpha - d1 * rr::rcp_fast(d2);
      if (!(next > lo && next < hi)) next = isfinite(hi) ? 0.5 * (lo + hi) : (alpha > 0 ? 2 * alpha : 1.0);
      if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
      alpha = next;
    }
    if (alpha == 0) break;
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = fma(alpha, search[i], a[i]);
  }
  StageOut out;
#pragma unroll
  for (int j = 0; j < NJ; j++) { out.qs[j] = in.qs[j] + fc[j]; out.qacc[j] = a[j]; }
  out.touch = touch;
  out.overflow = false;
  return out;
}

// The scene as the general constraint stage (mjs_arm_stage.h) sees it: the arm's own geoms against the floor come from the
// stage itself; this adds the two finger-tip spheres' contacts (body = the gripper lump on wrist_3: all six joints move it)
// in MuJoCo's pair order (geom ids: floor, arm, tip +y, tip -y, switch box, button): floor - tip 0, floor - tip 1,
// wrist cylinder - box (detected by the caller: one MPR run per substep), tip 0 - box, tip 0 - button, tip 1 - box,
// tip 1 - button; and the touch sensor's site test.
struct SceneButton {
  struct Extra { double sw[3], grip_th; bool wb_hit; double wb_dist, wb_pos[3], wb_nrm[3]; };
  MJS_DEV static double solver_scale(Extra) { return 1 / (UR5E_BP_MEANINERTIA * NJ); }
  MJS_DEV static double dof_invweight(int j) { return UR5E_BP_DOF_INVWEIGHT0[j]; }
  MJS_DEV static double link_invweight(int b) { return UR5E_BP_LINK_BODY_INVWEIGHT0[b]; }
  template <class E>
  MJS_DEV static void extra_contacts(const rr::Chain& ch, Extra ex, E emit) {
    ContactSet all;
    detect_contacts(ch, ex.grip_th, v3(ex.sw[0], ex.sw[1], ex.sw[2]), all);
    constexpr int order[NCS] = {0, 3, 1, 2, 4, 5};
#pragma unroll
    for (int o = 0; o < NCS; o++) {
      const int k = order[o];
      if (o == 2 && ex.wb_hit)
        emit(NJ, v3(ex.wb_pos[0], ex.wb_pos[1], ex.wb_pos[2]), v3(ex.wb_nrm[0], ex.wb_nrm[1], ex.wb_nrm[2]), -1.0, ex.wb_dist, UR5E_BP_LINK_BODY_INVWEIGHT0[6], true);
      if (all.hit[k]) emit(NJ, all.pos[k], all.nrm[k], all.sgn[k], all.dist[k], UR5E_BP_EEF_BODY_INVWEIGHT0[0], all.on_switch[k]);
    }
  }
  MJS_DEV static bool in_touch_site(Extra ex, V3 pos) {
    const double sr = MJS_SW_BUTTON_RADIUS * MJS_SW_SITE_SCALE, sh = MJS_SW_BUTTON_HALF * MJS_SW_SITE_SCALE;
    const V3 loc = pos - v3(ex.sw[0], ex.sw[1], ex.sw[2] + MJS_SW_BUTTON_Z);
    return !(loc.x * loc.x + loc.y * loc.y > sr * sr || fabs(loc.z) > sh);
  }
};

// mj_fwdConstraint's warm start across Physics.step() calls: qacc_warmstart = the previous step's SOLVER acceleration, i.e.
// the stage's result, or (M + armature)^-1 qfrc_smooth of a step without rows — kept as (pM, pqs) and evaluated lazily, only
// when the next step turns out to have rows (rr::solo_control_step does the same).
struct WarmState {
  double warm[NJ], pM[21], pqs[NJ];
  bool has_warm, lazy;
};

// One Physics.step() of a lane on the robust path: smooth dynamics (generated, Button-Push payload variant), detection of
// every row this scene can have (joint ranges, the arm's geoms on the floor, the finger tips on floor / switch box / button),
// the constraint stage for the lanes that have rows (a lean stage for up to four tip contacts, the static-slot stage with the
// joint limits, the general stage rr::gen_stage as soon as the lane has a link in the floor or the wrist on the switch box),
// then the implicitfast solve. Returns the integrator's acceleration.
MJS_DEV void physics_forces(const double* q, const double* v, const double* ctrl, const double* cs, const double* sn, V3 sw, double grip_th,
                            double* qacc_int, double& touch, int& ncon_proxy, bool& rows_active, bool& slot_overflow, WarmState& w, rr::Ws ws) {
  touch = 0;
  // cheap in-line detection: any joint beyond its range, an arm geom in the floor, an active contact of a finger-tip sphere?
  bool rows = rr::joint_outside_range(q);
  rr::Chain ch;
  rr::fk_cs(cs, sn, ch);
  const WristBox wb = detect_wrist_box(ch, sw);
  const bool arm = !(rr::min_floor_clearance(ch) >= 0.0) || (wb.hit && wb.dist < 0.0);  // rows only the general stage knows
  ContactSet con;
  detect_contacts(ch, grip_th, sw, con);
  ncon_proxy = con.n + (wb.hit ? 1 : 0);
  int n_tip = 0;
#pragma unroll
  for (int c = 0; c < NCS; c++) n_tip += (con.hit[c] && con.dist[c] < 0.0) ? 1 : 0;
  rows = rows || arm || n_tip > 0;
  // Rows are rare. Every out-of-line function below is CALLED BY THE WHOLE WAVEFRONT behind a wave-uniform branch and its
  // results are taken by per-lane selects (rr::solo_control_step says why: hipcc's split copies ahead of the exec restore of
  // a divergent join next to a call, _isa_lint.py).
  // 1. the warm start a row-free step left behind as (pM, pqs), needed now: qacc_warmstart = pM^-1 pqs
  const bool eval_lazy = rows && w.lazy;
  if (__any(eval_lazy)) {
    double wl[NJ];
    rr::smooth_acceleration(w.pM, w.pqs, wl);
#pragma unroll
    for (int j = 0; j < NJ; j++) w.warm[j] = eval_lazy ? wl[j] : w.warm[j];
    w.has_warm = w.has_warm || eval_lazy;
  }
  // 2. smooth dynamics; M + armature and qfrc_smooth go straight into (pM, pqs): what a row-free step leaves for the next one
  double bias[NJ], A[NJ][NJ], rhs[NJ], Dinv[NJ], fact[NJ];
  double* const M = w.pM;
  ur5e_bp_M_gen(cs, sn, M);
  ur5e_bp_bias_gen(cs, sn, v, bias);
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) A[i][j] = M[i * (i + 1) / 2 + j];
    M[i * (i + 1) / 2 + i] += MJS_UR_ARMATURE;
  }
  const int clamped = rr::actuator_forces(q, v, ctrl, fact);
#pragma unroll
  for (int j = 0; j < NJ; j++) { rhs[j] = fact[j] - bias[j]; w.pqs[j] = rhs[j]; }
  w.lazy = !rows;
  // 3. the constraint stage of the lanes with rows. Which stage? Decided per LANE from the lane's own rows (the stages agree to
  // the solver's rounding, not to the bit: a choice that looked at the other lanes of the wavefront would make an env's last
  // bits depend on its neighbours and break the bitwise shard invariance the multi-GPU path relies on); a wavefront whose
  // lanes differ runs their stages in turn and each lane keeps the result of its own. Exactly one of the four per lane: a lane
  // that has taken its stage's result feeds the later stages changed inputs and discards what they make of them.
  if (__any(rows)) {
    const bool lean = !(arm || n_tip > 4 || rr::joint_outside_range(q));
    const bool want_lean2 = rows && lean && n_tip <= 2, want_lean4 = rows && lean && n_tip > 2;
    const bool want_general = rows && !lean && arm, want_static = rows && !lean && !arm;
    auto run_lean = [&](auto tag, bool mine) {
      constexpr int NC = decltype(tag)::value;
      SmallIn<NC> in;
#pragma unroll
      for (int i = 0; i < NJ; i++) { in.v[i] = v[i]; in.cs[i] = cs[i]; in.sn[i] = sn[i]; in.qs[i] = rhs[i]; in.warm[i] = w.warm[i]; }
#pragma unroll
      for (int k = 0; k < 21; k++) in.M[k] = M[k];
      in.sw[0] = sw.x; in.sw[1] = sw.y; in.sw[2] = sw.z;
      in.has_warm = w.has_warm;
      // the lane's active tip contacts in detection order, by branch-free selects
#pragma unroll
      for (int j = 0; j < NC; j++) {
        in.on[j] = false; in.on_switch[j] = false; in.sgn[j] = 1.0; in.dist[j] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { in.pos[j][k] = 0.0; in.nrm[j][k] = k == 2 ? 1.0 : 0.0; }
      }
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < NCS; c++) {
        const bool act = con.hit[c] && con.dist[c] < 0.0;
#pragma unroll
        for (int j = 0; j < NC; j++) {
          const bool take = act && cnt == j;
          in.on[j] = take ? true : in.on[j];
          in.on_switch[j] = take ? con.on_switch[c] : in.on_switch[j];
          in.sgn[j] = take ? con.sgn[c] : in.sgn[j];
          in.dist[j] = take ? con.dist[c] : in.dist[j];
          in.pos[j][0] = take ? con.pos[c].x : in.pos[j][0]; in.pos[j][1] = take ? con.pos[c].y : in.pos[j][1]; in.pos[j][2] = take ? con.pos[c].z : in.pos[j][2];
          in.nrm[j][0] = take ? con.nrm[c].x : in.nrm[j][0]; in.nrm[j][1] = take ? con.nrm[c].y : in.nrm[j][1]; in.nrm[j][2] = take ? con.nrm[c].z : in.nrm[j][2];
        }
        cnt += act ? 1 : 0;
      }
      const StageOut out = constraint_stage_small<NC>(in);
#pragma unroll
      for (int i = 0; i < NJ; i++) { rhs[i] = mine ? out.qs[i] : rhs[i]; w.warm[i] = mine ? out.qacc[i] : w.warm[i]; }
      touch = mine ? out.touch : touch;
    };
    // (the 4-slot instance does a lane with <= 2 contacts bit for bit like the 2-slot one — unused slots are skipped, not
    // added as zeros — so it serves both kinds when the wavefront has both: one call instead of two, and still nothing in a
    // lane's result depends on its neighbours; test_contact_tasks_shard_invariance)
    if (__any(want_lean4)) run_lean(std::integral_constant<int, 4>{}, want_lean2 || want_lean4);
    else if (__any(want_lean2)) run_lean(std::integral_constant<int, 2>{}, want_lean2);
    if (__any(want_general)) {
      rr::GenStageIn gi;
#pragma unroll
      for (int i = 0; i < NJ; i++) { gi.q[i] = q[i]; gi.v[i] = v[i]; gi.cs[i] = cs[i]; gi.sn[i] = sn[i]; gi.qs[i] = rhs[i]; gi.warm[i] = w.warm[i]; }
#pragma unroll
      for (int k = 0; k < 21; k++) gi.M[k] = M[k];
      gi.has_warm = w.has_warm;
      const rr::GenStageOut go = rr::gen_stage<SceneButton>(
          gi, SceneButton::Extra{{sw.x, sw.y, sw.z}, grip_th, wb.hit, wb.dist, {wb.pos.x, wb.pos.y, wb.pos.z}, {wb.nrm.x, wb.nrm.y, wb.nrm.z}}, ws);
#pragma unroll
      for (int i = 0; i < NJ; i++) { rhs[i] = want_general ? go.qs[i] : rhs[i]; w.warm[i] = want_general ? go.qacc[i] : w.warm[i]; }
      touch = want_general ? go.touch : touch;
      slot_overflow = slot_overflow || (want_general && go.overflow);
    }
    if (__any(want_static)) {
      StageIn in;
#pragma unroll
      for (int i = 0; i < NJ; i++) { in.q[i] = q[i]; in.v[i] = v[i]; in.cs[i] = cs[i]; in.sn[i] = sn[i]; in.qs[i] = rhs[i]; in.warm[i] = w.warm[i]; }
#pragma unroll
      for (int k = 0; k < 21; k++) in.M[k] = M[k];
      in.sw[0] = sw.x; in.sw[1] = sw.y; in.sw[2] = sw.z;
      in.grip_th = grip_th;
      in.has_warm = w.has_warm;
      const StageOut out = constraint_stage(in);
#pragma unroll
      for (int i = 0; i < NJ; i++) { rhs[i] = want_static ? out.qs[i] : rhs[i]; w.warm[i] = want_static ? out.qacc[i] : w.warm[i]; }
      touch = want_static ? out.touch : touch;
      slot_overflow = slot_overflow || (want_static && out.overflow);
    }
    w.has_warm = w.has_warm || rows;
    rows_active = rows_active || rows;
  }
  rr::factor_system(A, clamped, Dinv);
  rr::udu_solve(A, Dinv, rhs);
#pragma unroll
  for (int j = 0; j < NJ; j++) qacc_int[j] = rhs[j];
}

// can this lane get constraint rows within the next `h` seconds? A joint within reach of its range, or a finger tip
// within reach of the floor / the switch, where "reach" is what the force-clamped servos add from rest
// (12 h^2 m: 0.12 m per 0.1 s control step; 60 h^2 rad for a joint) plus the distance covered at the CURRENT speed (the
// stand-in's Cartesian |J(q) v|; mjs_set_state can inject any velocity). The kernel asks for the whole control step first;
// a workgroup that fails that asks again before every segment of 5 substeps (a fast swing far from everything passes
// these) and only hands over to the robust path from the segment that fails. The fast path also re-checks its final state
// (a posteriori, in the kernel): a joint outside its range or a touching stand-in after a row-free step is reported as
// MJS_FAULT_FASTPATH_VIOLATED.
constexpr int SEGMENT_SUBSTEPS = 5;
static_assert(MJS_RR_NSUB % SEGMENT_SUBSTEPS == 0, "segments tile the control step");
MJS_DEV bool rows_possible(const double* q, const double* v, const rr::Chain& ch, V3 sw, Grip grip, double grip_ctrl, double h) {
  const V3 mid = proxy_centre(ch);  // the tool-axis point between the finger tips
  V3 vel = v3(0, 0, 0);
  double spin = 0;
  bool near = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    vel += v[j] * cross(rr::joint_axis(ch, j), mid - ch.p[j + 1]);
    spin += fabs(v[j]);
    const double margin = 60.0 * h * h + h * fabs(v[j]);
    near = near || (q[j] - MJS_UR_JNT_RANGE[j][0] < margin) || (MJS_UR_JNT_RANGE[j][1] - q[j] < margin);
  }
  // a finger tip sits `off` beside that point along the gripper's y. Within h it moves sideways by at most what the driver
  // can turn (|d off / d theta| <= 0.06 m/rad; velocity now + the force range over the reduced inertia: 300 h^2 rad) and never
  // past the commanded opening (the servo is over-damped; 3 mm cover its clamp chatter), and it is carried round the axis
  // point by the arm's rotation (|omega| <= sum |v_j|)
  const double off = 0.5 * grip_opening(grip.th) + MJS_G2F85_PROXY_RADIUS;
  const double off_cmd = 0.5 * grip_opening(grip_ctrl * (MJS_G2F85_MAX_DRIVER / MJS_G2F85_CTRL_MAX)) + MJS_G2F85_PROXY_RADIUS;
  const double sideways = fmin(fabs(off_cmd - off), 0.06 * (fabs(grip.vel) * h + 300.0 * h * h)) + 0.003;
  const double travel = 12.0 * h * h + h * sqrt(dot(vel, vel)) + sideways + off * spin * h;
  const double reach = 0.08 + travel;
  bool close = false;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const V3 c = madd(mid, t == 0 ? -off : off, ch.R[6].cz);
    const V3 d = c - v3(sw.x, sw.y, sw.z + 0.035);
    close = close || c.z < 0.03 + travel || dot(d, d) < reach * reach;
  }
  // the wrist cylinder against the switch box (bounding spheres 0.045 + 0.044 m): the cylinder sits 0.19 m up the tool axis
  // from the finger tips and is carried round them by the arm's rotation
  {
    const V3 wc = madd(ch.p[6], MJS_UR_COL_POS[MJS_UR_NCOLGEOM - 1][1], ch.R[6].cy);
    const V3 d = wc - v3(sw.x, sw.y, sw.z + MJS_SW_BOX_HALF);
    const double wreach = 0.09 + 12.0 * h * h + h * sqrt(dot(vel, vel)) + 0.2 * spin * h;
    close = close || dot(d, d) < wreach * wreach;
  }
  // the arm's own collision geoms and the floor: the configuration must be clear by rr::CLEAR_MARGIN now (what the arm may do
  // during the step is bounded per control step by the kernel: servo target clear as well, short joint-space travel)
  return near || close || !(rr::min_floor_clearance(ch) >= rr::CLEAR_MARGIN);
}

// Switch._update_activation (switch.py:51-60)
MJS_DEV void switch_update(double touch, uint8_t& flags) {
  bool was = flags & FLAG_SWITCH_PRESSED;
  bool pressed = touch >= MJS_SW_MIN_FORCE && touch <= MJS_SW_MAX_FORCE;
  if (pressed && !was) flags ^= FLAG_SWITCH_ACTIVE;  // flip on the rising edge
  flags = pressed ? (flags | FLAG_SWITCH_PRESSED) : (flags & ~FLAG_SWITCH_PRESSED);
}

MJS_DEV void make_obs(const rr::State& st, const rr::Chain& c, uint8_t flags, double* obs) {
#pragma unroll
  for (int j = 0; j < NJ; j++) obs[j] = st.q[j];  // ur5e/joint_configuration
  V3 tcp = rr::tcp_position(c);
  obs[6] = tcp.x; obs[7] = tcp.y; obs[8] = tcp.z;  // ur5e/tcp_position
  // Switch.get_position: button xpos + 0.5*size[1] added to ALL coordinates (switch.py:86-87)
  obs[9] = st.target[0] + MJS_SW_POSITION_OFFSET;
  obs[10] = st.target[1] + MJS_SW_POSITION_OFFSET;
  obs[11] = st.target[2] + MJS_SW_BUTTON_Z + MJS_SW_POSITION_OFFSET;
  obs[12] = (flags & FLAG_SWITCH_ACTIVE) ? 1.0 : 0.0;
}

// initialize_episode (robot_push_button.py:126-134, switch.py:62-65). st.target holds the switch position.
struct ResetOut {
  rr::State st;
  uint8_t flags;
  int ncon;
};  // the gripper of a fresh episode: driver angle 0 (open), at rest (mj_resetData)
__device__ __noinline__ ResetOut episode_init(DevRng rng, int i, uint8_t old_flags, rr::Ws ws) {
  ResetOut o;
  RngCursor c = rng_open(rng, i);
  double rp[3], q[NJ], zeros[NJ] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; k++) rp[k] = rng_uniform(rng, i, c, MJS_BP_ROBOT_SPACE_LO[k], MJS_BP_ROBOT_SPACE_HI[k]);
  bool ok = rr::tcp_pose_to_joints(rp, zeros, q);
#pragma unroll
  for (int j = 0; j < NJ; j++) { o.st.q[j] = ok ? q[j] : 0.0; o.st.v[j] = 0; }
#pragma unroll
  for (int k = 0; k < 3; k++) o.st.target[k] = rng_uniform(rng, i, c, MJS_BP_SWITCH_SPACE_LO[k], MJS_BP_SWITCH_SPACE_HI[k]);
  rng_close(rng, i, c);
  o.st.time = 0;
  // mj_forward at the reset state (ctrl = q) gives the touch force the switch sees in initialize_episode
  double cs[NJ], sn[NJ], qacc[NJ], touch;
  int ncp;
  bool rows = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) sincos(o.st.q[j], &sn[j], &cs[j]);
  V3 sw = v3(o.st.target[0], o.st.target[1], o.st.target[2]);
  bool overflow_ = false;
  WarmState w_;  // mj_resetData: qacc_warmstart = 0
#pragma unroll
  for (int j = 0; j < NJ; j++) { w_.warm[j] = 0; w_.pqs[j] = 0; }
#pragma unroll
  for (int k = 0; k < 21; k++) w_.pM[k] = 0;
  w_.has_warm = true; w_.lazy = false;
  physics_forces(o.st.q, o.st.v, o.st.q, cs, sn, sw, 0.0, qacc, touch, ncp, rows, overflow_, w_, ws);
  uint8_t f = old_flags & FLAG_SWITCH_PRESSED;  // was_pressed is stale from the previous episode (switch.py:53)
  switch_update(touch, f);                      // with _is_active = False
  o.flags = (uint8_t)(f | FLAG_WARM_VALID);     // mj_forward does not advance qacc_warmstart: the zeros of mj_resetData start the first step
  rr::Chain ch;
  rr::fk_cs(cs, sn, ch);
  o.ncon = rr::count_floor_contacts(ch) + ncp;
  return o;
}

__device__ __noinline__ uint8_t disturb(DevRng rng, int i, uint8_t flags) {
  RngCursor c = rng_open(rng, i);
  double u = rng_uniform(rng, i, c, 0.0, 1.0);
  rng_close(rng, i, c);
  return u < 0.01 ? (uint8_t)(flags & ~FLAG_SWITCH_ACTIVE) : flags;
}

// The 20 substeps of one control step on ONE wavefront with the constraint stage available (robust path: taken
// by a workgroup in which some env may get joint-limit or contact rows during this control step, and by
// kernel_variant = single wave). noinline + by value, as rr::solo_control_step.
struct SoloIn {
  double q[NJ], v[NJ], q0[NJ], q1[NJ], cs[NJ], sn[NJ], time, t0, t1, sw[3];
  uint8_t flags;
  int first_substep;  // the substeps before it were taken on the row-free path
  double grip_th, grip_vel, grip_ctrl;
  double warm[NJ];  // the state's qacc_warmstart rows (first_substep == 0 only)
  bool has_warm;
};
struct SoloOut {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], time, grip_th, grip_vel, warm[NJ];
  uint8_t flags;
  bool bad, rows_active, slot_overflow;
};
__device__ __noinline__ SoloOut solo_control_step(SoloIn in, rr::Ws ws) {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], q0[NJ], q1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) { q[j] = in.q[j]; v[j] = in.v[j]; cs[j] = in.cs[j]; sn[j] = in.sn[j]; q0[j] = in.q0[j]; q1[j] = in.q1[j]; }
  double time = in.time;
  const double t0 = in.t0, t1 = in.t1, inv_span = 1.0 / (in.t1 - in.t0);
  const V3 sw = v3(in.sw[0], in.sw[1], in.sw[2]);
  Grip grip{in.grip_th, in.grip_vel};
  uint8_t flags = in.flags;
  bool bad = false, rows_active = false, slot_overflow = false;
  WarmState w;
#pragma unroll
  for (int j = 0; j < NJ; j++) { w.warm[j] = in.warm[j]; w.pqs[j] = 0; }
#pragma unroll
  for (int k = 0; k < 21; k++) w.pM[k] = 0;
  w.has_warm = in.has_warm; w.lazy = false;
  int ncon_proxy = 0;
  const int first_substep = __builtin_amdgcn_readfirstlane(in.first_substep);  // wave-uniform (the kernel's solo_from): a scalar loop counter
#pragma unroll 1
  for (int s = first_substep; s < MJS_RR_NSUB; s++) {
    double t = fmin(fmax(time, t0), t1);
    double ctrl[NJ], qacc[NJ], touch;
#pragma unroll
    for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
    physics_forces(q, v, ctrl, cs, sn, sw, grip.th, qacc, touch, ncon_proxy, rows_active, slot_overflow, w, ws);
    grip_integrate(grip, in.grip_ctrl);
    double acc2 = 0, dq2 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      acc2 = fma(qacc[j], qacc[j], acc2);
      v[j] += MJS_RR_PHYSICS_DT * qacc[j];
      double dq = MJS_RR_PHYSICS_DT * v[j];
      q[j] += dq;
      dq2 = fma(dq, dq, dq2);
      rr::rotate_small(cs[j], sn[j], dq);
    }
    bad = bad || !(acc2 <= 1e20);
    if (!(dq2 <= 0.01)) {
#pragma unroll
      for (int j = 0; j < NJ; j++) sincos(q[j], &sn[j], &cs[j]);
    }
    time += MJS_RR_PHYSICS_DT;
    switch_update(touch, flags);  // Switch.after_substep (switch.py:71-72)
  }
  if (w.lazy) rr::smooth_acceleration(w.pM, w.pqs, w.warm);  // the next control step's warm start
  SoloOut o;
#pragma unroll
  for (int j = 0; j < NJ; j++) { o.q[j] = q[j]; o.v[j] = v[j]; o.cs[j] = cs[j]; o.sn[j] = sn[j]; o.warm[j] = w.warm[j]; }
  o.time = time;
  o.grip_th = grip.th; o.grip_vel = grip.vel;
  o.flags = flags;
  o.bad = bad;
  o.rows_active = rows_active;
  o.slot_overflow = slot_overflow;
  return o;
}

// reset of one lane (mjs_reset, or the next-step auto-reset of an env whose episode ended), out of line: state rows, flag byte
// (ONE store: a concurrent reader sees the old or the new byte), the FIRST time step's outputs
__device__ __noinline__ void reset_lane(KernelParams p, int i, uint8_t flags, rr::Ws ws, uint8_t extra_flags) {
  double obs[OBS_DIM];
  rr::Chain c;
  ResetOut r = episode_init(p.rng, i, flags, ws);
  rr::store_state(p, i, r.st);
  p.state[(size_t)S_GRIP * p.N + i] = 0.0;  // mj_resetData: gripper open, at rest
  p.state[(size_t)(S_GRIP + 1) * p.N + i] = 0.0;
  rr::store_warm(p, i, S_WARM, nullptr);
  p.flags[i] = (uint8_t)(r.flags | extra_flags);
  {
    double cs[NJ], sn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) sincos(r.st.q[j], &sn[j], &cs[j]);
    rr::store_cs(p, i, S_CS, cs, sn);
    rr::fk_cs(cs, sn, c);
  }
  make_obs(r.st, c, r.flags, obs);
  write_outputs<OBS_DIM>(p, i, obs, 0.0, 1.0, MJS_STEP_FIRST, false, false, false, 0, r.ncon);
}

// ROLES == 2 (default for stepping): the role-specialised pair of wavefronts of rr::kernel (role 0: M(q), U D U^T,
// U^-1; role 1: servo set-point, actuators, bias forces; two LDS exchanges per substep) with this scene's
// generated dynamics, for workgroups in which no env can get constraint rows during the control step; the others
// take solo_control_step on role 0.
template <bool IS_RESET, int ROLES>
__global__ __launch_bounds__(64 * ROLES) void kernel(KernelParams p) {
  const int lane = threadIdx.x & 63;
  const int role = (ROLES == 2) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  // MJS_VARIANT_RESET_GROUPS (p.reset_groups; rr::kernel3 explains the protocol): the grid's second half are reset workgroups,
  // workgroup G + g resets the envs of group g whose episode ended while workgroup g steps the others on another CU
  // Envs per workgroup: p.epg lanes of each wavefront carry an env (64; smaller groups are an A/B knob that lost:
  // profiles/r04_d_button_group_size.txt). The path decision below is per workgroup; results are a function of the env's group.
  const int epg = p.epg;
  const int groups = (p.N + epg - 1) / epg;
  const bool resetter = !IS_RESET && ROLES == 2 && (int)blockIdx.x >= groups;
  const int i = (resetter ? (int)blockIdx.x - groups : (int)blockIdx.x) * epg + lane;
  __shared__ double xch[ROLES == 2 ? 12 : 1][64];  // rows 0-5: qfrc_smooth (role 1 -> 0), 6-11: qacc (role 0 -> 1)
  if (lane >= epg || i >= p.N) return;
  uint8_t flags = p.flags[i];
  const bool pending = (flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP;
  const rr::Ws ws{p.ws, p.N, i};
  if (resetter) {
    if (role == 0 && pending && pending_is_due(p, flags)) reset_lane(p, i, flags, ws, (uint8_t)(FLAG_FRESH | (p.epoch ? FLAG_EPOCH : 0)));
    return;
  }
  if (ROLES == 2) __syncthreads();  // both wavefronts have read flags[i] before role 0 may rewrite it (see rr::kernel3)
  double obs[OBS_DIM];
  rr::Chain c;
  if (!IS_RESET && ROLES == 2 && p.reset_groups) {
    if (pending || ((flags & FLAG_FRESH) && ((flags & FLAG_EPOCH) != 0) == (p.epoch != 0))) return;
    flags = (uint8_t)(flags & ~(FLAG_FRESH | FLAG_EPOCH));
  } else if (IS_RESET || pending) {
    if (role != 0) return;
    if (IS_RESET && p.reset_mask && !p.reset_mask[i]) return;
    reset_lane(p, i, flags, ws, 0);
    return;
  }
  MJS_STAMP(p, 0);
  rr::State st = rr::load_state(p, i);
  Grip grip{p.state[(size_t)S_GRIP * p.N + i], p.state[(size_t)(S_GRIP + 1) * p.N + i]};
  const V3 sw = v3(st.target[0], st.target[1], st.target[2]);
  // before_step (robot_push_button.py:143-157); evaluated by both roles (same result)
  const int adim = p.action_type == MJS_ACTION_ABS_EEF ? ACT_DIM_EEF : ACT_DIM_JOINT;
  double q0[NJ], q1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) q0[j] = st.q[j];
  if (p.action_type == MJS_ACTION_ABS_EEF) {
    double act[3];
#pragma unroll
    for (int k = 0; k < 3; k++) act[k] = p.actions[(size_t)i * adim + k];
    if (!rr::tcp_pose_to_joints(act, q0, q1)) {
      flags |= FLAG_IK_FAILED;
#pragma unroll
      for (int j = 0; j < NJ; j++) q1[j] = q0[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; j++) q1[j] = p.actions[(size_t)i * adim + j];  // servoJ(a[:6])
  }
  // gripper.move(a[-1]) (robot_push_button.py:147,155): the commanded finger opening -> fingers_actuator ctrl
  const double grip_ctrl = grip_ctrl_of_opening(p.actions[(size_t)i * adim + (adim - 1)]);
  const double t0 = st.time, t1 = st.time + MJS_RR_CONTROL_DT, inv_span = 1.0 / (t1 - t0);
  bool bad = false, rows_active = false, slot_overflow = false;
  double cs[NJ], sn[NJ], warm_out[NJ] = {0, 0, 0, 0, 0, 0};
  rr::load_cs(p, i, S_CS, cs, sn);
  rr::fk_cs(cs, sn, c);
  // Which path? Both roles evaluate the same predicates on the same data, so the decisions agree without an exchange.
  // The arm's own geoms and the floor (the registered action space is +-3.14 rad on every joint, robot_push_button.py:193-203):
  // the row-free path needs the configuration now AND the servo target clear of the floor by rr::CLEAR_MARGIN, and a
  // joint-space travel (velocity included) short enough that the arm's points cannot dip from the chord between the two to the
  // floor: reach (<= 1 m) * travel^2 / 8 <= clearance - 2 cm. Otherwise the whole control step takes the robust path.
  bool link_unsafe;
  {
    double c1[NJ], s1[NJ], t2 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      sincos(q1[j], &s1[j], &c1[j]);
      const double d = fabs(q1[j] - q0[j]) + 0.05 * fabs(st.v[j]);
      t2 = fma(d, d, t2);
    }
    rr::Chain ch1;
    rr::fk_cs(c1, s1, ch1);
    const double clr = fmin(rr::min_floor_clearance(c), rr::min_floor_clearance(ch1));
    link_unsafe = !(clr >= rr::CLEAR_MARGIN) || !(t2 <= 8.0 * (clr - 0.02));
  }
  const bool links_solo = __any(link_unsafe);
  const bool guarded = links_solo || __any(rows_possible(st.q, st.v, c, sw, grip, grip_ctrl, MJS_RR_CONTROL_DT));  // some env may get rows during this control step
  MJS_STAMP(p, 1);
  int solo_from = (ROLES == 1 || links_solo) ? 0 : MJS_RR_NSUB;                      // first substep of the robust path
  if constexpr (ROLES == 2) {
#pragma unroll 1
    for (int seg = 0; seg < MJS_RR_NSUB / SEGMENT_SUBSTEPS && !links_solo; seg++) {
      if (guarded) {  // wave-uniform: the steady state never enters
        if (seg > 0) rr::fk_cs(cs, sn, c);
        if (__any(rows_possible(st.q, st.v, c, sw, grip, grip_ctrl, SEGMENT_SUBSTEPS * MJS_RR_PHYSICS_DT))) {
          solo_from = seg * SEGMENT_SUBSTEPS;
          break;
        }
      }
#pragma unroll 1
      for (int s = 0; s < SEGMENT_SUBSTEPS; s++) {
        double qacc[NJ];
        const double t = fmin(fmax(st.time, t0), t1);
        double ctrl[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
        if (role == 1) {
          double bias[NJ], fact[NJ];
          rr::actuator_forces(st.q, st.v, ctrl, fact);
          ur5e_bp_bias_gen(cs, sn, st.v, bias);
#pragma unroll
          for (int j = 0; j < NJ; j++) xch[j][lane] = fact[j] - bias[j];  // qfrc_smooth = -bias + actuator
          __syncthreads();  // qfrc_smooth published
          __syncthreads();  // qacc published
#pragma unroll
          for (int j = 0; j < NJ; j++) qacc[j] = xch[6 + j][lane];
        } else {
          double fdummy[NJ], M[21], A[NJ][NJ], W[NJ][NJ], Dinv[NJ], rhs[NJ];
          const int clamped = rr::actuator_forces(st.q, st.v, ctrl, fdummy);
          ur5e_bp_M_gen(cs, sn, M);
#pragma unroll
          for (int r = 0; r < NJ; r++) {
#pragma unroll
            for (int j = 0; j <= r; j++) A[r][j] = M[r * (r + 1) / 2 + j];
          }
          rr::factor_system(A, clamped, Dinv);
          rr::invert_unit_upper(A, W);
#pragma unroll
          for (int r = 0; r < NJ; r++) {  // opaque register uses pin the whole factorisation before the barrier
            asm volatile("" : "+v"(Dinv[r]));
#pragma unroll
            for (int j = 0; j < r; j++) asm volatile("" : "+v"(W[r][j]));
          }
          __syncthreads();  // qfrc_smooth published
#pragma unroll
          for (int j = 0; j < NJ; j++) rhs[j] = xch[j][lane];
          rr::apply_inverse(W, Dinv, rhs, qacc);
#pragma unroll
          for (int j = 0; j < NJ; j++) xch[6 + j][lane] = qacc[j];
          __syncthreads();  // qacc published
        }
        double acc2 = 0, dq2 = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          acc2 = fma(qacc[j], qacc[j], acc2);
          st.v[j] += MJS_RR_PHYSICS_DT * qacc[j];
          double dq = MJS_RR_PHYSICS_DT * st.v[j];
          st.q[j] += dq;
          dq2 = fma(dq, dq, dq2);
          rr::rotate_small(cs[j], sn[j], dq);
        }
        bad = bad || !(acc2 <= 1e20);
        if (!(dq2 <= 0.01)) {
#pragma unroll
          for (int j = 0; j < NJ; j++) sincos(st.q[j], &sn[j], &cs[j]);
        }
        st.time += MJS_RR_PHYSICS_DT;
        grip_integrate(grip, grip_ctrl);  // both roles, same arithmetic: no exchange
      }
    }
    // no contact was possible on these substeps: the touch sensor read 0 after each of them (Switch.after_substep)
    if (solo_from > 0) flags = (uint8_t)(flags & ~FLAG_SWITCH_PRESSED);
  }
  if (solo_from < MJS_RR_NSUB) {
    if (role != 0) return;
    SoloIn in;
#pragma unroll
    for (int j = 0; j < NJ; j++) { in.q[j] = st.q[j]; in.v[j] = st.v[j]; in.q0[j] = q0[j]; in.q1[j] = q1[j]; in.cs[j] = cs[j]; in.sn[j] = sn[j]; }
    in.time = st.time; in.t0 = t0; in.t1 = t1;
    in.sw[0] = sw.x; in.sw[1] = sw.y; in.sw[2] = sw.z;
    in.flags = flags;
    in.first_substep = solo_from;  // the whole wavefront runs the robust path from here: every substep detects rows for every lane of it
    in.grip_th = grip.th; in.grip_vel = grip.vel; in.grip_ctrl = grip_ctrl;
    // the state's qacc_warmstart belongs to the start of the control step; after row-free substeps the robust path starts
    // without one (its first substep has no rows by the guard's margin; later ones continue from it)
    in.has_warm = solo_from == 0 && (flags & FLAG_WARM_VALID);
#pragma unroll
    for (int j = 0; j < NJ; j++) in.warm[j] = in.has_warm ? p.state[(size_t)(S_WARM + j) * p.N + i] : 0.0;
    SoloOut o = solo_control_step(in, ws);
#pragma unroll
    for (int j = 0; j < NJ; j++) { st.q[j] = o.q[j]; st.v[j] = o.v[j]; cs[j] = o.cs[j]; sn[j] = o.sn[j]; warm_out[j] = o.warm[j]; }
    st.time = o.time;
    grip.th = o.grip_th; grip.vel = o.grip_vel;
    flags = o.flags;
    bad = bad || o.bad;
    rows_active = o.rows_active;
    slot_overflow = o.slot_overflow;
  }
  const bool solo = solo_from < MJS_RR_NSUB;
  if (role != 0) return;
  MJS_STAMP(p, 2);
  // after_step (robot_push_button.py:159-165): rand() is drawn only for an active, released switch
  if (p.button_disturbances && (flags & FLAG_SWITCH_ACTIVE) && !(flags & FLAG_SWITCH_PRESSED)) flags = disturb(p.rng, i, flags);
#pragma unroll
  for (int j = 0; j < NJ; j++) bad = bad || bad_value(st.q[j]) || bad_value(st.v[j]);
  rr::fk_cs(cs, sn, c);
  make_obs(st, c, flags, obs);
  MJS_STAMP(p, 3);
  // goal: switch active and TCP within 0.05 of the end position (robot_push_button.py:205-219)
  double dx = obs[6] - MJS_BP_ROBOT_END_POS[0], dy = obs[7] - MJS_BP_ROBOT_END_POS[1], dz = obs[8] - MJS_BP_ROBOT_END_POS[2];
  bool success = (flags & FLAG_SWITCH_ACTIVE) && sqrt(dx * dx + dy * dy + dz * dz) < MJS_BP_GOAL_THRESHOLD;
  double reward = success ? 1.0 : 0.0, discount = success ? 0.0 : 1.0;
  bool terminate = success;
  if (bad) { reward = 0; discount = 0; terminate = true; }
  if (st.time >= p.time_limit) terminate = true;
  // ncon after the step (mj_step1 of the last substep): arm-vs-floor + finger-tip sphere contacts
  int ncon = rr::floor_contacts_from_clearance(c, rr::min_floor_clearance(c));
  const bool arm_touches_floor = ncon > 0;
  int ncon_proxy;
  {
    ContactSet con;
    detect_contacts(c, grip.th, sw, con);
    ncon_proxy = con.n + (detect_wrist_box(c, sw).hit ? 1 : 0);
    ncon += ncon_proxy;
  }
  MJS_STAMP(p, 4);
  const bool violated = !solo && (ncon_proxy > 0 || arm_touches_floor || rr::joint_outside_range(st.q));  // a-posteriori check of the row-free path (solo: every lane detects)
  int fault = (bad ? MJS_FAULT_BAD_STATE : 0) | ((flags & FLAG_IK_FAILED) ? MJS_FAULT_IK_FAILED : 0) | (rows_active ? MJS_FAULT_LIMIT_COLDSTART : 0) |
              (slot_overflow ? MJS_FAULT_UNSUPPORTED_CONTACT : 0) | (violated ? MJS_FAULT_FASTPATH_VIOLATED : 0);
  bool terminated = terminate && discount == 0.0, truncated = terminate && discount > 0.0;
  uint8_t newflags = (uint8_t)((flags & (FLAG_IK_FAILED | FLAG_SWITCH_ACTIVE | FLAG_SWITCH_PRESSED)) | (terminate ? pending_mark(p) : 0) | (solo ? FLAG_WARM_VALID : 0));
  rr::store_state_stepped(p, i, st);  // q, v, time: a step only reads the switch rows
  rr::store_cs(p, i, S_CS, cs, sn);
  if (solo) rr::store_warm(p, i, S_WARM, warm_out);
  p.state[(size_t)S_GRIP * p.N + i] = grip.th;
  p.state[(size_t)(S_GRIP + 1) * p.N + i] = grip.vel;
  p.flags[i] = newflags;
  write_outputs<OBS_DIM>(p, i, obs, reward, discount, terminate ? MJS_STEP_LAST : MJS_STEP_MID, terminated, truncated, success, fault, ncon);
  MJS_STAMP(p, 5);
  if (terminate && p.autoreset == MJS_AUTORESET_SAME_STEP) {
    if (p.out.terminal_obs) {
#pragma unroll
      for (int k = 0; k < OBS_DIM; k++) p.out.terminal_obs[(size_t)i * OBS_DIM + k] = obs[k];
    }
    ResetOut r = episode_init(p.rng, i, newflags, ws);
    rr::store_state(p, i, r.st);
    p.state[(size_t)S_GRIP * p.N + i] = 0.0;
    p.state[(size_t)(S_GRIP + 1) * p.N + i] = 0.0;
    rr::store_warm(p, i, S_WARM, nullptr);
    p.flags[i] = r.flags;
    {
      double cs2[NJ], sn2[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) sincos(r.st.q[j], &sn2[j], &cs2[j]);
      rr::store_cs(p, i, S_CS, cs2, sn2);
      rr::fk_cs(cs2, sn2, c);
    }
    make_obs(r.st, c, r.flags, obs);
    if (p.out.obs) {
#pragma unroll
      for (int k = 0; k < OBS_DIM; k++) p.out.obs[(size_t)i * OBS_DIM + k] = obs[k];
    }
    if (p.out.ncon) p.out.ncon[i] = r.ncon;
  }
}

}  // namespace bp
