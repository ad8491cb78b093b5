// mjs_arm_stage.h — the robot scenes' GENERAL constraint stage: joint-limit rows + any number (<= ST_MAXC) of
// pyramidal condim-3 contacts of the arm's own collision geoms with the floor (and, per scene, further contacts of
// bodies on the arm: Button-Push's finger tips). Included by mjs_reach.h (namespace rr, after the 6x6 helpers).
//
// Path replaced: one mj_fwdConstraint of Physics.step() (mj_collision's plane-capsule / plane-cylinder pairs of the
// UR5e's collision geoms, include/mjs_scene_spec.h MJS_UR_COL_*; mj_instantiateLimit / mj_instantiateContact;
// mj_makeImpedance; mj_referenceConstraint; mj_solPrimal's Newton) for the lanes that have an active row. Reached from
// the registered action spaces: robot_push_button.py:193-203 (+-3.14 rad joint targets) swings links onto the floor.
// Same rules, row order and stopping criteria as oracle/om_engine.c (collide_plane, om_make_constraint,
// om_make_impedance, om_solve_constraint).
//
// Layout: the static-slot stages (mjs_button.h) keep every row in registers, which caps them at 4 contacts. An arm
// lying on the floor has up to 22 (9 capsules x 2 end spheres + 4 rim points of the wrist cylinder), so this stage keeps
// per-contact data in an HBM WORKSPACE owned by the handle ([ST_MAXC * ST_SLOT][N] doubles, struct-of-arrays like the
// state: a wavefront's access to one row is one coalesced 512-B transaction) and walks the contacts in wave-uniform
// loops; a contact's 3 x 6 frame Jacobian is never stored, it is rebuilt from (point, normal, last moving joint) and the
// joint axes / anchors the lane holds in registers (36 doubles) whenever a pass needs it:
//   setup pass      D, K * imp * dist, B * (Jc v)                                  per contact, once
//   eval pass       W = Jc a + B Jc v; edge residuals, active set, cost, J^T f, Hessian  once per Newton iteration
//   direction pass  Jc search                                                      once per Newton iteration
//   line search     reads W, Jc search, D per contact (no Jacobian)                 per 1-D Newton step
// Rare path by construction (the fast paths' guards keep steady-state workloads out of it); what matters here is that it
// is exact and has no row limit worth the name, not its instruction count.
#pragma once

namespace rr {

constexpr int ST_MAXC = 24;   // active (penetrating) contacts a lane can carry (the arm alone has 22 candidates); more -> MJS_FAULT_UNSUPPORTED_CONTACT
constexpr int ST_SLOT = 18;   // doubles per contact slot: pos3 nrm3 | invw->D, dist->kid, meta | B Jc v (3) | W (3) | Jc search (3)
constexpr int WS_ROWS = ST_MAXC * ST_SLOT;
struct Ws {
  double* base;  // [WS_ROWS][N]
  int N, i;
};
MJS_DEV double ws_ld(const Ws& w, int c, int k) { return w.base[(size_t)(c * ST_SLOT + k) * w.N + w.i]; }
MJS_DEV void ws_st(const Ws& w, int c, int k, double x) { w.base[(size_t)(c * ST_SLOT + k) * w.N + w.i] = x; }

// mju_makeFrame: two tangents for a unit normal
MJS_DEV void make_frame(V3 n, V3& t1, V3& t2) {
  V3 y = (n.y > -0.5 && n.y < 0.5) ? v3(0, 1, 0) : v3(0, 0, 1);
  double dp = dot(n, y);
  y = madd(y, -dp, n);
  double len = sqrt(dot(y, y));
  t1 = (1.0 / len) * y;
  t2 = cross(n, t1);
}

// world position of collision geom g's centre and its axis (geom z): local quat is identity or (1,1,0,0) = Rx(90 deg)
MJS_DEV void col_geom_pose(const Chain& c, int g, V3& gp, V3& axis) {
  const int b = MJS_UR_COL_BODY[g];
  const M3 R = c.R[b];
  gp = madd(madd(madd(c.p[b], MJS_UR_COL_POS[g][0], R.cx), MJS_UR_COL_POS[g][1], R.cy), MJS_UR_COL_POS[g][2], R.cz);
  axis = (MJS_UR_COL_QUAT[g][1] != 0.0) ? -R.cy : R.cz;
}

// Lower bound of the height of the arm's collision geometry above the floor: min over the capsules' end spheres of
// centre.z - radius; the wrist cylinder is bounded by the capsule of its own radius and half-length. > 0 means that
// mj_collision finds no arm-floor contact (count_floor_contacts == 0), exactly for the capsules, conservatively for the
// cylinder.
MJS_DEV double min_floor_clearance(const Chain& c) {
  double m = INFINITY;
#pragma unroll
  for (int g = 2; g < MJS_UR_NCOLGEOM; g++) {  // geoms 0 and 1 never touch the floor, whatever the joints do (see the assertions below)
    const int b = MJS_UR_COL_BODY[g];
    const M3 R = c.R[b];
    const double gz = c.p[b].z + MJS_UR_COL_POS[g][0] * R.cx.z + MJS_UR_COL_POS[g][1] * R.cy.z + MJS_UR_COL_POS[g][2] * R.cz.z;
    const double az = (MJS_UR_COL_QUAT[g][1] != 0.0) ? R.cy.z : R.cz.z;
    // geom 2 (the upper arm's long capsule along local z) has its near end sphere ON the shoulder-lift axis (offset - half-length
    // = 0): that end stays 0.163 - 0.05 m above the floor whatever the joints do; only its far (elbow) end can come down
    const double lowest_end = g == 2 ? MJS_UR_COL_SIZE[g][1] * az : -MJS_UR_COL_SIZE[g][1] * fabs(az);
    m = fmin(m, gz + lowest_end - MJS_UR_COL_SIZE[g][0]);
  }
  return m;
}
static_assert(MJS_UR_COL_BODY[2] == 2 && MJS_UR_COL_TYPE[2] == 3 && MJS_UR_COL_QUAT[2][1] == 0.0 && MJS_UR_COL_POS[2][0] == 0.0 && MJS_UR_COL_POS[2][1] == 0.0 &&
                  MJS_UR_COL_POS[2][2] == MJS_UR_COL_SIZE[2][1] && MJS_UR_BODY_POS[1][2] - MJS_UR_COL_SIZE[2][0] > 0,
              "the upper arm's long capsule is assumed to start on the shoulder-lift axis");
// geom 0: capsule on the shoulder link along its vertical joint axis: lowest point 3 mm above the floor, fixed.
// geom 1: capsule on the upper arm along the shoulder-lift axis (local y, horizontal in every configuration: the only joint
// before it turns about the vertical), centred on that axis at the shoulder's height: lowest point 0.163 - 0.06 m, fixed.
static_assert(MJS_UR_COL_BODY[0] == 1 && MJS_UR_COL_TYPE[0] == 3 && MJS_UR_COL_QUAT[0][1] == 0.0 && MJS_UR_COL_POS[0][0] == 0.0 && MJS_UR_COL_POS[0][1] == 0.0 &&
                  MJS_UR_BODY_POS[1][2] + MJS_UR_COL_POS[0][2] - MJS_UR_COL_SIZE[0][1] - MJS_UR_COL_SIZE[0][0] > 0,
              "the shoulder capsule is assumed to clear the floor in every configuration");
static_assert(MJS_UR_COL_BODY[1] == 2 && MJS_UR_COL_TYPE[1] == 3 && MJS_UR_COL_QUAT[1][1] != 0.0 && MJS_UR_COL_POS[1][0] == 0.0 && MJS_UR_COL_POS[1][2] == 0.0 &&
                  MJS_UR_BODY_POS[2][0] == 0.0 && MJS_UR_BODY_POS[2][2] == 0.0 && MJS_UR_BODY_POS[1][2] - MJS_UR_COL_SIZE[1][0] > 0,
              "the upper arm's shoulder-side capsule is assumed to lie on the (horizontal) shoulder-lift axis at the shoulder's height");

// mjc_PlaneCylinder against the floor z = 0 (normal +z): the deepest rim point, the same rim point of the other disk, and two
// side points of the lower disk, in MuJoCo's order; emit(pos, dist) for every DETECTED contact (dist <= 0). gp = centre, axis =
// the cylinder's axis (geom z), xaxis = geom x (used when the disks are parallel to the floor).
template <class Emit>
MJS_DEV void plane_cylinder_contacts(V3 gp, V3 axis, V3 xaxis, double rad, double half, Emit emit) {
  const double dist0 = gp.z;
  double prjaxis = axis.z;
  if (prjaxis > 0) { axis = -axis; prjaxis = -prjaxis; }
  V3 vec = prjaxis * axis - v3(0, 0, 1);
  const double len = sqrt(dot(vec, vec));
  if (len < MJS_MINVAL) vec = rad * xaxis;
  else vec = (rad / len) * vec;
  const double prjvec = vec.z;
  axis = half * axis;
  prjaxis *= half;
  double dd = dist0 + prjaxis + prjvec;
  if (dd > 0) return;
  emit(v3(gp.x + vec.x + axis.x, gp.y + vec.y + axis.y, gp.z + vec.z + axis.z - dd * 0.5), dd);
  dd = dist0 - prjaxis + prjvec;
  if (dd <= 0) emit(v3(gp.x + vec.x - axis.x, gp.y + vec.y - axis.y, gp.z + vec.z - axis.z - dd * 0.5), dd);
  V3 side = cross(vec, axis);
  const double sl = sqrt(dot(side, side));
  if (sl > MJS_MINVAL) {
    side = (rad * sqrt(3.0) * 0.5 / sl) * side;
    dd = dist0 + prjaxis - 0.5 * prjvec;
    if (dd <= 0) {
#pragma unroll
      for (int s = 1; s >= -1; s -= 2)  // point A = +side first, then B = -side (mjc_PlaneCylinder's order)
        emit(v3(gp.x + s * side.x + axis.x - 0.5 * vec.x, gp.y + s * side.y + axis.y - 0.5 * vec.y, gp.z + s * side.z + axis.z - 0.5 * vec.z - dd * 0.5), dd);
    }
  }
}

// every DETECTED arm-floor contact (dist <= 0, what mj_collision lists) in MuJoCo's pair order: emit(body, pos, dist).
// Same arithmetic as count_floor_contacts (mjc_PlaneCapsule: one contact per end sphere; mjc_PlaneCylinder: up to 4).
template <class Emit>
MJS_DEV void arm_floor_contacts(const Chain& c, Emit emit) {
#pragma unroll
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    const int b = MJS_UR_COL_BODY[g];
    V3 gp, axis;
    col_geom_pose(c, g, gp, axis);
    const double rad = MJS_UR_COL_SIZE[g][0], half = MJS_UR_COL_SIZE[g][1];
    if (MJS_UR_COL_TYPE[g] == 3) {
#pragma unroll
      for (int e = -1; e <= 1; e += 2) {
        const V3 ctr = madd(gp, e * half, axis);
        if (!(ctr.z > rad)) {
          const double dist = ctr.z - rad;
          emit(b, v3(ctr.x, ctr.y, ctr.z - (rad + 0.5 * dist)), dist);
        }
      }
    } else {
      plane_cylinder_contacts(gp, axis, c.R[b].cx, rad, half, [&](V3 pos, double dist) { emit(b, pos, dist); });
    }
  }
}

struct GenStageIn {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], M[21] /* lower triangle of M + armature */, qs[NJ] /* qfrc_smooth */;
  double warm[NJ];  // qacc_warmstart: the previous Physics.step()'s solver acceleration
  bool has_warm;
};
struct GenStageOut {
  double qs[NJ] /* qfrc_smooth + qfrc_constraint */, qacc[NJ] /* the solver's acceleration: the next step's warm start */, touch;
  bool overflow;
};

// SC (scene): dof_invweight(j), link_invweight(b), solver_scale(ex) = 1 / (meaninertia * nv) of the WHOLE model, struct Extra (by
// value), extra_contacts(ch, ex, emit) with emit(ndof, pos, nrm, sgn, dist, invweight, on_switch), in_touch_site(ex, pos).
template <class SC>
__device__ __noinline__ GenStageOut gen_stage(GenStageIn in, typename SC::Extra ex, Ws ws) {
  const double mu = MJS_GEOM_FRICTION_SLIDE;
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  double v[NJ], qs[NJ], Mf[NJ][NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    v[i] = in.v[i]; qs[i] = in.qs[i];
#pragma unroll
    for (int j = 0; j < NJ; j++) Mf[i][j] = i >= j ? in.M[i * (i + 1) / 2 + j] : in.M[j * (j + 1) / 2 + i];
  }
  // ---- joint-limit rows: 12 static slots (2j lower, 2j+1 upper; J = +-e_j), mj_instantiateLimit with jnt_margin = 0
  bool lon[NLIM], any_lim = false;
  double lD[NLIM], laref[NLIM];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int k = 2 * j + side;
      const double sgn = side == 0 ? 1.0 : -1.0;
      const double dist = side == 0 ? in.q[j] - MJS_UR_JNT_RANGE[j][0] : MJS_UR_JNT_RANGE[j][1] - in.q[j];
      const double imp = impedance_default(dist);
      lon[k] = dist < 0.0;
      lD[k] = 1 / fmax(MJS_MINVAL, (1 - imp) * SC::dof_invweight(j) / imp);
      laref[k] = -B * (sgn * v[j]) - K * imp * dist;
      any_lim = any_lim || lon[k];
    }
  }
  const bool use_lim = __any(any_lim);
  // ---- contacts: detect, keep the ACTIVE ones (dist < 0: a contact at dist == margin is listed but makes no rows)
  V3 ax[NJ], an[NJ];
  int n = 0;
  bool overflow = false;
  {
    Chain ch;
    fk_cs(in.cs, in.sn, ch);
#pragma unroll
    for (int j = 0; j < NJ; j++) { ax[j] = joint_axis(ch, j); an[j] = ch.p[j + 1]; }
    auto emit = [&](int ndof, V3 pos, V3 nrm, double sgn, double dist, double invw, bool on_switch) {
      if (!(dist < 0.0)) return;
      if (n >= ST_MAXC) { overflow = true; return; }
      ws_st(ws, n, 0, pos.x); ws_st(ws, n, 1, pos.y); ws_st(ws, n, 2, pos.z);
      ws_st(ws, n, 3, nrm.x); ws_st(ws, n, 4, nrm.y); ws_st(ws, n, 5, nrm.z);
      ws_st(ws, n, 6, invw); ws_st(ws, n, 7, dist);
      ws_st(ws, n, 8, (double)(ndof | (sgn < 0 ? 8 : 0) | (on_switch ? 16 : 0)));
      n++;
    };
    arm_floor_contacts(ch, [&](int b, V3 pos, double dist) { emit(b, pos, v3(0, 0, 1), 1.0, dist, fmax(MJS_MINVAL, SC::link_invweight(b)), false); });
    SC::extra_contacts(ch, ex, emit);
  }
  int nmax = 0;
  while (__any(n > nmax)) nmax++;
  // the contact frame's Jacobian (rows: normal, tangent 1, tangent 2; already geom2 - geom1) of slot c
  auto frame_jac = [&](int c, double Jc[3][NJ], V3& pos, int& meta) {
    pos = v3(ws_ld(ws, c, 0), ws_ld(ws, c, 1), ws_ld(ws, c, 2));
    const V3 nrm = v3(ws_ld(ws, c, 3), ws_ld(ws, c, 4), ws_ld(ws, c, 5));
    meta = (int)ws_ld(ws, c, 8);
    const int ndof = meta & 7;
    const double sgn = (meta & 8) ? -1.0 : 1.0;
    V3 t1, t2;
    make_frame(nrm, t1, t2);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const V3 col = sgn * cross(ax[j], pos - an[j]);  // (jac2 - jac1) column of a body moved by joints 0 .. ndof-1
      const bool moves = j < ndof;
      Jc[0][j] = moves ? dot(nrm, col) : 0.0; Jc[1][j] = moves ? dot(t1, col) : 0.0; Jc[2][j] = moves ? dot(t2, col) : 0.0;
    }
  };
  auto frame_mul = [&](const double Jc[3][NJ], const double* x, double* u) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) s += Jc[r][j] * x[j];
      u[r] = s;
    }
  };
  // ---- setup pass: mj_makeImpedance / mj_referenceConstraint per contact
#pragma unroll 1
  for (int c = 0; c < nmax; c++) {
    if (c < n) {
      double Jc[3][NJ], uv[3];
      V3 pos;
      int meta;
      frame_jac(c, Jc, pos, meta);
      frame_mul(Jc, v, uv);
      const double invw = ws_ld(ws, c, 6), dist = ws_ld(ws, c, 7);
      const double imp = impedance_default(dist);
      const double dA = invw + mu * mu * invw;
      ws_st(ws, c, 6, 1 / (2 * mu * mu * fmax(MJS_MINVAL, (1 - imp) * dA / imp)));  // D of the pyramid's edges
      ws_st(ws, c, 7, K * imp * dist);
#pragma unroll
      for (int r = 0; r < 3; r++) ws_st(ws, c, 9 + r, B * uv[r]);
    }
  }
  // ---- Newton (mj_solPrimal): cost, gradient, exact Hessian, Cholesky, exact line search, MuJoCo's stopping rules
  double a[NJ], a_s[NJ], Ma[NJ], fc[NJ], H[NJ][NJ], touch = 0;
  bool lact[NLIM];
  double ljar[NLIM];
  {
    double L[NJ][NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) {
#pragma unroll
      for (int j = 0; j < NJ; j++) L[i][j] = Mf[i][j];
      a_s[i] = qs[i];
    }
    chol6(L);
    chol6_solve(L, a_s);  // qacc_smooth
  }
  // cost at `a` with the active set, J^T force (fc) and, when asked, the Hessian M + J^T D_active J (lower triangle);
  // leaves W = Jc a + B Jc v of every contact in the workspace for the line search
  auto eval = [&](bool need_H) -> double {
    double cost = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m += Mf[i][k] * a[k];
      Ma[i] = m;
      fc[i] = 0;
#pragma unroll
      for (int j = 0; j <= i; j++) H[i][j] = Mf[i][j];
    }
    if (use_lim) {
#pragma unroll
      for (int k = 0; k < NLIM; k++) {
        ljar[k] = -laref[k] + ((k & 1) ? -a[k >> 1] : a[k >> 1]);
        const bool act = lon[k] && ljar[k] < 0;
        lact[k] = act;
        const double f = act ? -lD[k] * ljar[k] : 0.0;
        if (act) cost += 0.5 * lD[k] * ljar[k] * ljar[k];
        fc[k >> 1] += (k & 1) ? -f : f;
        if (act) H[k >> 1][k >> 1] += lD[k];
      }
    }
    touch = 0;
#pragma unroll 1
    for (int c = 0; c < nmax; c++) {
      if (c < n) {
        double Jc[3][NJ], W[3];
        V3 pos;
        int meta;
        frame_jac(c, Jc, pos, meta);
        frame_mul(Jc, a, W);
        const double D = ws_ld(ws, c, 6), kid = ws_ld(ws, c, 7);
#pragma unroll
        for (int r = 0; r < 3; r++) { W[r] += ws_ld(ws, c, 9 + r); ws_st(ws, c, 12 + r, W[r]); }
        // edge e: J = Jn +- mu Jt(1 + e/2); residual jar = J a - aref, aref = -B J v - K imp dist
        double f[4], nact[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double x = W[0] + ((e & 1) ? -mu : mu) * W[1 + (e >> 1)] + kid;
          const bool act = x < 0;
          nact[e] = act ? 1.0 : 0.0;
          f[e] = act ? -D * x : 0.0;
          if (act) cost += 0.5 * D * x * x;
        }
        const double fn = f[0] + f[1] + f[2] + f[3], f1 = mu * (f[0] - f[1]), f2 = mu * (f[2] - f[3]);
#pragma unroll
        for (int i = 0; i < NJ; i++) fc[i] += fn * Jc[0][i] + f1 * Jc[1][i] + f2 * Jc[2][i];
        if ((meta & 16) && SC::in_touch_site(ex, pos)) touch += fn;  // mj_sensorAcc, mjSENS_TOUCH
        if (need_H) {
          const double wn = D * (nact[0] + nact[1] + nact[2] + nact[3]), w1 = D * mu * (nact[0] - nact[1]), w2 = D * mu * (nact[2] - nact[3]);
          const double w11 = D * mu * mu * (nact[0] + nact[1]), w22 = D * mu * mu * (nact[2] + nact[3]);
          if (wn != 0.0) {
#pragma unroll
            for (int i = 0; i < NJ; i++) {
              const double jn = Jc[0][i], j1 = Jc[1][i], j2 = Jc[2][i];
              const double rn = wn * jn + w1 * j1 + w2 * j2, r1 = w1 * jn + w11 * j1, r2 = w2 * jn + w22 * j2;
#pragma unroll
              for (int j = 0; j <= i; j++) H[i][j] += rn * Jc[0][j] + r1 * Jc[1][j] + r2 * Jc[2][j];
            }
          }
        }
      }
    }
    double gauss = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) gauss += (Ma[i] - qs[i]) * (a[i] - a_s[i]);
    return cost + 0.5 * gauss;
  };
  // mj_fwdConstraint's warm start: the cheaper of qacc_warmstart and qacc_smooth (ties: the warm start, as the oracle's loop)
#pragma unroll
  for (int i = 0; i < NJ; i++) a[i] = a_s[i];
  if (__any(in.has_warm)) {
    const double cost_s = eval(false);
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = in.has_warm ? in.warm[i] : a_s[i];
    const double cost_w = eval(false);
    const bool keep = in.has_warm && !(cost_s < cost_w);
#pragma unroll
    for (int i = 0; i < NJ; i++) a[i] = keep ? a[i] : a_s[i];
  }
  const double scale = SC::solver_scale(ex);
  double oldcost = 0;
#pragma unroll 1
  for (int iter = 0; iter <= MJS_SOLVER_ITERATIONS; iter++) {
    const double cost = eval(true);
    double grad[NJ], gn = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) { grad[i] = Ma[i] - qs[i] - fc[i]; gn += grad[i] * grad[i]; }
    if (iter > 0 && (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE)) break;
    if (iter == MJS_SOLVER_ITERATIONS) break;
    oldcost = cost;
    double search[NJ], Mv[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) search[i] = -grad[i];
    if (!chol6(H)) break;
    chol6_solve(H, search);
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
    // direction pass: Jc search per contact
#pragma unroll 1
    for (int c = 0; c < nmax; c++) {
      if (c < n) {
        double Jc[3][NJ], us[3];
        V3 pos;
        int meta;
        frame_jac(c, Jc, pos, meta);
        frame_mul(Jc, search, us);
#pragma unroll
        for (int r = 0; r < 3; r++) ws_st(ws, c, 15 + r, us[r]);
      }
    }
    // 1-D Newton with bracketing on the piecewise-quadratic cost along `search` (PrimalSearch's gradient tolerance)
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale;
    double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 50; it++) {
      double d1 = g1 + alpha * g2, d2 = g2;
      if (use_lim) {
#pragma unroll
        for (int k = 0; k < NLIM; k++) {
          const double jv = (k & 1) ? -search[k >> 1] : search[k >> 1];
          const double x = ljar[k] + alpha * jv;
          if (lon[k] && x < 0) { d1 += lD[k] * x * jv; d2 += lD[k] * jv * jv; }
        }
      }
#pragma unroll 1
      for (int c = 0; c < nmax; c++) {
        if (c < n) {
          const double D = ws_ld(ws, c, 6), kid = ws_ld(ws, c, 7);
          const double W0 = ws_ld(ws, c, 12), W1 = ws_ld(ws, c, 13), W2 = ws_ld(ws, c, 14);
          const double u0 = ws_ld(ws, c, 15), u1 = ws_ld(ws, c, 16), u2 = ws_ld(ws, c, 17);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const double sm = (e & 1) ? -mu : mu;
            const double jar = W0 + sm * ((e >> 1) ? W2 : W1) + kid, jv = u0 + sm * ((e >> 1) ? u2 : u1);
            const double x = jar + alpha * jv;
            if (x < 0) { d1 += D * x * jv; d2 += D * jv * jv; }
          }
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
    for (int i = 0; i < NJ; i++) a[i] += alpha * search[i];
  }
  GenStageOut out;
#pragma unroll
  for (int j = 0; j < NJ; j++) { out.qs[j] = qs[j] + fc[j]; out.qacc[j] = a[j]; }
  out.touch = touch;
  out.overflow = overflow;
  return out;
}

}  // namespace rr
