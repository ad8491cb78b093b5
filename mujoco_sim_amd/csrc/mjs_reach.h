// mjs_reach.h — Robot-Reach (UR5e 6-DoF) fused control-step kernel (BASELINE config 3, the
// headline workload).
//
// One env per wavefront lane; persistent state (q6, v6, time, target3 = 16 doubles, SoA in HBM)
// is read and written once per control step; the analytic IK, the 20 physics substeps and the
// observation/reward/termination logic run in registers.
// Path replaced (reference, paths under /root/reference/mujoco_sim/):
//   environments/tasks/robot_reach.py:159-169 before_step -> entities/robots/robot.py:218-259
//       servoL/servoJ (TCP pose -> flange pose -> analytic IK closest to current q -> 2-waypoint
//       joint trajectory)
//   per substep: robot.py:261-263 + entities/robots/joint_trajectory.py:33-47 (ctrl = lerp) and
//       Physics.step() on the UR5e scene of robot_reach.py:90-119: forward kinematics,
//       composite-rigid-body inertia, Coriolis/centrifugal bias (RNE), gravity compensation on
//       the arm bodies (robot.py:80-82) with the un-compensated end-effector payload, affine
//       position servos with force clamp, implicitfast integration
//   robot.py:153-168,292-298 observables (tcp_position, joint_configuration),
//       robot_reach.py:122,171-181,206-207 target observable / reward / success
//   robot_reach.py:143-150 + environments/tasks/spaces.py:24-31 initialize_episode (6 uniforms + IK)
//   dm_control composer loop + environments/dmc2gym.py:144-153 termination / truncation split
// Dynamics are written in world axes about the world origin (same formulation as the oracle,
// hand-specialised to the UR5e chain; gravity compensation is folded analytically: only the
// payload's weight survives). Bound: per-lane FP64 dependency chains, not HBM.
#pragma once
#include "mjs_kernel_common.h"
#include "mjs_ur5e_dyn_gen.h"

#ifndef MJS_REACH_GENERIC_DYNAMICS
#define MJS_REACH_GENERIC_DYNAMICS 0  // 1 = first-version world-frame CRBA/RNE (kept for A/B profiles)
#endif

namespace rr {

constexpr int S_Q = 0, S_V = 6, S_TIME = 12, S_TARGET = 13;
// rows 16-21: qacc_warmstart (mjData state: the solver acceleration of the last Physics.step()). Only the robust path reads or
// writes them (flag FLAG_WARM_VALID says whether they belong to the current state); the row-free fast path never touches them.
constexpr int S_WARM = 16;
// rows 22-33: cos / sin of the six joint angles as the substep loop carries them (angle-addition updates, rotate_small): the step
// kernels start from them instead of six sincos calls on the critical path of the prologue (-2.4 k cycles of 9 k on the two
// dynamics wavefronts, profiles/r03_d_*). A cache of q, not state of its own: resets and mjs_set_state (when the rows do not
// belong to the given q) write exact values; the loaded pair is re-normalised, so rounding cannot accumulate in its length.
constexpr int S_CS = 22, S_SN = 28, STATE_DIM = 34;
constexpr int HOT_ROWS_READ = 16 + 12, HOT_ROWS_WRITTEN = 13 + 12;  // per env-step on the row-free path (target rows are only read)
constexpr int OBS_DIM = 12, ACT_DIM = 3, NJ = 6;
constexpr double PI = 3.14159265358979323846;

// ----------------------------------------------------------------------------- constants
// wrist_3 link merged with the lumped 2F-85 payload (deviation D-1). Both COMs lie on the local
// y axis and both tensors are diagonal in the wrist_3 frame (flange frame = Rx(-90deg)).
constexpr double W3_M = MJS_UR_BODY_MASS[6], W3_CY = MJS_UR_BODY_IPOS[6][1];
// inertial frame of wrist_3 is Rz(90deg): principal x -> body y
constexpr double W3_IXX = MJS_UR_BODY_DIAGINERTIA[6][1], W3_IYY = MJS_UR_BODY_DIAGINERTIA[6][0], W3_IZZ = MJS_UR_BODY_DIAGINERTIA[6][2];
constexpr double PL_M = MJS_G2F85_MASS, PL_CY = MJS_UR_FLANGE_POS[1] + MJS_G2F85_IPOS[2];
constexpr double PL_IXX = MJS_G2F85_DIAGINERTIA[0], PL_IYY = MJS_G2F85_DIAGINERTIA[2], PL_IZZ = MJS_G2F85_DIAGINERTIA[1];
constexpr double L6_M = W3_M + PL_M;
constexpr double L6_CY = (W3_M * W3_CY + PL_M * PL_CY) / L6_M;
constexpr double L6_D1 = W3_CY - L6_CY, L6_D2 = PL_CY - L6_CY;
constexpr double L6_IXX = W3_IXX + PL_IXX + W3_M * L6_D1 * L6_D1 + PL_M * L6_D2 * L6_D2;
constexpr double L6_IYY = W3_IYY + PL_IYY;
constexpr double L6_IZZ = W3_IZZ + PL_IZZ + W3_M * L6_D1 * L6_D1 + PL_M * L6_D2 * L6_D2;
constexpr double TCP_OFFSET = MJS_UR_FLANGE_POS[1] + MJS_G2F85_TCP_Z;  // along wrist_3 local y

struct Chain {
  V3 p[7];  // body origins, 0 = base .. 6 = wrist_3
  M3 R[7];
};

// mj_kinematics specialised to the UR5e tree (include/mjs_scene_spec.h MJS_UR_BODY_*)
MJS_DEV void fk_cs(const double* cs, const double* sn, Chain& c);
MJS_DEV void fk(const double* q, Chain& c) {
  double cs[6], sn[6];
#pragma unroll
  for (int j = 0; j < 6; j++) sincos(q[j], &sn[j], &cs[j]);
  fk_cs(cs, sn, c);
}
MJS_DEV void fk_cs(const double* cs, const double* sn, Chain& c) {
  double s, co;
  c.R[0] = M3{v3(-1, 0, 0), v3(0, -1, 0), v3(0, 0, 1)};  // base quat (0,0,0,-1): Rz(180deg), robot.py:320
  c.p[0] = v3(0, 0, 0);
  s = sn[0]; co = cs[0];  // shoulder: hinge z
  c.p[1] = madd(c.p[0], MJS_UR_BODY_POS[1][2], c.R[0].cz);
  c.R[1] = mul_rot_z(c.R[0], co, s);
  s = sn[1]; co = cs[1];  // upper arm: Ry(90) then hinge y
  c.p[2] = madd(c.p[1], MJS_UR_BODY_POS[2][1], c.R[1].cy);
  c.R[2] = mul_rot_y(mul_quarter_y(c.R[1]), co, s);
  s = sn[2]; co = cs[2];  // forearm: hinge y
  c.p[3] = madd(madd(c.p[2], MJS_UR_BODY_POS[3][1], c.R[2].cy), MJS_UR_BODY_POS[3][2], c.R[2].cz);
  c.R[3] = mul_rot_y(c.R[2], co, s);
  s = sn[3]; co = cs[3];  // wrist 1: Ry(90) then hinge y
  c.p[4] = madd(c.p[3], MJS_UR_BODY_POS[4][2], c.R[3].cz);
  c.R[4] = mul_rot_y(mul_quarter_y(c.R[3]), co, s);
  s = sn[4]; co = cs[4];  // wrist 2: hinge z
  c.p[5] = madd(c.p[4], MJS_UR_BODY_POS[5][1], c.R[4].cy);
  c.R[5] = mul_rot_z(c.R[4], co, s);
  s = sn[5]; co = cs[5];  // wrist 3: hinge y
  c.p[6] = madd(c.p[5], MJS_UR_BODY_POS[6][2], c.R[5].cz);
  c.R[6] = mul_rot_y(c.R[5], co, s);
}
MJS_DEV V3 joint_axis(const Chain& c, int j) {  // j = 0..5, joint j on body j+1
  return (j == 0 || j == 4) ? c.R[j + 1].cz : c.R[j + 1].cy;
}
MJS_DEV V3 tcp_position(const Chain& c) { return madd(c.p[6], TCP_OFFSET, c.R[6].cy); }

// spatial inertia about the world origin: axisymmetric-about-local-z tensor (a,a,cz) at COM com
MJS_DEV SI si_axisym(V3 axis, V3 com, double mass, double a, double cz) {
  double k = cz - a, c2 = dot(com, com);
  SI I;
  I.xx = a + k * axis.x * axis.x + mass * (c2 - com.x * com.x);
  I.xy = k * axis.x * axis.y - mass * com.x * com.y;
  I.xz = k * axis.x * axis.z - mass * com.x * com.z;
  I.yy = a + k * axis.y * axis.y + mass * (c2 - com.y * com.y);
  I.yz = k * axis.y * axis.z - mass * com.y * com.z;
  I.zz = a + k * axis.z * axis.z + mass * (c2 - com.z * com.z);
  I.h = mass * com;
  I.m = mass;
  return I;
}
MJS_DEV SI si_diag(M3 R, V3 com, double mass, double dx, double dy, double dz) {
  double c2 = dot(com, com);
  SI I;
  I.xx = dx * R.cx.x * R.cx.x + dy * R.cy.x * R.cy.x + dz * R.cz.x * R.cz.x + mass * (c2 - com.x * com.x);
  I.xy = dx * R.cx.x * R.cx.y + dy * R.cy.x * R.cy.y + dz * R.cz.x * R.cz.y - mass * com.x * com.y;
  I.xz = dx * R.cx.x * R.cx.z + dy * R.cy.x * R.cy.z + dz * R.cz.x * R.cz.z - mass * com.x * com.z;
  I.yy = dx * R.cx.y * R.cx.y + dy * R.cy.y * R.cy.y + dz * R.cz.y * R.cz.y + mass * (c2 - com.y * com.y);
  I.yz = dx * R.cx.y * R.cx.z + dy * R.cy.y * R.cy.z + dz * R.cz.y * R.cz.z - mass * com.y * com.z;
  I.zz = dx * R.cx.z * R.cx.z + dy * R.cy.z * R.cy.z + dz * R.cz.z * R.cz.z + mass * (c2 - com.z * com.z);
  I.h = mass * com;
  I.m = mass;
  return I;
}

// One forward-dynamics evaluation + implicitfast solve: returns the acceleration that the
// integrator applies, (M - dt*dF/dv)^-1 (qfrc_smooth) with no constraint rows active.
MJS_DEV void dynamics_generic(const double* q, const double* v, const double* ctrl, double* qacc_int) {
  Chain c;
  fk(q, c);
  // motion subspaces about the world origin
  SV S[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    V3 ax = joint_axis(c, j);
    S[j] = SV{ax, cross(c.p[j + 1], ax)};
  }
  // body inertias (bodies 1..6), all links axisymmetric about local z except the merged last link
  SI I[NJ];
  I[0] = si_axisym(c.R[1].cz, c.p[1], MJS_UR_BODY_MASS[1], MJS_UR_BODY_DIAGINERTIA[1][0], MJS_UR_BODY_DIAGINERTIA[1][2]);
  I[1] = si_axisym(c.R[2].cz, madd(c.p[2], MJS_UR_BODY_IPOS[2][2], c.R[2].cz), MJS_UR_BODY_MASS[2], MJS_UR_BODY_DIAGINERTIA[2][0], MJS_UR_BODY_DIAGINERTIA[2][2]);
  I[2] = si_axisym(c.R[3].cz, madd(c.p[3], MJS_UR_BODY_IPOS[3][2], c.R[3].cz), MJS_UR_BODY_MASS[3], MJS_UR_BODY_DIAGINERTIA[3][0], MJS_UR_BODY_DIAGINERTIA[3][2]);
  I[3] = si_axisym(c.R[4].cz, madd(c.p[4], MJS_UR_BODY_IPOS[4][1], c.R[4].cy), MJS_UR_BODY_MASS[4], MJS_UR_BODY_DIAGINERTIA[4][0], MJS_UR_BODY_DIAGINERTIA[4][2]);
  I[4] = si_axisym(c.R[5].cz, madd(c.p[5], MJS_UR_BODY_IPOS[5][2], c.R[5].cz), MJS_UR_BODY_MASS[5], MJS_UR_BODY_DIAGINERTIA[5][0], MJS_UR_BODY_DIAGINERTIA[5][2]);
  I[5] = si_diag(c.R[6], madd(c.p[6], L6_CY, c.R[6].cy), L6_M, L6_IXX, L6_IYY, L6_IZZ);
  // velocities and Coriolis accelerations (mj_comVel, mj_rne forward pass without gravity)
  SV vel[NJ], acc[NJ], F[NJ];
  {
    SV vp = SV{v3(0, 0, 0), v3(0, 0, 0)}, ap = vp;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      acc[j] = ap + v[j] * cross_motion(vp, S[j]);
      vel[j] = vp + v[j] * S[j];
      vp = vel[j];
      ap = acc[j];
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) F[j] = si_mul(I[j], acc[j]) + cross_force(vel[j], si_mul(I[j], vel[j]));
  // gravity: compensated on the arm bodies, acting on the payload only
  {
    V3 cpl = madd(c.p[6], PL_CY, c.R[6].cy);
    V3 W = v3(0, 0, PL_M * MJS_GRAVITY_Z);
    F[5].w = F[5].w - cross(cpl, W);
    F[5].v = F[5].v - W;
  }
#pragma unroll
  for (int j = NJ - 2; j >= 0; j--) F[j] = F[j] + F[j + 1];
  // composite inertias and the joint-space inertia matrix (mj_crb), lower triangle
#pragma unroll
  for (int j = NJ - 2; j >= 0; j--) I[j] = I[j] + I[j + 1];
  double A[NJ][NJ], rhs[NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    SV f = si_mul(I[i], S[i]);
#pragma unroll
    for (int j = 0; j <= i; j++) A[i][j] = dot(S[j], f);
  }
  // actuators (mj_fwdActuation) and the implicitfast system matrix
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    double cj = clampd(ctrl[j], MJS_UR_ACT_CTRLRANGE[j][0], MJS_UR_ACT_CTRLRANGE[j][1]);
    double f = MJS_UR_ACT_KP[j] * cj + 0.0 + (-MJS_UR_ACT_KP[j]) * q[j] + (-MJS_UR_ACT_KD[j]) * v[j];
    double fc = clampd(f, -MJS_UR_ACT_FRC[j], MJS_UR_ACT_FRC[j]);
    bool clamped = (fc <= -MJS_UR_ACT_FRC[j]) || (fc >= MJS_UR_ACT_FRC[j]);
    rhs[j] = -dot(S[j], F[j]) + fc;  // qfrc_smooth = passive - bias + actuator
    A[j][j] += MJS_UR_ARMATURE;
    if (!clamped) A[j][j] += MJS_RR_PHYSICS_DT * MJS_UR_ACT_KD[j];  // -dt * d(actuator)/dv
  }
  // LDL^T factorisation and solve, fully unrolled (L overwrites the strict lower triangle of A)
  double Dg[NJ], Dinv[NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j < i; j++) {
      double s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k] * Dg[k];
      A[i][j] = s * Dinv[j];
    }
    double d = A[i][i];
#pragma unroll
    for (int k = 0; k < i; k++) d -= A[i][k] * A[i][k] * Dg[k];
    Dg[i] = d;
    Dinv[i] = 1.0 / d;
  }
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int k = 0; k < i; k++) rhs[i] -= A[i][k] * rhs[k];
  }
#pragma unroll
  for (int i = 0; i < NJ; i++) rhs[i] *= Dinv[i];
#pragma unroll
  for (int i = NJ - 1; i >= 0; i--) {
#pragma unroll
    for (int k = i + 1; k < NJ; k++) rhs[i] -= A[k][i] * rhs[k];
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) qacc_int[j] = rhs[j];
}


// 1/x from the hardware estimate plus two Newton steps (error ~1 ulp): half the dependent-chain
// length of the IEEE division sequence, and the 6 pivot reciprocals of the LDL^T are serial.
MJS_DEV double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// actuator forces (mj_fwdActuation): fixed gain + affine bias, ctrl and force clamps; returns the
// bit mask of force-clamped actuators (their velocity derivative is dropped by implicitfast)
MJS_DEV int actuator_forces(const double* q, const double* v, const double* ctrl, double* fact) {
  int clamped = 0;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    double cj = clampd(ctrl[j], MJS_UR_ACT_CTRLRANGE[j][0], MJS_UR_ACT_CTRLRANGE[j][1]);
    double f = MJS_UR_ACT_KP[j] * cj + 0.0 + (-MJS_UR_ACT_KP[j]) * q[j] + (-MJS_UR_ACT_KD[j]) * v[j];
    double fc = clampd(f, -MJS_UR_ACT_FRC[j], MJS_UR_ACT_FRC[j]);
    if ((fc <= -MJS_UR_ACT_FRC[j]) || (fc >= MJS_UR_ACT_FRC[j])) clamped |= 1 << j;
    fact[j] = fc;
  }
  return clamped;
}

// implicitfast system matrix A = M + armature - dt * d(actuator)/dv and its factorisation
// A = U D U^T with U unit UPPER triangular, eliminating from the wrist end (joint 5) towards the
// base: the CRBA produces the wrist rows of M first, so the serial pivot chain overlaps the rest of
// the CRBA instead of starting after it (MuJoCo's L^T D L has the same order for sparsity).
// Storage: symmetric entry (i,j), i < j, lives in A[j][i]; on return U(i,j) is in A[j][i], 1/D in Dinv.
MJS_DEV void factor_system(double A[NJ][NJ], int clamped, double* Dinv, double dt = MJS_RR_PHYSICS_DT) {
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    A[j][j] += MJS_UR_ARMATURE;
    if (!((clamped >> j) & 1)) A[j][j] += dt * MJS_UR_ACT_KD[j];
  }
  double Dg[NJ];
#pragma unroll
  for (int j = NJ - 1; j >= 0; j--) {
    double d = A[j][j];
#pragma unroll
    for (int m = j + 1; m < NJ; m++) d -= A[m][j] * A[m][j] * Dg[m];
    Dg[j] = d;
    Dinv[j] = rcp_fast(d);
#pragma unroll
    for (int i = 0; i < j; i++) {
      double s = A[j][i];
#pragma unroll
      for (int m = j + 1; m < NJ; m++) s -= A[m][i] * A[m][j] * Dg[m];
      A[j][i] = s * Dinv[j];
    }
  }
}
MJS_DEV void actuate_and_factor(const double* q, const double* v, const double* ctrl, double A[NJ][NJ], double* Dinv, double* fact) {
  int clamped = actuator_forces(q, v, ctrl, fact);
  factor_system(A, clamped, Dinv);
}
// explicit inverse of the unit upper-triangular factor: V = U^-1 (V(i,j), i < j, stored in W[j][i]),
// so that x = V^T (Dinv .* (V b)) is two shallow mat-vecs (dependency depth ~5 each) instead of the
// ~25 serial steps of forward/backward substitution. Used where the solve is on the critical path.
MJS_DEV void invert_unit_upper(const double A[NJ][NJ], double W[NJ][NJ]) {
#pragma unroll
  for (int j = NJ - 1; j >= 1; j--) {
#pragma unroll
    for (int i = j - 1; i >= 0; i--) {
      double s = A[j][i];  // U(i,j)
#pragma unroll
      for (int k = i + 1; k < j; k++) s += A[k][i] * W[j][k];  // U(i,k) * V(k,j)
      W[j][i] = -s;
    }
  }
}
MJS_DEV void apply_inverse(const double W[NJ][NJ], const double* Dinv, const double* b, double* x) {
  double z[NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    double y = b[i];
#pragma unroll
    for (int j = i + 1; j < NJ; j++) y += W[j][i] * b[j];  // (V b)_i
    z[i] = y * Dinv[i];
  }
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    double v = z[i];
#pragma unroll
    for (int k = 0; k < i; k++) v += W[i][k] * z[k];  // (V^T z)_i
    x[i] = v;
  }
}

// solve U D U^T x = rhs in place
MJS_DEV void udu_solve(const double A[NJ][NJ], const double* Dinv, double* rhs) {
#pragma unroll
  for (int i = NJ - 1; i >= 0; i--) {
#pragma unroll
    for (int m = i + 1; m < NJ; m++) rhs[i] -= A[m][i] * rhs[m];
  }
#pragma unroll
  for (int i = 0; i < NJ; i++) rhs[i] *= Dinv[i];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int m = 0; m < i; m++) rhs[i] -= A[i][m] * rhs[m];
  }
}
MJS_DEV void actuate_and_solve(const double* q, const double* v, const double* ctrl, double A[NJ][NJ], double* rhs) {
  double Dinv[NJ], fact[NJ];
  actuate_and_factor(q, v, ctrl, A, Dinv, fact);
#pragma unroll
  for (int j = 0; j < NJ; j++) rhs[j] += fact[j];  // qfrc_smooth = passive - bias + actuator
  udu_solve(A, Dinv, rhs);
}

// ------------------------------------------------------------------ joint-limit rows
// mj_instantiateLimit / mj_makeImpedance / mj_referenceConstraint + the primal Newton solver
// (mj_solPrimal) for the only constraint rows the Robot-Reach scene can activate: joint limits
// (MJS_UR_JNT_RANGE; J = +-e_j). 12 static slots: 2j = lower side of joint j, 2j+1 = upper side, the
// oracle's row order. The solver is cold-started at qacc_smooth (the kernel does not carry
// qacc_warmstart; the strictly convex problem has one minimiser, so this agrees with the oracle to
// the solver tolerance; fault bit MJS_FAULT_LIMIT_COLDSTART reports that rows were active).
constexpr int NLIM = 2 * NJ;
struct LimitRows {
  double D[NLIM], aref[NLIM];
  bool on[NLIM];
};
MJS_DEV bool build_limit_rows(const double* q, const double* v, LimitRows& r) {
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  bool any = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int k = 2 * j + side;
      const double sgn = side == 0 ? 1.0 : -1.0;  // row Jacobian entry on dof j
      double dist = side == 0 ? q[j] - MJS_UR_JNT_RANGE[j][0] : MJS_UR_JNT_RANGE[j][1] - q[j];
      bool on = dist < 0.0;  // jnt_margin = 0
      double imp = impedance_default(dist);
      double R = fmax(MJS_MINVAL, (1 - imp) * UR5E_DOF_INVWEIGHT0[j] / imp);
      r.on[k] = on;
      r.D[k] = 1 / R;
      r.aref[k] = -B * (sgn * v[j]) - K * imp * dist;
      any = any || on;
    }
  }
  return any;
}
// reciprocal square root from v_rsq_f64 + two Newton steps (full double precision to ~1 ulp): the 6x6 Cholesky below
// is called several times per substep by the constraint stages and IEEE sqrt + division sequences dominated it
MJS_DEV double rsqrt_fast(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
// in-place Cholesky of a symmetric positive definite 6x6 (lower triangle used). The DIAGONAL of the factor is
// stored as its RECIPROCAL (chol6_solve multiplies instead of dividing).
MJS_DEV bool chol6(double H[NJ][NJ]) {
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double s = H[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= H[i][k] * H[j][k];
      if (i == j) {
        if (s < MJS_MINVAL) return false;
        H[i][i] = rsqrt_fast(s);
      } else
        H[i][j] = s * H[j][j];
    }
  }
  return true;
}
MJS_DEV void chol6_solve(const double L[NJ][NJ], double* x) {
#pragma unroll
  for (int i = 0; i < NJ; i++) {
    double s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
    x[i] = s * L[i][i];
  }
#pragma unroll
  for (int i = NJ - 1; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < NJ; k++) s -= L[k][i] * x[k];
    x[i] = s * L[i][i];
  }
}
// constraint force in joint space for the active limit rows; Mf = full symmetric M incl. armature
__device__ __noinline__ void solve_limits(const double Mf[NJ][NJ], const double* qfrc_smooth, const LimitRows& r, double* qfrc_constraint) {
  double L[NJ][NJ], a[NJ], a_s[NJ], Ma[NJ], jar[NLIM], force[NLIM];
  bool active[NLIM];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j < NJ; j++) L[i][j] = Mf[i][j];
    a_s[i] = qfrc_smooth[i];
  }
  chol6(L);
  chol6_solve(L, a_s);  // qacc_smooth
#pragma unroll
  for (int i = 0; i < NJ; i++) a[i] = a_s[i];
  auto update = [&](double& cost_out) {
    double cost = 0;
#pragma unroll
    for (int k = 0; k < NLIM; k++) {
      bool act = r.on[k] && jar[k] < 0;
      active[k] = act;
      force[k] = act ? -r.D[k] * jar[k] : 0.0;
      if (act) cost += 0.5 * r.D[k] * jar[k] * jar[k];
    }
    double gauss = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) gauss += (Ma[i] - qfrc_smooth[i]) * (a[i] - a_s[i]);
    cost_out = cost + 0.5 * gauss;
  };
  auto refresh = [&]() {
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) m += Mf[i][k] * a[k];
      Ma[i] = m;
    }
#pragma unroll
    for (int k = 0; k < NLIM; k++) jar[k] = -r.aref[k] + ((k & 1) ? -a[k >> 1] : a[k >> 1]);
  };
  refresh();
  double cost;
  update(cost);
  const double scale = 1 / (UR5E_MEANINERTIA * NJ);
#pragma unroll 1
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    double grad[NJ], search[NJ], Mv[NJ], H[NJ][NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      grad[i] = Ma[i] - qfrc_smooth[i] - (force[2 * i] - force[2 * i + 1]);
#pragma unroll
      for (int j = 0; j <= i; j++) H[i][j] = Mf[i][j];
      if (active[2 * i]) H[i][i] += r.D[2 * i];
      if (active[2 * i + 1]) H[i][i] += r.D[2 * i + 1];
      search[i] = -grad[i];
    }
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
    for (int i = 0; i < NJ; i++) { g1 += search[i] * (Ma[i] - qfrc_smooth[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
    // 1-D Newton with bracketing, MuJoCo's gradient tolerance (same routine as the pointmass kernel)
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale;
    double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 50; it++) {
      double d1 = g1 + alpha * g2, d2 = g2;
#pragma unroll
      for (int k = 0; k < NLIM; k++) {
        double jv = (k & 1) ? -search[k >> 1] : search[k >> 1];
        double x = jar[k] + alpha * jv;
        if (r.on[k] && x < 0) { d1 += r.D[k] * x * jv; d2 += r.D[k] * jv * jv; }
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
#pragma unroll
    for (int k = 0; k < NLIM; k++) jar[k] += alpha * ((k & 1) ? -search[k >> 1] : search[k >> 1]);
    double oldcost = cost;
    update(cost);
    double gn = 0;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
      double g = Ma[i] - qfrc_smooth[i] - (force[2 * i] - force[2 * i + 1]);
      gn += g * g;
    }
    if (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE) break;
  }
#pragma unroll
  for (int i = 0; i < NJ; i++) qfrc_constraint[i] = force[2 * i] - force[2 * i + 1];
}

// One forward-dynamics evaluation + implicitfast solve: (M - dt*dF/dv)^-1 qfrc_smooth, no
// constraint rows active. M(q) and the bias forces come from the generated straight-line code
// (tools/gen_ur5e_dynamics.py: link-local CRBA + RNE with every structural zero folded).
MJS_DEV void dynamics(const double* q, const double* v, const double* ctrl, const double* cs, const double* sn, double* qacc_int, bool& limit_rows_active) {
#if MJS_REACH_GENERIC_DYNAMICS
  dynamics_generic(q, v, ctrl, qacc_int);
#else
  double M[21], bias[NJ], A[NJ][NJ], rhs[NJ], Dinv[NJ], fact[NJ];
  ur5e_dynamics_gen(cs, sn, v, M, bias);
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) A[i][j] = M[i * (i + 1) / 2 + j];
  }
  const int clamped = actuator_forces(q, v, ctrl, fact);
#pragma unroll
  for (int j = 0; j < NJ; j++) rhs[j] = fact[j] - bias[j];  // qfrc_smooth = passive - bias + actuator
  LimitRows rows;
  if (build_limit_rows(q, v, rows)) {  // rare: some joint is beyond its range
    double Mf[NJ][NJ], fc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) {
#pragma unroll
      for (int j = 0; j < NJ; j++) Mf[i][j] = i >= j ? A[i][j] : A[j][i];
      Mf[i][i] += MJS_UR_ARMATURE;
    }
    solve_limits(Mf, rhs, rows, fc);
#pragma unroll
    for (int j = 0; j < NJ; j++) rhs[j] += fc[j];  // + qfrc_constraint
    limit_rows_active = true;
  }
  factor_system(A, clamped, Dinv);
  udu_solve(A, Dinv, rhs);
#pragma unroll
  for (int j = 0; j < NJ; j++) qacc_int[j] = rhs[j];
#endif
}

// cos/sin of q + d from cos/sin of q by the angle-addition formulas with a short Taylor kernel
// for the increment (|d| = dt*|qdot| is ~1e-2 at most; truncation error < 1e-18 for |d| <= 0.1).
// The exact sincos is re-evaluated at the start of every control step, so rounding drift is
// bounded by 20 substeps (~1e-15). The caller falls back to sincos when sum(d^2) > 0.01.
MJS_DEV void rotate_small(double& c, double& s, double d) {
  double z = d * d;
  double sd = d * (1.0 + z * (-1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880)))));
  double cd = 1.0 + z * (-0.5 + z * (1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800)))));
  double c2 = c * cd - s * sd, s2 = s * cd + c * sd;
  c = c2;
  s = s2;
}

// ------------------------------------------------------------------------- analytic IK
// (third-party ur_analytic_ik, call site robot.py:33-37; Hawkins 2013; same decision logic as
// oracle/om_ik.c). Pose = rotation R (row-major r[9]) + translation of the FLANGE.
MJS_DEV double wrap_pi(double x) {
  while (x > PI) x -= 2 * PI;
  while (x <= -PI) x += 2 * PI;
  return x;
}
MJS_DEV double clamp_unit(double x, bool& ok) {
  if (x > 1.0) { if (x > 1.0 + 1e-9) ok = false; return 1.0; }
  if (x < -1.0) { if (x < -1.0 - 1e-9) ok = false; return -1.0; }
  return x;
}
struct Aff {
  double r[9];
  double t[3];
};
MJS_DEV Aff aff_mul(const Aff& a, const Aff& b) {
  Aff o;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) o.r[3 * i + j] = a.r[3 * i] * b.r[j] + a.r[3 * i + 1] * b.r[3 + j] + a.r[3 * i + 2] * b.r[6 + j];
    o.t[i] = a.r[3 * i] * b.t[0] + a.r[3 * i + 1] * b.t[1] + a.r[3 * i + 2] * b.t[2] + a.t[i];
  }
  return o;
}
MJS_DEV Aff aff_inv(const Aff& a) {
  Aff o;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) o.r[3 * i + j] = a.r[3 * j + i];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) o.t[i] = -(o.r[3 * i] * a.t[0] + o.r[3 * i + 1] * a.t[1] + o.r[3 * i + 2] * a.t[2]);
  return o;
}
// DH transform Rz(th) Tz(d) Tx(a) Rx(al), al in {+pi/2, 0, -pi/2} given by (ca, sa)
MJS_DEV Aff dh(double th, double d, double a, double ca, double sa) {
  double st, ct;
  sincos(th, &st, &ct);
  Aff o;
  o.r[0] = ct; o.r[1] = -st * ca; o.r[2] = st * sa;
  o.r[3] = st; o.r[4] = ct * ca; o.r[5] = -ct * sa;
  o.r[6] = 0; o.r[7] = sa; o.r[8] = ca;
  o.t[0] = a * ct; o.t[1] = a * st; o.t[2] = d;
  return o;
}

// squared distance of one joint after mapping it to the 2*pi-equivalent nearest the guess
MJS_DEV double joint_dist2(double& qj, double guess) {
  double alt = qj + (guess > qj ? 2 * PI : -2 * PI);
  if (fabs(alt - guess) < fabs(qj - guess) && fabs(alt) <= 2 * PI) qj = alt;
  return (qj - guess) * (qj - guess);
}

// DH transform with the joint's cos/sin already known
MJS_DEV Aff dh_cs(double ct, double st, double d, double a, double ca, double sa) {
  Aff o;
  o.r[0] = ct; o.r[1] = -st * ca; o.r[2] = st * sa;
  o.r[3] = st; o.r[4] = ct * ca; o.r[5] = -ct * sa;
  o.r[6] = 0; o.r[7] = sa; o.r[8] = ca;
  o.t[0] = a * ct; o.t[1] = a * st; o.t[2] = d;
  return o;
}

// Closest of the (up to) 8 closed-form solutions to q_guess; false when none exists.
// Same winner as the oracle's exhaustive loop (oracle/om_ik.c: first strict minimum in index order
// idx = s1*4 + s5*2 + s3): exact ties are broken by the smaller idx, so the visiting order is free.
// The branch the guess itself lies on is visited first; every other candidate is abandoned as soon
// as the squared distance of the joints computed so far exceeds the best complete candidate.
// Intermediates are shared per theta1 and per (theta1, theta5) exactly as in the nested loops of
// the oracle; known sines/cosines are reused (sin(acos c) = sqrt(1-c^2), cos/sin(atan2(y,x)) =
// (x,y)/hypot, atan2(-y,x) = -atan2(y,x)) and theta4 = theta234 - theta2 - theta3.
// One theta1 branch (s1) of the closed form: its up to four candidates (s5, s3), the (s5p, s3p) pair first, each
// abandoned as soon as the squared distance of the joints computed so far exceeds `best` (in/out, with best_idx and
// q_out). ik_closest visits the guess's own branch first; the four-wavefront kernel gives each branch to its own
// wavefront (best = INFINITY there: nothing is known about the other branch) and merges by (distance, idx).
MJS_DEV void ik_branch(const Aff& T, const double* g, double psi, double phi, int s1, int s5p, int s3p, double& best, int& best_idx, double* q_out) {
  const double d1 = MJS_UR_DH_D1, a2 = MJS_UR_DH_A2, a3 = MJS_UR_DH_A3, d4 = MJS_UR_DH_D4, d5 = MJS_UR_DH_D5, d6 = MJS_UR_DH_D6;
  // R60 = R06^T: X60 = (r0, r1), Y60 = (r3, r4) read column-wise from T
  const double X60x = T.r[0], X60y = T.r[1], Y60x = T.r[3], Y60y = T.r[4];
  double th1 = psi + (s1 ? -phi : phi) + 0.5 * PI;
  double q0 = wrap_pi(th1);
  const double p0 = joint_dist2(q0, g[0]);
  if (p0 > best) return;
  double sn1, c1;
  sincos(th1, &sn1, &c1);
  bool ok5 = true;
  const double c5 = clamp_unit((T.t[0] * sn1 - T.t[1] * c1 - d4) / d6, ok5);
  if (!ok5) return;
  const double ac5 = acos(c5), root5 = sqrt(fmax(0.0, 1.0 - c5 * c5));
  const Aff T01i_T = aff_mul(aff_inv(dh_cs(c1, sn1, d1, 0, 0, 1)), T);
#pragma unroll 1
  for (int b = 0; b < 2; b++) {
    const int s5 = s5p ^ b;
    const double sg5 = s5 ? -1.0 : 1.0;
    double q4 = wrap_pi(sg5 * ac5);
    const double p4 = joint_dist2(q4, g[4]);
    if (p0 + p4 > best) continue;
    const double sn5 = sg5 * root5;
    double th6 = 0, c6 = 1.0, sn6 = 0.0;
    if (!(fabs(sn5) < 1e-12)) {
      double y6 = (-X60y * sn1 + Y60y * c1) / sn5, x6 = (X60x * sn1 - Y60x * c1) / sn5;
      th6 = atan2(y6, x6);
      double h6 = sqrt(x6 * x6 + y6 * y6);
      if (h6 > 0) { c6 = x6 / h6; sn6 = y6 / h6; }
    }
    double q5 = wrap_pi(th6);
    const double p5 = joint_dist2(q5, g[5]);
    if (p0 + p4 + p5 > best) continue;
    const Aff T14 = aff_mul(T01i_T, aff_inv(aff_mul(dh_cs(c5, sn5, d5, 0, 0, -1), dh_cs(c6, sn6, d6, 0, 1, 0))));
    const double px = T14.t[0], py = T14.t[1], r2 = px * px + py * py;
    bool ok3 = true;
    const double c3 = clamp_unit((r2 - a2 * a2 - a3 * a3) / (2 * a2 * a3), ok3);
    if (!ok3) continue;
    const double ac3 = acos(c3), root3 = sqrt(fmax(0.0, 1.0 - c3 * c3));
    const double base2 = atan2(py, px), at3 = atan2(a3 * root3, a2 + a3 * c3), th234 = atan2(T14.r[3], T14.r[0]);
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
      const int s3 = s3p ^ c;
      const double sg3 = s3 ? -1.0 : 1.0;
      const double th3 = sg3 * ac3;
      double q2 = wrap_pi(th3);
      const double p2 = joint_dist2(q2, g[2]);
      if (p0 + p4 + p5 + p2 > best) continue;
      const double th2 = base2 - sg3 * at3;
      const double th4 = th234 - th2 - th3;  // rotation of frame 1->4 is Rz(th2+th3+th4)
      double q1 = wrap_pi(th2), q3 = wrap_pi(th4);
      if (!(isfinite(q0) && isfinite(q1) && isfinite(q2) && isfinite(q3) && isfinite(q4) && isfinite(q5))) continue;
      // same summation order as the exhaustive evaluation: joints 0..5
      double dist = p0;
      dist += joint_dist2(q1, g[1]);
      dist += p2;
      dist += joint_dist2(q3, g[3]);
      dist += p4;
      dist += p5;
      const int idx = (s1 << 2) | (s5 << 1) | s3;
      if (dist < best || (dist == best && idx < best_idx)) {
        best = dist;
        best_idx = idx;
        q_out[0] = q0; q_out[1] = q1; q_out[2] = q2; q_out[3] = q3; q_out[4] = q4; q_out[5] = q5;
      }
    }
  }
}
// what every branch shares: the wrist-centre angles; false when the pose is out of reach for every branch
MJS_DEV bool ik_setup(const Aff& T, double& psi, double& phi) {
  const double d4 = MJS_UR_DH_D4, d6 = MJS_UR_DH_D6;
  double p05x = T.t[0] - d6 * T.r[2], p05y = T.t[1] - d6 * T.r[5];
  double rxy = sqrt(p05x * p05x + p05y * p05y);
  if (rxy < fabs(d4)) return false;
  psi = atan2(p05y, p05x);
  phi = acos(d4 / rxy);
  return true;
}
MJS_DEV bool ik_closest(const Aff& T, const double* g, double* q_out) {
  double psi, phi;
  if (!ik_setup(T, psi, phi)) return false;
  // branch of the guess
  double ta = wrap_pi(psi + phi + 0.5 * PI), tb = wrap_pi(psi - phi + 0.5 * PI);
  const int s1p = joint_dist2(tb, g[0]) < joint_dist2(ta, g[0]) ? 1 : 0;
  const int s5p = wrap_pi(g[4]) < 0 ? 1 : 0, s3p = wrap_pi(g[2]) < 0 ? 1 : 0;
  double best = INFINITY;
  int best_idx = 8;
#pragma unroll 1
  for (int a = 0; a < 2; a++) ik_branch(T, g, psi, phi, s1p ^ a, s5p, s3p, best, best_idx, q_out);
  return best_idx < 8;
}

// TCP pose (position + scalar-LAST quaternion, type_aliases.py:6-10) -> joints (robot.py:113-121,138-151)
MJS_DEV bool tcp_pose_to_joints_offset(const double* pos, double tcp_z, const double* q_guess, double* q_out);
MJS_DEV bool tcp_pose_to_joints(const double* pos, const double* q_guess, double* q_out) {
  return tcp_pose_to_joints_offset(pos, MJS_G2F85_TCP_Z, q_guess, q_out);
}
// flange pose of a TCP pose: TCP position + scalar-LAST quaternion, tcp_z = TCP offset of the attached end effector along
// the flange z axis (gripper 0.174, CylinderEEF 0.1, bare flange 0)  (robot.py:138-151)
MJS_DEV Aff flange_pose_of_tcp(const double* pos, const double* quat_xyzw, double tcp_z) {
  double x = quat_xyzw[0], y = quat_xyzw[1], z = quat_xyzw[2], w = quat_xyzw[3];
  double n = sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  Aff T;
  T.r[0] = 1 - 2 * (y * y + z * z); T.r[1] = 2 * (x * y - z * w); T.r[2] = 2 * (x * z + y * w);
  T.r[3] = 2 * (x * y + z * w); T.r[4] = 1 - 2 * (x * x + z * z); T.r[5] = 2 * (y * z - x * w);
  T.r[6] = 2 * (x * z - y * w); T.r[7] = 2 * (y * z + x * w); T.r[8] = 1 - 2 * (x * x + y * y);
  T.t[0] = pos[0] - T.r[2] * tcp_z;
  T.t[1] = pos[1] - T.r[5] * tcp_z;
  T.t[2] = pos[2] - T.r[8] * tcp_z;
  return T;
}
// ---- the same search for the ONE orientation every task commands: TOP_DOWN_QUATERNION = (1, 0, 0, 0) scalar-last
// (robot_reach.py:30, type_aliases.py:6-10), i.e. R = diag(1, -1, -1). With that rotation the closed form collapses
// (derivation: sympy on the DH chain, T14 = T01^-1 T06 (T45 T56)^-1):
//   cos(theta5) = (tx s1 - ty c1 - d4) / d6 = 0 up to rounding  =>  theta5 = +-pi/2, sin(theta5) = +-1
//   theta6 = atan2(-c1 / s5, s1 / s5) = theta1 -+ pi/2,  cos / sin(theta6) = (s1, -c1) * s5
//   T14:  px = tx c1 + ty s1 - s5 d5,  py = tz + d6 - d1,  theta2 + theta3 + theta4 = s5 * pi/2
// which removes two of the three acos, three of the five atan2, the 4x4 products and the divisions by sin(theta5) from the
// candidate evaluation (the IK wavefront of kernel3 is the critical path of the step's prologue: the two dynamics wavefronts
// wait for it). Same candidates, same visiting order, pruning and tie rule as ik_branch / the oracle's exhaustive loop; the
// joint angles agree with the general formulas to a few ulp (atan2(-c1, s1) vs theta1 - pi/2). A target that is not top-down
// to rounding (|cos theta5| >= 1e-8: never for this quaternion) is not handled here: the caller falls back to ik_closest.
MJS_DEV bool ik_branch_top_down(double tx, double ty, double tz, const double* g, double psi, double phi, int s1, int s5p, int s3p, double& best, int& best_idx,
                                double* q_out) {
  const double d1 = MJS_UR_DH_D1, a2 = MJS_UR_DH_A2, a3 = MJS_UR_DH_A3, d4 = MJS_UR_DH_D4, d5 = MJS_UR_DH_D5, d6 = MJS_UR_DH_D6;
  const double th1 = psi + (s1 ? -phi : phi) + 0.5 * PI;
  double q0 = wrap_pi(th1);
  const double p0 = joint_dist2(q0, g[0]);
  if (p0 > best) return true;
  double sn1, c1;
  sincos(th1, &sn1, &c1);
  bool ok5 = true;
  const double c5 = clamp_unit((tx * sn1 - ty * c1 - d4) / d6, ok5);
  if (!(fabs(c5) < 1e-8)) return false;  // not a top-down pose after all
  const double ac5 = 0.5 * PI - c5;      // acos(x) = pi/2 - x - x^3/6 - ...: exact in double for |x| < 1e-8; sqrt(1 - x^2) = 1
  const double py = tz + d6 - d1, c3den = 1.0 / (2 * a2 * a3);
#pragma unroll 1
  for (int b = 0; b < 2; b++) {
    const int s5 = s5p ^ b;
    const double sg5 = s5 ? -1.0 : 1.0;
    double q4 = wrap_pi(sg5 * ac5);
    const double p4 = joint_dist2(q4, g[4]);
    if (p0 + p4 > best) continue;
    double q5 = wrap_pi(th1 - sg5 * (0.5 * PI));
    const double p5 = joint_dist2(q5, g[5]);
    if (p0 + p4 + p5 > best) continue;
    const double px = tx * c1 + ty * sn1 - sg5 * d5, r2 = px * px + py * py;
    bool ok3 = true;
    const double c3 = clamp_unit((r2 - a2 * a2 - a3 * a3) * c3den, ok3);
    if (!ok3) continue;
    const double ac3 = acos(c3), root3 = sqrt(fmax(0.0, 1.0 - c3 * c3));
    const double base2 = atan2(py, px), at3 = atan2(a3 * root3, a2 + a3 * c3), th234 = sg5 * (0.5 * PI);
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
      const int s3 = s3p ^ c;
      const double sg3 = s3 ? -1.0 : 1.0;
      const double th3 = sg3 * ac3;
      double q2 = wrap_pi(th3);
      const double p2 = joint_dist2(q2, g[2]);
      if (p0 + p4 + p5 + p2 > best) continue;
      const double th2 = base2 - sg3 * at3;
      const double th4 = th234 - th2 - th3;
      double q1 = wrap_pi(th2), q3 = wrap_pi(th4);
      if (!(isfinite(q0) && isfinite(q1) && isfinite(q2) && isfinite(q3) && isfinite(q4) && isfinite(q5))) continue;
      double dist = p0;  // same summation order as the exhaustive evaluation: joints 0..5
      dist += joint_dist2(q1, g[1]);
      dist += p2;
      dist += joint_dist2(q3, g[3]);
      dist += p4;
      dist += p5;
      const int idx = (s1 << 2) | (s5 << 1) | s3;
      if (dist < best || (dist == best && idx < best_idx)) {
        best = dist;
        best_idx = idx;
        q_out[0] = q0; q_out[1] = q1; q_out[2] = q2; q_out[3] = q3; q_out[4] = q4; q_out[5] = q5;
      }
    }
  }
  return true;
}
// returns 1 found, 0 no solution, -1 "not top-down to rounding" (the caller uses the general search)
MJS_DEV int ik_closest_top_down_search(double tx, double ty, double tz, const double* g, double* q_out) {
  const double d4 = MJS_UR_DH_D4;
  const double rxy = sqrt(tx * tx + ty * ty);  // the wrist centre's xy is the flange's: R's z column is (0, 0, -1)
  if (rxy < fabs(d4)) return 0;
  const double psi = atan2(ty, tx), phi = acos(d4 / rxy);
  double ta = wrap_pi(psi + phi + 0.5 * PI), tb = wrap_pi(psi - phi + 0.5 * PI);
  const int s1p = joint_dist2(tb, g[0]) < joint_dist2(ta, g[0]) ? 1 : 0;
  const int s5p = wrap_pi(g[4]) < 0 ? 1 : 0, s3p = wrap_pi(g[2]) < 0 ? 1 : 0;
  double best = INFINITY;
  int best_idx = 8;
  bool top_down = true;
#pragma unroll 1
  for (int a = 0; a < 2; a++) top_down = ik_branch_top_down(tx, ty, tz, g, psi, phi, s1p ^ a, s5p, s3p, best, best_idx, q_out) && top_down;
  return top_down ? (best_idx < 8 ? 1 : 0) : -1;
}
struct IkTd { double q[NJ]; int status; };
__device__ __noinline__ IkTd ik_top_down_search_out_of_line(double tx, double ty, double tz, double g0, double g1, double g2, double g3, double g4, double g5) {
  const double g[NJ] = {g0, g1, g2, g3, g4, g5};
  IkTd o;
#pragma unroll
  for (int j = 0; j < NJ; j++) o.q[j] = g[j];
  o.status = ik_closest_top_down_search(tx, ty, tz, g, o.q);
  return o;
}
// The common case in STRAIGHT-LINE code: the candidate on the guess's own branch (s1p, s5p, s3p) is evaluated in full, with
// no loop and no divergence, and the other seven are excluded by LOWER BOUNDS of their joint distances that need no further
// inverse trigonometry: theta1' (the other shoulder branch), theta5' = -theta5 and theta6' = theta1' -+ pi/2 are arithmetic,
// and the other elbow branch of the same (theta1, theta5) is -theta3. A servo target near the current joints leaves every such
// bound pi^2-ish above the winner. Whenever a bound does not exclude its candidates strictly (or the guess branch has no
// solution) the lane asks for the full search (same winner: exact ties included, they go to the search); the caller runs it
// out of line for the wavefronts that need it. Loops with per-candidate pruning cost more in exec-mask / scalar instructions
// than in arithmetic: tools/ik_bench.py, profiles/r03_d_*.
MJS_DEV int ik_closest_top_down(double tx, double ty, double tz, const double* g, double* q_out, bool& need_search) {
  const double d1 = MJS_UR_DH_D1, a2 = MJS_UR_DH_A2, a3 = MJS_UR_DH_A3, d4 = MJS_UR_DH_D4, d5 = MJS_UR_DH_D5, d6 = MJS_UR_DH_D6;
  need_search = false;
  const double rxy = sqrt(tx * tx + ty * ty);
  if (rxy < fabs(d4)) return 0;
  const double psi = atan2(ty, tx), phi = acos(d4 / rxy);
  double ta = wrap_pi(psi + phi + 0.5 * PI), tb = wrap_pi(psi - phi + 0.5 * PI);
  const double pa = joint_dist2(ta, g[0]), pb = joint_dist2(tb, g[0]);
  const int s1p = pb < pa ? 1 : 0;
  const int s5p = wrap_pi(g[4]) < 0 ? 1 : 0, s3p = wrap_pi(g[2]) < 0 ? 1 : 0;
  // ---- the guess's branch
  const double th1 = psi + (s1p ? -phi : phi) + 0.5 * PI;
  const double q0 = s1p ? tb : ta, p0 = s1p ? pb : pa, p0_other = s1p ? pa : pb;
  // cos / sin(theta1) without a sincos call: theta1 = psi +- phi + pi/2 with (cos, sin) psi = (tx, ty) / rxy and
  // (cos, sin) phi = (d4, sqrt(rxy^2 - d4^2)) / rxy; agrees with sincos(theta1) to 2 ulp (theta1 itself carries the atan2 / acos rounding)
  const double inv_r = 1.0 / rxy, cps = tx * inv_r, sps = ty * inv_r, cph = d4 * inv_r;
  const double sph = (s1p ? -1.0 : 1.0) * sqrt(fmax(0.0, 1.0 - cph * cph));
  const double sn1 = cps * cph - sps * sph, c1 = -(sps * cph + cps * sph);  // sin(x + pi/2) = cos x, cos(x + pi/2) = -sin x, x = psi +- phi
  bool ok5 = true;
  const double c5 = clamp_unit((tx * sn1 - ty * c1 - d4) / d6, ok5);
  if (!(fabs(c5) < 1e-8)) return -1;
  const double ac5 = 0.5 * PI - c5, sg5 = s5p ? -1.0 : 1.0;
  double q4 = wrap_pi(sg5 * ac5), q4_other = wrap_pi(-sg5 * ac5);
  const double p4 = joint_dist2(q4, g[4]), p4_other = joint_dist2(q4_other, g[4]);
  double q5 = wrap_pi(th1 - sg5 * (0.5 * PI));
  const double p5 = joint_dist2(q5, g[5]);
  const double px = tx * c1 + ty * sn1 - sg5 * d5, py = tz + d6 - d1, r2 = px * px + py * py;
  bool ok3 = true;
  const double c3 = clamp_unit((r2 - a2 * a2 - a3 * a3) / (2 * a2 * a3), ok3);
  const double ac3 = acos(c3), root3 = sqrt(fmax(0.0, 1.0 - c3 * c3));
  const double sg3 = s3p ? -1.0 : 1.0, th3 = sg3 * ac3;
  double q2 = wrap_pi(th3), q2_other = wrap_pi(-th3);
  const double p2 = joint_dist2(q2, g[2]), p2_other = joint_dist2(q2_other, g[2]);
  // theta2 = atan2(py, px) - sg3 atan2(a3 sin|theta3|, a2 + a3 cos theta3): ONE atan2 of the rotated vector instead of the
  // difference of two (same angle modulo 2 pi, which wrap_pi removes; the two agree to a few ulp)
  const double A3 = a2 + a3 * c3, B3 = sg3 * (a3 * root3);
  const double th2 = atan2(py * A3 - px * B3, px * A3 + py * B3), th4 = sg5 * (0.5 * PI) - th2 - th3;
  double q1 = wrap_pi(th2), q3 = wrap_pi(th4);
  double dist = p0;  // joints 0..5, the exhaustive evaluation's order
  dist += joint_dist2(q1, g[1]);
  dist += p2;
  dist += joint_dist2(q3, g[3]);
  dist += p4;
  dist += p5;
  const bool valid = ok3 && isfinite(q0) && isfinite(q1) && isfinite(q2) && isfinite(q3) && isfinite(q4) && isfinite(q5);
  // ---- lower bounds of the other seven (a sum of some of a candidate's non-negative terms)
  //   same theta1: other theta5 (both theta3): p0 + p4';  same theta1, theta5: other theta3: p0 + p4 + p5 + p2'
  //   other theta1 (all four): p0'
  const double lb = fmin(fmin(p0 + p4_other, p0 + p4 + p5 + p2_other), p0_other);
  need_search = !valid || !(lb > dist);
  q_out[0] = q0; q_out[1] = q1; q_out[2] = q2; q_out[3] = q3; q_out[4] = q4; q_out[5] = q5;
  return 1;
}
__device__ __noinline__ bool ik_closest_general(Aff T, const double* g, double* q_out) { return ik_closest(T, g, q_out); }
static_assert(MJS_TOP_DOWN_QUAT_XYZW[0] == 1.0 && MJS_TOP_DOWN_QUAT_XYZW[1] == 0.0 && MJS_TOP_DOWN_QUAT_XYZW[2] == 0.0 && MJS_TOP_DOWN_QUAT_XYZW[3] == 0.0,
              "ik_closest_top_down is derived for R = diag(1, -1, -1)");
MJS_DEV bool tcp_pose_to_joints_offset(const double* pos, double tcp_z, const double* q_guess, double* q_out) {
  // flange = TCP - R z tcp_z with R z = (0, 0, -1) (robot.py:138-151)
  bool need_search;
  int r = ik_closest_top_down(pos[0], pos[1], pos[2] + tcp_z, q_guess, q_out, need_search);
  if (__any(r == 1 && need_search)) {  // rare: some lane's guess-branch candidate is not provably the closest
    const IkTd o = ik_top_down_search_out_of_line(pos[0], pos[1], pos[2] + tcp_z, q_guess[0], q_guess[1], q_guess[2], q_guess[3], q_guess[4], q_guess[5]);
    if (r == 1 && need_search) {
      r = o.status;
#pragma unroll
      for (int j = 0; j < NJ; j++) q_out[j] = o.q[j];
    }
  }
  if (__any(r < 0)) {  // never for this quaternion; kept so that the entry point is total
    double qg[NJ];
    const bool f = ik_closest_general(flange_pose_of_tcp(pos, MJS_TOP_DOWN_QUAT_XYZW, tcp_z), q_guess, qg);
    if (r < 0) {
      r = f ? 1 : 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) q_out[j] = qg[j];
    }
  }
  return r == 1;
}

// ----------------------------------------------------------------- contact detection
// floor plane z = 0 vs the arm's collision proxies (MJS_UR_COL_*): number of contacts MuJoCo
// would list (capsule: one per end sphere; cylinder: mjc_PlaneCylinder up to 4)
MJS_DEV int count_floor_contacts(const Chain& c) {
  int n = 0;
#pragma unroll
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    const int b = MJS_UR_COL_BODY[g];
    const M3 R = c.R[b];
    V3 gp = madd(madd(madd(c.p[b], MJS_UR_COL_POS[g][0], R.cx), MJS_UR_COL_POS[g][1], R.cy), MJS_UR_COL_POS[g][2], R.cz);
    // local quat is identity or (1,1,0,0) = Rx(90deg): geom z axis = -body y, geom x axis = body x
    const bool rot = MJS_UR_COL_QUAT[g][1] != 0.0;
    V3 axis = rot ? -R.cy : R.cz;
    const double rad = MJS_UR_COL_SIZE[g][0], half = MJS_UR_COL_SIZE[g][1];
    if (MJS_UR_COL_TYPE[g] == 3) {
      n += (gp.z - half * axis.z <= rad);
      n += (gp.z + half * axis.z <= rad);
    } else {
      // mjc_PlaneCylinder with plane normal (0,0,1) through the origin
      V3 nrm = v3(0, 0, 1);
      double dist0 = gp.z, prjaxis = axis.z;
      if (prjaxis > 0) { axis = -axis; prjaxis = -prjaxis; }
      V3 vec = prjaxis * axis - nrm;
      double len = sqrt(dot(vec, vec));
      if (len < MJS_MINVAL) vec = rad * R.cx;
      else vec = (rad / len) * vec;
      double prjvec = vec.z;
      axis = half * axis;
      prjaxis *= half;
      if (dist0 + prjaxis + prjvec > 0) continue;
      n += 1;
      n += (dist0 - prjaxis + prjvec <= 0);
      V3 side = cross(vec, axis);
      if (sqrt(dot(side, side)) > MJS_MINVAL && dist0 + prjaxis - 0.5 * prjvec <= 0) n += 2;
    }
  }
  return n;
}

}  // namespace rr
#include "mjs_arm_stage.h"
namespace rr {

struct SceneReach {  // Robot-Reach: UR5e + lumped gripper payload, no collision geom beyond the arm's own
  struct Extra {};
  MJS_DEV static double solver_scale(Extra) { return 1 / (UR5E_MEANINERTIA * NJ); }
  MJS_DEV static double dof_invweight(int j) { return UR5E_DOF_INVWEIGHT0[j]; }
  MJS_DEV static double link_invweight(int b) { return UR5E_LINK_BODY_INVWEIGHT0[b]; }
  template <class E>
  MJS_DEV static void extra_contacts(const Chain&, Extra, E) {}
  MJS_DEV static bool in_touch_site(Extra, V3) { return false; }
};

// ---- the fast path's guard ---------------------------------------------------------------------------------------
// The role-specialised fast path builds no constraint rows, so it may only run a control step during which no row can
// become active. Rows of this scene: joint limits, and contacts of the arm's collision geoms with the floor.
//   joints  A PRIORI: no joint can reach its range during the control step: margin = 0.6 rad (what the force-clamped
//           servos add within 0.1 s from rest) plus the distance the joint covers at its CURRENT velocity, |v| * 0.1 s
//           (mjs_set_state can inject any velocity).
//   floor   A PRIORI, from three facts: (1) the step STARTS with every collision geom at least CLEAR_MARGIN above the floor
//           (flag FLAG_CLEAR, computed from the final configuration of the previous step / reset / mjs_set_state: free, the
//           step's epilogue has the kinematics anyway); (2) the servo TARGET is a top-down TCP pose inside the task's own
//           action box (robot_reach.py:187-201), whose IK solutions keep the wrist geoms >= 0.15 m and the elbow far
//           above the floor; (3) the joint-space TRAVEL towards it is short: sum_j (|q1_j - q0_j| + 0.05 s |v_j|)^2 <=
//           TRAVEL2_MAX (the over-damped servos move every joint monotonically from q0 towards q1; between two
//           top-down poses of the action box the wrist stays > 0.1 m up; uniform in-box targets reach 1.1 at most, p99 0.6).
//           (2) is known at kernel entry (the action), (3) after the IK: a workgroup that fails (3) leaves the fast
//           path after substep 0, which needs no IK result and, by (1), has no rows.
// A POSTERIORI the final state is re-checked: a joint beyond its range or a geom on the floor after a row-free step
// (never observed; the bounds are not proofs) is REPORTED as MJS_FAULT_FASTPATH_VIOLATED for that env and step, so a wrong
// result is never published silently. Everything else — any action, any injected state — takes the robust path, which
// detects limits and contacts in every substep and solves them (mjs_arm_stage.h).
constexpr double CLEAR_MARGIN = 0.10, TRAVEL2_MAX = 2.5;
MJS_DEV bool joint_near_range(const double* q, const double* v) {
  bool near = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const double margin = 0.6 + MJS_RR_CONTROL_DT * fabs(v[j]);
    near = near || (q[j] - MJS_UR_JNT_RANGE[j][0] < margin) || (MJS_UR_JNT_RANGE[j][1] - q[j] < margin);
  }
  return near;
}
MJS_DEV bool joint_outside_range(const double* q) {
  bool out = false;
#pragma unroll
  for (int j = 0; j < NJ; j++) out = out || (q[j] < MJS_UR_JNT_RANGE[j][0]) || (q[j] > MJS_UR_JNT_RANGE[j][1]);
  return out;
}
MJS_DEV bool action_in_box(const double* act) {
  bool in = true;
#pragma unroll
  for (int k = 0; k < 3; k++) in = in && act[k] >= MJS_RR_SPACE_LO[k] && act[k] <= MJS_RR_SPACE_HI[k];
  return in;
}
MJS_DEV bool travel_is_long(const double* q_start, const double* q_target, const double* v) {
  double t2 = 0;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const double d = fabs(q_target[j] - q_start[j]) + 0.05 * fabs(v[j]);
    t2 = fma(d, d, t2);
  }
  return !(t2 <= TRAVEL2_MAX);
}
// FLAG_CLEAR of a configuration (mjs_set_state recomputes it: the caller may have edited the joints)
MJS_DEV bool config_is_clear(const double* q) {
  Chain c;
  fk(q, c);
  return min_floor_clearance(c) >= CLEAR_MARGIN;
}
// ncon of a configuration: the exact count only where the cheap bound says a geom may touch (rare)
MJS_DEV int floor_contacts_from_clearance(const Chain& c, double minclr) {
  int ncon = 0;
  if (__any(!(minclr > 0.0))) ncon = count_floor_contacts(c);
  return (minclr > 0.0) ? 0 : ncon;
}

// ---------------------------------------------------------------------------- kernel
struct State {
  double q[NJ], v[NJ], time, target[3];
};
MJS_DEV State load_state(const KernelParams& p, int i) {
  State st;
  const double* s = p.state + i;
  const size_t N = p.N;
#pragma unroll
  for (int j = 0; j < NJ; j++) { st.q[j] = s[(S_Q + j) * N]; st.v[j] = s[(S_V + j) * N]; }
  st.time = s[S_TIME * N];
#pragma unroll
  for (int k = 0; k < 3; k++) st.target[k] = s[(S_TARGET + k) * N];
  return st;
}
// the rows a control step changes: q, v, time (a step only reads the target / switch rows: resets write those)
MJS_DEV void store_state_stepped(const KernelParams& p, int i, const State& st) {
  double* s = p.state + i;
  const size_t N = p.N;
#pragma unroll
  for (int j = 0; j < NJ; j++) { s[(S_Q + j) * N] = st.q[j]; s[(S_V + j) * N] = st.v[j]; }
  s[S_TIME * N] = st.time;
}
MJS_DEV void store_state(const KernelParams& p, int i, const State& st) {
  store_state_stepped(p, i, st);
  double* s = p.state + i;
  const size_t N = p.N;
#pragma unroll
  for (int k = 0; k < 3; k++) s[(S_TARGET + k) * N] = st.target[k];
}
// the carried cos / sin rows (first_row = the task's S_CS; the sines follow the six cosines)
MJS_DEV void load_cs(const KernelParams& p, int i, int first_row, double* cs, double* sn) {
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const double c = p.state[(size_t)(first_row + j) * p.N + i], s_ = p.state[(size_t)(first_row + NJ + j) * p.N + i];
    const double f = fma(-0.5, fma(c, c, s_ * s_), 1.5);  // one Newton step towards unit length: 1 / sqrt(n2) ~ 1.5 - n2 / 2
    cs[j] = c * f;
    sn[j] = s_ * f;
  }
}
MJS_DEV void store_cs(const KernelParams& p, int i, int first_row, const double* cs, const double* sn) {
#pragma unroll
  for (int j = 0; j < NJ; j++) { p.state[(size_t)(first_row + j) * p.N + i] = cs[j]; p.state[(size_t)(first_row + NJ + j) * p.N + i] = sn[j]; }
}
// qacc_warmstart rows (first_row = the task's S_WARM): mj_resetData zeroes them
MJS_DEV void store_warm(const KernelParams& p, int i, int first_row, const double* w) {
#pragma unroll
  for (int j = 0; j < NJ; j++) p.state[(size_t)(first_row + j) * p.N + i] = w ? w[j] : 0.0;
}

// initialize_episode (robot_reach.py:143-150): robot xyz -> IK from qpos0 = 0 -> set joints;
// then target xyz. Returns the per-episode ik_failed flag (always clear after a reset).
// (noinline, RNG handle by value -- the kernel argument block must never have its address taken, or
// every lane copies it to scratch at kernel entry -- and by-value result: the rare reset path keeps its own IK copy out of the hot code and the
// long-lived state of the step path never has its address taken)
__device__ __noinline__ State episode_init(DevRng rng, int i) {
  State st;
  RngCursor c = rng_open(rng, i);
  double rp[3], q[NJ], zeros[NJ] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; k++) rp[k] = rng_uniform(rng, i, c, MJS_RR_SPACE_LO[k], MJS_RR_SPACE_HI[k]);
  bool ok = tcp_pose_to_joints(rp, zeros, q);
#pragma unroll
  for (int j = 0; j < NJ; j++) { st.q[j] = ok ? q[j] : 0.0; st.v[j] = 0; }
#pragma unroll
  for (int k = 0; k < 3; k++) st.target[k] = rng_uniform(rng, i, c, MJS_RR_SPACE_LO[k], MJS_RR_SPACE_HI[k]);
  rng_close(rng, i, c);
  st.time = 0;
  return st;
}
// everything a reset publishes: state, zeroed warm start, flags (fresh mjData: qacc_warmstart = 0 is the valid warm start of
// the first step), FIRST outputs. Returns nothing live.
MJS_DEV void publish_reset(const KernelParams& p, int i, const State& st, bool write_first, uint8_t extra_flags = 0) {
  Chain c;
  double obs[OBS_DIM];
  store_state(p, i, st);
  store_warm(p, i, S_WARM, nullptr);
  {
    double cs[NJ], sn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) sincos(st.q[j], &sn[j], &cs[j]);
    store_cs(p, i, S_CS, cs, sn);
    fk_cs(cs, sn, c);
  }
  const double minclr = min_floor_clearance(c);
  p.flags[i] = (uint8_t)(FLAG_WARM_VALID | (minclr >= CLEAR_MARGIN ? FLAG_CLEAR : 0) | extra_flags);  // ONE store: a concurrent reader sees the old or the new byte
  V3 tcp = tcp_position(c);
  obs[0] = tcp.x; obs[1] = tcp.y; obs[2] = tcp.z;
#pragma unroll
  for (int j = 0; j < NJ; j++) obs[3 + j] = st.q[j];
#pragma unroll
  for (int k = 0; k < 3; k++) obs[9 + k] = st.target[k];
  const int ncon = floor_contacts_from_clearance(c, minclr);
  if (write_first) write_outputs<OBS_DIM>(p, i, obs, 0.0, 1.0, MJS_STEP_FIRST, false, false, false, 0, ncon);
  else {  // same-step auto-reset: only the observation and ncon are replaced (the LAST step's reward / flags stay)
    if (p.out.obs) {
#pragma unroll
      for (int k = 0; k < OBS_DIM; k++) p.out.obs[(size_t)i * OBS_DIM + k] = obs[k];
    }
    if (p.out.ncon) p.out.ncon[i] = ncon;
  }
}

MJS_DEV void make_obs(const State& st, const Chain& c, double* obs) {
  V3 tcp = tcp_position(c);
  obs[0] = tcp.x; obs[1] = tcp.y; obs[2] = tcp.z;
#pragma unroll
  for (int j = 0; j < NJ; j++) obs[3 + j] = st.q[j];
#pragma unroll
  for (int k = 0; k < 3; k++) obs[9 + k] = st.target[k];
}

// The substeps first_substep .. 19 of one control step on ONE wavefront with every constraint row of the scene available
// (robust path: kernel_variant = single wave, and the default kernels for workgroups whose guard fails). Every substep
// checks the joint ranges and the arm's floor clearance; lanes with an active row go through the general constraint stage
// (mjs_arm_stage.h), warm-started exactly as mj_fwdConstraint does: qacc_warmstart = the previous Physics.step()'s solver
// acceleration — the stage's own result, or M^-1 qfrc_smooth of a row-free substep (evaluated lazily, only when the next
// substep turns out to have rows), or the persisted rows of the state at the control step's first substep.
// noinline + by value: keeps this rarely-taken code (and the stage's call frame) out of the role-specialised hot loop.
struct SoloIn {
  double q[NJ], v[NJ], q0[NJ], q1[NJ], cs[NJ], sn[NJ], time, t0, t1;
  double warm[NJ];
  bool has_warm;
  int first_substep;  // the substeps before it were taken on the row-free path
};
struct SoloOut {
  double q[NJ], v[NJ], cs[NJ], sn[NJ], time, warm[NJ];
  bool bad, rows_active, overflow;
};
// qacc_smooth = (M + armature)^-1 qfrc_smooth: what the solver returns for a step without rows
MJS_DEV void smooth_acceleration(const double* Marm /*21*/, const double* qs, double* out) {
  double L[NJ][NJ];
#pragma unroll
  for (int i = 0; i < NJ; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) L[i][j] = Marm[i * (i + 1) / 2 + j];
    out[i] = qs[i];
  }
  chol6(L);
  chol6_solve(L, out);
}
__device__ __noinline__ SoloOut solo_control_step(SoloIn in, Ws ws) {
  // work on register copies; the in/out structs live in the call frame
  double q[NJ], v[NJ], cs[NJ], sn[NJ], q0[NJ], q1[NJ], warm[NJ], pM[21], pqs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) { q[j] = in.q[j]; v[j] = in.v[j]; cs[j] = in.cs[j]; sn[j] = in.sn[j]; q0[j] = in.q0[j]; q1[j] = in.q1[j]; warm[j] = in.warm[j]; pqs[j] = 0; }
#pragma unroll
  for (int k = 0; k < 21; k++) pM[k] = 0;
  double time = in.time;
  const double t0 = in.t0, t1 = in.t1, inv_span = 1.0 / (in.t1 - in.t0);
  bool bad = false, rows_active = false, overflow = false;
  bool has_warm = in.has_warm, lazy = false;  // lazy: the warm start is M^-1 qfrc_smooth of (pM, pqs), not yet evaluated
  const int first_substep = __builtin_amdgcn_readfirstlane(in.first_substep);  // the same for every lane of a caller: a scalar loop counter
#pragma unroll 1
  for (int s = first_substep; s < MJS_RR_NSUB; s++) {
    // rows this substep? A joint beyond its range or a collision geom in the floor: rare
    bool rows = joint_outside_range(q);
    {
      Chain ch;
      fk_cs(cs, sn, ch);
      rows = rows || !(min_floor_clearance(ch) >= 0.0);
    }
    // The out-of-line functions below are CALLED BY THE WHOLE WAVEFRONT (wave-uniform branches only around a call, per-lane
    // selects of its results; a lane without rows discards them, so what a lane computes never depends on its neighbours):
    // hipcc places live-range split copies ahead of the exec-mask restore of a divergent join next to a call
    // (mujoco_sim_amd/_isa_lint.py), which lost the sub-step counter of the lanes that skipped the `if`.
    // 1. the warm start a row-free substep left behind as (pM, pqs), needed now: qacc_warmstart = pM^-1 pqs
    const bool eval_lazy = rows && lazy;
    if (__any(eval_lazy)) {
      double wl[NJ];
      smooth_acceleration(pM, pqs, wl);
#pragma unroll
      for (int j = 0; j < NJ; j++) warm[j] = eval_lazy ? wl[j] : warm[j];
      has_warm = has_warm || eval_lazy;
    }
    // 2. before_substep: ctrl = q0 + (q1 - q0) * (clip(t) - t0) / (t1 - t0)  (joint_trajectory.py:41-47), smooth dynamics;
    // M + armature and qfrc_smooth go straight into (pM, pqs): what a row-free substep leaves for the next one
    double t = fmin(fmax(time, t0), t1);
    double ctrl[NJ], bias[NJ], A[NJ][NJ], rhs[NJ], Dinv[NJ], fact[NJ];
    double* const M = pM;
#pragma unroll
    for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
    ur5e_dynamics_gen(cs, sn, v, M, bias);
#pragma unroll
    for (int i = 0; i < NJ; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = M[i * (i + 1) / 2 + j];
      M[i * (i + 1) / 2 + i] += MJS_UR_ARMATURE;
    }
    const int clamped = actuator_forces(q, v, ctrl, fact);
#pragma unroll
    for (int j = 0; j < NJ; j++) { rhs[j] = fact[j] - bias[j]; pqs[j] = rhs[j]; }  // qfrc_smooth = passive - bias + actuator
    lazy = !rows;
    // 3. the constraint stage
    if (__any(rows)) {
      GenStageIn gi;
#pragma unroll
      for (int j = 0; j < NJ; j++) { gi.q[j] = q[j]; gi.v[j] = v[j]; gi.cs[j] = cs[j]; gi.sn[j] = sn[j]; gi.qs[j] = rhs[j]; gi.warm[j] = warm[j]; }
#pragma unroll
      for (int k = 0; k < 21; k++) gi.M[k] = M[k];
      gi.has_warm = has_warm;
      const GenStageOut go = gen_stage<SceneReach>(gi, SceneReach::Extra{}, ws);
#pragma unroll
      for (int j = 0; j < NJ; j++) { rhs[j] = rows ? go.qs[j] : rhs[j]; warm[j] = rows ? go.qacc[j] : warm[j]; }
      has_warm = has_warm || rows;
      rows_active = rows_active || rows;
      overflow = overflow || (rows && go.overflow);
    }
    factor_system(A, clamped, Dinv);
    udu_solve(A, Dinv, rhs);
    double acc2 = 0, dq2 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      acc2 = fma(rhs[j], rhs[j], acc2);
      v[j] += MJS_RR_PHYSICS_DT * rhs[j];
      double dq = MJS_RR_PHYSICS_DT * v[j];
      q[j] += dq;
      dq2 = fma(dq, dq, dq2);
      rotate_small(cs[j], sn[j], dq);
    }
    bad = bad || !(acc2 <= 1e20);
    if (!(dq2 <= 0.01)) {
#pragma unroll
      for (int j = 0; j < NJ; j++) sincos(q[j], &sn[j], &cs[j]);
    }
    time += MJS_RR_PHYSICS_DT;
  }
  if (lazy) smooth_acceleration(pM, pqs, warm);  // the next control step's warm start
  SoloOut o;
#pragma unroll
  for (int j = 0; j < NJ; j++) { o.q[j] = q[j]; o.v[j] = v[j]; o.cs[j] = cs[j]; o.sn[j] = sn[j]; o.warm[j] = warm[j]; }
  o.time = time;
  o.bad = bad;
  o.rows_active = rows_active;
  o.overflow = overflow;
  return o;
}
// What every variant of the step does once the 20 substeps are done (role 0 / the only wavefront): observables, reward,
// termination, the a-posteriori checks, fault word, flags, state and output stores, same-step auto-reset.
struct StepEnd {
  bool solo, bad, rows_active, overflow;
  bool peer_stored_state = false;  // the other dynamics wavefront (same numbers) has written the q / v / time / cos / sin rows meanwhile
};
MJS_DEV void finish_step(const KernelParams& p, int i, int lane, State& st, const double* cs, const double* sn, uint8_t flags, const double* warm_out, StepEnd e,
                         double* obs_tile) {
  double obs[OBS_DIM];
  Chain c;
  bool bad = e.bad;
#pragma unroll
  for (int j = 0; j < NJ; j++) bad = bad || bad_value(st.q[j]) || bad_value(st.v[j]);  // mj_checkPos / mj_checkVel
  MJS_STAMP(p, 2);
  // observables, reward, termination (cos/sin carried from the last substep, <= 1e-15 from exact)
  fk_cs(cs, sn, c);
  make_obs(st, c, obs);
  double dx = obs[0] - st.target[0], dy = obs[1] - st.target[1], dz = obs[2] - st.target[2];
  double dist = sqrt(dx * dx + dy * dy + dz * dz);
  bool success = dist < MJS_RR_GOAL_THRESHOLD;
  double reward = (p.reward_type == MJS_REW_SPARSE) ? (success ? 1.0 : 0.0) : -dist;
  double discount = 1.0;
  bool terminate = false;
  if (p.terminate_on_success && success) { terminate = true; discount = 0.0; }
  if (bad) { reward = 0; discount = 0; terminate = true; }
  if (st.time >= p.time_limit) terminate = true;
  MJS_STAMP(p, 3);
  // ncon of the final configuration (mj_step1 of the last substep) and the next step's FLAG_CLEAR, both from the cheap
  // clearance bound; the exact count runs only when some lane of the wavefront may touch
  const double minclr = min_floor_clearance(c);
  const int ncon = floor_contacts_from_clearance(c, minclr);
  MJS_STAMP(p, 4);
  // fault word: the robust path solves every row of this scene (bit 8 only when a lane has more active contacts than the
  // stage's workspace holds); the row-free path's a-posteriori check is bit 16
  int fault = (bad ? MJS_FAULT_BAD_STATE : 0) | ((flags & FLAG_IK_FAILED) ? MJS_FAULT_IK_FAILED : 0) | (e.rows_active ? MJS_FAULT_LIMIT_COLDSTART : 0) |
              (e.overflow ? MJS_FAULT_UNSUPPORTED_CONTACT : 0) | ((!e.solo && (ncon > 0 || joint_outside_range(st.q))) ? MJS_FAULT_FASTPATH_VIOLATED : 0);
  bool terminated = terminate && discount == 0.0, truncated = terminate && discount > 0.0;
  // reset-groups launches mark a new "reset pending" with the launch's parity: a reset workgroup of THIS launch that reads the
  // byte after this store leaves the env to the next launch (pending_mark, mjs_kernel_common.h)
  uint8_t newflags = (uint8_t)((flags & FLAG_IK_FAILED) | (terminate ? pending_mark(p) : 0) | (minclr >= CLEAR_MARGIN ? FLAG_CLEAR : 0) | (e.solo ? FLAG_WARM_VALID : 0));
  // everything is written out BEFORE the (rare, real function call) same-step reset so that no value
  // has to stay live across that call
  if (!e.peer_stored_state) {
    store_state_stepped(p, i, st);
    store_cs(p, i, S_CS, cs, sn);
  }
  if (e.solo) store_warm(p, i, S_WARM, warm_out);
  p.flags[i] = newflags;
  // observations [N, 12] row-major: a lane-per-env store is a 96-B-strided scatter (2.5x write
  // amplification measured with WRITE_SIZE); transpose the wave's 64x12 block through LDS and write
  // it as 12 fully coalesced 512-B stores instead. Only this wavefront touches obs_tile.
  {
    KernelParams pn = p;
    pn.out.obs = nullptr;
    write_outputs<OBS_DIM>(pn, i, obs, reward, discount, terminate ? MJS_STEP_LAST : MJS_STEP_MID, terminated, truncated, success, fault, ncon);
    if (p.out.obs) {
      if (__ballot(1) == ~0ull) {  // whole wavefront alive: cooperative block store
#pragma unroll
        for (int k = 0; k < OBS_DIM; k++) obs_tile[lane * OBS_DIM + k] = obs[k];
        __builtin_amdgcn_wave_barrier();
        double* dst = p.out.obs + (size_t)blockIdx.x * 64 * OBS_DIM;
#pragma unroll
        for (int k = 0; k < OBS_DIM; k++) dst[k * 64 + lane] = obs_tile[k * 64 + lane];
      } else {  // some lanes left earlier (auto-reset path, tail of the batch): each lane writes its own row
#pragma unroll
        for (int k = 0; k < OBS_DIM; k++) p.out.obs[(size_t)i * OBS_DIM + k] = obs[k];
      }
    }
  }
  if (terminate && p.autoreset == MJS_AUTORESET_SAME_STEP) {
    if (p.out.terminal_obs) {
#pragma unroll
      for (int k = 0; k < OBS_DIM; k++) p.out.terminal_obs[(size_t)i * OBS_DIM + k] = obs[k];
    }
    State fresh = episode_init(p.rng, i);
    publish_reset(p, i, fresh, false);
  }
  MJS_STAMP(p, 5);
}

// The robust path as a TAIL of the kernel: takes the lane's state by value, runs the remaining substeps on this one wavefront
// (solo_control_step), and finishes the step itself (finish_step). The kernels call it and return, so the row-free fast path
// keeps nothing live across a call: with the call in the middle of the kernel the register allocator spilt around it on the
// fast path too (+3 us per launch at 4096 envs, profiles/r03_c_*). The persisted warm start (the state's qacc_warmstart rows)
// is read here: only the robust path touches those rows.
struct SoloTail {
  State st;
  double q0[NJ], q1[NJ], cs[NJ], sn[NJ], t0, t1;
  uint8_t flags;
  bool bad, use_persisted_warm;
  int first_substep;
};
__device__ __noinline__ void solo_tail(KernelParams p, int i, int lane, SoloTail a, double* obs_tile) {
  SoloIn in;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    in.q[j] = a.st.q[j]; in.v[j] = a.st.v[j]; in.q0[j] = a.q0[j]; in.q1[j] = a.q1[j]; in.cs[j] = a.cs[j]; in.sn[j] = a.sn[j];
    in.warm[j] = a.use_persisted_warm ? p.state[(size_t)(S_WARM + j) * p.N + i] : 0.0;
  }
  in.time = a.st.time; in.t0 = a.t0; in.t1 = a.t1;
  in.has_warm = a.use_persisted_warm && (a.flags & FLAG_WARM_VALID);
  in.first_substep = a.first_substep;
  const SoloOut o = solo_control_step(in, Ws{p.ws, p.N, i});
  State st = a.st;
  double cs[NJ], sn[NJ], warm_out[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) { st.q[j] = o.q[j]; st.v[j] = o.v[j]; cs[j] = o.cs[j]; sn[j] = o.sn[j]; warm_out[j] = o.warm[j]; }
  st.time = o.time;
  finish_step(p, i, lane, st, cs, sn, a.flags, warm_out, StepEnd{true, a.bad || o.bad, o.rows_active, o.overflow}, obs_tile);
}
#define MJS_RR_SOLO_TAIL(FIRST_SUBSTEP, USE_PERSISTED_WARM)                                                                   \
  do {                                                                                                                        \
    SoloTail a_;                                                                                                              \
    a_.st = st;                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; j++) { a_.q0[j] = q0[j]; a_.q1[j] = q1[j]; a_.cs[j] = cs[j]; a_.sn[j] = sn[j]; } \
    a_.t0 = t0; a_.t1 = t1; a_.flags = flags; a_.bad = bad; a_.use_persisted_warm = (USE_PERSISTED_WARM);                    \
    a_.first_substep = (FIRST_SUBSTEP);                                                                                       \
    solo_tail(p, i, lane, a_, obs_tile);                                                                                      \
    return;                                                                                                                   \
  } while (0)

// ROLES == 1: one wavefront steps 64 envs. ROLES == 2 (default for stepping): two wavefronts of
// one workgroup, placed on different SIMDs of the CU, step the same 64 envs: role 0 builds M(q),
// factorises the implicitfast matrix (U D U^T) and inverts U while role 1 evaluates the servo
// set-point, actuator forces and bias forces; they exchange 6 doubles per lane through LDS twice
// per substep (qfrc_smooth ->, <- qacc) and integrate redundantly. Measured alternatives (same
// session A/B, profiles/r01_c_ab_roles.txt): one barrier per substep with both roles applying the
// inverse (+2.0 us), contact detection on role 1 (+1.1 us), single wavefront (+5.0 us). Only role 0 touches HBM outputs and the RNG. The per-SIMD FP64 issue
// rate is what bounds this kernel, so splitting the substep across SIMDs is the lever at
// N = 4096 (64 env groups on a 1024-SIMD chip).
template <bool IS_RESET, int ROLES>
__global__ __launch_bounds__(64 * ROLES) void kernel(KernelParams p) {
  const int lane = threadIdx.x & 63;
  const int role = (ROLES == 2) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  const int i = blockIdx.x * 64 + lane;
  __shared__ double xch[1][ROLES == 2 ? 12 : 1][64];  // rows 0-5: qfrc_smooth (role 1 -> 0), 6-11: qacc (role 0 -> 1)
  __shared__ double obs_tile[IS_RESET ? 1 : OBS_DIM * 64];  // wave-private transpose buffer for coalesced obs stores
  if (i >= p.N) return;
  uint8_t flags = p.flags[i];
  if (ROLES == 2) __syncthreads();  // both wavefronts have read flags[i] before role 0 may rewrite it (see kernel3)
  if (IS_RESET || ((flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP)) {
    if (role != 0) return;
    if (IS_RESET && p.reset_mask && !p.reset_mask[i]) return;
    State fresh = episode_init(p.rng, i);
    publish_reset(p, i, fresh, true);
    return;
  }
  MJS_STAMP(p, 0);
  State st = load_state(p, i);
  // before_step: servoL (robot_reach.py:169 -> robot.py:218-259); evaluated by both roles (same result)
  double q0[NJ], q1[NJ], act[3];
#pragma unroll
  for (int k = 0; k < 3; k++) act[k] = p.actions[(size_t)i * ACT_DIM + k];
#pragma unroll
  for (int j = 0; j < NJ; j++) q0[j] = st.q[j];
  if (!tcp_pose_to_joints(act, q0, q1)) {
    flags |= FLAG_IK_FAILED;  // reference raises ValueError; batched: flag + hold position (D-4)
#pragma unroll
    for (int j = 0; j < NJ; j++) q1[j] = q0[j];
  }
  MJS_STAMP(p, 1);
  const double t0 = st.time, t1 = st.time + MJS_RR_CONTROL_DT;
  const double inv_span = 1.0 / (t1 - t0);
  bool bad = false;
  double cs[NJ], sn[NJ];
  load_cs(p, i, S_CS, cs, sn);
  // The guard (see above): both roles evaluate the same predicates on the same data, so the decision is consistent.
  const bool unsafe = joint_near_range(st.q, st.v) || !(flags & FLAG_CLEAR) || !action_in_box(act) || travel_is_long(q0, q1, st.v);
  const bool solo = (ROLES == 1) || __any(unsafe);
  if (solo) {
    if (role != 0) return;
    MJS_RR_SOLO_TAIL(0, true);
  } else if constexpr (ROLES == 2) {
#pragma unroll 1
    for (int s = 0; s < MJS_RR_NSUB; s++) {
      double qacc[NJ];
      if (role == 1) {
        // role 1: servo set-point, actuator forces and bias forces -> qfrc_smooth
        double t = fmin(fmax(st.time, t0), t1);
        double ctrl[NJ], bias[NJ], fact[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
        actuator_forces(st.q, st.v, ctrl, fact);
        ur5e_bias_gen(cs, sn, st.v, bias);
#pragma unroll
        for (int j = 0; j < NJ; j++) xch[0][j][lane] = fact[j] - bias[j];  // qfrc_smooth = -bias + actuator
        __syncthreads();  // qfrc_smooth published
        __syncthreads();  // qacc published
#pragma unroll
        for (int j = 0; j < NJ; j++) qacc[j] = xch[0][6 + j][lane];
      } else {
        // role 0: joint-space inertia, implicitfast matrix, U D U^T and U^-1, all before the barrier
        // (overlapping role 1); the clamp mask is recomputed here (cheap)
        if (s == 10) MJS_STAMP(p, 8);
        double t = fmin(fmax(st.time, t0), t1);
        double ctrl[NJ], fdummy[NJ], M[21], A[NJ][NJ], W[NJ][NJ], Dinv[NJ], rhs[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
        int clamped = actuator_forces(st.q, st.v, ctrl, fdummy);
        ur5e_M_gen(cs, sn, M);
#pragma unroll
        for (int r = 0; r < NJ; r++) {
#pragma unroll
          for (int j = 0; j <= r; j++) A[r][j] = M[r * (r + 1) / 2 + j];
        }
        factor_system(A, clamped, Dinv);
        invert_unit_upper(A, W);
        // opaque register uses pin the whole factorisation before the barrier
#pragma unroll
        for (int r = 0; r < NJ; r++) {
          asm volatile("" : "+v"(Dinv[r]));
#pragma unroll
          for (int j = 0; j < r; j++) asm volatile("" : "+v"(W[r][j]));
        }
        if (s == 10) MJS_STAMP(p, 9);
        __syncthreads();  // qfrc_smooth published
        if (s == 10) MJS_STAMP(p, 10);
#pragma unroll
        for (int j = 0; j < NJ; j++) rhs[j] = xch[0][j][lane];
        apply_inverse(W, Dinv, rhs, qacc);
#pragma unroll
        for (int j = 0; j < NJ; j++) xch[0][6 + j][lane] = qacc[j];
        __syncthreads();  // qacc published
        if (s == 10) MJS_STAMP(p, 11);
      }
      // mj_checkAcc: NaN / inf / |qacc| > 1e10 all make the sum of squares fail this one test
      double acc2 = 0, dq2 = 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        acc2 = fma(qacc[j], qacc[j], acc2);
        st.v[j] += MJS_RR_PHYSICS_DT * qacc[j];
        double dq = MJS_RR_PHYSICS_DT * st.v[j];
        st.q[j] += dq;
        dq2 = fma(dq, dq, dq2);
        rotate_small(cs[j], sn[j], dq);
      }
      bad = bad || !(acc2 <= 1e20);
      if (!(dq2 <= 0.01)) {  // some |dq| may exceed 0.1 rad (runaway state): fall back to the exact functions
#pragma unroll
        for (int j = 0; j < NJ; j++) sincos(st.q[j], &sn[j], &cs[j]);
      }
      st.time += MJS_RR_PHYSICS_DT;
      if (role == 0 && s == 10) MJS_STAMP(p, 12);
    }
  }
  // epilogue on both wavefronts (they hold the same numbers): role 1 writes the 25 state rows the step changed while role 0
  // works out observables, reward, flags and outputs. Not with the same-step auto-reset: role 0 may then rewrite the state rows
  // of a finished env in this launch.
  const bool split_stores = p.autoreset != MJS_AUTORESET_SAME_STEP;
  if (role != 0) {
    if (split_stores) {
      store_state_stepped(p, i, st);
      store_cs(p, i, S_CS, cs, sn);
    }
    return;
  }
  finish_step(p, i, lane, st, cs, sn, flags, nullptr, StepEnd{false, bad, false, false, split_stores}, obs_tile);
}


// ------------------------------------------------------------------------------------------------------------------
// Default stepping kernel: THREE wavefronts per 64 envs, each on its own SIMD of the CU.
//   wave 2      the analytic IK of servoL (the whole closest-of-8 search), published through LDS; then it exits.
//   waves 0, 1  meanwhile run substep 0 on their own, redundantly and without a barrier: its servo set-point is q0
//               itself (fraction 0/20 of the trajectory), so it does not need the IK result. Then one barrier, and
//               substeps 1..19 as the two role-specialised wavefronts of kernel<false, 2>.
// Why not more wavefronts or lanes per env — measured on gfx950 (tools/microbench/ub.hip, profiles/r02_a_microbench.txt):
// one LDS exchange + barrier between two wavefronts costs 150 cycles + ~20 per double, a cross-lane move of a double
// (2 x v_mov_b32_dpp) ~11 cycles = two FP64 issue slots, a wavefront with 16 active lanes issues FP64 no faster than a
// full one, and FP64 MFMA has 53-72 cycles of dependent latency. Splitting the IK by theta1 branch over two wavefronts
// was also measured and is SLOWER (profiles/r02_b_*): the serial search prunes the other 7 candidates against the
// guess-branch result, a wavefront that owns the other branch cannot and evaluates up to 4 full candidates (DESIGN.md
// section 4).
struct IkOut {
  double q[NJ];
  bool found;
};
template <bool INLINE>
__device__ __attribute__((always_inline)) inline IkOut ik_for_wave_body(double ax, double ay, double az, double g0, double g1, double g2, double g3, double g4, double g5) {
  const double act[3] = {ax, ay, az}, g[NJ] = {g0, g1, g2, g3, g4, g5};
  IkOut o;
#pragma unroll
  for (int j = 0; j < NJ; j++) o.q[j] = g[j];
  o.found = tcp_pose_to_joints(act, g, o.q);
  if (!o.found) {
#pragma unroll
    for (int j = 0; j < NJ; j++) o.q[j] = g[j];
  }
  return o;
}
// next-step auto-reset of one lane, out of line: the step kernel's own code stays what its fast path needs
__device__ __noinline__ void reset_lane_next_step(KernelParams p, int i, uint8_t extra_flags) {
  State fresh = episode_init(p.rng, i);
  publish_reset(p, i, fresh, true, extra_flags);
}
// out-of-line copy for the rare solo path (the IK wavefront inlines its own)
__device__ __noinline__ IkOut ik_out_of_line(double ax, double ay, double az, double g0, double g1, double g2, double g3, double g4, double g5) {
  return ik_for_wave_body<false>(ax, ay, az, g0, g1, g2, g3, g4, g5);
}

template <int DUMMY>
__global__ __launch_bounds__(192) void kernel3(KernelParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // The grid is twice the number of 64-env groups: workgroup g < G steps group g, workgroup G + g resets the envs of group g
  // whose episode ended (next-step auto-reset), on another CU and at the same time. An env is handled by exactly one of the
  // two: a stepping lane leaves when its flag byte says "reset pending" (read before the reset workgroup got there) or "reset
  // in THIS launch" (read after: FLAG_FRESH with this launch's parity). With the reset inside the stepping workgroup, episodes
  // that end at different times cost every launch the reset's time and its instruction-cache footprint next to the substep
  // loop (tools/desync_probe.py).
  const int groups = (p.N + 63) >> 6;
  const bool resetter = (int)blockIdx.x >= groups;  // only in launches with p.reset_groups
  const int i = (resetter ? (int)blockIdx.x - groups : (int)blockIdx.x) * 64 + lane;
  __shared__ double xch[12][64];    // rows 0-5: qfrc_smooth (role 1 -> 0), 6-11: qacc (role 0 -> 1)
  __shared__ double ikx[7][64];     // q1[6], found
  __shared__ double obs_tile[OBS_DIM * 64];
  if (i >= p.N) return;
  // Every wavefront takes the same branch below from its OWN read of flags[i]; wave 0 rewrites flags[i] in the epilogue. The
  // barrier orders every wavefront's read before that write.
  uint8_t flags = p.flags[i];
  const bool pending = (flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP;
  if (resetter) {
    if (wave == 0 && pending && pending_is_due(p, flags)) reset_lane_next_step(p, i, (uint8_t)(FLAG_FRESH | (p.epoch ? FLAG_EPOCH : 0)));
    return;
  }
  __syncthreads();
  if (p.reset_groups) {
    if (pending || ((flags & FLAG_FRESH) && ((flags & FLAG_EPOCH) != 0) == (p.epoch != 0))) return;
    flags = (uint8_t)(flags & ~(FLAG_FRESH | FLAG_EPOCH));
  } else if (pending) {  // synchronous episodes (the default): the workgroup resets its own envs
    if (wave == 0) reset_lane_next_step(p, i, 0);
    return;
  }
  MJS_STAMP(p, 0);
  State st = load_state(p, i);
  double q0[NJ], q1[NJ], act[3];
#pragma unroll
  for (int k = 0; k < 3; k++) act[k] = p.actions[(size_t)i * ACT_DIM + k];
#pragma unroll
  for (int j = 0; j < NJ; j++) q0[j] = st.q[j];
  // the part of the guard that is known at kernel entry (same data in every wavefront): a joint within reach of its range,
  // a start configuration that is not known to be clear of the floor, or a servo target outside the task's action box send
  // the whole workgroup to the robust single-wavefront path
  bool solo = __any(joint_near_range(st.q, st.v) || !(flags & FLAG_CLEAR) || !action_in_box(act));
  const double t0 = st.time, t1 = st.time + MJS_RR_CONTROL_DT;
  const double inv_span = 1.0 / (t1 - t0);
  bool bad = false;
  double cs[NJ], sn[NJ];
  if (solo) {
    if (wave != 0) return;
    {
      IkOut ik = ik_out_of_line(act[0], act[1], act[2], q0[0], q0[1], q0[2], q0[3], q0[4], q0[5]);
      if (!ik.found) flags |= FLAG_IK_FAILED;
#pragma unroll
      for (int j = 0; j < NJ; j++) q1[j] = ik.q[j];
    }
    load_cs(p, i, S_CS, cs, sn);
    MJS_RR_SOLO_TAIL(0, true);
  } else {
    if (wave == 2) {
      // before_step: servoL (robot_reach.py:169 -> robot.py:218-259)
      IkOut o = ik_for_wave_body<true>(act[0], act[1], act[2], q0[0], q0[1], q0[2], q0[3], q0[4], q0[5]);
#pragma unroll
      for (int j = 0; j < NJ; j++) ikx[j][lane] = o.q[j];
      ikx[6][lane] = o.found ? 1.0 : 0.0;
      __syncthreads();  // IK published
      return;
    }
    const int role = wave;
    double v_start[NJ];
    load_cs(p, i, S_CS, cs, sn);
#pragma unroll
    for (int j = 0; j < NJ; j++) v_start[j] = st.v[j];
    {
      // substep 0, whole on this wavefront (both of them, same bits): ctrl = q0 (joint_trajectory.py:41-47 at t = t0)
      double M[21], bias[NJ], A[NJ][NJ], rhs[NJ], Dinv[NJ], fact[NJ], ctrl[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) ctrl[j] = q0[j];
      ur5e_M_gen(cs, sn, M);
      ur5e_bias_gen(cs, sn, st.v, bias);
#pragma unroll
      for (int r = 0; r < NJ; r++) {
#pragma unroll
        for (int j = 0; j <= r; j++) A[r][j] = M[r * (r + 1) / 2 + j];
      }
      const int clamped = actuator_forces(st.q, st.v, ctrl, fact);
#pragma unroll
      for (int j = 0; j < NJ; j++) rhs[j] = fact[j] - bias[j];
      factor_system(A, clamped, Dinv);
      udu_solve(A, Dinv, rhs);
      double acc2 = 0, dq2 = 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        acc2 = fma(rhs[j], rhs[j], acc2);
        st.v[j] += MJS_RR_PHYSICS_DT * rhs[j];
        double dq = MJS_RR_PHYSICS_DT * st.v[j];
        st.q[j] += dq;
        dq2 = fma(dq, dq, dq2);
        rotate_small(cs[j], sn[j], dq);
      }
      bad = bad || !(acc2 <= 1e20);
      if (!(dq2 <= 0.01)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) sincos(st.q[j], &sn[j], &cs[j]);
      }
      st.time += MJS_RR_PHYSICS_DT;
    }
    MJS_STAMP(p, 6);
    __syncthreads();  // IK published
    MJS_STAMP(p, 1);
#pragma unroll
    for (int j = 0; j < NJ; j++) q1[j] = ikx[j][lane];
    if (ikx[6][lane] == 0.0) flags |= FLAG_IK_FAILED;  // reference raises ValueError; batched: flag + hold position (D-4)
    // the rest of the guard needs the servo target: a long joint-space travel leaves the fast path here (substep 0 had no
    // rows: the start configuration was clear and no joint near its range). Both dynamics wavefronts hold the same numbers.
    if (__any(travel_is_long(q0, q1, v_start))) {
      if (role != 0) return;
      MJS_RR_SOLO_TAIL(1, false);  // no warm start across the hand-over: the first robust substep has no rows (guard), later ones start from it
    } else {
#pragma unroll 1
      for (int s = 1; s < MJS_RR_NSUB; s++) {
        double qacc[NJ];
        if (role == 1) {
          double t = fmin(fmax(st.time, t0), t1);
          double ctrl[NJ], bias[NJ], fact[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
          actuator_forces(st.q, st.v, ctrl, fact);
          ur5e_bias_gen(cs, sn, st.v, bias);
#pragma unroll
          for (int j = 0; j < NJ; j++) xch[j][lane] = fact[j] - bias[j];  // qfrc_smooth = -bias + actuator
          __syncthreads();  // qfrc_smooth published
          __syncthreads();  // qacc published
#pragma unroll
          for (int j = 0; j < NJ; j++) qacc[j] = xch[6 + j][lane];
        } else {
          if (s == 10) MJS_STAMP(p, 8);
          double t = fmin(fmax(st.time, t0), t1);
          double ctrl[NJ], fdummy[NJ], M[21], A[NJ][NJ], W[NJ][NJ], Dinv[NJ], rhs[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
          int clamped = actuator_forces(st.q, st.v, ctrl, fdummy);
#ifndef MJS_RR_SPLIT_FACTOR
          // M(q) + armature + dt kd -> U D U^T -> U^-1 as ONE generated, list-scheduled block (tools/gen_ur5e_dynamics.py factor_inverse:
          // the pivot chain runs next to the tail of the CRBA by construction; profiles/r04_g_*)
          {
            double dd[NJ], Wp[15];
#pragma unroll
            for (int j = 0; j < NJ; j++) dd[j] = MJS_UR_ARMATURE + (((clamped >> j) & 1) ? 0.0 : MJS_RR_PHYSICS_DT * MJS_UR_ACT_KD[j]);
            ur5e_MW_gen(cs, sn, dd, Wp, Dinv);
#pragma unroll
            for (int r = 1; r < NJ; r++) {
#pragma unroll
              for (int j = 0; j < r; j++) W[r][j] = Wp[r * (r - 1) / 2 + j];
            }
            (void)M; (void)A;
          }
#else
          ur5e_M_gen(cs, sn, M);
#pragma unroll
          for (int r = 0; r < NJ; r++) {
#pragma unroll
            for (int j = 0; j <= r; j++) A[r][j] = M[r * (r + 1) / 2 + j];
          }
          factor_system(A, clamped, Dinv);
          invert_unit_upper(A, W);
#endif
#pragma unroll
          for (int r = 0; r < NJ; r++) {
            asm volatile("" : "+v"(Dinv[r]));
#pragma unroll
            for (int j = 0; j < r; j++) asm volatile("" : "+v"(W[r][j]));
          }
          if (s == 10) MJS_STAMP(p, 9);
          __syncthreads();  // qfrc_smooth published
          if (s == 10) MJS_STAMP(p, 10);
#pragma unroll
          for (int j = 0; j < NJ; j++) rhs[j] = xch[j][lane];
          apply_inverse(W, Dinv, rhs, qacc);
#pragma unroll
          for (int j = 0; j < NJ; j++) xch[6 + j][lane] = qacc[j];
          __syncthreads();  // qacc published
          if (s == 10) MJS_STAMP(p, 11);
        }
        double acc2 = 0, dq2 = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          acc2 = fma(qacc[j], qacc[j], acc2);
          st.v[j] += MJS_RR_PHYSICS_DT * qacc[j];
          double dq = MJS_RR_PHYSICS_DT * st.v[j];
          st.q[j] += dq;
          dq2 = fma(dq, dq, dq2);
          rotate_small(cs[j], sn[j], dq);
        }
        bad = bad || !(acc2 <= 1e20);
        if (!(dq2 <= 0.01)) {
#pragma unroll
          for (int j = 0; j < NJ; j++) sincos(st.q[j], &sn[j], &cs[j]);
        }
        st.time += MJS_RR_PHYSICS_DT;
        if (role == 0 && s == 10) MJS_STAMP(p, 12);
      }
      // epilogue on both wavefronts (same numbers): role 1 writes the 25 state rows the step changed while role 0 works out
      // observables, reward, flags and outputs (not with the same-step auto-reset: role 0 may then rewrite those rows)
      if (role == 1) {
        if (p.autoreset != MJS_AUTORESET_SAME_STEP) {
          store_state_stepped(p, i, st);
          store_cs(p, i, S_CS, cs, sn);
        }
        return;
      }
      finish_step(p, i, lane, st, cs, sn, flags, nullptr, StepEnd{false, bad, false, false, p.autoreset != MJS_AUTORESET_SAME_STEP}, obs_tile);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The Robot entity's control API on a stand-alone UR5e (entities/robots/robot.py:113-272): moveJ / movej_IK / servoL /
// servoJ -> JointTrajectory, then n x (Robot.before_substep; Physics.step), then Robot.get_tcp_pose. This is the
// component the reference's own tests drive (test/test_ur_control_api.py:7-82: UR5e() + raw mjcf.Physics, the XML's
// default timestep, no task), exposed as mjs_ur5e_robot_run so that those tests run literally on the HIP path. One
// robot per lane, same device code as the task kernels (analytic IK, generated M / bias, implicitfast U D U^T solve);
// exact sin/cos every substep (a utility, not a hot path). State block: include/mjsim.h MJS_UR_STATE.
constexpr int UR_STATE = 34;
// SE3Container.orientation_as_quaternion (SE3Container.py:102-106) of a rotation with columns (cx, cy, cz): angle-axis
// (spatialmath tr2angvec, incl. its theta = pi branch) -> [sin(theta/2) axis, cos(theta/2)], scalar LAST
MJS_DEV void rotation_to_quat_xyzw(V3 cx, V3 cy, V3 cz, double* q) {
  const double R[9] = {cx.x, cy.x, cz.x, cx.y, cy.y, cz.y, cx.z, cy.z, cz.z};
  double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  c = fmin(fmax(c, -1.0), 1.0);
  const double theta = acos(c);
  double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (theta < 1e-12 || (n < 1e-12 && c > 0)) { q[0] = q[1] = q[2] = 0; q[3] = 1; return; }
  if (n < 1e-9) {  // theta = pi: R = 2 a a^T - I
    int k = 0;
    if (R[4] > R[0]) k = 1;
    if (R[8] > R[4 * k]) k = 2;
    const double m = sqrt(2.0 * (1.0 + R[4 * k]));
#pragma unroll
    for (int i = 0; i < 3; i++) ax[i] = (R[3 * i + k] + (i == k ? 1.0 : 0.0)) / m;
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) ax[i] /= n;
  }
  const double sh = sin(0.5 * theta);
  q[0] = sh * ax[0]; q[1] = sh * ax[1]; q[2] = sh * ax[2]; q[3] = cos(0.5 * theta);
}

__global__ __launch_bounds__(64) void ur_robot_kernel(double* state, const double* target, int command, double param, int n_substeps, int eef,
                                                      double dt, double* pose_out, uint8_t* status, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double* st = state + (size_t)i * UR_STATE;
  double q[NJ], v[NJ], ctrl[NJ], q0[NJ], q1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) { q[j] = st[j]; v[j] = st[6 + j]; ctrl[j] = st[12 + j]; q0[j] = st[20 + j]; q1[j] = st[26 + j]; }
  double time = st[18], t0 = st[32], t1 = st[33];
  bool active = st[19] != 0.0;
  const double tcp_z = eef == MJS_UR_EEF_GRIPPER ? MJS_G2F85_TCP_Z : 0.0;  // robot.py:104-107: the bare flange has no TCP offset
  int stat = 1;
  if (command != MJS_UR_CMD_NONE) {
    double tgt[NJ];
    bool have = true;
    const double* tg = target + (size_t)i * 7;
    if (command == MJS_UR_CMD_MOVEJ_IK || command == MJS_UR_CMD_SERVOL) {  // robot.py:204-205,219-220
      have = ik_closest(flange_pose_of_tcp(tg, tg + 3, tcp_z), q, tgt);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; j++) tgt[j] = tg[j];
    }
    if (!have) stat = 0;  // movej_IK prints and returns, servoL raises: the trajectory stays as it was
    else {
      double span = param;  // servoJ / servoL: robot.py:254-258
      if (command == MJS_UR_CMD_MOVEJ || command == MJS_UR_CMD_MOVEJ_IK) {  // robot.py:211-216: time = max |dq| / speed
        double mx = 0;
#pragma unroll
        for (int j = 0; j < NJ; j++) mx = fmax(mx, fabs(tgt[j] - q[j]));
        span = mx / param;
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) { q0[j] = q[j]; q1[j] = tgt[j]; }
      t0 = time; t1 = time + span; active = true;
    }
  }
#pragma unroll 1
  for (int s = 0; s < n_substeps; s++) {
    if (active) {  // Robot.before_substep, robot.py:261-263; joint_trajectory.py:33-47
      const double t = fmin(fmax(time, t0), t1);
#pragma unroll
      for (int j = 0; j < NJ; j++) ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) / (t1 - t0);
      if (dt >= t1) active = false;  // robot.py:271: is_finished is handed physics.timestep()
    }
    double cs[NJ], sn[NJ], M[21], bias[NJ], A[NJ][NJ], rhs[NJ], Dinv[NJ], fact[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      sincos(q[j], &sn[j], &cs[j]);
      if (q[j] < MJS_UR_JNT_RANGE[j][0] || q[j] > MJS_UR_JNT_RANGE[j][1]) stat |= 2;  // limit rows are not modelled here
    }
    if (eef == MJS_UR_EEF_GRIPPER) { ur5e_M_gen(cs, sn, M); ur5e_bias_gen(cs, sn, v, bias); }
    else { ur5e_bare_M_gen(cs, sn, M); ur5e_bare_bias_gen(cs, sn, v, bias); }
#pragma unroll
    for (int r = 0; r < NJ; r++) {
#pragma unroll
      for (int j = 0; j <= r; j++) A[r][j] = M[r * (r + 1) / 2 + j];
    }
    // actuator gain of the implicitfast derivative uses THIS dt
    const int clamped = actuator_forces(q, v, ctrl, fact);
#pragma unroll
    for (int j = 0; j < NJ; j++) rhs[j] = fact[j] - bias[j];
    factor_system(A, clamped, Dinv, dt);
    udu_solve(A, Dinv, rhs);
    double acc2 = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      acc2 = fma(rhs[j], rhs[j], acc2);
      v[j] += dt * rhs[j];
      q[j] += dt * v[j];
    }
    if (!(acc2 <= 1e20)) stat |= 4;
    time += dt;
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) { st[j] = q[j]; st[6 + j] = v[j]; st[12 + j] = ctrl[j]; st[20 + j] = q0[j]; st[26 + j] = q1[j]; }
  st[18] = time; st[19] = active ? 1.0 : 0.0; st[32] = t0; st[33] = t1;
  if (pose_out) {  // Robot.get_tcp_pose, robot.py:153-168: flange site pose, TCP offset along the flange z axis
    Chain c;
    fk(q, c);
    // flange frame = wrist_3 * Rx(-90 deg) (MJS_UR_FLANGE_QUAT): x = body x, y = -body z, z = body y
    const V3 fx = c.R[6].cx, fy = -c.R[6].cz, fz = c.R[6].cy;
    const V3 p = madd(madd(c.p[6], MJS_UR_FLANGE_POS[1], c.R[6].cy), tcp_z, fz);
    double* o = pose_out + (size_t)i * 7;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
    rotation_to_quat_xyzw(fx, fy, fz, o + 3);
  }
  if (status) status[i] = (uint8_t)stat;
}

}  // namespace rr
