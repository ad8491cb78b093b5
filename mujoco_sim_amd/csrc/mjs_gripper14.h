// mjs_gripper14.h — Robot Button-Push with the ARTICULATED Robotiq 2F-85: nv = 14 fused control-step kernel.
//
// Path replaced (reference, paths under /root/reference/mujoco_sim/): the same as mjs_button.h —
//   environments/tasks/robot_push_button.py:143-157 before_step (gripper.move + servoJ / servoL), per substep
//   entities/robots/robot.py:261-263 + Physics.step() + entities/props/switch.py:51-72, robot_push_button.py:167-219 reward /
//   termination, :126-134 initialize_episode — on the scene the reference really builds (robot_push_button.py:66-108): the UR5e
//   with entities/eef/gripper.py:36-98's Robotiq2f85 attached, i.e. the menagerie MJCF with its eight hinges, the two `connect`
//   equalities that close the finger linkages, the `joint` equality that couples the drivers (gripper.py:62-66 "others are held
//   by equality constraints"), the fixed tendon with `fingers_actuator` (gripper.py:58-60,79-84), the pad boxes, and
//   <option cone="elliptic" impratio="10"/>. Constants: include/mjs_scene_spec.h MJS_G85_* ([MEN], recalled).
//
// Design (first correct version of the 14-dof tree; the nv = 6 kernels of mjs_button.h stay the fast default):
//   * one env per lane, EPW envs per 64-lane workgroup (the other lanes leave at once: at 4096 envs the chip has 4x more CUs
//     than 64-env wavefronts would use, FP64 issue does not speed up with idle lanes, so fewer envs per wavefront = more CUs busy);
//   * the kinematic tree is compiled into a table of 14 MOVING bodies, one hinge each (body index = dof index): bodies without a
//     joint (arm base, attachment frame, base_mount, base, wrist camera, pads) are welded into their moving ancestor when the
//     model is built (build_model_kernel: composite mass / COM / inertia, mj_setConst's invweight0 / meaninertia), which is the
//     same rigid-body system; gravity compensation and the contact rows' diagApprox keep using the ORIGINAL bodies' own mass / COM;
//   * generic Featherstone passes over that table (kinematics, CRB, RNE), dense 14 x 14 Cholesky, MuJoCo's constraint rows in
//     mj_makeConstraint's order (connect x2, joint coupling, joint limits, elliptic condim-3 contacts), impedance / reference
//     acceleration, the primal Newton solver with elliptic cones (cost, gradient, cone Hessian, exact line search) warm-started
//     like mj_fwdConstraint, the touch sensor, implicitfast integration with the tendon actuator's velocity derivative;
//   * rows live in a per-handle HBM workspace ws[env][row][entry] (env-major: see struct Rows); the per-env state is in LDS;
//   * convex pairs (pad - pad, pad - switch box, button - pad, wrist cylinder - switch box) go through the own MPR of
//     mjs_push_impl.h, one contact per pair (DESIGN.md D-9's rule); pad boxes on the floor: mjc_PlaneBox's corner rule.
// The oracle (oracle/om_engine.c + om_tasks.c build_button(.., OM_GRIPPER_ARTICULATED)) states the same model on MuJoCo's
// own 23-body tree with quaternions; the two agree to rounding (tests/test_gpu_parity.py::test_articulated_*).
#pragma once
#include "mjs_kernel_common.h"
#include "mjs_reach.h"
#include "mjs_push.h"
#include "mjs_button.h"

namespace bg {

#define MJS_HD __host__ __device__ __forceinline__  // shared by the kernels and the host-side model compiler (build_model)
// the loops over the 14 bodies of the tree passes (kinematics, crb, velocity stage): fully unrolled - static LDS offsets and scalar model
// loads that the scheduler can batch (6.9 -> 6.5 ms per launch against the rolled loops, profiles/r04_f_*; -DBG_TREE_LOOP='_Pragma("unroll 1")')
#ifndef BG_TREE_LOOP
#define BG_TREE_LOOP _Pragma("unroll")
#endif
// the register-resident blocks (Hessian + Cholesky + substitutions, M x, factor-solve) stay OUT of line: inlined into their stages the
// launch takes 8.4 instead of 6.7 ms (register allocation over the larger function; profiles/r04_f_*)
#ifndef BG_NEWTON_INLINE
#define BG_NEWTON_INLINE __device__ __noinline__
#endif

constexpr int NV = 14, NA = 6;
constexpr int MAXCON = 16;     // contacts per env (detected); the surplus is dropped and reported (MJS_FAULT_UNSUPPORTED_CONTACT)
constexpr int MAXLIM = 8;      // active joint-limit rows
constexpr int NTRI = NV * (NV + 1) / 2;  // packed lower triangle of a symmetric 14 x 14 matrix: entry (i, j <= i) at i (i + 1) / 2 + j
constexpr int NEQ_ROWS = 7;
constexpr int MAXEFC = NEQ_ROWS + MAXLIM + 3 * MAXCON;
// row workspace: per row 14 Jacobian entries + pos, D, aref, jar, jv, force
constexpr int ROW_J = 0, ROW_POS = NV, ROW_D = NV + 1, ROW_AREF = NV + 2, ROW_JAR = NV + 3, ROW_JV = NV + 4, ROW_FORCE = NV + 5, ROW_STRIDE = NV + 6;
constexpr int WS_DOUBLES = MAXEFC * ROW_STRIDE;  // per env
// state rows: the first 36 keep mjs_button.h's meaning (arm q, v, time, switch position, driver angle + velocity of the RIGHT
// driver for the cameras / the gripper tests, arm qacc_warmstart, cos / sin of the arm joints), then the gripper's 8 q, 8 v,
// 8 qacc_warmstart
constexpr int S_Q = bp::S_Q, S_V = bp::S_V, S_TIME = bp::S_TIME, S_SWITCH = bp::S_SWITCH, S_GRIP = bp::S_GRIP, S_WARM = bp::S_WARM, S_CS = bp::S_CS, S_SN = bp::S_SN;
constexpr int S_GQ = 36, S_GV = 44, S_GWARM = 52, STATE_DIM = 60;
constexpr int OBS_DIM = bp::OBS_DIM;
using bp::FLAG_SWITCH_ACTIVE;
using bp::FLAG_SWITCH_PRESSED;

// ------------------------------------------------------------------------------------------------ the compiled model
// moving bodies: 0-5 the arm links shoulder .. wrist_3; 6-9 right driver, coupler, spring_link, follower; 10-13 the left ones
// parent of a moving body: {-1, 0, 1, 2, 3, 4, 5, 6, 5, 8, 5, 10, 5, 12} (a function: the loops below run with run-time indices)
MJS_HD int PBf(int b) { return (b >= 8 && (b & 1) == 0) ? 5 : b - 1; }
constexpr int B_RDRIVER = 6, B_RCOUPLER = 7, B_RFOLLOWER = 9, B_LDRIVER = 10, B_LCOUPLER = 11, B_LFOLLOWER = 13, B_WRIST3 = 5;
struct Model {
  double pos[NV][3], rot[NV][9];      // body frame in the parent's frame at q = 0 (rot row-major)
  double axis[NV][3], jpos[NV][3];    // hinge axis and anchor in the body frame
  double mass[NV], com[NV][3], inertia[NV][6];  // welded composite: mass, COM in the body frame, inertia about the COM (xx xy xz yy yz zz, body axes)
  double own_mass[NV], own_com[NV][3];          // the original body alone (gravcomp, invweight0)
  double gravcomp[NV], armature[NV], damping[NV], stiffness[NV], springref[NV], range[NV][2], lim_solref[NV], lim_solimp[NV][3];
  // mj_setConst
  double dof_invweight0[NV], meaninertia;
  double invw_body[NV];               // body_invweight0 (translation) of the ORIGINAL moving bodies at their own COM
  double invw_pad[2];                 // of the two pad bodies (right, left)
  double anchor2[2][3];               // connect: the follower's origin expressed in the coupler's frame at qpos0 (right, left)
  // geoms: arm collision proxies (moving body, centre, axis, radius, half length), pads (frame in the follower's frame), flange site
  double col_pos[MJS_UR_NCOLGEOM][3], col_axis[MJS_UR_NCOLGEOM][3], col_xaxis[MJS_UR_NCOLGEOM][3], col_size[MJS_UR_NCOLGEOM][2];
  int col_body[MJS_UR_NCOLGEOM], col_type[MJS_UR_NCOLGEOM];
  double pad_friction[2];
  double pad_pos[2][2][3], pad_rot[2][9];  // [side][box] centre, [side] orientation in the follower's frame
  double pad_com[2][3];                    // pad body's own COM in the follower's frame
  double pad_reach[2];                     // radius about the follower's origin that holds both pad boxes of the side (collision guards)
  double site_pos[3], site_rot[9];         // flange site in wrist_3's frame
};
__constant__ Model g_model;  // constant address space: a wave-uniform index makes the load a scalar one

MJS_HD void quat_to_mat(const double* q_in, double* m) {  // normalises; row-major
  double n = sqrt(q_in[0] * q_in[0] + q_in[1] * q_in[1] + q_in[2] * q_in[2] + q_in[3] * q_in[3]);
  double w = q_in[0] / n, x = q_in[1] / n, y = q_in[2] / n, z = q_in[3] / n;
  m[0] = w * w + x * x - y * y - z * z; m[4] = w * w - x * x + y * y - z * z; m[8] = w * w - x * x - y * y + z * z;
  m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y); m[3] = 2 * (x * y + w * z);
  m[5] = 2 * (y * z - w * x); m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x);
}
MJS_HD void mat_mul(const double* a, const double* b, double* c) {  // 3x3 row-major
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) t[3 * r + k] = a[3 * r] * b[k] + a[3 * r + 1] * b[3 + k] + a[3 * r + 2] * b[6 + k];
  for (int k = 0; k < 9; k++) c[k] = t[k];
}
MJS_HD void mat_vec(const double* a, const double* v, double* r) {
  double x = a[0] * v[0] + a[1] * v[1] + a[2] * v[2], y = a[3] * v[0] + a[4] * v[1] + a[5] * v[2], z = a[6] * v[0] + a[7] * v[1] + a[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MJS_HD void mat_t_vec(const double* a, const double* v, double* r) {
  double x = a[0] * v[0] + a[3] * v[1] + a[6] * v[2], y = a[1] * v[0] + a[4] * v[1] + a[7] * v[2], z = a[2] * v[0] + a[5] * v[1] + a[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MJS_HD void cross3(const double* a, const double* b, double* r) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
MJS_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---- per-env data (lane-private)
struct SIn { double I[6], h[3], mass; };  // spatial inertia about the world origin: I (xx xy xz yy yz zz), h = m c, mass
struct Contact {
  double dist, pos[3], nrm[3];  // (the contact frame is rebuilt from the normal where the rows are made: mju_makeFrame)
  int b1, b2;          // moving bodies (-1: static); the normal points from geom1 to geom2
  double friction;     // tangential friction coefficient of the pair (mj_contactParam)
  double solref, solimp[3], tran;  // time constant; d0, dmax, width; body_invweight0 of both bodies (translation), summed
  bool on_switch;      // one of the geoms belongs to the switch body (touch sensor)
  int row;             // first row (-1: detected but not active: dist >= 0)
  int zone;            // elliptic cone: 0 top, 1 middle, 2 bottom
  double mu;
};
constexpr int TMP_DOUBLES = 382;  // 19 rows of the constraint problem in LDS; the exact size serves the bank rule below
struct Env {
  double q[NV], v[NV], ctrl[7], time, warm[NV];
  double sw[3];
  // position stage (mj_step1)
  double xpos[NV][3], xmat[NV][9], S[NV][6];
  double M[NTRI];           // joint-space inertia (+ armature), packed lower triangle
  double cs[NV], sn[NV];    // cos / sin of the joint angles, carried through the substeps by angle addition (rr::rotate_small; exact at
                            // the start of every control step and whenever a substep turns a joint by more than 0.1 rad)
  double bias[NV], passive[NV];
  Contact con[MAXCON];
  int ncon, nefc, nlim;
  bool overflow;
  uint8_t rtype[MAXEFC];  // 0 equality, 1 limit, 2 contact normal, 3 contact friction
  uint8_t rcon[MAXEFC];   // contact index of a contact row
  // acceleration stage
  double qfrc_smooth[NV], qacc_smooth[NV], qacc[NV], qfrc_constraint[NV], act_force[7], touch;
  int clamped;  // bit u: actuator u sits on its force range
  double sx[NV], sy[NV];  // vector arguments / results of the out-of-line register blocks (pointer arguments would be FLAT + scratch)
  double cost_s;          // cost of the smooth candidate (role 2 -> role 0)
  int r1_bad;   // the integrating wavefront (role 1) saw a non-finite state during this control step
  // the first LROWS rows of the constraint problem (struct Rows below); no stage uses it as scratch any more
  double tmp[TMP_DOUBLES];
#ifdef MJS_BG_PROFILE
  double prof1[8];  // role 1's stage clocks (see the kernel)
  double dbg[6];  // clocks inside st_forces: (spare) | newton_direction | line search | update pass | factor-solve of M | the whole solver
#endif
};
// LDS banks: lane l reads field f of ITS env at l * sizeof(Env) + f, 16 lanes at a time, up to 16 bytes per lane (ds_read_b128 /
// ds_read2_b64). The stride in dwords must spread 16 lanes x 4 dwords over the 64 banks: 34 mod 64 does (l * 34 mod 64 = 0, 34, 4, 38,
// 8, ..: every lane its own group of four banks); 26 mod 64 - the struct without the padding - cost 21 M conflict cycles per launch
// (SQ_LDS_BANK_CONFLICT, profiles/r04_f_*).
#ifndef MJS_BG_PROFILE
static_assert(sizeof(Env) % 8 == 0 && (sizeof(Env) / 4) % 64 == 34, "Env stride vs the LDS banks");
#endif
MJS_HD int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// Row workspace of one env. The first LROWS rows live in LDS (e.tmp); a typical substep has 9 - 12 rows (7 equality rows, the couplers'
// stops, a contact), so the solver's row passes - chains of dependent loads on a wavefront that has its SIMD to itself - stay out of
// HBM. Rows from LROWS on are in the handle's HBM workspace, CONTIGUOUS per env (ws[env][row][entry]): with 16 of 64 lanes carrying an
// env a struct-of-arrays layout buys no coalescing, while env-major makes every entry a constant offset from one row pointer. One
// accessor serves both (the row pointer is a generic one: FLAT loads); RowsLds below is the all-LDS case.
constexpr int LROWS = TMP_DOUBLES / ROW_STRIDE;
struct Rows {
  double* base;  // the env's rows in HBM
  double* lds;   // the env's e.tmp
  MJS_DEV double* row(int r) const { return (r < LROWS ? lds : base) + r * ROW_STRIDE; }
  MJS_DEV double& at(int r, int k) const { return row(r)[k]; }
};
// The same rows when ALL of them are in LDS (nefc <= LROWS: the usual case). The accessor above selects between two address spaces,
// which makes every row access a FLAT instruction - to the LDS aperture a slow one that also ties the vector-memory and the LDS
// counters together (each access waits for the previous); with this type the compiler sees an LDS pointer and emits ds_read / ds_write.
struct RowsLds {
  double* lds;
  MJS_DEV double* row(int r) const { return lds + r * ROW_STRIDE; }
  MJS_DEV double& at(int r, int k) const { return row(r)[k]; }
};

// ------------------------------------------------------------------------------------------------ tree passes
MJS_HD void kinematics(const Model& m, Env& e) {
BG_TREE_LOOP
  for (int b = 0; b < NV; b++) {
    const int p = PBf(b);
    double R0[9], x0[3], tmp[3];
    if (p < 0) {
      for (int k = 0; k < 9; k++) R0[k] = m.rot[b][k];
      for (int k = 0; k < 3; k++) x0[k] = m.pos[b][k];
    } else {
      mat_mul(e.xmat[p], m.rot[b], R0);
      mat_vec(e.xmat[p], m.pos[b], tmp);
      for (int k = 0; k < 3; k++) x0[k] = e.xpos[p][k] + tmp[k];
    }
    double anchor[3], axw[3];
    mat_vec(R0, m.jpos[b], tmp);
    for (int k = 0; k < 3; k++) anchor[k] = x0[k] + tmp[k];
    mat_vec(R0, m.axis[b], axw);
    // R = R0 * Rot(axis, q) with the Rodrigues formula in the body frame
    const double sn = e.sn[b], cs = e.cs[b];
    const double* a = m.axis[b];
    const double oc = 1 - cs;
    double Rl[9] = {cs + oc * a[0] * a[0], oc * a[0] * a[1] - sn * a[2], oc * a[0] * a[2] + sn * a[1],
                    oc * a[0] * a[1] + sn * a[2], cs + oc * a[1] * a[1], oc * a[1] * a[2] - sn * a[0],
                    oc * a[0] * a[2] - sn * a[1], oc * a[1] * a[2] + sn * a[0], cs + oc * a[2] * a[2]};
    mat_mul(R0, Rl, e.xmat[b]);
    mat_vec(e.xmat[b], m.jpos[b], tmp);
    for (int k = 0; k < 3; k++) e.xpos[b][k] = anchor[k] - tmp[k];
    for (int k = 0; k < 3; k++) e.S[b][k] = axw[k];
    cross3(anchor, axw, e.S[b] + 3);
  }
}

// spatial inertia of a moving body (welded composite) about the world origin: I (xx xy xz yy yz zz), h = m c, mass
MJS_HD void body_inertia(const Model& m, const Env& e, int b, SIn& s) {
  const double* R = e.xmat[b];
  double c[3], tmp[3];
  mat_vec(R, m.com[b], tmp);
  for (int k = 0; k < 3; k++) c[k] = e.xpos[b][k] + tmp[k];
  const double* ib = m.inertia[b];
  const double Ib[9] = {ib[0], ib[1], ib[2], ib[1], ib[3], ib[4], ib[2], ib[4], ib[5]};
  double RI[9], Ic[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
  mat_mul(R, Ib, RI);
  mat_mul(RI, Rt, Ic);
  const double mass = m.mass[b], c2 = dot3(c, c);
  s.I[0] = Ic[0] + mass * (c2 - c[0] * c[0]);
  s.I[1] = Ic[1] - mass * c[0] * c[1];
  s.I[2] = Ic[2] - mass * c[0] * c[2];
  s.I[3] = Ic[4] + mass * (c2 - c[1] * c[1]);
  s.I[4] = Ic[5] - mass * c[1] * c[2];
  s.I[5] = Ic[8] + mass * (c2 - c[2] * c[2]);
  for (int k = 0; k < 3; k++) s.h[k] = mass * c[k];
  s.mass = mass;
}
MJS_HD void sin_mul(const SIn& s, const double* v, double* f) {  // f = I v, v = [ang, lin]
  double hv[3], wh[3];
  cross3(s.h, v + 3, hv);
  cross3(v, s.h, wh);
  f[0] = s.I[0] * v[0] + s.I[1] * v[1] + s.I[2] * v[2] + hv[0];
  f[1] = s.I[1] * v[0] + s.I[3] * v[1] + s.I[4] * v[2] + hv[1];
  f[2] = s.I[2] * v[0] + s.I[4] * v[1] + s.I[5] * v[2] + hv[2];
  for (int k = 0; k < 3; k++) f[3 + k] = s.mass * v[3 + k] + wh[k];
}
MJS_DEV void cross_motion6(const double* v, const double* s, double* r) {
  double a[3], b[3], c[3];
  cross3(v, s, a); cross3(v, s + 3, b); cross3(v + 3, s, c);
  for (int k = 0; k < 3; k++) { r[k] = a[k]; r[3 + k] = b[k] + c[k]; }
}
MJS_DEV void cross_force6(const double* v, const double* f, double* r) {
  double a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  for (int k = 0; k < 3; k++) { r[k] = a[k] + b[k]; r[3 + k] = c[k]; }
}

// dense Cholesky A = L L^T on packed lower triangles, false when not positive definite. Host (model compiler): plain loops.
// Device: the 105 entries are loaded into registers, factorised by fully unrolled code (no indexed memory: the loop version's
// dependent LDS reads cost ~450 cycles per multiply-add on a wavefront that has the SIMD to itself, 125 us per factorisation)
// and stored back; out of line, one copy for the five call sites.
inline bool chol_factor_host(const double* A, double* L) {
  bool ok = true;
  for (int i = 0; i < NV; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[tri(i, j)];
      for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) {
        if (!(s >= MJS_MINVAL)) { ok = false; s = MJS_MINVAL; }
        L[tri(i, i)] = sqrt(s);
      } else
        L[tri(i, j)] = s / L[tri(j, j)];
    }
  return ok;
}
inline void chol_solve_host(const double* L, double* x) {
  for (int i = 0; i < NV; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
    x[i] = s / L[tri(i, i)];
  }
  for (int i = NV - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < NV; k++) s -= L[tri(k, i)] * x[k];
    x[i] = s / L[tri(i, i)];
  }
}
extern __shared__ double lds_envs[];
MJS_DEV Env& my_env() { return reinterpret_cast<Env*>(lds_envs)[threadIdx.x & 63]; }  // both wavefronts of a workgroup: lane l <-> env l
#define BG_OFF(field) ((int)(offsetof(Env, field) / sizeof(double)))
// 1 / sqrt(d) for d >= MJS_MINVAL: v_rsq_f64 refined by two Newton steps (y <- y + y (1 - d y^2) / 2; the second in the residual form
// brings the result to within an ulp or two of the correctly rounded value); replaces a sqrt and 14 - j divisions per Cholesky column
MJS_DEV double inv_sqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  double h = 0.5 * y, r = fma(-d * y, y, 1.0);
  y = fma(h, r, y);
  h = 0.5 * y; r = fma(-d * y, y, 1.0);
  return fma(h, r, y);
}
// x <- A^-1 x for a packed symmetric positive definite A: factorisation and both substitutions in registers, nothing stored but x
template <int XO>
BG_NEWTON_INLINE void factor_solve_env() {  // e[XO..] <- M^-1 e[XO..] (fields of the env, addressed by their offset in doubles)
  Env& e = my_env();
  double* const x = reinterpret_cast<double*>(&e) + XO;
  const double* const A = e.M;
  double a[NTRI], y[NV];
#pragma unroll
  for (int k = 0; k < NTRI; k++) a[k] = A[k];
#pragma unroll
  for (int k = 0; k < NV; k++) y[k] = x[k];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double d = a[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= a[tri(j, k)] * a[tri(j, k)];
    if (!(d >= MJS_MINVAL)) d = MJS_MINVAL;
    const double inv = inv_sqrt(d);  // the diagonal holds 1 / l_jj: one reciprocal square root per column instead of 14 - j divisions
    a[tri(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < NV; i++) {
      double sij = a[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) sij -= a[tri(i, k)] * a[tri(j, k)];
      a[tri(i, j)] = sij * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double sy = y[i];
#pragma unroll
    for (int k = 0; k < i; k++) sy -= a[tri(i, k)] * y[k];
    y[i] = sy * a[tri(i, i)];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; i--) {
    double sy = y[i];
#pragma unroll
    for (int k = i + 1; k < NV; k++) sy -= a[tri(k, i)] * y[k];
    y[i] = sy * a[tri(i, i)];
  }
#pragma unroll
  for (int k = 0; k < NV; k++) x[k] = y[k];
}
// y = M x with the packed symmetric M in registers
#ifndef BG_SYMMUL_ATTR
#define BG_SYMMUL_ATTR BG_NEWTON_INLINE
#endif
template <int XO, int YO>
BG_SYMMUL_ATTR void sym_mul_env() {  // e[YO..] = M e[XO..]
  Env& e = my_env();
  const double* const M = e.M;
  const double* const x = reinterpret_cast<double*>(&e) + XO;
  double* const y = reinterpret_cast<double*>(&e) + YO;
  double a[NTRI], xx[NV];
#pragma unroll
  for (int k = 0; k < NTRI; k++) a[k] = M[k];
#pragma unroll
  for (int k = 0; k < NV; k++) xx[k] = x[k];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < NV; k++) s += a[k <= i ? tri(i, k) : tri(k, i)] * xx[k];
    y[i] = s;
  }
}
// mj_crb + armature -> M (packed); the callers factorise it (factor_solve_dev / chol_factor_host). The tree is the arm's chain with four
// two-body branches on wrist_3 (PBf), so the composite inertias are accumulated in registers - branch by branch, then up the chain -
// and nothing but M is written: the pass needs no scratch and can run next to the velocity stage of the other wavefront.
MJS_HD void sin_add(SIn& a, const SIn& b) {
  for (int k = 0; k < 6; k++) a.I[k] += b.I[k];
  for (int k = 0; k < 3; k++) a.h[k] += b.h[k];
  a.mass += b.mass;
}
MJS_HD void crb_row(const Model& m, Env& e, int i, const SIn& c) {  // M[i][j] for j = i and its ancestors
  double f[6];
  sin_mul(c, e.S[i], f);
  for (int j = i; j >= 0; j = PBf(j)) {
    double v = 0;
    for (int k = 0; k < 6; k++) v += e.S[j][k] * f[k];
    e.M[tri(i, j)] = v;
  }
  e.M[tri(i, i)] += m.armature[i];
}
MJS_HD void crb(const Model& m, Env& e) {
BG_TREE_LOOP
  for (int k = 0; k < NTRI; k++) e.M[k] = 0;
  SIn acc;
  for (int k = 0; k < 6; k++) acc.I[k] = 0;
  for (int k = 0; k < 3; k++) acc.h[k] = 0;
  acc.mass = 0;
BG_TREE_LOOP
  for (int br = 0; br < 4; br++) {  // (driver, coupler), (spring_link, follower) of the right finger, then of the left one
    const int pb = NA + 2 * br, cb = pb + 1;
    SIn c, own;
    body_inertia(m, e, cb, c);
    crb_row(m, e, cb, c);
    body_inertia(m, e, pb, own);
    sin_add(c, own);
    crb_row(m, e, pb, c);
    sin_add(acc, c);
  }
BG_TREE_LOOP
  for (int b = NA - 1; b >= 0; b--) {
    SIn own;
    body_inertia(m, e, b, own);
    sin_add(acc, own);
    crb_row(m, e, b, acc);
  }
}

// translational Jacobian of the world point p attached to moving body b: jt[k][j] (zero for dofs that do not move b)
MJS_HD void jac_point(const Env& e, int b, const double* p, double jt[3][NV]) {
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < NV; j++) jt[k][j] = 0;
  for (int j = b; j >= 0; j = PBf(j)) {
    double wxp[3];
    cross3(e.S[j], p, wxp);
    for (int k = 0; k < 3; k++) jt[k][j] = e.S[j][3 + k] + wxp[k];
  }
}

// mj_comVel + mj_rne (flg_acc = 0) + mj_passive (damping, springs, gravity compensation of the arm's own bodies)
MJS_DEV void velocity_stage(const Model& m, Env& e) {
  // spatial velocities, accelerations and forces of the 14 bodies: LOCAL arrays - with the body loops unrolled every index is static and
  // they live in registers (a body's velocity / acceleration only until its children are done, the forces until the backward pass),
  // so the stage needs no scratch in LDS and can run while another wavefront writes the rows into e.tmp
  double cvel[NV][6], cacc[NV][6], cfrc[NV][6];
BG_TREE_LOOP
  for (int b = 0; b < NV; b++) {
    const int p = PBf(b);
    double vp[6] = {0, 0, 0, 0, 0, 0}, ap[6] = {0, 0, 0, 0, 0, -MJS_GRAVITY_Z};
    if (p >= 0)
      for (int k = 0; k < 6; k++) { vp[k] = cvel[p][k]; ap[k] = cacc[p][k]; }
    double sd[6];
    cross_motion6(vp, e.S[b], sd);
    for (int k = 0; k < 6; k++) { cacc[b][k] = ap[k] + sd[k] * e.v[b]; cvel[b][k] = vp[k] + e.S[b][k] * e.v[b]; }
    SIn I;  // the body's own spatial inertia of this configuration (recomputed here rather than shared with crb: the two run on
    body_inertia(m, e, b, I);  // different wavefronts at the same time)
    double Ia[6], Iv[6], vIv[6];
    sin_mul(I, cacc[b], Ia);
    sin_mul(I, cvel[b], Iv);
    cross_force6(cvel[b], Iv, vIv);
    for (int k = 0; k < 6; k++) cfrc[b][k] = Ia[k] + vIv[k];
  }
BG_TREE_LOOP
  for (int b = NV - 1; b > 0; b--) {
    const int p = PBf(b);
    if (p >= 0)
      for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k];
  }
BG_TREE_LOOP
  for (int b = 0; b < NV; b++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += e.S[b][k] * cfrc[b][k];
    e.bias[b] = s;
    e.passive[b] = -m.damping[b] * e.v[b] - (m.stiffness[b] != 0 ? m.stiffness[b] * (e.q[b] - m.springref[b]) : 0.0);
  }
BG_TREE_LOOP
  for (int b = 0; b < NV; b++) {
    if (m.gravcomp[b] == 0) continue;
    double c[3], tmp[3];
    mat_vec(e.xmat[b], m.own_com[b], tmp);
    for (int k = 0; k < 3; k++) c[k] = e.xpos[b][k] + tmp[k];
    const double Fz = -MJS_GRAVITY_Z * m.own_mass[b] * m.gravcomp[b];
    for (int j = b; j >= 0; j = PBf(j)) {
      double wxp[3];
      cross3(e.S[j], c, wxp);
      e.passive[j] += (e.S[j][5] + wxp[2]) * Fz;
    }
  }
}

// ------------------------------------------------------------------------------------------------ collision
#pragma clang fp contract(off)
MJS_DEV void make_frame9(const double* n_in, double* frame) {  // mju_makeFrame
  double n = sqrt(dot3(n_in, n_in));
  for (int k = 0; k < 3; k++) frame[k] = n_in[k] / n;
  double y[3] = {0, 1, 0};
  if (!(frame[1] > -0.5 && frame[1] < 0.5)) { y[1] = 0; y[2] = 1; }
  double dp = dot3(frame, y);
  for (int k = 0; k < 3; k++) y[k] -= dp * frame[k];
  n = sqrt(dot3(y, y));
  for (int k = 0; k < 3; k++) frame[3 + k] = y[k] / n;
  cross3(frame, frame + 3, frame + 6);
}
struct PairParam { double friction, solref, d0, dmax, width; };
MJS_DEV PairParam default_pair() { return PairParam{MJS_GEOM_FRICTION_SLIDE, MJS_SOLREF_TIMECONST, MJS_SOLIMP_D0, MJS_SOLIMP_DWIDTH, MJS_SOLIMP_WIDTH}; }
MJS_DEV PairParam pad_pair(double friction) { return PairParam{friction, MJS_G85_PAD_SOLREF[0], MJS_G85_PAD_SOLIMP[0], MJS_G85_PAD_SOLIMP[1], MJS_G85_PAD_SOLIMP[2]}; }
MJS_DEV void add_contact(Env& e, double dist, const double* pos, const double* nrm, int b1, int b2, PairParam pp_, double tran, bool on_switch) {
  if (e.ncon >= MAXCON) { e.overflow = true; return; }
  Contact& c = e.con[e.ncon++];
  c.dist = dist;
  for (int k = 0; k < 3; k++) { c.pos[k] = pos[k]; c.nrm[k] = nrm[k]; }
  c.b1 = b1; c.b2 = b2;
  c.friction = pp_.friction; c.solref = pp_.solref; c.solimp[0] = pp_.d0; c.solimp[1] = pp_.dmax; c.solimp[2] = pp_.width;
  c.tran = tran; c.on_switch = on_switch;
  c.row = -1; c.zone = 0; c.mu = 0;
}
// geom frames
MJS_DEV void pad_geom(const Model& m, const Env& e, int side, int box, pp::Geom& g, double* centre, double* R) {
  const int b = side == 0 ? B_RFOLLOWER : B_LFOLLOWER;
  double tmp[3];
  mat_vec(e.xmat[b], m.pad_pos[side][box], tmp);
  for (int k = 0; k < 3; k++) centre[k] = e.xpos[b][k] + tmp[k];
  mat_mul(e.xmat[b], m.pad_rot[side], R);
  g.c = v3(centre[0], centre[1], centre[2]);
  g.R = M3{v3(R[0], R[3], R[6]), v3(R[1], R[4], R[7]), v3(R[2], R[5], R[8])};
  g.s = v3(MJS_G85_PAD_SIZE[0], MJS_G85_PAD_SIZE[1], MJS_G85_PAD_SIZE[2]);
  g.box = true; g.cat = -1;
}
MJS_DEV pp::Geom static_box(const double* sw) {
  pp::Geom g;
  g.c = v3(sw[0], sw[1], sw[2] + MJS_SW_BOX_HALF);
  g.R = M3{v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
  g.s = v3(MJS_SW_BOX_HALF, MJS_SW_BOX_HALF, MJS_SW_BOX_HALF);
  g.box = true; g.cat = -1;
  return g;
}
MJS_DEV pp::Geom static_button(const double* sw) {
  pp::Geom g;
  g.c = v3(sw[0], sw[1], sw[2] + MJS_SW_BUTTON_Z);
  g.R = M3{v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
  g.s = v3(MJS_SW_BUTTON_RADIUS, MJS_SW_BUTTON_HALF, 0);
  g.box = false; g.cat = -1;
  return g;
}
// one convex pair through the own MPR (role of mjc_Convex): normal from g1 to g2
MJS_DEV void convex_pair(Env& e, const pp::Geom& g1, const pp::Geom& g2, int b1, int b2, PairParam pp_, double tran, bool on_switch) {
  pp::Contact c;
  if (!pp::collide_convex(g1, g2, 0, 0, 0.0, c)) return;
  const double pos[3] = {c.pos.x, c.pos.y, c.pos.z}, nrm[3] = {c.n.x, c.n.y, c.n.z};
  add_contact(e, c.dist, pos, nrm, b1, b2, pp_, tran, on_switch);
}
// mj_collision of the scene in MuJoCo's pair order (geoms: floor, the arm's ten proxies, right pads 1 2, left pads 1 2, switch box, button)
MJS_DEV void collision(const Model& m, Env& e) {
  e.ncon = 0;
  const double up[3] = {0, 0, 1};
  // floor vs the arm's capsules / cylinder (mjc_PlaneCapsule, mjc_PlaneCylinder). Unrolled over the geoms with the spec's constants
  // (body, type, offset, size; the geom axis is the body's -y or z column: MJS_UR_COL_QUAT is the identity or Rx(90 deg)), so that
  // the ten tests are straight-line code on static LDS offsets instead of a rolled loop of dependent scalar + LDS loads.
#pragma unroll
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    const int b = MJS_UR_COL_BODY[g] - 1;
    const double* R = e.xmat[b];
    double gp[3], axis[3];
    for (int k = 0; k < 3; k++)
      gp[k] = e.xpos[b][k] + (R[3 * k] * MJS_UR_COL_POS[g][0] + R[3 * k + 1] * MJS_UR_COL_POS[g][1] + R[3 * k + 2] * MJS_UR_COL_POS[g][2]);
    const bool rx90 = MJS_UR_COL_QUAT[g][1] != 0.0;
    for (int k = 0; k < 3; k++) axis[k] = rx90 ? -R[3 * k + 1] : R[3 * k + 2];
    const double rad = MJS_UR_COL_SIZE[g][0], half = MJS_UR_COL_SIZE[g][1];
    const double tran = m.invw_body[b];
    if (MJS_UR_COL_TYPE[g] == 3) {
      for (int s = -1; s <= 1; s += 2) {
        double c[3], pos[3];
        for (int k = 0; k < 3; k++) c[k] = gp[k] + s * half * axis[k];
        const double cd = c[2];
        if (cd > rad) continue;
        const double dist = cd - rad;
        for (int k = 0; k < 3; k++) pos[k] = c[k] - up[k] * (rad + 0.5 * dist);
        add_contact(e, dist, pos, up, -1, b, default_pair(), tran, false);
      }
    } else {
      double ax[3] = {axis[0], axis[1], axis[2]};
      const double dist0 = gp[2];
      double prjaxis = ax[2];
      if (prjaxis > 0) { for (int k = 0; k < 3; k++) ax[k] = -ax[k]; prjaxis = -prjaxis; }
      double vec[3], pos[3];
      for (int k = 0; k < 3; k++) vec[k] = ax[k] * prjaxis - up[k];
      const double len = sqrt(dot3(vec, vec));
      if (len < MJS_MINVAL) {  // disk parallel to the plane: the geom's x axis scaled by the radius
        for (int k = 0; k < 3; k++) vec[k] = R[3 * k] * rad;  // (geom x = body x under Rx(90 deg))
      } else {
        for (int k = 0; k < 3; k++) vec[k] *= rad / len;
      }
      const double prjvec = vec[2];
      for (int k = 0; k < 3; k++) ax[k] *= half;
      prjaxis *= half;
      double dd = dist0 + prjaxis + prjvec;
      if (dd > 0) continue;
      for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] + ax[k] - up[k] * dd * 0.5;
      add_contact(e, dd, pos, up, -1, b, default_pair(), tran, false);
      dd = dist0 - prjaxis + prjvec;
      if (dd <= 0) {
        for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] - ax[k] - up[k] * dd * 0.5;
        add_contact(e, dd, pos, up, -1, b, default_pair(), tran, false);
      }
      double side[3];
      cross3(vec, ax, side);
      const double sl = sqrt(dot3(side, side));
      if (sl > MJS_MINVAL) {
        for (int k = 0; k < 3; k++) side[k] *= rad * sqrt(3.0) * 0.5 / sl;
        dd = dist0 + prjaxis - 0.5 * prjvec;
        if (dd <= 0)
          for (int s = 1; s >= -1; s -= 2) {  // point A = +side first, then B = -side
            for (int k = 0; k < 3; k++) pos[k] = gp[k] + s * side[k] + ax[k] - 0.5 * vec[k] - up[k] * dd * 0.5;
            add_contact(e, dd, pos, up, -1, b, default_pair(), tran, false);
          }
      }
    }
  }
  // Guards for the pads' pairs: both pad boxes of a side lie within pad_reach of the follower's origin, so a side whose sphere is
  // clear of the floor / the switch / the other side's sphere cannot produce a contact there (detection has no margin: a contact
  // exists only where the geoms touch), and its geoms are not even built. Pair ORDER is unchanged for the pairs that remain.
  double fo[2][3];
  for (int s = 0; s < 2; s++)
    for (int k = 0; k < 3; k++) fo[s][k] = e.xpos[s == 0 ? B_RFOLLOWER : B_LFOLLOWER][k];
  bool near_floor[2], near_switch[2];
  static_assert(MJS_SW_BOX_HALF == 0.025 && MJS_SW_BUTTON_Z == 0.05 && MJS_SW_BUTTON_HALF == 0.02 && MJS_SW_BUTTON_RADIUS == 0.02, "the switch's bounding sphere below");
  constexpr double SW_REACH = 0.0567;  // the switch box and the button about (sw.x, sw.y, sw.z + 0.035): |(0.025, 0.025, 0.035)| = 0.0495, button top corner 0.0403
  for (int s = 0; s < 2; s++) {
    near_floor[s] = fo[s][2] <= m.pad_reach[s];
    const double dx = fo[s][0] - e.sw[0], dy = fo[s][1] - e.sw[1], dz = fo[s][2] - (e.sw[2] + 0.035), rr_ = m.pad_reach[s] + SW_REACH;
    near_switch[s] = dx * dx + dy * dy + dz * dz <= rr_ * rr_;
  }
  bool near_pads;
  {
    const double dx = fo[0][0] - fo[1][0], dy = fo[0][1] - fo[1][1], dz = fo[0][2] - fo[1][2], rr_ = m.pad_reach[0] + m.pad_reach[1];
    near_pads = dx * dx + dy * dy + dz * dz <= rr_ * rr_;
  }
  const bool need_side[2] = {near_floor[0] || near_switch[0] || near_pads, near_floor[1] || near_switch[1] || near_pads};
  // the pads' geoms
  pp::Geom pad[2][2];
  double pc[2][2][3], pR[2][9];
#pragma unroll 1
  for (int s = 0; s < 2; s++)
    if (need_side[s])
      for (int k = 0; k < 2; k++) pad_geom(m, e, s, k, pad[s][k], pc[s][k], pR[s]);
  // floor vs pad boxes (mjc_PlaneBox: corners at or below the plane, x index fastest, at most 4)
#pragma unroll 1
  for (int s = 0; s < 2; s++) {
    if (!near_floor[s]) continue;
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
      int cnt = 0;
      const int b = s == 0 ? B_RFOLLOWER : B_LFOLLOWER;
      for (int i = 0; i < 8 && cnt < 4; i++) {
        const double loc[3] = {(i & 1) ? MJS_G85_PAD_SIZE[0] : -MJS_G85_PAD_SIZE[0], (i & 2) ? MJS_G85_PAD_SIZE[1] : -MJS_G85_PAD_SIZE[1],
                               (i & 4) ? MJS_G85_PAD_SIZE[2] : -MJS_G85_PAD_SIZE[2]};
        double corner[3], pos[3];
        mat_vec(pR[s], loc, corner);
        for (int j = 0; j < 3; j++) corner[j] += pc[s][k][j];
        const double dist = corner[2];
        if (dist > 0) continue;
        for (int j = 0; j < 3; j++) pos[j] = corner[j] - up[j] * dist * 0.5;
        add_contact(e, dist, pos, up, -1, b, pad_pair(m.pad_friction[k]), m.invw_pad[s], false);
        cnt++;
      }
    }
  }
  // the arm's wrist cylinder (proxy 9) vs the switch box (cylinder - box: the cylinder is geom1)
  const pp::Geom box = static_box(e.sw), button = static_button(e.sw);
  {
    pp::Geom cyl;
    constexpr int g = MJS_UR_NCOLGEOM - 1;
    const int b = B_WRIST3;
    double gp[3], tmp[3];
    mat_vec(e.xmat[b], m.col_pos[g], tmp);
    for (int k = 0; k < 3; k++) gp[k] = e.xpos[b][k] + tmp[k];
    const double dx = gp[0] - box.c.x, dy = gp[1] - box.c.y, dz = gp[2] - box.c.z;
    const double rb = 0.0434 + sqrt(MJS_UR_COL_SIZE[g][0] * MJS_UR_COL_SIZE[g][0] + MJS_UR_COL_SIZE[g][1] * MJS_UR_COL_SIZE[g][1]);  // box: |(0.025)^3| = 0.04331
    if (dx * dx + dy * dy + dz * dz <= rb * rb) {
      double gR[9], lq[9];
      quat_to_mat(MJS_UR_COL_QUAT[g], lq);
      mat_mul(e.xmat[b], lq, gR);
      cyl.c = v3(gp[0], gp[1], gp[2]);
      cyl.R = M3{v3(gR[0], gR[3], gR[6]), v3(gR[1], gR[4], gR[7]), v3(gR[2], gR[5], gR[8])};
      cyl.s = v3(MJS_UR_COL_SIZE[g][0], MJS_UR_COL_SIZE[g][1], 0);
      cyl.box = false; cyl.cat = -1;
      convex_pair(e, cyl, box, b, -1, default_pair(), m.invw_body[b], true);
    }
  }
  // pad pairs in geom order: right pad k vs (left pad 0, left pad 1, switch box, button), then left pad k vs (switch box, button)
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    if (near_pads)
      for (int l = 0; l < 2; l++) convex_pair(e, pad[0][k], pad[1][l], B_RFOLLOWER, B_LFOLLOWER, pad_pair(fmax(m.pad_friction[k], m.pad_friction[l])), m.invw_pad[0] + m.invw_pad[1], false);
    if (near_switch[0]) {
      convex_pair(e, pad[0][k], box, B_RFOLLOWER, -1, pad_pair(m.pad_friction[k]), m.invw_pad[0], true);
      convex_pair(e, button, pad[0][k], -1, B_RFOLLOWER, pad_pair(m.pad_friction[k]), m.invw_pad[0], true);  // cylinder before box (geom type order)
    }
  }
  if (near_switch[1]) {
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
      convex_pair(e, pad[1][k], box, B_LFOLLOWER, -1, pad_pair(m.pad_friction[k]), m.invw_pad[1], true);
      convex_pair(e, button, pad[1][k], -1, B_LFOLLOWER, pad_pair(m.pad_friction[k]), m.invw_pad[1], true);
    }
  }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------ constraint rows
MJS_DEV double impedance(double d0, double dmax, double width, double pos) {  // getimpedance, midpoint 0.5, power 2
  double x = pos / width;
  if (x < 0) x = -x;
  if (x >= 1 || x <= 0) return x >= 1 ? dmax : d0;
  double y;
  if (x <= 0.5) y = 2 * (x * x); else y = 1 - 2 * ((1 - x) * (1 - x));
  return d0 + y * (dmax - d0);
}
// R, K imp, B of a row group from (solref time constant with refsafe, solimp, position for the impedance, diagApprox)
struct KBI { double K, B, imp; };
MJS_DEV KBI kbi(double timeconst, double d0, double dmax, double width, double pos) {
  const double tc = timeconst < 2 * MJS_RR_PHYSICS_DT ? 2 * MJS_RR_PHYSICS_DT : timeconst;  // refsafe
  KBI o;
  o.imp = impedance(d0, dmax, width, pos);
  o.K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc);  // damping ratio 1
  o.B = 2 / fmax(MJS_MINVAL, dmax * tc);
  return o;
}
template <class RW>
MJS_DEV void put_row(const RW& w, int r, const double* J, double pos, double D, double aref) {
  for (int k = 0; k < NV; k++) w.at(r, ROW_J + k) = J[k];
  w.at(r, ROW_POS) = pos; w.at(r, ROW_D) = D; w.at(r, ROW_AREF) = aref;
}
MJS_DEV double row_vel(const Env& e, const double* J) {
  double s = 0;
  for (int k = 0; k < NV; k++) s += J[k] * e.v[k];
  return s;
}
// mj_makeConstraint + mj_makeImpedance + mj_referenceConstraint, in two parts: the equality and joint-limit rows need the positions and
// velocities only; the contact rows need the contacts of the collision stage (which runs on the other wavefront meanwhile)
template <class RW>
MJS_DEV void make_rows_eq(const Model& m, Env& e, const RW& w) {
  int r = 0;
  double J[NV], jt1[3][NV], jt2[3][NV];
  // connect: follower origin (body1, anchor 0 0 0) = the coupler's anchor
  for (int s = 0; s < 2; s++) {
    const int b1 = s == 0 ? B_RFOLLOWER : B_LFOLLOWER, b2 = s == 0 ? B_RCOUPLER : B_LCOUPLER;
    double p1[3], p2[3], tmp[3], res[3];
    for (int k = 0; k < 3; k++) p1[k] = e.xpos[b1][k];
    mat_vec(e.xmat[b2], m.anchor2[s], tmp);
    for (int k = 0; k < 3; k++) { p2[k] = e.xpos[b2][k] + tmp[k]; res[k] = p1[k] - p2[k]; }
    jac_point(e, b1, p1, jt1);
    jac_point(e, b2, p2, jt2);
    const KBI kb = kbi(MJS_G85_SOLREF[0], MJS_G85_SOLIMP[0], MJS_G85_SOLIMP[1], MJS_G85_SOLIMP[2], sqrt(dot3(res, res)));
    const double R = fmax(MJS_MINVAL, (1 - kb.imp) * (m.invw_body[b1] + m.invw_body[b2]) / kb.imp);
    for (int k = 0; k < 3; k++) {
      for (int j = 0; j < NV; j++) J[j] = jt1[k][j] - jt2[k][j];
      put_row(w, r, J, res[k], 1 / R, -kb.B * row_vel(e, J) - kb.K * kb.imp * res[k]);
      e.rtype[r++] = 0;
    }
  }
  {  // joint coupling: right driver - left driver = 0
    for (int j = 0; j < NV; j++) J[j] = 0;
    J[B_RDRIVER] = 1; J[B_LDRIVER] = -1;
    const double res = e.q[B_RDRIVER] - e.q[B_LDRIVER];
    const KBI kb = kbi(MJS_G85_SOLREF[0], MJS_G85_SOLIMP[0], MJS_G85_SOLIMP[1], MJS_G85_SOLIMP[2], res);
    const double R = fmax(MJS_MINVAL, (1 - kb.imp) * (m.dof_invweight0[B_RDRIVER] + m.dof_invweight0[B_LDRIVER]) / kb.imp);
    put_row(w, r, J, res, 1 / R, -kb.B * row_vel(e, J) - kb.K * kb.imp * res);
    e.rtype[r++] = 0;
  }
  // joint limits, lower side first
  e.nlim = 0;
#pragma unroll 1
  for (int j = 0; j < NV; j++)
    for (int side = -1; side <= 1; side += 2) {
      const double dist = side * (m.range[j][(side + 1) / 2] - e.q[j]);
      if (!(dist < 0)) continue;
      if (e.nlim >= MAXLIM) { e.overflow = true; continue; }
      for (int k = 0; k < NV; k++) J[k] = 0;
      J[j] = -side;
      const KBI kb = kbi(m.lim_solref[j], m.lim_solimp[j][0], m.lim_solimp[j][1], m.lim_solimp[j][2], dist);
      const double R = fmax(MJS_MINVAL, (1 - kb.imp) * m.dof_invweight0[j] / kb.imp);
      put_row(w, r, J, dist, 1 / R, -kb.B * (-side * e.v[j]) - kb.K * kb.imp * dist);
      e.rcon[r] = (uint8_t)j;  // the row's dof (the solver's shortcuts for +-e_j rows)
      e.rtype[r++] = 1;
      e.nlim++;
    }
  e.nefc = r;
}
// contacts: elliptic cones, condim 3: the contact frame's rows applied to (jac2 - jac1)
template <class RW>
MJS_DEV void make_rows_contacts(const Model& m, Env& e, const RW& w) {
  int r = e.nefc;
  double J[NV], jt1[3][NV], jt2[3][NV];
#pragma unroll 1
  for (int c = 0; c < e.ncon; c++) {
    Contact& con = e.con[c];
    con.row = -1;
    if (!(con.dist < 0)) continue;  // detected but not active (includemargin 0)
    if (con.b1 >= 0) jac_point(e, con.b1, con.pos, jt1);
    if (con.b2 >= 0) jac_point(e, con.b2, con.pos, jt2);
    const KBI kb = kbi(con.solref, con.solimp[0], con.solimp[1], con.solimp[2], fabs(con.dist));
    const double R0 = fmax(MJS_MINVAL, (1 - kb.imp) * con.tran / kb.imp), R1 = R0 / MJS_G85_IMPRATIO;
    con.mu = con.friction * sqrt(R1 / R0);
    con.row = r;
    double frame[9];
    make_frame9(con.nrm, frame);
    for (int k = 0; k < 3; k++) {
      for (int j = 0; j < NV; j++) {
        double s = 0;
        for (int a = 0; a < 3; a++) s += frame[3 * k + a] * ((con.b2 >= 0 ? jt2[a][j] : 0.0) - (con.b1 >= 0 ? jt1[a][j] : 0.0));
        J[j] = s;
      }
      const double pos = k == 0 ? con.dist : 0.0;
      put_row(w, r, J, pos, 1 / (k == 0 ? R0 : R1), -kb.B * row_vel(e, J) - kb.K * kb.imp * pos);
      e.rtype[r] = k == 0 ? 2 : 3;
      e.rcon[r] = (uint8_t)c;
      r++;
    }
  }
  e.nefc = r;
}

// Diagnostic build only (-DMJS_BG_PROFILE, never shipped; tools/art_profile.py): shader-clock totals per stage, written over the
// observation of the env (kinematics, crb + factor, collision, rows, velocity stage, forces + solver, integration)
#ifdef MJS_BG_PROFILE
#define BG_T(k, stmt) do { const long long t0_ = clock64(); stmt; prof[k] += (double)(clock64() - t0_); } while (0)
#define BG_COUNT(e, k) ((void)0)
#define BG_S(e, k, stmt) do { const long long t0_ = clock64(); stmt; (e).dbg[k] += (double)(clock64() - t0_); } while (0)
#else
#define BG_T(k, stmt) do { stmt; } while (0)
#define BG_COUNT(e, k) ((void)0)
#define BG_S(e, k, stmt) do { stmt; } while (0)
#endif
// The stages are OUT OF LINE, one copy each, and find their env in LDS themselves (an Env& argument would make every access a
// FLAT one): a control step runs them 20 times, resets run them too, and inlined into the four call contexts the kernel was
// 62 k instructions - far beyond the instruction cache, which the substep loop streams through once per substep.
__device__ __noinline__ void st_kinematics() { kinematics(g_model, my_env()); }
__device__ __noinline__ void st_crb() { crb(g_model, my_env()); }
__device__ __noinline__ void st_collision() { collision(g_model, my_env()); }
static_assert(NEQ_ROWS + MAXLIM <= LROWS, "the equality and limit rows are always in LDS");
__device__ __noinline__ void st_rows_eq() { Env& e = my_env(); make_rows_eq(g_model, e, RowsLds{e.tmp}); }
__device__ __noinline__ void st_rows_contacts(double* ws_lane) { Env& e = my_env(); make_rows_contacts(g_model, e, Rows{ws_lane, e.tmp}); }
__device__ __noinline__ void st_velocity() { velocity_stage(g_model, my_env()); }
// mj_step1 on ONE wavefront (resets): position + velocity stages
MJS_DEV void step1(const Model& m, Env& e, const Rows& w) {
  st_kinematics();
  st_crb();
  st_velocity();
  st_collision();
  st_rows_eq();
  st_rows_contacts(w.base);
}

// ------------------------------------------------------------------------------------------------ forces and the solver
MJS_DEV void actuation(const Model& m, Env& e, double* qfrc_act) {
  for (int j = 0; j < NV; j++) qfrc_act[j] = 0;
  e.clamped = 0;
#pragma unroll
  for (int u = 0; u < NA; u++) {
    const double c = clampd(e.ctrl[u], MJS_UR_ACT_CTRLRANGE[u][0], MJS_UR_ACT_CTRLRANGE[u][1]);
    double f = MJS_UR_ACT_KP[u] * c - MJS_UR_ACT_KP[u] * e.q[u] - MJS_UR_ACT_KD[u] * e.v[u];
    const double fc = clampd(f, -MJS_UR_ACT_FRC[u], MJS_UR_ACT_FRC[u]);
    if (fc <= -MJS_UR_ACT_FRC[u] || fc >= MJS_UR_ACT_FRC[u]) e.clamped |= 1 << u;
    e.act_force[u] = fc;
    qfrc_act[u] += fc;
  }
  {  // fingers_actuator on the tendon 0.5 (right driver + left driver)
    const double c = clampd(e.ctrl[6], 0.0, MJS_G2F85_CTRL_MAX);
    const double len = MJS_G85_TENDON_COEF * e.q[B_RDRIVER] + MJS_G85_TENDON_COEF * e.q[B_LDRIVER];
    const double vel = MJS_G85_TENDON_COEF * e.v[B_RDRIVER] + MJS_G85_TENDON_COEF * e.v[B_LDRIVER];
    const double f = MJS_G2F85_ACT_GAIN * c - MJS_G2F85_ACT_KP * len - MJS_G2F85_ACT_KV * vel;
    const double fc = clampd(f, -MJS_G2F85_ACT_FORCE, MJS_G2F85_ACT_FORCE);
    if (fc <= -MJS_G2F85_ACT_FORCE || fc >= MJS_G2F85_ACT_FORCE) e.clamped |= 1 << 6;
    e.act_force[6] = fc;
    qfrc_act[B_RDRIVER] += MJS_G85_TENDON_COEF * fc;
    qfrc_act[B_LDRIVER] += MJS_G85_TENDON_COEF * fc;
  }
}

// ONE pass over the rows (PrimalUpdateConstraint + the gradient): jar (fresh: J qacc - aref; update: += alpha jv), forces, cone zones, the
// constraint cost, and grad = Ma - qfrc_smooth - J^T f; returns the cost incl. the Gauss term. The oracle walks the rows once per
// quantity; fused here because every pass over the HBM row workspace is a chain of dependent loads on a wavefront that has its
// SIMD to itself.
// column pattern of the equality rows (make_rows_eq's order): rows 0-2 connect the right follower (body 9: chain 8, 5..0) to the right
// coupler (7: 6, 5..0), rows 3-5 the left ones (13: 12, 5..0; 11: 10, 5..0), row 6 couples the drivers (6, 10)
MJS_HD constexpr bool eq_row_touches(int r, int k) {
  return r < 3 ? (k < NA + 4) : r < 6 ? (k < NA || k >= NA + 4) : (k == B_RDRIVER || k == B_LDRIVER);
}
// COST_ONLY: nothing is written (no jar / force / zone, no gradient): the pass another wavefront runs on the smooth candidate while the
// solving wavefront evaluates the warm one; same arithmetic for the cost, bit for bit. FRESH: jar = J q - aref with the candidate q_mem
// (a field of the env in LDS: a limit row reads its one component by a run-time index), else jar += alpha jv. The seven equality rows
// go first as straight-line code on their static column pattern (eq_row_touches); a limit row is +-e_j (dof in rcon) and hands its
// J^T f to the gradient through e.sx (no run-time index into a register array).
template <class RW, bool COST_ONLY, bool FRESH>
MJS_DEV double rows_pass(Env& e, const RW& w, const double* q_mem, const double* qacc, const double* Ma, double alpha, double* grad) {
  double cost = 0;
  if (!COST_ONLY) {
    for (int i = 0; i < NV; i++) grad[i] = Ma[i] - e.qfrc_smooth[i];
    for (int i = 0; i < NV; i++) e.sx[i] = 0;
  }
#pragma unroll
  for (int r = 0; r < NEQ_ROWS; r++) {
    double Jr[NV], z;
#pragma unroll
    for (int k = 0; k < NV; k++) Jr[k] = (eq_row_touches(r, k) && (FRESH || !COST_ONLY)) ? w.at(r, ROW_J + k) : 0.0;
    if (FRESH) {
      z = -w.at(r, ROW_AREF);
#pragma unroll
      for (int k = 0; k < NV; k++)
        if (eq_row_touches(r, k)) z += Jr[k] * q_mem[k];
    } else
      z = w.at(r, ROW_JAR) + alpha * w.at(r, ROW_JV);
    const double D = w.at(r, ROW_D), f = -D * z;
    cost += 0.5 * D * z * z;
    if (!COST_ONLY) {
      w.at(r, ROW_JAR) = z;
      w.at(r, ROW_FORCE) = f;
#pragma unroll
      for (int k = 0; k < NV; k++)
        if (eq_row_touches(r, k)) grad[k] -= Jr[k] * f;
    }
  }
#pragma unroll 1
  for (int r = NEQ_ROWS; r < e.nefc; r++) {
    const int t = e.rtype[r];
    if (t == 1) {
      const int j = e.rcon[r];
      const double Jj = w.at(r, ROW_J + j);
      const double z = FRESH ? Jj * q_mem[j] - w.at(r, ROW_AREF) : w.at(r, ROW_JAR) + alpha * w.at(r, ROW_JV);
      const double D = w.at(r, ROW_D);
      const bool act = z < 0;
      const double f = act ? -D * z : 0.0;
      if (act) cost += 0.5 * D * z * z;
      if (!COST_ONLY) {
        w.at(r, ROW_JAR) = z;
        w.at(r, ROW_FORCE) = f;
        if (act) e.sx[j] += Jj * f;
      }
      continue;
    }
    double Jr[3][NV], z[3], f[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) {
      for (int k = 0; k < NV; k++) Jr[a][k] = w.at(r + a, ROW_J + k);
      if (FRESH) {
        double sj = -w.at(r + a, ROW_AREF);
        for (int k = 0; k < NV; k++) sj += Jr[a][k] * q_mem[k];
        z[a] = sj;
      } else
        z[a] = w.at(r + a, ROW_JAR) + alpha * w.at(r + a, ROW_JV);
      if (!COST_ONLY) w.at(r + a, ROW_JAR) = z[a];
    }
    {
      Contact& con = e.con[e.rcon[r]];
      const double D0 = w.at(r, ROW_D), D1 = w.at(r + 1, ROW_D);
      const double mu = con.mu, fr = con.friction;
      const double N = z[0] * mu, U1 = z[1] * fr, U2 = z[2] * fr, T = sqrt(U1 * U1 + U2 * U2);
      if (N >= mu * T) { if (!COST_ONLY) con.zone = 0; }
      else if (mu * N + T <= 0) {
        if (!COST_ONLY) con.zone = 2;
        f[0] = -D0 * z[0]; f[1] = -D1 * z[1]; f[2] = -D1 * z[2];
        cost += 0.5 * (D0 * z[0] * z[0] + D1 * z[1] * z[1] + D1 * z[2] * z[2]);
      } else {
        if (!COST_ONLY) con.zone = 1;
        const double Dm = D0 / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
        cost += 0.5 * Dm * NT * NT;
        f[0] = -Dm * NT * mu;
        f[1] = -f[0] / T * U1 * fr; f[2] = -f[0] / T * U2 * fr;
      }
    }
    if (!COST_ONLY)
      for (int a = 0; a < 3; a++) {
        w.at(r + a, ROW_FORCE) = f[a];
        if (f[a] != 0)
          for (int k = 0; k < NV; k++) grad[k] -= Jr[a][k] * f[a];
      }
    r += 2;
  }
  if (!COST_ONLY)
    for (int i = 0; i < NV; i++) grad[i] -= e.sx[i];
  double gauss = 0;
  for (int i = 0; i < NV; i++) gauss += (Ma[i] - e.qfrc_smooth[i]) * (qacc[i] - e.qacc_smooth[i]);
  return cost + 0.5 * gauss;
}
// exact 1-D minimiser of the cost along the search direction (1-D Newton with bracketing, MuJoCo's gradient stopping rule)
template <class RW>
MJS_DEV double line_search(Env& e, const RW& w, double g1, double g2, double gtol) {
  double alpha = 0, lo = 0, hi = INFINITY;
  // the equality rows are always active: their share of the derivatives is the same quadratic in every iteration, folded into (g1, g2)
#pragma unroll
  for (int r = 0; r < NEQ_ROWS; r++) {
    const double D = w.at(r, ROW_D), jar = w.at(r, ROW_JAR), jv = w.at(r, ROW_JV);
    g1 += D * jar * jv; g2 += D * jv * jv;
  }
#pragma unroll 1
  for (int it = 0; it < 50; it++) {
    BG_COUNT(e, 1);
    double d1 = g1 + alpha * g2, d2 = g2;
#pragma unroll 1
    for (int r = NEQ_ROWS; r < e.nefc; r++) {
      const int t = e.rtype[r];
      if (t == 2) {
        const Contact& con = e.con[e.rcon[r]];
        const double mu = con.mu, fr = con.friction;
        const double j0 = w.at(r, ROW_JAR), j1 = w.at(r + 1, ROW_JAR), j2 = w.at(r + 2, ROW_JAR);
        const double v0 = w.at(r, ROW_JV), v1 = w.at(r + 1, ROW_JV), v2 = w.at(r + 2, ROW_JV);
        const double D0 = w.at(r, ROW_D), D1 = w.at(r + 1, ROW_D);
        const double U0 = j0 * mu, V0 = v0 * mu, u1 = j1 * fr, u2 = j2 * fr, w1 = v1 * fr, w2 = v2 * fr;
        const double UU = u1 * u1 + u2 * u2, UV = u1 * w1 + u2 * w2, VV = w1 * w1 + w2 * w2;
        const double N = U0 + alpha * V0, Tsq = UU + alpha * (2 * UV + alpha * VV), T = Tsq > 0 ? sqrt(Tsq) : 0.0;
        if (N >= mu * T) {
        } else if (mu * N + T <= 0) {
          const double x0 = j0 + alpha * v0, x1 = j1 + alpha * v1, x2 = j2 + alpha * v2;
          d1 += D0 * x0 * v0 + D1 * x1 * v1 + D1 * x2 * v2;
          d2 += D0 * v0 * v0 + D1 * v1 * v1 + D1 * v2 * v2;
        } else {
          const double Dm = D0 / (mu * mu * (1 + mu * mu));
          const double T1 = (UV + alpha * VV) / T, T2 = VV / T - (UV + alpha * VV) * (UV + alpha * VV) / (T * T * T);
          const double NT = N - mu * T, NT1 = V0 - mu * T1;
          d1 += Dm * NT * NT1;
          d2 += Dm * (NT1 * NT1 - NT * mu * T2);
        }
        r += 2;
        continue;
      }
      const double x = w.at(r, ROW_JAR) + alpha * w.at(r, ROW_JV);
      if (t == 0 || x < 0) {
        const double D = w.at(r, ROW_D), jv = w.at(r, ROW_JV);
        d1 += D * x * jv; d2 += D * jv * jv;
      }
    }
    if (fabs(d1) < gtol) break;
    if (d1 < 0) lo = alpha; else hi = alpha;
    if (d2 <= 0) break;
    double next = alpha - d1 / d2;
    if (!(next > lo && next < hi)) next = isfinite(hi) ? 0.5 * (lo + hi) : (alpha > 0 ? 2 * alpha : 1.0);
    if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
    alpha = next;
  }
  return alpha;
}
// Newton direction, out of line and in REGISTERS: H = M + sum over the active rows D J^T J + the cone blocks is accumulated in 105
// registers (rows in a run-time loop, the 14 x 14 triangle unrolled), factorised in place and back-substituted for search = -H^-1 grad
// and J search is written to the rows' JV entries; nothing of H touches memory. Returns false when H is not positive definite.
template <class RW> MJS_DEV RW rows_of(double* ws_env, Env& e);
template <> MJS_DEV Rows rows_of<Rows>(double* ws_env, Env& e) { return Rows{ws_env, e.tmp}; }
template <> MJS_DEV RowsLds rows_of<RowsLds>(double*, Env& e) { return RowsLds{e.tmp}; }
template <class RW>
BG_NEWTON_INLINE bool newton_direction(double* ws_env) {  // gradient in e.sx, search direction out in e.sy
  Env& e = my_env();
  const double* const grad = e.sx;
  double* const search = e.sy;
  const RW w = rows_of<RW>(ws_env, e);
  double h[NTRI];
#pragma unroll
  for (int k = 0; k < NTRI; k++) h[k] = e.M[k];
  // The seven equality rows are always there, always active and in a fixed order with a STATIC column pattern (make_rows_eq: a connect
  // row touches the arm and the two chains of its finger, 10 of 14 dofs; the coupling row the two drivers): unrolled with static LDS
  // offsets, 55 / 3 multiply-adds per row instead of 105. A limit row is +-e_j: its D goes to one diagonal entry, collected per dof
  // in e.sy (free until the search direction is written) because a register array has no run-time index.
#pragma unroll
  for (int r = 0; r < NEQ_ROWS; r++) {
    const double D = w.at(r, ROW_D);
    double Jr[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) Jr[k] = eq_row_touches(r, k) ? w.at(r, ROW_J + k) : 0.0;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      if (!eq_row_touches(r, i)) continue;
      const double dj = D * Jr[i];
#pragma unroll
      for (int j = 0; j <= i; j++)
        if (eq_row_touches(r, j)) h[tri(i, j)] += dj * Jr[j];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; k++) e.sy[k] = 0;
#pragma unroll 1
  for (int r = NEQ_ROWS; r < e.nefc; r++) {
    const int t = e.rtype[r];
    if (t == 1) {  // joint limit: J = +-e_j with j in rcon
      if (w.at(r, ROW_JAR) < 0) e.sy[e.rcon[r]] += w.at(r, ROW_D);
      continue;
    }
    if (t == 2) {
      const Contact& con = e.con[e.rcon[r]];
      if (con.zone != 0) {
        double J0[NV], J1[NV], J2[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) { J0[k] = w.at(r, ROW_J + k); J1[k] = w.at(r + 1, ROW_J + k); J2[k] = w.at(r + 2, ROW_J + k); }
        double Hc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        const double D0 = w.at(r, ROW_D), D1 = w.at(r + 1, ROW_D);
        if (con.zone == 2) { Hc[0][0] = D0; Hc[1][1] = D1; Hc[2][2] = D1; }
        else {
          const double mu = con.mu, fr = con.friction;
          const double N = w.at(r, ROW_JAR) * mu, U1 = w.at(r + 1, ROW_JAR) * fr, U2 = w.at(r + 2, ROW_JAR) * fr, T = sqrt(U1 * U1 + U2 * U2);
          const double Dm = D0 / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
          const double wv[3] = {0, fr * U1, fr * U2}, vv[3] = {mu, -mu * wv[1] / T, -mu * wv[2] / T};
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
              double curv = -wv[a] * wv[b] / (T * T * T);
              if (a == b && a > 0) curv += fr * fr / T;
              Hc[a][b] = Dm * (vv[a] * vv[b] - mu * NT * curv);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const double t0 = Hc[0][0] * J0[i] + Hc[0][1] * J1[i] + Hc[0][2] * J2[i];
          const double t1 = Hc[1][0] * J0[i] + Hc[1][1] * J1[i] + Hc[1][2] * J2[i];
          const double t2 = Hc[2][0] * J0[i] + Hc[2][1] * J1[i] + Hc[2][2] * J2[i];
#pragma unroll
          for (int j = 0; j <= i; j++) h[tri(i, j)] += J0[j] * t0 + J1[j] * t1 + J2[j] * t2;
        }
      }
      r += 2;
      continue;
    }
  }
#pragma unroll
  for (int k = 0; k < NV; k++) h[tri(k, k)] += e.sy[k];
  // Cholesky in place (the diagonal holds 1 / l_jj)
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double d = h[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= h[tri(j, k)] * h[tri(j, k)];
    if (!(d >= MJS_MINVAL)) { ok = false; d = MJS_MINVAL; }
    const double inv = inv_sqrt(d);
    h[tri(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < NV; i++) {
      double sij = h[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) sij -= h[tri(i, k)] * h[tri(j, k)];
      h[tri(i, j)] = sij * inv;
    }
  }
  double y[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) y[i] = -grad[i];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double sy = y[i];
#pragma unroll
    for (int k = 0; k < i; k++) sy -= h[tri(i, k)] * y[k];
    y[i] = sy * h[tri(i, i)];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; i--) {
    double sy = y[i];
#pragma unroll
    for (int k = i + 1; k < NV; k++) sy -= h[tri(k, i)] * y[k];
    y[i] = sy * h[tri(i, i)];
  }
#pragma unroll
  for (int i = 0; i < NV; i++) search[i] = y[i];
  // J search for the line search (equality rows by their pattern; a limit row reads its one component of the search from LDS)
#pragma unroll
  for (int r = 0; r < NEQ_ROWS; r++) {
    double sj = 0;
#pragma unroll
    for (int k = 0; k < NV; k++)
      if (eq_row_touches(r, k)) sj += w.at(r, ROW_J + k) * y[k];
    w.at(r, ROW_JV) = sj;
  }
#pragma unroll 1
  for (int r = NEQ_ROWS; r < e.nefc; r++) {
    double sj = 0;
    if (e.rtype[r] == 1) {
      const int j = e.rcon[r];
      sj = w.at(r, ROW_J + j) * search[j];
    } else {
#pragma unroll
      for (int k = 0; k < NV; k++) sj += w.at(r, ROW_J + k) * y[k];
    }
    w.at(r, ROW_JV) = sj;
  }
  return ok;
}
// mj_fwdConstraint: primal Newton (mj_solPrimal) warm-started from the cheaper of qacc_warmstart and qacc_smooth
// mj_fwdConstraint's solver in two phases, so that the cost of the smooth candidate can come from another wavefront:
//   solve_warm:   the warm candidate's pass (qacc_warmstart; the rows hold its jar / forces afterwards)
//   cost_smooth:  the cost of qacc_smooth (M qacc_smooth = qfrc_smooth by definition: no Gauss term, nothing written)
//   solve_newton: qacc_warmstart unless qacc_smooth is STRICTLY cheaper (then its pass is redone with the writes), the Newton iterations
// PRE: M qacc_warmstart was computed by another wavefront meanwhile (it waits in e.qacc, which the solver only writes at its end).
struct SolveState { double qacc[NV], Ma[NV], grad[NV], cost; };
template <bool PRE, class RW>
MJS_DEV void solve_warm(Env& e, const RW& w, SolveState& st) {
  if (PRE) { for (int i = 0; i < NV; i++) st.Ma[i] = e.qacc[i]; }
  else {
    sym_mul_env<BG_OFF(warm), BG_OFF(sy)>();
    for (int i = 0; i < NV; i++) st.Ma[i] = e.sy[i];
  }
  for (int i = 0; i < NV; i++) st.qacc[i] = e.warm[i];
  st.cost = rows_pass<RW, false, true>(e, w, e.warm, st.qacc, st.Ma, 0.0, st.grad);
}
template <class RW>
MJS_DEV double cost_smooth(Env& e, const RW& w) {
  double ma_s[NV];
  for (int i = 0; i < NV; i++) ma_s[i] = e.qfrc_smooth[i];
  return rows_pass<RW, true, true>(e, w, e.qacc_smooth, e.qacc_smooth, ma_s, 0.0, nullptr);
}
template <class RW>
MJS_DEV void solve_newton(const Model& m, Env& e, const RW& w, double* ws_env, SolveState& st, double c_s) {
  double search[NV], Mv[NV];
  double* const qacc = st.qacc; double* const Ma = st.Ma; double* const grad = st.grad;
  double cost = st.cost;
  if (c_s < cost) {
    for (int i = 0; i < NV; i++) { qacc[i] = e.qacc_smooth[i]; Ma[i] = e.qfrc_smooth[i]; }
    cost = rows_pass<RW, false, true>(e, w, e.qacc_smooth, qacc, Ma, 0.0, grad);
  }
  const double scale = 1 / (m.meaninertia * NV);
#pragma unroll 1
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    bool pd;
    for (int i = 0; i < NV; i++) e.sx[i] = grad[i];
    BG_S(e, 1, pd = newton_direction<RW>(ws_env));
    if (!pd) break;
    sym_mul_env<BG_OFF(sy), BG_OFF(sx)>();
    for (int i = 0; i < NV; i++) { search[i] = e.sy[i]; Mv[i] = e.sx[i]; }
    double g1 = 0, g2 = 0, snorm = 0;
    for (int i = 0; i < NV; i++) { g1 += search[i] * (Ma[i] - e.qfrc_smooth[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
    double alpha;
    BG_S(e, 2, alpha = line_search(e, w, g1, g2, MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale));
    if (alpha == 0) break;
    for (int i = 0; i < NV; i++) { qacc[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
    const double oldcost = cost;
    BG_S(e, 3, (cost = rows_pass<RW, false, false>(e, w, nullptr, qacc, Ma, alpha, grad)));
    double gn = 0;
    for (int i = 0; i < NV; i++) gn += grad[i] * grad[i];
    if (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE) break;
  }
  // qfrc_constraint = J^T f = M a - qfrc_smooth - gradient
  for (int i = 0; i < NV; i++) { e.qacc[i] = qacc[i]; e.qfrc_constraint[i] = Ma[i] - e.qfrc_smooth[i] - grad[i]; }
}
// mj_sensorAcc, touch: normal forces of the contacts of the switch body whose point lies in the site cylinder (button x 1.01)
template <class RW>
MJS_DEV void touch_sensor(Env& e, const RW& w) {
  e.touch = 0;
  for (int c = 0; c < e.ncon; c++) {
    const Contact& con = e.con[c];
    if (con.row < 0 || !con.on_switch) continue;
    const double lx = con.pos[0] - e.sw[0], ly = con.pos[1] - e.sw[1], lz = con.pos[2] - (e.sw[2] + MJS_SW_BUTTON_Z);
    const double rs = MJS_SW_BUTTON_RADIUS * MJS_SW_SITE_SCALE, hs = MJS_SW_BUTTON_HALF * MJS_SW_SITE_SCALE;
    if (lx * lx + ly * ly > rs * rs || fabs(lz) > hs) continue;
    e.touch += w.at(con.row, ROW_FORCE);
  }
}
// mj_fwdActuation + mj_fwdAcceleration + mj_fwdConstraint + mj_sensorAcc on the rows of the last step1
MJS_DEV void forces_smooth(const Model& m, Env& e) {
  double act[NV];
  actuation(m, e, act);
  for (int i = 0; i < NV; i++) { e.qfrc_smooth[i] = e.passive[i] - e.bias[i] + act[i]; e.qacc_smooth[i] = e.qfrc_smooth[i]; }
  BG_S(e, 4, factor_solve_env<BG_OFF(qacc_smooth)>());  // mj_fwdAcceleration (the factor of M is not needed again)
}
// (every env has its seven equality rows: nefc > 0.) SPLIT: the smooth candidate's cost comes from another wavefront (e.cost_s) behind a
// workgroup barrier - ONE barrier for the wavefront whatever accessor its lanes take, hence outside the two instantiations.
template <bool PRE, bool SPLIT>
MJS_DEV void forces_constraint(const Model& m, Env& e, double* ws_env) {
  SolveState st;
  const bool all_lds = e.nefc <= LROWS;  // every row is in LDS
  const RowsLds wl{e.tmp};
  const Rows wm{ws_env, e.tmp};
  if (all_lds) solve_warm<PRE>(e, wl, st); else solve_warm<PRE>(e, wm, st);
  double c_s;
  if (SPLIT) {
    __syncthreads();  // B3b
    c_s = e.cost_s;
  } else
    c_s = all_lds ? cost_smooth(e, wl) : cost_smooth(e, wm);
  if (all_lds) { BG_S(e, 5, solve_newton(m, e, wl, ws_env, st, c_s)); touch_sensor(e, wl); }
  else { BG_S(e, 5, solve_newton(m, e, wm, ws_env, st, c_s)); touch_sensor(e, wm); }
}
// mj_step2 after the forces (warm start, implicitfast velocity update, position integration) on the INTEGRATING wavefront (role 1): the
// matrix M + dt (damping + the unclamped actuators' velocity gains) depends on what is known before the solver starts, so it is
// factorised - in registers - WHILE role 0 solves the constraints; the workgroup barrier in the middle is the solver's completion
// (role 0 executes the matching barrier after st_solve), after which only the two substitutions and the state update remain.
MJS_DEV void integrate_split(const Model& m, Env& e) {
  double a[NTRI], y[NV];
#pragma unroll
  for (int k = 0; k < NTRI; k++) a[k] = e.M[k];
#pragma unroll
  for (int i = 0; i < NV; i++) a[tri(i, i)] += MJS_RR_PHYSICS_DT * m.damping[i];
#pragma unroll
  for (int u = 0; u < NA; u++)
    if (!(e.clamped & (1 << u))) a[tri(u, u)] += MJS_RR_PHYSICS_DT * MJS_UR_ACT_KD[u];
  if (!(e.clamped & (1 << 6))) {
    const double kk = MJS_RR_PHYSICS_DT * MJS_G2F85_ACT_KV * MJS_G85_TENDON_COEF * MJS_G85_TENDON_COEF;
    a[tri(B_RDRIVER, B_RDRIVER)] += kk; a[tri(B_LDRIVER, B_LDRIVER)] += kk; a[tri(B_LDRIVER, B_RDRIVER)] += kk;
  }
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double d = a[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= a[tri(j, k)] * a[tri(j, k)];
    if (!(d >= MJS_MINVAL)) d = MJS_MINVAL;
    const double inv = inv_sqrt(d);
    a[tri(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < NV; i++) {
      double sij = a[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) sij -= a[tri(i, k)] * a[tri(j, k)];
      a[tri(i, j)] = sij * inv;
    }
  }
  __syncthreads();  // the solver is done: qacc, qfrc_constraint
  bool bad = false, far = false;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const double qa = e.qacc[i];
    bad = bad || bad_value(qa);
    e.warm[i] = qa;
    y[i] = e.qfrc_smooth[i] + e.qfrc_constraint[i];
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double sy = y[i];
#pragma unroll
    for (int k = 0; k < i; k++) sy -= a[tri(i, k)] * y[k];
    y[i] = sy * a[tri(i, i)];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; i--) {
    double sy = y[i];
#pragma unroll
    for (int k = i + 1; k < NV; k++) sy -= a[tri(k, i)] * y[k];
    y[i] = sy * a[tri(i, i)];
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const double vi = e.v[i] + MJS_RR_PHYSICS_DT * y[i];
    const double dq = MJS_RR_PHYSICS_DT * vi;
    const double qi = e.q[i] + dq;
    e.v[i] = vi; e.q[i] = qi;
    bad = bad || bad_value(qi) || bad_value(vi);
    double c = e.cs[i], sn_ = e.sn[i];
    rr::rotate_small(c, sn_, dq);  // angle addition; exact again below when some joint of the env turned by more than 0.1 rad
    e.cs[i] = c; e.sn[i] = sn_;
    far = far || !(dq * dq <= 0.01);
  }
  if (far) {
#pragma unroll 1
    for (int i = 0; i < NV; i++) sincos(e.q[i], &e.sn[i], &e.cs[i]);
  }
  e.time += MJS_RR_PHYSICS_DT;
  if (bad) e.r1_bad = 1;
}

__device__ __noinline__ void st_integrate_split() { integrate_split(g_model, my_env()); }
__device__ __noinline__ void st_forces(double* ws_lane) { Env& e = my_env(); forces_smooth(g_model, e); forces_constraint<false, false>(g_model, e, ws_lane); }  // one wavefront (resets)
__device__ __noinline__ void st_smooth() { forces_smooth(g_model, my_env()); }
#ifndef BG_PRE_MW
#define BG_PRE_MW true
#endif
#ifndef MJS_BG_ROLES
#define MJS_BG_ROLES 3
#endif
#ifndef BG_SPLIT_COST
#define BG_SPLIT_COST (MJS_BG_ROLES == 3)
#endif
__device__ __noinline__ void st_solve(double* ws_lane) { Env& e = my_env(); forces_constraint<BG_PRE_MW, BG_SPLIT_COST>(g_model, e, ws_lane); }
MJS_DEV void cost_smooth_stage(Env& e, double* ws_lane) { e.cost_s = e.nefc <= LROWS ? cost_smooth(e, RowsLds{e.tmp}) : cost_smooth(e, Rows{ws_lane, e.tmp}); }
__device__ __noinline__ void st_cost_smooth(double* ws_lane) { cost_smooth_stage(my_env(), ws_lane); }  // role 2, next to role 0's warm candidate
// stage call in the role loops: out of line (one shared copy, callee-saved registers through scratch on every call) or in line
#ifndef BG_STAGE_INLINE
#define BG_STAGE_INLINE 1
#endif
#if BG_STAGE_INLINE
#define BG_ST(call, body) body
#else
#define BG_ST(call, body) call
#endif
__device__ __noinline__ void st_mul_warm() { sym_mul_env<BG_OFF(warm), BG_OFF(qacc)>(); }  // role 1, while role 0 runs st_smooth

// ------------------------------------------------------------------------------------------------ model compilation (host)
// The role of MuJoCo's model compiler + mj_setConst for this scene: runs on the host at mjs_create, the result is copied into
// the device's g_model. (It shares kinematics / crb / Cholesky with the kernels: MJS_HD.)
struct OBody { int parent; double pos[3], quat[4], mass, ipos[3], iquat[4], inertia[3]; int moving; };
inline void build_model(Model& m) {
  // the original tree: 0 UR base, 1-6 UR links, 7 attachment frame, 8-19 gripper bodies, 20 wrist camera
  constexpr int NOB = 21;
  OBody ob[NOB];
  const double ident[4] = {1, 0, 0, 0}, zero3[3] = {0, 0, 0};
  auto set = [&](int o, int parent, const double* pos, const double* quat, double mass, const double* ipos, const double* iquat, const double* inertia, int moving) {
    ob[o].parent = parent; ob[o].mass = mass; ob[o].moving = moving;
    for (int k = 0; k < 3; k++) { ob[o].pos[k] = pos[k]; ob[o].ipos[k] = ipos[k]; ob[o].inertia[k] = inertia[k]; }
    for (int k = 0; k < 4; k++) { ob[o].quat[k] = quat[k]; ob[o].iquat[k] = iquat[k]; }
  };
  for (int b = 0; b < MJS_UR_NBODY; b++)
    set(b, b - 1, MJS_UR_BODY_POS[b], MJS_UR_BODY_QUAT[b], MJS_UR_BODY_MASS[b], MJS_UR_BODY_IPOS[b], MJS_UR_BODY_IQUAT[b], MJS_UR_BODY_DIAGINERTIA[b], b - 1);
  set(7, 6, MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, 0, zero3, ident, zero3, -1);
  const int gmov[MJS_G85_NBODY] = {-1, -1, 6, 7, 8, 9, -1, 10, 11, 12, 13, -1};
  for (int b = 0; b < MJS_G85_NBODY; b++)
    set(8 + b, MJS_G85_PARENT[b] < 0 ? 7 : 8 + MJS_G85_PARENT[b], MJS_G85_POS[b], MJS_G85_QUAT[b], MJS_G85_MASS[b], MJS_G85_IPOS[b], MJS_G85_IQUAT[b], MJS_G85_DIAGINERTIA[b], gmov[b]);
  {  // wrist camera: box + sphere of default density, concentric at MJS_WCAM_POS (mass only)
    const double bx = MJS_CAM_BOX_HALF[0], by = MJS_CAM_BOX_HALF[1], bz = MJS_CAM_BOX_HALF[2], rs = MJS_CAM_SPHERE_RADIUS;
    const double mb = MJS_GEOM_DENSITY * 8 * bx * by * bz, ms = MJS_GEOM_DENSITY * 4.0 / 3.0 * 3.14159265358979323846 * rs * rs * rs, Is = 0.4 * ms * rs * rs;
    const double inertia[3] = {mb * (by * by + bz * bz) / 3 + Is, mb * (bx * bx + bz * bz) / 3 + Is, mb * (bx * bx + by * by) / 3 + Is};
    set(20, 6, MJS_UR_FLANGE_POS, MJS_UR_FLANGE_QUAT, mb + ms, MJS_WCAM_POS, MJS_WCAM_QUAT, inertia, -1);
  }
  // pose of every original body in the frame of its moving ancestor (mov < 0 and never under a joint: the world)
  int mov[NOB];
  double rp[NOB][3], rR[NOB][9];
  for (int o = 0; o < NOB; o++) {
    double R[9];
    quat_to_mat(ob[o].quat, R);
    const int p = ob[o].parent;
    double pp_[3] = {0, 0, 0}, pR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int pm = -1;
    if (p >= 0) { pm = mov[p]; for (int k = 0; k < 3; k++) pp_[k] = rp[p][k]; for (int k = 0; k < 9; k++) pR[k] = rR[p][k]; }
    double cp[3], cR[9], tmp[3];
    mat_vec(pR, ob[o].pos, tmp);
    for (int k = 0; k < 3; k++) cp[k] = pp_[k] + tmp[k];
    mat_mul(pR, R, cR);
    if (ob[o].moving >= 0) {
      const int b = ob[o].moving;
      for (int k = 0; k < 3; k++) m.pos[b][k] = cp[k];
      for (int k = 0; k < 9; k++) m.rot[b][k] = cR[k];
      mov[o] = b;
      for (int k = 0; k < 3; k++) rp[o][k] = 0;
      for (int k = 0; k < 9; k++) rR[o][k] = (k % 4 == 0) ? 1.0 : 0.0;
    } else {
      mov[o] = pm;
      for (int k = 0; k < 3; k++) rp[o][k] = cp[k];
      for (int k = 0; k < 9; k++) rR[o][k] = cR[k];
    }
  }
  // welded composites
  for (int b = 0; b < NV; b++) {
    double mass = 0, mc[3] = {0, 0, 0};
    for (int o = 0; o < NOB; o++) {
      if (mov[o] != b) continue;
      double c[3], tmp[3];
      mat_vec(rR[o], ob[o].ipos, tmp);
      for (int k = 0; k < 3; k++) { c[k] = rp[o][k] + tmp[k]; mc[k] += ob[o].mass * c[k]; }
      mass += ob[o].mass;
    }
    m.mass[b] = mass;
    for (int k = 0; k < 3; k++) m.com[b][k] = mc[k] / mass;
    double I[6] = {0, 0, 0, 0, 0, 0};
    for (int o = 0; o < NOB; o++) {
      if (mov[o] != b) continue;
      double c[3], tmp[3], iR[9], R[9], RI[9], Ic[9];
      mat_vec(rR[o], ob[o].ipos, tmp);
      for (int k = 0; k < 3; k++) c[k] = rp[o][k] + tmp[k] - m.com[b][k];
      quat_to_mat(ob[o].iquat, iR);
      mat_mul(rR[o], iR, R);
      const double D[9] = {ob[o].inertia[0], 0, 0, 0, ob[o].inertia[1], 0, 0, 0, ob[o].inertia[2]};
      const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
      mat_mul(R, D, RI);
      mat_mul(RI, Rt, Ic);
      const double mo = ob[o].mass, c2 = dot3(c, c);
      I[0] += Ic[0] + mo * (c2 - c[0] * c[0]); I[1] += Ic[1] - mo * c[0] * c[1]; I[2] += Ic[2] - mo * c[0] * c[2];
      I[3] += Ic[4] + mo * (c2 - c[1] * c[1]); I[4] += Ic[5] - mo * c[1] * c[2]; I[5] += Ic[8] + mo * (c2 - c[2] * c[2]);
    }
    for (int k = 0; k < 6; k++) m.inertia[b][k] = I[k];
  }
  // joints, own inertials, limits
  for (int b = 0; b < NV; b++) {
    const int o = b < NA ? b + 1 : -1;
    if (b < NA) {
      for (int k = 0; k < 3; k++) { m.axis[b][k] = MJS_UR_JNT_AXIS[b][k]; m.jpos[b][k] = 0; m.own_com[b][k] = ob[o].ipos[k]; }
      m.own_mass[b] = ob[o].mass;
      m.gravcomp[b] = 1.0;  // robot.py:80-82: the arm's bodies, set before the end effector is attached
      m.armature[b] = MJS_UR_ARMATURE; m.damping[b] = 0; m.stiffness[b] = 0; m.springref[b] = 0;
      m.range[b][0] = MJS_UR_JNT_RANGE[b][0]; m.range[b][1] = MJS_UR_JNT_RANGE[b][1];
      m.lim_solref[b] = MJS_SOLREF_TIMECONST; m.lim_solimp[b][0] = MJS_SOLIMP_D0; m.lim_solimp[b][1] = MJS_SOLIMP_DWIDTH; m.lim_solimp[b][2] = MJS_SOLIMP_WIDTH;
    } else {
      int gb = -1;
      for (int k = 0; k < MJS_G85_NBODY; k++) if (gmov[k] == b) gb = k;
      const int c = MJS_G85_JCLASS[gb];
      m.axis[b][0] = 1; m.axis[b][1] = 0; m.axis[b][2] = 0;
      for (int k = 0; k < 3; k++) { m.jpos[b][k] = MJS_G85_JPOS[c][k]; m.own_com[b][k] = MJS_G85_IPOS[gb][k]; }
      m.own_mass[b] = MJS_G85_MASS[gb];
      m.gravcomp[b] = 0;
      m.armature[b] = MJS_G85_JARMATURE[c]; m.damping[b] = MJS_G85_JDAMPING[c]; m.stiffness[b] = MJS_G85_JSTIFFNESS[c]; m.springref[b] = MJS_G85_JSPRINGREF[c];
      m.range[b][0] = MJS_G85_JRANGE[c][0]; m.range[b][1] = MJS_G85_JRANGE[c][1];
      if (MJS_G85_JSTIFFLIMIT[c]) { m.lim_solref[b] = MJS_G85_SOLREF[0]; m.lim_solimp[b][0] = MJS_G85_SOLIMP[0]; m.lim_solimp[b][1] = MJS_G85_SOLIMP[1]; m.lim_solimp[b][2] = MJS_G85_SOLIMP[2]; }
      else { m.lim_solref[b] = MJS_SOLREF_TIMECONST; m.lim_solimp[b][0] = MJS_SOLIMP_D0; m.lim_solimp[b][1] = MJS_SOLIMP_DWIDTH; m.lim_solimp[b][2] = MJS_SOLIMP_WIDTH; }
    }
  }
  // geoms and the flange site
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    double R[9];
    quat_to_mat(MJS_UR_COL_QUAT[g], R);
    for (int k = 0; k < 3; k++) { m.col_pos[g][k] = MJS_UR_COL_POS[g][k]; m.col_axis[g][k] = R[3 * k + 2]; m.col_xaxis[g][k] = R[3 * k]; }
    m.col_size[g][0] = MJS_UR_COL_SIZE[g][0]; m.col_size[g][1] = MJS_UR_COL_SIZE[g][1];
    m.col_body[g] = MJS_UR_COL_BODY[g] - 1; m.col_type[g] = MJS_UR_COL_TYPE[g];
  }
  m.pad_friction[0] = MJS_G85_PAD_FRICTION[0]; m.pad_friction[1] = MJS_G85_PAD_FRICTION[1];
  for (int s = 0; s < 2; s++) {
    const int o = 8 + (s == 0 ? MJS_G85_B_RIGHT_PAD : MJS_G85_B_LEFT_PAD);
    for (int k = 0; k < 9; k++) m.pad_rot[s][k] = rR[o][k];
    double tmp[3];
    for (int bx = 0; bx < 2; bx++) {
      mat_vec(rR[o], MJS_G85_PAD_POS[bx], tmp);
      for (int k = 0; k < 3; k++) m.pad_pos[s][bx][k] = rp[o][k] + tmp[k];
    }
    mat_vec(rR[o], ob[o].ipos, tmp);
    for (int k = 0; k < 3; k++) m.pad_com[s][k] = rp[o][k] + tmp[k];
    m.pad_reach[s] = 0;
    for (int bx = 0; bx < 2; bx++)
      m.pad_reach[s] = fmax(m.pad_reach[s], sqrt(dot3(m.pad_pos[s][bx], m.pad_pos[s][bx])) + sqrt(dot3(MJS_G85_PAD_SIZE, MJS_G85_PAD_SIZE)));
  }
  for (int k = 0; k < 3; k++) m.site_pos[k] = MJS_UR_FLANGE_POS[k];
  quat_to_mat(MJS_UR_FLANGE_QUAT, m.site_rot);
  // mj_setConst at qpos0 = 0: connect anchors, meaninertia, invweight0
  static thread_local Env e;
  static thread_local double L0[NTRI];
  for (int i = 0; i < NV; i++) { e.q[i] = 0; e.v[i] = 0; e.cs[i] = 1; e.sn[i] = 0; }
  kinematics(m, e);
  crb(m, e);
  chol_factor_host(e.M, L0);
  for (int s = 0; s < 2; s++) {
    const int b1 = s == 0 ? B_RFOLLOWER : B_LFOLLOWER, b2 = s == 0 ? B_RCOUPLER : B_LCOUPLER;
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = e.xpos[b1][k] - e.xpos[b2][k];
    mat_t_vec(e.xmat[b2], d, m.anchor2[s]);
  }
  double tr = 0;
  for (int i = 0; i < NV; i++) tr += e.M[tri(i, i)];
  m.meaninertia = tr / NV;
  for (int i = 0; i < NV; i++) {
    double x[NV];
    for (int k = 0; k < NV; k++) x[k] = k == i ? 1.0 : 0.0;
    chol_solve_host(L0, x);
    m.dof_invweight0[i] = x[i];
  }
  auto invweight_tran = [&](int b, const double* local) {
    double c[3], tmp[3], jt[3][NV];
    mat_vec(e.xmat[b], local, tmp);
    for (int k = 0; k < 3; k++) c[k] = e.xpos[b][k] + tmp[k];
    jac_point(e, b, c, jt);
    double s = 0;
    for (int k = 0; k < 3; k++) {
      double x[NV];
      for (int j = 0; j < NV; j++) x[j] = jt[k][j];
      chol_solve_host(L0, x);
      for (int j = 0; j < NV; j++) s += jt[k][j] * x[j];
    }
    return fmax(MJS_MINVAL, s / 3);
  };
  for (int b = 0; b < NV; b++) m.invw_body[b] = invweight_tran(b, m.own_com[b]);
  m.invw_pad[0] = invweight_tran(B_RFOLLOWER, m.pad_com[0]);
  m.invw_pad[1] = invweight_tran(B_LFOLLOWER, m.pad_com[1]);
}

// ------------------------------------------------------------------------------------------------ task glue
MJS_DEV void tcp_of(const Model& m, const Env& e, double* tcp) {  // robot.py:153-168: flange site + its z axis * 0.174
  double tmp[3], R[9];
  mat_vec(e.xmat[B_WRIST3], m.site_pos, tmp);
  mat_mul(e.xmat[B_WRIST3], m.site_rot, R);
  for (int k = 0; k < 3; k++) tcp[k] = e.xpos[B_WRIST3][k] + tmp[k] + R[3 * k + 2] * MJS_G2F85_TCP_Z;
}
MJS_DEV void make_obs(const Model& m, const Env& e, uint8_t flags, double* obs) {
  for (int j = 0; j < NA; j++) obs[j] = e.q[j];
  tcp_of(m, e, obs + 6);
  obs[9] = e.sw[0] + MJS_SW_POSITION_OFFSET;
  obs[10] = e.sw[1] + MJS_SW_POSITION_OFFSET;
  obs[11] = e.sw[2] + MJS_SW_BUTTON_Z + MJS_SW_POSITION_OFFSET;
  obs[12] = (flags & FLAG_SWITCH_ACTIVE) ? 1.0 : 0.0;
}
MJS_DEV void load_env(const KernelParams& p, int i, Env& e) {
  const double* s = p.state + i;
  const size_t N = p.N;
  for (int j = 0; j < NA; j++) { e.q[j] = s[(S_Q + j) * N]; e.v[j] = s[(S_V + j) * N]; e.warm[j] = s[(S_WARM + j) * N]; }
  for (int j = 0; j < NV - NA; j++) { e.q[NA + j] = s[(S_GQ + j) * N]; e.v[NA + j] = s[(S_GV + j) * N]; e.warm[NA + j] = s[(S_GWARM + j) * N]; }
  e.time = s[S_TIME * N];
  for (int k = 0; k < 3; k++) e.sw[k] = s[(S_SWITCH + k) * N];
  for (int j = 0; j < NV; j++) sincos(e.q[j], &e.sn[j], &e.cs[j]);  // exact at the start of every control step
}
MJS_DEV void store_env(const KernelParams& p, int i, const Env& e) {
  double* s = p.state + i;
  const size_t N = p.N;
  for (int j = 0; j < NA; j++) {
    s[(S_Q + j) * N] = e.q[j]; s[(S_V + j) * N] = e.v[j]; s[(S_WARM + j) * N] = e.warm[j];
    double sn, cs;
    sincos(e.q[j], &sn, &cs);
    s[(S_CS + j) * N] = cs; s[(S_SN + j) * N] = sn;
  }
  for (int j = 0; j < NV - NA; j++) { s[(S_GQ + j) * N] = e.q[NA + j]; s[(S_GV + j) * N] = e.v[NA + j]; s[(S_GWARM + j) * N] = e.warm[NA + j]; }
  s[S_TIME * N] = e.time;
  for (int k = 0; k < 3; k++) s[(S_SWITCH + k) * N] = e.sw[k];
  s[S_GRIP * N] = e.q[B_RDRIVER]; s[(S_GRIP + 1) * N] = e.v[B_RDRIVER];
}
// initialize_episode (robot_push_button.py:126-134, switch.py:62-65) + mj_forward; returns the new flag byte
MJS_DEV uint8_t episode_init(const Model& m, const KernelParams& p, int i, Env& e, const Rows& w, uint8_t old_flags) {
  RngCursor c = rng_open(p.rng, i);
  double rp[3], q[NA], zeros[NA] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 3; k++) rp[k] = rng_uniform(p.rng, i, c, MJS_BP_ROBOT_SPACE_LO[k], MJS_BP_ROBOT_SPACE_HI[k]);
  const bool ok = rr::tcp_pose_to_joints(rp, zeros, q);
  for (int k = 0; k < 3; k++) e.sw[k] = rng_uniform(p.rng, i, c, MJS_BP_SWITCH_SPACE_LO[k], MJS_BP_SWITCH_SPACE_HI[k]);
  rng_close(p.rng, i, c);
  for (int j = 0; j < NV; j++) { e.q[j] = 0; e.v[j] = 0; e.warm[j] = 0; }
  for (int j = 0; j < NA; j++) { e.q[j] = ok ? q[j] : 0.0; e.ctrl[j] = e.q[j]; }
  for (int j = 0; j < NV; j++) sincos(e.q[j], &e.sn[j], &e.cs[j]);
  e.ctrl[6] = 0;  // mj_resetData
  e.time = 0;
  e.overflow = false;
  step1(m, e, w);
  st_forces(w.base);
  uint8_t f = old_flags & FLAG_SWITCH_PRESSED;  // was_pressed is stale from the previous episode (switch.py:53)
  bp::switch_update(e.touch, f);
  return f;
}
MJS_DEV void reset_env(const Model& m, const KernelParams& p, int i, Env& e, const Rows& w, uint8_t old_flags, uint8_t extra_flags) {
  const uint8_t f = episode_init(m, p, i, e, w, old_flags);
  store_env(p, i, e);
  p.flags[i] = (uint8_t)(f | extra_flags);
  double obs[OBS_DIM];
  make_obs(m, e, f, obs);
  write_outputs<OBS_DIM>(p, i, obs, 0.0, 1.0, MJS_STEP_FIRST, false, false, false, e.overflow ? MJS_FAULT_UNSUPPORTED_CONTACT : 0, e.ncon);
}

// Envs per workgroup: `epw` lanes of each wavefront carry an env, the others leave at once (FP64 issue does not get faster with idle
// lanes; the LDS holds 16 envs). A STEPPING workgroup has ROLES wavefronts that share the envs in LDS, lane l of each working on env l:
//   role 0: kinematics | crb (M), eq. / limit rows | actuation, M^-1 qfrc_smooth | warm candidate's pass  | Newton iterations, touch, switch |           |
//   role 1:    (waits) | velocity stage            | M qacc_warmstart            |                        | factorises M + dt D              | integrate |
//   role 2:    (waits) | collision                 | contact rows                | smooth candidate's cost|                                  |           |
//                      B1                          B2                            B3                       B3b                                B4          B5
// (ROLES == 2: collision on role 0, all rows on role 1, no B3b.) Six workgroup barriers per Physics.step(); every stage reads what another
// role finished before the last barrier and writes fields no other role touches until the next one. What bounds the launch is the
// latency of role 0's chain (profiles/r04_f_*): the roles exist to take everything off it that does not depend on its last result.
// Resets run all stages on role 0.
extern __shared__ double lds_envs[];
#ifndef MJS_BG_WAVES
#define MJS_BG_WAVES 1  // wavefronts per SIMD the register budget is cut for (tools/ab experiments: -DMJS_BG_WAVES=2 / 4)
#endif
#ifndef MJS_BG_ROLES
#define MJS_BG_ROLES 3
#endif
constexpr int ROLES = MJS_BG_ROLES;
static_assert(ROLES == 2 || ROLES == 3, "two or three wavefronts per env group");
template <bool IS_RESET>
__global__ __launch_bounds__(IS_RESET ? 64 : 64 * ROLES) __attribute__((amdgpu_waves_per_eu(MJS_BG_WAVES, MJS_BG_WAVES))) void kernel(KernelParams p, double* ws_base, int epw) {
  const int lane = threadIdx.x & 63;
  const int role = IS_RESET ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (lane >= epw) return;
  const int i = blockIdx.x * epw + lane;
  if (i >= p.N) return;
  const Model& m = g_model;
  const Rows w{ws_base + (size_t)i * WS_DOUBLES, nullptr};  // (the kernel only hands the HBM pointer to the stages)
  uint8_t flags = p.flags[i];
  if (!IS_RESET) __syncthreads();  // every wavefront has read flags[i] before role 0 may rewrite it
  Env& e = reinterpret_cast<Env*>(lds_envs)[lane];
  if (IS_RESET || ((flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP)) {
    if (role != 0) return;
    if (IS_RESET && p.reset_mask && !p.reset_mask[i]) return;
    e.overflow = false;
    reset_env(m, p, i, e, w, flags, 0);
    return;
  }
#ifdef MJS_BG_PROFILE
  double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // Each role's loop names every stage ONCE (the loop starts with the mj_step1 of the state it finds and ends after the last one), so
  // that BG_STAGE_INLINE can put the stage bodies in line without duplicating them: an out-of-line stage saves and restores the
  // callee-saved registers it uses on every call (scratch traffic: profiles/r04_f_*), an in-line one does not.
  if (role == 1) {  // ---------------------------------------------------------------- role 1
#pragma unroll 1
    for (int s = 0;; s++) {
      __syncthreads();                     // B1: (the env is loaded,) kinematics done
      if (s < MJS_RR_NSUB) {               // (the last mj_step1 only serves the observation and ncon: kinematics + collision)
        BG_T(4, BG_ST(st_velocity(), velocity_stage(g_model, e)));
        if (ROLES == 2) BG_T(3, BG_ST(st_rows_eq(), make_rows_eq(g_model, e, RowsLds{e.tmp})));
      }
      __syncthreads();                     // B2
      if (s == MJS_RR_NSUB) break;
      if (ROLES == 2) BG_T(3, BG_ST(st_rows_contacts(w.base), make_rows_contacts(g_model, e, Rows{w.base, e.tmp})));
      if (BG_PRE_MW) st_mul_warm();
      __syncthreads();                     // B3: the rows are complete
      if (BG_SPLIT_COST) __syncthreads();  // B3b (roles 0 and 2 exchange the smooth candidate's cost)
      BG_T(6, BG_ST(st_integrate_split(), integrate_split(g_model, e)));  // (B4 inside)
#ifdef MJS_BG_PROFILE
      if (s == MJS_RR_NSUB - 1)
        for (int k = 0; k < 8; k++) e.prof1[k] = prof[k];
#endif
      __syncthreads();                     // B5: the new state
    }
    return;
  }
  if (ROLES == 3 && role == 2) {  // --------------------------------------------------- role 2
#pragma unroll 1
    for (int s = 0;; s++) {
      __syncthreads();                     // B1
      BG_ST(st_collision(), collision(g_model, e));
      __syncthreads();                     // B2: the equality / limit rows are made (role 0)
      if (s == MJS_RR_NSUB) break;
      BG_ST(st_rows_contacts(w.base), make_rows_contacts(g_model, e, Rows{w.base, e.tmp}));
      __syncthreads();                     // B3
      if (BG_SPLIT_COST) {
        BG_ST(st_cost_smooth(w.base), cost_smooth_stage(e, w.base));  // qacc_smooth is role 0's, from before B3
        __syncthreads();                   // B3b
      }
      __syncthreads();                     // B4
      __syncthreads();                     // B5
    }
    return;
  }
  // ------------------------------------------------------------------------------------ role 0
  e.overflow = false;
  e.r1_bad = 0;
  load_env(p, i, e);
  // before_step (robot_push_button.py:143-157): gripper.move -> fingers_actuator ctrl, servoJ / servoL -> joint trajectory
  double q0[NA], q1[NA];
  for (int j = 0; j < NA; j++) q0[j] = e.q[j];
  const int A = p.action_type == MJS_ACTION_ABS_EEF ? bp::ACT_DIM_EEF : bp::ACT_DIM_JOINT;
  const double* act = p.actions + (size_t)i * A;
  if (p.action_type == MJS_ACTION_ABS_EEF) {
    const double tp[3] = {act[0], act[1], act[2]};
    e.ctrl[6] = bp::grip_ctrl_of_opening(act[3]);
    if (!rr::tcp_pose_to_joints(tp, q0, q1)) {
      flags |= FLAG_IK_FAILED;
      for (int j = 0; j < NA; j++) q1[j] = q0[j];
    }
  } else {
    for (int j = 0; j < NA; j++) q1[j] = act[j];
    e.ctrl[6] = bp::grip_ctrl_of_opening(act[6]);
  }
  const double t0 = e.time, t1 = e.time + MJS_RR_CONTROL_DT, inv_span = 1.0 / (t1 - t0);
  bool rows_active = false;
#ifdef MJS_BG_PROFILE
  for (int k = 0; k < 6; k++) e.dbg[k] = 0;
#endif
#pragma unroll 1
  for (int s = 0;; s++) {
    BG_T(0, BG_ST(st_kinematics(), kinematics(g_model, e)));  // mj_step1 of the state (first pass: the previous Physics.step()'s; dm_control's legacy order)
    __syncthreads();                       // B1
    if (s < MJS_RR_NSUB) BG_T(1, BG_ST(st_crb(), crb(g_model, e)));
    if (ROLES == 2) BG_T(2, BG_ST(st_collision(), collision(g_model, e)));
    else if (s < MJS_RR_NSUB) BG_T(3, BG_ST(st_rows_eq(), make_rows_eq(g_model, e, RowsLds{e.tmp})));
    __syncthreads();                       // B2
    if (s == MJS_RR_NSUB) break;
    const double t = fmin(fmax(e.time, t0), t1);
    for (int j = 0; j < NA; j++) e.ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;  // robot.py:261-263
    BG_T(5, BG_ST(st_smooth(), forces_smooth(g_model, e)));  // mj_step2: mj_fwdActuation, mj_fwdAcceleration ...
    __syncthreads();                       // B3
    rows_active = rows_active || e.nefc > NEQ_ROWS;
    BG_T(5, BG_ST(st_solve(w.base), (forces_constraint<BG_PRE_MW, BG_SPLIT_COST>(g_model, e, w.base))));  // ... mj_fwdConstraint, mj_sensorAcc (B3b inside)
    bp::switch_update(e.touch, flags);     // Switch.after_substep (switch.py:71-72): the touch force of this Physics.step()
    __syncthreads();                       // B4: role 1 integrates
    __syncthreads();                       // B5
  }
  const bool bad = e.r1_bad != 0;
  if (p.button_disturbances && (flags & FLAG_SWITCH_ACTIVE) && !(flags & FLAG_SWITCH_PRESSED)) flags = bp::disturb(p.rng, i, flags);
  double obs[OBS_DIM];
  make_obs(m, e, flags, obs);
#ifdef MJS_BG_PROFILE
  for (int k = 0; k < 7; k++) obs[k] = prof[k] + e.prof1[k];
  for (int k = 0; k < 6; k++) obs[7 + k] = e.dbg[k];
#endif
  const double dx = obs[6] - MJS_BP_ROBOT_END_POS[0], dy = obs[7] - MJS_BP_ROBOT_END_POS[1], dz = obs[8] - MJS_BP_ROBOT_END_POS[2];
  const bool success = (flags & FLAG_SWITCH_ACTIVE) && sqrt(dx * dx + dy * dy + dz * dz) < MJS_BP_GOAL_THRESHOLD;
  double reward = success ? 1.0 : 0.0, discount = success ? 0.0 : 1.0;
  bool terminate = success;
  if (bad) { reward = 0; discount = 0; terminate = true; }
  if (e.time >= p.time_limit) terminate = true;
  const int fault = (bad ? MJS_FAULT_BAD_STATE : 0) | ((flags & FLAG_IK_FAILED) ? MJS_FAULT_IK_FAILED : 0) | (rows_active ? MJS_FAULT_LIMIT_COLDSTART : 0) |
                    (e.overflow ? MJS_FAULT_UNSUPPORTED_CONTACT : 0);
  const bool terminated = terminate && discount == 0.0, truncated = terminate && discount > 0.0;
  const uint8_t newflags = (uint8_t)((flags & (FLAG_IK_FAILED | FLAG_SWITCH_ACTIVE | FLAG_SWITCH_PRESSED)) | (terminate ? FLAG_RESET_PENDING : 0) | FLAG_WARM_VALID);
  store_env(p, i, e);
  p.flags[i] = newflags;
  write_outputs<OBS_DIM>(p, i, obs, reward, discount, terminate ? MJS_STEP_LAST : MJS_STEP_MID, terminated, truncated, success, fault, e.ncon);
  if (terminate && p.autoreset == MJS_AUTORESET_SAME_STEP) {
    if (p.out.terminal_obs)
      for (int k = 0; k < OBS_DIM; k++) p.out.terminal_obs[(size_t)i * OBS_DIM + k] = obs[k];
    e.overflow = false;
    const uint8_t f = episode_init(m, p, i, e, w, newflags);
    store_env(p, i, e);
    p.flags[i] = f;
    make_obs(m, e, f, obs);
    if (p.out.obs)
      for (int k = 0; k < OBS_DIM; k++) p.out.obs[(size_t)i * OBS_DIM + k] = obs[k];
    if (p.out.ncon) p.out.ncon[i] = e.ncon;
  }
}

}  // namespace bg
