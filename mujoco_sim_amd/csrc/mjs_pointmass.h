// mjs_pointmass.h — Pointmass-Reach fused control-step kernel (BASELINE configs 1-2).
//
// One env per wavefront lane; the 13-double state is struct-of-arrays in HBM and is read and
// written exactly once per control step; the 5 physics substeps run in registers.
// Path replaced (reference, paths under /root/reference/mujoco_sim/):
//   environments/tasks/point_reach.py:150-163 before_step (mocap target = clip(xpos + action))
//   5 x Physics.step() for the scene of point_reach.py:75-102 + entities/pointmass.py:52-66 +
//       mjcf/walled_pointmass_arena.xml:15-19: nv = 2 slide dofs, weld-to-mocap equality
//       (6 rows, 2 with non-zero Jacobian), sphere-vs-4-wall pyramidal contacts, sphere-vs-ground
//       detection, primal Newton solver, semi-implicit Euler
//   point_reach.py:165-202 after_step / get_reward / should_terminate_episode / get_discount
//   point_reach.py:125-144 + pointmass.py:100-130 initialize_episode (4 uniforms)
//   dm_control composer loop + environments/dmc2gym.py:144-153 termination / truncation split
// HBM-bound in principle (260 B per env-step); at N=4096 the state fits in L2.
#pragma once
#include "mjs_kernel_common.h"

namespace pm {

constexpr int S_QX = 0, S_QY = 1, S_VX = 2, S_VY = 3, S_MX = 4, S_MY = 5, S_TX = 6, S_TY = 7, S_TIME = 8, S_DIST = 9, S_PREV = 10,
              S_WX = 11, S_WY = 12, STATE_DIM = 13;
constexpr int OBS_DIM = 4, ACT_DIM = 2;

struct State {
  double qx, qy, vx, vy, mx, my, tx, ty, time, dist, prev, wx, wy;
};

__device__ __forceinline__ State load_state(const KernelParams& p, int i) {
  const double* s = p.state + i;
  const size_t N = p.N;
  return State{s[S_QX * N], s[S_QY * N], s[S_VX * N], s[S_VY * N], s[S_MX * N], s[S_MY * N], s[S_TX * N],
               s[S_TY * N], s[S_TIME * N], s[S_DIST * N], s[S_PREV * N], s[S_WX * N], s[S_WY * N]};
}
__device__ __forceinline__ void store_state(const KernelParams& p, int i, const State& st) {
  double* s = p.state + i;
  const size_t N = p.N;
  s[S_QX * N] = st.qx; s[S_QY * N] = st.qy; s[S_VX * N] = st.vx; s[S_VY * N] = st.vy;
  s[S_MX * N] = st.mx; s[S_MY * N] = st.my; s[S_TX * N] = st.tx; s[S_TY * N] = st.ty;
  s[S_TIME * N] = st.time; s[S_DIST * N] = st.dist; s[S_PREV * N] = st.prev; s[S_WX * N] = st.wx; s[S_WY * N] = st.wy;
}

// Constraint rows of one substep (nv = 2) live in 10 STATIC slots so that every index is a compile-time constant
// (registers, no scratch): slots 0-1 the two weld rows with non-zero Jacobian, then two groups of 4 pyramid edges for
// the (at most two: the arena is 1 m wide) walls THIS LANE touches, in geom order (wall_x, wall_y, wall_neg_x,
// wall_neg_y of walled_pointmass_arena.xml:16-19) = MuJoCo's row order, which only instantiates rows of active
// contacts. Row Jacobians are J = n + mu*(+t1, -t1, +t2, -t2) with the tangents mju_makeFrame derives from each
// normal (one tangent is always the world z axis, along which the pointmass has no dof): per-wall constants,
// selected per lane. (The first version kept 4 x 4 wall slots with compile-time Jacobians and took that 18-slot path
// whenever ANY lane of the wavefront touched a wall: 7.6 % of the envs do, i.e. 98 % of the 64-env wavefronts.)
constexpr int NSLOT = 10;
constexpr double MU = MJS_GEOM_FRICTION_SLIDE;
constexpr double WALL_JX[4][4] = {{1, 1, 1, 1}, {0, 0, MU, -MU}, {-1, -1, -1, -1}, {0, 0, -MU, MU}};
constexpr double WALL_JY[4][4] = {{MU, -MU, 0, 0}, {1, 1, 1, 1}, {MU, -MU, 0, 0}, {-1, -1, -1, -1}};

struct Rows {
  double D[NSLOT], aref[NSLOT], JX[NSLOT], JY[NSLOT];
  bool on[NSLOT];
};

template <int BASE>
__device__ __forceinline__ void set_contact(Rows& r, int wall, double dist, double vx, double vy, double K, double B, double tran) {
  // pyramidal condim-3 contact (mj_instantiateContact / mj_diagApprox / mj_makeImpedance) with wall `wall` (< 0: none)
  bool act = wall >= 0;
  double imp = impedance_default(act ? dist : -1.0e-3);
  double dA = tran + MU * MU * tran;
  double R0 = fmax(MJS_MINVAL, (1 - imp) * dA / imp);
  double Rpy = 2 * MU * MU * R0;
  double D = 1 / Rpy;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double jx = wall == 0 ? WALL_JX[0][k] : wall == 1 ? WALL_JX[1][k] : wall == 2 ? WALL_JX[2][k] : WALL_JX[3][k];
    const double jy = wall == 0 ? WALL_JY[0][k] : wall == 1 ? WALL_JY[1][k] : wall == 2 ? WALL_JY[2][k] : WALL_JY[3][k];
    r.JX[BASE + k] = jx; r.JY[BASE + k] = jy;
    double vel = jx * vx + jy * vy;
    r.D[BASE + k] = D;
    r.on[BASE + k] = act;
    r.aref[BASE + k] = -B * vel - K * imp * dist;
  }
}

// value of the constraint+Gauss cost, forces and active set (mj_constraintUpdate), nv = 2,
// M = m*I, qfrc_smooth = qacc_smooth = 0
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
template <int NS>
__device__ __forceinline__ double cost_update(const Rows& r, const double* jar, double ax, double ay, double Max, double May, bool* active, double* force) {
  double cost = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    bool act = r.on[k] && (k < 2 || jar[k] < 0);
    active[k] = act;
    force[k] = act ? -r.D[k] * jar[k] : 0.0;
    if (act) cost += 0.5 * r.D[k] * jar[k] * jar[k];
  }
  double gauss = Max * ax + May * ay;
  return cost + 0.5 * gauss;
}

// 1-D Newton with bracketing on the piecewise-quadratic cost (role of MuJoCo's PrimalSearch)
template <int NS>
__device__ __forceinline__ double line_search(const Rows& r, const double* jar, const double* jv, double g1, double g2, double gtol) {
  double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
  for (int it = 0; it < 50; it++) {
    double d1 = g1 + alpha * g2, d2 = g2;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      double x = jar[k] + alpha * jv[k];
      if (r.on[k] && (k < 2 || x < 0)) {
        d1 += r.D[k] * x * jv[k];
        d2 += r.D[k] * jv[k] * jv[k];
      }
    }
    if (fabs(d1) < gtol) break;  // MuJoCo's PrimalSearch stop: tolerance * ls_tolerance * |search| / scale
    if (d1 < 0) lo = alpha; else hi = alpha;
    if (d2 <= 0) break;
    double next = alpha + (-d1 / d2);
    if (!(next > lo && next < hi)) next = isfinite(hi) ? 0.5 * (lo + hi) : (alpha > 0 ? 2 * alpha : 1.0);
    if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
    alpha = next;
  }
  return alpha;
}

// primal Newton solver (mj_solPrimal with Newton), nv = 2
template <int NS>
__device__ __forceinline__ void solve(const Rows& r, double wx, double wy, double mass, double& ax_out, double& ay_out) {
  double jar[NSLOT], jv[NSLOT], force[NSLOT];
  bool active[NSLOT];
  // warmstart: cheaper of qacc_warmstart and qacc_smooth (= 0)
  double ax = 0, ay = 0, best = INFINITY;
#pragma unroll
  for (int trial = 0; trial < 2; trial++) {
    double tx = trial == 0 ? wx : 0.0, ty = trial == 0 ? wy : 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++) jar[k] = -r.aref[k] + r.JX[k] * tx + r.JY[k] * ty;
    double c = cost_update<NS>(r, jar, tx, ty, mass * tx, mass * ty, active, force);
    if (c < best) { best = c; ax = tx; ay = ty; }
  }
#pragma unroll
  for (int k = 0; k < NS; k++) jar[k] = -r.aref[k] + r.JX[k] * ax + r.JY[k] * ay;
  double Max = mass * ax, May = mass * ay;
  double cost = cost_update<NS>(r, jar, ax, ay, Max, May, active, force);
  const double scale = 1 / (mass * 2);  // 1/(meaninertia * nv)
#pragma unroll 1
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    double gx = Max, gy = May, hxx = mass, hxy = 0, hyy = mass;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      if (r.on[k]) {
        gx -= r.JX[k] * force[k];
        gy -= r.JY[k] * force[k];
        if (active[k]) {
          hxx += r.JX[k] * r.D[k] * r.JX[k];
          hxy += r.JY[k] * r.D[k] * r.JX[k];
          hyy += r.JY[k] * r.D[k] * r.JY[k];
        }
      }
    }
    // Cholesky of the 2x2 Hessian, search = -H^-1 grad; reciprocal square roots (v_rsq_f64 + 2 Newton steps) instead
    // of the IEEE sqrt and five divisions, which were a quarter of the substep's instructions
    if (hxx < MJS_MINVAL) break;
    const double i00 = rsqrt_newton(hxx), l10 = hxy * i00, s = hyy - l10 * l10;
    if (s < MJS_MINVAL) break;
    const double i11 = rsqrt_newton(s);
    const double y0 = -gx * i00, y1 = (-gy - l10 * y0) * i11;
    const double sy = y1 * i11, sx = (y0 - l10 * sy) * i00;
    double Mvx = mass * sx, Mvy = mass * sy;
    double g1 = sx * Max + sy * May, g2 = sx * Mvx + sy * Mvy, snorm = sx * sx + sy * sy;
    if (sqrt(snorm) < MJS_MINVAL) break;
#pragma unroll
    for (int k = 0; k < NS; k++) jv[k] = r.JX[k] * sx + r.JY[k] * sy;
    double alpha = line_search<NS>(r, jar, jv, g1, g2, MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale);
    if (alpha == 0) break;
    ax += alpha * sx; ay += alpha * sy;
    Max += alpha * Mvx; May += alpha * Mvy;
#pragma unroll
    for (int k = 0; k < NS; k++) jar[k] += alpha * jv[k];
    double oldcost = cost;
    cost = cost_update<NS>(r, jar, ax, ay, Max, May, active, force);
    double ngx = Max, ngy = May;
#pragma unroll
    for (int k = 0; k < NS; k++)
      if (r.on[k]) { ngx -= r.JX[k] * force[k]; ngy -= r.JY[k] * force[k]; }
    double improvement = scale * (oldcost - cost), gradient = scale * sqrt(ngx * ngx + ngy * ngy);
    if (improvement < MJS_SOLVER_TOLERANCE || gradient < MJS_SOLVER_TOLERANCE) break;
  }
  ax_out = ax; ay_out = ay;
}

// one Physics.step(): constraints from the current state -> Newton -> Euler
// (mx, my): mocap position the constraint rows were built with (see kernel: stale on substep 0)
__device__ __forceinline__ void physics_step(State& st, double mx, double my, bool& bad) {
  const double mass = MJS_PM_MASS, dt = MJS_PM_PHYSICS_DT, radius = MJS_PM_RADIUS;
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * dt);  // refsafe
  const double dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  const double tran = (1 / mass + 1 / mass + 0.0) / 3;  // body_invweight0 (translation) of the pointmass body
  Rows r;
  // weld rows x, y (the z and three rotational rows have zero Jacobian and zero residual)
  {
    double ex = mx - st.qx, ey = my - st.qy;
    double pn = sqrt(ex * ex + ey * ey);  // weld impedance uses the norm of the 6-vector residual
    double imp = impedance_default(pn);
    double R = fmax(MJS_MINVAL, (1 - imp) * tran / imp);
    double D = 1 / R;
    r.D[0] = D; r.on[0] = true; r.aref[0] = -B * (-st.vx) - K * imp * ex; r.JX[0] = -1; r.JY[0] = 0;
    r.D[1] = D; r.on[1] = true; r.aref[1] = -B * (-st.vy) - K * imp * ey; r.JX[1] = 0; r.JY[1] = -1;
  }
  // walls: rows are active when dist < 0
  const double d0 = (st.qx - MJS_PM_ARENA_LO) - radius, d1 = (st.qy - MJS_PM_ARENA_LO) - radius;
  const double d2 = -(st.qx - MJS_PM_ARENA_HI) - radius, d3 = -(st.qy - MJS_PM_ARENA_HI) - radius;
  // This lane's active walls in geom order (MuJoCo's contact order); the wavefront takes the smallest slot count that
  // covers all its lanes: 2 (welds only), 6 (one wall group), 10 (two: a lane in a corner). Row order and arithmetic
  // do not depend on the path (inactive slots contribute nothing), so a lane gets the same bits whichever path runs.
  const double dw[4] = {d0, d1, d2, d3};
  int wa = -1, wb = -1;
  double da = 0, db = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const bool hit = dw[w] < 0;
    const bool first = hit && wa < 0, second = hit && !first && wb < 0;
    if (second) { wb = w; db = dw[w]; }
    if (first) { wa = w; da = dw[w]; }
  }
  double ax, ay;
  if (__any(wb >= 0)) {
    set_contact<2>(r, wa, da, st.vx, st.vy, K, B, tran);
    set_contact<6>(r, wb, db, st.vx, st.vy, K, B, tran);
    solve<10>(r, st.wx, st.wy, mass, ax, ay);
  } else if (__any(wa >= 0)) {
    set_contact<2>(r, wa, da, st.vx, st.vy, K, B, tran);
    solve<6>(r, st.wx, st.wy, mass, ax, ay);
  } else
    solve<2>(r, st.wx, st.wy, mass, ax, ay);
  bad = bad || bad_value(ax) || bad_value(ay) || bad_value(st.qx) || bad_value(st.qy) || bad_value(st.vx) || bad_value(st.vy);
  st.wx = ax; st.wy = ay;
  st.vx += dt * ax; st.vy += dt * ay;
  st.qx += dt * st.vx; st.qy += dt * st.vy;
  st.time += dt;
}

__device__ __forceinline__ int count_contacts(const State& st) {
  const double radius = MJS_PM_RADIUS;
  int n = 1;  // sphere rests exactly on the ground plane: detected (dist == margin), not active
  n += ((st.qx - MJS_PM_ARENA_LO) <= radius);
  n += ((st.qy - MJS_PM_ARENA_LO) <= radius);
  n += (-(st.qx - MJS_PM_ARENA_HI) <= radius);
  n += (-(st.qy - MJS_PM_ARENA_HI) <= radius);
  return n;
}

// initialize_episode (point_reach.py:125-144): goal_x, goal_y, point_x, point_y ~ U(-0.45, 0.45)
__device__ __forceinline__ void episode_init(const KernelParams& p, int i, State& st) {
  const double lo = MJS_PM_ARENA_LO + MJS_PM_RADIUS, hi = MJS_PM_ARENA_HI - MJS_PM_RADIUS;
  RngCursor c = rng_open(p.rng, i);
  double gx = rng_uniform(p.rng, i, c, lo, hi), gy = rng_uniform(p.rng, i, c, lo, hi);
  double px = rng_uniform(p.rng, i, c, lo, hi), py = rng_uniform(p.rng, i, c, lo, hi);
  rng_close(p.rng, i, c);
  st.tx = gx; st.ty = gy;
  st.qx = px; st.qy = py; st.vx = 0; st.vy = 0; st.mx = px; st.my = py;
  st.time = 0; st.wx = 0; st.wy = 0;
  // distance bookkeeping is NOT reset per episode (point_reach.py:112-113)
}

template <bool IS_RESET>
__global__ __launch_bounds__(64) void kernel(KernelParams p) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N) return;
  State st = load_state(p, i);
  uint8_t flags = p.flags[i];
  double obs[OBS_DIM];
  if (IS_RESET || ((flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP)) {
    if (IS_RESET && p.reset_mask && !p.reset_mask[i]) return;
    episode_init(p, i, st);
    store_state(p, i, st);
    p.flags[i] = 0;
    obs[0] = st.qx; obs[1] = st.qy; obs[2] = st.tx; obs[3] = st.ty;
    write_outputs<OBS_DIM>(p, i, obs, 0.0, 1.0, MJS_STEP_FIRST, false, false, false, 0, count_contacts(st));
    return;
  }
  // before_step: mocap target (point_reach.py:160-163, pointmass.py:133-148)
  // dm_control steps as mj_step2; mj_step1 ("legacy" mode): the constraint rows
  // used by the first substep were built by the previous mj_step1, i.e. with the OLD mocap position.
  const double mx_old = st.mx, my_old = st.my;
  st.mx = clampd(st.qx + p.actions[(size_t)i * ACT_DIM + 0], MJS_PM_ARENA_LO, MJS_PM_ARENA_HI);
  st.my = clampd(st.qy + p.actions[(size_t)i * ACT_DIM + 1], MJS_PM_ARENA_LO, MJS_PM_ARENA_HI);
  bool bad = false;
#pragma unroll 1
  for (int s = 0; s < MJS_PM_NSUB; s++) physics_step(st, s == 0 ? mx_old : st.mx, s == 0 ? my_old : st.my, bad);
  // after_step / reward / termination (point_reach.py:165-202)
  st.prev = st.dist;
  double dx = st.qx - st.tx, dy = st.qy - st.ty;
  st.dist = sqrt(dx * dx + dy * dy);
  bool success = st.dist < MJS_PM_GOAL_THRESHOLD;
  double reward;
  switch (p.reward_type) {
    case MJS_REW_SPARSE: reward = success ? 1.0 : 0.0; break;
    case MJS_REW_DENSE_NEG_DISTANCE: reward = -st.dist; break;
    case MJS_REW_DENSE_POTENTIAL: reward = st.prev - st.dist; break;
    default: reward = -st.dist + 0.5; break;
  }
  bool terminate = success;
  double discount = success ? 0.0 : 1.0;
  if (bad) { reward = 0; discount = 0; terminate = true; }
  if (st.time >= p.time_limit) terminate = true;
  obs[0] = st.qx; obs[1] = st.qy; obs[2] = st.tx; obs[3] = st.ty;
  int ncon = count_contacts(st);
  bool terminated = terminate && discount == 0.0, truncated = terminate && discount > 0.0;
  uint8_t newflags = terminate ? FLAG_RESET_PENDING : 0;
  if (terminate && p.autoreset == MJS_AUTORESET_SAME_STEP) {
    if (p.out.terminal_obs) {
#pragma unroll
      for (int k = 0; k < OBS_DIM; k++) p.out.terminal_obs[(size_t)i * OBS_DIM + k] = obs[k];
    }
    episode_init(p, i, st);
    newflags = 0;
    obs[0] = st.qx; obs[1] = st.qy; obs[2] = st.tx; obs[3] = st.ty;
    ncon = count_contacts(st);  // d->ncon as read after the reset's mj_forward
  }
  store_state(p, i, st);
  p.flags[i] = newflags;
  write_outputs<OBS_DIM>(p, i, obs, reward, discount, terminate ? MJS_STEP_LAST : MJS_STEP_MID, terminated, truncated, success,
                         bad ? MJS_FAULT_BAD_STATE : 0, ncon);
}

}  // namespace pm
