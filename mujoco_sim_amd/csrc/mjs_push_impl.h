// mjs_push_impl.h — Robot Planar-Push fused control-step kernel (BASELINE config 4: the contact-solver path).
// Included by mjs_push.h once per block-slot count (MJS_PP_NS = namespace, MJS_PP_NB = slots): `pp` with 2 slots
// (n_objects <= 2, the registered env and BASELINE config 4) and `pp5` with 5 (the reference's dataclass default).
//
// Path replaced (reference, paths under /root/reference/mujoco_sim/; INTENDED semantics where the reference is
// broken at HEAD):
//   environments/tasks/robot_planar_push.py:185-201 before_step (episode_step += 1, servoL to (ax, ay, 0.02)),
//   per substep robot.py:261-263 servo interpolation + Physics.step() on the scene of :81-117 (UR5e + CylinderEEF +
//   n free blocks on the floor), :203-241 + tasks/base.py:47-57 reward / accomplished / step limit / discount,
//   :149-176 initialize_episode (robot, target, blocks re-drawn until nothing touches, 150 settle steps).
// Blocks (google_block.py:37-68): one of the reference's four meshes (cube / moon / pentagon / star), collided by its
// convex hull (include/mjs_block_hulls.h, generated from the reference's .obj files), category / colour / scale in
// [0.8, 1.2] drawn per episode from the env's seeded stream (deviation D-5), mass 0.1 kg, inertia of the closed mesh;
// block_shape = box keeps round 1's stand-in (a box of the cube mesh's bounding box, scale 1) as a fast variant.
// Convex pairs (cylinder-hull, hull-hull) go through an own MPR (role of mjc_Convex -> libccd), one contact per pair;
// hull-floor gives up to 4 vertex contacts (first four hull vertices at or below the plane in the table's
// farthest-point order; box: mjc_PlaneBox's corner order). Contacts are condim 4 (block) pyramids: 6 rows each.
//
// First correct version: lane per env, one wavefront per 64 envs, generic dense in-lane Newton over nv = 6 + 6 n
// dofs with the rows in per-lane scratch arrays. The arm's M and bias come from the generated code (ur5e_pp_*),
// the free blocks' from closed forms. Collision arithmetic is + - * / sqrt without FMA contraction so that contact
// sets match the CPU restatement bit for bit.
namespace MJS_PP_NS {

using rr::NJ;
constexpr int NB = MJS_PP_NB, NV = NJ + 6 * NB;
constexpr int OBS_DIM = 5 + 2 * NB, ACT_DIM = 2;
// state rows (float64 SoA): arm q, v, time, target xyz, episode_step, then per block pos3 quat4 vel6
// per block: pos3 quat4 vel6, then shape code (-1 = box stand-in, else category + 8 * colour) and mesh scale
constexpr int S_Q = 0, S_V = 6, S_TIME = 12, S_TARGET = 13, S_STEP = 16, S_BLOCK = 17, BLOCK_DIM = 15;
constexpr int STATE_DIM = S_BLOCK + BLOCK_DIM * NB;
constexpr int MAXAF = 5;  // (the LDS of the 2-slot instance has room for 20 more rows) arm-floor contacts (arm collision geoms + the EEF cylinder) the cooperative solver takes next to an arm-block coupling
constexpr int MAXCON_BLOCKS = 4 * NB + 2 * NB + (NB * (NB - 1)) / 2;  // floor-block corners, wrist proxy-block, eef-block, block-block
constexpr int MAXCON = MAXAF + MAXCON_BLOCKS;
// rows: 12 limit rows + 6 per block contact (condim 4) + 4 per arm-floor contact (condim 3). The 5-slot instance keeps its 4 x 64 row
// slots: the arm-floor rows share them with block contacts that can never all be active at once (checked when the rows are counted).
constexpr int MAXROW = NB <= 2 ? 2 * NJ + 6 * MAXCON_BLOCKS + 4 * MAXAF : 256;
static_assert(MAXROW >= 2 * NJ + 6 * MAXCON_BLOCKS, "every block contact has its rows");
// wavefronts per workgroup: same-CU wavefronts walk the same (large) code and share its cache lines; the 5-slot
// instance's cooperative workspace (~62 KB with compact rows, 113 KB dense) leaves room for two wavefronts per CU
constexpr int WAVES = NB <= 2 ? 4 : 2;
// Envs per wavefront. The coupled constraint problems of a wavefront's envs are solved one after the other by the
// whole wavefront, so the critical path of a launch is set by the wavefront with the most coupled envs; at 4096 envs
// the chip has 16x more SIMDs than 64-env wavefronts would use, so fewer envs per wavefront (the other lanes only
// help in the cooperative solves) shorten that path.
#ifndef MJS_PP_ENVS_PER_WAVE
#define MJS_PP_ENVS_PER_WAVE 4
#endif
#ifndef MJS_PP5_ENVS_PER_WAVE
#define MJS_PP5_ENVS_PER_WAVE 8
#endif
// the 5-slot instance has room for TWO wavefronts per CU (LDS, compact rows): 8 envs per wavefront keep all envs of a
// 4096-env launch resident at once (measured: 4 / 8 / 16 envs per wavefront = 297 / 403 / 308 k env-steps/s)
constexpr int EPW = NB <= 2 ? MJS_PP_ENVS_PER_WAVE : MJS_PP5_ENVS_PER_WAVE;
static_assert(EPW >= 1 && EPW <= 16 && (EPW & (EPW - 1)) == 0, "quads of lanes per (env, block): 4 * EPW lanes per block");
constexpr int QUAD_BLOCKS = 16 / EPW;  // blocks whose quads fit the wavefront in one pass

struct Block {
  V3 p;          // body origin = centre of the bottom face (free joint qpos[0:3])
  double q[4];   // orientation (w, x, y, z)
  V3 v, w;       // linear velocity of the origin (world), angular velocity (body frame)
  double shape, scale;  // shape code (-1 = box stand-in, else category + 8 * colour), mesh scale: constant within an episode
};
// what the dynamics and the collision code need to know about a block's shape
struct Shape {
  int cat;               // -1 = box stand-in, else hull category
  double scale;
  V3 c;                  // centre of mass in the body frame
  double Ixx, Iyy, Izz;  // inertia about the COM, diagonal in the body axes
  double rbound;         // bounding radius about the COM
};
static_assert(MJS_HULL_NCAT == 4, "shape code: category in bits 0-1 (always a valid table index), colour from bit 3 up");
MJS_DEV Shape shape_of(const Block& b) {
  Shape sh;
  if (b.shape < 0.0) {
    sh.cat = -1; sh.scale = 1.0;
    sh.c = v3(0, 0, MJS_BLOCK_GEOM_Z);
    sh.Ixx = MJS_BLOCK_MASS * (MJS_BLOCK_HALF[1] * MJS_BLOCK_HALF[1] + MJS_BLOCK_HALF[2] * MJS_BLOCK_HALF[2]) / 3;
    sh.Iyy = MJS_BLOCK_MASS * (MJS_BLOCK_HALF[0] * MJS_BLOCK_HALF[0] + MJS_BLOCK_HALF[2] * MJS_BLOCK_HALF[2]) / 3;
    sh.Izz = MJS_BLOCK_MASS * (MJS_BLOCK_HALF[0] * MJS_BLOCK_HALF[0] + MJS_BLOCK_HALF[1] * MJS_BLOCK_HALF[1]) / 3;
    sh.rbound = sqrt(MJS_BLOCK_HALF[0] * MJS_BLOCK_HALF[0] + MJS_BLOCK_HALF[1] * MJS_BLOCK_HALF[1] + MJS_BLOCK_HALF[2] * MJS_BLOCK_HALF[2]);
  } else {
    const int cat = ((int)b.shape) & 3;
    const double sc = b.scale;
    sh.cat = cat; sh.scale = sc;
    sh.c = v3(MJS_HULL_COM[cat][0] * sc, MJS_HULL_COM[cat][1] * sc, MJS_HULL_COM[cat][2] * sc);
    sh.Ixx = MJS_BLOCK_MASS * MJS_HULL_INERTIA_PER_MASS[cat][0] * sc * sc;
    sh.Iyy = MJS_BLOCK_MASS * MJS_HULL_INERTIA_PER_MASS[cat][1] * sc * sc;
    sh.Izz = MJS_BLOCK_MASS * MJS_HULL_INERTIA_PER_MASS[cat][2] * sc * sc;
    sh.rbound = MJS_HULL_RBOUND[cat] * sc;
  }
  return sh;
}
// a free body's share of mj_setConst's meaninertia (sum of the diagonal of M at qpos0): 3 m + tr(I_c) + 2 m |c|^2
MJS_DEV double shape_inertia_trace(const Shape& sh) {
  return 3 * MJS_BLOCK_MASS + sh.Ixx + sh.Iyy + sh.Izz + 2 * MJS_BLOCK_MASS * dot(sh.c, sh.c);
}
struct World {
  double q[NJ], v[NJ], time, target[3], episode_step;
  Block b[NB];
};

MJS_DEV World load_world_at(const double* base, size_t N, int i) {
  World s;
  const double* st = base + i;
#pragma unroll
  for (int j = 0; j < NJ; j++) { s.q[j] = st[(S_Q + j) * N]; s.v[j] = st[(S_V + j) * N]; }
  s.time = st[S_TIME * N];
#pragma unroll
  for (int k = 0; k < 3; k++) s.target[k] = st[(S_TARGET + k) * N];
  s.episode_step = st[S_STEP * N];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const double* bs = st + (size_t)(S_BLOCK + BLOCK_DIM * b) * N;
    s.b[b].p = v3(bs[0], bs[N], bs[2 * N]);
#pragma unroll
    for (int k = 0; k < 4; k++) s.b[b].q[k] = bs[(3 + k) * N];
    s.b[b].v = v3(bs[7 * N], bs[8 * N], bs[9 * N]);
    s.b[b].w = v3(bs[10 * N], bs[11 * N], bs[12 * N]);
    s.b[b].shape = bs[13 * N]; s.b[b].scale = bs[14 * N];
  }
  return s;
}
MJS_DEV World load_world(const KernelParams& p, int i) { return load_world_at(p.state, (size_t)p.N, i); }
MJS_DEV void store_world_at(double* base, size_t N, int i, const World& s) {
  double* st = base + i;
#pragma unroll
  for (int j = 0; j < NJ; j++) { st[(S_Q + j) * N] = s.q[j]; st[(S_V + j) * N] = s.v[j]; }
  st[S_TIME * N] = s.time;
#pragma unroll
  for (int k = 0; k < 3; k++) st[(S_TARGET + k) * N] = s.target[k];
  st[S_STEP * N] = s.episode_step;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    double* bs = st + (size_t)(S_BLOCK + BLOCK_DIM * b) * N;
    bs[0] = s.b[b].p.x; bs[N] = s.b[b].p.y; bs[2 * N] = s.b[b].p.z;
#pragma unroll
    for (int k = 0; k < 4; k++) bs[(3 + k) * N] = s.b[b].q[k];
    bs[7 * N] = s.b[b].v.x; bs[8 * N] = s.b[b].v.y; bs[9 * N] = s.b[b].v.z;
    bs[10 * N] = s.b[b].w.x; bs[11 * N] = s.b[b].w.y; bs[12 * N] = s.b[b].w.z;
    bs[13 * N] = s.b[b].shape; bs[14 * N] = s.b[b].scale;
  }
}
MJS_DEV void store_world(const KernelParams& p, int i, const World& s) { store_world_at(p.state, (size_t)p.N, i, s); }
// The handle's state buffer has a SECOND SLOT per env: rows [STATE_DIM, 2 STATE_DIM) hold the env's NEXT episode while it is being
// prepared (the draws of initialize_episode + as many of its 150 settle steps as have run), row 2 STATE_DIM how far that has got:
// -1 nothing drawn yet, 0 .. 149 settle steps done, 150 ready. An env's reset is a function of its own RNG stream only, so it can be
// worked out ahead of time: in launches with p.prefetch the grid's second half are prefetch workgroups that advance the next-episode
// slot of the envs of their group by PREFETCH_CHUNK (10) substeps per launch, on CUs the stepping workgroups leave idle; when an
// episode ends, the next launch swaps the slots (no settle steps inside a step launch). Ownership is decided by what both roles
// read at launch start: an env whose reset is pending belongs to the stepping workgroup (which finishes whatever is missing
// inline: the old path, bit for bit the same arithmetic), every other env's second slot to the prefetch workgroup. mjs_seed
// empties the slot; mjs_get_state / mjs_set_state carry it (a checkpoint resumes with the prepared episode).
// Rows PROG_ROW + 1 .. + 6: the prepared episode's reset joints = the servo set-point of its settle steps (robot.py:185-189); the
// next 12: the carried cos / sin of its joint angles (the substep loop updates them incrementally: with them in the slot a settle
// phase that is cut into chunks is bit for bit the uninterrupted one, so WHEN the chunks ran - which depends on the order in
// which a launch's workgroups start - cannot show in any result).
constexpr int NEXT_ROW0 = STATE_DIM, PROG_ROW = 2 * STATE_DIM, CTRL_ROW0 = PROG_ROW + 1, CS_ROW0 = CTRL_ROW0 + NJ, FULL_STATE_DIM = CS_ROW0 + 2 * NJ;
constexpr int PREFETCH_CHUNK = 10;

MJS_DEV M3 quat_to_m3(const double* q) {  // unit quaternion -> rotation (columns = body axes in the world)
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  return M3{v3(1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)), v3(2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)),
            v3(2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y))};
}
MJS_DEV V3 rot(const M3& R, V3 a) { return madd(madd(a.x * R.cx, a.y, R.cy), a.z, R.cz); }
MJS_DEV V3 rot_t(const M3& R, V3 a) { return v3(dot(R.cx, a), dot(R.cy, a), dot(R.cz, a)); }

// ------------------------------------------------------------------------------------------------ collision
// Same operations, in the same order, as oracle/om_engine.c (collide_plane box branch, mpr_penetration): the two
// sides are written independently but must take the same branches, hence no FMA contraction here.
#pragma clang fp contract(off)

// biased tie-break thresholds, identical to oracle/om_engine.c (see the comment there)
constexpr double MPR_EPS_DIR = 1e-10, MPR_EPS_LEN = 1e-13, MPR_EPS_VOL = 1e-16;

struct Geom {  // a convex collision geom in the world: box (half extents s), cylinder (radius s.x, half length s.y) or the
  V3 c;        // convex hull of a block mesh (category cat, scale s.x, vertices relative to the COM = the geom centre c)
  M3 R;
  V3 s;
  bool box;
  int cat;     // >= 0: hull
};
constexpr double MPR_EPS_TIE = 1e-12;  // hull vertices whose projections differ by less are tied: the first in table order wins
struct Contact {
  double dist;
  V3 pos, n;
  int ba, bb;  // bodies: 0 = world, 1 = arm (a geom welded to link 6), 2 + i = block i; normal points from a to b
  double tran; // body_invweight0 (translation) of the two bodies, summed: diagApprox of the pyramid rows
};

// the hull tables (vertices relative to the COM) staged in LDS behind the cooperative workspaces (kernel prologue): the scans
// below read them with per-lane category indices, an uncoalesced gather when served from the constant segment in global memory
MJS_DEV const double* hull_lds();
MJS_DEV double hull_c(int cat, int i, int k) { return hull_lds()[(cat * MJS_HULL_MAXV + i) * 3 + k]; }
// Lanes per env: only EPW lanes of a wavefront carry an env, so the hot path's convex pairs are evaluated by GROUPS of
// LPE = 64 / EPW lanes (one DPP row of 16, or half a row) that hold the same pair and share the hull scans (COOP = true)
constexpr int LPE = 64 / EPW;
static_assert(LPE == 16 || LPE == 8, "group reductions below: one DPP row or half a row per env");
template <int CTRL>
MJS_DEV double dppd(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
MJS_DEV int dppi(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }
// all-reduce over the lanes of an env group, the same bits in every lane: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
MJS_DEV double group_max(double x) {
  x = fmax(x, dppd<0xB1>(x)); x = fmax(x, dppd<0x4E>(x)); x = fmax(x, dppd<0x141>(x));
  if constexpr (LPE == 16) x = fmax(x, dppd<0x140>(x));
  return x;
}
MJS_DEV int group_min(int x) {
  x = min(x, dppi<0xB1>(x)); x = min(x, dppi<0x4E>(x)); x = min(x, dppi<0x141>(x));
  if constexpr (LPE == 16) x = min(x, dppi<0x140>(x));
  return x;
}
// COOP: called by all lanes of an env group with identical arguments; the result is identical in every lane
template <bool COOP>
MJS_DEV V3 support(const Geom& g, V3 dir) {
  const V3 loc = rot_t(g.R, dir);
  V3 res;
  if (g.cat >= 0) {
    // hull vertex with the largest projection on the direction; structural ties (a face or an edge square to the direction) go
    // to the LOWEST table index among the vertices within MPR_EPS_TIE of the maximum, whatever the rounding noise
    const int nvx = MJS_HULL_NV[g.cat];
    const double sc = g.s.x;
    const double lsx = loc.x * sc, lsy = loc.y * sc, lsz = loc.z * sc;
    double best = -1e300;
    int idx = 1 << 20;
    if constexpr (COOP) {
      // the lane's share of the vertices (sub, sub + LPE, ...) is scanned once: the projections stay in registers for the tie
      // pass (the table is padded to MJS_HULL_MAXV rows: reads past nvx are in bounds and masked)
      constexpr int MAXJ = (MJS_HULL_MAXV + LPE - 1) / LPE;
      const int sub = (int)(threadIdx.x & (LPE - 1));
      double prj[MAXJ];
#pragma unroll
      for (int j = 0; j < MAXJ; j++) {
        const int i = sub + j * LPE;
        const int ic = i < MJS_HULL_MAXV ? i : MJS_HULL_MAXV - 1;
        const double pr = lsx * hull_c(g.cat, ic, 0) + lsy * hull_c(g.cat, ic, 1) + lsz * hull_c(g.cat, ic, 2);
        prj[j] = i < nvx ? pr : -1e300;
        best = prj[j] > best ? prj[j] : best;
      }
      best = group_max(best);
#pragma unroll
      for (int j = MAXJ - 1; j >= 0; j--)  // descending: the lowest index of this lane within the tie band survives
        if (prj[j] >= best - MPR_EPS_TIE) idx = sub + j * LPE;
      idx = group_min(idx);
    } else {
      for (int i = 0; i < nvx; i++) {
        const double pr = lsx * hull_c(g.cat, i, 0) + lsy * hull_c(g.cat, i, 1) + lsz * hull_c(g.cat, i, 2);
        best = pr > best ? pr : best;
      }
      for (int i = 0; i < nvx; i++) {
        const double pr = lsx * hull_c(g.cat, i, 0) + lsy * hull_c(g.cat, i, 1) + lsz * hull_c(g.cat, i, 2);
        if (pr >= best - MPR_EPS_TIE) { idx = i; break; }
      }
    }
    res = v3(hull_c(g.cat, idx, 0) * sc, hull_c(g.cat, idx, 1) * sc, hull_c(g.cat, idx, 2) * sc);
  } else if (g.box) {
    res = v3(loc.x >= -MPR_EPS_DIR ? g.s.x : -g.s.x, loc.y >= -MPR_EPS_DIR ? g.s.y : -g.s.y, loc.z >= -MPR_EPS_DIR ? g.s.z : -g.s.z);
  } else {
    const double len = sqrt(loc.x * loc.x + loc.y * loc.y);
    if (len > MPR_EPS_DIR) res = v3(g.s.x * loc.x / len, g.s.x * loc.y / len, 0); else res = v3(0, 0, 0);
    res.z = loc.z >= -MPR_EPS_DIR ? g.s.y : -g.s.y;
  }
  const V3 w = v3(g.R.cx.x * res.x + g.R.cy.x * res.y + g.R.cz.x * res.z, g.R.cx.y * res.x + g.R.cy.y * res.y + g.R.cz.y * res.z,
                  g.R.cx.z * res.x + g.R.cy.z * res.y + g.R.cz.z * res.z);
  return v3(w.x + g.c.x, w.y + g.c.y, w.z + g.c.z);
}
struct MprVert { V3 v, a, b; };
template <bool COOP>
MJS_DEV MprVert mpr_support(const Geom& g1, const Geom& g2, V3 dir) {
  MprVert s;
  s.a = support<COOP>(g1, dir);
  s.b = support<COOP>(g2, v3(-dir.x, -dir.y, -dir.z));
  s.v = v3(s.a.x - s.b.x, s.a.y - s.b.y, s.a.z - s.b.z);
  return s;
}
MJS_DEV double dot_nc(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MJS_DEV V3 cross_nc(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MJS_DEV bool normalize_nc(V3& v) {
  const double n = sqrt(dot_nc(v, v));
  if (n < 1e-14) return false;
  v = v3(v.x / n, v.y / n, v.z / n);
  return true;
}
MJS_DEV V3 sub_nc(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MJS_DEV V3 any_perpendicular(V3 v) {
  const double ax = fabs(v.x), ay = fabs(v.y), az = fabs(v.z);
  const int k = ax <= ay ? (ax <= az ? 0 : 2) : (ay <= az ? 1 : 2);
  return cross_nc(v, v3(k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0));
}
// Minkowski portal refinement on g1 - g2; true + (depth, normal g1 -> g2, pos) when the geoms overlap
template <bool COOP>
MJS_DEV bool mpr_core(const Geom& g1, const Geom& g2, double& depth, V3& normal, V3& pos) {
  MprVert v0, v1, v2, v3_, v4;
  v0.a = g1.c; v0.b = g2.c; v0.v = sub_nc(g1.c, g2.c);
  if (sqrt(dot_nc(v0.v, v0.v)) < 1e-12) v0.v.x = 1e-5;
  V3 dir = v3(-v0.v.x, -v0.v.y, -v0.v.z);
  normalize_nc(dir);
  v1 = mpr_support<COOP>(g1, g2, dir);
  if (dot_nc(v1.v, dir) <= 0) return false;
  dir = cross_nc(v0.v, v1.v);
  if (!normalize_nc(dir)) { dir = any_perpendicular(v0.v); normalize_nc(dir); }
  v2 = mpr_support<COOP>(g1, g2, dir);
  if (dot_nc(v2.v, dir) <= 0) return false;
  dir = cross_nc(sub_nc(v1.v, v0.v), sub_nc(v2.v, v0.v));
  if (!normalize_nc(dir)) return false;
  if (dot_nc(dir, v0.v) > MPR_EPS_LEN) { MprVert t = v1; v1 = v2; v2 = t; dir = v3(-dir.x, -dir.y, -dir.z); }
  for (int it = 0;; it++) {  // portal discovery
    if (it >= MJS_MPR_MAX_ITER) return false;
    v3_ = mpr_support<COOP>(g1, g2, dir);
    if (dot_nc(v3_.v, dir) <= 0) return false;
    bool cont = false;
    if (dot_nc(cross_nc(v1.v, v3_.v), v0.v) < -MPR_EPS_VOL) { v2 = v3_; cont = true; }
    else if (dot_nc(cross_nc(v3_.v, v2.v), v0.v) < -MPR_EPS_VOL) { v1 = v3_; cont = true; }
    if (!cont) break;
    dir = cross_nc(sub_nc(v1.v, v0.v), sub_nc(v2.v, v0.v));
    if (!normalize_nc(dir)) return false;
  }
  bool hit = false;
  for (int it = 0; it < MJS_MPR_MAX_ITER; it++) {  // portal refinement
    dir = cross_nc(sub_nc(v2.v, v1.v), sub_nc(v3_.v, v1.v));
    if (!normalize_nc(dir)) return false;
    if (dot_nc(dir, v1.v) >= -MPR_EPS_LEN) hit = true;
    v4 = mpr_support<COOP>(g1, g2, dir);
    const double reach = dot_nc(v4.v, dir);
    if (!hit && reach < 0) return false;
    const double progress = dot_nc(sub_nc(v4.v, v3_.v), dir);
    if (progress <= MJS_MPR_TOLERANCE || it == MJS_MPR_MAX_ITER - 1) {
      if (!hit) return false;
      depth = dot_nc(v1.v, dir);
      normal = dir;
      const V3 c23 = cross_nc(v2.v, v3_.v), c13 = cross_nc(v1.v, v3_.v), c12 = cross_nc(v1.v, v2.v);
      double b0 = dot_nc(v1.v, c23), b1 = -dot_nc(v0.v, c23), b2 = dot_nc(v0.v, c13), b3 = -dot_nc(v0.v, c12);
      double sum = b0 + b1 + b2 + b3;
      if (fabs(sum) < 1e-30) { b0 = 0; b1 = b2 = b3 = 1; sum = 3; }
      const double pax = (b0 * v0.a.x + b1 * v1.a.x + b2 * v2.a.x + b3 * v3_.a.x) / sum, pbx = (b0 * v0.b.x + b1 * v1.b.x + b2 * v2.b.x + b3 * v3_.b.x) / sum;
      const double pay = (b0 * v0.a.y + b1 * v1.a.y + b2 * v2.a.y + b3 * v3_.a.y) / sum, pby = (b0 * v0.b.y + b1 * v1.b.y + b2 * v2.b.y + b3 * v3_.b.y) / sum;
      const double paz = (b0 * v0.a.z + b1 * v1.a.z + b2 * v2.a.z + b3 * v3_.a.z) / sum, pbz = (b0 * v0.b.z + b1 * v1.b.z + b2 * v2.b.z + b3 * v3_.b.z) / sum;
      pos = v3(0.5 * (pax + pbx), 0.5 * (pay + pby), 0.5 * (paz + pbz));
      return true;
    }
    const V3 t1 = cross_nc(v4.v, v0.v);
    if (dot_nc(v1.v, t1) > MPR_EPS_VOL) {
      if (dot_nc(v2.v, t1) > MPR_EPS_VOL) v1 = v4; else v3_ = v4;
    } else {
      if (dot_nc(v3_.v, t1) > MPR_EPS_VOL) v2 = v4; else v1 = v4;
    }
  }
  return false;
}
// serial version (reset / contact counting on single lanes): out of line. The group-parallel version of the hot path is inlined
// into its one call site (a loop over the env's pairs) with the two geoms read from LDS into registers: as a by-reference call
// every access to a geom was a scratch load by all 64 lanes (profiles/r02_g_*: 1.9 GB of HBM-side traffic per launch).
__device__ __noinline__ bool mpr_penetration_serial(const Geom& g1, const Geom& g2, double& depth, V3& normal, V3& pos) {
  return mpr_core<false>(g1, g2, depth, normal, pos);
}
MJS_DEV double rbound(const Geom& g) {
  return g.cat >= 0 ? MJS_HULL_RBOUND[g.cat] * g.s.x : g.box ? sqrt(g.s.x * g.s.x + g.s.y * g.s.y + g.s.z * g.s.z) : sqrt(g.s.x * g.s.x + g.s.y * g.s.y);
}
MJS_DEV bool collide_convex(const Geom& g1, const Geom& g2, int ba, int bb, double tran, Contact& c) {
  const V3 diff = sub_nc(g2.c, g1.c);
  const double bound = rbound(g1) + rbound(g2);
  if (dot_nc(diff, diff) > bound * bound) return false;
  double depth;
  if (!mpr_penetration_serial(g1, g2, depth, c.n, c.pos)) return false;
  c.dist = -depth;
  c.ba = ba; c.bb = bb;
  c.tran = tran;
  return true;
}
MJS_DEV Geom block_geom(const Block& b, const M3& R) {
  Geom g;
  g.R = R;
  if (b.shape < 0.0) {
    const double gz = MJS_BLOCK_GEOM_Z;  // geom_xpos = xpos + xmat * geom_pos
    g.c = v3(b.p.x + R.cz.x * gz, b.p.y + R.cz.y * gz, b.p.z + R.cz.z * gz);
    g.s = v3(MJS_BLOCK_HALF[0], MJS_BLOCK_HALF[1], MJS_BLOCK_HALF[2]);
    g.box = true;
    g.cat = -1;
  } else {  // a mesh geom's frame sits at the mesh's centre of mass: geom_pos = COM * scale
    const int cat = ((int)b.shape) & 3;
    const double sc = b.scale;
    const double cx = MJS_HULL_COM[cat][0] * sc, cy = MJS_HULL_COM[cat][1] * sc, cz = MJS_HULL_COM[cat][2] * sc;
    g.c = v3(b.p.x + (R.cx.x * cx + R.cy.x * cy + R.cz.x * cz), b.p.y + (R.cx.y * cx + R.cy.y * cy + R.cz.y * cz), b.p.z + (R.cx.z * cx + R.cy.z * cy + R.cz.z * cz));
    g.s = v3(sc, sc, sc);
    g.box = false;
    g.cat = cat;
  }
  return g;
}
// local coordinates (geom frame) of the k-th floor-contact candidate of a block geom and how many there are: the 8 corners
// of the box in mjc_PlaneBox's order (x index fastest), or the hull's vertices in the table's farthest-point order
MJS_DEV int floor_candidates(const Geom& g) { return g.cat >= 0 ? MJS_HULL_NV[g.cat] : 8; }
MJS_DEV V3 floor_candidate(const Geom& g, int i) {
  if (g.cat >= 0)
    return v3(hull_c(g.cat, i, 0) * g.s.x, hull_c(g.cat, i, 1) * g.s.x, hull_c(g.cat, i, 2) * g.s.x);
  return v3((i & 1) ? g.s.x : -g.s.x, (i & 2) ? g.s.y : -g.s.y, (i & 4) ? g.s.z : -g.s.z);
}
// mjc_PlaneBox against the floor z = 0: corners at or below the plane, at most 4, x index fastest (hull: the same rule on
// its vertices in table order)
MJS_DEV int floor_box(const Geom& g, int bb, Contact* out) {
  int cnt = 0;
  const int ncand = floor_candidates(g);
  for (int i = 0; i < ncand && cnt < 4; i++) {
    const V3 lc = floor_candidate(g, i);
    const double lx = lc.x, ly = lc.y, lz = lc.z;
    const V3 corner = v3(g.R.cx.x * lx + g.R.cy.x * ly + g.R.cz.x * lz + g.c.x, g.R.cx.y * lx + g.R.cy.y * ly + g.R.cz.y * lz + g.c.y,
                         g.R.cx.z * lx + g.R.cy.z * ly + g.R.cz.z * lz + g.c.z);
    const double dist = corner.z;  // (corner - plane pos) . n with n = +z, plane through the origin
    if (dist > 0.0) continue;
    Contact& c = out[cnt++];
    c.dist = dist;
    c.n = v3(0, 0, 1);
    c.pos = v3(corner.x, corner.y, corner.z - dist * 0.5);
    c.ba = 0; c.bb = bb;
    c.tran = 1.0 / MJS_BLOCK_MASS;  // world 0 + free block 1/m
  }
  return cnt;
}
#pragma clang fp contract(fast)  // back to hipcc's default (-ffp-contract=fast): "on" would stop cross-statement fusion in everything included later

MJS_DEV Geom eef_geom(const rr::Chain& ch) {  // CylinderEEF: axis = flange z = wrist_3 y, centre at flange z = 0.051
  Geom g;
  const M3 R6 = ch.R[6];
  g.R = M3{R6.cx, -R6.cz, R6.cy};  // flange frame (MJS_UR_FLANGE_QUAT): x = wrist_3 x, y = -wrist_3 z, z = wrist_3 y
  g.c = madd(ch.p[6], MJS_UR_FLANGE_POS[1] + MJS_CYL_POS_Z, R6.cy);
  g.s = v3(MJS_CYL_RADIUS, MJS_CYL_HALFLEN, 0);
  g.box = false;
  g.cat = -1;
  return g;
}
// number of DETECTED EEF cylinder - floor contacts (what mj_collision lists: 0..4)
MJS_DEV int count_eef_floor_contacts(const rr::Chain& ch) {
  const Geom eg = eef_geom(ch);
  int n = 0;
  rr::plane_cylinder_contacts(eg.c, eg.R.cz, eg.R.cx, eg.s.x, eg.s.y, [&](V3, double) { n++; });
  return n;
}
MJS_DEV Geom wrist3_proxy_geom(const rr::Chain& ch) {  // the arm's last collision proxy is a CYLINDER (MJS_UR_COL_* index 9):
  Geom g;                                             // the only arm geom whose pair with a box is evaluated (convex-convex)
  constexpr int G = MJS_UR_NCOLGEOM - 1;
  const M3 R6 = ch.R[6];
  g.R = M3{R6.cx, R6.cz, -R6.cy};  // geom quat (1,1,0,0): +90 deg about x of the body frame
  g.c = madd(madd(madd(ch.p[6], MJS_UR_COL_POS[G][0], R6.cx), MJS_UR_COL_POS[G][1], R6.cy), MJS_UR_COL_POS[G][2], R6.cz);
  g.s = v3(MJS_UR_COL_SIZE[G][0], MJS_UR_COL_SIZE[G][1], 0);
  g.box = false;
  g.cat = -1;
  return g;
}
MJS_DEV V3 eef_tcp_position(const rr::Chain& c) { return madd(c.p[6], MJS_UR_FLANGE_POS[1] + MJS_CYL_TCP_Z, c.R[6].cy); }

// ---- hot-path detection: static slots only (no indexed memory)
struct FloorSlots {
  bool on[4];       // slot k = k-th corner at or below the floor in mjc_PlaneBox order (active when dist < 0)
  double dist[4];
  V3 r[4];          // contact point - body origin
  int n;            // slots found (detected contacts, dist <= 0): what mj_collision counts
};
#pragma clang fp contract(off)
MJS_DEV FloorSlots floor_slots(const Geom& g, V3 origin) {
  FloorSlots fs;
#pragma unroll
  for (int k = 0; k < 4; k++) { fs.on[k] = false; fs.dist[k] = 0; fs.r[k] = v3(0, 0, 0); }
  int rank = 0;  // number of detected corners / vertices so far
  const int ncand = floor_candidates(g);
#pragma unroll 1
  for (int i = 0; i < ncand && rank < 4; i++) {
    const V3 lc = floor_candidate(g, i);
    const double lx = lc.x, ly = lc.y, lz = lc.z;
    const V3 corner = v3(g.R.cx.x * lx + g.R.cy.x * ly + g.R.cz.x * lz + g.c.x, g.R.cx.y * lx + g.R.cy.y * ly + g.R.cz.y * lz + g.c.y,
                         g.R.cx.z * lx + g.R.cy.z * ly + g.R.cz.z * lz + g.c.z);
    const double dist = corner.z;
    const bool hit = !(dist > 0.0);
    const V3 r = v3(corner.x - origin.x, corner.y - origin.y, (corner.z - dist * 0.5) - origin.z);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (hit && rank == k) { fs.on[k] = dist < 0.0; fs.dist[k] = dist; fs.r[k] = r; }
    rank += hit ? 1 : 0;
  }
  fs.n = rank;
  return fs;
}
// the same slots computed by all lanes of an env group together (identical arguments, identical result in every lane): the
// candidates are dealt to the lanes, the (at most) four LOWEST candidate indices at or below the floor are found by group
// minima, and every lane rebuilds their slot data. A block that is being pushed rocks on fewer than four bottom vertices, and
// the serial scan then walks the whole hull (96 vertices) in every substep.
#pragma clang fp contract(off)
MJS_DEV FloorSlots floor_slots_group(const Geom& g, V3 origin) {
  FloorSlots fs;
#pragma unroll
  for (int k = 0; k < 4; k++) { fs.on[k] = false; fs.dist[k] = 0; fs.r[k] = v3(0, 0, 0); }
  const int ncand = floor_candidates(g), sub = (int)(threadIdx.x & (LPE - 1));
  constexpr int NONE = 1 << 20;
  fs.n = 0;
  unsigned long long mine = 0;  // bit j: this lane's j-th candidate (index sub + j * LPE) is at or below the floor
  for (int j = 0, i = sub; i < ncand; j++, i += LPE) {
    const V3 lc = floor_candidate(g, i);
    const double z = g.R.cx.z * lc.x + g.R.cy.z * lc.y + g.R.cz.z * lc.z + g.c.z;
    if (!(z > 0.0)) mine |= 1ull << j;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int my = mine ? sub + (__ffsll((long long)mine) - 1) * LPE : NONE;
    const int idx = group_min(my);
    if (idx == NONE) break;  // group-uniform
    if (my == idx) mine &= mine - 1;
    const V3 lc = floor_candidate(g, idx);
    const double lx = lc.x, ly = lc.y, lz = lc.z;
    const V3 corner = v3(g.R.cx.x * lx + g.R.cy.x * ly + g.R.cz.x * lz + g.c.x, g.R.cx.y * lx + g.R.cy.y * ly + g.R.cz.y * lz + g.c.y,
                         g.R.cx.z * lx + g.R.cy.z * ly + g.R.cz.z * lz + g.c.z);
    const double dist = corner.z;
    fs.on[k] = dist < 0.0; fs.dist[k] = dist;
    fs.r[k] = v3(corner.x - origin.x, corner.y - origin.y, (corner.z - dist * 0.5) - origin.z);
    fs.n = k + 1;
  }
  return fs;
}
#pragma clang fp contract(fast)  // back to hipcc's default (-ffp-contract=fast): "on" would stop cross-statement fusion in everything included later
// convex pairs of the scene in MuJoCo's pair order: wrist proxy - block b (NB slots), EEF - block b (NB), block a - block b
// (a < b, row-major)
constexpr int NCVX = 2 * NB + (NB * (NB - 1)) / 2;
constexpr int pair_slot(int a, int b) { return 2 * NB + a * NB - (a * (a + 1)) / 2 + (b - a - 1); }
struct ConvexHits {
  bool hit[NCVX];
  double dist[NCVX];
  V3 pos[NCVX], n[NCVX];
};
// one convex pair, evaluated by all lanes of an env group together (identical arguments, identical result in every lane)
struct PairHit { bool hit; double dist; V3 pos, n; };
// pair slot k -> indices of its two geoms in DetLds::g (0 wrist proxy, 1 EEF cylinder, 2 + b block b), MuJoCo's pair order
MJS_DEV void pair_geoms(int k, int& ia, int& ib) {
  if (k < NB) { ia = 0; ib = 2 + k; }
  else if (k < 2 * NB) { ia = 1; ib = 2 + (k - NB); }
  else {
    ia = 2; ib = 3;
#pragma unroll
    for (int a = 0; a < NB; a++) {
#pragma unroll
      for (int b = a + 1; b < NB; b++)
        if (pair_slot(a, b) == k) { ia = 2 + a; ib = 2 + b; }
    }
  }
}
MJS_DEV PairHit convex_pair_group(const Geom& g1, const Geom& g2) {
  PairHit h{false, 0.0, v3(0, 0, 0), v3(0, 0, 1)};
  const V3 diff = sub_nc(g2.c, g1.c);
  const double bound = rbound(g1) + rbound(g2);
  if (dot_nc(diff, diff) > bound * bound) return h;
  double depth;
  V3 nn, pp_;
  if (!mpr_core<true>(g1, g2, depth, nn, pp_)) return h;
  h.hit = true; h.dist = -depth; h.pos = pp_; h.n = nn;
  return h;
}
#pragma clang fp contract(fast)  // back to hipcc's default (-ffp-contract=fast): "on" would stop cross-statement fusion in everything included later

// all contacts of the scene in MuJoCo's pair order (geom ids: floor, arm capsules, EEF cylinder, blocks): floor-block i
// (<= 4 each), EEF-block i, block-block. Arm capsules vs floor and EEF vs floor are only COUNTED (D-8): `extra`.
MJS_DEV int detect_contacts(const rr::Chain& ch, const World& s, const M3* Rb, int nb, Contact* con, int& extra, bool& eef_floor_active) {
  int n = 0;
  extra = rr::count_floor_contacts(ch);
  const Geom eg = eef_geom(ch);
  {  // mjc_PlaneCylinder: up to four contacts once the deepest rim point of the EEF cylinder reaches the floor (the arm sagging by 19 mm)
    const double prj = eg.R.cz.z, rad = sqrt(fmax(0.0, 1.0 - prj * prj));
    const double lowest = eg.c.z - fabs(prj) * eg.s.y - rad * eg.s.x;
    eef_floor_active = lowest < 0.0;
    extra += count_eef_floor_contacts(ch);
  }
  Geom bg[NB];
  for (int b = 0; b < nb; b++) {
    bg[b] = block_geom(s.b[b], Rb[b]);
    n += floor_box(bg[b], 2 + b, con + n);
  }
  const Geom wg = wrist3_proxy_geom(ch);
  for (int b = 0; b < nb; b++)
    if (collide_convex(wg, bg[b], 1, 2 + b, UR5E_PP_WRIST3_BODY_INVWEIGHT0[0] + 1.0 / MJS_BLOCK_MASS, con[n])) n++;
  for (int b = 0; b < nb; b++)
    if (collide_convex(eg, bg[b], 1, 2 + b, UR5E_PP_EEF_BODY_INVWEIGHT0[0] + 1.0 / MJS_BLOCK_MASS, con[n])) n++;
  for (int a = 0; a < nb; a++)
    for (int b = a + 1; b < nb; b++)
      if (collide_convex(bg[a], bg[b], 2 + a, 2 + b, 2.0 / MJS_BLOCK_MASS, con[n])) n++;
  return n;
}

// ------------------------------------------------------------------------------------------------ dynamics
// Free block, generalised velocity (v_origin in the world, w in the body frame), COM at c in the body frame, inertia I_c about
// the COM diagonal in the body axes (Shape):
//   M = [[m I, -m R C], [m C R^T, I_c - m C C]],  C = [c]x
//   smooth force = -( m R (w x (w x c)) - m g ;  w x I_c w + m c x (w x (w x c)) - m c x R^T g )
constexpr double BLK_INVW_TRAN = 1.0 / MJS_BLOCK_MASS;
MJS_DEV void block_smooth_force(const M3& R, V3 w, const Shape& sh, double* f) {
  const double m = MJS_BLOCK_MASS;
  const V3 c = sh.c, grav = v3(0, 0, MJS_GRAVITY_Z);
  const V3 wwc = cross(w, cross(w, c));
  const V3 lin = m * rot(R, wwc) - m * grav;
  const V3 Iw = v3(sh.Ixx * w.x, sh.Iyy * w.y, sh.Izz * w.z);
  const V3 ang = cross(w, Iw) + m * cross(c, wwc) - m * cross(c, rot_t(R, grav));
  f[0] = -lin.x; f[1] = -lin.y; f[2] = -lin.z; f[3] = -ang.x; f[4] = -ang.y; f[5] = -ang.z;
}

// Decoupled case (the common one): a block without an active arm-block or block-block contact. With its floor contacts
// it is an independent 6-dof problem with at most 4 corner contacts x 6 pyramid edges whose frame is constant (n = +z,
// t1 = +y, t2 = -x: mju_makeFrame of (0, 0, 1)): a row is (F, (axis_d x r) . F) for the three frame vectors F and
// (0, axis_d . n) for the torsional row. Solved by quad_block_floor below. Cold start at qacc_smooth: a warm start from
// the previous substep's accelerations (mjData.qacc_warmstart; tried per component) needed MORE Newton iterations
// (2.4 -> 3.8 per coupled solve) and moved hand-set scenarios by the solver tolerance (5e-10) instead of 1e-16.
// x = M^-1 f for a block in closed form: with the 3x3 blocks M = [[m I, B], [B^T, D]], B = -m R C and D = I_c - m C C, the
// Schur complement D - B^T B / m is I_c = diag(Ixx, Iyy, Izz) for ANY c: alpha = I_c^-1 (f_ang - c x R^T f_lin),
// a = f_lin / m + R (c x alpha): no factorisation
MJS_DEV void block_minv(const M3& R, const Shape& sh, const double* f, double* x) {
  const V3 fl = v3(f[0], f[1], f[2]);
  const V3 cu = cross(sh.c, rot_t(R, fl));
  const double ax = (f[3] - cu.x) / sh.Ixx, ay = (f[4] - cu.y) / sh.Iyy, az = (f[5] - cu.z) / sh.Izz;
  const V3 a = (1.0 / MJS_BLOCK_MASS) * fl + rot(R, cross(sh.c, v3(ax, ay, az)));
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = ax; x[4] = ay; x[5] = az;
}
MJS_DEV void block_mass_matrix6(const M3& R, const Shape& sh, double (*M)[6]) {  // full symmetric matrix
  const double m = MJS_BLOCK_MASS;
  const V3 c = sh.c;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) M[i][j] = 0;
  }
  M[0][0] = M[1][1] = M[2][2] = m;
  const double c2 = dot(c, c);
  M[3][3] = sh.Ixx + m * (c2 - c.x * c.x); M[4][4] = sh.Iyy + m * (c2 - c.y * c.y); M[5][5] = sh.Izz + m * (c2 - c.z * c.z);
  M[4][3] = M[3][4] = -m * c.x * c.y; M[5][3] = M[3][5] = -m * c.x * c.z; M[5][4] = M[4][5] = -m * c.y * c.z;
  // B = -m R C, column d = -m R (C e_d): C e_0 = (0, cz, -cy), C e_1 = (-cz, 0, cx), C e_2 = (cy, -cx, 0)
  const V3 b0 = (-m) * (c.z * R.cy - c.y * R.cz), b1 = (-m) * (c.x * R.cz - c.z * R.cx), b2 = (-m) * (c.y * R.cx - c.x * R.cy);
  M[3][0] = M[0][3] = b0.x; M[3][1] = M[1][3] = b0.y; M[3][2] = M[2][3] = b0.z;
  M[4][0] = M[0][4] = b1.x; M[4][1] = M[1][4] = b1.y; M[4][2] = M[2][4] = b1.z;
  M[5][0] = M[0][5] = b2.x; M[5][1] = M[1][5] = b2.y; M[5][2] = M[2][5] = b2.z;
}

// ---- the decoupled case, lane-parallel -------------------------------------------------------------------------
// Only EPW of the 64 lanes of a wavefront carry an env and FP64 has no lane skipping, so the per-block floor problems
// are dealt to QUADS of lanes: quad (block, env), one floor corner per lane. A lane holds only its corner's four
// contact-frame Jacobian rows (24 values instead of 96: no spills); the 6x6 Hessian, the constraint force, the cost
// and the line-search sums are quad reductions (two DPP quad_perm butterflies: every lane of the quad gets the same
// bits, so the quad's control flow stays uniform); the 6x6 factorisations are replicated. Inputs and results travel
// through the wavefront's LDS workspace.
template <int CTRL>
MJS_DEV double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
MJS_DEV double quad_sum(double x) {
  x += dpp_f64<0xB1>(x);  // quad_perm [1, 0, 3, 2]
  x += dpp_f64<0x4E>(x);  // quad_perm [2, 3, 0, 1]
  return x;
}
MJS_DEV void quad_block_floor(const M3 R, const Shape& sh, const double* qvel, const double* f, bool on, double dist, V3 r, double meaninertia, int nv_total, double* f_out) {
  const double mu[3] = {MJS_BLOCK_FRICTION[0], MJS_BLOCK_FRICTION[0], MJS_BLOCK_FRICTION[1]};
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  double Mb[6][6];
  block_mass_matrix6(R, sh, Mb);
  double Jc[4][6], aref[6];  // this lane's corner: rows normal, t1, t2, torsion (frame of n = +z as in solve_block_floor)
  const V3 axs[3] = {R.cx, R.cy, R.cz};
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const V3 lin = cross(axs[d], r);
    Jc[0][d] = d == 2 ? 1.0 : 0.0; Jc[0][3 + d] = lin.z;
    Jc[1][d] = d == 1 ? 1.0 : 0.0; Jc[1][3 + d] = lin.y;
    Jc[2][d] = d == 0 ? -1.0 : 0.0; Jc[2][3 + d] = -lin.x;
    Jc[3][d] = 0.0; Jc[3][3 + d] = axs[d].z;
  }
  const double imp = impedance_default(dist);
  const double tran = 1.0 / MJS_BLOCK_MASS;
  const double D = 1 / (2 * mu[0] * mu[0] * fmax(MJS_MINVAL, (1 - imp) * (tran + mu[0] * mu[0] * tran) / imp));
  auto edge_values = [&](const double* x, double* out6) {
    double u[4] = {0, 0, 0, 0};
#pragma unroll
    for (int rw = 0; rw < 4; rw++) {
#pragma unroll
      for (int d = 0; d < 6; d++) u[rw] += Jc[rw][d] * x[d];
    }
#pragma unroll
    for (int e = 0; e < 6; e++) out6[e] = u[0] + ((e & 1) ? -mu[e >> 1] : mu[e >> 1]) * u[1 + (e >> 1)];
  };
  {
    double ev[6];
    edge_values(qvel, ev);
#pragma unroll
    for (int e = 0; e < 6; e++) aref[e] = -B * ev[e] - K * imp * dist;
  }
  double a[6], a_s[6], Ma[6], jar[6], force[6];
  bool act[6];
  block_minv(R, sh, f, a_s);
#pragma unroll
  for (int i = 0; i < 6; i++) a[i] = a_s[i];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double m = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) m += (i >= k ? Mb[i][k] : Mb[k][i]) * a[k];
    Ma[i] = m;
  }
  edge_values(a, jar);
#pragma unroll
  for (int e = 0; e < 6; e++) jar[e] -= aref[e];
  auto update = [&]() {
    double cost = 0;
#pragma unroll
    for (int e = 0; e < 6; e++) {
      const bool o = on && jar[e] < 0;
      act[e] = o;
      force[e] = o ? -D * jar[e] : 0.0;
      cost += o ? 0.5 * D * jar[e] * jar[e] : 0.0;
    }
    double gauss = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) gauss += (Ma[i] - f[i]) * (a[i] - a_s[i]);
    return quad_sum(cost) + 0.5 * gauss;
  };
  auto constraint_force = [&](double* fc) {
    const double fn = force[0] + force[1] + force[2] + force[3] + force[4] + force[5];
    const double f1 = mu[0] * (force[0] - force[1]), f2 = mu[1] * (force[2] - force[3]), f3 = mu[2] * (force[4] - force[5]);
#pragma unroll
    for (int i = 0; i < 6; i++) fc[i] = quad_sum(fn * Jc[0][i] + f1 * Jc[1][i] + f2 * Jc[2][i] + f3 * Jc[3][i]);
  };
  double cost = update();
  const double scale = 1 / (meaninertia * nv_total);
#pragma unroll 1
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    double grad[6], search[6], Mv[6], H[6][6], fc[6];
    constraint_force(fc);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      grad[i] = Ma[i] - f[i] - fc[i];
      search[i] = -grad[i];
    }
    {  // this corner's share of J^T D J over its active edges, then the quad sum
      double wn = 0, w[3], ww[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double np_ = act[2 * k], nm = act[2 * k + 1];
        wn += np_ + nm;
        w[k] = D * mu[k] * (np_ - nm);
        ww[k] = D * mu[k] * mu[k] * (np_ + nm);
      }
      wn *= D;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const double jn = Jc[0][i];
        const double rn = wn * jn + w[0] * Jc[1][i] + w[1] * Jc[2][i] + w[2] * Jc[3][i];
        const double r1 = w[0] * jn + ww[0] * Jc[1][i], r2 = w[1] * jn + ww[1] * Jc[2][i], r3 = w[2] * jn + ww[2] * Jc[3][i];
#pragma unroll
        for (int j = 0; j <= i; j++) H[i][j] = Mb[i][j] + quad_sum(rn * Jc[0][j] + r1 * Jc[1][j] + r2 * Jc[2][j] + r3 * Jc[3][j]);
      }
    }
    if (!rr::chol6(H)) break;
    rr::chol6_solve(H, search);
    double g1 = 0, g2 = 0, snorm = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double m = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) m += (i >= k ? Mb[i][k] : Mb[k][i]) * search[k];
      Mv[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { g1 += search[i] * (Ma[i] - f[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
    double jv[6];
    edge_values(search, jv);
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(snorm) / scale;
    double alpha = 0, lo = 0, hi = INFINITY;
#pragma unroll 1
    for (int it = 0; it < 50; it++) {
      double p1 = 0, p2 = 0;
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const double x = jar[e] + alpha * jv[e];
        const bool o = on && x < 0;
        p1 += o ? D * x * jv[e] : 0.0;
        p2 += o ? D * jv[e] * jv[e] : 0.0;
      }
      const double d1 = g1 + alpha * g2 + quad_sum(p1), d2 = g2 + quad_sum(p2);
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
    for (int i = 0; i < 6; i++) { a[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
#pragma unroll
    for (int e = 0; e < 6; e++) jar[e] += alpha * jv[e];
    const double oldcost = cost;
    cost = update();
    constraint_force(fc);
    double gn = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const double g = Ma[i] - f[i] - fc[i];
      gn += g * g;
    }
    if (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE) break;
  }
  double fc[6];
  constraint_force(fc);
#pragma unroll
  for (int i = 0; i < 6; i++) f_out[i] = f[i] + fc[i];
}

struct StepInfo {
  bool bad, rows_active, unsupported;
  bool arm_floor;  // this substep: an arm collision geom or the EEF cylinder is in the floor (set by the detection phase)
  int ncon;
#ifdef MJS_STAMPS
  unsigned long long cyc[16];  // 0..5 phase cycles, 6 = cooperative solves, 7 = their Newton iterations, 8..15 = phases inside them
#endif
};
#ifndef PP_TIC
#ifdef MJS_STAMPS
#define PP_TIC(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_ACC(info, k, t0) do { unsigned long long t1_; PP_TIC(t1_); (info).cyc[k] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define PP_TIC(t) do { } while (0)
#define PP_ACC(info, k, t0) do { } while (0)
#endif
#endif

// The coupled case (an arm-block or block-block contact is active, or a joint is beyond its range): the constraint
// problem couples all nv = 6 + 6 n dofs. It is solved by the WHOLE WAVEFRONT for one env at a time: the env's lane
// publishes M, qfrc_smooth and its rows (<= 78 x 18) in LDS, then the 64 lanes share the dense Newton iteration of
// mj_solPrimal: rows are dealt to lanes (residuals, forces, line-search sums with wave reductions), the Hessian
// entries are dealt to lanes, the 18 x 18 Cholesky and the triangular solves run column by column in LDS.
// The function must be called by all 64 lanes of the workgroup (uniform control flow).
constexpr int LDP = NV + 1;  // padded leading dimension in LDS
constexpr bool COMPACT = NB > 2;
constexpr int COL_NONE = 1 << 20;
// The cooperating lanes are ONE wavefront (several wavefronts share a workgroup only to share the instruction
// cache). A wavefront's LDS instructions execute in program order, so the only things a "sync" has to do are (1) keep
// the COMPILER from moving one lane's LDS load above another lane's earlier store and (2) drain the LDS queue. A
// release/acquire fence pair also waited for every outstanding scratch/global access (s_waitcnt vmcnt(0)) at each of
// the ~60 syncs of a solve; an explicit lgkmcnt wait with a compiler memory barrier does not.
#ifndef MJS_WAVE_SYNC
#define MJS_WAVE_SYNC()                                          \
  do {                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_wave_barrier();                             \
  } while (0)
#endif
struct CoopLds {
  // row MAXROW is the null row (J = 0, D = 0, force = 0): padding of the active-row list.
  // Rows are dense (LDP columns) in the 2-slot instance. In the 5-slot instance they are COMPACT: a row touches at most
  // two bodies, so it stores [6 columns of group 0 | 6 of group 1] and cb = the first dense column of each group
  // (COL_NONE: no such group) - 26 KB instead of 75 KB, which is what lets two wavefronts share a CU.
  double J[MAXROW + 1][COMPACT ? 12 : LDP], D[MAXROW + 1], aref[MAXROW], jar[MAXROW], force[MAXROW + 1];
  int cb[COMPACT ? MAXROW + 1 : 1][2];
  double M[NV][LDP], H[NV][LDP];
  double qs[NV], a[NV], a_s[NV], Ma[NV], search[NV], Mv[NV], fc[NV];
  int arow[MAXROW], nact;  // indices of the active rows (ascending), rebuilt by every constraint update
  int nrow;
  // the sub-system being solved: the bodies with an active arm-block / block-block contact or joint-limit row; its
  // dofs are packed (arm first when present). off[0] = arm offset (0) or -1, off[1 + b] = offset of block b or -1.
  int nv, off[NB + 1];
  double meaninertia;  // mj_setConst's statistic of the owner env's model (depends on its blocks' shapes)
  double jtf[NV];  // J^T force accumulator of the compact-row instance
  // problem description written by the owner lane; the rows are then built by all lanes
  int ncon, c_ba[MAXCON], c_bb[MAXCON], c_act[MAXCON], lim_act[2 * NJ];
  int c_nr[MAXCON], c_nd[MAXCON];  // rows of the contact's pyramid (6: condim 4, block contacts; 4: condim 3, arm-floor) and, for an arm body, the joints that move it
  int overflow;                    // more arm-floor contacts or rows than the workspace holds
  double c_dist[MAXCON], c_tran[MAXCON], c_pos[MAXCON][3], c_n[MAXCON][3];
  double ax[NJ][3], an[NJ][3];     // joint axes and anchors (world)
  double bp[NB][3], bR[NB][9];     // block origins and rotation columns (cx, cy, cz)
  double qvel[NV], q[NJ];
};
// The cooperative workspace lives in dynamic LDS declared at namespace scope so that every device function reaches it
// as an LDS (address space 3) object: passing it by reference through a non-inlined call would degrade every access
// to a FLAT instruction (measured: no ds_* instruction at all in physics_step, all fences waiting on vmcnt).
extern __shared__ double pp_lds_raw[];
MJS_DEV CoopLds& coop_lds() { return reinterpret_cast<CoopLds*>(pp_lds_raw)[threadIdx.x >> 6]; }
static_assert(sizeof(CoopLds) % sizeof(double) == 0, "the hull tables follow the workspaces");
constexpr int HULL_LDS_DOUBLES = MJS_HULL_NCAT * MJS_HULL_MAXV * 3;
MJS_DEV const double* hull_lds() { return pp_lds_raw + (sizeof(CoopLds) * WAVES) / sizeof(double); }
// Per-env substep state: the env's world, servo set-point, joint sines / cosines and step flags live in LDS for the
// length of the substep loop, one slot per env lane (+ one dummy slot shared by a wavefront's helper lanes). The
// substep is a non-inlined function: state handed to it by reference sat in per-lane scratch and every access was a
// FLAT load / store behind L2 (184 + 56 per substep, ~0.7 GB of HBM-side traffic per 4096-env launch).
constexpr bool SPLIT_DETECT = NB <= 2;
struct EnvLds {
  World w;
  double ctrl[NJ], cs[NJ], sn[NJ];
  StepInfo info;
  // results of the substep's detection phase (detect_phase), read by the solve / integrate phase; the 5-slot instance has
  // no LDS left for them (17 slots x 2.6 KB per wavefront) and keeps the substep in one function
  double Rb[SPLIT_DETECT ? NB : 1][9];      // block rotations (columns cx, cy, cz)
  double fl[SPLIT_DETECT ? NB : 1][4][5];   // floor slots: on, dist, r[3]
  double cv[SPLIT_DETECT ? NCVX : 1][8];    // convex pairs: hit, dist, pos[3], n[3]
  int arm_in, blk_in[NB], fln[NB];  // fln: floor contacts detected per block
  double Marm[SPLIT_DETECT ? 21 : 1], qacc[SPLIT_DETECT ? NV : 1];  // arm mass matrix (packed), qfrc_smooth -> qacc of the substep
};
MJS_DEV FloorSlots env_fs(const EnvLds& e, int b) {
  FloorSlots f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double* o = e.fl[b][k];
    f.on[k] = o[0] != 0.0; f.dist[k] = o[1]; f.r[k] = v3(o[2], o[3], o[4]);
  }
  f.n = e.fln[b];
  return f;
}
MJS_DEV ConvexHits env_cvx(const EnvLds& e) {
  ConvexHits c;
#pragma unroll
  for (int k = 0; k < NCVX; k++) {
    const double* o = e.cv[k];
    c.hit[k] = o[0] != 0.0; c.dist[k] = o[1]; c.pos[k] = v3(o[2], o[3], o[4]); c.n[k] = v3(o[5], o[6], o[7]);
  }
  return c;
}
MJS_DEV M3 env_Rb(const EnvLds& e, int b) { return M3{v3(e.Rb[b][0], e.Rb[b][1], e.Rb[b][2]), v3(e.Rb[b][3], e.Rb[b][4], e.Rb[b][5]), v3(e.Rb[b][6], e.Rb[b][7], e.Rb[b][8])}; }
constexpr size_t ENV_LDS_OFFSET = sizeof(CoopLds) * WAVES + sizeof(double) * HULL_LDS_DOUBLES;
constexpr size_t LDS_BYTES = ENV_LDS_OFFSET + sizeof(EnvLds) * WAVES * (EPW + 1);  // dynamic LDS of a launch
MJS_DEV EnvLds& env_lds() {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return reinterpret_cast<EnvLds*>(pp_lds_raw + ENV_LDS_OFFSET / sizeof(double))[wave * (EPW + 1) + (lane < EPW ? lane : EPW)];
}
static_assert(sizeof(EnvLds) % sizeof(double) == 0, "slots are arrays of doubles");
// exchange area of the lane-parallel decoupled solves (quad_block_floor); shares the wavefront's workspace with the
// cooperative solver, which runs after it
struct QuadIn {
  double R[9], qv[6], f[6], dist[4], r[4][3], shape, scale, meaninertia;
  int on[4], need;
};
struct QuadLds {
  QuadIn in[EPW][NB];
  double out[EPW][NB][6];
};
static_assert(sizeof(QuadLds) <= sizeof(CoopLds), "the exchange area aliases the cooperative workspace");
MJS_DEV QuadLds& quad_lds() { return *reinterpret_cast<QuadLds*>(&coop_lds()); }
// exchange area of the group-parallel convex pairs (before the quads use the workspace): the env lanes publish their geoms
// (wrist proxy, EEF cylinder, blocks), every env group evaluates the env's pairs, the env lanes collect the hits
constexpr int GEOM_DOUBLES = 17;  // c[3], R[9], s[3], box, cat
struct DetLds {
  double g[EPW][NB + 2][GEOM_DOUBLES];
  double org[EPW][NB][3];          // block body origins
  double out[EPW][NCVX][8];        // hit, dist, pos[3], n[3]
  double fl[EPW][NB][4][5];        // floor slots: on, dist, r[3]
  int fln[EPW][NB];                // floor slots found per block
  int live[EPW];
};
static_assert(sizeof(DetLds) <= sizeof(CoopLds), "the exchange area aliases the cooperative workspace");
MJS_DEV DetLds& det_lds() { return *reinterpret_cast<DetLds*>(&coop_lds()); }
MJS_DEV void put_geom(double* d, const Geom& g) {
  d[0] = g.c.x; d[1] = g.c.y; d[2] = g.c.z;
  d[3] = g.R.cx.x; d[4] = g.R.cx.y; d[5] = g.R.cx.z; d[6] = g.R.cy.x; d[7] = g.R.cy.y; d[8] = g.R.cy.z; d[9] = g.R.cz.x; d[10] = g.R.cz.y; d[11] = g.R.cz.z;
  d[12] = g.s.x; d[13] = g.s.y; d[14] = g.s.z; d[15] = g.box ? 1.0 : 0.0; d[16] = (double)g.cat;
}
MJS_DEV Geom get_geom(const double* d) {
  Geom g;
  g.c = v3(d[0], d[1], d[2]);
  g.R = M3{v3(d[3], d[4], d[5]), v3(d[6], d[7], d[8]), v3(d[9], d[10], d[11])};
  g.s = v3(d[12], d[13], d[14]);
  g.box = d[15] != 0.0;
  g.cat = (int)d[16];
  return g;
}
// wave broadcast of a double from a compile-time lane
MJS_DEV double bcast(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
// x + (x moved by the DPP control, 0 where the source lane does not exist or the row / bank mask disables the lane)
template <int CTRL, int ROW_MASK>
MJS_DEV double dpp_add(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, false);
  return x + __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, the same bits in every lane: inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8),
// row totals forwarded with row_bcast15 / row_bcast31, the total read from lane 63. Register-only: the ds_bpermute
// butterfly (__shfl_xor) it replaces cost a round trip through the LDS crossbar per stage.
MJS_DEV double wave_sum(double x) {
  x = dpp_add<0x111, 0xF>(x);
  x = dpp_add<0x112, 0xF>(x);
  x = dpp_add<0x114, 0xF>(x);
  x = dpp_add<0x118, 0xF>(x);
  x = dpp_add<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  x = dpp_add<0x143, 0xC>(x);  // row_bcast31 into rows 2 and 3
  return bcast(x, 63);
}
// Solve H x = b (H = sh.H, lower triangle, SPD; b and x in `vec`, LDS) with the factorisation in REGISTERS: lane i owns
// row i of H; the pivot, the column-j entries of the other rows and the right-hand side travel by v_readlane
// broadcasts (all lane indices are compile-time constants of the unrolled loops), so the 153 trailing updates and
// the forward substitution touch no memory. The rows of L then go to LDS once and every lane fetches its column of
// L for the backward substitution. Rows >= nv are identity padding (one block instead of two).
template <int NVT>
MJS_DEV bool coop_chol_solve_n(CoopLds& sh, double* vec, int lane) {
  constexpr int NV = NVT;  // the unrolled loops below run over the sub-system's size
  const int nv = NVT;
  double row[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) row[j] = (lane < nv && j <= lane) ? sh.H[lane < NV ? lane : 0][j] : (j == lane ? 1.0 : 0.0);
  double b = lane < nv ? vec[lane < NV ? lane : 0] : 0.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const double d = bcast(row[j], j);
    if (d < MJS_MINVAL) ok = false;
    const double inv = rr::rsqrt_fast(fmax(d, MJS_MINVAL));
    row[j] = lane == j ? inv : row[j] * inv;  // L_ij for the rows below; the pivot row keeps 1 / L_jj
    const double yj = bcast(b, j) * inv;      // forward substitution fused in
    b = lane == j ? yj : (lane > j ? b - row[j] * yj : b);
#pragma unroll
    for (int k = j + 1; k < NV; k++) {
      const double lkj = bcast(row[j], k);
      row[k] -= row[j] * lkj;  // entries above the diagonal (lane < k) are never read
    }
  }
  // rows of L to LDS, then each lane reads its column: L[k][lane], k > lane
  if (lane < NV) {
#pragma unroll
    for (int j = 0; j < NV; j++)
      if (j <= lane) sh.H[lane][j] = row[j];
  }
  MJS_WAVE_SYNC();
  double col[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) col[k] = (lane < NV && k > lane) ? sh.H[k][lane < NV ? lane : 0] : 0.0;
#pragma unroll
  for (int k = NV - 1; k >= 0; k--) {
    const double xk = bcast(b, k) * bcast(row[k], k);  // y_k / L_kk
    b = lane == k ? xk : (lane < k ? b - col[k] * xk : b);
  }
  if (lane < nv) vec[lane] = b;
  MJS_WAVE_SYNC();
  return ok;
}
template <int NVT>
MJS_DEV bool coop_chol_dispatch(CoopLds& sh, int nv, double* vec, int lane) {  // nv is a multiple of 6, wave-uniform
  if (nv == NVT) return coop_chol_solve_n<NVT>(sh, vec, lane);
  if constexpr (NVT > 6) return coop_chol_dispatch<NVT - 6>(sh, nv, vec, lane);
  return false;
}
MJS_DEV bool coop_chol_solve(CoopLds& sh, int nv, double* vec, int lane) { return coop_chol_dispatch<NV>(sh, nv, vec, lane); }
// cooperative mj_solPrimal on the problem published in sh; result sh.fc = J^T force
constexpr int NCH = (MAXROW + 63) / 64;  // rows per lane when the rows are dealt to the 64 lanes
// Out of line ON PURPOSE: inlined into the substep its loops inherit the caller's register pressure (the env lane's world
// state, contact slots and mass matrices are live across the call) and reload spilled values from scratch inside the
// row loops; as a function it is allocated on its own. The workspace is reached through coop_lds() (address space 3).
__device__ __noinline__ int coop_newton(int nv, double scale, int lane, StepInfo& info) {
  CoopLds& sh = coop_lds();
  const int nrow = sh.nrow;
  unsigned long long tn = 0;
  PP_TIC(tn);
  // qacc_smooth = M^-1 qfrc_smooth: M is block diagonal (arm, blocks), one lane per 6x6 block
  if (lane < nv / 6) {
    double L[6][6], x[6];
    const int o = 6 * lane;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) L[i][j] = sh.M[o + i][o + j];
      x[i] = sh.qs[o + i];
    }
    rr::chol6(L);
    rr::chol6_solve(L, x);
#pragma unroll
    for (int i = 0; i < 6; i++) { sh.a_s[o + i] = x[i]; sh.a[o + i] = x[i]; }
  }
  MJS_WAVE_SYNC();
  const int col = lane < nv ? lane : 0;  // lanes >= nv shadow column 0 (uniform control flow), their results are dropped
  auto m_times = [&](const double* x) {  // (M x)[col]; nv is a multiple of 6
    double m = 0;
    for (int k = 0; k < nv; k += 6) {
#pragma unroll
      for (int u = 0; u < 6; u++) m += sh.M[col][k + u] * x[k + u];
    }
    return m;
  };
  auto j_times = [&](int r, const double* x) {  // (J x)[r]
    double y = 0;
    if constexpr (!COMPACT) {
      for (int k = 0; k < nv; k += 6) {
#pragma unroll
        for (int u = 0; u < 6; u++) y += sh.J[r][k + u] * x[k + u];
      }
    } else {
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const int c0 = sh.cb[r][g];
        if (c0 == COL_NONE) continue;
#pragma unroll
        for (int u = 0; u < 6; u++) y += sh.J[r][6 * g + u] * x[c0 + u];
      }
    }
    return y;
  };
  {
    const double m = m_times(sh.a);
    if (lane < nv) sh.Ma[lane] = m;
  }
#pragma unroll
  for (int q = 0; q < NCH; q++) {
    const int r = lane + 64 * q;
    if (r < nrow) sh.jar[r] = j_times(r, sh.a) - sh.aref[r];
  }
  MJS_WAVE_SYNC();
  // forces and the active set from jar: the active rows are compacted into sh.arow (ballot + prefix count) so that the
  // sums over active rows below run over a dense list with their LDS reads in flight together. Returns the cost.
  // The list then lives in registers (entry k in lane k & 63 of myrow[k >> 6], padded with the null row): a sum over
  // active rows fetches its row indices with v_readlane instead of a dependent LDS read.
  int nact = 0, myrow[NCH];
  auto update = [&]() {
    double cost = 0;
    int base = 0;
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int r = lane + 64 * q;
      const bool in = r < nrow;
      const double x = in ? sh.jar[r] : 0.0, d = in ? sh.D[r] : 0.0;
      const bool act = in && x < 0;
      if (in) sh.force[r] = act ? -d * x : 0.0;
      cost += act ? 0.5 * d * x * x : 0.0;
      const unsigned long long m = __ballot(act);
      if (act) sh.arow[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
      base += __popcll(m);
    }
    nact = base;
    if (lane < nv) cost += 0.5 * (sh.Ma[lane] - sh.qs[lane]) * (sh.a[lane] - sh.a_s[lane]);
    MJS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < NCH; q++) myrow[q] = lane + 64 * q < nact ? sh.arow[lane + 64 * q] : MAXROW;
    return wave_sum(cost);
  };
  auto jt_force = [&]() {  // (J^T force)[col] over the active rows
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if constexpr (!COMPACT) {
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const int nq = min(nact - 64 * q, 64);
        for (int k = 0; k < nq; k += 4) {  // lanes past the list hold the null row
          const int r0 = __builtin_amdgcn_readlane(myrow[q], k), r1 = __builtin_amdgcn_readlane(myrow[q], (k + 1) & 63);
          const int r2 = __builtin_amdgcn_readlane(myrow[q], (k + 2) & 63), r3 = __builtin_amdgcn_readlane(myrow[q], (k + 3) & 63);
          s0 += sh.J[r0][col] * sh.force[r0]; s1 += sh.J[r1][col] * sh.force[r1];
          s2 += sh.J[r2][col] * sh.force[r2]; s3 += sh.J[r3][col] * sh.force[r3];
        }
      }
      return (s0 + s1) + (s2 + s3);
    } else {
      // compact rows, row-major: every active row scatters force * J into the <= 12 columns it touches (ds_add_f64,
      // rows one after the other = deterministic order); lanes 0..11 of a trip serve one row, 5 rows per trip
      if (lane < NV) sh.jtf[lane] = 0;
      MJS_WAVE_SYNC();
      const int slot = lane / 12, u = lane - 12 * slot;  // 5 rows x 12 columns per trip (lanes 60..63 idle)
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const int nq = min(nact - 64 * q, 64);
        for (int k = 0; k < nq; k += 5) {
          const int kk = k + slot;
          const int r = __shfl(myrow[q], kk < 64 ? kk : 63);
          if (slot < 5 && kk < nq) {
            const int c0 = sh.cb[r][u < 6 ? 0 : 1];
            if (c0 != COL_NONE) atomicAdd(&sh.jtf[c0 + (u < 6 ? u : u - 6)], sh.J[r][u] * sh.force[r]);
          }
        }
      }
      MJS_WAVE_SYNC();
      return sh.jtf[col];
    }
  };
  double cost = update();
  PP_ACC(info, 8, tn);
  // this lane's (up to NHE) entries of the lower triangle of the Hessian
  constexpr int NHE = (NV * (NV + 1) / 2 + 63) / 64;
  int hi[NHE], hj[NHE];
  bool he[NHE];
#pragma unroll
  for (int q = 0; q < NHE; q++) {
    const int e = lane + 64 * q;
    he[q] = e < nv * (nv + 1) / 2;
    int i = (int)((sqrt(8.0 * (he[q] ? e : 0) + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= (he[q] ? e : 0)) i++;
    while (i * (i + 1) / 2 > (he[q] ? e : 0)) i--;
    hi[q] = i;
    hj[q] = (he[q] ? e : 0) - i * (i + 1) / 2;
  }
  {  // search = -gradient
    const double g = sh.Ma[col] - sh.qs[col] - jt_force();
    if (lane < nv) sh.search[lane] = -g;
  }
  int iters = 0;
  for (int iter = 0; iter < MJS_SOLVER_ITERATIONS; iter++) {
    iters++;
    if constexpr (!COMPACT) {
      // Hessian = M + J^T diag(D active) J on the matrix cores: v_mfma_f64_16x16x4_f64 takes A[i][k] = J[r_k][i] and
      // B[k][j] = D[r_k] J[r_k][j] for four active rows r_k per instruction (lane l: i = j = l & 15, k = l >> 4), so a lane
      // reads ONE J value and one D per four rows - the entry-major VALU version read 14 values per two rows and was
      // bound by LDS latency (7.4 k cycles per Newton iteration at 12 dofs). nv <= 16: one 16x16 tile; nv = 18: the
      // three lower tiles of a 32x32 product. The FP64 MFMA rate equals the vector rate on gfx950: the gain is operand
      // sharing, not FLOPs. C/D layout (guide): col = lane & 15, row = (lane >> 4) + 4 * reg.
      typedef double d4 __attribute__((ext_vector_type(4)));
      d4 c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
      const int li = lane & 15, lk = lane >> 4;
      const bool two = nv > 16;  // wave-uniform
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int nq = min(nact - 64 * c, 64);
        for (int k = 0; k < nq; k += 4) {  // lanes past the list hold the null row (J = 0, D = 0)
          const int r0 = __builtin_amdgcn_readlane(myrow[c], k), r1 = __builtin_amdgcn_readlane(myrow[c], (k + 1) & 63);
          const int r2 = __builtin_amdgcn_readlane(myrow[c], (k + 2) & 63), r3 = __builtin_amdgcn_readlane(myrow[c], (k + 3) & 63);
          const int r = lk == 0 ? r0 : lk == 1 ? r1 : lk == 2 ? r2 : r3;
          const double d = sh.D[r];
          const double a0 = li < nv ? sh.J[r][li] : 0.0;
          c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d * a0, c00, 0, 0, 0);
          if (two) {
            const double a1 = 16 + li < nv ? sh.J[r][16 + li] : 0.0;
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d * a0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d * a1, c11, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = lk + 4 * g;
        if (row < nv && li <= row) sh.H[row][li] = sh.M[row][li] + c00[g];
        if (two) {
          if (16 + row < nv) sh.H[16 + row][li] = sh.M[16 + row][li] + c10[g];
          if (16 + row < nv && li <= row) sh.H[16 + row][16 + li] = sh.M[16 + row][16 + li] + c11[g];
        }
      }
    } else {
      // row-major for the compact rows: a row touches <= 12 columns, i.e. <= 78 of the up to 666 entries. H starts as M;
      // every active row then adds D J_a J_b for its local column pairs (a >= b, one or two pairs per lane) with LDS
      // ds_add_f64. Rows are processed one after the other (LDS operations of a wavefront execute in order), so every
      // entry sees its contributions in row order: deterministic.
#pragma unroll
      for (int q = 0; q < NHE; q++)
        if (he[q]) sh.H[hi[q]][hj[q]] = sh.M[hi[q]][hj[q]];
      int pa[2], pb[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int e = min(lane + 64 * q, 77);
        int a = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= e) a++;
        while (a * (a + 1) / 2 > e) a--;
        pa[q] = a; pb[q] = e - a * (a + 1) / 2;
      }
      MJS_WAVE_SYNC();
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int nq = min(nact - 64 * c, 64);
        for (int k = 0; k < nq; k++) {
          const int r = __builtin_amdgcn_readlane(myrow[c], k);
          const double d = sh.D[r];
          const int c0 = sh.cb[r][0], c1 = sh.cb[r][1];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            if (q == 1 && lane >= 78 - 64) continue;
            const int ca = pa[q] < 6 ? c0 : c1, cbq = pb[q] < 6 ? c0 : c1;
            if (ca == COL_NONE || cbq == COL_NONE) continue;  // wave-uniform per pair class only when both groups exist; per-lane otherwise
            const int ia = ca + (pa[q] < 6 ? pa[q] : pa[q] - 6), ib = cbq + (pb[q] < 6 ? pb[q] : pb[q] - 6);
            const double v = sh.J[r][pa[q]] * d * sh.J[r][pb[q]];
            atomicAdd(&sh.H[ia > ib ? ia : ib][ia > ib ? ib : ia], v);
          }
        }
      }
    }
    MJS_WAVE_SYNC();
    PP_ACC(info, 9, tn);
    if (!coop_chol_solve(sh, nv, sh.search, lane)) break;
    PP_ACC(info, 10, tn);
    {
      const double m = m_times(sh.search);
      if (lane < nv) sh.Mv[lane] = m;
    }
    double jr[NCH], jvr[NCH], dr[NCH];  // this lane's rows, kept for the line search
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int r = lane + 64 * q;
      const bool in = r < nrow;
      jvr[q] = in ? j_times(r, sh.search) : 0.0;
      jr[q] = in ? sh.jar[r] : 1.0;  // rows beyond nrow: never active
      dr[q] = in ? sh.D[r] : 0.0;
    }
    MJS_WAVE_SYNC();
    PP_ACC(info, 11, tn);
    double g1 = 0, g2 = 0, sn2 = 0;
    if (lane < nv) { g1 = sh.search[lane] * (sh.Ma[lane] - sh.qs[lane]); g2 = sh.search[lane] * sh.Mv[lane]; sn2 = sh.search[lane] * sh.search[lane]; }
    g1 = wave_sum(g1); g2 = wave_sum(g2); sn2 = wave_sum(sn2);
    if (sqrt(sn2) < MJS_MINVAL) break;
    const double gtol = MJS_SOLVER_TOLERANCE * 0.01 * sqrt(sn2) / scale;
    double alpha = 0, lo = 0, hi_ = INFINITY;
    for (int it = 0; it < 50; it++) {  // exact 1-D Newton; every lane follows the same alpha sequence
      double p1 = 0, p2 = 0;
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const double x = jr[q] + alpha * jvr[q];
        const bool o = x < 0;
        p1 += o ? dr[q] * x * jvr[q] : 0.0;
        p2 += o ? dr[q] * jvr[q] * jvr[q] : 0.0;
      }
      const double d1 = g1 + alpha * g2 + wave_sum(p1), d2 = g2 + wave_sum(p2);
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi_ = alpha;
      if (d2 <= 0) break;
      double next = alpha + (-d1 / d2);
      if (!(next > lo && next < hi_)) next = isfinite(hi_) ? 0.5 * (lo + hi_) : (alpha > 0 ? 2 * alpha : 1.0);
      if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
      alpha = next;
    }
    PP_ACC(info, 12, tn);
    if (alpha == 0) break;
    if (lane < nv) { sh.a[lane] += alpha * sh.search[lane]; sh.Ma[lane] += alpha * sh.Mv[lane]; }
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int r = lane + 64 * q;
      if (r < nrow) sh.jar[r] = jr[q] + alpha * jvr[q];
    }
    MJS_WAVE_SYNC();
    const double oldcost = cost;
    cost = update();
    // gradient at the new point: its norm for the stopping rule, its negative as the next right-hand side
    const double g = sh.Ma[col] - sh.qs[col] - jt_force();
    const double gn = wave_sum(lane < nv ? g * g : 0.0);
    if (lane < nv) sh.search[lane] = -g;
    PP_ACC(info, 13, tn);
    if (scale * (oldcost - cost) < MJS_SOLVER_TOLERANCE || scale * sqrt(gn) < MJS_SOLVER_TOLERANCE) break;
  }
  MJS_WAVE_SYNC();
  {
    const double f = jt_force();
    if (lane < nv) sh.fc[lane] = f;
  }
  MJS_WAVE_SYNC();
  PP_ACC(info, 14, tn);
  return iters;
}
// the lane that owns the env describes its problem in LDS: contacts, kinematics, mass matrix blocks, forces
MJS_DEV void publish_problem(const World& s, const double* cs, const double* sn, const double* Marm, const double* qs_arm, int nb,
                             const FloorSlots* fs, const ConvexHits& cvx, const M3* Rb, bool arm_in, const bool* blk_in, double meaninertia, bool arm_floor) {
  CoopLds& sh = coop_lds();
  sh.meaninertia = meaninertia;
  int off[NB + 1], nvs = arm_in ? NJ : 0;
  off[0] = arm_in ? 0 : -1;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    off[1 + b] = (b < nb && blk_in[b]) ? nvs : -1;
    if (b < nb && blk_in[b]) nvs += 6;
  }
#pragma unroll
  for (int k = 0; k <= NB; k++) sh.off[k] = off[k];
  sh.nv = nvs;
  // contact list in MuJoCo's pair order from the hot path's static slots (no second collision pass)
  int ncon = 0;
  sh.overflow = 0;
  auto put = [&](double dist, V3 pos, V3 n, int ba, int bb, double tran, int nr = 6, int nd = NJ) {
    sh.c_ba[ncon] = ba; sh.c_bb[ncon] = bb; sh.c_act[ncon] = dist < 0.0;
    sh.c_nr[ncon] = nr; sh.c_nd[ncon] = nd;
    sh.c_dist[ncon] = dist; sh.c_tran[ncon] = tran;
    sh.c_pos[ncon][0] = pos.x; sh.c_pos[ncon][1] = pos.y; sh.c_pos[ncon][2] = pos.z;
    sh.c_n[ncon][0] = n.x; sh.c_n[ncon][1] = n.y; sh.c_n[ncon][2] = n.z;
    ncon++;
  };
  // The arm's own geoms / the EEF cylinder in the floor while the arm is coupled to a block (the EEF drags over the floor and pushes):
  // floor (geom 0) vs arm geoms come first in mj_collision's pair order; condim 3 (4 pyramid rows), body b of the arm is moved by
  // joints 0 .. b-1. Same detection and order as the general stage (rr::gen_stage + ScenePush::extra_contacts), which solves the
  // arm alone when no block is coupled to it. Only ACTIVE contacts take a slot (a listed contact at dist == 0 makes no rows).
  if (arm_in && arm_floor) {
    rr::Chain chf;
    rr::fk_cs(cs, sn, chf);
    int naf = 0;
    auto emit_af = [&](int ndof, V3 pos, double dist, double invw) {
      if (!(dist < 0.0)) return;
      if (naf >= MAXAF) { sh.overflow = 1; return; }
      put(dist, pos, v3(0, 0, 1), 0, 1, invw, 4, ndof);
      naf++;
    };
    rr::arm_floor_contacts(chf, [&](int b, V3 pos, double dist) { emit_af(b, pos, dist, fmax(MJS_MINVAL, UR5E_PP_LINK_BODY_INVWEIGHT0[b])); });
    const Geom eg = eef_geom(chf);
    rr::plane_cylinder_contacts(eg.c, eg.R.cz, eg.R.cx, eg.s.x, eg.s.y, [&](V3 pos, double dist) { emit_af(NJ, pos, dist, UR5E_PP_EEF_BODY_INVWEIGHT0[0]); });
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b >= nb || !blk_in[b]) continue;
    const FloorSlots& f = fs[b];
#pragma unroll
    for (int k = 0; k < 4; k++)  // only ACTIVE corners make rows (a corner exactly on the floor is detected but inactive)
      if (f.on[k]) put(f.dist[k], f.r[k] + s.b[b].p, v3(0, 0, 1), 0, 2 + b, 1.0 / MJS_BLOCK_MASS);
  }
#pragma unroll
  for (int b = 0; b < NB; b++)
    if (b < nb && blk_in[b] && cvx.hit[b]) put(cvx.dist[b], cvx.pos[b], cvx.n[b], 1, 2 + b, UR5E_PP_WRIST3_BODY_INVWEIGHT0[0] + 1.0 / MJS_BLOCK_MASS);
#pragma unroll
  for (int b = 0; b < NB; b++)
    if (b < nb && blk_in[b] && cvx.hit[NB + b]) put(cvx.dist[NB + b], cvx.pos[NB + b], cvx.n[NB + b], 1, 2 + b, UR5E_PP_EEF_BODY_INVWEIGHT0[0] + 1.0 / MJS_BLOCK_MASS);
#pragma unroll
  for (int a = 0; a < NB; a++) {
#pragma unroll
    for (int b = a + 1; b < NB; b++) {
      const int k = pair_slot(a, b);
      if (b < nb && blk_in[a] && blk_in[b] && cvx.hit[k]) put(cvx.dist[k], cvx.pos[k], cvx.n[k], 2 + a, 2 + b, 2.0 / MJS_BLOCK_MASS);
    }
  }
  sh.ncon = ncon;
  for (int k = 0; k < 2 * NJ; k++) sh.lim_act[k] = 0;
  if (arm_in) {
  rr::Chain ch;
  rr::fk_cs(cs, sn, ch);
  for (int j = 0; j < NJ; j++) {
    const V3 a = rr::joint_axis(ch, j), p = ch.p[j + 1];
    sh.ax[j][0] = a.x; sh.ax[j][1] = a.y; sh.ax[j][2] = a.z;
    sh.an[j][0] = p.x; sh.an[j][1] = p.y; sh.an[j][2] = p.z;
    sh.q[j] = s.q[j];
    sh.qvel[j] = s.v[j];
    sh.qs[j] = qs_arm[j];
    sh.lim_act[2 * j] = s.q[j] - MJS_UR_JNT_RANGE[j][0] < 0.0;
    sh.lim_act[2 * j + 1] = MJS_UR_JNT_RANGE[j][1] - s.q[j] < 0.0;
    for (int k = 0; k <= j; k++) sh.M[j][k] = sh.M[k][j] = Marm[j * (j + 1) / 2 + k];
    sh.M[j][j] += MJS_UR_ARMATURE;
  }
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (off[1 + b] < 0) continue;
    const int o = off[1 + b];
    double Mb[6][6], f[6];
    const Shape shp = shape_of(s.b[b]);
    block_mass_matrix6(Rb[b], shp, Mb);
    block_smooth_force(Rb[b], s.b[b].w, shp, f);
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) sh.M[o + i][o + j] = Mb[i][j];
      sh.qs[o + i] = f[i];
    }
    sh.qvel[o] = s.b[b].v.x; sh.qvel[o + 1] = s.b[b].v.y; sh.qvel[o + 2] = s.b[b].v.z;
    sh.qvel[o + 3] = s.b[b].w.x; sh.qvel[o + 4] = s.b[b].w.y; sh.qvel[o + 5] = s.b[b].w.z;
    sh.bp[b][0] = s.b[b].p.x; sh.bp[b][1] = s.b[b].p.y; sh.bp[b][2] = s.b[b].p.z;
    sh.bR[b][0] = Rb[b].cx.x; sh.bR[b][1] = Rb[b].cx.y; sh.bR[b][2] = Rb[b].cx.z;
    sh.bR[b][3] = Rb[b].cy.x; sh.bR[b][4] = Rb[b].cy.y; sh.bR[b][5] = Rb[b].cy.z;
    sh.bR[b][6] = Rb[b].cz.x; sh.bR[b][7] = Rb[b].cz.y; sh.bR[b][8] = Rb[b].cz.z;
  }
}
// all lanes: limit rows + pyramid rows (6 per active contact, condim 4) in the oracle's order
MJS_DEV void coop_build_rows(CoopLds& sh, int lane) {
  const int nv = sh.nv;
  const bool arm_in = sh.off[0] >= 0;
  const double tc = fmax(MJS_SOLREF_TIMECONST, 2 * MJS_RR_PHYSICS_DT), dmax = MJS_SOLIMP_DWIDTH;
  const double K = 1 / fmax(MJS_MINVAL, dmax * dmax * tc * tc * MJS_SOLREF_DAMPRATIO * MJS_SOLREF_DAMPRATIO);
  const double B = 2 / fmax(MJS_MINVAL, dmax * tc);
  int nlim = 0;
  for (int k = 0; k < 2 * NJ; k++) nlim += sh.lim_act[k];
  if (lane < 2 * NJ && sh.lim_act[lane]) {
    int row = 0;
    for (int k = 0; k < lane; k++) row += sh.lim_act[k];
    const int j = lane >> 1;
    const double sgn = (lane & 1) ? -1.0 : 1.0;
    const double dist = (lane & 1) ? MJS_UR_JNT_RANGE[j][1] - sh.q[j] : sh.q[j] - MJS_UR_JNT_RANGE[j][0];
    if constexpr (!COMPACT) {
      for (int d = 0; d < nv; d++) sh.J[row][d] = d == j ? sgn : 0.0;
    } else {
      for (int d = 0; d < 12; d++) sh.J[row][d] = d == j ? sgn : 0.0;
      sh.cb[row][0] = 0; sh.cb[row][1] = COL_NONE;
    }
    const double imp = impedance_default(dist);
    sh.D[row] = 1 / fmax(MJS_MINVAL, (1 - imp) * UR5E_PP_DOF_INVWEIGHT0[j] / imp);
    sh.aref[row] = -B * (sgn * sh.qvel[j]) - K * imp * dist;
  }
  int ncrow = 0;  // contact rows in all
  for (int c = 0; c < sh.ncon; c++) ncrow += sh.c_act[c] ? sh.c_nr[c] : 0;
  const bool rows_fit = nlim + ncrow <= MAXROW;  // (always, in the 2-slot instance)
  for (int t = lane; t < 6 * sh.ncon && rows_fit; t += 64) {  // task = (contact, pyramid edge)
    const int c = t / 6, e = t - 6 * c;
    if (!sh.c_act[c] || e >= sh.c_nr[c]) continue;
    int before = 0;
    for (int k = 0; k < c; k++) before += sh.c_act[k] ? sh.c_nr[k] : 0;
    const int row = nlim + before + e;
    const V3 n = v3(sh.c_n[c][0], sh.c_n[c][1], sh.c_n[c][2]);
    V3 t1, t2;
    {  // mju_makeFrame
      V3 y = (n.y > -0.5 && n.y < 0.5) ? v3(0, 1, 0) : v3(0, 0, 1);
      y = madd(y, -dot(n, y), n);
      t1 = (1.0 / sqrt(dot(y, y))) * y;
      t2 = cross(n, t1);
    }
    const int kk = e >> 1;  // 0: t1, 1: t2, 2: torsion
    const double sgn = (e & 1) ? -1.0 : 1.0;
    const int ba = sh.c_ba[c], bb = sh.c_bb[c];
    const bool blocks_only = ba >= 2 && bb >= 2, arm_floor = bb == 1;
    const double slide = arm_floor ? MJS_GEOM_FRICTION_SLIDE : MJS_BLOCK_FRICTION[0];
    const double fri[3] = {slide, slide, fmax(MJS_BLOCK_FRICTION[1], blocks_only ? 0.0 : MJS_GEOM_FRICTION_SPIN)};
    // row = Jn + sgn mu Jk with Jn / Jk the normal and the kk-th frame row of (body b - body a), written column by
    // column straight into LDS (static column indices: nothing lives in indexed scratch). The arm is body a of an arm-block
    // contact and body b of a floor-arm contact (bb == 1: columns of the joints that move the touching link only).
    const V3 pos = v3(sh.c_pos[c][0], sh.c_pos[c][1], sh.c_pos[c][2]);
    const V3 Fk = kk == 0 ? t1 : kk == 1 ? t2 : n;
    const bool rotk = kk == 2;
    const double mu = sgn * fri[kk];
    double vel = 0;
    const double nd[3] = {n.x, n.y, n.z}, fd[3] = {Fk.x, Fk.y, Fk.z};
    // the six columns of block b (sign sb) at row offset `base`, dense offset o
    auto block_cols = [&](int b, double sb, int base, int o) {
      const V3 rvec = pos - v3(sh.bp[b][0], sh.bp[b][1], sh.bp[b][2]);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const double v = sb * nd[d] + mu * (rotk ? 0.0 : sb * fd[d]);
        sh.J[row][base + d] = v;
        vel += v * sh.qvel[o + d];
      }
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const V3 axs = v3(sh.bR[b][3 * d], sh.bR[b][3 * d + 1], sh.bR[b][3 * d + 2]);
        const V3 lin = cross(axs, rvec);
        const double v = sb * dot(n, lin) + mu * (sb * dot(Fk, rotk ? axs : lin));
        sh.J[row][base + 3 + d] = v;
        vel += v * sh.qvel[o + 3 + d];
      }
    };
    const int nmov = sh.c_nd[c];
    auto arm_cols = [&](double sa, int base) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const V3 ax = v3(sh.ax[j][0], sh.ax[j][1], sh.ax[j][2]);
        const V3 lin = cross(ax, pos - v3(sh.an[j][0], sh.an[j][1], sh.an[j][2]));
        const double v = j < nmov ? sa * dot(n, lin) + mu * (sa * dot(Fk, rotk ? ax : lin)) : 0.0;
        sh.J[row][base + j] = v;
        vel += v * sh.qvel[j];
      }
    };
    if constexpr (!COMPACT) {
      if (arm_in) arm_cols(ba == 1 ? -1.0 : arm_floor ? 1.0 : 0.0, 0);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int o = sh.off[1 + b];
        if (o < 0) continue;
        block_cols(b, (bb == 2 + b ? 1.0 : 0.0) - (ba == 2 + b ? 1.0 : 0.0), o, o);
      }
    } else {
      // group 0 = body a (arm, a block or the world), group 1 = body b (a block, or the arm for a floor-arm contact)
      if (ba == 1) { arm_cols(-1.0, 0); sh.cb[row][0] = 0; }
      else if (ba >= 2) { block_cols(ba - 2, -1.0, 0, sh.off[1 + ba - 2]); sh.cb[row][0] = sh.off[1 + ba - 2]; }
      else {
#pragma unroll
        for (int d = 0; d < 6; d++) sh.J[row][d] = 0.0;
        sh.cb[row][0] = COL_NONE;
      }
      if (arm_floor) { arm_cols(1.0, 6); sh.cb[row][1] = 0; }
      else {
        block_cols(bb - 2, 1.0, 6, sh.off[1 + bb - 2]);
        sh.cb[row][1] = sh.off[1 + bb - 2];
      }
    }
    const double imp = impedance_default(sh.c_dist[c]);
    const double R0 = fmax(MJS_MINVAL, (1 - imp) * (sh.c_tran[c] + fri[0] * fri[0] * sh.c_tran[c]) / imp);
    sh.D[row] = 1 / (2 * fri[0] * fri[0] * R0);
    sh.aref[row] = -B * vel - K * imp * sh.c_dist[c];
  }
  if (lane == 0) {
    sh.nrow = rows_fit ? nlim + ncrow : 0;
    if (!rows_fit) sh.overflow = 1;
    sh.D[MAXROW] = 0; sh.force[MAXROW] = 0;
  }
  if (lane < (COMPACT ? 12 : NV)) sh.J[MAXROW][lane] = 0;
  if constexpr (COMPACT) {
    if (lane < 2) sh.cb[MAXROW][lane] = COL_NONE;
  }
}
MJS_DEV void coop_coupled(bool need, const World& s, const double* cs, const double* sn, const double* Marm, int nb, double* qacc, StepInfo& info,
                          const FloorSlots* fs, const ConvexHits& cvx, const M3* Rb, bool arm_in, const bool* blk_in, double meaninertia) {
  CoopLds& sh = coop_lds();
  const int lane = threadIdx.x & 63, nv_all = NJ + 6 * nb;
  unsigned long long todo = __ballot(need);
  while (todo) {  // wave-uniform loop over the lanes whose env needs the coupled solve
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    for (int e = lane; e < NV * LDP; e += 64) (&sh.M[0][0])[e] = 0;  // all lanes: clear M, the owner fills its diagonal blocks
    MJS_WAVE_SYNC();
    unsigned long long tp = 0;
    PP_TIC(tp);
    if (lane == owner) {
      if constexpr (SPLIT_DETECT) {  // the detection results are read from the env's LDS slot here, where they are needed
        const EnvLds& env = env_lds();
        FloorSlots fs_l[NB];
        M3 Rb_l[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { fs_l[b] = env_fs(env, b); Rb_l[b] = env_Rb(env, b); }
        const ConvexHits cvx_l = env_cvx(env);
        publish_problem(s, cs, sn, Marm, qacc, nb, fs_l, cvx_l, Rb_l, arm_in, blk_in, meaninertia, info.arm_floor);
      } else
        publish_problem(s, cs, sn, Marm, qacc, nb, fs, cvx, Rb, arm_in, blk_in, meaninertia, info.arm_floor);
    }
    MJS_WAVE_SYNC();
    // mj_solPrimal's stopping rules are scaled by the WHOLE model (meaninertia, nv) of the owner env, also when a sub-system is solved
    const double scale = 1 / (sh.meaninertia * nv_all);
    PP_ACC(info, 5, tp);
    PP_TIC(tp);
    coop_build_rows(sh, lane);
    MJS_WAVE_SYNC();
    PP_ACC(info, 15, tp);
    const int iters = sh.nrow > 0 ? coop_newton(sh.nv, scale, lane, info) : 0;
#ifdef MJS_STAMPS
    info.cyc[6] += 1; info.cyc[7] += iters;
#else
    (void)iters;
#endif
    if (lane == owner && sh.overflow) info.unsupported = true;
    if (lane == owner && sh.nrow > 0) {
      if (arm_in) {
#pragma unroll
        for (int i = 0; i < NJ; i++) qacc[i] = sh.qs[i] + sh.fc[i];
      }
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int o = sh.off[1 + b];
        if (o < 0) continue;
#pragma unroll
        for (int k = 0; k < 6; k++) qacc[NJ + 6 * b + k] = sh.qs[o + k] + sh.fc[o + k];
      }
    }
    MJS_WAVE_SYNC();
  }
}

// sin / cos of the half rotation angle of one substep (0.5 |w| dt: < 0.125 rad unless a block spins faster than 50 rad/s):
// Taylor kernels (truncation error < 3e-20 there) instead of ocml's range-reducing sincos
MJS_DEV void sincos_small(double x, double* s, double* c) {
  if (fabs(x) < 0.125) {
    const double x2 = x * x;
    *s = x * (1.0 - x2 * (1.0 / 6) * (1.0 - x2 * (1.0 / 20) * (1.0 - x2 * (1.0 / 42) * (1.0 - x2 * (1.0 / 72) * (1.0 - x2 * (1.0 / 110) * (1.0 - x2 * (1.0 / 156)))))));
    *c = 1.0 - x2 * 0.5 * (1.0 - x2 * (1.0 / 12) * (1.0 - x2 * (1.0 / 30) * (1.0 - x2 * (1.0 / 56) * (1.0 - x2 * (1.0 / 90) * (1.0 - x2 * (1.0 / 132))))));
  } else
    sincos(x, s, c);
}

// One Physics.step() (mj_step2 of the current state; the next mj_step1 is the start of the next call): smooth
// dynamics, constraint solve, implicitfast for the servo'd arm / plain Euler for the blocks, position integration.
// `live` = this lane really steps its env; lanes that do not still take part in the cooperative solve of their
// neighbours (all 64 lanes of the workgroup must call this function together).
// Substep phase 1: kinematics, collision detection by the env groups, the coupled sub-system. In the 2-slot instance it
// is its own non-inlined function (detect_phase: the group-parallel MPR is the most register-hungry code of the substep and
// nothing of it is needed afterwards) that hands its results over through the env's LDS slot; the 5-slot instance, which has
// no LDS left for them, calls it in line and keeps them in registers.
MJS_DEV void detect_body(int nb, bool live, M3* Rb, FloorSlots* fs, ConvexHits& cvx, bool& arm_in, bool* blk_in) {
  EnvLds& env = env_lds();
  World& s = env.w;
  double* cs = env.cs;
  double* sn = env.sn;
  StepInfo& info = env.info;
  if (live) {
  rr::Chain ch;
  rr::fk_cs(cs, sn, ch);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    double qn[4];
    const double nrm = sqrt(s.b[b].q[0] * s.b[b].q[0] + s.b[b].q[1] * s.b[b].q[1] + s.b[b].q[2] * s.b[b].q[2] + s.b[b].q[3] * s.b[b].q[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) qn[k] = s.b[b].q[k] * (1.0 / nrm);
    Rb[b] = quat_to_m3(qn);
  }
  // contacts (static slots): floor corners per block; is any arm-block / block-block pair penetrating?
  Geom bg[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    bg[b] = block_geom(s.b[b], Rb[b]);
  }
  {
    const Geom eg = eef_geom(ch), wg = wrist3_proxy_geom(ch);
    const double prj = eg.R.cz.z, rad = sqrt(fmax(0.0, 1.0 - prj * prj));
    // the arm's own geoms or the EEF cylinder in the floor: rows of the arm alone, solved by the general stage in physics_step
    info.arm_floor = !(rr::min_floor_clearance(ch) >= 0.0) || (eg.c.z - fabs(prj) * eg.s.y - rad * eg.s.x < 0.0);
    // publish this env's geoms for the group-parallel pair evaluation below
    DetLds& dl = det_lds();
    const int lane_ = threadIdx.x & 63;
    put_geom(dl.g[lane_][0], wg);
    put_geom(dl.g[lane_][1], eg);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      put_geom(dl.g[lane_][2 + b], bg[b]);
      dl.org[lane_][b][0] = s.b[b].p.x; dl.org[lane_][b][1] = s.b[b].p.y; dl.org[lane_][b][2] = s.b[b].p.z;
    }
  }
  }  // live (first part)
  {
    // convex pairs in MuJoCo's pair order (wrist proxy - block b, EEF - block b, block a - block b): every env group of LPE
    // lanes evaluates the pairs of its env together (bounding spheres, then MPR with the hull scans split over the lanes)
    DetLds& dl = det_lds();
    const int lane_ = threadIdx.x & 63, grp = lane_ / LPE;
    if (lane_ < EPW) dl.live[lane_] = live ? 1 : 0;
    MJS_WAVE_SYNC();
    if (dl.live[grp]) {  // group-uniform
      // floor contacts of every block (static slots), by the whole group
#pragma unroll 1
      for (int b = 0; b < NB; b++) {
        const Geom gb = get_geom(dl.g[grp][2 + b]);
        const FloorSlots f = floor_slots_group(gb, v3(dl.org[grp][b][0], dl.org[grp][b][1], dl.org[grp][b][2]));
        if ((lane_ & (LPE - 1)) == 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            double* o = dl.fl[grp][b][k];
            o[0] = f.on[k] ? 1.0 : 0.0; o[1] = f.dist[k]; o[2] = f.r[k].x; o[3] = f.r[k].y; o[4] = f.r[k].z;
          }
          dl.fln[grp][b] = f.n;
        }
      }
    }
    {
      // The pairs of ALL the wavefront's envs that pass the bounding-sphere test form lists of up to 64 (one lane per (env,
      // pair) does the sphere test, a ballot is the list) and the env groups take the listed pairs in turn, whichever env
      // they belong to: an env with a block on the arm, or with blocks leaning on each other, needs a full MPR run for
      // several pairs in every substep while the groups of the other envs had nothing to do (the launch is as slow as that
      // wavefront). Every group runs the same MPR code on its own item.
      constexpr int NIT = EPW * NCVX, GROUPS = 64 / LPE;
#pragma unroll 1
      for (int base = 0; base < NIT; base += 64) {
        const int item_ = base + lane_;
        bool want = false;
        if (item_ < NIT) {
          const int e = item_ / NCVX, k = item_ % NCVX;
          int ia, ib;
          pair_geoms(k, ia, ib);
          if (dl.live[e] && ib - 2 < nb) {
            const Geom g1 = get_geom(dl.g[e][ia]), g2 = get_geom(dl.g[e][ib]);
            const V3 diff = sub_nc(g2.c, g1.c);
            const double bound = rbound(g1) + rbound(g2);
            want = !(dot_nc(diff, diff) > bound * bound);
            if (!want) {
              double* o = dl.out[e][k];
              o[0] = 0.0; o[1] = 0.0; o[2] = 0.0; o[3] = 0.0; o[4] = 0.0; o[5] = 0.0; o[6] = 0.0; o[7] = 1.0;
            }
          }
        }
        const unsigned long long todo = __ballot(want);
        const int nitems = __popcll(todo);
#pragma unroll 1
        for (int r = 0; r * GROUPS < nitems; r++) {  // wave-uniform
          const int target = r * GROUPS + grp;
          int item = -1;
          {
            unsigned long long m = todo;
            for (int c = 0; m; c++) {
              const int bit = __ffsll((long long)m) - 1;
              m &= m - 1;
              if (c == target) { item = base + bit; break; }
            }
          }
          if (item >= 0) {  // group-uniform
            const int e = item / NCVX, k = item % NCVX;
            int ia, ib;
            pair_geoms(k, ia, ib);
            const Geom g1 = get_geom(dl.g[e][ia]), g2 = get_geom(dl.g[e][ib]);
            const PairHit h = convex_pair_group(g1, g2);
            if ((lane_ & (LPE - 1)) == 0) {
              double* o = dl.out[e][k];
              o[0] = h.hit ? 1.0 : 0.0; o[1] = h.dist; o[2] = h.pos.x; o[3] = h.pos.y; o[4] = h.pos.z; o[5] = h.n.x; o[6] = h.n.y; o[7] = h.n.z;
            }
          }
        }
      }
    }
    MJS_WAVE_SYNC();
  }
  if (live) {
  {
    DetLds& dl = det_lds();
    const int lane_ = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NCVX; k++) { cvx.hit[k] = false; cvx.dist[k] = 0; cvx.pos[k] = v3(0, 0, 0); cvx.n[k] = v3(0, 0, 1); }
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double* o = dl.fl[lane_][b][k];
        fs[b].on[k] = o[0] != 0.0; fs[b].dist[k] = o[1]; fs[b].r[k] = v3(o[2], o[3], o[4]);
      }
      fs[b].n = dl.fln[lane_][b];
    }
    auto take = [&](int slot) {
      const double* o = dl.out[lane_][slot];
      cvx.hit[slot] = o[0] != 0.0; cvx.dist[slot] = o[1]; cvx.pos[slot] = v3(o[2], o[3], o[4]); cvx.n[slot] = v3(o[5], o[6], o[7]);
    };
#pragma unroll
    for (int b = 0; b < NB; b++)
      if (b < nb) { take(b); take(NB + b); }
#pragma unroll
    for (int a = 0; a < NB; a++) {
#pragma unroll
      for (int b = a + 1; b < NB; b++)
        if (b < nb) take(pair_slot(a, b));
    }
    // the coupled sub-system: the bodies joined by an ACTIVE arm-block / block-block contact; every other block only
    // touches the floor and stays an independent 6-dof problem
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const bool act = (cvx.hit[b] && cvx.dist[b] < 0.0) || (cvx.hit[NB + b] && cvx.dist[NB + b] < 0.0);
      arm_in = arm_in || act;
      blk_in[b] = blk_in[b] || act;
    }
#pragma unroll
    for (int a = 0; a < NB; a++) {
#pragma unroll
      for (int b = a + 1; b < NB; b++) {
        const int k = pair_slot(a, b);
        const bool act = cvx.hit[k] && cvx.dist[k] < 0.0;
        blk_in[a] = blk_in[a] || act;
        blk_in[b] = blk_in[b] || act;
      }
    }
  }
  }  // live
}

// (round 3) inlined into the substep: as a call it saved / restored the callee-saved registers it used in every substep for all 64
// lanes, 0.37 GB of scratch writes per 4096-env launch; inlined the launch moves 0.14 GB instead of 0.44 GB at the same speed
// within the run-to-run spread (profiles/r03_f_push_traffic.txt; in round 2's build the same change cost 8 %).
__device__ __forceinline__ void detect_phase(int nb, bool live) {
  EnvLds& env = env_lds();
  bool arm_in = false, blk_in[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) blk_in[b] = false;
  ConvexHits cvx;
  FloorSlots fs[NB];
  M3 Rb[NB];
  detect_body(nb, live, Rb, fs, cvx, arm_in, blk_in);
  if constexpr (SPLIT_DETECT) {
  if (live) {
    // hand the results to the next phase through the env's LDS slot
#pragma unroll
    for (int b = 0; b < NB; b++) {
      env.Rb[b][0] = Rb[b].cx.x; env.Rb[b][1] = Rb[b].cx.y; env.Rb[b][2] = Rb[b].cx.z; env.Rb[b][3] = Rb[b].cy.x; env.Rb[b][4] = Rb[b].cy.y; env.Rb[b][5] = Rb[b].cy.z;
      env.Rb[b][6] = Rb[b].cz.x; env.Rb[b][7] = Rb[b].cz.y; env.Rb[b][8] = Rb[b].cz.z;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double* o = env.fl[b][k];
        o[0] = fs[b].on[k] ? 1.0 : 0.0; o[1] = fs[b].dist[k]; o[2] = fs[b].r[k].x; o[3] = fs[b].r[k].y; o[4] = fs[b].r[k].z;
      }
      env.blk_in[b] = blk_in[b] ? 1 : 0;
      env.fln[b] = fs[b].n;
    }
#pragma unroll
    for (int k = 0; k < NCVX; k++) {
      double* o = env.cv[k];
      o[0] = cvx.hit[k] ? 1.0 : 0.0; o[1] = cvx.dist[k]; o[2] = cvx.pos[k].x; o[3] = cvx.pos[k].y; o[4] = cvx.pos[k].z; o[5] = cvx.n[k].x; o[6] = cvx.n[k].y; o[7] = cvx.n[k].z;
    }
    env.arm_in = arm_in ? 1 : 0;
  }
  }
}

// mj_collision's ncon of the env's CURRENT state (the state after the last substep of a control step), counted from one more
// run of the group-parallel detection phase. The serial count_contacts below (still used by the reset's rejection sampling)
// walks the hulls on one lane per env (full MPR runs with 96-vertex scans for an env that is pushing a block: 9 % of the
// launch); all 64 lanes must call this one.
__device__ __noinline__ int count_contacts_group(int nb, bool live) {
  int n = 0;
  // floor contacts per block and convex-pair hits of the current state
  int fln[NB];
  bool hit[NCVX];
  if constexpr (SPLIT_DETECT) {
    detect_phase(nb, live);  // results in the env's LDS slot
    const EnvLds& env = env_lds();
#pragma unroll
    for (int b = 0; b < NB; b++) fln[b] = live ? env.fln[b] : 0;
#pragma unroll
    for (int k = 0; k < NCVX; k++) hit[k] = live && env.cv[k][0] != 0.0;
  } else {  // the 5-slot instance keeps the detection results in registers: its own copy of the phase
    bool arm_in = false, blk_in[NB];
    ConvexHits cvx;
    FloorSlots fs[NB];
    M3 Rb[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { blk_in[b] = false; fs[b].n = 0; }
#pragma unroll
    for (int k = 0; k < NCVX; k++) cvx.hit[k] = false;
    detect_body(nb, live, Rb, fs, cvx, arm_in, blk_in);
#pragma unroll
    for (int b = 0; b < NB; b++) fln[b] = live ? fs[b].n : 0;
#pragma unroll
    for (int k = 0; k < NCVX; k++) hit[k] = live && cvx.hit[k];
  }
  if (live) {
    const EnvLds& env = env_lds();
    rr::Chain ch;
    rr::fk_cs(env.cs, env.sn, ch);
    n = rr::count_floor_contacts(ch) + count_eef_floor_contacts(ch);
#pragma unroll
    for (int b = 0; b < NB; b++)
      if (b < nb) n += fln[b] + (hit[b] ? 1 : 0) + (hit[NB + b] ? 1 : 0);
#pragma unroll
    for (int a = 0; a < NB; a++) {
#pragma unroll
      for (int b = a + 1; b < NB; b++)
        if (b < nb) n += hit[pair_slot(a, b)] ? 1 : 0;
    }
  }
  return n;
}

// The arm's floor contacts in this scene (rr::gen_stage, mjs_arm_stage.h): the arm's ten collision geoms come from the stage
// itself; this adds the CylinderEEF (geom on the EEF body welded to wrist_3: all six joints move it) against the floor,
// mjc_PlaneCylinder's up to four contacts, after the arm geoms in MuJoCo's pair order. The solver's stopping rules are scaled
// by the WHOLE model of this env (arm + blocks: mj_setConst's meaninertia, nv = 6 + 6 n), as in the oracle's one Newton problem.
struct ScenePush {
  struct Extra { double scale; };
  MJS_DEV static double solver_scale(Extra ex) { return ex.scale; }
  MJS_DEV static double dof_invweight(int j) { return UR5E_PP_DOF_INVWEIGHT0[j]; }
  MJS_DEV static double link_invweight(int b) { return UR5E_PP_LINK_BODY_INVWEIGHT0[b]; }
  template <class E>
  MJS_DEV static void extra_contacts(const rr::Chain& ch, Extra, E emit) {
    const Geom eg = eef_geom(ch);
    rr::plane_cylinder_contacts(eg.c, eg.R.cz, eg.R.cx, eg.s.x, eg.s.y,
                                [&](V3 pos, double dist) { emit(NJ, pos, v3(0, 0, 1), 1.0, dist, UR5E_PP_EEF_BODY_INVWEIGHT0[0], false); });
  }
  MJS_DEV static bool in_touch_site(Extra, V3) { return false; }
};

// The substep is inlined into its two call sites (the control-step loop and the same-step auto-reset's settle loop): as a
// non-inlined function it saved and restored ~130 callee-saved registers per call, every lane, every substep.
__device__ __forceinline__ void physics_step(int nb, bool live, rr::Ws ws) {
  EnvLds& env = env_lds();
  World& s = env.w;
  const double* ctrl = env.ctrl;
  double* cs = env.cs;
  double* sn = env.sn;
  StepInfo& info = env.info;
  const int nv = NJ + 6 * nb;
  double Marm_l[SPLIT_DETECT ? 1 : 21], qacc_l[SPLIT_DETECT ? 1 : NV];
  double* Marm = SPLIT_DETECT ? env.Marm : Marm_l;
  double* qacc = SPLIT_DETECT ? env.qacc : qacc_l;
  int clamped = 0;
  bool coupled = false, arm_in = false, blk_in[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) blk_in[b] = false;
  ConvexHits cvx;        // 5-slot instance only (one function: the results are locals)
  FloorSlots fs[NB];
  M3 Rb[NB];
  auto get_Rb = [&](int b) { if constexpr (SPLIT_DETECT) return env_Rb(env, b); else return Rb[b]; };
  auto get_fs = [&](int b) { if constexpr (SPLIT_DETECT) return env_fs(env, b); else return fs[b]; };
  unsigned long long tt = 0;
  PP_TIC(tt);
  if constexpr (SPLIT_DETECT) detect_phase(nb, live);
  else detect_body(nb, live, Rb, fs, cvx, arm_in, blk_in);
  PP_ACC(info, 0, tt);
  if (live) {
    if constexpr (SPLIT_DETECT) {  // the detection results stay in the env's LDS slot and are read where they are used
#pragma unroll
      for (int b = 0; b < NB; b++) blk_in[b] = env.blk_in[b] != 0;
      arm_in = env.arm_in != 0;
    }
  // arm smooth dynamics
  double bias[NJ], fact[NJ];
  ur5e_pp_M_gen(cs, sn, Marm);
  ur5e_pp_bias_gen(cs, sn, s.v, bias);
  clamped = rr::actuator_forces(s.q, s.v, ctrl, fact);
#pragma unroll
  for (int i = 0; i < NJ; i++) qacc[i] = fact[i] - bias[i];
  PP_ACC(info, 1, tt);
  // which constraint problem?
#pragma unroll
  for (int j = 0; j < NJ; j++) arm_in = arm_in || s.q[j] < MJS_UR_JNT_RANGE[j][0] || s.q[j] > MJS_UR_JNT_RANGE[j][1];
  coupled = arm_in;
#pragma unroll
  for (int b = 0; b < NB; b++) coupled = coupled || blk_in[b];
  if (coupled) info.rows_active = true;
#pragma unroll
  for (int b = 0; b < NB; b++)
    if (b < nb) block_smooth_force(get_Rb(b), s.b[b].w, shape_of(s.b[b]), qacc + NJ + 6 * b);
  }  // live
  // mj_setConst's meaninertia of THIS env's model: the arm's share + every block's (shape-dependent) share
  double meaninertia = UR5E_PP_MEANINERTIA * NJ;
#pragma unroll
  for (int b = 0; b < NB; b++)
    if (b < nb) meaninertia += shape_inertia_trace(shape_of(s.b[b]));
  meaninertia /= nv;
  // decoupled case (nothing but floor contacts): every block with its corner contacts is an independent 6-dof problem,
  // solved by a quad of lanes (quad_block_floor). Env lanes publish, all lanes solve, env lanes collect.
  {
    QuadLds& qx = quad_lds();
    const int lane = threadIdx.x & 63;
    if (lane < EPW) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const FloorSlots fsb = get_fs(b);
        const bool need = live && !blk_in[b] && b < nb && (fsb.on[0] || fsb.on[1] || fsb.on[2] || fsb.on[3]);
        QuadIn& in = qx.in[lane][b];
        in.need = need;
        if (need) {
          const M3 Rbb = get_Rb(b);
          info.rows_active = true;
          in.shape = s.b[b].shape; in.scale = s.b[b].scale; in.meaninertia = meaninertia;
          in.R[0] = Rbb.cx.x; in.R[1] = Rbb.cx.y; in.R[2] = Rbb.cx.z; in.R[3] = Rbb.cy.x; in.R[4] = Rbb.cy.y; in.R[5] = Rbb.cy.z;
          in.R[6] = Rbb.cz.x; in.R[7] = Rbb.cz.y; in.R[8] = Rbb.cz.z;
          in.qv[0] = s.b[b].v.x; in.qv[1] = s.b[b].v.y; in.qv[2] = s.b[b].v.z; in.qv[3] = s.b[b].w.x; in.qv[4] = s.b[b].w.y; in.qv[5] = s.b[b].w.z;
#pragma unroll
          for (int k = 0; k < 6; k++) in.f[k] = qacc[NJ + 6 * b + k];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            in.on[k] = fsb.on[k]; in.dist[k] = fsb.dist[k];
            in.r[k][0] = fsb.r[k].x; in.r[k][1] = fsb.r[k].y; in.r[k][2] = fsb.r[k].z;
          }
        }
      }
    }
    MJS_WAVE_SYNC();
#pragma unroll 1
    for (int b0 = 0; b0 < nb; b0 += QUAD_BLOCKS) {
      const int c = lane & 3, e = (lane >> 2) & (EPW - 1), b = b0 + (lane >> 2) / EPW;
      if (b < nb && qx.in[e][b].need) {  // quad-uniform
        const QuadIn& in = qx.in[e][b];
        M3 R;
        R.cx = v3(in.R[0], in.R[1], in.R[2]); R.cy = v3(in.R[3], in.R[4], in.R[5]); R.cz = v3(in.R[6], in.R[7], in.R[8]);
        double qv[6], f[6], out[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { qv[k] = in.qv[k]; f[k] = in.f[k]; }
        Block shp_src;
        shp_src.shape = in.shape; shp_src.scale = in.scale;
        quad_block_floor(R, shape_of(shp_src), qv, f, in.on[c] != 0, in.dist[c], v3(in.r[c][0], in.r[c][1], in.r[c][2]), in.meaninertia, nv, out);
        if (c == 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) qx.out[e][b][k] = out[k];
        }
      }
    }
    MJS_WAVE_SYNC();
    if (lane < EPW) {
#pragma unroll
      for (int b = 0; b < NB; b++)
        if (qx.in[lane][b].need) {
#pragma unroll
          for (int k = 0; k < 6; k++) qacc[NJ + 6 * b + k] = qx.out[lane][b][k];
        }
    }
    MJS_WAVE_SYNC();  // the cooperative solver reuses the area
  }
  PP_ACC(info, 2, tt);
  coop_coupled(live && coupled, s, cs, sn, Marm, nb, qacc, info, fs, cvx, Rb, arm_in, blk_in, meaninertia);  // all lanes
  PP_ACC(info, 3, tt);
  if (!live) return;
  // Arm geoms / EEF cylinder in the floor (an unreachable or low target drags the tool over the floor; mjs_set_state): while the arm
  // is not coupled to a block its own 6-dof constraint problem (floor contacts + joint limits) goes through the general stage,
  // cold-started like every solve of this kernel; an arm that is ALSO in the coupled sub-system had its floor contacts solved there
  // (publish_problem: floor-arm rows with the touching link's columns; more than MAXAF of them are reported).
  if (info.arm_floor) {
    if (arm_in && coupled) { /* solved by coop_coupled */ }
    else {
      rr::GenStageIn gi;
#pragma unroll
      for (int j = 0; j < NJ; j++) { gi.q[j] = s.q[j]; gi.v[j] = s.v[j]; gi.cs[j] = cs[j]; gi.sn[j] = sn[j]; gi.qs[j] = qacc[j]; gi.warm[j] = 0.0; }
#pragma unroll
      for (int i = 0; i < NJ; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) gi.M[i * (i + 1) / 2 + j] = Marm[i * (i + 1) / 2 + j] + (i == j ? MJS_UR_ARMATURE : 0.0);
      }
      gi.has_warm = false;
      const rr::GenStageOut go = rr::gen_stage<ScenePush>(gi, ScenePush::Extra{1.0 / (meaninertia * nv)}, ws);
#pragma unroll
      for (int j = 0; j < NJ; j++) qacc[j] = go.qs[j];
      info.rows_active = true;
      info.unsupported = info.unsupported || go.overflow;
    }
  }
  // integrator: arm implicitfast (M + armature + dt * kd on unclamped actuators), blocks M qacc = f
  {
    double A[NJ][NJ], rhs[NJ], Dinv[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = Marm[i * (i + 1) / 2 + j];
      rhs[i] = qacc[i];
    }
    rr::factor_system(A, clamped, Dinv);
    rr::udu_solve(A, Dinv, rhs);
#pragma unroll
    for (int i = 0; i < NJ; i++) qacc[i] = rhs[i];
  }
  double acc2 = 0, dq2 = 0;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b >= nb) continue;
    const int o = NJ + 6 * b;
    double x[6];
    block_minv(get_Rb(b), shape_of(s.b[b]), qacc + o, x);
#pragma unroll
    for (int i = 0; i < 6; i++) { qacc[o + i] = x[i]; acc2 = fma(x[i], x[i], acc2); }
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    acc2 = fma(qacc[j], qacc[j], acc2);
    s.v[j] += MJS_RR_PHYSICS_DT * qacc[j];
    const double dq = MJS_RR_PHYSICS_DT * s.v[j];
    s.q[j] += dq;
    dq2 = fma(dq, dq, dq2);
    rr::rotate_small(cs[j], sn[j], dq);
  }
  if (!(dq2 <= 0.01)) {
#pragma unroll
    for (int j = 0; j < NJ; j++) sincos(s.q[j], &sn[j], &cs[j]);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b >= nb) continue;
    const int o = NJ + 6 * b;
    Block& k = s.b[b];
    k.v = madd(k.v, MJS_RR_PHYSICS_DT, v3(qacc[o], qacc[o + 1], qacc[o + 2]));
    k.w = madd(k.w, MJS_RR_PHYSICS_DT, v3(qacc[o + 3], qacc[o + 4], qacc[o + 5]));
    k.p = madd(k.p, MJS_RR_PHYSICS_DT, k.v);
    // mju_quatIntegrate with the body-frame angular velocity
    double nrm = sqrt(k.q[0] * k.q[0] + k.q[1] * k.q[1] + k.q[2] * k.q[2] + k.q[3] * k.q[3]);
    const double ang = sqrt(dot(k.w, k.w));
    if (ang >= MJS_MINVAL) {
      const V3 ax = (1.0 / ang) * k.w;
      double sh, chf;
      sincos_small(0.5 * ang * MJS_RR_PHYSICS_DT, &sh, &chf);
      const double inrm = 1.0 / nrm;
      const double q0 = k.q[0] * inrm, q1 = k.q[1] * inrm, q2 = k.q[2] * inrm, q3 = k.q[3] * inrm;
      const double r0 = chf, r1 = ax.x * sh, r2 = ax.y * sh, r3 = ax.z * sh;
      k.q[0] = q0 * r0 - q1 * r1 - q2 * r2 - q3 * r3;
      k.q[1] = q0 * r1 + q1 * r0 + q2 * r3 - q3 * r2;
      k.q[2] = q0 * r2 - q1 * r3 + q2 * r0 + q3 * r1;
      k.q[3] = q0 * r3 + q1 * r2 - q2 * r1 + q3 * r0;
      nrm = sqrt(k.q[0] * k.q[0] + k.q[1] * k.q[1] + k.q[2] * k.q[2] + k.q[3] * k.q[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) k.q[i] *= 1.0 / nrm;
  }
  info.bad = info.bad || !(acc2 <= 1e20);
  s.time += MJS_RR_PHYSICS_DT;
  PP_ACC(info, 4, tt);
}

// contacts of the current state as mj_forward / the trailing mj_step1 would report them (d->ncon)
__device__ __noinline__ int count_contacts(const World& s, const double* cs, const double* sn, int nb) {
  rr::Chain ch;
  rr::fk_cs(cs, sn, ch);
  M3 Rb[NB];
  for (int b = 0; b < nb; b++) {
    double qn[4];
    const double nrm = sqrt(s.b[b].q[0] * s.b[b].q[0] + s.b[b].q[1] * s.b[b].q[1] + s.b[b].q[2] * s.b[b].q[2] + s.b[b].q[3] * s.b[b].q[3]);
    for (int k = 0; k < 4; k++) qn[k] = s.b[b].q[k] * (1.0 / nrm);
    Rb[b] = quat_to_m3(qn);
  }
  Contact con[MAXCON];
  int extra;
  bool eef_floor;
  return detect_contacts(ch, s, Rb, nb, con, extra, eef_floor) + extra;
}

MJS_DEV bool tcp_to_joints(const double* pos, const double* guess, double* q_out) { return rr::tcp_pose_to_joints_offset(pos, MJS_CYL_TCP_Z, guess, q_out); }

MJS_DEV void make_obs(const World& s, const double* cs, const double* sn, int nb, double* obs) {
  rr::Chain c;
  rr::fk_cs(cs, sn, c);
  const V3 tcp = eef_tcp_position(c);
  obs[0] = tcp.x; obs[1] = tcp.y; obs[2] = tcp.z;            // ur5e/tcp_position
  obs[3] = s.target[0]; obs[4] = s.target[1];                // target_position = site.pos[:2]
  for (int b = 0; b < NB; b++) {                             // block_positions = body xpos[:2]
    obs[5 + 2 * b] = b < nb ? s.b[b].p.x : 0.0;
    obs[6 + 2 * b] = b < nb ? s.b[b].p.y : 0.0;
  }
}

// initialize_episode (robot_planar_push.py:149-176, intended semantics), first part: the draws. The 150 settle steps
// run in the kernel's uniform substep loop. `commit` = false leaves the env's RNG stream untouched (padding lanes).
__device__ __noinline__ void episode_draws(DevRng rng, int i, int nb, World& s, bool commit, bool mesh_blocks) {
  RngCursor c = rng_open(rng, i);
  double rp[3], q[NJ], zeros[NJ] = {0, 0, 0, 0, 0, 0};
  // initialize_episode_mjcf (robot_planar_push.py:144-147,163-167) comes first: every block is replaced by
  // GoogleBlockProp.sample_random_object() (google_block.py:55-68): category, colour, scale in [0.8, 1.2]; three uniforms per block
  // from the env's seeded stream (deviation D-5: the reference uses Python's unseeded global `random`)
  for (int b = 0; b < NB; b++) { s.b[b].shape = -1.0; s.b[b].scale = 1.0; }
  if (mesh_blocks) {
    for (int b = 0; b < nb; b++) {
      int cat = (int)rng_uniform(rng, i, c, 0.0, (double)MJS_HULL_NCAT), col = (int)rng_uniform(rng, i, c, 0.0, 6.0);
      cat = cat < MJS_HULL_NCAT ? cat : MJS_HULL_NCAT - 1;
      col = col < 6 ? col : 5;
      s.b[b].shape = (double)(cat + 8 * col);
      s.b[b].scale = rng_uniform(rng, i, c, MJS_BLOCK_SCALE_LO, MJS_BLOCK_SCALE_HI);
    }
  }
  for (int k = 0; k < 3; k++) rp[k] = rng_uniform(rng, i, c, MJS_PP_ROBOT_SPACE_LO[k], MJS_PP_ROBOT_SPACE_HI[k]);
  const bool ok = tcp_to_joints(rp, zeros, q);
  for (int j = 0; j < NJ; j++) { s.q[j] = ok ? q[j] : 0.0; s.v[j] = 0; }
  for (int k = 0; k < 3; k++) s.target[k] = rng_uniform(rng, i, c, MJS_PP_TARGET_SPACE_LO[k], MJS_PP_TARGET_SPACE_HI[k]);
  s.time = 0;
  s.episode_step = 0;
  double cs[NJ], sn[NJ];
  for (int j = 0; j < NJ; j++) sincos(s.q[j], &sn[j], &cs[j]);
  for (int b = 0; b < NB; b++) { s.b[b].p = v3(0, 0, 0); s.b[b].q[0] = 1; s.b[b].q[1] = s.b[b].q[2] = s.b[b].q[3] = 0; s.b[b].v = v3(0, 0, 0); s.b[b].w = v3(0, 0, 0); }
  for (int attempt = 0; attempt < 1000; attempt++) {  // randomize_object_position: until mj_forward reports ncon == 0
    for (int b = 0; b < nb; b++) {
      double bp[3];
      for (int k = 0; k < 3; k++) bp[k] = rng_uniform(rng, i, c, MJS_PP_OBJECT_SPACE_LO[k], MJS_PP_OBJECT_SPACE_HI[k]);
      s.b[b].p = v3(bp[0], bp[1], bp[2]);
    }
    if (count_contacts(s, cs, sn, nb) == 0) break;
  }
  if (commit) rng_close(rng, i, c);
}

// One workgroup = one wavefront = 64 envs, lane per env. Control flow is UNIFORM across the wavefront (the coupled
// constraint problems are solved cooperatively): every lane runs the same substep loop, with `live` masking lanes
// that have nothing to do in an iteration (padding lanes of the last workgroup, lanes that step while their neighbours
// run the 150 settle steps of a reset).
template <bool IS_RESET>
__global__ __launch_bounds__(64 * WAVES) void kernel(KernelParams p) {
  {  // prologue: hull tables -> LDS (every lane of the workgroup runs the whole kernel: uniform control flow)
    double* dst = pp_lds_raw + (sizeof(CoopLds) * WAVES) / sizeof(double);
    const double* src = &MJS_HULL_VERT_C[0][0][0];
    for (int k = threadIdx.x; k < HULL_LDS_DOUBLES; k += 64 * WAVES) dst[k] = src[k];
    __syncthreads();
  }
  const int groups = (int)gridDim.x >> ((!IS_RESET && p.prefetch) ? 1 : 0);
  const bool prefetcher = !IS_RESET && p.prefetch && (int)blockIdx.x >= groups;
  const int wg = prefetcher ? (int)blockIdx.x - groups : (int)blockIdx.x;
  const int gi = (wg * WAVES + (threadIdx.x >> 6)) * EPW + (threadIdx.x & 63);
  const bool valid = (threadIdx.x & 63) < EPW && gi < p.N;
  const int i = valid ? gi : 0;  // helper / padding lanes shadow env 0 and never write
  const int nb = p.n_objects;
  uint8_t flags = p.flags[i];
  double obs[OBS_DIM], cs[NJ], sn[NJ], ctrl0[NJ], q0[NJ], q1[NJ];
  World s;
  double* const next_base = p.state + (size_t)NEXT_ROW0 * p.N;
  double* const prog_row = p.state + (size_t)PROG_ROW * p.N;
  int prog = valid ? (int)prog_row[i] : -1;  // how far the env's next episode has been prepared (see NEXT_ROW0)
  const bool masked_out = IS_RESET && p.reset_mask && !p.reset_mask[i];
  const bool pending = (flags & FLAG_RESET_PENDING) && p.autoreset == MJS_AUTORESET_NEXT_STEP;
  const bool resetting = valid && !prefetcher && !masked_out && (IS_RESET || pending);
  const bool stepping = valid && !prefetcher && !IS_RESET && !resetting;
  const bool filling = valid && prefetcher && !pending && prog < MJS_PP_SETTLE_STEPS;
  if (prefetcher && !__syncthreads_or(filling)) return;  // nothing to prepare in this group
  double t0 = 0, t1 = 1, inv_span = 1;
  int nsub = 0;
  bool carried = false;  // the slot holds the cos / sin rows of a settle phase that is under way
  if (resetting || filling) {
    if (prog >= 0) {  // drawn earlier (and settled for `prog` steps)
      s = load_world_at(next_base, (size_t)p.N, i);
      for (int j = 0; j < NJ; j++) ctrl0[j] = p.state[(size_t)(CTRL_ROW0 + j) * p.N + i];
      carried = prog > 0;
    } else {
      episode_draws(p.rng, i, nb, s, valid, p.block_shape == MJS_BLOCKS_MESH);
      prog = 0;
      for (int j = 0; j < NJ; j++) ctrl0[j] = s.q[j];  // Robot.set_joint_positions leaves ctrl = the reset joints (robot.py:185-189)
    }
    nsub = MJS_PP_SETTLE_STEPS - prog;  // 0 when the episode was ready: a reset is then a swap of the slots
    if (filling && nsub > PREFETCH_CHUNK) nsub = PREFETCH_CHUNK;
  } else if (stepping) {
    s = load_world(p, i);
    // before_step (base.py:31-32, robot_planar_push.py:185-201)
    s.episode_step += 1.0;
    double act[3] = {p.actions[(size_t)i * ACT_DIM], p.actions[(size_t)i * ACT_DIM + 1], MJS_PP_ACTION_Z};
    for (int j = 0; j < NJ; j++) q0[j] = s.q[j];
    if (!tcp_to_joints(act, q0, q1)) {
      flags |= FLAG_IK_FAILED;
      for (int j = 0; j < NJ; j++) q1[j] = q0[j];
    }
    t0 = s.time; t1 = s.time + MJS_RR_CONTROL_DT; inv_span = 1.0 / (t1 - t0);
    nsub = MJS_RR_NSUB;
  } else {
    s = load_world(p, i);
  }
  for (int j = 0; j < NJ; j++) sincos(s.q[j], &sn[j], &cs[j]);
  if (carried)
    for (int j = 0; j < NJ; j++) { cs[j] = p.state[(size_t)(CS_ROW0 + j) * p.N + i]; sn[j] = p.state[(size_t)(CS_ROW0 + NJ + j) * p.N + i]; }
  if (!valid) nsub = 0;  // helper lanes only take part in the cooperative solves
  StepInfo info{false, false, false, 0};
#ifdef MJS_STAMPS
  for (int k = 0; k < 16; k++) info.cyc[k] = 0;
#endif
  int nmax = nsub;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m));
  EnvLds& env = env_lds();
  bool bad = false, terminate = false, again = false;
  // Two passes through ONE substep loop (the substep is inlined: a single call site keeps one copy of it in the
  // instruction cache): pass 0 = this control step (or the reset's settle steps), pass 1 = the settle steps of the
  // same-step auto-reset (SB3 VecEnv convention) of the lanes whose episode just ended; the others idle through it.
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    int nloop = nmax;
    if (pass == 1) {
      if (IS_RESET || p.autoreset != MJS_AUTORESET_SAME_STEP || !__any(stepping && terminate)) break;
      again = stepping && terminate;
      if (again) {
        if (valid && p.out.terminal_obs)
          for (int k = 0; k < OBS_DIM; k++) p.out.terminal_obs[(size_t)i * OBS_DIM + k] = obs[k];
        episode_draws(p.rng, i, nb, s, valid, p.block_shape == MJS_BLOCKS_MESH);
        for (int j = 0; j < NJ; j++) { ctrl0[j] = s.q[j]; sincos(s.q[j], &sn[j], &cs[j]); }
      }
      info = StepInfo{false, false, false, 0};
#ifdef MJS_STAMPS
      for (int k = 0; k < 16; k++) info.cyc[k] = 0;
#endif
      nloop = MJS_PP_SETTLE_STEPS;
    }
    env.w = s;
    for (int j = 0; j < NJ; j++) { env.cs[j] = cs[j]; env.sn[j] = sn[j]; }
    env.info = info;
    const bool lerp = stepping && pass == 0;
#pragma unroll 1
    for (int sub = 0; sub < nloop; sub++) {
      if (lerp) {
        const double t = fmin(fmax(env.w.time, t0), t1);
        for (int j = 0; j < NJ; j++) env.ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - t0) * inv_span;
      } else {
        for (int j = 0; j < NJ; j++) env.ctrl[j] = ctrl0[j];
      }
      physics_step(nb, pass == 0 ? sub < nsub : (again && valid), rr::Ws{p.ws, p.N, i});
    }
    s = env.w;
    for (int j = 0; j < NJ; j++) { cs[j] = env.cs[j]; sn[j] = env.sn[j]; }
    info = env.info;
    // ncon of the state just reached (mj_step1 of the last substep): by the env groups where the detection results live in LDS
    if (prefetcher) {  // the prepared episode goes back to its slot; nothing is reported
      if (filling) {
        store_world_at(next_base, (size_t)p.N, i, s);
        prog_row[i] = (double)(prog + nsub);
        for (int j = 0; j < NJ; j++) {
          p.state[(size_t)(CTRL_ROW0 + j) * p.N + i] = ctrl0[j];
          p.state[(size_t)(CS_ROW0 + j) * p.N + i] = cs[j];
          p.state[(size_t)(CS_ROW0 + NJ + j) * p.N + i] = sn[j];
        }
      }
      return;
    }
    const int ncon_now = count_contacts_group(nb, valid && (pass == 0 ? (resetting || stepping) : again));
    if (pass == 1) {
      if (again) {
        const int ncon2 = ncon_now;
        if (valid) {
          store_world(p, i, s);
          p.flags[i] = 0;
          make_obs(s, cs, sn, nb, obs);
          if (p.out.obs)
            for (int k = 0; k < OBS_DIM; k++) p.out.obs[(size_t)i * OBS_DIM + k] = obs[k];
          if (p.out.ncon) p.out.ncon[i] = ncon2;
        }
      }
      break;
    }
#ifdef MJS_STAMPS
    if (p.stamps && threadIdx.x == 0 && !IS_RESET)
      for (int k = 0; k < 16; k++) p.stamps[(size_t)blockIdx.x * 16 + k] = info.cyc[k];
#endif
    bad = info.bad;
    if (resetting) {
      const int ncon = ncon_now;
      for (int j = 0; j < NJ; j++) bad = bad || bad_value(s.q[j]) || bad_value(s.v[j]);  // (a prepared episode's settle ran in earlier launches)
      for (int b = 0; b < nb; b++) bad = bad || bad_value(s.b[b].p.x) || bad_value(s.b[b].p.y) || bad_value(s.b[b].p.z);
      if (valid) {
        store_world(p, i, s);
        prog_row[i] = -1.0;  // the next-episode slot is consumed
        p.flags[i] = 0;
        make_obs(s, cs, sn, nb, obs);
        write_outputs<OBS_DIM>(p, i, obs, 0.0, 1.0, MJS_STEP_FIRST, false, false, false, bad ? MJS_FAULT_BAD_STATE : 0, ncon);
      }
    } else if (stepping) {
      for (int j = 0; j < NJ; j++) bad = bad || bad_value(s.q[j]) || bad_value(s.v[j]);
      for (int b = 0; b < nb; b++) bad = bad || bad_value(s.b[b].p.x) || bad_value(s.b[b].p.y) || bad_value(s.b[b].p.z) || bad_value(s.b[b].v.x) || bad_value(s.b[b].v.y) || bad_value(s.b[b].v.z);
      make_obs(s, cs, sn, nb, obs);
      // reward / accomplished / step limit (robot_planar_push.py:203-241, base.py:47-57)
      double sum = 0, nearest = INFINITY;
      int inside = 0;
      for (int b = 0; b < nb; b++) {
        const double dx = s.b[b].p.x - s.target[0], dy = s.b[b].p.y - s.target[1], rx = obs[0] - s.b[b].p.x, ry = obs[1] - s.b[b].p.y;
        const double dt = sqrt(dx * dx + dy * dy), dr = sqrt(rx * rx + ry * ry);
        sum += dt;
        inside += dt < MJS_PP_TARGET_RADIUS;
        nearest = fmin(nearest, dr);
      }
      const bool success = inside == nb;
      double reward = p.reward_type == MJS_REW_SPARSE ? (double)inside : (-sum / nb - MJS_PP_NEAREST_COEF * nearest) * MJS_PP_REWARD_SCALE;
      double discount = success ? 0.0 : 1.0;
      terminate = success || s.episode_step >= (double)p.max_episode_steps;
      if (bad) { reward = 0; discount = 0; terminate = true; }
      if (s.time >= p.time_limit) terminate = true;
      const int ncon = ncon_now;
      const int fault = (bad ? MJS_FAULT_BAD_STATE : 0) | ((flags & FLAG_IK_FAILED) ? MJS_FAULT_IK_FAILED : 0) | (info.rows_active ? MJS_FAULT_LIMIT_COLDSTART : 0) |
                        (info.unsupported ? MJS_FAULT_UNSUPPORTED_CONTACT : 0);
      const bool terminated = terminate && discount == 0.0, truncated = terminate && discount > 0.0;
      if (valid) {
        store_world(p, i, s);
        p.flags[i] = (uint8_t)((flags & FLAG_IK_FAILED) | (terminate ? FLAG_RESET_PENDING : 0));
        write_outputs<OBS_DIM>(p, i, obs, reward, discount, terminate ? MJS_STEP_LAST : MJS_STEP_MID, terminated, truncated, success, fault, ncon);
      }
    }
}
}

}  // namespace MJS_PP_NS
