// mjsim.hip — libmjsim.so: C ABI (include/mjsim.h) + kernel launches. gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mjsim.h"
#include "mjs_kernel_common.h"
#include "mjs_pointmass.h"
#include "mjs_reach.h"
#include "mjs_button.h"
#include "mjs_gripper14.h"
#include "mjs_push.h"
#include "mjs_render.h"
#include <cmath>

struct mjs_handle {
  mjs_config cfg;
  // render primitive list of the CURRENT state already built on this stream? (the visual configs render two cameras per
  // control step; reset / step / rollout / set_state invalidate)
  bool prims_valid = false;
  void* prims_stream = nullptr;
  int state_dim, obs_dim, act_dim;
  double* state;     // [state_dim][N]
  uint8_t* flags;    // [N]
  uint32_t* rng_mt;  // [624][N]
  int32_t* rng_pos;  // [N]
  unsigned long long* stamps;  // diagnostic builds only
  float* prims;                // [N][nprim][PRIM_FLOATS] render primitive list (robot scenes)
  float4* bg_ray = nullptr;    // [bg_H * bg_W] scene-camera ray table (rend::Background), built at the first render of a size
  uint32_t* bg_rgb = nullptr;
  int bg_H = 0, bg_W = 0;
  float* cams;                 // [N][12] wrist-camera poses (Button-Push)
  double* ws = nullptr;        // [rr::WS_ROWS][N] contact workspace of the general constraint stage (Robot-Reach, Button-Push)
  int epoch = 0;               // parity of the next step launch (FLAG_EPOCH)
  double* ws14 = nullptr;      // [bg::WS_DOUBLES][N] constraint rows of the articulated-gripper kernel (mjs_gripper14.h)
  std::string err;
};

namespace {

thread_local std::string g_err;

int fail(mjs_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}
int hip_fail(mjs_handle* h, hipError_t e, const char* what) {
  return fail(h, MJS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, expr)                                   \
  do {                                                     \
    hipError_t e_ = (expr);                                \
    if (e_ != hipSuccess) return hip_fail(h, e_, #expr);   \
  } while (0)

// Select the handle's device for the duration of one ABI call and restore the caller's current device afterwards
// (torch's notion of the current device must not change behind its back). Same device: one hipGetDevice, no switch.
struct DeviceGuard {
  int prev = -1;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

double default_time_limit(int task) {
  if (task == MJS_TASK_POINTMASS_REACH) return MJS_PM_MAX_CONTROL_STEPS * MJS_PM_CONTROL_DT;  // mujoco_sim/__init__.py:21,28
  if (task == MJS_TASK_BUTTON_PUSH) return MJS_BP_MAX_CONTROL_STEPS * MJS_RR_CONTROL_DT;      // mujoco_sim/__init__.py:45-49
  if (task == MJS_TASK_PLANAR_PUSH) return 1e300;  // scripts/sb3/planar_push.py:72: no Environment time limit, the task counts steps
  return MJS_RR_MAX_CONTROL_STEPS * MJS_RR_CONTROL_DT;                                       // BASELINE config 3
}
int default_reward(int task) { return task == MJS_TASK_POINTMASS_REACH ? MJS_REW_DENSE_BIASED_NEG_DISTANCE : MJS_REW_DENSE_NEG_DISTANCE; }

__global__ __launch_bounds__(64) void seed_kernel(DevRng rng, uint32_t base_seed, int offset) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rng.N) return;
  rng_seed_lane(rng, i, base_seed + (uint32_t)(offset + i));
}

// Pointmass bookkeeping starts at 1.0 at construction (point_reach.py:112-113); flags = reset pending
// host_pending: the byte of an env that waits for its next-step reset, as the host writes it (reset-groups handles: with the
// parity OPPOSITE to the next step launch's, i.e. "ended in an earlier launch", mjs_kernel_common.h)
__global__ void fill_row_kernel(double* row, int N, double value) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) row[i] = value;
}
__global__ void init_kernel(double* state, uint8_t* flags, int N, int task, uint8_t host_pending) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  flags[i] = host_pending;
  if (task == MJS_TASK_POINTMASS_REACH) {
    state[(size_t)pm::S_DIST * N + i] = 1.0;
    state[(size_t)pm::S_PREV * N + i] = 1.0;
  }
}

__global__ void get_state_kernel(const double* state, const uint8_t* flags, double* out, int N, int S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < S; k++) out[(size_t)k * N + i] = state[(size_t)k * N + i];
  // the launch-parity protocol of the reset-groups variant is not part of a checkpoint: a restored env is neither "reset in
  // this launch" nor marked with a parity of the handle it came from
  out[(size_t)S * N + i] = (double)(uint8_t)(flags[i] & ~(FLAG_FRESH | FLAG_EPOCH));
}
__global__ void set_state_kernel(double* state, uint8_t* flags, const double* in, int N, int S, int task, uint8_t host_pending) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < S; k++) state[(size_t)k * N + i] = in[(size_t)k * N + i];
  uint8_t f = (uint8_t)((uint8_t)in[(size_t)S * N + i] & ~(FLAG_FRESH | FLAG_EPOCH));
  if (f & FLAG_RESET_PENDING) f = (uint8_t)(f | host_pending);
  if (task == MJS_TASK_ROBOT_REACH || task == MJS_TASK_BUTTON_PUSH) {
    // the caller may have edited the joints: the carried cos / sin rows are kept when they belong to the given q (a checkpoint
    // resumes bit for bit) and rewritten with exact values when they do not
    const int first = task == MJS_TASK_ROBOT_REACH ? rr::S_CS : bp::S_CS;
    double q[rr::NJ];
    for (int j = 0; j < rr::NJ; j++) {
      q[j] = in[(size_t)(rr::S_Q + j) * N + i];
      double sn, cs;
      sincos(q[j], &sn, &cs);
      const double c_in = in[(size_t)(first + j) * N + i], s_in = in[(size_t)(first + rr::NJ + j) * N + i];
      if (!(fabs(c_in - cs) < 1e-11 && fabs(s_in - sn) < 1e-11)) {
        state[(size_t)(first + j) * N + i] = cs;
        state[(size_t)(first + rr::NJ + j) * N + i] = sn;
      }
    }
    if (task == MJS_TASK_ROBOT_REACH)  // FLAG_CLEAR is a function of the joints
      f = (uint8_t)((f & ~FLAG_CLEAR) | (rr::config_is_clear(q) ? FLAG_CLEAR : 0));
  }
  flags[i] = f;
}

__global__ void debug_ik_kernel(const double* T, const double* guess, double* q, uint8_t* ok, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rr::Aff A;
  for (int k = 0; k < 9; k++) A.r[k] = T[(size_t)i * 12 + k];
  for (int k = 0; k < 3; k++) A.t[k] = T[(size_t)i * 12 + 9 + k];
  double g[6], out[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 6; k++) g[k] = guess[(size_t)i * 6 + k];
  bool found = rr::ik_closest(A, g, out);
  for (int k = 0; k < 6; k++) q[(size_t)i * 6 + k] = out[k];
  ok[i] = found;
}

__global__ void tcp_to_joints_kernel(const double* pos, const double* guess, double* q, uint8_t* ok, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p3[3], g[6], out[6];
  for (int k = 0; k < 3; k++) p3[k] = pos[(size_t)i * 3 + k];
  for (int k = 0; k < 6; k++) { g[k] = guess[(size_t)i * 6 + k]; out[k] = g[k]; }
  bool found = rr::tcp_pose_to_joints(p3, g, out);
  for (int k = 0; k < 6; k++) q[(size_t)i * 6 + k] = found ? out[k] : g[k];
  ok[i] = found;
}

// A/B knob (MJS_LDS_PAD = bytes of unused dynamic LDS per workgroup of the Robot-Reach / Button-Push step kernels): a pad above half of the
// CU's 160 KB leaves room for ONE workgroup per CU, which tells whether the dispatcher packs the few workgroups of a 4096-env launch onto
// shared CUs (profiles/r04_e_workgroup_placement.txt).
size_t lds_pad() {
  static const size_t pad = [] { const char* ev = std::getenv("MJS_LDS_PAD"); return ev ? (size_t)std::atol(ev) : (size_t)0; }();
  return pad;
}
// Button-Push: envs per workgroup = 64. Smaller groups were measured (MJS_BP_EPG, profiles/r04_d_button_group_size.txt) and LOSE: with
// 16 envs per group the steady state goes 53 -> 84 us per launch and the desynchronised case stays at 260 us - four times the
// wavefronts share SIMDs (the dispatcher packs workgroups onto CUs), which costs more than the smaller groups save.
int button_epg(const mjs_handle* h) {
  if (h->cfg.task != MJS_TASK_BUTTON_PUSH) return 64;
  if (const char* ev = std::getenv("MJS_BP_EPG")) { const int v = std::atoi(ev); if (v == 8 || v == 16 || v == 32 || v == 64) return v; }
  return 64;
}
int push_prog_row(const mjs_handle* h) { return h->cfg.n_objects > MJS_PP_FAST_OBJECTS ? pp5::PROG_ROW : pp::PROG_ROW; }
bool articulated(const mjs_handle* h) { return h->cfg.task == MJS_TASK_BUTTON_PUSH && h->cfg.gripper_model == MJS_GRIPPER_ARTICULATED; }
bool uses_reset_groups(const mjs_handle* h) {
  if (articulated(h)) return false;  // one kernel shape: a workgroup resets its own envs
  return (h->cfg.task == MJS_TASK_ROBOT_REACH || h->cfg.task == MJS_TASK_BUTTON_PUSH) && h->cfg.kernel_variant == MJS_VARIANT_RESET_GROUPS &&
         h->cfg.autoreset == MJS_AUTORESET_NEXT_STEP;
}
uint8_t host_pending_byte(const mjs_handle* h) {
  return (uint8_t)(FLAG_RESET_PENDING | ((uses_reset_groups(h) && !h->epoch) ? FLAG_EPOCH : 0));
}

KernelParams make_params(const mjs_handle* h, const double* actions, const uint8_t* mask, const mjs_outputs* out) {
  KernelParams p;
  p.N = h->cfg.num_envs;
  p.reward_type = h->cfg.reward_type;
  p.autoreset = h->cfg.autoreset;
  p.terminate_on_success = h->cfg.terminate_on_success;
  p.action_type = h->cfg.action_type;
  p.button_disturbances = h->cfg.button_disturbances;
  p.n_objects = h->cfg.n_objects;
  p.max_episode_steps = h->cfg.max_episode_steps;
  p.block_shape = h->cfg.block_shape;
  p.epoch = h->epoch;
  p.reset_groups = uses_reset_groups(h);
  p.epg = button_epg(h);
  // Planar-Push: next episodes are prepared ahead of time by prefetch workgroups (mjs_push_impl.h NEXT_ROW0); variant 1 keeps
  // round 3's behaviour (the settle steps of a reset run inside the step launch) for A/B measurements
  p.prefetch = h->cfg.task == MJS_TASK_PLANAR_PUSH && h->cfg.autoreset == MJS_AUTORESET_NEXT_STEP && h->cfg.kernel_variant != MJS_VARIANT_SINGLE_WAVE;
  p.time_limit = h->cfg.time_limit;
  p.state = h->state;
  p.flags = h->flags;
  p.ws = h->ws;
  p.rng = DevRng{h->rng_mt, h->rng_pos, h->cfg.num_envs};
  p.actions = actions;
  p.reset_mask = mask;
  if (out) p.out = *out; else std::memset(&p.out, 0, sizeof p.out);
  p.stamps = h->stamps;
  return p;
}

// envs per workgroup of the articulated-gripper kernel. The kernel is bound by instruction DELIVERY (its substep streams ~150 KB of
// code through the instruction cache), so fewer, fuller wavefronts win as soon as every CU has one: 16 envs per workgroup (what
// the LDS holds: 16 x 9.8 KB) from 4096 envs on - 11.7 ms per launch against 14.9 with 4 (profiles/r04_b_*)
int bg_epw_for(int n) {
  if (const char* ev = std::getenv("MJS_BG_EPW")) { const int v = std::atoi(ev); if (v >= 1 && v <= 16) return v; }  // A/B experiments
  return n < 1024 ? 4 : n < 4096 ? 8 : 16;
}

constexpr int BLOCK = 64;  // one wavefront per workgroup: N/64 workgroups spread over the CUs
inline dim3 grid_for(int n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

template <bool IS_RESET>
int launch(mjs_handle* h, const KernelParams& p, hipStream_t s) {
  bool flip_epoch = false;
  if (h->cfg.task == MJS_TASK_POINTMASS_REACH) pm::kernel<IS_RESET><<<grid_for(p.N), BLOCK, 0, s>>>(p);
  else if (h->cfg.task == MJS_TASK_PLANAR_PUSH && h->cfg.n_objects <= MJS_PP_FAST_OBJECTS)
    pp::kernel<IS_RESET><<<dim3((unsigned)(((!IS_RESET && p.prefetch) ? 2 : 1) * ((p.N + pp::EPW * pp::WAVES - 1) / (pp::EPW * pp::WAVES)))), BLOCK * pp::WAVES, pp::LDS_BYTES, s>>>(p);
  else if (h->cfg.task == MJS_TASK_PLANAR_PUSH)  // 3..5 blocks: the 5-slot instance, its cooperative workspace needs the large-LDS opt-in (mjs_create)
    pp5::kernel<IS_RESET><<<dim3((unsigned)(((!IS_RESET && p.prefetch) ? 2 : 1) * ((p.N + pp5::EPW * pp5::WAVES - 1) / (pp5::EPW * pp5::WAVES)))), BLOCK * pp5::WAVES, pp5::LDS_BYTES, s>>>(p);
  else if (articulated(h)) {
    const int epw = bg_epw_for(p.N);
    bg::kernel<IS_RESET><<<dim3((unsigned)((p.N + epw - 1) / epw)), IS_RESET ? BLOCK : bg::ROLES * BLOCK, (size_t)epw * sizeof(bg::Env), s>>>(p, h->ws14, epw);  // stepping: bg::ROLES wavefronts share the envs
  }
  else if (h->cfg.task == MJS_TASK_BUTTON_PUSH) {
    const dim3 bgrid((unsigned)((p.N + p.epg - 1) / p.epg));
    if (IS_RESET || h->cfg.kernel_variant == MJS_VARIANT_SINGLE_WAVE) bp::kernel<IS_RESET, 1><<<bgrid, BLOCK, 0, s>>>(p);
    else if (p.reset_groups) {  // MJS_VARIANT_RESET_GROUPS: the second half of the grid resets the envs whose episode ended
      bp::kernel<false, 2><<<dim3(2 * bgrid.x), 2 * BLOCK, 0, s>>>(p);
      flip_epoch = true;
    }
    else bp::kernel<false, 2><<<bgrid, 2 * BLOCK, lds_pad(), s>>>(p);
  }
  else if (IS_RESET || h->cfg.kernel_variant == MJS_VARIANT_SINGLE_WAVE) rr::kernel<IS_RESET, 1><<<grid_for(p.N), BLOCK, 0, s>>>(p);
  // Two shapes of the same step (results equal to rounding): an IK wave + two role-specialised dynamics waves per 64 envs
  // is the faster one while every workgroup has a CU to itself (the launch is one workgroup's latency: -10 % at 4096 envs);
  // past 16384 envs per GPU the chip is full and the third, mostly idle wave costs a SIMD: the two-role kernel of round 1
  // then has twice the throughput (profiles/r02_j_batch_size_scaling.txt: 65536 envs 435 vs 822 M env-steps/s).
  else if (h->cfg.kernel_variant == MJS_VARIANT_TWO_ROLES || (h->cfg.kernel_variant == MJS_VARIANT_DEFAULT && p.N > 16384))
    rr::kernel<false, 2><<<grid_for(p.N), 2 * BLOCK, 0, s>>>(p);
  else if (p.reset_groups) {
    // MJS_VARIANT_RESET_GROUPS, for episodes that end at different times: the second half of the grid are reset workgroups
    // (one per 64 envs, on other CUs: an env whose episode ended is reset there while the first half steps the others: 54 -> 38 us
    // per launch with 1 % of the envs ending in every step); the launch parity tells a freshly reset env from one that waits.
    // Not the default for synchronous episodes: the 64 extra workgroups cost 0.9 us per launch at 4096 envs.
    rr::kernel3<0><<<dim3(2 * grid_for(p.N).x), 3 * BLOCK, 0, s>>>(p);
    flip_epoch = true;
  }
  else rr::kernel3<0><<<grid_for(p.N), 3 * BLOCK, lds_pad(), s>>>(p);
  HIP_TRY(h, hipGetLastError());
  if (flip_epoch) h->epoch ^= 1;  // only a launch that was accepted consumed its parity
  return MJS_OK;
}

}  // namespace

extern "C" {

#ifndef MJS_SOURCE_HASH
#define MJS_SOURCE_HASH "unhashed"  // the in-tree build (mujoco_sim_amd/_native.py) passes -DMJS_SOURCE_HASH=<sha256 of csrc/ + include/>
#endif
#define MJS_STR2(x) #x
#define MJS_STR(x) MJS_STR2(x)
const char* mjs_version(void) { return "mjsim-hip 0.3 (gfx950, abi " MJS_STR(MJS_ABI_VERSION) ") src=" MJS_SOURCE_HASH; }

int mjs_obs_dim(int task) {
  return task == MJS_TASK_POINTMASS_REACH ? pm::OBS_DIM : task == MJS_TASK_ROBOT_REACH ? rr::OBS_DIM : task == MJS_TASK_BUTTON_PUSH ? bp::OBS_DIM : task == MJS_TASK_PLANAR_PUSH ? pp::OBS_DIM : -1;
}
int mjs_action_dim_for(int task, int action_type) {
  if (task == MJS_TASK_BUTTON_PUSH) return action_type == MJS_ACTION_ABS_EEF ? bp::ACT_DIM_EEF : action_type == MJS_ACTION_ABS_JOINT ? bp::ACT_DIM_JOINT : -1;
  return task == MJS_TASK_POINTMASS_REACH ? pm::ACT_DIM : task == MJS_TASK_ROBOT_REACH ? rr::ACT_DIM : task == MJS_TASK_PLANAR_PUSH ? pp::ACT_DIM : -1;
}
int mjs_action_dim(int task) { return mjs_action_dim_for(task, MJS_ACTION_ABS_JOINT); }
int mjs_state_dim(int task) {
  return task == MJS_TASK_POINTMASS_REACH ? pm::STATE_DIM + 1 : task == MJS_TASK_ROBOT_REACH ? rr::STATE_DIM + 1 : task == MJS_TASK_BUTTON_PUSH ? bp::STATE_DIM + 1 : task == MJS_TASK_PLANAR_PUSH ? pp::FULL_STATE_DIM + 1 : -1;
}
int mjs_env_obs_dim(const mjs_handle* h) { return h ? h->obs_dim : -1; }
int mjs_env_state_dim(const mjs_handle* h) { return h ? h->state_dim + 1 : -1; }
int mjs_substeps(int task) {
  return task == MJS_TASK_POINTMASS_REACH ? MJS_PM_NSUB : (task == MJS_TASK_ROBOT_REACH || task == MJS_TASK_BUTTON_PUSH || task == MJS_TASK_PLANAR_PUSH) ? MJS_RR_NSUB : -1;
}

int mjs_algorithmic_bytes_per_env_step(int task) {
  // state read + state written + flag byte r/w + action + obs + reward + discount + 5 flag bytes + ncon
  const int out_fixed = 8 + 8 + 5 + 4;
  if (task == MJS_TASK_POINTMASS_REACH)
    return 8 * pm::STATE_DIM /*R*/ + 8 * (pm::STATE_DIM - 2) /*W: target unchanged*/ + 2 + 8 * pm::ACT_DIM + 8 * pm::OBS_DIM + out_fixed;
  if (task == MJS_TASK_ROBOT_REACH)
    return 8 * rr::HOT_ROWS_READ + 8 * rr::HOT_ROWS_WRITTEN + 2 + 8 * rr::ACT_DIM + 8 * rr::OBS_DIM + out_fixed;  // q v time target + carried cos / sin; the qacc_warmstart rows are only touched by the robust path
  if (task == MJS_TASK_PLANAR_PUSH)  // everything but the target is rewritten
    return 8 * pp::STATE_DIM /*R*/ + 8 * (pp::STATE_DIM - 3) /*W*/ + 2 + 8 * pp::ACT_DIM + 8 * pp::OBS_DIM + out_fixed;
  if (task == MJS_TASK_BUTTON_PUSH)
    return 8 * bp::HOT_ROWS_READ + 8 * bp::HOT_ROWS_WRITTEN + 2 + 8 * bp::ACT_DIM_JOINT + 8 * bp::OBS_DIM + out_fixed;
  return -1;
}

const char* mjs_last_error(const mjs_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int mjs_create(const mjs_config* cfg, mjs_handle** out) {
  if (!cfg || !out) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(mjs_config))
    return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: mjs_config.struct_size does not match this library's header (abi " MJS_STR(MJS_ABI_VERSION) "): the caller was built against another include/mjsim.h");
  if (cfg->task != MJS_TASK_POINTMASS_REACH && cfg->task != MJS_TASK_ROBOT_REACH && cfg->task != MJS_TASK_BUTTON_PUSH && cfg->task != MJS_TASK_PLANAR_PUSH)
    return fail(nullptr, MJS_ERR_UNSUPPORTED, "mjs_create: unknown task id");
  if (mjs_action_dim_for(cfg->task, cfg->action_type) < 0) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: bad action_type");
  if (cfg->gripper_model != MJS_GRIPPER_REDUCED && cfg->gripper_model != MJS_GRIPPER_ARTICULATED) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: bad gripper_model");
  if (cfg->gripper_model == MJS_GRIPPER_ARTICULATED && cfg->task != MJS_TASK_BUTTON_PUSH)
    return fail(nullptr, MJS_ERR_UNSUPPORTED, "mjs_create: the articulated 2F-85 is built for Button-Push only (Robot-Reach keeps the rigid payload, BASELINE config 3 \"UR5e 6-DoF\")");
  if (cfg->task == MJS_TASK_PLANAR_PUSH && cfg->n_objects > MJS_PP_MAX_OBJECTS) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: n_objects exceeds MJS_PP_MAX_OBJECTS");
  if (cfg->task == MJS_TASK_PLANAR_PUSH && cfg->block_shape != MJS_BLOCKS_MESH && cfg->block_shape != MJS_BLOCKS_BOX)
    return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: bad block_shape");
  if (cfg->num_envs <= 0) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: num_envs must be positive");
  if (cfg->autoreset < MJS_AUTORESET_NEXT_STEP || cfg->autoreset > MJS_AUTORESET_DISABLED)
    return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: bad autoreset mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MJS_ERR_NO_DEVICE, "mjs_create: no HIP device visible (this library has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_create: device ordinal out of range");
  mjs_handle* h = new (std::nothrow) mjs_handle();
  if (!h) return fail(nullptr, MJS_ERR_ALLOC, "mjs_create: out of host memory");
  h->cfg = *cfg;
  if (h->cfg.reward_type < 0) h->cfg.reward_type = default_reward(cfg->task);
  if (h->cfg.n_objects <= 0) h->cfg.n_objects = MJS_PP_FAST_OBJECTS;
  if (h->cfg.max_episode_steps <= 0) h->cfg.max_episode_steps = MJS_PP_MAX_CONTROL_STEPS;
  if (!(h->cfg.time_limit > 0)) h->cfg.time_limit = default_time_limit(cfg->task);
  h->state_dim = mjs_state_dim(cfg->task) - 1;
  h->obs_dim = mjs_obs_dim(cfg->task);
  if (cfg->task == MJS_TASK_PLANAR_PUSH && h->cfg.n_objects > MJS_PP_FAST_OBJECTS) { h->state_dim = pp5::FULL_STATE_DIM; h->obs_dim = pp5::OBS_DIM; }
  if (cfg->task == MJS_TASK_BUTTON_PUSH && h->cfg.gripper_model == MJS_GRIPPER_ARTICULATED) h->state_dim = bg::STATE_DIM;
  h->act_dim = mjs_action_dim_for(cfg->task, cfg->action_type);
  h->state = nullptr; h->flags = nullptr; h->rng_mt = nullptr; h->rng_pos = nullptr; h->stamps = nullptr; h->prims = nullptr; h->cams = nullptr;
  const size_t N = (size_t)cfg->num_envs;
  DeviceGuard dev_(cfg->device);
  hipError_t e = dev_.err;
  if (e == hipSuccess && cfg->task == MJS_TASK_PLANAR_PUSH) {
    // > 64 KB of dynamic LDS per workgroup is an opt-in (gfx950 has 160 KB per CU): cooperative workspaces + hull tables
#ifndef MJS_STAMPS  // the diagnostic build's phase counters live in the env slots: its 5-slot instance does not fit and is not used
    static_assert(pp5::LDS_BYTES <= 160 * 1024, "cooperative workspace exceeds the CU's LDS");
#endif
    static_assert(pp::LDS_BYTES <= 160 * 1024, "cooperative workspace exceeds the CU's LDS");
    if (h->cfg.n_objects > MJS_PP_FAST_OBJECTS) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pp5::kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp5::LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pp5::kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp5::LDS_BYTES);
    } else {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pp::kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pp::kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp::LDS_BYTES);
    }
  }
  if (e == hipSuccess) e = hipMalloc(&h->state, sizeof(double) * h->state_dim * N);
  if (e == hipSuccess) e = hipMalloc(&h->flags, N);
  if (e == hipSuccess) e = hipMalloc(&h->rng_mt, sizeof(uint32_t) * 624 * N);
  if (e == hipSuccess) e = hipMalloc(&h->rng_pos, sizeof(int32_t) * N);
  if (e == hipSuccess) e = hipMemset(h->state, 0, sizeof(double) * h->state_dim * N);
  if (e == hipSuccess && cfg->task == MJS_TASK_ROBOT_REACH)  // render primitive list (no allocation in launch paths)
    e = hipMalloc(&h->prims, sizeof(float) * rend::PRIM_FLOATS * rend::RR_NPRIM * N);
  if (e == hipSuccess && cfg->task == MJS_TASK_BUTTON_PUSH) e = hipMalloc(&h->prims, sizeof(float) * rend::PRIM_FLOATS * rend::BP_NPRIM * N);
  if (e == hipSuccess && cfg->task == MJS_TASK_PLANAR_PUSH) e = hipMalloc(&h->prims, sizeof(float) * rend::PRIM_FLOATS * rend::PP_NPRIM * N);
  if (e == hipSuccess && cfg->task == MJS_TASK_BUTTON_PUSH) e = hipMalloc(&h->cams, sizeof(float) * 12 * N);
  if (e == hipSuccess && articulated(h)) e = hipMalloc(&h->ws14, sizeof(double) * (size_t)bg::WS_DOUBLES * N);  // 15 KB per env: every row of every env
  if (e == hipSuccess && cfg->task != MJS_TASK_POINTMASS_REACH)  // 3.4 KB per env, touched only by lanes with an arm geom in the floor
    e = hipMalloc(&h->ws, sizeof(double) * rr::WS_ROWS * N);
#ifdef MJS_STAMPS
  if (e == hipSuccess) e = hipMalloc(&h->stamps, sizeof(unsigned long long) * 16 * N);  // one slot block per workgroup, at most N workgroups
  if (e == hipSuccess) e = hipMemset(h->stamps, 0, sizeof(unsigned long long) * 16 * N);
#endif
  if (e != hipSuccess) {
    int rc = hip_fail(nullptr, e, "mjs_create: device allocation");
    mjs_destroy(h);
    return rc;
  }
  if (lds_pad() > 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rr::kernel3<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bp::kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad());
  }
  if (articulated(h)) {  // > 64 KB of dynamic LDS per workgroup is an opt-in (16 envs per workgroup at large batches)
    static_assert(16 * sizeof(bg::Env) <= 160 * 1024, "16 envs per workgroup fit the CU's LDS");
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bg::kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * sizeof(bg::Env)));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bg::kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * sizeof(bg::Env)));
    if (e != hipSuccess) {
      int rc = hip_fail(nullptr, e, "mjs_create: LDS opt-in");
      mjs_destroy(h);
      return rc;
    }
  }
  if (articulated(h)) {  // the compiled 14-body model: a __device__ global of this device, the same constants for every handle
    static const bg::Model host_model = [] { bg::Model m; std::memset(&m, 0, sizeof m); bg::build_model(m); return m; }();
    e = hipMemcpyToSymbol(HIP_SYMBOL(bg::g_model), &host_model, sizeof host_model);
    if (e != hipSuccess) {
      int rc = hip_fail(nullptr, e, "mjs_create: model upload");
      mjs_destroy(h);
      return rc;
    }
  }
  init_kernel<<<grid_for((int)N), BLOCK>>>(h->state, h->flags, (int)N, cfg->task, host_pending_byte(h));
  if (cfg->task == MJS_TASK_PLANAR_PUSH) fill_row_kernel<<<grid_for((int)N), BLOCK>>>(h->state + (size_t)push_prog_row(h) * N, (int)N, -1.0);  // no next episode prepared
  seed_kernel<<<grid_for((int)N), BLOCK>>>(DevRng{h->rng_mt, h->rng_pos, (int)N}, 0u, cfg->env_index_offset);
  e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    int rc = hip_fail(nullptr, e, "mjs_create: init kernels");
    mjs_destroy(h);
    return rc;
  }
  *out = h;
  return MJS_OK;
}

void mjs_destroy(mjs_handle* h) {
  if (!h) return;
#ifdef MJS_STAMPS
  if (h->stamps) {  // diagnostic build: mean phase lengths (shader cycles) of the LAST step launch
    const int W = (h->cfg.num_envs + 63) / 64;
    unsigned long long* host = new unsigned long long[16 * (size_t)h->cfg.num_envs];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, h->stamps, sizeof(unsigned long long) * 16 * (size_t)h->cfg.num_envs, hipMemcpyDeviceToHost);
    double d[5] = {0, 0, 0, 0, 0}, e[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < W; w++) {
      for (int k = 0; k < 5; k++) d[k] += (double)(host[16 * w + k + 1] - host[16 * w + k]) / W;
      for (int k = 0; k < 5; k++) e[k] += (double)(host[16 * w + 8 + k + 1] - host[16 * w + 8 + k]) / W;
    }
    if (h->cfg.task == MJS_TASK_PLANAR_PUSH) {
      double c[16] = {0}, worst[16] = {0}, worst_total = 0;
      int hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int WG = (h->cfg.num_envs + pp::EPW * pp::WAVES - 1) / (pp::EPW * pp::WAVES);
      for (int w = 0; w < WG; w++) {
        double tot = 0;
        for (int k = 0; k < 5; k++) tot += (double)host[16 * w + k];
        for (int k = 0; k < 16; k++) c[k] += (double)host[16 * w + k] / WG;
        if (tot > worst_total) { worst_total = tot; for (int k = 0; k < 16; k++) worst[k] = (double)host[16 * w + k]; }
        const int ns = (int)host[16 * w + 6];
        hist[ns == 0 ? 0 : ns <= 5 ? 1 : ns <= 10 ? 2 : ns <= 20 ? 3 : ns <= 30 ? 4 : ns <= 40 ? 5 : ns <= 60 ? 6 : ns <= 80 ? 7 : 8]++;
      }
      std::fprintf(stderr, "[MJS_STAMPS] planar-push, wave 0 of each workgroup, cycles per control step, MEAN: detect %.0f | arm dynamics %.0f | decoupled solves %.0f | cooperative coupled %.0f (of which the owner lane's publish %.0f) | integrate %.0f\n", c[0], c[1], c[2], c[3], c[5], c[4]);
      std::fprintf(stderr, "[MJS_STAMPS] cooperative solves per wavefront and control step: mean %.2f (%.2f Newton iterations each), slowest wavefront %.0f solves / %.0f iterations; wavefronts with 0 | 1-5 | 6-10 | 11-20 | 21-30 | 31-40 | 41-60 | 61-80 | >80 solves: %d %d %d %d %d %d %d %d %d\n", c[6], c[6] > 0 ? c[7] / c[6] : 0.0, worst[6], worst[7], hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
      std::fprintf(stderr, "[MJS_STAMPS] slowest wavefront, inside its cooperative solves: rows %.0f | init (M^-1 f, J a, update) %.0f | gradient+Hessian %.0f | Cholesky+solves %.0f | M v, J v %.0f | line search %.0f | update+gradient norm %.0f | J^T f %.0f\n", worst[15], worst[8], worst[9], worst[10], worst[11], worst[12], worst[13], worst[14]);
      std::fprintf(stderr, "[MJS_STAMPS] SLOWEST workgroup: detect %.0f | arm dynamics %.0f | decoupled solves %.0f | cooperative coupled %.0f (publish %.0f) | integrate %.0f\n", worst[0], worst[1], worst[2], worst[3], worst[5], worst[4]);
    }
    double own = 0;  // Robot-Reach kernel3: stamp 6 = wave 0 has finished its own prologue (load, sincos, substep 0) and reaches the IK barrier
    for (int w = 0; w < W; w++) own += (double)(host[16 * w + 6] - host[16 * w + 0]) / W;
    std::fprintf(stderr, "[MJS_STAMPS] cycles: load+IK %.0f (of which wave 0's own load + sincos + substep 0: %.0f) | substeps %.0f | fk+obs %.0f | contacts %.0f | store %.0f\n", d[0], own, d[1], d[2], d[3], d[4]);
    std::fprintf(stderr, "[MJS_STAMPS] role-0 substep 10: CRBA+factor+invert %.0f | barrier %.0f | apply inverse+publish+barrier %.0f | integrate %.0f\n", e[0], e[1], e[2], e[3]);
    delete[] host;
    (void)hipFree(h->stamps);
  }
#endif
  if (h->state) (void)hipFree(h->state);
  if (h->flags) (void)hipFree(h->flags);
  if (h->rng_mt) (void)hipFree(h->rng_mt);
  if (h->rng_pos) (void)hipFree(h->rng_pos);
  if (h->prims) (void)hipFree(h->prims);
  if (h->bg_ray) (void)hipFree(h->bg_ray);
  if (h->bg_rgb) (void)hipFree(h->bg_rgb);
  if (h->cams) (void)hipFree(h->cams);
  if (h->ws) (void)hipFree(h->ws);
  if (h->ws14) (void)hipFree(h->ws14);
  delete h;
}

int mjs_seed(mjs_handle* h, uint32_t base_seed, void* stream) {
  if (!h) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_seed: null handle");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  seed_kernel<<<grid_for(h->cfg.num_envs), BLOCK, 0, (hipStream_t)stream>>>(DevRng{h->rng_mt, h->rng_pos, h->cfg.num_envs}, base_seed,
                                                                           h->cfg.env_index_offset);
  if (h->cfg.task == MJS_TASK_PLANAR_PUSH)  // episodes prepared from the old streams are void
    fill_row_kernel<<<grid_for(h->cfg.num_envs), BLOCK, 0, (hipStream_t)stream>>>(h->state + (size_t)push_prog_row(h) * h->cfg.num_envs, h->cfg.num_envs, -1.0);
  HIP_TRY(h, hipGetLastError());
  return MJS_OK;
}

int mjs_reset(mjs_handle* h, const uint8_t* mask_dev, const mjs_outputs* out, void* stream) {
  if (!h) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_reset: null handle");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  h->prims_valid = false;
  return launch<true>(h, make_params(h, nullptr, mask_dev, out), (hipStream_t)stream);
}

int mjs_step(mjs_handle* h, const double* actions_dev, const mjs_outputs* out, void* stream) {
  if (!h) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_step: null handle");
  if (!actions_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_step: actions_dev is null");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  h->prims_valid = false;
  return launch<false>(h, make_params(h, actions_dev, nullptr, out), (hipStream_t)stream);
}

int mjs_rollout(mjs_handle* h, const double* actions_dev, int32_t T, const mjs_outputs* out, void* stream) {
  if (!h) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_rollout: null handle");
  if (!actions_dev || T < 0) return fail(h, MJS_ERR_INVALID_ARG, "mjs_rollout: bad arguments");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  h->prims_valid = false;
  const size_t N = (size_t)h->cfg.num_envs;
  for (int32_t t = 0; t < T; t++) {
    mjs_outputs o;
    std::memset(&o, 0, sizeof o);
    if (out) {
      o = *out;
      if (o.obs) o.obs += (size_t)t * N * h->obs_dim;
      if (o.terminal_obs) o.terminal_obs += (size_t)t * N * h->obs_dim;
      if (o.reward) o.reward += t * N;
      if (o.discount) o.discount += t * N;
      if (o.terminated) o.terminated += t * N;
      if (o.truncated) o.truncated += t * N;
      if (o.is_success) o.is_success += t * N;
      if (o.step_type) o.step_type += t * N;
      if (o.fault) o.fault += t * N;
      if (o.ncon) o.ncon += t * N;
    }
    int rc = launch<false>(h, make_params(h, actions_dev + (size_t)t * N * h->act_dim, nullptr, &o), (hipStream_t)stream);
    if (rc != MJS_OK) return rc;
  }
  return MJS_OK;
}

int mjs_render(mjs_handle* h, int32_t camera, int32_t height, int32_t width, uint8_t* rgb_dev, void* stream) {
  if (!h || !rgb_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_render: null argument");
  const int task = h->cfg.task;
  const bool wrist = camera == MJS_CAMERA_WRIST;
  if ((camera != MJS_CAMERA_SCENE && !(wrist && task == MJS_TASK_BUTTON_PUSH)) || height <= 0 || width <= 0)
    return fail(h, MJS_ERR_INVALID_ARG, "mjs_render: bad camera or size");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  rend::RenderParams p;
  p.N = h->cfg.num_envs; p.H = height; p.W = width; p.state = h->state; p.out = rgb_dev;
  p.env_cams = nullptr; p.nprim = 0;
  // camera frame from the MuJoCo quaternion (w,x,y,z): columns of R are the local x (right), y (up), z (back) axes
  const double* q = task == MJS_TASK_POINTMASS_REACH ? MJS_PM_CAM_QUAT : task == MJS_TASK_BUTTON_PUSH ? MJS_BP_CAM_QUAT : MJS_RR_CAM_QUAT;
  const double* cpos = task == MJS_TASK_POINTMASS_REACH ? MJS_PM_CAM_POS : task == MJS_TASK_BUTTON_PUSH ? MJS_BP_CAM_POS : MJS_RR_CAM_POS;
  const double fovy = wrist ? MJS_WCAM_FOVY : task == MJS_TASK_POINTMASS_REACH ? MJS_PM_CAM_FOVY : task == MJS_TASK_BUTTON_PUSH ? MJS_BP_CAM_FOVY : MJS_RR_CAM_FOVY;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                       2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  for (int k = 0; k < 3; k++) {
    p.cam.pos[k] = (float)cpos[k];
    p.cam.right[k] = (float)R[3 * k + 0];
    p.cam.up[k] = (float)R[3 * k + 1];
    p.cam.back[k] = (float)R[3 * k + 2];
  }
  p.cam.tan_half = (float)std::tan(fovy * 3.14159265358979323846 / 360.0);
  dim3 grid((unsigned)((height * width + 255) / 256), (unsigned)p.N);
  const int tiles = ((height + 7) / 8) * ((width + 7) / 8);  // robot scenes: one wavefront per 8x8 tile
  dim3 tile_grid((unsigned)((tiles + 3) / 4), (unsigned)p.N);
  // robot scenes: the rectangle walk (per-primitive pixel rectangles, colours staged in 16 KB of LDS) with one workgroup per
  // env image, or per band of rows when the image has more than 4096 pixels; the fixed scene cameras add the table of
  // env-independent rays and floor colours, computed once per size. Sizes that are not multiples of 8 (and kernel_variant 1)
  // keep the 8x8-tile walk. Ladder: profiles/r02_g_render_ladder.txt.
  int band_rows = height;
  if (height * width > rend::RECT_WALK_MAX_PIXELS) band_rows = (rend::RECT_WALK_MAX_PIXELS / width) & ~7;
  const bool rect_walk = height % 8 == 0 && width % 8 == 0 && band_rows >= 8 && h->cfg.kernel_variant != MJS_VARIANT_SINGLE_WAVE;
  const unsigned nbands = rect_walk ? (unsigned)((height + band_rows - 1) / band_rows) : 1u;
  if (rect_walk && !wrist && (h->bg_H != height || h->bg_W != width)) {  // the scene camera of a handle never moves: keyed by size only
    if (h->bg_ray) (void)hipFree(h->bg_ray);
    if (h->bg_rgb) (void)hipFree(h->bg_rgb);
    h->bg_ray = nullptr; h->bg_rgb = nullptr; h->bg_H = h->bg_W = 0;
    HIP_TRY(h, hipMalloc(&h->bg_ray, sizeof(float4) * height * width));
    HIP_TRY(h, hipMalloc(&h->bg_rgb, sizeof(uint32_t) * height * width));
    rend::scene_background_kernel<<<(unsigned)((height * width + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, h->bg_ray, h->bg_rgb);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));  // once per size: later renders may come on another stream
    h->bg_H = height; h->bg_W = width;
  }
  auto robot_scene = [&]() {
    const dim3 rw_grid((unsigned)p.N, nbands);
    if (rect_walk && p.env_cams) rend::robot_scene_rect_walk_kernel<false><<<rw_grid, 256, 0, (hipStream_t)stream>>>(p, h->prims, rend::Background{nullptr, nullptr}, band_rows);
    else if (rect_walk) rend::robot_scene_rect_walk_kernel<true><<<rw_grid, 256, 0, (hipStream_t)stream>>>(p, h->prims, rend::Background{h->bg_ray, h->bg_rgb}, band_rows);
    else rend::robot_scene_kernel<<<tile_grid, 256, 0, (hipStream_t)stream>>>(p, h->prims);
  };
  const bool fresh = h->prims_valid && h->prims_stream == stream;  // same state, same stream: the list is still good
  h->prims_valid = true;
  h->prims_stream = stream;
  if (task == MJS_TASK_POINTMASS_REACH) {
    rend::pointmass_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
  } else if (task == MJS_TASK_BUTTON_PUSH) {
    if (!fresh) rend::button_prims_kernel<<<grid_for(p.N), BLOCK, 0, (hipStream_t)stream>>>(h->state, h->flags, h->prims, h->cams, p.N);
    p.nprim = rend::BP_NPRIM;
    if (wrist) p.env_cams = h->cams;
    robot_scene();
  } else if (task == MJS_TASK_PLANAR_PUSH) {  // robot_planar_push.py:45,66: the FRONT_TILTED camera of Robot-Reach
    const int nslots = MJS_PP_OBJECT_SLOTS(h->cfg.n_objects);
    if (!fresh) rend::push_prims_kernel<<<grid_for(p.N), BLOCK, 0, (hipStream_t)stream>>>(h->state, h->prims, p.N, h->cfg.n_objects, nslots);
    p.nprim = rend::ARM_NREC + 1 + nslots;
    robot_scene();
  } else {
    if (!fresh) rend::reach_prims_kernel<<<grid_for(p.N), BLOCK, 0, (hipStream_t)stream>>>(h->state, h->prims, p.N);
    p.nprim = rend::RR_NPRIM;
    robot_scene();
  }
  HIP_TRY(h, hipGetLastError());
  return MJS_OK;
}

int mjs_debug_ur5e_ik(const double* T_dev, const double* guess_dev, double* q_dev, uint8_t* ok_dev, int32_t n, void* stream) {
  if (!T_dev || !guess_dev || !q_dev || !ok_dev || n < 0) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_debug_ur5e_ik: bad argument");
  debug_ik_kernel<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(T_dev, guess_dev, q_dev, ok_dev, n);
  HIP_TRY(nullptr, hipGetLastError());
  return MJS_OK;
}

int mjs_ur5e_tcp_to_joints(const double* tcp_pos_dev, const double* guess_dev, double* q_dev, uint8_t* ok_dev, int32_t n, void* stream) {
  if (!tcp_pos_dev || !guess_dev || !q_dev || !ok_dev || n < 0) return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_ur5e_tcp_to_joints: bad argument");
  tcp_to_joints_kernel<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(tcp_pos_dev, guess_dev, q_dev, ok_dev, n);
  HIP_TRY(nullptr, hipGetLastError());
  return MJS_OK;
}

int mjs_ur5e_robot_run(double* state_dev, const double* target_dev, int32_t command, double param, int32_t n_substeps, int32_t eef, double dt,
                       double* tcp_pose_out_dev, uint8_t* status_dev, int32_t n, void* stream) {
  if (!state_dev || n < 0 || n_substeps < 0 || command < MJS_UR_CMD_NONE || command > MJS_UR_CMD_SERVOJ || (command != MJS_UR_CMD_NONE && !target_dev) ||
      (eef != MJS_UR_EEF_NONE && eef != MJS_UR_EEF_GRIPPER) || !(dt > 0) || (command != MJS_UR_CMD_NONE && !(param > 0)))
    return fail(nullptr, MJS_ERR_INVALID_ARG, "mjs_ur5e_robot_run: bad argument");
  static_assert(rr::UR_STATE == MJS_UR_STATE, "state block layout");
  rr::ur_robot_kernel<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(state_dev, target_dev, command, param, n_substeps, eef, dt, tcp_pose_out_dev, status_dev, n);
  HIP_TRY(nullptr, hipGetLastError());
  return MJS_OK;
}

int mjs_get_state(mjs_handle* h, double* state_dev, void* stream) {
  if (!h || !state_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_get_state: null argument");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  get_state_kernel<<<grid_for(h->cfg.num_envs), BLOCK, 0, (hipStream_t)stream>>>(h->state, h->flags, state_dev, h->cfg.num_envs, h->state_dim);
  HIP_TRY(h, hipGetLastError());
  return MJS_OK;
}

int mjs_set_state(mjs_handle* h, const double* state_dev, void* stream) {
  if (!h || !state_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_set_state: null argument");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  h->prims_valid = false;
  set_state_kernel<<<grid_for(h->cfg.num_envs), BLOCK, 0, (hipStream_t)stream>>>(h->state, h->flags, state_dev, h->cfg.num_envs, h->state_dim, h->cfg.task, host_pending_byte(h));
  HIP_TRY(h, hipGetLastError());
  return MJS_OK;
}

int mjs_get_rng_state(mjs_handle* h, uint32_t* mt_dev, int32_t* pos_dev, void* stream) {
  if (!h || !mt_dev || !pos_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_get_rng_state: null argument");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  const size_t N = (size_t)h->cfg.num_envs;
  HIP_TRY(h, hipMemcpyAsync(mt_dev, h->rng_mt, sizeof(uint32_t) * 624 * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(h, hipMemcpyAsync(pos_dev, h->rng_pos, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MJS_OK;
}

int mjs_set_rng_state(mjs_handle* h, const uint32_t* mt_dev, const int32_t* pos_dev, void* stream) {
  if (!h || !mt_dev || !pos_dev) return fail(h, MJS_ERR_INVALID_ARG, "mjs_set_rng_state: null argument");
  DeviceGuard dev_(h->cfg.device);
  HIP_TRY(h, dev_.err);
  const size_t N = (size_t)h->cfg.num_envs;
  HIP_TRY(h, hipMemcpyAsync(h->rng_mt, mt_dev, sizeof(uint32_t) * 624 * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(h, hipMemcpyAsync(h->rng_pos, pos_dev, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MJS_OK;
}

}  // extern "C"
