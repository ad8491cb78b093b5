// mjs_kernel_common.h — launch parameters shared by the task kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mjsim.h"
#include "../../include/mjs_scene_spec.h"
#include "../../include/mjs_block_hulls.h"
#include "mjs_dev_math.h"
#include "mjs_dev_rng.h"

// per-env flag byte
// (4, 8: Button-Push switch bits.) FLAG_WARM_VALID: the state's qacc_warmstart rows hold the solver acceleration of the last
// Physics.step() (written by the robust path and by resets; a row-free step leaves them stale and clears the bit).
// FLAG_CLEAR (Robot-Reach): every arm collision geom was >= rr::CLEAR_MARGIN above the floor in the step's final configuration.
// FLAG_FRESH / FLAG_EPOCH (Robot-Reach's default step kernel): the env was reset by a reset workgroup of the step launch whose
// parity FLAG_EPOCH holds; a stepping wavefront of THAT launch which reads the byte after the reset leaves the lane alone, the
// next launch (other parity) steps it and clears both bits (rr::kernel3). In such launches a NEW "reset pending" carries the parity
// of the launch whose stepping workgroup set it (FLAG_EPOCH without FLAG_FRESH, pending_mark): a reset workgroup acts only on a
// pending env of the OTHER parity (pending_is_due), i.e. one whose episode ended in an earlier launch - the reset workgroups are
// the grid's second half and may start after a stepping workgroup of the same launch has retired (> 256 resident workgroups).
// Host-written bytes are normalised to that rule (init_kernel, mjs_set_state); mjs_get_state exports neither bit.
enum { FLAG_RESET_PENDING = 1, FLAG_IK_FAILED = 2, FLAG_WARM_VALID = 16, FLAG_CLEAR = 32, FLAG_FRESH = 64, FLAG_EPOCH = 128 };

struct KernelParams {
  int N;
  int reward_type;
  int autoreset;
  int terminate_on_success;
  int action_type;
  int button_disturbances;
  int n_objects;          // Planar-Push
  int max_episode_steps;  // Planar-Push
  int block_shape;        // Planar-Push: MJS_BLOCKS_MESH / MJS_BLOCKS_BOX
  int epoch;              // parity of this step launch (0 / 1, toggled per launch): see FLAG_EPOCH
  int reset_groups;       // Robot-Reach kernel3: 1 = the grid's second half are reset workgroups, 0 = a workgroup resets its own envs
  int epg;                // Button-Push: envs per workgroup (lanes >= epg leave at once): 64, or 16 while the chip has CUs to spare (mjsim.hip)
  int prefetch;           // Planar-Push: 1 = the grid's second half are PREFETCH workgroups that prepare every env's next episode
                          // (draws + settle steps, a few substeps per launch) in the state's second slot (mjs_push_impl.h)
  double time_limit;
  double* state;    // [state_dim][N] struct-of-arrays float64
  uint8_t* flags;   // [N]
  double* ws;       // [rr::WS_ROWS][N] contact workspace of the robot scenes' general constraint stage (mjs_arm_stage.h)
  DevRng rng;
  const double* actions;      // [N, A] (step) or nullptr (reset)
  const uint8_t* reset_mask;  // reset kernel only; nullptr = all
  mjs_outputs out;
  unsigned long long* stamps;  // diagnostic builds only (-DMJS_STAMPS): [workgroup][16] shader-clock stamps
};

__device__ __forceinline__ uint8_t pending_mark(const KernelParams& p) {
  return (uint8_t)(FLAG_RESET_PENDING | ((p.reset_groups && p.epoch) ? FLAG_EPOCH : 0));
}
__device__ __forceinline__ bool pending_is_due(const KernelParams& p, uint8_t flags) {
  return !(flags & FLAG_FRESH) && ((flags & FLAG_EPOCH) != 0) != (p.epoch != 0);
}

// In-kernel phase stamps for a SEPARATE diagnostic build (never in the shipped library): the real
// kernel executes no stamp. s_memtime + lgkmcnt(0) as one asm statement (guide section 7).
#ifdef MJS_STAMPS
#define MJS_STAMP(p, slot)                                                                  \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if ((p).stamps && threadIdx.x == 0) (p).stamps[(size_t)blockIdx.x * 16 + (slot)] = t_; \
  } while (0)
#else
#define MJS_STAMP(p, slot) do { } while (0)
#endif

template <int OBS>
__device__ __forceinline__ void write_outputs(const KernelParams& p, int i, const double* obs, double reward, double discount,
                                              int step_type, bool terminated, bool truncated, bool success, int fault, int ncon) {
  if (p.out.obs) {
#pragma unroll
    for (int k = 0; k < OBS; k++) p.out.obs[(size_t)i * OBS + k] = obs[k];
  }
  if (p.out.reward) p.out.reward[i] = reward;
  if (p.out.discount) p.out.discount[i] = discount;
  if (p.out.terminated) p.out.terminated[i] = terminated;
  if (p.out.truncated) p.out.truncated[i] = truncated;
  if (p.out.is_success) p.out.is_success[i] = success;
  if (p.out.step_type) p.out.step_type[i] = (uint8_t)step_type;
  if (p.out.fault) p.out.fault[i] = (uint8_t)fault;
  if (p.out.ncon) p.out.ncon[i] = ncon;
}

// MuJoCo impedance curve d(r) (mj_makeImpedance / getimpedance) for the default solimp
__device__ __forceinline__ double impedance_default(double pos_minus_margin) {
  const double d0 = MJS_SOLIMP_D0, d1 = MJS_SOLIMP_DWIDTH, width = MJS_SOLIMP_WIDTH, mid = MJS_SOLIMP_MIDPOINT;
  double x = pos_minus_margin / width;
  if (x < 0) x = -x;
  if (x >= 1 || x <= 0) return (x >= 1) ? d1 : d0;
  double y;
  if (x <= mid) {
    double a = 1 / mid;  // power 2: 1/mid^(p-1)
    y = a * (x * x);
  } else {
    double b = 1 / (1 - mid);
    y = 1 - b * ((1 - x) * (1 - x));
  }
  return d0 + y * (d1 - d0);
}
