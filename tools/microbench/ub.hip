// ub.hip — micro-benchmarks behind the Robot-Reach multi-wavefront design: what one LDS exchange
// + workgroup barrier costs between 2 / 4 / 8 wavefronts of a workgroup, the FP64 FMA issue interval and dependent
// latency of one wavefront alone on its SIMD, the cost of a DPP cross-lane move of a double, and the latency of the FP64
// MFMA shapes. Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/ub tools/microbench/ub.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// ---- A: NV doubles per lane exchanged through LDS + NB barriers per iteration, W wavefronts
template <int W, int NV, int NB>
__global__ __launch_bounds__(64 * W) void k_exchange(unsigned long long* out, double* sink, int iters) {
  __shared__ double buf[W][NV][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[NV];
  for (int k = 0; k < NV; k++) v[k] = lane * 0.001 + k;
  __syncthreads();
  unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
    for (int k = 0; k < NV; k++) buf[wave][k][lane] = v[k];
    __syncthreads();
    for (int k = 0; k < NV; k++) v[k] = fma(buf[(wave + 1) % W][k][lane], 1.0000001, 1e-9);
    if (NB == 2) __syncthreads();
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  double s = 0;
  for (int k = 0; k < NV; k++) s += v[k];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- B: CH independent dependent chains of FP64 FMAs, one wavefront
template <int CH>
__global__ __launch_bounds__(64) void k_fma(unsigned long long* out, double* sink, int iters) {
  double a[CH];
  for (int k = 0; k < CH; k++) a[k] = threadIdx.x * 1e-3 + k;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int k = 0; k < CH; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
    }
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  double s = 0;
  for (int k = 0; k < CH; k++) s += a[k];
  sink[blockIdx.x * 64 + threadIdx.x] = s;
}

// ---- B2: the same with only the low `active` lanes enabled (does a partly filled wavefront issue FP64 faster?)
__global__ __launch_bounds__(64) void k_fma_masked(unsigned long long* out, double* sink, int iters, int active) {
  double a[4];
  for (int k = 0; k < 4; k++) a[k] = threadIdx.x * 1e-3 + k;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0 = now();
  if ((int)threadIdx.x < active) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
      }
    }
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}

// ---- C: cross-lane move of a double by two v_mov_b32_dpp (quad_perm rotate) feeding an FMA
template <int CH>
__global__ __launch_bounds__(64) void k_dpp(unsigned long long* out, double* sink, int iters) {
  double a[CH];
  for (int k = 0; k < CH; k++) a[k] = threadIdx.x * 1e-3 + k;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int k = 0; k < CH; k++) {
        double b;
        asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %1, %3 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"
                     : "=&v"(((int*)&b)[0]), "=&v"(((int*)&b)[1]) : "v"(((int*)&a[k])[0]), "v"(((int*)&a[k])[1]));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[k]) : "v"(b), "v"(m), "v"(c));
      }
    }
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  double s = 0;
  for (int k = 0; k < CH; k++) s += a[k];
  sink[blockIdx.x * 64 + threadIdx.x] = s;
}

// ---- D: dependent chain of FP64 MFMAs (result fed back as the B operand / accumulator)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_mfma16(unsigned long long* out, double* sink, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-3;
  d4 acc = {0, 0, 0, 0};
  unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(64) void k_mfma4(unsigned long long* out, double* sink, int iters, int feed_b) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-3, acc = 0;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
      if (feed_b) b = acc;
    }
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc + b;
}
// layout probe of v_mfma_f64_4x4x4: A = one-hot at lane la, B = one-hot at lane lb -> which lanes of D are non-zero
__global__ __launch_bounds__(64) void k_mfma4_probe(double* out, int la, int lb) {
  double a = (int)threadIdx.x == la ? 1.0 : 0.0, b = (int)threadIdx.x == lb ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[threadIdx.x] = d;
}

template <typename F>
static double run(F launch, int blocks, int iters, int per_iter) {
  unsigned long long* d_out;
  double* d_sink;
  hipMalloc(&d_out, sizeof(unsigned long long) * blocks);
  hipMalloc(&d_sink, sizeof(double) * blocks * 1024);
  launch(d_out, d_sink, 8);
  hipDeviceSynchronize();
  launch(d_out, d_sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto x : h) s += (double)x;
  hipFree(d_out);
  hipFree(d_sink);
  return s / blocks / iters / per_iter;
}

int main() {
  const int B = 64, IT = 2000;
#define EX(W, NV, NB) printf("exchange W=%d waves, %2d doubles/lane, %d barrier(s)/iter: %7.1f cycles/iter\n", W, NV, NB, \
    run([&](unsigned long long* o, double* s, int it) { k_exchange<W, NV, NB><<<B, 64 * W>>>(o, s, it); }, B, IT, 1))
  EX(2, 6, 1); EX(2, 6, 2); EX(4, 6, 1); EX(4, 6, 2); EX(8, 6, 1); EX(4, 12, 1); EX(4, 24, 1); EX(2, 1, 1); EX(4, 1, 1);
#define FM(CH) printf("FP64 FMA, %d independent chain(s), one wave per SIMD: %6.2f cycles/instruction\n", CH, \
    run([&](unsigned long long* o, double* s, int it) { k_fma<CH><<<B, 64>>>(o, s, it); }, B, IT, 16 * CH))
  FM(1); FM(2); FM(3); FM(4); FM(8);
  for (int act : {64, 32, 16, 4})
    printf("FP64 FMA, 4 chains, %2d active lanes: %6.2f cycles/instruction\n", act,
           run([&](unsigned long long* o, double* s, int it) { k_fma_masked<<<B, 64>>>(o, s, it, act); }, B, IT, 64));
#define DP(CH) printf("2x v_mov_b32_dpp + FMA, %d chain(s): %6.2f cycles per (move+FMA)\n", CH, \
    run([&](unsigned long long* o, double* s, int it) { k_dpp<CH><<<B, 64>>>(o, s, it); }, B, IT, 16 * CH))
  DP(1); DP(4);
  printf("v_mfma_f64_16x16x4 dependent (acc) chain: %6.2f cycles/instruction\n", run([&](unsigned long long* o, double* s, int it) { k_mfma16<<<B, 64>>>(o, s, it); }, B, IT, 16));
  printf("v_mfma_f64_4x4x4 acc chain: %6.2f cycles/instruction\n", run([&](unsigned long long* o, double* s, int it) { k_mfma4<<<B, 64>>>(o, s, it, 0); }, B, IT, 16));
  printf("v_mfma_f64_4x4x4 result fed back as B: %6.2f cycles/instruction\n", run([&](unsigned long long* o, double* s, int it) { k_mfma4<<<B, 64>>>(o, s, it, 1); }, B, IT, 16));
  double* d;
  hipMalloc(&d, 64 * sizeof(double));
  double h[64];
  for (int la : {0, 1, 4, 5, 16, 21}) for (int lb : {0, 1, 4, 5, 16, 21}) {
    k_mfma4_probe<<<1, 64>>>(d, la, lb);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mfma4x4x4 probe A@lane%2d B@lane%2d -> D nonzero at lanes:", la, lb);
    for (int l = 0; l < 64; l++) if (h[l] != 0) printf(" %d", l);
    printf("\n");
  }
  return 0;
}
