// mfma_valu_mix.hip -- how much vector-ALU / transcendental / LDS work hides under v_mfma_f32_32x32x2_f32 (64 cycles per SIMD)?
// (developer tool, GPU box only)   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_mix.hip -o tools/build/mfma_valu_mix
//
// Decides the structure of the 64-row tile kernel: if K independent VALU instructions between two dependent MFMAs of ONE wave cost
// nothing up to K = K0, a wave can carry the tanh / loss / dZ2 arithmetic of one 32-row half in the shadow of the other half's MFMAs.
//   test A  one wave per SIMD: loop { MFMA; K x filler } for K = 0..24, filler = v_fma_f32 / v_exp_f32 / ds_read_b128 / ds_write_b32
//   test B  two waves per SIMD (512-thread workgroup): waves 0-3 MFMA only, waves 4-7 filler only; each group's own duration
//           against its duration alone (what the round-1 builder saw as "non-GEMM phases crawl next to a dense MFMA stream")
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { F_FMA = 0, F_EXP = 1, F_DSREAD = 2, F_DSWRITE = 3, F_MIXED = 4 };

template <int KIND>
__device__ __forceinline__ void filler(float (&x)[8], int j, float a, float b, const float* lds_r, float* lds_w, f32x4& sink) {
  float& v = x[j & 7];
  if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v) : "v"(a), "v"(b));
  if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
  if (KIND == F_DSREAD) {
    f32x4 t;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lds_r + 16u * (j & 7)));
    sink = t;   // consumed after the loop only
  }
  if (KIND == F_DSWRITE) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)lds_w + 4u * (j & 7)), "v"(v));
  if (KIND == F_MIXED) {   // roughly a tanh epilogue element: 6 fma-class, 1 exp, 1 rcp, 1 LDS store per 9 slots
    const int m = j % 9;
    if (m == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
    else if (m == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
    else if (m == 8) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)lds_w + 4u * (j & 7)), "v"(v));
    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v) : "v"(a), "v"(b));
  }
}

// test A: one wave per SIMD, K fillers after every MFMA
template <int KIND, int K>
__global__ void __launch_bounds__(256, 1) inwave(float* out, unsigned long long* clk, int iters) {
  __shared__ float lds[256 * 64];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 256 * 64; i += 256) lds[i] = 1.0f;
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.001f * (lane + j);
  const float a = 0.999f, b = 1e-4f;
  const float* lr = lds + threadIdx.x * 64;
  float* lw = lds + threadIdx.x * 64 + 32;
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K; ++j) filler<KIND>(x, u * K + j, a, b, lr, lw, sink);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long w1 = wall_clock64();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = w1 - w0;
}

// test B: waves 0-3 MFMA only (mfma_iters x 8 MFMAs), waves 4-7 filler only (fill_iters x 64 fillers); either count may be 0
template <int KIND>
__global__ void __launch_bounds__(512, 1) crosswave(float* out, unsigned long long* clk, int mfma_iters, int fill_iters, int prio_fill) {
  __shared__ float lds[512 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 512 * 32; i += 512) lds[i] = 1.0f;
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.001f * (lane + j);
  const float a = 0.999f, b = 1e-4f;
  const float* lr = lds + threadIdx.x * 32;
  float* lw = lds + threadIdx.x * 32 + 16;
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  if (wave >= 4 && prio_fill) __builtin_amdgcn_s_setprio(3);
  const unsigned long long w0 = wall_clock64();
  if (wave < 4) {
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    for (int it = 0; it < fill_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) filler<KIND>(x, j, a, b, lr, lw, sink);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long w1 = wall_clock64();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) clk[blockIdx.x * 8 + wave] = w1 - w0;
}

static float* g_out;
static unsigned long long* g_clk;

template <int KIND, int K>
void run_inwave(const char* what) {
  const int wgs = 256, iters = 400;   // 3200 MFMAs per wave
  std::vector<unsigned long long> h(wgs * 4);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((inwave<KIND, K>), dim3(wgs), dim3(256), 0, 0, g_out, g_clk, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), g_clk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = std::min(best, (double)h[h.size() / 2]);
  }
  const double ns_per_mfma = best * 10.0 / (iters * 8.0);
  printf("A %-8s K=%2d  %6.1f ns per MFMA slot  (+%5.1f ns over K=0 budget 30.5)  -> %5.2f ns per filler\n", what, K, ns_per_mfma, ns_per_mfma - 30.5,
         K ? ns_per_mfma / K : 0.0);
}

template <int KIND>
void run_cross(const char* what, int mfma_iters, int fill_iters, int prio) {
  const int wgs = 256;
  std::vector<unsigned long long> h(wgs * 8);
  hipMemset(g_clk, 0, sizeof(unsigned long long) * h.size());
  hipLaunchKernelGGL((crosswave<KIND>), dim3(wgs), dim3(512), 0, 0, g_out, g_clk, mfma_iters, fill_iters, prio);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), g_clk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> m, f;
  for (int b = 0; b < wgs; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? m : f).push_back((double)h[b * 8 + w]);
  std::sort(m.begin(), m.end());
  std::sort(f.begin(), f.end());
  printf("B %-8s mfma_iters %5d fill_iters %5d prio %d :  MFMA waves %8.2f us (%5.1f ns/MFMA)   filler waves %8.2f us (%6.2f ns/filler)\n", what, mfma_iters,
         fill_iters, prio, m[m.size() / 2] / 100.0, mfma_iters ? m[m.size() / 2] * 10.0 / (mfma_iters * 8.0) : 0.0, f[f.size() / 2] / 100.0,
         fill_iters ? f[f.size() / 2] * 10.0 / (fill_iters * 64.0) : 0.0);
}

template <int KIND>
void sweep_inwave(const char* what) {
  run_inwave<KIND, 0>(what);
  run_inwave<KIND, 1>(what);
  run_inwave<KIND, 2>(what);
  run_inwave<KIND, 4>(what);
  run_inwave<KIND, 6>(what);
  run_inwave<KIND, 8>(what);
  run_inwave<KIND, 10>(what);
  run_inwave<KIND, 12>(what);
  run_inwave<KIND, 16>(what);
  run_inwave<KIND, 24>(what);
}

int main() {
  hipMalloc(&g_out, sizeof(float) * 256 * 512);
  hipMalloc(&g_clk, sizeof(unsigned long long) * 256 * 8);
  // warm the clock
  for (int i = 0; i < 20; ++i) run_cross<F_FMA>("warm", 400, 0, 0);
  sweep_inwave<F_FMA>("v_fma");
  sweep_inwave<F_EXP>("v_exp");
  sweep_inwave<F_DSREAD>("ds_rd128");
  sweep_inwave<F_DSWRITE>("ds_wr32");
  sweep_inwave<F_MIXED>("tanhmix");
  // cross-wave: alone, then together
  run_cross<F_FMA>("v_fma", 400, 0, 0);
  run_cross<F_FMA>("v_fma", 0, 400, 0);
  run_cross<F_FMA>("v_fma", 400, 400, 0);
  run_cross<F_FMA>("v_fma", 400, 400, 1);
  run_cross<F_FMA>("v_fma", 400, 100, 0);
  run_cross<F_MIXED>("tanhmix", 0, 400, 0);
  run_cross<F_MIXED>("tanhmix", 400, 400, 0);
  run_cross<F_MIXED>("tanhmix", 400, 400, 1);
  run_cross<F_DSREAD>("ds_rd128", 0, 400, 0);
  run_cross<F_DSREAD>("ds_rd128", 400, 400, 0);
  return 0;
}
