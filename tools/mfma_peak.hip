// mfma_peak.hip -- what the v_mfma_f32_32x32x2_f32 pipe of this MI355X actually sustains (developer tool, GPU box only).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
// Every CU runs `waves` waves that issue back-to-back MFMAs from registers (two independent accumulator chains, random
// operands), for roughly `us` microseconds; reports TFLOP/s from HIP events and the shader clock from s_memtime vs the
// 100 MHz wall clock, for bursts as short as one GEMM phase of the fused kernel and for long runs (DVFS settles).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC independent accumulator chains per wave (2 = a 32x64 wave tile, 4 = a 64x64 wave tile)
template <int NACC>
__global__ void __launch_bounds__(512) burn_n(float* out, unsigned long long* clk, int iters, float seed) {
  f32x16 a[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int e = 0; e < 16; ++e) a[k][e] = 0.f;
  float x[4], y[4];
  for (int k = 0; k < 4; ++k) {
    x[k] = seed + threadIdx.x * 1e-3f + k;
    y[k] = 1.f - threadIdx.x * 2e-3f - k;
  }
  const unsigned long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) a[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(u + k) & 3], y[k & 3], a[k], 0, 0, 0);
  }
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < NACC; ++k)
    for (int e = 0; e < 16; ++e) s += a[k][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[2 * blockIdx.x] = w1 - w0;
}

__global__ void __launch_bounds__(512) burn(float* out, unsigned long long* clk, int iters, float seed) {
  f32x16 a0, a1;
  for (int e = 0; e < 16; ++e) {
    a0[e] = 0.f;
    a1[e] = 0.f;
  }
  float x = seed + threadIdx.x * 1e-3f, y = 1.f - threadIdx.x * 2e-3f;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    }
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = w1 - w0;
    clk[2 * blockIdx.x + 1] = c1 - c0;
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * 1024 * 512);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    for (int iters : {16, 64, 256, 4096}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(burn, dim3(256), dim3(threads), 0, 0, out, clk, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost);
        double wall = 0, cyc = 0;
        for (int b = 0; b < 256; ++b) {
          wall += h[2 * b];
          cyc += h[2 * b + 1];
        }
        wall /= 256;
        cyc /= 256;
        const double flops = 256.0 * (threads / 64) * iters * 16.0 * 32 * 32 * 2 * 2;
        const double mfma_per_simd = (threads / 256.0) * iters * 16.0;
        if (rep == 2)
          printf("waves/CU %d  mfma/SIMD %6.0f  event %8.2f us  %6.1f TFLOP/s | in-kernel %8.2f us  %6.1f TFLOP/s  s_memtime/wall = %.3f ticks per 10 ns  -> %.0f ns per MFMA per SIMD\n",
                 threads / 64, mfma_per_simd, ms * 1e3, flops / (ms * 1e-3) / 1e12, wall / 100.0, flops / (wall * 1e-8) / 1e12, cyc / wall,
                 wall * 10.0 / mfma_per_simd);
      }
    }
  }
  // accumulator-count sweep at a GEMM-phase-sized burst (512 MFMAs per SIMD)
  for (int threads : {256, 512}) {
    for (int nacc : {1, 2, 4, 8}) {
      const int iters = threads == 256 ? 32 : 16;
      for (int rep = 0; rep < 3; ++rep) {
        if (nacc == 1) hipLaunchKernelGGL(burn_n<1>, dim3(256), dim3(threads), 0, 0, out, clk, iters, 0.5f);
        if (nacc == 2) hipLaunchKernelGGL(burn_n<2>, dim3(256), dim3(threads), 0, 0, out, clk, iters, 0.5f);
        if (nacc == 4) hipLaunchKernelGGL(burn_n<4>, dim3(256), dim3(threads), 0, 0, out, clk, iters, 0.5f);
        if (nacc == 8) hipLaunchKernelGGL(burn_n<8>, dim3(256), dim3(threads), 0, 0, out, clk, iters, 0.5f);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(512);
      hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost);
      double wall = 0;
      for (int b = 0; b < 256; ++b) wall += h[2 * b];
      wall /= 256;
      const double mfma_per_simd = (threads / 256.0) * iters * 16.0;
      printf("accumulators/wave %d  waves/CU %d  mfma/SIMD %4.0f  in-kernel %7.2f us  -> %.1f ns per MFMA per SIMD\n", nacc, threads / 64, mfma_per_simd,
             wall / 100.0, wall * 10.0 / mfma_per_simd);
    }
  }
  return 0;
}
