// mfma_load_mix.hip -- does the fp32 MFMA pipe slow down when its operands come straight from global loads?
// (developer tool, GPU box only)   hipcc -O3 --offload-arch=gfx950 tools/mfma_load_mix.hip -o gpurun_out/mfma_load_mix
// Mimics the reduction loop of gemm_tn_frag_kernel (2 register sets x 4 k8-groups, 4 accumulators per wave, 256-thread
// workgroups, 2 per CU) with the operand source as the only variable:
//   mode 0  no loads, operands are loop-invariant registers
//   mode 1  double-buffered global loads, MFMAs read the loaded registers directly
//   mode 2  same loads, the A operand goes through a v_mov first
//   mode 3  same loads, kept alive but not consumed: MFMAs use loop-invariant registers
//   mode 4  as mode 1 with s_setprio 1 around the load block
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SG = 4;

template <int MODE>
__global__ void __launch_bounds__(256, 2) mix(const float* __restrict__ src, size_t region_floats, float* out, unsigned long long* clk, int groups) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 2; ++c)
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
  f32x4 ra[2][SG][2], rb[2][SG][2];
  const f32x4 ka = {0.5f + lane * 1e-3f, 0.25f, -0.5f, 1.f}, kb = {1.f - lane * 2e-3f, 0.75f, 0.125f, -1.f};
  // every workgroup walks its own slice of the region; one group = 2 KB of A per wave pair + 2 KB of B
  const size_t wg_span = (size_t)groups * 2048;                       // floats per workgroup stream
  const float* ap = src + ((size_t)blockIdx.x * 2 * wg_span) % region_floats + wave * 512 + lane * 4;
  const float* bp = ap + wg_span;
#define LOAD(set, g0)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < SG; ++u) {                                                      \
    const int gg_ = min((g0) + u, groups - 1);                                                          \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) ra[set][u][r] = *reinterpret_cast<const f32x4*>(ap + (size_t)gg_ * 2048 + r * 256); \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) rb[set][u][c] = *reinterpret_cast<const f32x4*>(bp + (size_t)gg_ * 2048 + c * 256); \
  }
#define MFMA(set)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < SG; ++u)                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      float a_[2], b_[2];                                                                               \
      _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                   \
        if (MODE == 0 || MODE == 3) { a_[r] = ka[i]; b_[r] = kb[i]; }                                   \
        else { a_[r] = ra[set][u][r][i]; b_[r] = rb[set][u][r][i]; }                                    \
        if (MODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(a_[r]) : "v"(ra[set][u][r][i]));          \
      }                                                                                                 \
      if (MODE == 2) asm volatile("s_nop 1");                                                           \
      _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                     \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                   \
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[r], b_[c], acc[r][c], 0, 0, 0);           \
    }                                                                                                   \
  if (MODE == 3) {                                                                                      \
    _Pragma("unroll") for (int u = 0; u < SG; ++u)                                                      \
      _Pragma("unroll") for (int r = 0; r < 2; ++r) asm volatile("" ::"v"(ra[set][u][r]), "v"(rb[set][u][r])); \
  }
  const unsigned long long w0 = wall_clock64();
  if (MODE != 0) { LOAD(0, 0) }
  for (int g = 0; g < groups; g += 2 * SG) {
    if (MODE == 4) __builtin_amdgcn_s_setprio(1);
    if (MODE != 0) { LOAD(1, g + SG) }
    if (MODE == 4) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 4) __builtin_amdgcn_s_setprio(1);
    if (MODE != 0) { LOAD(0, g + 2 * SG) }
    if (MODE == 4) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    MFMA(1)
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 2; ++c)
      for (int e = 0; e < 16; ++e) s += acc[r][c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = w1 - w0;
}

template <int MODE>
void run(const float* src, size_t region_floats, float* out, unsigned long long* clk, int groups, int wgs, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  std::vector<unsigned long long> h(wgs);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix<MODE>, dim3(wgs), dim3(256), 0, 0, src, region_floats, out, clk, groups);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double mfma_per_simd = (double)groups * 16 * (wgs / 256.0);
  printf("mode %d (%-44s) region %4zu MB  groups %4d  event %7.2f us  in-kernel median %7.2f max %7.2f us  -> %5.1f ns per MFMA per SIMD (event)\n", MODE, what,
         region_floats * 4 >> 20, groups, best * 1e3, h[wgs / 2] / 100.0, h[wgs - 1] / 100.0, best * 1e6 / mfma_per_simd);
}

int main() {
  const size_t max_floats = (size_t)256 << 20;   // 1 GB
  float *src, *out;
  unsigned long long* clk;
  hipMalloc(&src, (max_floats + ((size_t)16 << 20)) * 4);   // slack: a stream may start at the end of the region
  hipMemset(src, 0, (max_floats + ((size_t)16 << 20)) * 4);
  hipMalloc(&out, sizeof(float) * 1024 * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * 1024);
  for (int groups : {32, 128}) {
    for (size_t region_mb : {(size_t)2, (size_t)64, (size_t)1024}) {
      const size_t rf = region_mb << 18;
      run<0>(src, rf, out, clk, groups, 512, "no loads");
      run<1>(src, rf, out, clk, groups, 512, "operands straight from the loaded registers");
      run<2>(src, rf, out, clk, groups, 512, "A operand through v_mov");
      run<3>(src, rf, out, clk, groups, 512, "loads in flight, operands loop-invariant");
      run<4>(src, rf, out, clk, groups, 512, "as mode 1, s_setprio 1 around the loads");
    }
  }
  return 0;
}
