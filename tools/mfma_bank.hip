// mfma_bank.hip -- does v_mfma_f32_32x32x2_f32 issue slower when its A and B operands sit in the same VGPR bank (index mod 4)?
// (developer tool, GPU box only)   hipcc -O3 --offload-arch=gfx950 tools/mfma_bank.hip -o tools/build/mfma_bank
//
// Why: gemm_tn_frag_kernel got 11 us SLOWER (32 -> 43 us) when the v_mul that copied every A operand was removed, and both it and
// mlp_tile_kernel feed element i of one float4 (A) together with element i of another float4 (B) to the same MFMA -- float4 register
// quads are 4-aligned, so A and B always share a bank.  Explicit registers, two accumulators alternating, all 256 CUs busy:
//   same    A = v[4+i], B = v[8+i]      (same bank)
//   shift1  A = v[4+i], B = v[9+i]      (banks differ by 1)
//   shift2  A = v[4+i], B = v[10+i]
//   rot     A = v[4+i], B = v[8+(i+1)%4] (the data-layout fix: B stored rotated by one element)
// 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define MF(a, b, c) "v_mfma_f32_32x32x2_f32 a[" c "], v" a ", v" b ", a[" c "]\n"
#define BODY_SAME MF("4", "8", "0:15") MF("4", "8", "16:31") MF("5", "9", "0:15") MF("5", "9", "16:31") MF("6", "10", "0:15") MF("6", "10", "16:31") MF("7", "11", "0:15") MF("7", "11", "16:31")
#define BODY_SHIFT1 MF("4", "9", "0:15") MF("4", "9", "16:31") MF("5", "10", "0:15") MF("5", "10", "16:31") MF("6", "11", "0:15") MF("6", "11", "16:31") MF("7", "12", "0:15") MF("7", "12", "16:31")
#define BODY_SHIFT2 MF("4", "10", "0:15") MF("4", "10", "16:31") MF("5", "11", "0:15") MF("5", "11", "16:31") MF("6", "12", "0:15") MF("6", "12", "16:31") MF("7", "13", "0:15") MF("7", "13", "16:31")
#define BODY_ROT MF("4", "9", "0:15") MF("4", "9", "16:31") MF("5", "10", "0:15") MF("5", "10", "16:31") MF("6", "11", "0:15") MF("6", "11", "16:31") MF("7", "8", "0:15") MF("7", "8", "16:31")

template <int KIND>
__global__ void __launch_bounds__(512, 1) bank_kernel(float* out, unsigned long long* clk, int iters) {
  // operands: small finite values in v4..v13
  asm volatile(
      "v_mov_b32 v4, 0x3a83126f\n v_mov_b32 v5, 0x3a83126f\n v_mov_b32 v6, 0x3a83126f\n v_mov_b32 v7, 0x3a83126f\n"
      "v_mov_b32 v8, 0x3a83126f\n v_mov_b32 v9, 0x3a83126f\n v_mov_b32 v10, 0x3a83126f\n v_mov_b32 v11, 0x3a83126f\n"
      "v_mov_b32 v12, 0x3a83126f\n v_mov_b32 v13, 0x3a83126f\n" ::
          : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
  asm volatile(
      "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n"
      "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
      "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n"
      "v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n"
      "v_accvgpr_write_b32 a16, 0\n v_accvgpr_write_b32 a17, 0\n v_accvgpr_write_b32 a18, 0\n v_accvgpr_write_b32 a19, 0\n"
      "v_accvgpr_write_b32 a20, 0\n v_accvgpr_write_b32 a21, 0\n v_accvgpr_write_b32 a22, 0\n v_accvgpr_write_b32 a23, 0\n"
      "v_accvgpr_write_b32 a24, 0\n v_accvgpr_write_b32 a25, 0\n v_accvgpr_write_b32 a26, 0\n v_accvgpr_write_b32 a27, 0\n"
      "v_accvgpr_write_b32 a28, 0\n v_accvgpr_write_b32 a29, 0\n v_accvgpr_write_b32 a30, 0\n v_accvgpr_write_b32 a31, 0\n" ::
          : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20",
            "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) asm volatile(BODY_SAME ::: "a0", "a16", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
    if (KIND == 1) asm volatile(BODY_SHIFT1 ::: "a0", "a16", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
    if (KIND == 2) asm volatile(BODY_SHIFT2 ::: "a0", "a16", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
    if (KIND == 3) asm volatile(BODY_ROT ::: "a0", "a16", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13");
  }
  float r;
  asm volatile("s_nop 15\n s_nop 15\n v_accvgpr_read_b32 %0, a0" : "=v"(r));
  const unsigned long long w1 = wall_clock64();
  if (threadIdx.x % 64 == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = w1 - w0;
  if (r == 12345.f) out[0] = r;
}

template <int KIND>
void run(const char* name, int threads, float* out, unsigned long long* clk, int n_cus) {
  const int iters = 64;   // 512 MFMAs per wave
  std::vector<unsigned long long> h(n_cus * threads / 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(bank_kernel<KIND>, dim3(n_cus), dim3(threads), 0, 0, out, clk, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), clk, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double us = sum / h.size() / 100.0;   // 100 MHz wall clock
  const int waves_per_simd = threads / 256;
  printf("%-7s waves/SIMD %d : %8.2f us per wave for %d MFMAs -> %.1f ns per MFMA per SIMD\n", name, waves_per_simd, us, iters * 8, us * 1e3 / (iters * 8 * waves_per_simd));
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int n_cus = p.multiProcessorCount;
  float* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&clk, sizeof(unsigned long long) * n_cus * 8);
  for (int threads : {256, 512}) {
    run<0>("same", threads, out, clk, n_cus);
    run<1>("shift1", threads, out, clk, n_cus);
    run<2>("shift2", threads, out, clk, n_cus);
    run<3>("rot", threads, out, clk, n_cus);
  }
  return 0;
}
