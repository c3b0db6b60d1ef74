// kp1_mlp.hip -- actor-critic MLP forward / loss / backward / Adam on gfx950 matrix cores.
//
// Arithmetic: fp32 in, fp32 accumulate (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
// The reference trains in fp32 (SB3 on torch defaults), so no reduced-precision operands are used.
//
// GEMM shapes per optimiser step (B = minibatch rows, H = hidden, both nets batched in grid.z):
//   fwd   Z1 = X[B,64] W1p^T      Z2 = H1[B,H] W2^T                         -> gemm_nt  (A row-major, W row-major [N][K])
//   bwd   dZ1 = (dZ2[B,H] W2) * (1 - H1^2)   (uses the packed transpose W2T) -> gemm_nt
//   wgrad dW2 = dZ2^T H1 , dW1 = dZ1^T X      (reduction over B, split-K)     -> gemm_tn
// The 7+1 wide heads, the PPO loss and its gradient run on the vector ALUs in head kernels.
//
// MFMA operand scheme (32x32x2): lane l feeds A[i = l&31][kslot = l>>5] and B[kslot][j = l&31]; C/D element
// (row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31).  K order inside a tile is free, so each lane reads 4 (NT) or
// 1 (TN) consecutive k per LDS read and both operands use the same k for the same kslot.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/kp1_ppo.h"
#include "kp1_device.hpp"
#include "kp1_host.hpp"

using kp1::fail;
using namespace kp1;   // kp1_device.hpp: the env arithmetic the rollout kernel steps its tile's envs with

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: arrays of it are reliably scalar-replaced (HIP's float4 struct was not)

namespace {

constexpr int ACT = KP1_MLP_ACT, HEADS = 8;  // observation width: ParamLayout::IN (56, or 80 with the route keys), padded to INP (64 / 128)
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;

enum { EPI_BIAS_TANH = 0, EPI_DTANH = 1 };

// Developer timeline (tools/nt_timeline.py builds a private copy of the library with -DKP1_NT_TRACE): thread 0 of every
// workgroup stamps the 100 MHz wall clock at the phase boundaries of gemm_nt_kernel.  Compiled out of the product.
#ifdef KP1_NT_TRACE
constexpr int KP1_TRACE_SLOTS = 16, KP1_TRACE_WGS = 1024;
__device__ unsigned long long kp1_nt_trace_buf[KP1_TRACE_SLOTS * KP1_TRACE_WGS];
#define KP1_TR(slot)                                                                                          \
  if (threadIdx.x == 0) {                                                                                     \
    const int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                           \
    if (wg_ < KP1_TRACE_WGS) kp1_nt_trace_buf[wg_ * KP1_TRACE_SLOTS + (slot)] = wall_clock64();               \
  }
// where the workgroup runs: HW_ID (cu_id [11:8], sh_id [12], se_id [15:13]) in the low word, XCC_ID in the high word
#define KP1_TR_HW(slot)                                                                                       \
  if (threadIdx.x == 0) {                                                                                     \
    const int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                           \
    if (wg_ < KP1_TRACE_WGS)                                                                                  \
      kp1_nt_trace_buf[wg_ * KP1_TRACE_SLOTS + (slot)] =                                                      \
          (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
  }
#else
#define KP1_TR(slot)
#define KP1_TR_HW(slot)
#endif

// Developer check of the shader clock the matrix kernels actually run at (the guide's DVFS item 6: cycles of s_memtime per tick of the 100 MHz
// s_memrealtime, stamped once at the start and once at the end of a workgroup; tools/mfma_clock.py builds with -DKP1_CLK_TRACE).  The stamps go to
// a buffer nothing else reads.  Compiled out of the product.
#ifdef KP1_CLK_TRACE
__device__ unsigned long long kp1_clk_buf[2 * 1024 * 4];   // [kernel: 0 tile, 1 weight gradients][workgroup][cycles at start, ticks, cycles at end, ticks]
#define KP1_CLK(kern, at)                                                                                     \
  if (threadIdx.x == 0) {                                                                                     \
    const int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                           \
    if (wg_ < 1024) {                                                                                         \
      kp1_clk_buf[((kern) * 1024 + wg_) * 4 + 2 * (at)] = __builtin_readcyclecounter();                       \
      kp1_clk_buf[((kern) * 1024 + wg_) * 4 + 2 * (at) + 1] = wall_clock64();                                 \
    }                                                                                                         \
  }
#else
#define KP1_CLK(kern, at)
#endif

// Cache policy of the big once-written / once-read streams (build switches; the defaults are the measured optimum).
// aux bits of the gfx950 buffer instructions: 1 = sc0, 2 = nt, 16 = sc1.  sc1 stores are write-through: the line leaves the XCD's L2 when it
// is written instead of at the kernel's end-of-launch write-back, and the consumer (another launch, mostly on another XCD) reads it from
// HBM / the memory-side cache either way.
#ifndef KP1_FU_ST_AUX
#define KP1_FU_ST_AUX 0    // tile kernel: X / h1 / dZ2 / dZ1 activation stores
#endif
#ifndef KP1_TNF_ST_AUX
#define KP1_TNF_ST_AUX 16  // weight-gradient kernel: partial-slab stores (write-through: 30.4 -> 29.0 us in situ, profiles/r02_ab_cache_policy.log)
#endif
#ifndef KP1_TNF_LD_NT
#define KP1_TNF_LD_NT 0    // weight-gradient kernel: activation operand loads non-temporal
#endif
#ifndef KP1_TNS_LD_AUX
#define KP1_TNS_LD_AUX 16  // wave-split weight-gradient kernel: the activation-fragment loads bypass the CU's L1 (sc1; no reuse inside a CU): 25.4 -> 24.4 us in situ, nt 26.5 (profiles/r03_ab_tn_load_policy.log)
#endif
#ifndef KP1_FIN_LD_NT
#define KP1_FIN_LD_NT 0    // finalize kernel: partial-slab loads non-temporal
#endif
#ifndef KP1_FIN_SPLIT
#define KP1_FIN_SPLIT 1    // finalize kernel: adjacent lanes that share one float4 item and split its batch chunks (1, 2 or 4)
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte store of `v` at float index `idx` of the (wave-uniform) base pointer with cache policy AUX; AUX = 0 is a plain global store
template <int AUX>
__device__ __forceinline__ void store16(float* __restrict__ base, int64_t idx, const f32x4 v) {
  if constexpr (AUX == 0) {
    *reinterpret_cast<f32x4*>(base + idx) = v;
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(idx * 4), 0, AUX);
  }
}
// 16-byte load at base[off] (floats) with cache-policy bits (raw buffer load: 1 sc0, 2 nt, 16 sc1); AUX = 0: a plain global load
template <int AUX>
__device__ __forceinline__ f32x4 load16_aux(const float* __restrict__ base, int64_t off) {
  if constexpr (AUX == 0) {
    return *reinterpret_cast<const f32x4*>(base + off);
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * 4), 0, AUX));
  }
}
template <int NT>
__device__ __forceinline__ f32x4 load16(const float* __restrict__ p) {
  if constexpr (NT == 0) return *reinterpret_cast<const f32x4*>(p);
  else return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}

// tanh(x) = 2 / (1 + 2^(-2 log2(e) x)) - 1 in FIVE vector instructions (v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma = 28 issue cycles).
// Instruction count matters more here than anywhere else in the kernel: v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 FMA lanes, so
// vector-ALU work never overlaps with fp32 MFMAs -- not inside a wave (one v_fma between two MFMAs costs its full 4 cycles on top of the
// MFMA's 64) and not across co-resident waves (a VALU-only wave next to an MFMA-only wave finishes in the SUM of both times);
// tools/mfma_valu_mix.hip, profiles/r02_mfma_valu_mix.log.  Every vector instruction of an epilogue is therefore paid in MFMA time, and
// the 64 tanh evaluations per lane were 38 % of the tile kernel's vector instructions in the round-1 form (14 instructions each: odd
// polynomial below |x| = 0.25, exp form above, select, copysign).  Saturates correctly (2^(+big) = inf -> -1, 2^(-big) = 0 -> +1), NaN
// propagates.  Absolute error <= ~1.5e-7 everywhere (the rounding of 1 + t and of the final 2r - 1), i.e. fp32 epsilon of the result's
// scale; the RELATIVE error grows as |x| -> 0 (result ~ x +- 1e-7), which a tanh hidden unit's consumers -- dot products of 256 such
// values, and 1 - h^2 in the backward pass -- do not resolve.  Checked against torch.tanh in tests/test_ppo_kernels_gpu.py.
__device__ __forceinline__ float kp_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * -2.8853900817779268f);
  return fmaf(2.0f, __builtin_amdgcn_rcpf(t + 1.0f), -1.0f);
}

struct GemmNT {
  const float* A; int64_t lda; int64_t strideA;       // [M][lda], per-net stride
  const int64_t* gather;                               // optional row indices into A (shared by both nets)
  const float* W; int64_t strideW;                     // k-slab major [K/32][N][32] (struct Packed)
  const float* bias; int64_t strideBias;               // [N]
  float* C; int64_t ldc; int64_t strideC;              // [M][ldc]
  const float* aux; int64_t strideAux;                 // EPI_DTANH: activation H at C's coordinates (ld = ldc)
  float* colsum; int64_t strideColsum;                 // EPI_DTANH: per-row-block column sums of C (bias-gradient partials)
                                                       //   written to colsum[blockIdx.x * 2*strideColsum + z*strideColsum + n]
  int M, N, K, Kreal;                                  // K multiple of 32; A columns >= Kreal read as 0
};

// C = epi(A W^T).  One workgroup owns 64 rows x BN columns (BN = 256: the whole hidden width, or 128 for small
// batches) for one net.  The 64 x K block of A is loaded ONCE and stays in LDS; W streams through a double-buffered
// [BN][32] stage, so every stage carries 4096 MFMA cycles per wave (BN = 256) against one L2 round trip, and A is
// never re-read.  4 waves, each 64 columns wide (2 MFMA column blocks) and RB row blocks tall.
// amdgpu_waves_per_eu(1, 2): the 140 KB LDS footprint allows one workgroup (one wave per SIMD) per CU anyway; without
// the hint hipcc spills the prefetch registers to scratch to stay under the 256-register budget of two waves per SIMD.
template <int BN, int EPI, int K, int NTH, int BM, int BKS>
__global__ void __launch_bounds__(NTH) gemm_nt_kernel(const GemmNT g) {
  constexpr int NW = NTH / 64, LDS_T = BKS + 4, KC = BKS / 4;  // BKS = k depth of one W stage, LDS_T its LDS pitch  // NTH = 512: two waves per SIMD share the MFMA pipe and split the epilogue VALU work
  constexpr int KQ = K / 4, A_LOADS = BM * KQ / NTH;  // float4 per A row / per thread: all issued before any is used
  constexpr int WN = BN / 64, WM = NW / WN;
  constexpr int RB = BM / WM / 32, CB = 2;
  constexpr int W_LOADS = BN * KC / NTH;
  extern __shared__ float lds[];
  constexpr int lda_s = K + 4;               // LDS pitch of the resident A block (bank-conflict-free b128 reads)
  float* As = lds;
  float* Ws0 = lds + BM * lda_s;
  float* Ws1 = Ws0 + BN * LDS_T;

  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* __restrict__ A = g.A + z * g.strideA;
  const float* __restrict__ W = g.W + z * g.strideW + (int64_t)n0 * 32;  // k-slab major: [K/32][N][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;

  // ---- prologue: W stages 0 and 1 to registers, whole A block to LDS.  W is prefetched TWO stages ahead (register
  // sets rwa / rwb alternate): with one wave per SIMD nothing else hides an L2 round trip, and one 4096-cycle stage
  // of cover was not enough under 256 workgroups streaming the same W.
  constexpr int KT = K / BKS;
  f32x4 rwa[W_LOADS], rwb[W_LOADS];
#define KP1_NT_WLOAD(dst, stage)                                                                        \
  {                                                                                                     \
    const int kk = ((stage) < KT ? (stage) : KT - 1) * BKS;                                              \
    _Pragma("unroll") for (int j = 0; j < W_LOADS; ++j) {                                               \
      const int f = tid + NTH * j;                                                                      \
      dst[j] = *reinterpret_cast<const f32x4*>(W + (int64_t)(kk >> 5) * g.N * 32 + (f / KC) * 32 + (kk & 31) + 4 * (f % KC)); \
    }                                                                                                   \
  }
#define KP1_NT_WSTORE(src, ws)                                                                          \
  _Pragma("unroll") for (int j = 0; j < W_LOADS; ++j) {                                                 \
    const int f = tid + NTH * j;                                                                        \
    *reinterpret_cast<f32x4*>((ws) + (f / KC) * LDS_T + 4 * (f % KC)) = src[j];                            \
  }
  KP1_TR(0)
  KP1_NT_WLOAD(rwa, 0)
  KP1_NT_WLOAD(rwb, 1)
  {
    // branch-free and fully unrolled: (gather indices) -> all A_LOADS row loads in flight -> mask -> LDS
    int64_t arow[A_LOADS];
    f32x4 av[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      const int m = min(m0 + (tid + NTH * j) / KQ, g.M - 1);
      arow[j] = g.gather ? g.gather[m] : (int64_t)m;
    }
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      const int k = 4 * ((tid + NTH * j) % KQ);
      av[j] = *reinterpret_cast<const f32x4*>(A + arow[j] * g.lda + (k < g.Kreal ? k : 0));
    }
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      const int f = tid + NTH * j, row = f / KQ, k = 4 * (f % KQ);
      const float keep = (m0 + row < g.M && k < g.Kreal) ? 1.f : 0.f;
      *reinterpret_cast<f32x4*>(As + row * lda_s + k) = av[j] * keep;
    }
  }
  KP1_NT_WSTORE(rwa, Ws0)
  __syncthreads();
  KP1_TR(1)

  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

  const float* as_base = As + (wr * (BM / WM) + (lane & 31)) * lda_s + 4 * (lane >> 5);
#define KP1_NT_COMPUTE(kt_, ws_cur_)                                                                    \
  {                                                                                                     \
    const float* as = as_base + (kt_) * BKS;                                                             \
    const float* ws = (ws_cur_) + (wc * 64 + (lane & 31)) * LDS_T + 4 * (lane >> 5);                      \
    _Pragma("unroll") for (int kg = 0; kg < BKS / 8; ++kg) {                                             \
      float4 a[RB], b[CB];                                                                              \
      _Pragma("unroll") for (int r = 0; r < RB; ++r) a[r] = *reinterpret_cast<const float4*>(as + r * 32 * lda_s + kg * 8); \
      _Pragma("unroll") for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const float4*>(ws + c * 32 * LDS_T + kg * 8);   \
      _Pragma("unroll") for (int r = 0; r < RB; ++r)                                                    \
        _Pragma("unroll") for (int c = 0; c < CB; ++c) {                                                \
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b[c].x, acc[r][c], 0, 0, 0);         \
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b[c].y, acc[r][c], 0, 0, 0);         \
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, b[c].z, acc[r][c], 0, 0, 0);         \
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, b[c].w, acc[r][c], 0, 0, 0);         \
        }                                                                                               \
    }                                                                                                   \
  }
  // stage kt computes from LDS buffer (kt & 1); registers hold stage kt+1 (written to the other buffer at the end of
  // the stage) while the loads of stage kt+2 are issued at its start.  Unrolled by two so the register sets are static.
  for (int kt = 0; kt < KT; kt += 2) {
    KP1_NT_WLOAD(rwa, kt + 2)          // rwa is free: stage kt already sits in Ws0
    KP1_NT_COMPUTE(kt, Ws0)
    KP1_NT_WSTORE(rwb, Ws1)            // stage kt+1
    __syncthreads();
    KP1_NT_WLOAD(rwb, kt + 3)          // KT is even (K % 64 == 0, checked on the host): no conditional, no spills
    KP1_NT_COMPUTE(kt + 1, Ws1)
    KP1_NT_WSTORE(rwa, Ws0)            // stage kt+2
    __syncthreads();
    KP1_TR(2 + kt / 2)
  }
#undef KP1_NT_WLOAD
#undef KP1_NT_WSTORE
#undef KP1_NT_COMPUTE

  // ---- epilogue: accumulators -> LDS (row-major [64][BN+4], reusing the A/W staging space) -> each thread owns
  // float4 column groups of 16 B, so bias/aux loads and the C stores are fully coalesced dwordx4 (the MFMA register
  // layout would give 64 scalar stores per lane, which are issue-bound with one wave per SIMD).
  constexpr int LDC = BN + 4, CQ = BN / 4, C_ITERS = BM * CQ / NTH;
  float* Cs = lds;
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wr * (BM / WM) + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        Cs[row * LDC + wc * 64 + c * 32 + (lane & 31)] = acc[r][c][e];
      }
  __syncthreads();
  KP1_TR(10)
  float* __restrict__ C = g.C + z * g.strideC;
  const int c4 = tid % CQ;                      // fixed column group per thread
  const int ncol = n0 + 4 * c4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_BIAS_TANH) bias4 = *reinterpret_cast<const f32x4*>(g.bias + z * g.strideBias + ncol);
  f32x4 hv[EPI == EPI_DTANH ? C_ITERS : 1];
  if constexpr (EPI == EPI_DTANH) {
#pragma unroll
    for (int j = 0; j < C_ITERS; ++j) {
      const int m = min(m0 + (tid + NTH * j) / CQ, g.M - 1);
      hv[j] = *reinterpret_cast<const f32x4*>(g.aux + z * g.strideAux + (int64_t)m * g.ldc + ncol);
    }
  }
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < C_ITERS; ++j) {
    const int row = (tid + NTH * j) / CQ;
    const int m = m0 + row;
    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LDC + 4 * c4);
    if constexpr (EPI == EPI_BIAS_TANH) {
      v += bias4;
      v.x = kp_tanh(v.x); v.y = kp_tanh(v.y); v.z = kp_tanh(v.z); v.w = kp_tanh(v.w);
    } else {
      v = v * (1.f - hv[j] * hv[j]);
      if (m < g.M) csum += v;
    }
    if (m < g.M) *reinterpret_cast<f32x4*>(C + (int64_t)m * g.ldc + ncol) = v;
  }
  KP1_TR(11)
  if constexpr (EPI == EPI_DTANH) {
    if (g.colsum) {
      // bias-gradient partial of this row block: threads tid, tid + CQ, ... own the same column group; combine them in
      // LDS (space behind the C tile) and write one row of partials with plain stores (no contended atomics)
      float* red = lds + BM * LDC;
      *reinterpret_cast<f32x4*>(red + (tid / CQ) * BN + 4 * c4) = csum;
      __syncthreads();
      if (tid < BN) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NTH / CQ; ++k) t += red[k * BN + tid];
        g.colsum[(int64_t)blockIdx.x * 2 * g.strideColsum + z * g.strideColsum + n0 + tid] = t;
      }
    }
  }
}

struct GemmTN {
  const float* D; int64_t ldd; int64_t strideD;     // dZ [B][ldd] : output-feature columns ("M" of the product)
  const float* X; int64_t ldx; int64_t strideX;     // previous activations [B][ldx] : input-feature columns ("N")
  const int64_t* gatherX;                            // optional row gather for X (layer 1 reads the obs buffer)
  float* slab;                                       // partial products [chunk][net][Mpad][slab_ld], plain stores
  int64_t slab_ld, slab_net_stride, slab_chunk_stride;
  int B, Nload;                                      // X columns < Nload are readable (the rest of the tile is zero)
  int chunk;                                         // rows of B reduced per workgroup (multiple of 64)
  int n_i_tiles;
};

// partial[chunk][net][o][i] = sum_{b in chunk} D[b][o] X[b][i].   Block tile 128(o) x 128(i), 4 waves 2x2, wave tile
// 64x64 with interleaved row/col blocks (o = base + 2*(l&31) + rb): one ds_read_b64 per operand and step feeds two MFMA
// row (column) blocks.  Both operands sit in LDS exactly as in memory (batch-row major), no transposes.  128x128 tiles
// need 32 FLOP per byte streamed from L2 (64x64 tiles: 16 FLOP/B, which demanded ~10 TB/s and stalled every stage).
// Stages are 64 batch rows (8192 MFMA cycles per wave), double buffered.  The batch axis is split over workgroups; each
// writes its partial tile with plain coalesced stores and tn_reduce_kernel sums the partials in fixed order, so weight
// gradients are bitwise reproducible (float atomics are not).
// grid: x = B chunk, y = o_tile * n_i_tiles + i_tile, z = net.
template <bool GATHER>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) gemm_tn_kernel(const GemmTN g) {
  constexpr int TB = 64, TT = 128, LDW = TT + 4, LOADS = TB * TT / 4 / 256;  // 8 float4 per thread per operand
  extern __shared__ float lds[];
  const int z = blockIdx.z;
  const int o0 = (blockIdx.y / g.n_i_tiles) * TT, i0 = (blockIdx.y % g.n_i_tiles) * TT;
  const int b_begin = blockIdx.x * g.chunk;
  const int b_end = min(b_begin + g.chunk, g.B);
  const float* __restrict__ D = g.D + z * g.strideD + o0;
  const float* __restrict__ X = g.X + z * g.strideX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int srow = tid >> 5, sc4 = tid & 31;  // staging: f = tid + 256*j -> row = srow + 8*j, c4 = sc4
  const bool x_col_ok = i0 + 4 * sc4 < g.Nload;
  const int x_col = x_col_ok ? i0 + 4 * sc4 : 0;
  const int b_last = b_end - 1;

  // Branch-free staging: out-of-range rows are clamped to the last valid row and zeroed at LDS-store time, so all 16
  // global loads of a stage are issued back to back and nothing waits on them until the stage's MFMAs are issued.
  f32x4 rd[LOADS], rx[LOADS];
#define KP1_TN_LOAD(b0)                                                                                   \
  {                                                                                                       \
    int64_t xrow[LOADS];                                                                                  \
    _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                   \
      const int b = min((b0) + srow + 8 * j, b_last);                                                     \
      xrow[j] = GATHER ? g.gatherX[b] : (int64_t)b;                                                       \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                   \
      const int b = min((b0) + srow + 8 * j, b_last);                                                     \
      rd[j] = *reinterpret_cast<const f32x4*>(D + (int64_t)b * g.ldd + 4 * sc4);                          \
      rx[j] = *reinterpret_cast<const f32x4*>(X + xrow[j] * g.ldx + x_col);                               \
    }                                                                                                     \
  }
#define KP1_TN_STORE(buf, b0)                                                                             \
  _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                     \
    const float keep = ((b0) + srow + 8 * j <= b_last) ? 1.f : 0.f;                                       \
    const float keepx = x_col_ok ? keep : 0.f;                                                            \
    float* base = lds + (buf) * 2 * TB * LDW + (srow + 8 * j) * LDW + 4 * sc4;                            \
    *reinterpret_cast<f32x4*>(base) = rd[j] * keep;                                                       \
    *reinterpret_cast<f32x4*>(base + TB * LDW) = rx[j] * keepx;                                           \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

  const int KT = (b_end - b_begin + TB - 1) / TB;
  KP1_TR(0)
  if (KT > 0) {  // uniform per workgroup
    KP1_TN_LOAD(b_begin)
    KP1_TN_STORE(0, b_begin)
    __syncthreads();
    KP1_TR(1)
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      KP1_TN_LOAD(b_begin + (kt + 1) * TB)  // past the chunk end: clamped rows, zeroed at store time
      const float* ds = lds + buf * 2 * TB * LDW + (lane >> 5) * LDW + wr * 64 + 2 * (lane & 31);
      const float* xs = lds + buf * 2 * TB * LDW + TB * LDW + (lane >> 5) * LDW + wc * 64 + 2 * (lane & 31);
#pragma unroll
      for (int s0 = 0; s0 < TB / 2; s0 += 4) {
        float2 av[4], xv[4];  // LDS reads of 4 steps (16 MFMAs) in flight ahead of their use
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = *reinterpret_cast<const float2*>(ds + 2 * (s0 + u) * LDW);
          xv[u] = *reinterpret_cast<const float2*>(xs + 2 * (s0 + u) * LDW);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, xv[u].x, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, xv[u].y, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, xv[u].x, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, xv[u].y, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // 8 DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // then their 16 MFMAs
      }
      KP1_TN_STORE(buf ^ 1, b_begin + (kt + 1) * TB)
      __syncthreads();
      KP1_TR(2 + (kt < 8 ? kt : 8))
    }
  }
#undef KP1_TN_LOAD
#undef KP1_TN_STORE
  // partial tile -> slab (zeros when this chunk had no rows): transposed through LDS so rows are written as float4
  constexpr int LDP = TT + 4;
  float* Ps = lds;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = wr * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) + r;
        Ps[o * LDP + wc * 64 + 2 * (lane & 31) + c] = acc[r][c][e];
      }
  __syncthreads();
  KP1_TR(11)
  float* __restrict__ P = g.slab + blockIdx.x * g.slab_chunk_stride + z * g.slab_net_stride;
#pragma unroll
  for (int j = 0; j < TT * TT / 4 / 256; ++j) {
    const int f = tid + 256 * j, o = f >> 5, q = f & 31;
    const int col = i0 + 4 * q;
    if (col < g.slab_ld) *reinterpret_cast<f32x4*>(P + (int64_t)(o0 + o) * g.slab_ld + col) = *reinterpret_cast<const f32x4*>(Ps + o * LDP + 4 * q);
  }
  KP1_TR(12)
}

// ---------------------------------------------------------------------------------------------- heads
struct HeadArgs {
  const float* h2;        // [2][n][Hp]  (net 0 = policy, 1 = value)
  int64_t strideH;        // n_alloc * Hp
  int Hp, n;
  const float* w3;        // [8][Hp] rows 0..6 action_net, row 7 value_net
  const float* b3;        // [8]
  const float* log_std;   // [7]
  // inference outputs
  const float* noise; float* mean; float* value; float* action; float* clipped; float* log_prob;
  // training inputs
  const int64_t* idx; const float* actions; const float* old_logp; const float* adv; const float* ret;
  const double* adv_partials; int n_adv_partials; const float* adv_stats; float adv_mean, adv_inv_std; int adv_mode;
  float clip_range, ent_coef, vf_coef, inv_count;
  // training outputs
  float* dz2;             // [2][n][Hp]
  float* hpart;           // per-block partials [n_blocks][hpart_stride]: dW action [7][Hp], dW value [Hp], db2 pi [Hp], db2 vf [Hp],
  int hpart_stride;       //   then 8 head-bias grads, 7 log_std grads, 3 loss sums (policy, value, kl)
  int H;                  // real hidden (<= Hp)
};

constexpr int HEAD_ROWS = 32;  // rows per 256-thread block: thread = (row = t/8, out = t%8)

__device__ __forceinline__ float head_dot(const float* __restrict__ h, const float* __restrict__ w, int Hp) {
  float s = 0.f;
  for (int k = 0; k < Hp; k += 4) {
    const float4 a = *reinterpret_cast<const float4*>(h + k);
    const float4 b = *reinterpret_cast<const float4*>(w + k);
    s = fmaf(a.x, b.x, s);
    s = fmaf(a.y, b.y, s);
    s = fmaf(a.z, b.z, s);
    s = fmaf(a.w, b.w, s);
  }
  return s;
}

// policy.forward tail: heads + Gaussian sampling (SB3 DiagGaussianDistribution)
__global__ void __launch_bounds__(256) head_infer_kernel(const HeadArgs a) {
  extern __shared__ float w3s[];  // [8][Hp]
  for (int k = threadIdx.x; k < HEADS * a.Hp; k += 256) w3s[k] = a.w3[k];
  __syncthreads();
  const int row = blockIdx.x * HEAD_ROWS + (threadIdx.x >> 3), out = threadIdx.x & 7;
  const bool ok = row < a.n;
  float v = 0.f;
  if (ok) {
    const float* h = a.h2 + (out == 7 ? a.strideH : 0) + (int64_t)row * a.Hp;
    v = head_dot(h, w3s + out * a.Hp, a.Hp) + a.b3[out];
  }
  float lp = 0.f;
  if (ok && out < ACT) {
    if (a.mean) a.mean[(int64_t)row * ACT + out] = v;
    float act = v;
    if (a.noise) {
      const float ls = a.log_std[out];
      const float nz = a.noise[(int64_t)row * ACT + out];
      act = fmaf(expf(ls), nz, v);
      lp = -0.5f * nz * nz - ls - LOG_SQRT_2PI;
    }
    if (a.action) a.action[(int64_t)row * ACT + out] = act;
    if (a.clipped) a.clipped[(int64_t)row * ACT + out] = fminf(fmaxf(act, -1.f), 1.f);
  }
  if (ok && out == 7 && a.value) a.value[row] = v;
  // log_prob = sum over the 7 action lanes of the 8-lane group
  lp += __shfl_xor(lp, 1);
  lp += __shfl_xor(lp, 2);
  lp += __shfl_xor(lp, 4);
  if (ok && out == 0 && a.log_prob) a.log_prob[row] = lp;
}

// per-block partial (sum, sum of squares) of the gathered advantages, fp64, fixed order => deterministic
__global__ void __launch_bounds__(256) adv_partials_kernel(const float* __restrict__ adv, const int64_t* __restrict__ idx, int n, double* __restrict__ partials) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double v = (double)adv[idx ? idx[i] : (int64_t)i];
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      s1[threadIdx.x] += s1[threadIdx.x + s];
      s2[threadIdx.x] += s2[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s1[0];
    partials[2 * blockIdx.x + 1] = s2[0];
  }
}

// heads forward + PPO loss gradient + dZ2 for both nets + head weight/bias gradients
template <int HP>
__global__ void __launch_bounds__(256) head_train_kernel(const HeadArgs a) {
  // all LDS is one dynamic array (16-B aligned base: head_dot reads float4; guide G17)
  extern __shared__ float smem[];
  const int hpitch = a.Hp + 4;             // row pitch of the staged activations: conflict-free float4 row reads
  float* w3s = smem;                       // [8][Hp]
  float* dout = smem + HEADS * a.Hp;       // [HEAD_ROWS][8] : d loss / d (mean_0..6, value)
  float* red = dout + HEAD_ROWS * 8;       // [4 waves][20] per-wave partial sums
  float* adv_ms = red + 80;                // [2] (+2 pad)
  float* h2s = adv_ms + 4;                 // [2][HEAD_ROWS][hpitch]: this block's rows of both nets, read from HBM once
  for (int k = threadIdx.x; k < HEADS * a.Hp; k += 256) w3s[k] = a.w3[k];
  {
    // all loads of this thread in flight before the first LDS store (a load->store loop serialises 16 round trips)
    constexpr int Q = HP / 4, LOADS = 2 * HEAD_ROWS * Q / 256;
    f32x4 v[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      const int f = threadIdx.x + 256 * j;
      const int net = f / (HEAD_ROWS * Q), rem = f % (HEAD_ROWS * Q), r = rem / Q, c4 = rem % Q;
      const int grow = min(blockIdx.x * HEAD_ROWS + r, a.n - 1);
      v[j] = *reinterpret_cast<const f32x4*>(a.h2 + net * a.strideH + (int64_t)grow * HP + 4 * c4);
    }
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      const int f = threadIdx.x + 256 * j;
      const int net = f / (HEAD_ROWS * Q), rem = f % (HEAD_ROWS * Q), r = rem / Q, c4 = rem % Q;
      *reinterpret_cast<f32x4*>(h2s + (net * HEAD_ROWS + r) * hpitch + 4 * c4) = v[j];
    }
  }
  // fixed-order parallel reduction of the per-block advantage partials (wave 0; a.n_adv_partials <= 128)
  double ps = 0.0, pss = 0.0;
  if (a.adv_mode == 1 && threadIdx.x < 64) {
    for (int k = threadIdx.x; k < a.n_adv_partials; k += 64) {
      ps += a.adv_partials[2 * k];
      pss += a.adv_partials[2 * k + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
      ps += __shfl_xor(ps, off);
      pss += __shfl_xor(pss, off);
    }
  }
  if (threadIdx.x == 0) {
    float mean = 0.f, inv_std = 1.f;
    if (a.adv_mode == 1) {  // statistics of this minibatch, torch .mean() / .std() (unbiased)
      const double s = ps, ss = pss;
      const double m = s / a.n;
      const double var = a.n > 1 ? fmax((ss - a.n * m * m) / (a.n - 1), 0.0) : 0.0;
      mean = (float)m;
      inv_std = (float)(1.0 / (sqrt(var) + 1e-8));
    } else if (a.adv_mode == 2) {
      mean = a.adv_stats[0];
      inv_std = a.adv_stats[1];
    } else if (a.adv_mode == 3) {
      mean = a.adv_mean;
      inv_std = a.adv_inv_std;
    }
    adv_ms[0] = mean;
    adv_ms[1] = inv_std;
  }
  __syncthreads();
  const int row_l = threadIdx.x >> 3, out = threadIdx.x & 7;
  const int row = blockIdx.x * HEAD_ROWS + row_l;
  const bool ok = row < a.n;
  const int64_t src = ok ? (a.idx ? a.idx[row] : (int64_t)row) : 0;
  float v = 0.f;
  if (ok) v = head_dot(h2s + ((out == 7 ? HEAD_ROWS : 0) + row_l) * hpitch, w3s + out * a.Hp, a.Hp) + a.b3[out];
  // Gaussian log-prob of the stored action under the current policy
  float z = 0.f, inv_sd = 1.f, lp = 0.f;
  if (ok && out < ACT) {
    const float ls = a.log_std[out];
    inv_sd = expf(-ls);
    z = (a.actions[src * ACT + out] - v) * inv_sd;
    lp = -0.5f * z * z - ls - LOG_SQRT_2PI;
  }
  lp += __shfl_xor(lp, 1);
  lp += __shfl_xor(lp, 2);
  lp += __shfl_xor(lp, 4);
  float g_logp = 0.f, d = 0.f, pl = 0.f, vl = 0.f, kl = 0.f;
  if (ok) {
    const float old = a.old_logp[src];
    const float A = (a.adv[src] - adv_ms[0]) * adv_ms[1];
    const float lr_ = lp - old;
    const float ratio = expf(lr_);
    const float rc = fminf(fmaxf(ratio, 1.f - a.clip_range), 1.f + a.clip_range);
    const float s1 = ratio * A, s2 = rc * A;
    // d/dlogp of -min(s1, s2): the unclipped branch carries the gradient unless the clipped one is strictly smaller
    const bool inside = ratio >= 1.f - a.clip_range && ratio <= 1.f + a.clip_range;
    g_logp = (inside || s1 < s2) ? -A * ratio * a.inv_count : 0.f;
    if (out < ACT) {
      d = g_logp * z * inv_sd;                           // d loss / d mean_out
    } else {
      const float R = a.ret[src];
      d = a.vf_coef * 2.f * (v - R) * a.inv_count;       // d loss / d value
      vl = (R - v) * (R - v);
      pl = -fminf(s1, s2);
      kl = (ratio - 1.f) - lr_;
    }
  }
  dout[row_l * 8 + out] = ok ? d : 0.f;
  // per-block sums over rows of: d loss/d log_std_out, d loss/d head bias_out, loss terms.  Lanes with equal (t & 7) hold
  // the same `out`: reduce inside the wave by shuffles, across the 4 waves through per-wave LDS slots summed in a fixed order.
  float gls = (ok && out < ACT) ? g_logp * (z * z - 1.f) : 0.f;
  gls += __shfl_xor(gls, 8);
  gls += __shfl_xor(gls, 16);
  gls += __shfl_xor(gls, 32);
  float dsum = ok ? d : 0.f;
  dsum += __shfl_xor(dsum, 8);
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);
  // per-wave partials in LDS, combined in a fixed order below (no float atomics: bitwise reproducible run to run).
  // red layout: [wave 0..3][20] = 3 loss sums, pad, 8 bias grads at +4, 7 log_std grads at +12
  float pl_w = out == 7 ? pl : 0.f, vl_w = out == 7 ? vl : 0.f, kl_w = out == 7 ? kl : 0.f;
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    pl_w += __shfl_xor(pl_w, off);
    vl_w += __shfl_xor(vl_w, off);
    kl_w += __shfl_xor(kl_w, off);
  }
  const int wave_ = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) {
    red[wave_ * 20 + 4 + out] = dsum;
    if (out < ACT) red[wave_ * 20 + 12 + out] = gls;
    if (out == 7) {
      red[wave_ * 20 + 0] = pl_w;
      red[wave_ * 20 + 1] = vl_w;
      red[wave_ * 20 + 2] = kl_w;
    }
  }
  __syncthreads();
  float* part = a.hpart + (int64_t)blockIdx.x * a.hpart_stride;
  if (threadIdx.x < 18) {
    const int k = threadIdx.x;  // 0..7 bias grads, 8..14 log_std grads, 15..17 loss sums
    const int slot = k < 8 ? 4 + k : (k < 15 ? 12 + (k - 8) : k - 15);
    part[10 * a.Hp + k] = ((red[slot] + red[20 + slot]) + red[40 + slot]) + red[60 + slot];
  }
  // phase 2: one thread per hidden column; dZ2 = (dOut W3) * (1 - h2^2), head weight grads, layer-2 bias grads
  const int rows_here = min(HEAD_ROWS, a.n - blockIdx.x * HEAD_ROWS);
  for (int h = threadIdx.x; h < a.Hp; h += 256) {
    float wcol[HEADS];
#pragma unroll
    for (int o = 0; o < HEADS; ++o) wcol[o] = w3s[o * a.Hp + h];
    float gw[HEADS];
#pragma unroll
    for (int o = 0; o < HEADS; ++o) gw[o] = 0.f;
    float gb2p = 0.f, gb2v = 0.f;
    for (int r = 0; r < rows_here; ++r) {
      const int64_t rr = (int64_t)(blockIdx.x * HEAD_ROWS + r) * a.Hp + h;
      const float hp = h2s[r * hpitch + h], hv = h2s[(HEAD_ROWS + r) * hpitch + h];
      float dp = 0.f;
#pragma unroll
      for (int o = 0; o < ACT; ++o) {
        const float dd = dout[r * 8 + o];
        dp = fmaf(dd, wcol[o], dp);
        gw[o] = fmaf(dd, hp, gw[o]);
      }
      const float dv = dout[r * 8 + 7];
      gw[7] = fmaf(dv, hv, gw[7]);
      const float dzp = dp * (1.f - hp * hp);
      const float dzv = dv * wcol[7] * (1.f - hv * hv);
      a.dz2[rr] = dzp;
      a.dz2[a.strideH + rr] = dzv;
      gb2p += dzp;
      gb2v += dzv;
    }
#pragma unroll
    for (int o = 0; o < HEADS; ++o) part[o * a.Hp + h] = gw[o];
    part[8 * a.Hp + h] = gb2p;
    part[9 * a.Hp + h] = gb2v;
  }
}

// ---------------------------------------------------------------------------------------------- optimiser
struct ParamLayout {  // offsets into the flat SB3-order vector
  int H, Hp, IN, INP;  // INP: IN rounded up to the k depth the GEMMs are instantiated for (64 or 128)
  int64_t log_std, p_w1, p_b1, p_w2, p_b2, v_w1, v_b1, v_w2, v_b2, a_w, a_b, c_w, c_b, total;
};

__host__ __device__ inline ParamLayout make_layout(int H, int IN = KP1_MLP_IN) {
  ParamLayout L;
  L.H = H;
  L.IN = IN;
  L.INP = IN <= 64 ? 64 : 128;
  L.Hp = (H + 127) / 128 * 128;
  int64_t o = 0;
  L.log_std = o; o += ACT;
  L.p_w1 = o; o += (int64_t)H * IN;
  L.p_b1 = o; o += H;
  L.p_w2 = o; o += (int64_t)H * H;
  L.p_b2 = o; o += H;
  L.v_w1 = o; o += (int64_t)H * IN;
  L.v_b1 = o; o += H;
  L.v_w2 = o; o += (int64_t)H * H;
  L.v_b2 = o; o += H;
  L.a_w = o; o += (int64_t)ACT * H;
  L.a_b = o; o += ACT;
  L.c_w = o; o += H;
  L.c_b = o; o += 1;
  L.total = o;
  return L;
}

// Kernel-format GEMM weights are "k-slab major": W[net][K/32][N][32], i.e. element (out row n, in column k) of a net sits
// at (k/32)*N*32 + n*32 + k%32.  One 32-deep W stage of a GEMM is then ONE contiguous N*128-byte run.  With the natural
// row-major [N][K] layout a stage is N separate 128-byte pieces at a 1 KB pitch, which (256-byte channel interleave) land
// on 4 of the 16 L2 channels of an XCD -- every CU streams the same stage at the same time, and the MFMA loops of all
// three GEMM kinds sat at ~50 % waiting on those four channels.
// The fused tile kernel reads a second copy, "fragment major": inside each 32-deep k stage the order is [MFMA column block
// n/32][k group (k%32)/8][lane = n%32 + 32*((k%8)/4)][k%4] -- exactly the float4 every lane feeds to four consecutive
// v_mfma_f32_32x32x2_f32 as B operand, so the weights go global -> VGPR in coalesced 1 KB wave loads, never through LDS.
struct Packed {
  float *w1p, *b1, *w2, *w2t, *b2, *w3, *b3, *log_std;  // [2][2][Hp][32], [2][Hp], [2][Hp/32][Hp][32] x2, [2][Hp], [8][Hp], [8], [8]
  float *w1f, *w2f, *w2tf;                               // fragment-major copies of w1p, w2, w2t
  int formats;                                           // which GEMM-weight copies pack_one writes: bit 0 k-slab major, bit 1 fragment major
};
constexpr int PACK_SLAB = 1, PACK_FRAG = 2;
__host__ __device__ inline int64_t slab_at(int64_t n, int64_t k, int64_t N) { return (k >> 5) * N * 32 + n * 32 + (k & 31); }
__host__ __device__ inline int64_t frag_at(int64_t n, int64_t k, int64_t N) {
  return (k >> 5) * N * 32 + (n >> 5) * 1024 + ((k & 31) >> 3) * 256 + ((n & 31) + 32 * ((k & 7) >> 2)) * 4 + (k & 3);
}

// flat SB3 vector element i -> its place(s) in the kernel-format weights (zero padding pre-set once at creation)
__device__ __forceinline__ void pack_one(int64_t i, float v, const ParamLayout& L, const Packed& k) {
  // 32-bit element offsets and unsigned divisions: a 64-bit signed division is a ~100-instruction software routine on this ISA and sat on the
  // per-thread critical path of adam_kernel (the vector has < 2^31 elements)
  const int H = L.H, Hp = L.Hp, IN = L.IN, INP = L.INP;
  const unsigned uH = (unsigned)H, uIN = (unsigned)IN;
  if (i < L.p_w1) {
    k.log_std[i] = v;
  } else if (i < L.p_b1) {
    const unsigned e = (unsigned)(i - L.p_w1), r = e / uIN, c = e - r * uIN;
    if (k.formats & PACK_SLAB) k.w1p[slab_at(r, c, Hp)] = v;
    if (k.formats & PACK_FRAG) k.w1f[frag_at(r, c, Hp)] = v;
  } else if (i < L.p_w2) {
    k.b1[i - L.p_b1] = v;
  } else if (i < L.p_b2) {
    const unsigned e = (unsigned)(i - L.p_w2), r = e / uH, c = e - r * uH;
    if (k.formats & PACK_SLAB) {
      k.w2[slab_at(r, c, Hp)] = v;
      k.w2t[slab_at(c, r, Hp)] = v;
    }
    if (k.formats & PACK_FRAG) {
      k.w2f[frag_at(r, c, Hp)] = v;
      k.w2tf[frag_at(c, r, Hp)] = v;
    }
  } else if (i < L.v_w1) {
    k.b2[i - L.p_b2] = v;
  } else if (i < L.v_b1) {
    const unsigned e = (unsigned)(i - L.v_w1), r = e / uIN, c = e - r * uIN;
    if (k.formats & PACK_SLAB) k.w1p[(int64_t)Hp * INP + slab_at(r, c, Hp)] = v;
    if (k.formats & PACK_FRAG) k.w1f[(int64_t)Hp * INP + frag_at(r, c, Hp)] = v;
  } else if (i < L.v_w2) {
    k.b1[Hp + i - L.v_b1] = v;
  } else if (i < L.v_b2) {
    const unsigned e = (unsigned)(i - L.v_w2), r = e / uH, c = e - r * uH;
    if (k.formats & PACK_SLAB) {
      k.w2[(int64_t)Hp * Hp + slab_at(r, c, Hp)] = v;
      k.w2t[(int64_t)Hp * Hp + slab_at(c, r, Hp)] = v;
    }
    if (k.formats & PACK_FRAG) {
      k.w2f[(int64_t)Hp * Hp + frag_at(r, c, Hp)] = v;
      k.w2tf[(int64_t)Hp * Hp + frag_at(c, r, Hp)] = v;
    }
  } else if (i < L.a_w) {
    k.b2[Hp + i - L.v_b2] = v;
  } else if (i < L.a_b) {
    const unsigned e = (unsigned)(i - L.a_w), r = e / uH;
    k.w3[(int64_t)r * Hp + (e - r * uH)] = v;
  } else if (i < L.c_w) {
    k.b3[i - L.a_b] = v;
  } else if (i < L.c_b) {
    k.w3[(int64_t)7 * Hp + (i - L.c_w)] = v;
  } else {
    k.b3[7] = v;
  }
}

__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ p, const ParamLayout L, const Packed k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < L.total) pack_one(i, p[i], L, k);
}

// Gradient finalisation: every producer kernel (TN GEMMs, backward-GEMM epilogue, head kernel) wrote per-workgroup
// partials with plain stores; this kernel sums them in a fixed order into the flat SB3-order gradient, adds the entropy
// term, accumulates the loss statistics and (fused) the per-block sum of squares for clip_grad_norm_.  No float atomics
// touch a gradient anywhere, so results are bitwise reproducible run to run.
struct FinalizeArgs {
  ParamLayout L;
  const float* slab2; int64_t s2_ld, s2_net, s2_chunk; int s2_n;   // dW2 partials [chunk][net][Hp][Hp]
  const float* slab1; int64_t s1_ld, s1_net, s1_chunk; int s1_n;   // dW1 partials [chunk][net][Hp][64]
  const float* bslab; int64_t b_net, b_tile; int b_n;              // db1 partials [row tile][net][Hp]
  const float* hpart; int64_t h_stride; int h_n;                   // head partials [block][10 Hp + 18]
  float ent_coef, inv_count;
  const float* log_std;
  float* grad; float* stats; double* sumsq;
  int* step_counter;  // Adam step count kept on the device (hipGraph replays cannot change a scalar kernel argument)
};

template <int UNROLL>
__device__ __forceinline__ float sum_strided(const float* __restrict__ p, int64_t stride, int n) {
  float s = 0.f;
  int c = 0;
  for (; c + UNROLL <= n; c += UNROLL) {
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[(int64_t)(c + u) * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u];
  }
  for (; c < n; ++c) s += p[(int64_t)c * stride];
  return s;
}

// where the partials of flat gradient element i live: n values at p[c * stride], plus a constant
struct PartialSrc {
  const float* p; int64_t stride; int n; float add;
  bool wide;  // few elements with one partial per 32-row tile (biases, heads, log_std): summed cooperatively
};

__device__ __forceinline__ PartialSrc finalize_source(const FinalizeArgs& a, int64_t i) {
  const ParamLayout& L = a.L;
  const int H = L.H, Hp = L.Hp, IN = L.IN;
  PartialSrc r;
  r.add = 0.f;
  r.wide = true;
  r.stride = a.h_stride;
  r.n = a.h_n;
  if (i < L.p_w1) {
    r.p = a.hpart + 10 * Hp + 8 + i;
    r.add = -a.ent_coef;  // d(-ent_coef * sum log_std)
  } else if (i < L.p_b1 || (i >= L.v_w1 && i < L.v_b1)) {
    const int net = i >= L.v_w1;
    const int64_t e = i - (net ? L.v_w1 : L.p_w1);
    r.p = a.slab1 + net * a.s1_net + (e / IN) * a.s1_ld + e % IN;
    r.stride = a.s1_chunk; r.n = a.s1_n; r.wide = false;
  } else if (i < L.p_w2 || (i >= L.v_b1 && i < L.v_w2)) {
    const int net = i >= L.v_b1;
    r.p = a.bslab + net * a.b_net + (i - (net ? L.v_b1 : L.p_b1));
    r.stride = a.b_tile; r.n = a.b_n;
  } else if (i < L.p_b2 || (i >= L.v_w2 && i < L.v_b2)) {
    const int net = i >= L.v_w2;
    const int64_t e = i - (net ? L.v_w2 : L.p_w2);
    r.p = a.slab2 + net * a.s2_net + (e / H) * a.s2_ld + e % H;
    r.stride = a.s2_chunk; r.n = a.s2_n; r.wide = false;
  } else if (i < L.v_w1) {
    r.p = a.hpart + 8 * Hp + (i - L.p_b2);
  } else if (i < L.a_w) {
    r.p = a.hpart + 9 * Hp + (i - L.v_b2);
  } else if (i < L.a_b) {
    const int64_t e = i - L.a_w;
    r.p = a.hpart + (e / H) * Hp + e % H;
  } else if (i < L.c_w) {
    r.p = a.hpart + 10 * Hp + (i - L.a_b);
  } else if (i < L.c_b) {
    r.p = a.hpart + 7 * Hp + (i - L.c_w);
  } else {
    r.p = a.hpart + 10 * Hp + 7;
  }
  return r;
}

// the "wide" elements in enumeration order: log_std, pi b1, pi b2, vf b1, vf b2, then everything from action_net on
__host__ __device__ inline int64_t finalize_wide_count(const ParamLayout& L) { return ACT + 4 * (int64_t)L.H + (L.total - L.a_w); }
__device__ __forceinline__ int64_t finalize_wide_index(const ParamLayout& L, int64_t e) {
  const int H = L.H;
  if (e < ACT) return L.log_std + e;
  e -= ACT;
  if (e < H) return L.p_b1 + e;
  e -= H;
  if (e < H) return L.p_b2 + e;
  e -= H;
  if (e < H) return L.v_b1 + e;
  e -= H;
  if (e < H) return L.v_b2 + e;
  e -= H;
  return L.a_w + e;
}

// Blocks [0, n_main): the four weight matrices (dW2, dW1 of both nets; few partials per element: one per batch chunk of the TN GEMMs).
// One thread owns FOUR consecutive columns of one weight row and sums its float4 partials over the chunks with all loads in flight: a
// quarter of the load instructions of a thread-per-element walk over the same 25 MB (the kernel is bound by load issue + latency, not
// bandwidth), same summation order per element (chunk 0, 1, 2, ...).  Blocks [n_main, ...): 32 "wide" elements each (biases, heads,
// log_std: one partial per 32-row tile), 8 threads per element walk the per-tile partials with a stride of 8 (32 loads in flight each) and
// LDS combines the 8 strands in a fixed order -- a single thread summing 256 tiles was the tail of this kernel (8 dependent load rounds).
// Every block leaves the sum of squares of what it wrote in sumsq[blockIdx.x] for clip_grad_norm_.
__host__ __device__ inline int64_t finalize_vec_items(const ParamLayout& L) { return 2 * ((int64_t)L.H * L.H / 4 + (int64_t)L.H * L.IN / 4); }
// KP1_FIN_SPLIT adjacent lanes share one item: lane part p sums the chunks [p * ceil(n / SPLIT), ...) in order and the parts are added in
// part order by shuffles -- a fixed order, so still bitwise reproducible; more workgroups pull on the 25 MB of slabs at once
constexpr int FIN_SPLIT = KP1_FIN_SPLIT;
static_assert(FIN_SPLIT == 1 || FIN_SPLIT == 2 || FIN_SPLIT == 4, "KP1_FIN_SPLIT must be 1, 2 or 4");

__global__ void __launch_bounds__(256) grad_finalize_kernel(const FinalizeArgs a, int n_main) {
  const ParamLayout& L = a.L;
  __shared__ double sq[4];
  __shared__ float red[8][32];
  float gval = 0.f;
  double gsq = 0.0;
  if ((int)blockIdx.x < n_main) {
    const int64_t j = (int64_t)((blockIdx.x * 256u + threadIdx.x) / (unsigned)FIN_SPLIT);
    const int part = threadIdx.x % FIN_SPLIT;
    const int64_t per2 = (int64_t)L.H * L.H / 4, per1 = (int64_t)L.H * L.IN / 4;
    if (j < 2 * (per2 + per1)) {
      const float* src;
      int64_t stride, dst;
      int n;
      if (j < 2 * per2) {
        const int net = j >= per2;
        const unsigned rem = (unsigned)(j - net * per2), q4 = (unsigned)(L.H / 4), row = rem / q4, c4 = rem - row * q4;   // 32-bit: see pack_one
        src = a.slab2 + net * a.s2_net + (int64_t)row * a.s2_ld + 4 * c4;
        stride = a.s2_chunk; n = a.s2_n;
        dst = (net ? L.v_w2 : L.p_w2) + (int64_t)row * L.H + 4 * c4;
      } else {
        const int64_t k = j - 2 * per2;
        const int net = k >= per1;
        const unsigned rem = (unsigned)(k - net * per1), q4 = (unsigned)(L.IN / 4), row = rem / q4, c4 = rem - row * q4;
        src = a.slab1 + net * a.s1_net + (int64_t)row * a.s1_ld + 4 * c4;
        stride = a.s1_chunk; n = a.s1_n;
        dst = (net ? L.v_w1 : L.p_w1) + (int64_t)row * L.IN + 4 * c4;
      }
      if (FIN_SPLIT > 1) {   // this lane's share of the chunks
        const int per = (n + FIN_SPLIT - 1) / FIN_SPLIT, c0 = min(part * per, n);
        src += (int64_t)c0 * stride;
        n = min(per, n - c0);
      }
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int c = 0;
      // the kernel is bound by memory round trips, not bandwidth: 32 partials (one whole dW2 element at the default batch split) are in
      // flight before the first add; the adds keep the chunk order 0, 1, 2, ... whatever the unroll
      for (; c + 32 <= n; c += 32) {
        f32x4 v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = load16<KP1_FIN_LD_NT>(src + (int64_t)(c + u) * stride);
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
      }
      for (; c + 16 <= n; c += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = load16<KP1_FIN_LD_NT>(src + (int64_t)(c + u) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
      for (; c + 8 <= n; c += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load16<KP1_FIN_LD_NT>(src + (int64_t)(c + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; c < n; ++c) s += load16<KP1_FIN_LD_NT>(src + (int64_t)c * stride);
      if (FIN_SPLIT > 1) {   // parts added in part order: ((p0 + p1) + p2) + p3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = __shfl(s[q], (threadIdx.x & 63) - part);
#pragma unroll
          for (int k = 1; k < FIN_SPLIT; ++k) t += __shfl(s[q], (threadIdx.x & 63) - part + k);
          s[q] = t;
        }
      }
      if (part == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a.grad[dst + q] = s[q];
          gsq += (double)s[q] * (double)s[q];
        }
      }
    }
    if (a.step_counter && blockIdx.x == 0 && threadIdx.x == 0) *a.step_counter += 1;  // read by the adam kernel that follows
    if (a.stats && blockIdx.x == 0 && threadIdx.x == 3) {  // entropy of the state-independent diagonal Gaussian
      float ent = 0.f;
      for (int k = 0; k < ACT; ++k) ent += 0.5f + LOG_SQRT_2PI + a.log_std[k];
      a.stats[2] += ent;
    }
  } else {
    const int64_t n_wide = finalize_wide_count(L);
    const int el = threadIdx.x & 31, strand = threadIdx.x >> 5;
    const int64_t e = (int64_t)(blockIdx.x - n_main) * 32 + el;
    // elements n_wide .. n_wide+2 are the three loss sums (policy, value, kl), not gradients
    const bool is_grad = e < n_wide, is_stat = !is_grad && e < n_wide + 3;
    PartialSrc s{};
    int64_t flat = 0;
    if (is_grad) {
      flat = finalize_wide_index(L, e);
      s = finalize_source(a, flat);
    } else if (is_stat) {
      s.p = a.hpart + 10 * L.Hp + 15 + (e - n_wide);
      s.stride = a.h_stride; s.n = a.h_n; s.add = 0.f;
    }
    float part = 0.f;
    if (is_grad || is_stat) {
      int c = strand;
      for (; c + 8 * 31 < s.n; c += 8 * 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = s.p[(int64_t)(c + 8 * u) * s.stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) part += v[u];
      }
      for (; c < s.n; c += 8) part += s.p[(int64_t)c * s.stride];
    }
    red[strand][el] = part;
    __syncthreads();
    if (threadIdx.x < 32) {
      float t = red[0][el];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += red[k][el];
      if (is_grad) {
        gval = t + s.add;
        a.grad[flat] = gval;
      } else if (is_stat && a.stats) {
        const int k = (int)(e - n_wide);
        a.stats[k == 2 ? 3 : k] += t * a.inv_count;  // single writer
      }
    }
  }
  // block sum of squares: shuffles inside a wave, the four wave sums through LDS (fixed order) -- one barrier instead of nine
  double w = gsq + (double)gval * (double)gval;
  for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
  if ((threadIdx.x & 63) == 0) sq[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) a.sumsq[blockIdx.x] = ((sq[0] + sq[1]) + sq[2]) + sq[3];
}

__global__ void __launch_bounds__(256) sumsq_partials_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partials) {
  __shared__ double s[256];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += (double)g[i] * (double)g[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = s[0];
}

// torch.nn.utils.clip_grad_norm_ (coef = max_norm / (norm + 1e-6), clamped to 1) + torch.optim.Adam
// The same pass repacks the updated element into the kernel-format weights and (zero_grad) clears the gradient, so the
// next minibatch's atomic accumulation starts from zero without a memset launch.
// VEC = 4: a thread owns four consecutive elements (float4 loads / stores of p, g, m, v; the caller checks 16-byte alignment): a quarter of
// the workgroups -- each of them sums the norm partials again -- and of the load instructions; per-element arithmetic unchanged.
#ifndef KP1_ADAM_VEC
#define KP1_ADAM_VEC 1     // 4 measured SLOWER (8.1 -> 12.9 us in situ, profiles/r02_ab_adam_vec4.log): the kernel is a latency chain per thread
#endif                     // (scattered repack stores behind div / sqrt), so it wants more threads, not fatter ones
#ifndef KP1_ADAM_BLOCK
#define KP1_ADAM_BLOCK 256
#endif
template <int VEC>
__global__ void __launch_bounds__(KP1_ADAM_BLOCK) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, const double* __restrict__ partials, int n_partials, float lr, float eps, float max_norm,
                                                   float bc1, float bc2_sqrt, const ParamLayout L, const Packed k, int zero_grad,
                                                   const int* __restrict__ step_counter, int host_step, const int* __restrict__ actor_extra) {
  __shared__ float scale_s;
  // this thread's elements: their loads do not depend on the norm, so they go out first and share one memory round trip with the
  // step count and the norm partials below (the kernel is a chain of round trips: it moves 2.6 MB)
  const int64_t i0 = ((int64_t)blockIdx.x * KP1_ADAM_BLOCK + threadIdx.x) * VEC;
  float g_in[VEC], m_in[VEC], v_in[VEC], p_in[VEC];
  if (VEC == 4 && i0 + 3 < n) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i0), mv = *reinterpret_cast<const f32x4*>(m + i0);
    const f32x4 vv = *reinterpret_cast<const f32x4*>(v + i0), pv = *reinterpret_cast<const f32x4*>(p + i0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { g_in[j] = gv[j]; m_in[j] = mv[j]; v_in[j] = vv[j]; p_in[j] = pv[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int64_t il = i0 + j < n ? i0 + j : n - 1;
      g_in[j] = g[il]; m_in[j] = m[il]; v_in[j] = v[il]; p_in[j] = p[il];
    }
  }
  const int extra = *actor_extra;
  float base_step = (float)host_step;
  if (step_counter) base_step = (float)*step_counter;
  double s = 0.0;
  if (threadIdx.x < 64) {
    for (int q = threadIdx.x; q < n_partials; q += 64) s += partials[q];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s);
    scale_s = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
  }
  if (step_counter) {  // bias corrections from the device-resident step count
    bc1 = 1.f - powf(0.9f, base_step);
    bc2_sqrt = sqrtf(1.f - powf(0.999f, base_step));
  }
  // torch.optim.Adam keeps one step count per tensor: the actor tensors (policy_net.*, action_net.*) have taken *actor_extra
  // more steps than the rest when a teacher-anchor side loss updates them between rollouts (route/teacher_anchor.py:68-87)
  float bc1_a = bc1, bc2_sqrt_a = bc2_sqrt;
  if (extra != 0) {
    const float st = base_step + (float)extra;
    bc1_a = 1.f - powf(0.9f, st);
    bc2_sqrt_a = sqrtf(1.f - powf(0.999f, st));
  }
  __syncthreads();
  const float scale = scale_s;
  float pn[VEC], mn[VEC], vn[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int64_t i = i0 + j;
    const bool actor = extra != 0 && ((i >= L.p_w1 && i < L.v_w1) || (i >= L.a_w && i < L.c_w));
    const float b1 = actor ? bc1_a : bc1, b2 = actor ? bc2_sqrt_a : bc2_sqrt;
    const float gi = g_in[j] * scale;
    mn[j] = 0.9f * m_in[j] + 0.1f * gi;
    vn[j] = 0.999f * v_in[j] + 0.001f * gi * gi;
    const float denom = sqrtf(vn[j]) / b2 + eps;
    pn[j] = p_in[j] - (lr / b1) * (mn[j] / denom);
  }
  if (VEC == 4 && i0 + 3 < n) {
    *reinterpret_cast<f32x4*>(m + i0) = f32x4{mn[0], mn[1], mn[2], mn[3]};
    *reinterpret_cast<f32x4*>(v + i0) = f32x4{vn[0], vn[1], vn[2], vn[3]};
    *reinterpret_cast<f32x4*>(p + i0) = f32x4{pn[0], pn[1], pn[2], pn[3]};
    if (zero_grad) *reinterpret_cast<f32x4*>(g + i0) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < VEC; ++j) pack_one(i0 + j, pn[j], L, k);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int64_t i = i0 + j;
      if (i < n) {
        m[i] = mn[j]; v[i] = vn[j]; p[i] = pn[j];
        pack_one(i, pn[j], L, k);
        if (zero_grad) g[i] = 0.f;
      }
    }
  }
}

#include "kp1_env_step.inc"
#include "kp1_mlp_tile.inc"
#include "kp1_mlp_fused.inc"

}  // namespace

// ============================================================================================ host
struct kp1_mlp {
  int device = 0, H = 0, Hp = 0, max_batch = 0;
  ParamLayout L;
  Packed k{};
  float* h1 = nullptr;   // [2][max_batch][Hp]
  float* h2 = nullptr;
  float* dz2 = nullptr;
  float* dz1 = nullptr;
  float* xf = nullptr;    // [max_batch][64] gathered observations, k8-fragment major (fused path)
  // [r3 experiment, KP1_MLP_OPT_BF16X3_WGRAD] the same four tensors as three bf16 planes each, allocated when the option is first switched on
  int bf16x3 = 0;
  unsigned short* sp_h1 = nullptr; unsigned short* sp_dz2 = nullptr; unsigned short* sp_dz1 = nullptr; unsigned short* sp_xf = nullptr;
  double* partials = nullptr;  // [512]
  float* slab = nullptr;       // [64 chunks][2 nets][Hp][Hp] partial dW2
  float* slab1 = nullptr;      // [64 chunks][2 nets][Hp][64] partial dW1
  float* bslab = nullptr;      // [max_batch/64 row tiles][2 nets][Hp] partial db1
  float* hpart = nullptr;      // [max_batch/32 blocks][10 Hp + 32] head partials
  int n_finalize_blocks = 0;   // sum-of-squares partials written by the last grad_finalize_kernel
  int* step_dev = nullptr;     // device Adam step counter
  int last_s2_n = 0, last_s1_n = 0;
  int fused = 1;               // KP1_MLP_OPT_FUSED: one workgroup carries a row tile through the whole chain (H = 256 only)
  // While the fused path is selected the Adam kernel refreshes only the fragment-major weight copies (the transposed / slab copies are
  // scattered 4-byte writes and were half of the kernel's HBM write traffic); the k-slab copies are then stale until the next full pack, which switching the
  // option off triggers from the parameter vector last seen.
  const float* last_params = nullptr;
  bool slab_stale = false;
  std::vector<void*> allocs;
  // KP1_MLP_OPT_PROFILE: a HIP-event pair attached to every launch of the optimiser step (hipExtLaunchKernelGGL: the dispatch's own begin
  // and end), on the launch stream, in the real launch sequence (tile -> weight gradients -> finalize -> Adam), read back by
  // kp1_mlp_profile_read.  Off in production (an eager epoch: events do not travel into a captured graph replay).
  int profile = 0;
  struct ProfPair { hipEvent_t a, b; };
  std::vector<ProfPair> prof[KP1_MLP_PROFILE_SLOTS];
  int prof_used[KP1_MLP_PROFILE_SLOTS] = {0, 0, 0, 0};
};

namespace {
// The event pair of the profiled launch in flight on this thread.  KP1_LAUNCH hands it to hipExtLaunchKernelGGL, which attaches both events
// to the kernel's own dispatch packet: their elapsed time is the dispatch's begin -> end, the interval rocprofv3 --kernel-trace reports.
// (Round 2 recorded the events as separate stream markers around the launch; that interval also holds the marker -> dispatch hand-over,
// 2-4 us on this stack, and read 50.6 us where rocprofv3 read 48.5 for the same launches.)  A scope whose launch site does not use
// KP1_LAUNCH falls back to the marker pair.
struct ProfSlotInFlight { hipEvent_t a = nullptr, b = nullptr; bool attached = false; };
thread_local ProfSlotInFlight tl_prof;

struct ProfScope {
  kp1_mlp* m; int slot; hipStream_t stream; bool on;
  ProfScope(kp1_mlp* m_, int slot_, hipStream_t s_) : m(m_), slot(slot_), stream(s_), on(m_->profile != 0) {
    if (!on) return;
    auto& v = m->prof[slot];
    if (m->prof_used[slot] >= (int)v.size()) {
      kp1_mlp::ProfPair p{};
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
      v.push_back(p);
    }
    tl_prof.a = v[m->prof_used[slot]].a;
    tl_prof.b = v[m->prof_used[slot]].b;
    tl_prof.attached = false;
  }
  ~ProfScope() {
    if (!on) return;
    if (!tl_prof.attached) {   // nothing was launched through KP1_LAUNCH inside the scope: an empty marker pair (reads ~0)
      (void)hipEventRecord(tl_prof.a, stream);
      (void)hipEventRecord(tl_prof.b, stream);
    }
    tl_prof = ProfSlotInFlight{};
    m->prof_used[slot] += 1;
  }
};

// launch of a kernel that may be inside a ProfScope
#define KP1_LAUNCH(kernel, grid, block, bytes, stream, ...)                                                        \
  do {                                                                                                             \
    if (tl_prof.a != nullptr && !tl_prof.attached) {                                                               \
      tl_prof.attached = true;                                                                                     \
      hipExtLaunchKernelGGL(kernel, grid, block, bytes, stream, tl_prof.a, tl_prof.b, 0, __VA_ARGS__);             \
    } else {                                                                                                       \
      hipLaunchKernelGGL(kernel, grid, block, bytes, stream, __VA_ARGS__);                                         \
    }                                                                                                              \
  } while (0)
}  // namespace

namespace {

constexpr int N_PARTIALS = 128;
#ifndef KP1_TN_FORM
#define KP1_TN_FORM 1      // 0: gemm_tn_frag_kernel (128 x 128 tiles, 32 batch chunks), 1: gemm_tn_split_kernel (64 x 64 tiles, 8 chunks, waves split the rows; default: profiles/r02_ab_tn_wave_split.log)
#endif
#ifndef KP1_TN_SPLIT2
#if KP1_TN_FORM == 1
#define KP1_TN_SPLIT2 8    // batch chunks of the dW2 / dW1 partial tiles
#define KP1_TN_SPLIT1 16
#else
#define KP1_TN_SPLIT2 32
#define KP1_TN_SPLIT1 64
#endif
#endif

// rows of the batch each TN workgroup reduces: aim at ~256 workgroups (one per CU) for the H x H gradient
int tn_chunk_rows(int n, int tiles) {
  int chunks = (256 + tiles - 1) / tiles;
  if (chunks > 64) chunks = 64;     // slab capacity (kp1_mlp_create)
  if (chunks < 1) chunks = 1;
  int rows = (n + chunks - 1) / chunks;
  rows = (rows + 63) / 64 * 64;     // whole 64-row stages
  return rows < 64 ? 64 : rows;
}

int mlp_check_device(const kp1_mlp* m) {
  HIP_TRY(hipSetDevice(m->device));
  return KP1_OK;
}

constexpr size_t TN_LDS_BYTES = sizeof(float) * 2 * 2 * 64 * (128 + 4);

// Tile shapes.  Wide (whole hidden width per workgroup, large batches): 64 rows x 256 columns, 32-deep W stages, 512
// threads (two waves per SIMD split the epilogue), one workgroup per CU.  Narrow (small batches, more workgroups):
// 64 rows x 128 columns, 256 threads.  Measured alternatives at M = 8192 (all within 7 %): 32 x 256 tiles with 16-deep
// stages and two workgroups per CU 29.9 us, 64 x 256 / 256 threads 28.9 us, this one 28.0 us -- the three GEMM kinds all
// plateau near 75 TFLOP/s, see DESIGN.md section 5.
template <int BN, int EPI, int K>
int launch_nt_inst(const GemmNT& g, hipStream_t stream) {
  constexpr int NTH = BN == 256 ? 512 : 256;
  constexpr int BM = 64;
  constexpr int BKS = 32;
  size_t bytes = sizeof(float) * (BM * (size_t)(K + 4) + 2 * (size_t)BN * (BKS + 4));
  const size_t epi = sizeof(float) * (BM * (size_t)(BN + 4) + (size_t)(NTH / (BN / 4)) * BN);  // C tile + bias-partial scratch
  if (epi > bytes) bytes = epi;
  HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<BN, EPI, K, NTH, BM, BKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  hipLaunchKernelGGL((gemm_nt_kernel<BN, EPI, K, NTH, BM, BKS>), dim3((g.M + BM - 1) / BM, g.N / BN, 2), dim3(NTH), bytes, stream, g);
  return KP1_OK;
}

// number of row tiles launch_nt uses for an n-row GEMM with N = Hp columns (the bias-partial count of the backward GEMM)
int nt_row_tiles(int n, int Hp) {
  (void)Hp;
  return (n + 63) / 64;
}

template <int EPI>
int launch_nt(const GemmNT& g, hipStream_t stream) {
  if (g.N % 128 != 0) return fail(KP1_ERR_INVALID, "gemm_nt needs N % 128 == 0");
  // 256-column workgroups (A read once per row block) when that still fills the chip, else 128-column ones
  const int row_tiles = (g.M + 63) / 64;
  const bool wide = (g.N % 256 == 0) && row_tiles * 2 >= 200;
  if (g.K == 64) return wide ? launch_nt_inst<256, EPI, 64>(g, stream) : launch_nt_inst<128, EPI, 64>(g, stream);
  if (g.K == 128) return wide ? launch_nt_inst<256, EPI, 128>(g, stream) : launch_nt_inst<128, EPI, 128>(g, stream);
  if (g.K == 256) return wide ? launch_nt_inst<256, EPI, 256>(g, stream) : launch_nt_inst<128, EPI, 256>(g, stream);
  return fail(KP1_ERR_UNSUPPORTED, "gemm_nt is instantiated for K in {64, 128, 256}");
}

// dW (both nets) = D^T X over n rows: split-B partial tiles into m->slab, then the fixed-order reduce into G
int launch_tn(kp1_mlp* m, GemmTN t, int n_o_tiles, int slab_cols, float* slab, int* n_chunks_out, hipStream_t stream);

int launch_fused(const FusedArgs& fa_in, hipStream_t stream) {
  using G = FuGeom<true>;
  FusedArgs fa = fa_in;
  static const int stagger_us = [] { const char* e = std::getenv("KP1_FU_STAGGER_US"); return e ? std::atoi(e) : 11; }();  // tuning knob; re-swept in round 3 (profiles/r03_ab_tile_stagger.log): 10-13 us within 0.4 us of each other, a cliff at 14 (+6 us)
  static const int n_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n;
  }();
  const dim3 grid((fa.n + G::BM - 1) / G::BM, 1, 2);
  fa.n_cus = n_cus;
  fa.stagger_ticks = (G::RB == 1 && (int)(grid.x * grid.z) > n_cus) ? stagger_us * 100 : 0;   // only when CUs hold two workgroups at once
  const size_t bytes = sizeof(float) * G::LDS_FLOATS;
  if (fa.sp_h1 != nullptr) {   // [r3 experiment] bf16 x 3 planes instead of the fp32 activation copies
    if (fa.inp == 64) {
      HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<true, 2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      KP1_LAUNCH((mlp_tile_kernel<true, 2, 0, true>), grid, dim3(G::NTH), bytes, stream, fa);
    } else {
      HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<true, 4, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      KP1_LAUNCH((mlp_tile_kernel<true, 4, 0, true>), grid, dim3(G::NTH), bytes, stream, fa);
    }
    return KP1_OK;
  }
  if (fa.inp == 64) {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    KP1_LAUNCH((mlp_tile_kernel<true, 2>), grid, dim3(G::NTH), bytes, stream, fa);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    KP1_LAUNCH((mlp_tile_kernel<true, 4>), grid, dim3(G::NTH), bytes, stream, fa);
  }
  return KP1_OK;
}

// policy forward + env step in one launch (kp1_mlp_forward_env_step): env_mode = KP1_MODE_APPROACH / KP1_MODE_DOCK, 64-float observation rows
int launch_fused_infer_env(const FusedArgs& fa, int env_mode, hipStream_t stream) {
  using G = FuGeom<false>;
  const size_t bytes = sizeof(float) * G::LDS_FLOATS;
  const dim3 grid((fa.n + G::BM - 1) / G::BM, 1, fa.value ? 2 : 1);
  if (env_mode == KP1_MODE_DOCK) {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((mlp_tile_kernel<false, 2, 2>), grid, dim3(G::NTH), bytes, stream, fa);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((mlp_tile_kernel<false, 2, 1>), grid, dim3(G::NTH), bytes, stream, fa);
  }
  return KP1_OK;
}

int launch_fused_infer(const FusedArgs& fa, hipStream_t stream) {
  using G = FuGeom<false>;  // 32-row tiles, 4 waves, two workgroups per CU
  const size_t bytes = sizeof(float) * G::LDS_FLOATS;
  // the value net is skipped when no value is asked for (deterministic evaluators), the policy net when only values are
  const bool want_pi = fa.mean || fa.action || fa.clipped || fa.log_prob;
  if (!want_pi && !fa.value) return KP1_OK;
  FusedArgs f = fa;
  const bool both = want_pi && fa.value;
  if (!both) f.net_base = want_pi ? 0 : 1;
  const dim3 grid((fa.n + G::BM - 1) / G::BM, 1, both ? 2 : 1);
  if (fa.inp == 64) {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((mlp_tile_kernel<false, 2>), grid, dim3(G::NTH), bytes, stream, f);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)mlp_tile_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((mlp_tile_kernel<false, 4>), grid, dim3(G::NTH), bytes, stream, f);
  }
  return KP1_OK;
}

int launch_tn_frag(const TnFragArgs& t, hipStream_t stream) {
  if (t.n_chunks2 > 64 || t.n_chunks1 > 64) return fail(KP1_ERR_INVALID, "too many batch chunks for the partial-gradient slabs");
#if KP1_TN_FORM == 1
  {
    const size_t bytes = sizeof(float) * TN_SPLIT_LDS_FLOATS;
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    KP1_LAUNCH(gemm_tn_split_kernel, dim3(32 * t.n_chunks2 + 16 * t.n_chunks1), dim3(256), bytes, stream, t);
    return KP1_OK;
  }
#endif
  const size_t bytes = sizeof(float) * 128 * (128 + 4);
  HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_frag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  hipLaunchKernelGGL(gemm_tn_frag_kernel, dim3(8 * t.n_chunks2 + 4 * t.n_chunks1), dim3(256), bytes, stream, t);
  return KP1_OK;
}

// forward layers 1 and 2 for n rows (both nets): h1, h2 filled
int launch_forward_layers(kp1_mlp* m, const float* obs, int obs_stride, const int64_t* idx, int n, hipStream_t stream) {
  const int Hp = m->Hp, IN = m->L.IN, INP = m->L.INP;
  const int64_t act_stride = (int64_t)m->max_batch * Hp;
  GemmNT g{};
  g.A = obs; g.lda = obs_stride; g.strideA = 0; g.gather = idx;
  g.W = m->k.w1p; g.strideW = (int64_t)Hp * INP;
  g.bias = m->k.b1; g.strideBias = Hp;
  g.C = m->h1; g.ldc = Hp; g.strideC = act_stride;
  g.aux = nullptr; g.strideAux = 0; g.colsum = nullptr; g.strideColsum = 0;
  g.M = n; g.N = Hp; g.K = INP; g.Kreal = obs_stride >= INP ? INP : IN;
  int rc = launch_nt<EPI_BIAS_TANH>(g, stream);
  if (rc != KP1_OK) return rc;
  g.A = m->h1; g.lda = Hp; g.strideA = act_stride; g.gather = nullptr;
  g.W = m->k.w2; g.strideW = (int64_t)Hp * Hp;
  g.bias = m->k.b2;
  g.C = m->h2;
  g.K = Hp; g.Kreal = Hp;
  rc = launch_nt<EPI_BIAS_TANH>(g, stream);
  if (rc != KP1_OK) return rc;
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

}  // namespace

namespace {
int launch_tn(kp1_mlp* m, GemmTN t, int n_o_tiles, int slab_cols, float* slab, int* n_chunks_out, hipStream_t stream) {
  const int Hp = m->Hp;
  const int n_chunks = (t.B + t.chunk - 1) / t.chunk;
  if (n_chunks > 64) return fail(KP1_ERR_INVALID, "too many batch chunks for the partial-gradient slab");
  t.slab = slab;
  t.slab_ld = slab_cols;
  t.slab_net_stride = (int64_t)Hp * slab_cols;
  t.slab_chunk_stride = 2 * t.slab_net_stride;
  const dim3 grid(n_chunks, n_o_tiles * t.n_i_tiles, 2);
  if (t.gatherX) {
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES));
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), TN_LDS_BYTES, stream, t);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES));
    hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), TN_LDS_BYTES, stream, t);
  }
  if (n_chunks_out) *n_chunks_out = n_chunks;
  return KP1_OK;
}
}  // namespace

extern "C" {

int64_t kp1_mlp_num_params(int32_t hidden) { return hidden > 0 ? make_layout(hidden).total : 0; }
int64_t kp1_mlp_num_params_ex(int32_t hidden, int32_t obs_dim) { return hidden > 0 && obs_dim > 0 && obs_dim <= 128 ? make_layout(hidden, obs_dim).total : 0; }

int kp1_mlp_create(int32_t device, int32_t hidden, int32_t max_batch, kp1_mlp** out) { return kp1_mlp_create_ex(device, hidden, KP1_MLP_IN, max_batch, out); }

int kp1_mlp_create_ex(int32_t device, int32_t hidden, int32_t obs_dim, int32_t max_batch, kp1_mlp** out) {
  if (!out || hidden <= 0 || hidden > 1024 || max_batch <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_create");
  if (obs_dim != KP1_MLP_IN && obs_dim != KP1_MLP_IN_ROUTE)
    return fail(KP1_ERR_UNSUPPORTED, "obs_dim must be 56 (ArmKinematicEnv) or 80 (route observation keys)");
  // 256: the fused tile path (BASELINE config 2).  128 and 64 (SB3's default net_arch, what the reference trains and what its checkpoints
  // hold): the layer-wise MFMA kernels on a 128-wide padded layout -- padded hidden units have zero weights and biases, so they stay at
  // tanh(0) = 0 in the forward pass and receive exactly zero gradient; finalize / Adam / the gradient norm only ever touch the H real rows.
  if (hidden != 64 && hidden != 128 && hidden != 256)
    return fail(KP1_ERR_UNSUPPORTED, "hidden must be 64 (SB3 default), 128 or 256 (the widths the MFMA kernels are instantiated for)");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= count) return fail(KP1_ERR_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  kp1_mlp* m = new kp1_mlp();
  m->device = device;
  m->H = hidden;
  m->L = make_layout(hidden, obs_dim);
  m->Hp = m->L.Hp;
  m->max_batch = (max_batch + 127) / 128 * 128;
  const int64_t Hp = m->Hp, mb = m->max_batch, INP = m->L.INP;
  auto alloc = [&](void** p, size_t bytes) -> int {
    if (hipMalloc(p, bytes) != hipSuccess) return fail(KP1_ERR_ALLOC, "hipMalloc failed in kp1_mlp_create");
    m->allocs.push_back(*p);
    return hipMemset(*p, 0, bytes) == hipSuccess ? KP1_OK : fail(KP1_ERR_NO_DEVICE, "hipMemset failed");
  };
  int rc = KP1_OK;
#define MLP_ALLOC(ptr, count) if (rc == KP1_OK) rc = alloc((void**)&(ptr), sizeof(*(ptr)) * (size_t)(count))
  MLP_ALLOC(m->k.w1p, 2 * Hp * INP);
  MLP_ALLOC(m->k.b1, 2 * Hp);
  MLP_ALLOC(m->k.w2, 2 * Hp * Hp);
  MLP_ALLOC(m->k.w2t, 2 * Hp * Hp);
  MLP_ALLOC(m->k.w1f, 2 * Hp * INP);
  MLP_ALLOC(m->k.w2f, 2 * Hp * Hp);
  MLP_ALLOC(m->k.w2tf, 2 * Hp * Hp);
  MLP_ALLOC(m->k.b2, 2 * Hp);
  MLP_ALLOC(m->k.w3, HEADS * Hp);
  MLP_ALLOC(m->k.b3, HEADS);
  MLP_ALLOC(m->k.log_std, HEADS);
  MLP_ALLOC(m->h1, 2 * mb * Hp);
  MLP_ALLOC(m->h2, 2 * mb * Hp);
  MLP_ALLOC(m->dz2, 2 * mb * Hp);
  MLP_ALLOC(m->dz1, 2 * mb * Hp);
  MLP_ALLOC(m->xf, mb * INP);
  MLP_ALLOC(m->partials, 2 * N_PARTIALS + 2048);
  MLP_ALLOC(m->slab, (int64_t)64 * 2 * Hp * Hp);
  MLP_ALLOC(m->slab1, (int64_t)64 * 2 * Hp * INP);
  MLP_ALLOC(m->step_dev, 4);
  MLP_ALLOC(m->bslab, (mb / 32) * 2 * Hp);
  MLP_ALLOC(m->hpart, (mb / 32) * (10 * Hp + 32));
#undef MLP_ALLOC
  if (rc != KP1_OK) {
    kp1_mlp_destroy(m);
    return rc;
  }
  *out = m;
  return KP1_OK;
}

int kp1_mlp_destroy(kp1_mlp* m) {
  if (!m) return KP1_OK;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  for (void* p : m->allocs) (void)hipFree(p);
  for (void* p : {(void*)m->sp_h1, (void*)m->sp_dz2, (void*)m->sp_dz1, (void*)m->sp_xf}) (void)hipFree(p);
  for (auto& v : m->prof)
    for (auto& pr : v) {
      (void)hipEventDestroy(pr.a);
      (void)hipEventDestroy(pr.b);
    }
  delete m;
  return KP1_OK;
}

int kp1_mlp_set_option(kp1_mlp* m, int32_t option, int32_t value) {
  if (!m) return fail(KP1_ERR_INVALID, "NULL argument");
  if (option == KP1_MLP_OPT_FUSED) {
    m->fused = value ? 1 : 0;
    if (!m->fused && m->slab_stale && m->last_params) {   // the layer-wise kernels read the k-slab copies: bring them up to date
      HIP_TRY(hipSetDevice(m->device));
      HIP_TRY(hipDeviceSynchronize());
      Packed k = m->k;
      k.formats = PACK_SLAB | PACK_FRAG;
      hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((m->L.total + 255) / 256)), dim3(256), 0, (hipStream_t)0, m->last_params, m->L, k);
      HIP_TRY(kp1::launch_status());
      HIP_TRY(hipDeviceSynchronize());
      m->slab_stale = false;
    }
    return KP1_OK;
  }
  if (option == KP1_MLP_OPT_BF16X3_WGRAD) {
    if (value && !(m->fused && m->Hp == FU_HP)) return fail(KP1_ERR_UNSUPPORTED, "the bf16x3 weight-gradient experiment needs the 2x256 tile kernels");
    if (value && !m->sp_h1) {
      HIP_TRY(hipSetDevice(m->device));
      const size_t act = (size_t)3 * 2 * m->max_batch * m->Hp * sizeof(unsigned short), xb = (size_t)3 * m->max_batch * m->L.INP * sizeof(unsigned short);
      for (unsigned short** p : {&m->sp_h1, &m->sp_dz2, &m->sp_dz1}) {
        HIP_TRY(hipMalloc((void**)p, act));
        HIP_TRY(hipMemset(*p, 0, act));
      }
      HIP_TRY(hipMalloc((void**)&m->sp_xf, xb));
      HIP_TRY(hipMemset(m->sp_xf, 0, xb));
    }
    m->bf16x3 = value ? 1 : 0;
    return KP1_OK;
  }
  if (option == KP1_MLP_OPT_PROFILE) {
    m->profile = value ? 1 : 0;
    for (int k = 0; k < KP1_MLP_PROFILE_SLOTS; ++k) m->prof_used[k] = 0;
    return KP1_OK;
  }
  if (option == KP1_MLP_OPT_STEP_COUNT) {   // optimiser steps taken so far (resuming from a checkpoint's Adam state)
    if (value < 0) return fail(KP1_ERR_INVALID, "step count must be >= 0");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpy(m->step_dev, &value, sizeof(int), hipMemcpyHostToDevice));
    return KP1_OK;
  }
  if (option == KP1_MLP_OPT_ACTOR_EXTRA_STEPS) {
    if (value < 0) return fail(KP1_ERR_INVALID, "actor extra steps must be >= 0");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpy(m->step_dev + 1, &value, sizeof(int), hipMemcpyHostToDevice));
    return KP1_OK;
  }
  return fail(KP1_ERR_INVALID, "unknown kp1_mlp option");
}

int kp1_mlp_pack_weights(kp1_mlp* m, const float* params, void* stream) {
  if (!m || !params) return fail(KP1_ERR_INVALID, "NULL argument");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  Packed k = m->k;
  k.formats = PACK_SLAB | PACK_FRAG;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((m->L.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, m->L, k);
  HIP_TRY(kp1::launch_status());
  m->last_params = params;
  m->slab_stale = false;
  return KP1_OK;
}

int kp1_mlp_forward(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, const float* noise, float* mean, float* value, float* action,
                    float* clipped_action, float* log_prob, void* stream) {
  if (!m || !obs) return fail(KP1_ERR_INVALID, "NULL argument");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  const int IN = m->L.IN, INP = m->L.INP;
  if (obs_stride != IN && obs_stride != INP) return fail(KP1_ERR_INVALID, "obs_stride must be the observation width or its padded width (56 / 64, or 80 / 128)");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  if (m->fused && m->Hp == FU_HP) {
    FusedArgs fa{};
    fa.obs = obs; fa.obs_stride = obs_stride; fa.Kreal = obs_stride >= INP ? INP : IN; fa.inp = INP; fa.idx = nullptr; fa.n = n;
    fa.k = m->k;
    fa.noise = noise; fa.mean = mean; fa.value = value; fa.action = action; fa.clipped = clipped_action; fa.log_prob = log_prob;
    rc = launch_fused_infer(fa, (hipStream_t)stream);
    if (rc != KP1_OK) return rc;
    HIP_TRY(kp1::launch_status());
    return KP1_OK;
  }
  rc = launch_forward_layers(m, obs, obs_stride, nullptr, n, (hipStream_t)stream);
  if (rc != KP1_OK) return rc;
  HeadArgs a{};
  a.h2 = m->h2; a.strideH = (int64_t)m->max_batch * m->Hp; a.Hp = m->Hp; a.n = n; a.H = m->H;
  a.w3 = m->k.w3; a.b3 = m->k.b3; a.log_std = m->k.log_std;
  a.noise = noise; a.mean = mean; a.value = value; a.action = action; a.clipped = clipped_action; a.log_prob = log_prob;
  hipLaunchKernelGGL(head_infer_kernel, dim3((n + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), sizeof(float) * HEADS * m->Hp, (hipStream_t)stream, a);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_mlp_forward_env_step(kp1_mlp* m, kp1_env* env, const float* obs, int32_t obs_stride, const float* noise, float* value, float* action,
                             float* log_prob, float* next_obs, float* reward, uint8_t* done, float* terminal_obs, void* stream) {
  if (!m || !env || !obs || !action || !next_obs || !reward || !done) return fail(KP1_ERR_INVALID, "NULL argument to kp1_mlp_forward_env_step");
  if (!(m->fused && m->Hp == FU_HP) || m->L.INP != 64)
    return fail(KP1_ERR_UNSUPPORTED, "kp1_mlp_forward_env_step needs the 2x256 tile kernels and the 56-float observation (padded to 64)");
  if (obs_stride != m->L.IN && obs_stride != m->L.INP) return fail(KP1_ERR_INVALID, "obs_stride must be 56 or 64");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  FusedArgs fa{};
  int mode = 0, device = 0;
  int64_t n_envs = 0;
  rc = kp1::env_step_args_f32(env, &fa.env, sizeof fa.env, nullptr, next_obs, reward, done, terminal_obs, 1, &mode, &n_envs, &device);
  if (rc != KP1_OK) return rc;
  if (device != m->device) return fail(KP1_ERR_INVALID, "the env handle and the MLP workspace live on different devices");
  if (n_envs > m->max_batch) return fail(KP1_ERR_INVALID, "more envs than the workspace max_batch");
  fa.obs = obs; fa.obs_stride = obs_stride; fa.Kreal = obs_stride >= m->L.INP ? m->L.INP : m->L.IN; fa.inp = m->L.INP; fa.idx = nullptr; fa.n = (int)n_envs;
  fa.k = m->k;
  fa.noise = noise; fa.value = value; fa.action = action; fa.log_prob = log_prob;
  rc = launch_fused_infer_env(fa, mode, (hipStream_t)stream);
  if (rc != KP1_OK) return rc;
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_mlp_loss_grad(kp1_mlp* m, const float* obs, int32_t obs_stride, const int64_t* idx, int32_t n, const float* actions,
                      const float* old_log_prob, const float* advantages, const float* returns, float adv_mean, float adv_inv_std,
                      const float* adv_stats_dev, float clip_range, float ent_coef, float vf_coef, float inv_count, float* grad_out,
                      float* stats_out, int32_t grad_is_zero, void* stream_) {
  if (!m || !obs || !actions || !old_log_prob || !advantages || !returns || !grad_out) return fail(KP1_ERR_INVALID, "NULL argument");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  const int IN = m->L.IN, INP = m->L.INP;
  if (obs_stride != IN && obs_stride != INP) return fail(KP1_ERR_INVALID, "obs_stride must be the observation width or its padded width (56 / 64, or 80 / 128)");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int Hp = m->Hp, H = m->H;
  const int64_t act_stride = (int64_t)m->max_batch * Hp;
  const ParamLayout& L = m->L;
  (void)grad_is_zero;  // every gradient element is written (not accumulated) by grad_finalize_kernel
  const bool fused = m->fused && Hp == FU_HP;
  int adv_mode = 0;
  if (adv_stats_dev) adv_mode = 2;
  else if (adv_inv_std > 0.f) adv_mode = 3;
  else if (adv_inv_std == 0.f) adv_mode = 1;
  if (adv_mode == 1) hipLaunchKernelGGL(adv_partials_kernel, dim3(N_PARTIALS), dim3(256), 0, stream, advantages, idx, n, m->partials);
  const int hpart_stride = 10 * Hp + 32;
  if (fused) {
    FusedArgs fa{};
    fa.obs = obs; fa.obs_stride = obs_stride; fa.Kreal = obs_stride >= INP ? INP : IN; fa.inp = INP; fa.idx = idx; fa.n = n;
    fa.k = m->k;
    fa.actions = actions; fa.old_logp = old_log_prob; fa.adv = advantages; fa.ret = returns;
    fa.adv_partials = m->partials; fa.n_adv_partials = N_PARTIALS; fa.adv_stats = adv_stats_dev; fa.adv_mean = adv_mean; fa.adv_inv_std = adv_inv_std;
    fa.adv_mode = adv_mode;
    fa.clip_range = clip_range; fa.vf_coef = vf_coef; fa.inv_count = inv_count;
    fa.h1 = m->h1; fa.dz2 = m->dz2; fa.dz1 = m->dz1; fa.act_stride = act_stride; fa.xf = m->xf;
    fa.bslab = m->bslab; fa.hpart = m->hpart; fa.hpart_stride = hpart_stride;
    if (m->bf16x3) {
      fa.sp_h1 = m->sp_h1; fa.sp_dz2 = m->sp_dz2; fa.sp_dz1 = m->sp_dz1; fa.sp_xf = m->sp_xf;
      fa.sp_plane = (int64_t)2 * m->max_batch * Hp; fa.sp_plane_x = (int64_t)m->max_batch * INP;
    }
    {
      ProfScope ps(m, KP1_MLP_PROFILE_TILE, stream);
      rc = launch_fused(fa, stream);
    }
    if (rc != KP1_OK) return rc;
  } else {
    rc = launch_forward_layers(m, obs, obs_stride, idx, n, stream);
    if (rc != KP1_OK) return rc;
    HeadArgs a{};
    a.h2 = m->h2; a.strideH = act_stride; a.Hp = Hp; a.n = n; a.H = H;
    a.w3 = m->k.w3; a.b3 = m->k.b3; a.log_std = m->k.log_std;
    a.idx = idx; a.actions = actions; a.old_logp = old_log_prob; a.adv = advantages; a.ret = returns;
    a.adv_partials = m->partials; a.n_adv_partials = N_PARTIALS; a.adv_stats = adv_stats_dev; a.adv_mean = adv_mean; a.adv_inv_std = adv_inv_std;
    a.adv_mode = adv_mode;
    a.clip_range = clip_range; a.ent_coef = ent_coef; a.vf_coef = vf_coef; a.inv_count = inv_count;
    a.dz2 = m->dz2;
    a.hpart = m->hpart; a.hpart_stride = hpart_stride;
    {
      const dim3 hgrid((n + HEAD_ROWS - 1) / HEAD_ROWS);
      const size_t hbytes = sizeof(float) * (HEADS * Hp + HEAD_ROWS * 8 + 84 + 2 * HEAD_ROWS * (Hp + 4));
      if (Hp == 256) {
        HIP_TRY(hipFuncSetAttribute((const void*)head_train_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hbytes));
        hipLaunchKernelGGL(head_train_kernel<256>, hgrid, dim3(256), hbytes, stream, a);
      } else {
        HIP_TRY(hipFuncSetAttribute((const void*)head_train_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hbytes));
        hipLaunchKernelGGL(head_train_kernel<128>, hgrid, dim3(256), hbytes, stream, a);
      }
    }
    // dZ1 = (dZ2 W2) * (1 - h1^2), bias-1 gradient partials = per-row-tile column sums of dZ1
    GemmNT g{};
    g.A = m->dz2; g.lda = Hp; g.strideA = act_stride; g.gather = nullptr;
    g.W = m->k.w2t; g.strideW = (int64_t)Hp * Hp;
    g.bias = nullptr; g.strideBias = 0;
    g.C = m->dz1; g.ldc = Hp; g.strideC = act_stride;
    g.aux = m->h1; g.strideAux = act_stride;
    g.colsum = m->bslab; g.strideColsum = Hp;
    g.M = n; g.N = Hp; g.K = Hp; g.Kreal = Hp;
    rc = launch_nt<EPI_DTANH>(g, stream);
    if (rc != KP1_OK) return rc;
  }

  // weight gradients (split over the batch axis)
  int s2_n = 0, s1_n = 0;
  if (fused) {
    TnFragArgs t{};
    t.inp = INP;
    t.dz2 = m->dz2; t.h1 = m->h1; t.dz1 = m->dz1; t.act_stride = act_stride; t.xf = m->xf;
    t.slab2 = m->slab; t.s2_net = (int64_t)Hp * Hp; t.s2_chunk = 2 * t.s2_net;
    t.slab1 = m->slab1; t.s1_net = (int64_t)Hp * INP; t.s1_chunk = 2 * t.s1_net;
    t.groups = (n + FU_BM_TRAIN - 1) / FU_BM_TRAIN * (FU_BM_TRAIN / 8);   // the tile kernel writes whole 64-row tiles
    auto up8 = [](int v) { return (v + 7) / 8 * 8; };
    t.cg2 = up8((t.groups + KP1_TN_SPLIT2 - 1) / KP1_TN_SPLIT2);   // form 0: ~32 chunks x 8 tiles, form 1: 8 chunks x 32 tiles = one dW2 workgroup per CU
    t.cg1 = up8((t.groups + KP1_TN_SPLIT1 - 1) / KP1_TN_SPLIT1);   // form 0: ~64 chunks x 4 tiles, form 1: 16 chunks x 16 tiles of quarter-size dW1 workgroups
    t.n_chunks2 = (t.groups + t.cg2 - 1) / t.cg2;
    t.n_chunks1 = (t.groups + t.cg1 - 1) / t.cg1;
    if (m->bf16x3) {   // [r3 experiment] same slabs, same chunking in 16-row fragments
      TnBf16Args b{};
      b.dz2 = m->sp_dz2; b.h1 = m->sp_h1; b.dz1 = m->sp_dz1; b.plane = (int64_t)2 * m->max_batch * Hp; b.net = (int64_t)m->max_batch * Hp;
      b.xf = m->sp_xf; b.plane_x = (int64_t)m->max_batch * INP; b.inp = INP;
      b.slab2 = t.slab2; b.s2_net = t.s2_net; b.s2_chunk = t.s2_chunk; b.slab1 = t.slab1; b.s1_net = t.s1_net; b.s1_chunk = t.s1_chunk;
      b.frags = t.groups / 2; b.cf2 = t.cg2 / 2; b.cf1 = t.cg1 / 2; b.n_chunks2 = t.n_chunks2; b.n_chunks1 = t.n_chunks1;   // groups, cg2, cg1 are multiples of 4 / 8
      const size_t bytes = sizeof(float) * TN_SPLIT_LDS_FLOATS;
      HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      ProfScope ps(m, KP1_MLP_PROFILE_WGRAD, stream);
      KP1_LAUNCH(gemm_tn_bf16x3_kernel, dim3(32 * b.n_chunks2 + 16 * b.n_chunks1), dim3(256), bytes, stream, b);
    } else {
      ProfScope ps(m, KP1_MLP_PROFILE_WGRAD, stream);
      rc = launch_tn_frag(t, stream);
    }
    if (rc != KP1_OK) return rc;
    s2_n = t.n_chunks2;
    s1_n = t.n_chunks1;
  } else {
    GemmTN t{};
    t.B = n;
    t.chunk = tn_chunk_rows(n, (Hp / 128) * (Hp / 128) * 2);
    // dW2[o][i] = sum_b dZ2[b][o] h1[b][i]
    t.D = m->dz2; t.ldd = Hp; t.strideD = act_stride;
    t.X = m->h1; t.ldx = Hp; t.strideX = act_stride; t.gatherX = nullptr;
    t.Nload = Hp; t.n_i_tiles = Hp / 128;
    rc = launch_tn(m, t, Hp / 128, Hp, m->slab, &s2_n, stream);
    if (rc != KP1_OK) return rc;
    // dW1[o][i] = sum_b dZ1[b][o] x[b][i]   (i < 56)
    t.D = m->dz1;
    t.X = obs; t.ldx = obs_stride; t.strideX = 0; t.gatherX = idx;
    t.Nload = obs_stride >= INP ? INP : IN; t.n_i_tiles = 1;
    t.chunk = tn_chunk_rows(n, (Hp / 128) * 2);  // few tiles: more, shorter batch chunks keep the CUs busy
    rc = launch_tn(m, t, Hp / 128, INP, m->slab1, &s1_n, stream);
    if (rc != KP1_OK) return rc;
  }
  FinalizeArgs f{};
  f.L = L;
  f.slab2 = m->slab; f.s2_ld = Hp; f.s2_net = (int64_t)Hp * Hp; f.s2_chunk = 2 * f.s2_net; f.s2_n = s2_n;
  f.slab1 = m->slab1; f.s1_ld = INP; f.s1_net = (int64_t)Hp * INP; f.s1_chunk = 2 * f.s1_net; f.s1_n = s1_n;
  f.bslab = m->bslab; f.b_net = Hp; f.b_tile = 2 * Hp; f.b_n = fused ? (n + FU_BM_TRAIN - 1) / FU_BM_TRAIN * (FU_BM_TRAIN / 32) : nt_row_tiles(n, Hp);
  f.hpart = m->hpart; f.h_stride = 10 * Hp + 32; f.h_n = (n + HEAD_ROWS - 1) / HEAD_ROWS;
  f.ent_coef = ent_coef; f.inv_count = inv_count; f.log_std = m->k.log_std;
  f.grad = grad_out; f.stats = stats_out; f.sumsq = m->partials + 2 * N_PARTIALS;
  f.step_counter = m->step_dev;
  const int n_main = (int)((finalize_vec_items(L) * FIN_SPLIT + 255) / 256);
  m->n_finalize_blocks = n_main + (int)((finalize_wide_count(L) + 3 + 31) / 32);
  if (m->n_finalize_blocks > 2048) return fail(KP1_ERR_INVALID, "parameter vector too large for the sum-of-squares partial buffer");
  {
    ProfScope ps(m, KP1_MLP_PROFILE_FINALIZE, stream);
    KP1_LAUNCH(grad_finalize_kernel, dim3(m->n_finalize_blocks), dim3(256), 0, stream, f, n_main);
  }
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_mlp_profile_read(kp1_mlp* m, float* out_us, int32_t* out_launches) {
  if (!m || !out_us || !out_launches) return fail(KP1_ERR_INVALID, "NULL argument to kp1_mlp_profile_read");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  for (int k = 0; k < KP1_MLP_PROFILE_SLOTS; ++k) {
    double total_ms = 0.0;
    for (int i = 0; i < m->prof_used[k]; ++i) {
      HIP_TRY(hipEventSynchronize(m->prof[k][i].b));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, m->prof[k][i].a, m->prof[k][i].b));
      total_ms += ms;
    }
    out_launches[k] = m->prof_used[k];
    out_us[k] = m->prof_used[k] > 0 ? (float)(total_ms * 1e3 / m->prof_used[k]) : 0.f;
    m->prof_used[k] = 0;
  }
  return KP1_OK;
}

int kp1_mlp_time_kernels(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, int32_t iters, float* out_ms, double* out_flops,
                         void* stream_) {
  if (!m || !obs || !out_ms || !out_flops || iters <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_time_kernels");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int Hp = m->Hp, IN = m->L.IN, INP = m->L.INP;
  const int64_t act_stride = (int64_t)m->max_batch * Hp;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  rc = launch_forward_layers(m, obs, obs_stride, nullptr, n, stream);  // fills h1/h2 with real activations
  if (rc != KP1_OK) return rc;
  HIP_TRY(hipMemcpyAsync(m->dz2, m->h2, sizeof(float) * 2 * (size_t)act_stride, hipMemcpyDeviceToDevice, stream));
  auto time_it = [&](int which) -> int {
    HIP_TRY(hipEventRecord(e0, stream));
    for (int it = 0; it < iters; ++it) {
      GemmNT g{};
      g.gather = nullptr; g.ldc = Hp; g.strideC = act_stride; g.M = n; g.N = Hp;
      if (which == 0) {
        g.A = m->h1; g.lda = Hp; g.strideA = act_stride; g.W = m->k.w2; g.strideW = (int64_t)Hp * Hp; g.bias = m->k.b2; g.strideBias = Hp;
        g.C = m->h2; g.K = Hp; g.Kreal = Hp;
        if (launch_nt<EPI_BIAS_TANH>(g, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else if (which == 1) {
        g.A = m->dz2; g.lda = Hp; g.strideA = act_stride; g.W = m->k.w2t; g.strideW = (int64_t)Hp * Hp; g.C = m->dz1; g.aux = m->h1;
        g.strideAux = act_stride; g.K = Hp; g.Kreal = Hp;
        if (launch_nt<EPI_DTANH>(g, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else if (which == 2) {
        GemmTN t{};
        t.B = n; t.chunk = tn_chunk_rows(n, (Hp / 128) * (Hp / 128) * 2);
        t.D = m->dz2; t.ldd = Hp; t.strideD = act_stride; t.X = m->h1; t.ldx = Hp; t.strideX = act_stride;
        t.Nload = Hp; t.n_i_tiles = Hp / 128;
        if (launch_tn(m, t, Hp / 128, Hp, m->slab, nullptr, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else if (which == 3) {
        g.A = obs; g.lda = obs_stride; g.strideA = 0; g.W = m->k.w1p; g.strideW = (int64_t)Hp * INP; g.bias = m->k.b1; g.strideBias = Hp;
        g.C = m->h1; g.K = INP; g.Kreal = obs_stride >= INP ? INP : IN;
        if (launch_nt<EPI_BIAS_TANH>(g, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else if (which == 4) {
        // the fused tile kernel on this workspace: rows 0..n of obs, loss inputs = finite scratch values (layer-2 activations)
        FusedArgs fa{};
        fa.obs = obs; fa.obs_stride = obs_stride; fa.Kreal = obs_stride >= INP ? INP : IN; fa.inp = INP; fa.idx = nullptr; fa.n = n;
        fa.k = m->k;
        fa.actions = m->h2; fa.old_logp = m->h2 + (int64_t)ACT * n; fa.adv = m->h2 + (int64_t)(ACT + 1) * n; fa.ret = m->h2 + (int64_t)(ACT + 2) * n;
        fa.adv_mode = 3; fa.adv_mean = 0.f; fa.adv_inv_std = 1.f;
        fa.clip_range = 0.2f; fa.vf_coef = 0.5f; fa.inv_count = 1.f / n;
        fa.h1 = m->h1; fa.dz2 = m->dz2; fa.dz1 = m->dz1; fa.act_stride = act_stride; fa.xf = m->xf;
        fa.bslab = m->bslab; fa.hpart = m->hpart; fa.hpart_stride = 10 * Hp + 32;
        if (launch_fused(fa, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else {
        TnFragArgs t{};
    t.inp = INP;
        t.dz2 = m->dz2; t.h1 = m->h1; t.dz1 = m->dz1; t.act_stride = act_stride; t.xf = m->xf;
        t.slab2 = m->slab; t.s2_net = (int64_t)Hp * Hp; t.s2_chunk = 2 * t.s2_net;
        t.slab1 = m->slab1; t.s1_net = (int64_t)Hp * INP; t.s1_chunk = 2 * t.s1_net;
        t.groups = (n + FU_BM_TRAIN - 1) / FU_BM_TRAIN * (FU_BM_TRAIN / 8);   // the tile kernel writes whole 64-row tiles
        t.cg2 = ((t.groups + KP1_TN_SPLIT2 - 1) / KP1_TN_SPLIT2 + 7) / 8 * 8;
        t.cg1 = ((t.groups + KP1_TN_SPLIT1 - 1) / KP1_TN_SPLIT1 + 7) / 8 * 8;
        t.n_chunks2 = (t.groups + t.cg2 - 1) / t.cg2;
        t.n_chunks1 = (t.groups + t.cg1 - 1) / t.cg1;
        if (launch_tn_frag(t, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      }
    }
    HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    out_ms[which] = ms / iters;
    return KP1_OK;
  };
  for (int which : {3, 0, 1, 2}) {  // layer 1 first so h1 stays a valid activation for the others
    rc = time_it(which);
    if (rc != KP1_OK) return rc;
  }
  out_ms[4] = out_ms[5] = 0.f;
  out_flops[4] = out_flops[5] = 0.0;
  if (Hp == FU_HP) {  // the fused training path (it rewrites h1 / dz2 / dz1 in fragment-major order, so it runs last)
    for (int which : {4, 5}) {
      rc = time_it(which);
      if (rc != KP1_OK) return rc;
    }
  }
  const double M = n, H = Hp;
  out_flops[0] = 2.0 * 2.0 * M * H * H;
  out_flops[1] = 2.0 * 2.0 * M * H * H;
  out_flops[2] = 2.0 * 2.0 * M * H * H;
  out_flops[3] = 2.0 * 2.0 * M * H * INP;
  if (Hp == FU_HP) {
    out_flops[4] = 2.0 * 2.0 * M * (H * (double)INP + 2.0 * H * H);  // layer 1 + layer 2 + activation backward (heads: < 3 % more)
    out_flops[5] = 2.0 * 2.0 * M * (H * H + H * (double)INP);        // dW2 + dW1
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return KP1_OK;
}

// ---------------------------------------------------------------------------------------------- placement self-check
// Two SPEED assumptions of the update kernels rest on how the hardware places workgroups (never correctness): the training tile holds back
// the workgroups with linear index in [#CUs, 2 #CUs) on the assumption that workgroup k + #CUs lands on the CU of workgroup k (the second tile
// of a CU), and the weight-gradient kernel makes the batch chunk the fast block index on the assumption that block b runs on XCD b % 8.  This
// probe launches a kernel with the training tile's launch shape (grid, block, dynamic LDS, two workgroups per CU), keeps every workgroup resident
// until all have arrived (bounded wait), and records where each one ran.
__global__ void __launch_bounds__(FuGeom<true>::NTH, FuGeom<true>::WG_PER_CU * FuGeom<true>::NTH / 256) placement_probe_kernel(unsigned int* __restrict__ where, unsigned int* __restrict__ arrived,
                                                                                                                            unsigned int total) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave, SIMD, pipe, CU, SH, SE
    const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // HW_REG_XCC_ID[3:0]
    const unsigned int linear = blockIdx.x + gridDim.x * blockIdx.z;
    where[linear] = (xcc << 16) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);   // XCD | SE | SH | CU
    lds[0] = 0.f;
    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total && wall_clock64() - t0 < 200000ull) __builtin_amdgcn_s_sleep(8);   // <= 2 ms
  }
  __syncthreads();
}

int kp1_mlp_placement_check(int32_t device, int32_t n_rows, int32_t* out, void* stream_) {
  // out[0] workgroups, out[1] pairs (k, k + #CUs) probed, out[2] pairs on the same CU, out[3] workgroups on the XCD of workgroup (index % 8),
  // out[4] #CUs, out[5] distinct CUs used, out[6] workgroups that arrived while the probe waited, out[7] distinct XCDs among workgroups 0..7
  if (!out || n_rows <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_placement_check");
  HIP_TRY(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)stream_;
  using G = FuGeom<true>;
  int n_cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, device));
  const dim3 grid((n_rows + G::BM - 1) / G::BM, 1, 2);
  const unsigned int total = grid.x * grid.z;
  if ((int)total > G::WG_PER_CU * n_cus) return fail(KP1_ERR_UNSUPPORTED, "placement probe: the grid must be resident at once");
  unsigned int* dev = nullptr;
  HIP_TRY(hipMalloc(&dev, sizeof(unsigned int) * (total + 1)));
  HIP_TRY(hipMemsetAsync(dev, 0, sizeof(unsigned int) * (total + 1), stream));
  const size_t bytes = sizeof(float) * G::LDS_FLOATS;
  HIP_TRY(hipFuncSetAttribute((const void*)placement_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  hipLaunchKernelGGL(placement_probe_kernel, grid, dim3(G::NTH), bytes, stream, dev, dev + total, total);
  std::vector<unsigned int> host(total + 1);
  hipError_t e = hipMemcpyAsync(host.data(), dev, sizeof(unsigned int) * (total + 1), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dev);
  HIP_TRY(e);
  int pairs = 0, same = 0, rr = 0;
  std::vector<unsigned int> seen;
  for (unsigned int k = 0; k < total; ++k) {
    if ((host[k] >> 16) == (host[k & 7u] >> 16)) ++rr;      // same XCD as the first workgroup of its residue class (the XCD numbering itself is the hardware's)
    if (k + (unsigned)n_cus < total) {
      ++pairs;
      if (host[k] == host[k + n_cus]) ++same;
    }
    bool fresh = true;
    for (unsigned int v : seen) fresh = fresh && v != host[k];
    if (fresh) seen.push_back(host[k]);
  }
  int xcds = 0;
  for (unsigned int a = 0; a < 8 && a < total; ++a) {
    bool fresh = true;
    for (unsigned int b = 0; b < a; ++b) fresh = fresh && (host[a] >> 16) != (host[b] >> 16);
    xcds += fresh ? 1 : 0;
  }
  out[0] = (int)total; out[1] = pairs; out[2] = same; out[3] = rr; out[4] = n_cus; out[5] = (int)seen.size(); out[6] = (int)host[total]; out[7] = xcds;
  return KP1_OK;
}

#ifdef KP1_CLK_TRACE
int kp1_debug_clk_trace(unsigned long long* out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(kp1_clk_buf), sizeof(unsigned long long) * 2 * 1024 * 4));
  return KP1_OK;
}
#endif

#ifdef KP1_NT_TRACE
int kp1_debug_nt_trace(unsigned long long* out, int clear) {
  HIP_TRY(hipDeviceSynchronize());
  if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(kp1_nt_trace_buf), sizeof(unsigned long long) * KP1_TRACE_SLOTS * KP1_TRACE_WGS));
  if (clear) {
    std::vector<unsigned long long> z(KP1_TRACE_SLOTS * KP1_TRACE_WGS, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kp1_nt_trace_buf), z.data(), sizeof(unsigned long long) * z.size()));
  }
  return KP1_OK;
}
#endif

int kp1_mlp_adam_step(kp1_mlp* m, float* params, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float eps, float max_grad_norm,
                      int32_t step, int32_t zero_grad, void* stream_) {
  if (!m || !params || !grad || !exp_avg || !exp_avg_sq) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_adam_step");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = m->L.total;
  // zero_grad & 2: the caller did not touch grad since kp1_mlp_loss_grad (single GPU), so the sum-of-squares partials
  // the finalize kernel left are the norm of exactly this gradient and the extra reduction launch is skipped
  const bool fused_norm = (zero_grad & 2) && m->n_finalize_blocks > 0;
  const double* norm_partials = fused_norm ? m->partials + 2 * N_PARTIALS : m->partials + N_PARTIALS;
  const int n_norm_partials = fused_norm ? m->n_finalize_blocks : N_PARTIALS;
  ProfScope ps(m, KP1_MLP_PROFILE_ADAM, stream);
  if (!fused_norm) hipLaunchKernelGGL(sumsq_partials_kernel, dim3(N_PARTIALS), dim3(256), 0, stream, grad, n, m->partials + N_PARTIALS);
  zero_grad = 0;  // gradients are overwritten by the next finalize; nothing to clear
  // step <= 0: use the device-resident counter that kp1_mlp_loss_grad's finalize kernel increments (graph-replay safe)
  const int host_step = step > 0 ? step : 1;
  const float bc1 = 1.f - std::pow(0.9f, (float)host_step);
  const float bc2 = 1.f - std::pow(0.999f, (float)host_step);
  Packed kfmt = m->k;
  const bool frag_only = m->fused && m->Hp == FU_HP;
  kfmt.formats = frag_only ? PACK_FRAG : (PACK_SLAB | PACK_FRAG);
  m->last_params = params;
  if (frag_only) m->slab_stale = true;
  const bool vec4 = KP1_ADAM_VEC == 4 && ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                                          reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
  const int* step_arg = step > 0 ? (const int*)nullptr : (const int*)m->step_dev;
  if (vec4)
    KP1_LAUNCH(adam_kernel<4>, dim3((unsigned)((n + 4 * KP1_ADAM_BLOCK - 1) / (4 * KP1_ADAM_BLOCK))), dim3(KP1_ADAM_BLOCK), 0, stream, params, grad, exp_avg, exp_avg_sq, n, norm_partials,
                       n_norm_partials, lr, eps, max_grad_norm, bc1, std::sqrt(bc2), m->L, kfmt, zero_grad, step_arg, host_step, (const int*)m->step_dev + 1);
  else
    KP1_LAUNCH(adam_kernel<1>, dim3((unsigned)((n + KP1_ADAM_BLOCK - 1) / KP1_ADAM_BLOCK)), dim3(KP1_ADAM_BLOCK), 0, stream, params, grad, exp_avg, exp_avg_sq, n, norm_partials,
                       n_norm_partials, lr, eps, max_grad_norm, bc1, std::sqrt(bc2), m->L, kfmt, zero_grad, step_arg, host_step, (const int*)m->step_dev + 1);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

}  // extern "C"
