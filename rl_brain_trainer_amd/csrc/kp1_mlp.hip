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

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/kp1_ppo.h"
#include "kp1_host.hpp"

using kp1::fail;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int IN = KP1_MLP_IN, INP = KP1_MLP_IN_PAD, ACT = KP1_MLP_ACT, HEADS = 8;
constexpr int BK = 32, LDT = BK + 4;  // LDS row pitch 36 floats: ds_read_b128 conflict-free (guide: pad by one access width)
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;

enum { EPI_BIAS_TANH = 0, EPI_DTANH = 1 };

struct GemmNT {
  const float* A; int64_t lda; int64_t strideA;       // [M][lda], per-net stride
  const int64_t* gather;                               // optional row indices into A (shared by both nets)
  const float* W; int64_t strideW;                     // [N][K] row-major
  const float* bias; int64_t strideBias;               // [N]
  float* C; int64_t ldc; int64_t strideC;              // [M][ldc]
  const float* aux; int64_t strideAux;                 // EPI_DTANH: activation H at C's coordinates (ld = ldc)
  float* colsum; int64_t strideColsum;                 // EPI_DTANH: += column sums of C (bias gradient), may be null
  int M, N, K, Kreal;                                  // K multiple of 32; A columns >= Kreal read as 0
};

// C = epi(A W^T).  One workgroup owns 64 rows x BN columns (BN = 256: the whole hidden width, or 128 for small
// batches) for one net.  The 64 x K block of A is loaded ONCE and stays in LDS; W streams through a double-buffered
// [BN][32] stage, so every stage carries 4096 MFMA cycles per wave (BN = 256) against one L2 round trip, and A is
// never re-read.  4 waves, each 64 columns wide (2 MFMA column blocks) and RB row blocks tall.
template <int BN, int EPI>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const GemmNT g) {
  constexpr int BM = 64;
  constexpr int WN = BN / 64, WM = 4 / WN;
  constexpr int RB = BM / WM / 32, CB = 2;
  constexpr int W_LOADS = BN * 8 / 256;
  extern __shared__ float lds[];
  const int lda_s = g.K + 4;                 // LDS pitch of the resident A block (bank-conflict-free b128 reads)
  float* As = lds;
  float* Ws0 = lds + BM * lda_s;
  float* Ws1 = Ws0 + BN * LDT;

  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* __restrict__ A = g.A + z * g.strideA;
  const float* __restrict__ W = g.W + z * g.strideW + (int64_t)n0 * g.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;

  // ---- prologue: first W stage to registers, whole A block to LDS
  float4 rw[W_LOADS];
#pragma unroll
  for (int j = 0; j < W_LOADS; ++j) {
    const int f = tid + 256 * j;
    rw[j] = *reinterpret_cast<const float4*>(W + (int64_t)(f >> 3) * g.K + 4 * (f & 7));
  }
  {
    const int kq = g.K >> 2;                 // float4 per A row
    for (int f = tid; f < BM * kq; f += 256) {
      const int row = f / kq, c4 = f - row * kq;
      const int m = m0 + row, k = 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < g.M && k < g.Kreal) {
        const int64_t src = g.gather ? g.gather[m] : (int64_t)m;
        v = *reinterpret_cast<const float4*>(A + src * g.lda + k);
      }
      *reinterpret_cast<float4*>(As + row * lda_s + k) = v;
    }
  }
#pragma unroll
  for (int j = 0; j < W_LOADS; ++j) {
    const int f = tid + 256 * j;
    *reinterpret_cast<float4*>(Ws0 + (f >> 3) * LDT + 4 * (f & 7)) = rw[j];
  }
  __syncthreads();

  f32x16 acc[RB][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

  const int KT = g.K / BK;
  const float* as_base = As + (wr * (BM / WM) + (lane & 31)) * lda_s + 4 * (lane >> 5);
  for (int kt = 0; kt < KT; ++kt) {
    const float* ws_cur = (kt & 1) ? Ws1 : Ws0;
    float* ws_nxt = (kt & 1) ? Ws0 : Ws1;
    // unconditional prefetch (the last iteration re-reads its own stage): keeping the loads and the LDS stores out of
    // conditional blocks lets hipcc hold rw[] in registers; inside `if`s it spills the array to scratch and waits
    // for the loads before the MFMAs, which destroys the overlap
    const int k_next = (kt + 1 < KT ? kt + 1 : kt) * BK;
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) {
      const int f = tid + 256 * j;
      rw[j] = *reinterpret_cast<const float4*>(W + (int64_t)(f >> 3) * g.K + k_next + 4 * (f & 7));
    }
    const float* as = as_base + kt * BK;
    const float* ws = ws_cur + (wc * 64 + (lane & 31)) * LDT + 4 * (lane >> 5);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float4 a[RB], b[CB];
#pragma unroll
      for (int r = 0; r < RB; ++r) a[r] = *reinterpret_cast<const float4*>(as + r * 32 * lda_s + kg * 8);
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<const float4*>(ws + c * 32 * LDT + kg * 8);
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b[c].x, acc[r][c], 0, 0, 0);
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b[c].y, acc[r][c], 0, 0, 0);
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, b[c].z, acc[r][c], 0, 0, 0);
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, b[c].w, acc[r][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) {
      const int f = tid + 256 * j;
      *reinterpret_cast<float4*>(ws_nxt + (f >> 3) * LDT + 4 * (f & 7)) = rw[j];
    }
    __syncthreads();
  }

  // Epilogue in three passes so no memory operation is pending while tanhf's divergent blocks run: (1) issue every
  // load, (2) compute in registers, (3) issue every store.  (Interleaving them makes hipcc place s_waitcnt vmcnt(0)
  // in each conditional block, which serialises the stores: 32 dependent store round trips per lane.)
  float* __restrict__ C = g.C + z * g.strideC;
  const int mbase = m0 + wr * (BM / WM) + 4 * (lane >> 5);
  float bias[CB];
  float hval[EPI == EPI_DTANH ? CB : 1][RB][16];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int n = n0 + wc * 64 + c * 32 + (lane & 31);
    if constexpr (EPI == EPI_BIAS_TANH) bias[c] = g.bias[z * g.strideBias + n];
    else {
      bias[c] = 0.f;
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + r * 32 + (e & 3) + 8 * (e >> 2);
          hval[c][r][e] = m < g.M ? g.aux[z * g.strideAux + (int64_t)m * g.ldc + n] : 0.f;
        }
    }
  }
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    float csum = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[r][c][e];
        if constexpr (EPI == EPI_BIAS_TANH) {
          v = tanhf(v + bias[c]);
        } else {
          const float h = hval[c][r][e];
          v = v * (1.f - h * h);
          csum += v;  // rows >= M carry zero accumulators (their A rows were loaded as 0)
        }
        acc[r][c][e] = v;
      }
    if constexpr (EPI == EPI_DTANH) {
      if (g.colsum) {
        const int n = n0 + wc * 64 + c * 32 + (lane & 31);
        csum += __shfl_xor(csum, 32);
        if (lane < 32) atomicAdd(g.colsum + z * g.strideColsum + n, csum);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int n = n0 + wc * 64 + c * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mbase + r * 32 + (e & 3) + 8 * (e >> 2);
        if (m < g.M) C[(int64_t)m * g.ldc + n] = acc[r][c][e];
      }
  }
}

struct GemmTN {
  const float* D; int64_t ldd; int64_t strideD;     // dZ [B][ldd] : output-feature columns ("M" of the product)
  const float* X; int64_t ldx; int64_t strideX;     // previous activations [B][ldx] : input-feature columns ("N")
  const int64_t* gatherX;                            // optional row gather for X (layer 1 reads the obs buffer)
  float* G; int64_t ldg; int64_t strideG;           // dW [Mreal][ldg] += D^T X  (atomic accumulate)
  int B, Mreal, Nreal, Nload;                        // G rows o < Mreal, columns i < Nreal; X columns < Nload are readable
  int chunk;                                         // rows of B reduced per workgroup (multiple of 32)
  int n_i_tiles;
};

// dW[o][i] += sum_b D[b][o] X[b][i].  Block tile 64(o) x 64(i), 4 waves 2x2, one 32x32 MFMA tile per wave.
// Both operands are K(b)-major in LDS exactly as they sit in memory (rows of dZ / H), so lane l reads
// D[b = 2s + (l>>5)][o = wr*32 + (l&31)] and X[b][i = wc*32 + (l&31)]: conflict-free ds_read_b32, no transposes.
// Stages are 128 batch rows deep (4096 MFMA cycles per wave per stage, double buffered) so the next stage's loads
// are covered.  Small tiles + long batch chunks keep 256 workgroups busy while the split-K atomic traffic stays at
// n_chunks * H*H*4 B (4 MB at B = 8192, H = 256) instead of 16.8 MB with 128x128 tiles.
// grid: x = B chunk, y = o_tile * n_i_tiles + i_tile, z = net.
template <bool GATHER>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const GemmTN g) {
  constexpr int TB = 128, TT = 64, LDW = TT + 4, LOADS = TB * TT / 4 / 256;  // 8 float4 per thread per operand
  extern __shared__ float lds[];
  const int z = blockIdx.z;
  const int o0 = (blockIdx.y / g.n_i_tiles) * TT, i0 = (blockIdx.y % g.n_i_tiles) * TT;
  const int b_begin = blockIdx.x * g.chunk;
  const int b_end = min(b_begin + g.chunk, g.B);
  const float* __restrict__ D = g.D + z * g.strideD + o0;
  const float* __restrict__ X = g.X + z * g.strideX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int srow = tid >> 4, sc4 = tid & 15;  // staging: f = tid + 256*j -> row = srow + 16*j, c4 = sc4
  // columns of X past Nload are not readable (layer 1: 56 or 64 obs columns under a 64-wide tile): clamp + zero
  const bool x_col_ok = i0 + 4 * sc4 < g.Nload;
  const int x_col = x_col_ok ? i0 + 4 * sc4 : 0;
  const int b_last = b_end - 1;

  // Branch-free staging: out-of-range rows are clamped to the last valid row and zeroed after the load, so all 16
  // global loads of a stage are issued back to back (conditional blocks made hipcc wait vmcnt(0) between them).
  float4 rd[LOADS], rx[LOADS];
#define KP1_TN_LOAD(b0)                                                                                   \
  {                                                                                                       \
    int64_t xrow[LOADS];                                                                                  \
    _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                   \
      const int b = min((b0) + srow + 16 * j, b_last);                                                    \
      xrow[j] = GATHER ? g.gatherX[b] : (int64_t)b;                                                       \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                   \
      const int b = min((b0) + srow + 16 * j, b_last);                                                    \
      rd[j] = *reinterpret_cast<const float4*>(D + (int64_t)b * g.ldd + 4 * sc4);                         \
      rx[j] = *reinterpret_cast<const float4*>(X + xrow[j] * g.ldx + x_col);                              \
    }                                                                                                     \
  }
  // the zeroing of clamped rows happens here, at LDS-store time, so nothing touches the loaded registers (and no
  // vmcnt wait is needed) until the MFMAs of the current stage have been issued
#define KP1_TN_STORE(buf, b0)                                                                             \
  _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                     \
    const float keep = ((b0) + srow + 16 * j <= b_last) ? 1.f : 0.f;                                      \
    const float keepx = x_col_ok ? keep : 0.f;                                                            \
    float* base = lds + (buf) * 2 * TB * LDW + (srow + 16 * j) * LDW + 4 * sc4;                           \
    *reinterpret_cast<float4*>(base) = make_float4(rd[j].x * keep, rd[j].y * keep, rd[j].z * keep, rd[j].w * keep);          \
    *reinterpret_cast<float4*>(base + TB * LDW) = make_float4(rx[j].x * keepx, rx[j].y * keepx, rx[j].z * keepx, rx[j].w * keepx); \
  }

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  const int KT = (b_end - b_begin + TB - 1) / TB;
  if (KT <= 0) return;  // whole workgroup (uniform): nothing to reduce
  KP1_TN_LOAD(b_begin)
  KP1_TN_STORE(0, b_begin)
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    KP1_TN_LOAD(b_begin + (kt + 1) * TB)  // past the chunk end: clamped rows, zeroed
    const float* ds = lds + buf * 2 * TB * LDW + (lane >> 5) * LDW + wr * 32 + (lane & 31);
    const float* xs = ds + TB * LDW + (wc - wr) * 32;
#pragma unroll
    for (int s0 = 0; s0 < TB / 2; s0 += 8) {
      float av[8], xv[8];  // LDS reads of 8 steps in flight ahead of their MFMAs
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        av[u] = ds[2 * (s0 + u) * LDW];
        xv[u] = xs[2 * (s0 + u) * LDW];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], xv[u], acc, 0, 0, 0);
      // schedule: the 16 LDS reads of this group first, then its 8 MFMAs (otherwise hipcc waits lgkmcnt(0) per pair)
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    KP1_TN_STORE(buf ^ 1, b_begin + (kt + 1) * TB)
    __syncthreads();
  }
#undef KP1_TN_LOAD
#undef KP1_TN_STORE
  float* __restrict__ G = g.G + z * g.strideG;
  const int i = i0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int o = o0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    if (o < g.Mreal && i < g.Nreal) atomicAdd(G + (int64_t)o * g.ldg + i, acc[e]);
  }
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
  float* g_w3;            // [8][H] in SB3 layout: action_net.weight [7][H] then value_net.weight [1][H] (pointers below)
  float* g_action_w; float* g_action_b; float* g_value_w; float* g_value_b; float* g_log_std; float* g_b2p; float* g_b2v;
  int H;                  // real hidden (<= Hp)
  float* stats;           // [4]
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
__global__ void __launch_bounds__(256) head_train_kernel(const HeadArgs a) {
  // all LDS is one dynamic array (16-B aligned base: head_dot reads w3s as float4; guide G17)
  extern __shared__ float smem[];
  float* w3s = smem;                       // [8][Hp]
  float* dout = smem + HEADS * a.Hp;       // [HEAD_ROWS][8] : d loss / d (mean_0..6, value)
  float* red = dout + HEAD_ROWS * 8;       // [4]
  float* adv_ms = red + 4;                 // [2]
  for (int k = threadIdx.x; k < HEADS * a.Hp; k += 256) w3s[k] = a.w3[k];
  if (threadIdx.x < 4) red[threadIdx.x] = 0.f;
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
  if (ok) {
    const float* h = a.h2 + (out == 7 ? a.strideH : 0) + (int64_t)row * a.Hp;
    v = head_dot(h, w3s + out * a.Hp, a.Hp) + a.b3[out];
  }
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
  // d loss / d log_std_out = sum_rows g_logp (z^2 - 1)  (+ entropy term added once by block 0)
  float gls = (ok && out < ACT) ? g_logp * (z * z - 1.f) : 0.f;
  // reduce over the 32 rows that share `out`: lanes with equal (t & 7) inside a wave, then across the 4 waves via atomics
  gls += __shfl_xor(gls, 8);
  gls += __shfl_xor(gls, 16);
  gls += __shfl_xor(gls, 32);
  float dsum = ok ? d : 0.f;
  dsum += __shfl_xor(dsum, 8);
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);
  if ((threadIdx.x & 63) < 8) {
    if (out < ACT) {
      atomicAdd(a.g_log_std + out, gls);
      atomicAdd(a.g_action_b + out, dsum);
    } else {
      atomicAdd(a.g_value_b, dsum);
    }
  }
  if (out == 7) {
    atomicAdd(&red[0], pl);
    atomicAdd(&red[1], vl);
    atomicAdd(&red[3], kl);
  }
  __syncthreads();
  if (threadIdx.x == 0 && a.stats) {
    atomicAdd(a.stats + 0, red[0] * a.inv_count);
    atomicAdd(a.stats + 1, red[1] * a.inv_count);
    atomicAdd(a.stats + 3, red[3] * a.inv_count);
  }
  if (blockIdx.x == 0 && threadIdx.x < ACT) {
    atomicAdd(a.g_log_std + threadIdx.x, -a.ent_coef);  // d(-ent_coef * sum_a log_std_a)/d log_std
    if (threadIdx.x == 0 && a.stats) {
      float ent = 0.f;
      for (int k = 0; k < ACT; ++k) ent += 0.5f + LOG_SQRT_2PI + a.log_std[k];
      atomicAdd(a.stats + 2, ent);
    }
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
      const float hp = a.h2[rr], hv = a.h2[a.strideH + rr];
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
    if (h < a.H) {
#pragma unroll
      for (int o = 0; o < ACT; ++o) atomicAdd(a.g_action_w + (int64_t)o * a.H + h, gw[o]);
      atomicAdd(a.g_value_w + h, gw[7]);
      atomicAdd(a.g_b2p + h, gb2p);
      atomicAdd(a.g_b2v + h, gb2v);
    }
  }
}

// ---------------------------------------------------------------------------------------------- optimiser
struct ParamLayout {  // offsets into the flat SB3-order vector
  int H, Hp;
  int64_t log_std, p_w1, p_b1, p_w2, p_b2, v_w1, v_b1, v_w2, v_b2, a_w, a_b, c_w, c_b, total;
};

__host__ __device__ inline ParamLayout make_layout(int H) {
  ParamLayout L;
  L.H = H;
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

struct Packed {
  float *w1p, *b1, *w2, *w2t, *b2, *w3, *b3, *log_std;  // [2][Hp][64], [2][Hp], [2][Hp][Hp], [2][Hp][Hp], [2][Hp], [8][Hp], [8], [8]
};

// flat SB3 vector -> kernel-format weights (zero padding pre-set once at creation)
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ p, const ParamLayout L, const Packed k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L.total) return;
  const float v = p[i];
  const int H = L.H, Hp = L.Hp;
  if (i < L.p_w1) {
    k.log_std[i] = v;
  } else if (i < L.p_b1) {
    const int64_t e = i - L.p_w1;
    k.w1p[(e / IN) * INP + e % IN] = v;
  } else if (i < L.p_w2) {
    k.b1[i - L.p_b1] = v;
  } else if (i < L.p_b2) {
    const int64_t e = i - L.p_w2, r = e / H, c = e % H;
    k.w2[r * Hp + c] = v;
    k.w2t[c * Hp + r] = v;
  } else if (i < L.v_w1) {
    k.b2[i - L.p_b2] = v;
  } else if (i < L.v_b1) {
    const int64_t e = i - L.v_w1;
    k.w1p[(int64_t)Hp * INP + (e / IN) * INP + e % IN] = v;
  } else if (i < L.v_w2) {
    k.b1[Hp + i - L.v_b1] = v;
  } else if (i < L.v_b2) {
    const int64_t e = i - L.v_w2, r = e / H, c = e % H;
    k.w2[(int64_t)Hp * Hp + r * Hp + c] = v;
    k.w2t[(int64_t)Hp * Hp + c * Hp + r] = v;
  } else if (i < L.a_w) {
    k.b2[Hp + i - L.v_b2] = v;
  } else if (i < L.a_b) {
    const int64_t e = i - L.a_w;
    k.w3[(e / H) * Hp + e % H] = v;
  } else if (i < L.c_w) {
    k.b3[i - L.a_b] = v;
  } else if (i < L.c_b) {
    k.w3[(int64_t)7 * Hp + (i - L.c_w)] = v;
  } else {
    k.b3[7] = v;
  }
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
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, const double* __restrict__ partials, int n_partials, float lr, float eps, float max_norm,
                                                   float bc1, float bc2_sqrt) {
  __shared__ float scale_s;
  double s = 0.0;
  if (threadIdx.x < 64) {
    for (int k = threadIdx.x; k < n_partials; k += 64) s += partials[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s);
    scale_s = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * scale_s;
  const float mi = 0.9f * m[i] + 0.1f * gi;
  const float vi = 0.999f * v[i] + 0.001f * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

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
  double* partials = nullptr;  // [512]
  std::vector<void*> allocs;
};

namespace {

constexpr int N_PARTIALS = 128;

// rows of the batch each TN workgroup reduces: aim at ~256 workgroups (one per CU) for the H x H gradient
int tn_chunk_rows(int n, int Hp) {
  const int tiles = (Hp / 64) * (Hp / 64) * 2;
  int chunks = (256 + tiles - 1) / tiles;
  if (chunks < 1) chunks = 1;
  int rows = (n + chunks - 1) / chunks;
  rows = (rows + 127) / 128 * 128;  // whole 128-row stages
  return rows < 128 ? 128 : rows;
}

int mlp_check_device(const kp1_mlp* m) {
  HIP_TRY(hipSetDevice(m->device));
  return KP1_OK;
}

constexpr size_t TN_LDS_BYTES = sizeof(float) * 2 * 2 * 128 * (64 + 4);

template <int EPI>
int launch_nt(const GemmNT& g, hipStream_t stream) {
  // 256-column workgroups (A read once per row block) when that still fills the chip, else 128-column ones
  const int row_tiles = (g.M + 63) / 64;
  const bool wide = (g.N % 256 == 0) && row_tiles * 2 >= 200;
  const size_t lds_a = sizeof(float) * 64 * (size_t)(g.K + 4);
  if (wide) {
    const size_t bytes = lds_a + sizeof(float) * 2 * 256 * LDT;
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<256, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((gemm_nt_kernel<256, EPI>), dim3(row_tiles, g.N / 256, 2), dim3(256), bytes, stream, g);
  } else {
    const size_t bytes = lds_a + sizeof(float) * 2 * 128 * LDT;
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<128, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((gemm_nt_kernel<128, EPI>), dim3(row_tiles, g.N / 128, 2), dim3(256), bytes, stream, g);
  }
  return KP1_OK;
}

int launch_tn(const GemmTN& t, int n_o_tiles, hipStream_t stream) {
  const dim3 grid((t.B + t.chunk - 1) / t.chunk, n_o_tiles * t.n_i_tiles, 2);
  if (t.gatherX) {
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES));
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), TN_LDS_BYTES, stream, t);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES));
    hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), TN_LDS_BYTES, stream, t);
  }
  return KP1_OK;
}

// forward layers 1 and 2 for n rows (both nets): h1, h2 filled
int launch_forward_layers(kp1_mlp* m, const float* obs, int obs_stride, const int64_t* idx, int n, hipStream_t stream) {
  const int Hp = m->Hp;
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
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

}  // namespace

extern "C" {

int64_t kp1_mlp_num_params(int32_t hidden) { return hidden > 0 ? make_layout(hidden).total : 0; }

int kp1_mlp_create(int32_t device, int32_t hidden, int32_t max_batch, kp1_mlp** out) {
  if (!out || hidden <= 0 || hidden > 1024 || max_batch <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_create");
  if (hidden % 128 != 0) return fail(KP1_ERR_UNSUPPORTED, "hidden must be a multiple of 128 (the 128-wide MFMA column tile); 2x256 is BASELINE config 2");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= count) return fail(KP1_ERR_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  kp1_mlp* m = new kp1_mlp();
  m->device = device;
  m->H = hidden;
  m->L = make_layout(hidden);
  m->Hp = m->L.Hp;
  m->max_batch = (max_batch + 127) / 128 * 128;
  const int64_t Hp = m->Hp, mb = m->max_batch;
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
  MLP_ALLOC(m->k.b2, 2 * Hp);
  MLP_ALLOC(m->k.w3, HEADS * Hp);
  MLP_ALLOC(m->k.b3, HEADS);
  MLP_ALLOC(m->k.log_std, HEADS);
  MLP_ALLOC(m->h1, 2 * mb * Hp);
  MLP_ALLOC(m->h2, 2 * mb * Hp);
  MLP_ALLOC(m->dz2, 2 * mb * Hp);
  MLP_ALLOC(m->dz1, 2 * mb * Hp);
  MLP_ALLOC(m->partials, 2 * N_PARTIALS);
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
  delete m;
  return KP1_OK;
}

int kp1_mlp_pack_weights(kp1_mlp* m, const float* params, void* stream) {
  if (!m || !params) return fail(KP1_ERR_INVALID, "NULL argument");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((m->L.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, m->L, m->k);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

int kp1_mlp_forward(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, const float* noise, float* mean, float* value, float* action,
                    float* clipped_action, float* log_prob, void* stream) {
  if (!m || !obs) return fail(KP1_ERR_INVALID, "NULL argument");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  if (obs_stride != IN && obs_stride != INP) return fail(KP1_ERR_INVALID, "obs_stride must be 56 or 64");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  rc = launch_forward_layers(m, obs, obs_stride, nullptr, n, (hipStream_t)stream);
  if (rc != KP1_OK) return rc;
  HeadArgs a{};
  a.h2 = m->h2; a.strideH = (int64_t)m->max_batch * m->Hp; a.Hp = m->Hp; a.n = n; a.H = m->H;
  a.w3 = m->k.w3; a.b3 = m->k.b3; a.log_std = m->k.log_std;
  a.noise = noise; a.mean = mean; a.value = value; a.action = action; a.clipped = clipped_action; a.log_prob = log_prob;
  hipLaunchKernelGGL(head_infer_kernel, dim3((n + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), sizeof(float) * HEADS * m->Hp, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

int kp1_mlp_loss_grad(kp1_mlp* m, const float* obs, int32_t obs_stride, const int64_t* idx, int32_t n, const float* actions,
                      const float* old_log_prob, const float* advantages, const float* returns, float adv_mean, float adv_inv_std,
                      const float* adv_stats_dev, float clip_range, float ent_coef, float vf_coef, float inv_count, float* grad_out,
                      float* stats_out, void* stream_) {
  if (!m || !obs || !actions || !old_log_prob || !advantages || !returns || !grad_out) return fail(KP1_ERR_INVALID, "NULL argument");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  if (obs_stride != IN && obs_stride != INP) return fail(KP1_ERR_INVALID, "obs_stride must be 56 or 64");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int Hp = m->Hp, H = m->H;
  const int64_t act_stride = (int64_t)m->max_batch * Hp;
  const ParamLayout& L = m->L;
  HIP_TRY(hipMemsetAsync(grad_out, 0, sizeof(float) * (size_t)L.total, stream));
  rc = launch_forward_layers(m, obs, obs_stride, idx, n, stream);
  if (rc != KP1_OK) return rc;

  HeadArgs a{};
  a.h2 = m->h2; a.strideH = act_stride; a.Hp = Hp; a.n = n; a.H = H;
  a.w3 = m->k.w3; a.b3 = m->k.b3; a.log_std = m->k.log_std;
  a.idx = idx; a.actions = actions; a.old_logp = old_log_prob; a.adv = advantages; a.ret = returns;
  a.adv_partials = m->partials; a.n_adv_partials = N_PARTIALS; a.adv_stats = adv_stats_dev; a.adv_mean = adv_mean; a.adv_inv_std = adv_inv_std;
  if (adv_stats_dev) a.adv_mode = 2;
  else if (adv_inv_std > 0.f) a.adv_mode = 3;
  else if (adv_inv_std == 0.f) a.adv_mode = 1;
  else a.adv_mode = 0;
  if (a.adv_mode == 1) hipLaunchKernelGGL(adv_partials_kernel, dim3(N_PARTIALS), dim3(256), 0, stream, advantages, idx, n, m->partials);
  a.clip_range = clip_range; a.ent_coef = ent_coef; a.vf_coef = vf_coef; a.inv_count = inv_count;
  a.dz2 = m->dz2;
  a.g_action_w = grad_out + L.a_w; a.g_action_b = grad_out + L.a_b; a.g_value_w = grad_out + L.c_w; a.g_value_b = grad_out + L.c_b;
  a.g_log_std = grad_out + L.log_std; a.g_b2p = grad_out + L.p_b2; a.g_b2v = grad_out + L.v_b2;
  a.stats = stats_out;
  hipLaunchKernelGGL(head_train_kernel, dim3((n + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), sizeof(float) * (HEADS * Hp + HEAD_ROWS * 8 + 8), stream, a);

  // dZ1 = (dZ2 W2) * (1 - h1^2), bias-1 gradient = column sums of dZ1 (into a padded scratch, copied below by the TN stage)
  GemmNT g{};
  g.A = m->dz2; g.lda = Hp; g.strideA = act_stride; g.gather = nullptr;
  g.W = m->k.w2t; g.strideW = (int64_t)Hp * Hp;
  g.bias = nullptr; g.strideBias = 0;
  g.C = m->dz1; g.ldc = Hp; g.strideC = act_stride;
  g.aux = m->h1; g.strideAux = act_stride;
  g.colsum = nullptr; g.strideColsum = 0;
  g.M = n; g.N = Hp; g.K = Hp; g.Kreal = Hp;
  // b1 gradients live at different flat offsets for the two nets; colsum stride expresses that when H == Hp
  if (H == Hp) {
    g.colsum = grad_out + L.p_b1;
    g.strideColsum = L.v_b1 - L.p_b1;
  }
  rc = launch_nt<EPI_DTANH>(g, stream);
  if (rc != KP1_OK) return rc;

  // weight gradients (split over the batch axis)
  GemmTN t{};
  t.B = n;
  t.chunk = tn_chunk_rows(n, Hp);
  // dW2[o][i] = sum_b dZ2[b][o] h1[b][i]
  t.D = m->dz2; t.ldd = Hp; t.strideD = act_stride;
  t.X = m->h1; t.ldx = Hp; t.strideX = act_stride; t.gatherX = nullptr;
  t.G = grad_out + L.p_w2; t.ldg = H; t.strideG = L.v_w2 - L.p_w2;
  t.Mreal = H; t.Nreal = H; t.Nload = Hp; t.n_i_tiles = Hp / 64;
  rc = launch_tn(t, Hp / 64, stream);
  if (rc != KP1_OK) return rc;
  // dW1[o][i] = sum_b dZ1[b][o] x[b][i]   (i < 56)
  t.D = m->dz1;
  t.X = obs; t.ldx = obs_stride; t.strideX = 0; t.gatherX = idx;
  t.G = grad_out + L.p_w1; t.ldg = IN; t.strideG = L.v_w1 - L.p_w1;
  t.Nreal = IN; t.Nload = obs_stride >= INP ? INP : IN; t.n_i_tiles = 1;
  rc = launch_tn(t, Hp / 64, stream);
  if (rc != KP1_OK) return rc;
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

int kp1_mlp_time_kernels(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, int32_t iters, float* out_ms, double* out_flops,
                         void* stream_) {
  if (!m || !obs || !out_ms || !out_flops || iters <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_time_kernels");
  if (n <= 0 || n > m->max_batch) return fail(KP1_ERR_INVALID, "n exceeds the workspace max_batch");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int Hp = m->Hp;
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
        t.B = n; t.chunk = tn_chunk_rows(n, Hp);
        t.D = m->dz2; t.ldd = Hp; t.strideD = act_stride; t.X = m->h1; t.ldx = Hp; t.strideX = act_stride;
        t.G = m->dz1; t.ldg = Hp; t.strideG = act_stride;  // scratch target (dz1 is rewritten by kernel 1 anyway)
        t.Mreal = Hp; t.Nreal = Hp; t.Nload = Hp; t.n_i_tiles = Hp / 64;
        if (launch_tn(t, Hp / 64, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
      } else {
        g.A = obs; g.lda = obs_stride; g.strideA = 0; g.W = m->k.w1p; g.strideW = (int64_t)Hp * INP; g.bias = m->k.b1; g.strideBias = Hp;
        g.C = m->h1; g.K = INP; g.Kreal = obs_stride >= INP ? INP : IN;
        if (launch_nt<EPI_BIAS_TANH>(g, stream) != KP1_OK) return KP1_ERR_NO_DEVICE;
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
  const double M = n, H = Hp;
  out_flops[0] = 2.0 * 2.0 * M * H * H;
  out_flops[1] = 2.0 * 2.0 * M * H * H;
  out_flops[2] = 2.0 * 2.0 * M * H * H;
  out_flops[3] = 2.0 * 2.0 * M * H * INP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return KP1_OK;
}

int kp1_mlp_adam_step(kp1_mlp* m, float* params, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float eps, float max_grad_norm,
                      int32_t step, void* stream_) {
  if (!m || !params || !grad || !exp_avg || !exp_avg_sq || step <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_mlp_adam_step");
  int rc = mlp_check_device(m);
  if (rc != KP1_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = m->L.total;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(N_PARTIALS), dim3(256), 0, stream, grad, n, m->partials + N_PARTIALS);
  const float bc1 = 1.f - std::pow(0.9f, (float)step);
  const float bc2 = 1.f - std::pow(0.999f, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, params, grad, exp_avg, exp_avg_sq, n,
                     m->partials + N_PARTIALS, N_PARTIALS, lr, eps, max_grad_norm, bc1, std::sqrt(bc2));
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, params, m->L, m->k);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

}  // extern "C"
