// kp1_device.hpp -- device-side env arithmetic of the MI355X kinematic_phase1 engine (gfx950).
//
// One wavefront lane owns one environment.  Everything a lane needs per step lives in
// registers; configuration is wave-uniform and comes in through scalar loads (SGPR operands)
// from a read-only DevCfg block, per-env state is SoA [field][env] in HBM so each field access
// is one fully coalesced wave instruction.
//
// Real type R: float (production; north_star tolerance 1e-5 on pose error) or double (strict
// parity build, used by the tests to check counters/flags bit-exactly against the oracle).
// Reset sampling is always fp64 so the PCG64 draw sequence, the stage-index choice and the
// sampled joint vectors are bit-identical to the reference; they are rounded to R when stored.
//
// Reference semantics cited per function (paths relative to hrl_trainer/):
//   KP1/ = kinematic_phase1/ ,  V51/ = v5_1/
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kp1.h"

namespace kp1 {

constexpr int NJ = KP1_NJ;

// ---- math overloads -----------------------------------------------------------------------
__device__ __forceinline__ float kp_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double kp_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ void kp_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void kp_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ float kp_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double kp_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float kp_fmod(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double kp_fmod(double a, double b) { return fmod(a, b); }
__device__ __forceinline__ float kp_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double kp_pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ bool kp_isfinite(float a) { return isfinite(a); }
__device__ __forceinline__ bool kp_isfinite(double a) { return isfinite(a); }
__device__ __forceinline__ float kp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double kp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Quotients of the R-typed arithmetic downstream of pose6 (normalisers, margins, reward ratios).  fp64 handle: IEEE division, as the
// reference.  fp32 handle: v_rcp_f32 (1 ulp) and a multiply -- 2 instructions instead of the 10 of the correctly rounded fp32 division
// sequence (48 of them on the step path = a tenth of its instructions); the result differs from the rounded quotient by <= 2 ulp, far
// inside the fp32 handle's tolerances (observations 2e-6, rewards 1e-5 relative).
__device__ __forceinline__ float kp_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double kp_div(double a, double b) { return a / b; }
template <typename R> __device__ __forceinline__ R kp_max(R a, R b) { return a > b ? a : b; }
template <typename R> __device__ __forceinline__ R kp_min(R a, R b) { return a < b ? a : b; }
template <typename R> __device__ __forceinline__ R kp_clip(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ int kp_clipi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ int kp_mini(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int kp_maxi(int a, int b) { return a > b ? a : b; }

// ---- device config ------------------------------------------------------------------------
#define KP1_DEV_DECL_f64(name) R name;
#define KP1_DEV_DECL_i32(name) int name;
#define KP1_DEV_DECL(type, name, dflt) KP1_DEV_DECL_##type(name)

// FK chain constants folded on the host in fp64 (V51/ee_fk.py:14-61,98-117):
//   joint 0 is prismatic, joints 1..6 revolute.  With A_i = [RA_i | p_i] the constant origin
//   transform and a_i the normalised local axis,  RA_i * Rodrigues(a_i, q) = K1_i + cos(q) Kc_i + sin(q) Ks_i
//   with K1 = RA a a^T, Kc = RA (I - a a^T), Ks = RA [a]x.
template <typename R>
struct DevFk {
  R p01[3];      // p_0 + RA_0 p_1  (position before joint-1 rotation, at q0 = 0)
  R v0[3];       // RA_0 a_0        (prismatic direction)
  R k1[6][9];    // joints 1..6 (joint 1's matrices pre-multiplied by RA_0)
  R kc[6][9];
  R ks[6][9];
  R p[5][3];     // p_2..p_6
};

template <typename R>
struct DevCfg {
  struct Env { KP1_ENV_FIELDS(KP1_DEV_DECL) } env;
  struct Reward {
    KP1_APPROACH_REWARD_FIELDS(KP1_DEV_DECL)
    R ms_thr[KP1_MAX_MILESTONES];
    R ms_bonus[KP1_MAX_MILESTONES];
  } reward;
  struct DockReward { KP1_DOCK_REWARD_FIELDS(KP1_DEV_DECL) } dock;
  struct Term { KP1_TERMINATION_FIELDS(KP1_DEV_DECL) } term;
  struct Obs { KP1_OBSERVATION_FIELDS(KP1_DEV_DECL) } obs;
  R lower[NJ], upper[NJ], dlim[NJ];
  // The kinematic chain -- q integration (q += clip(action) * delta_limit * scale, joint clip), the FK products and the Euler-angle
  // extraction -- runs in fp64 on BOTH handles (V51/ee_fk.py:98-134 and arm_kinematic_env.py:237-246 are fp64 in the reference).
  // Roll and yaw are atan2 of rotation entries of size cos(pitch): an absolute error e in those entries, or in q, becomes e / cos(pitch)
  // in the angles, so fp32 there (1e-7) cannot hold the 1e-5 pose bar on workspaces that reach pitch -> +-pi/2 (BASELINE configs[3]).
  // Everything downstream of pose6 (errors, counters, reward, observation) stays in R.
  struct Kin {
    double lower[NJ], upper[NJ], dlim[NJ];
    double action_delta_scale, dock_action_delta_scale;
    DevFk<double> fk;
  } kin;
};

// A wave-uniform, read-only block (DevCfg, DevSampler): route its loads through the constant address space, so that they are scalar loads
// (SGPR operands, no vector registers, no vector-memory round trip) whatever the stores around them might alias.  Without this the
// compiler falls back to uniform-address VECTOR loads for every config read it cannot prove unclobbered by the state stores.
template <typename T>
__device__ __forceinline__ const T& uniform_block(const T* p) {
  typedef const __attribute__((address_space(4))) T* const_space_ptr;
  return *(const T*)(const_space_ptr)p;
}

// The same block seen through a per-lane pointer (block + a zero the compiler cannot see through): its reads become VECTOR loads of a
// wave-uniform address -- one cache line broadcast to all lanes -- with the SGPR-base + immediate-offset addressing form.  Why one would
// want that for uniform data: scalar loads return out of order, so every use waits for ALL outstanding ones (s_waitcnt lgkmcnt(0)), and
// 102 SGPRs hold only a few batches ahead; vector loads return in order (vmcnt(N) waits for exactly the one needed) and a wave that is
// alone on its SIMD has 512 vector registers to prefetch into.
template <typename T>
__device__ __forceinline__ const T& lane_view(const T& block) {
  uint32_t zero;
  asm("v_mov_b32 %0, 0" : "=v"(zero));
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(&block) + zero);
}

// Warm the scalar data cache with a block the wave is about to read through scalar loads.  The step kernel reads ~400 scalars of DevCfg, and
// the compiler (102 SGPRs) can only load them a few at a time next to their uses: each batch is then a dependent round trip to L2
// (~200-500 cycles), and with ONE wave per SIMD nothing else runs meanwhile -- SQ_WAIT_ANY was 53 % of the wave's cycles
// (profiles/r03_sq_counters.json).  One dword from every 64-byte line, all in flight at once, turns those round trips into scalar-cache
// hits for the price of a single one.  Returns the block pointer made dependent on the loaded words (plus a run-time zero), so that every
// later read of the block is ordered behind the warm-up without a volatile asm (a volatile asm counts as a store to everything for the
// compiler, which then reads the whole block with uniform-address VECTOR loads).
#ifndef KP1_CFG_WARM
#define KP1_CFG_WARM 2
#endif
#define KP1_WARM_BYTES 4096   // blocks handed to scalar_cache_warm are allocated in multiples of this
template <typename T>
__device__ __forceinline__ const T* scalar_cache_warm(const T* __restrict__ block) {
#if KP1_CFG_WARM == 2
  // ONE asm statement: 64 loads (4 KB, the block's allocation is padded to that) into the same scratch SGPR -- the words are never used --,
  // one wait, and the scratch register forced to zero as the statement's only output
  static_assert(KP1_WARM_BYTES == 4096, "the load list below covers 4 KB");
  uint32_t zero;
  asm(
      "s_load_dword %0, %1, 0x0\n\t"
      "s_load_dword %0, %1, 0x40\n\t"
      "s_load_dword %0, %1, 0x80\n\t"
      "s_load_dword %0, %1, 0xc0\n\t"
      "s_load_dword %0, %1, 0x100\n\t"
      "s_load_dword %0, %1, 0x140\n\t"
      "s_load_dword %0, %1, 0x180\n\t"
      "s_load_dword %0, %1, 0x1c0\n\t"
      "s_load_dword %0, %1, 0x200\n\t"
      "s_load_dword %0, %1, 0x240\n\t"
      "s_load_dword %0, %1, 0x280\n\t"
      "s_load_dword %0, %1, 0x2c0\n\t"
      "s_load_dword %0, %1, 0x300\n\t"
      "s_load_dword %0, %1, 0x340\n\t"
      "s_load_dword %0, %1, 0x380\n\t"
      "s_load_dword %0, %1, 0x3c0\n\t"
      "s_load_dword %0, %1, 0x400\n\t"
      "s_load_dword %0, %1, 0x440\n\t"
      "s_load_dword %0, %1, 0x480\n\t"
      "s_load_dword %0, %1, 0x4c0\n\t"
      "s_load_dword %0, %1, 0x500\n\t"
      "s_load_dword %0, %1, 0x540\n\t"
      "s_load_dword %0, %1, 0x580\n\t"
      "s_load_dword %0, %1, 0x5c0\n\t"
      "s_load_dword %0, %1, 0x600\n\t"
      "s_load_dword %0, %1, 0x640\n\t"
      "s_load_dword %0, %1, 0x680\n\t"
      "s_load_dword %0, %1, 0x6c0\n\t"
      "s_load_dword %0, %1, 0x700\n\t"
      "s_load_dword %0, %1, 0x740\n\t"
      "s_load_dword %0, %1, 0x780\n\t"
      "s_load_dword %0, %1, 0x7c0\n\t"
      "s_load_dword %0, %1, 0x800\n\t"
      "s_load_dword %0, %1, 0x840\n\t"
      "s_load_dword %0, %1, 0x880\n\t"
      "s_load_dword %0, %1, 0x8c0\n\t"
      "s_load_dword %0, %1, 0x900\n\t"
      "s_load_dword %0, %1, 0x940\n\t"
      "s_load_dword %0, %1, 0x980\n\t"
      "s_load_dword %0, %1, 0x9c0\n\t"
      "s_load_dword %0, %1, 0xa00\n\t"
      "s_load_dword %0, %1, 0xa40\n\t"
      "s_load_dword %0, %1, 0xa80\n\t"
      "s_load_dword %0, %1, 0xac0\n\t"
      "s_load_dword %0, %1, 0xb00\n\t"
      "s_load_dword %0, %1, 0xb40\n\t"
      "s_load_dword %0, %1, 0xb80\n\t"
      "s_load_dword %0, %1, 0xbc0\n\t"
      "s_load_dword %0, %1, 0xc00\n\t"
      "s_load_dword %0, %1, 0xc40\n\t"
      "s_load_dword %0, %1, 0xc80\n\t"
      "s_load_dword %0, %1, 0xcc0\n\t"
      "s_load_dword %0, %1, 0xd00\n\t"
      "s_load_dword %0, %1, 0xd40\n\t"
      "s_load_dword %0, %1, 0xd80\n\t"
      "s_load_dword %0, %1, 0xdc0\n\t"
      "s_load_dword %0, %1, 0xe00\n\t"
      "s_load_dword %0, %1, 0xe40\n\t"
      "s_load_dword %0, %1, 0xe80\n\t"
      "s_load_dword %0, %1, 0xec0\n\t"
      "s_load_dword %0, %1, 0xf00\n\t"
      "s_load_dword %0, %1, 0xf40\n\t"
      "s_load_dword %0, %1, 0xf80\n\t"
      "s_load_dword %0, %1, 0xfc0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, 0"
      : "=&s"(zero) : "s"(block));
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(block) + zero);
#else
  return block;
#endif
}

// sampler configuration, always fp64 (reset path only)
struct DevSampler {
  double lower[NJ], upper[NJ];
  int curriculum_enabled, n_stages;
  kp1_stage stages[KP1_MAX_STAGES];
  kp1_stage_sampling ss;
  kp1_random_start rs;
  kp1_dock_reset dr;
  double start_sample_margin_fraction, goal_sample_margin_fraction;
  int n_handoff;        // states a dock reset may draw from: handoff[handoff_offset .. handoff_offset + n_handoff)
  int handoff_offset;   // 0 unless the dock reverse curriculum selected a stage-specific slice of the buffer
  DevFk<double> fk;  // goal_q -> goal_pose6 is part of the sampled state; close-bucket rejection needs fp64 FK
};

// ---- per-env state layout (SoA [field][N]) -------------------------------------------------
enum RealField {
  F_Q = 0, F_DQ = 7, F_PREV_ACTION = 14, F_GOAL_Q = 21, F_GOAL_POSE = 28, F_EE_POSE = 34, F_ENTRY = 40,
  F_MIN_POS = 44, F_POS_ERR = 45, F_ORI_ERR = 46, F_EXEC_DQ = 47, F_ACTION_L2 = 48, F_DQ_CHANGE = 49,
  F_QLO = 50,    // fp32 handle: q = (double)F_Q + (double)F_QLO (F_Q is q rounded to fp32, what every fp32 consumer reads); unused in fp64
  F_NUM_REAL = 57
};
enum IntField { I_STEP = 0, I_DWELL, I_ENTRY, I_DRIFT, I_FLAGS, I_STAGE, I_NUM_INT };
enum { FLAG_PRE_NEAR_HIT = 1, FLAG_NEAR_HIT = 2, FLAG_SUCCESS = 4 };

template <typename R>
struct EnvState {
  R* real;         // [F_NUM_REAL][N]
  int32_t* ints;   // [I_NUM_INT][N]
  uint64_t* rng64; // [4][N]: state_hi, state_lo, inc_hi, inc_lo
  uint32_t* rng32; // [2][N]: has_uint32, uinteger
  int64_t n;
  // Address of element i of field plane f = ONE wave-uniform base (the array itself, an SGPR pair) + a 32-bit byte offset per lane,
  // (f * n + i) * sizeof(T): the loads and stores take the SGPR-base + 32-bit-offset form and the offset costs one 32-bit multiply-add,
  // instead of a 64-bit multiply-add chain per field per lane (it was a tenth of the step's vector instructions).  A separate 64-bit
  // scalar base per PLANE would be cheaper still per access, but 57 of them do not fit the 102 SGPRs: the compiler spills them to
  // vector-register lanes, and in the largest kernel (kp1_route_step_kernel<double>: 512 registers + scratch) those lanes were themselves
  // spilled to scratch inside the divergent auto-reset region and came back wrong -- a wild scalar base, HSA_STATUS_ERROR_MEMORY_APERTURE_
  // VIOLATION at the first in-launch route reset (round 3, gpurun_out/r03_dbg1.log).  kp1_create bounds n so that the offset fits 32 bits.
  template <typename T>
  static __device__ __forceinline__ T& at(T* base, int f, int64_t n, int64_t i) {
#ifdef KP1_OLD_PLANE_ADDR
    return base[(int64_t)f * n + i];
#else
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + ((uint32_t)f * (uint32_t)n + (uint32_t)i) * (uint32_t)sizeof(T));
#endif
  }
  __device__ __forceinline__ R& r(int f, int64_t i) const { return at<R>(real, f, n, i); }
#ifndef KP1_ENV_LD_AUX
#define KP1_ENV_LD_AUX 16  // cache-policy bits of the step's state loads (fp32 handle; 0 = plain loads): sc1, not allocated in the CU's L1: 7.9 -> 7.7 us at 4096 envs, 10.09 -> 9.89 us at 32768 (profiles/r03_ab_env_load_policy.log)
#endif
  // state loads of the step path: every field is read once per step, nothing is reused through the CU's L1
  __device__ __forceinline__ R rl(int f, int64_t i) const {
    if constexpr (KP1_ENV_LD_AUX == 0 || sizeof(R) != 4) {
      return at<R>(real, f, n, i);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(real, 0, 0x7fffffff, 0x00020000);
      return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(((uint32_t)f * (uint32_t)n + (uint32_t)i) * 4u), 0, KP1_ENV_LD_AUX));
    }
  }
  __device__ __forceinline__ int32_t ivl(int f, int64_t i) const {
    if constexpr (KP1_ENV_LD_AUX == 0) {
      return at<int32_t>(ints, f, n, i);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ints, 0, 0x7fffffff, 0x00020000);
      return __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(((uint32_t)f * (uint32_t)n + (uint32_t)i) * 4u), 0, KP1_ENV_LD_AUX));
    }
  }
  __device__ __forceinline__ double q_loadl(int k, int64_t i) const {
    if constexpr (sizeof(R) == 4) return (double)rl(F_Q + k, i) + (double)rl(F_QLO + k, i);
    else return (double)r(F_Q + k, i);
  }
  __device__ __forceinline__ int32_t& iv(int f, int64_t i) const { return at<int32_t>(ints, f, n, i); }
  // joint position as the kinematic chain carries it (fp64).  fp32 handle: a two-float value, 48 significant bits.
  __device__ __forceinline__ double q_load(int k, int64_t i) const {
    if constexpr (sizeof(R) == 4) return (double)r(F_Q + k, i) + (double)r(F_QLO + k, i);
    else return (double)r(F_Q + k, i);
  }
  __device__ __forceinline__ void q_store(int k, int64_t i, double q) const {
    if constexpr (sizeof(R) == 4) {
      const float hi = (float)q;
      r(F_Q + k, i) = (R)hi;
      r(F_QLO + k, i) = (R)(float)(q - (double)hi);
    } else {
      r(F_Q + k, i) = (R)q;
    }
  }
};

// ---- forward kinematics (V51/ee_fk.py:98-134) ----------------------------------------------
// Bit-reproducibility: contraction is off and every multiply-add is an explicit fma, so the same q gives the same
// pose bits at every call site (reset, step, set_state).  A zero action therefore leaves the pose error exactly
// unchanged, as in the reference, and cannot bump the drift counter through last-bit noise.
// fp64 sin and cos of a joint angle for the fp32 handle's chain: |x| <= 64 (joint limits are inside +-2 pi), so the argument reduction is
// two fused steps against a 33-bit / tail split of pi/2 (n * PIO2_HI is exact for |n| < 2^20) and needs no large-argument path; kernels
// are the fdlibm k_sin / k_cos minimax polynomials on [-pi/4, pi/4] (Sun Microsystems' public coefficients; < 1 ulp).  ~40 instructions
// against ~130 for the general-argument library sincos.
__device__ __forceinline__ void kp_sincos_kin(double x, double* s, double* c) {
  const double n = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
  const double z = r * r, w = z * z;
  const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * 2.75573137070700676789e-06) +
                    z * w * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10);
  const double sn = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                    (w * w) * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
  const double hz = 0.5 * z, om = 1.0 - hz;
  const double cs = om + (((1.0 - om) - hz) + z * pc);
  const int quad = (int)n;
  const double a = (quad & 1) ? cs : sn, b = (quad & 1) ? sn : cs;
  *s = (quad & 2) ? -a : a;
  *c = ((quad + 1) & 2) ? -b : b;
}

// the same chain as straight-line code specialised to the robot constants (tools/gen_fk_chain.py): what the fp32 handle runs
#include "kp1_fk_generated.inc"
#ifndef KP1_FK_GENERATED
#define KP1_FK_GENERATED 1   // 0: the fp32 handle runs the generic chain on the constants of DevFk (A/B switch; the fp64 handle always does)
#endif

// Rotation and position of the chain.  FAST selects kp_sincos_kin (fp32 handle); the fp64 handle keeps the library sincos it is pinned with.
template <typename R, bool FAST>
__device__ __forceinline__ void fk_chain(const DevFk<R>& __restrict__ k, const R* __restrict__ q, R* __restrict__ p, R* __restrict__ Rm) {
#pragma clang fp contract(off)
  R s, c;
  // joint 0 (prismatic) + joint 1 origin: pure translation
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = kp_fma(k.v0[i], q[0], k.p01[i]);
  if constexpr (FAST) kp_sincos_kin(q[1], &s, &c);
  else kp_sincos(q[1], &s, &c);
#pragma unroll
  for (int e = 0; e < 9; ++e) Rm[e] = kp_fma(s, k.ks[0][e], kp_fma(c, k.kc[0][e], k.k1[0][e]));
#pragma unroll
  for (int j = 2; j < NJ; ++j) {
    const int m = j - 1;
    if constexpr (FAST) kp_sincos_kin(q[j], &s, &c);
    else kp_sincos(q[j], &s, &c);
    R D[9], Rn[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) D[e] = kp_fma(s, k.ks[m][e], kp_fma(c, k.kc[m][e], k.k1[m][e]));
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      p[r] = kp_fma(Rm[3 * r + 2], k.p[j - 2][2], kp_fma(Rm[3 * r + 1], k.p[j - 2][1], kp_fma(Rm[3 * r + 0], k.p[j - 2][0], p[r])));
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Rn[3 * r + cc] = kp_fma(Rm[3 * r + 2], D[6 + cc], kp_fma(Rm[3 * r + 1], D[3 + cc], Rm[3 * r + 0] * D[cc]));
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) Rm[e] = Rn[e];
  }
}

template <typename R>
__device__ __forceinline__ void fk_pose6(const DevFk<R>& __restrict__ k, const R* __restrict__ q, R* __restrict__ pose) {
#pragma clang fp contract(off)
  R Rm[9], p[3];
  fk_chain<R, false>(k, q, p, Rm);
  pose[0] = p[0];
  pose[1] = p[1];
  pose[2] = p[2];
  pose[3] = kp_atan2(Rm[7], Rm[8]);                                              // roll  = atan2(R21, R22)
  pose[4] = kp_atan2(-Rm[6], kp_sqrt(kp_fma(Rm[3], Rm[3], Rm[0] * Rm[0])));      // pitch = atan2(-R20, sqrt(R00^2 + R10^2))
  pose[5] = kp_atan2(Rm[3], Rm[0]);                                              // yaw   = atan2(R10, R00)
}

// The chain as the handles call it: fp64 constants and fp64 q on both.
//   fp64 handle: fk_pose6<double> as is.
//   fp32 handle: fp64 products (the small entries of a near-gimbal-lock rotation keep their RELATIVE accuracy), then the three angles from
//   the entries rounded to fp32: atan2 is a function of the ratio, an fp32 ratio costs <= 1.2e-7 relative, i.e. <= 6e-8 rad -- below the
//   fp32 spacing of the angle it is stored as.  Same bits at every call site (reset, step, set_state), as above.
template <typename R>
__device__ __forceinline__ void fk_pose6_kin(const DevFk<double>& __restrict__ k, const double* __restrict__ q, R* __restrict__ pose) {
  if constexpr (sizeof(R) == 8) {
    fk_pose6<double>(k, q, pose);
  } else {
#pragma clang fp contract(off)
    double Rm[9], p[3];
#if KP1_FK_GENERATED
    fk_chain_generated(q, p, Rm);      // (k is not read: kp1_create has checked that the generated constants are fold_fk's, bit for bit)
#else
    fk_chain<double, true>(k, q, p, Rm);
#endif
    const float r0 = (float)Rm[0], r3 = (float)Rm[3], r6 = (float)Rm[6], r7 = (float)Rm[7], r8 = (float)Rm[8];
    pose[0] = (R)p[0];
    pose[1] = (R)p[1];
    pose[2] = (R)p[2];
    pose[3] = atan2f(r7, r8);
    pose[4] = atan2f(-r6, sqrtf(__builtin_fmaf(r3, r3, r0 * r0)));
    pose[5] = atan2f(r3, r0);
  }
}

// KP1/kinematics/pose_utils.py:11-12 wrap_to_pi, numpy floor-mod
template <typename R>
__device__ __forceinline__ R wrap_to_pi(R v) {
#pragma clang fp contract(off)
  const R PI = (R)3.141592653589793;
  const R TWO_PI = (R)2.0 * PI;
  const R x = v + PI;
  // Both angles of a pose difference are atan2 results, so x lies in [-pi, 3 pi]: there fmod is the identity, or ONE subtraction that is
  // exact (Sterbenz: pi <= x <= 4 pi) -- the same bits as the library loop, without the loop.  Anything else (poses handed in through
  // set_state / reset options) takes the general path.
  R m;
  if (x >= (R)0 && x < TWO_PI) m = x;
  else if (x >= TWO_PI && x < (R)2 * TWO_PI) m = x - TWO_PI;
  else if (x < (R)0 && x > -TWO_PI) m = x + TWO_PI;
  else {
    m = kp_fmod(x, TWO_PI);
    if (m < (R)0) m += TWO_PI;
  }
  return m - PI;
}
// KP1/kinematics/pose_utils.py:21-30: both error norms of curr vs goal
template <typename R>
__device__ __forceinline__ void pose_error_norms(const R* curr, const R* goal, R* pos_err, R* ori_err, R* pos_norm, R* ori_norm) {
#pragma clang fp contract(off)
  R sp = 0, so = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pos_err[i] = goal[i] - curr[i];
    ori_err[i] = wrap_to_pi<R>(goal[3 + i] - curr[3 + i]);
    sp = kp_fma(pos_err[i], pos_err[i], sp);  // explicit: same bits at every call site (prev vs curr comparison, :263)
    so = kp_fma(ori_err[i], ori_err[i], so);
  }
  *pos_norm = kp_sqrt(sp);
  *ori_norm = kp_sqrt(so);
}
template <typename R>
__device__ __forceinline__ R norm7(const R* v) {
#pragma clang fp contract(off)
  R s = 0;
#pragma unroll
  for (int i = 0; i < NJ; ++i) s = kp_fma(v[i], v[i], s);   // explicit: the same bits in every kernel this is inlined into
  return kp_sqrt(s);
}

// KP1/envs/arm_kinematic_env.py:489-506
template <typename R>
__device__ __forceinline__ R interpolate_control(R pos, R near_t, R far_t, R near_v, R far_v, R fallback) {
  if (near_t <= (R)0 || far_t <= near_t) return fallback;
  if (pos <= near_t) return near_v;
  if (pos >= far_t) return far_v;
  R alpha = kp_div(pos - near_t, kp_max<R>(far_t - near_t, (R)1e-9));
  return near_v + alpha * (far_v - near_v);
}

// KP1/envs/arm_kinematic_env.py:432-444
template <typename R>
__device__ __forceinline__ bool is_near_goal(const DevCfg<R>& c, R pos, R ori) {
  if (pos > c.reward.near_goal_pos_threshold_m) return false;
  if (c.reward.use_orientation_gate && ori > c.reward.near_goal_ori_threshold_rad) return false;
  return true;
}
template <typename R>
__device__ __forceinline__ bool is_pre_near_goal(const DevCfg<R>& c, R pos, R ori) {
  if (pos > c.reward.pre_near_goal_pos_threshold_m) return false;
  if (c.reward.use_orientation_gate && ori > c.reward.near_goal_ori_threshold_rad) return false;
  return true;
}

// inputs shared by both reward functions (everything already reduced to scalars: the reference
// recomputes the four norms inside each reward; they are the same numbers)
template <typename R>
struct RewardIn {
  R prev_pos, curr_pos, prev_ori, curr_ori;
  R action_norm, prev_action_norm, action_msq, action_delta_msq;
  R dq_norm, prev_dq_norm, dq_change_l2, margin_min;
  R entry_pos, entry_ori, entry_action, entry_dq;
  int dwell, entry_count, drift_count;
  bool pre, cn, pn, success;
};

#define KP1_N_APPROACH_COMPONENTS 54
#define KP1_N_DOCK_COMPONENTS 60

// KP1/envs/reward_approach.py:75-373.  comps (optional): strided output, component k at comps[k*stride].
template <typename R, bool COMPS>
__device__ __forceinline__ R approach_reward(const typename DevCfg<R>::Reward& __restrict__ cfg, const RewardIn<R>& in,
                                             R* comps, int64_t stride) {
  const R Z = (R)0;
  const R prev_pos = in.prev_pos, curr_pos = in.curr_pos, prev_ori = in.prev_ori, curr_ori = in.curr_ori;
  const bool pre = in.pre, cn = in.cn, pn = in.pn;
  R position_progress = cfg.position_progress_weight * (prev_pos - curr_pos);
  R global_ori = cfg.orientation_progress_weight * (prev_ori - curr_ori);
  R nf_ori = pre ? cfg.near_field_orientation_progress_weight * (prev_ori - curr_ori) : Z;
  R orientation_progress = global_ori + nf_ori;
  R milestone = Z;
  if (pre)
    for (int i = 0; i < cfg.n_orientation_milestones; ++i)
      if (curr_ori <= cfg.ms_thr[i]) milestone += cfg.ms_bonus[i];
  R nf_center = pre ? -cfg.near_field_orientation_center_weight * curr_ori : Z;
  R pre_near_goal = (pre && !cn) ? cfg.pre_near_goal_bonus : Z;
  R bonus_scale = (R)1;
  {  // decay ** max(entry_count - 1, 0): integer power by repeated multiply only when an entry happens
    int e = kp_maxi(in.entry_count - 1, 0);
    if (cn && !pn && e > 0) bonus_scale = kp_pow(cfg.near_goal_bonus_decay, (R)e);
  }
  R near_goal = (cn && !pn) ? cfg.near_goal_bonus * bonus_scale : Z;
  R inner = (pre && !cn) ? cfg.pre_near_to_near_progress_weight * kp_max<R>(prev_pos - curr_pos, Z) : Z;
  R coarse = (pre && curr_ori <= cfg.coarse_orientation_bonus_threshold_rad) ? cfg.coarse_orientation_bonus : Z;
  bool curr_ho = cfg.handover_pos_threshold_m > Z && curr_pos <= cfg.handover_pos_threshold_m &&
                 (cfg.handover_ori_threshold_rad <= Z || curr_ori <= cfg.handover_ori_threshold_rad);
  bool prev_ho = cfg.handover_pos_threshold_m > Z && prev_pos <= cfg.handover_pos_threshold_m &&
                 (cfg.handover_ori_threshold_rad <= Z || prev_ori <= cfg.handover_ori_threshold_rad);
  R ho_bonus = (curr_ho && !prev_ho) ? cfg.handover_bonus : Z;
  R ho_ret = (curr_ho && prev_ho) ? cfg.handover_retention_bonus : Z;
  R ho_dwell = (curr_ho && in.dwell >= 2) ? cfg.handover_dwell_bonus : Z;
  R ho_leave = (prev_ho && !curr_ho) ? -cfg.handover_leave_penalty : Z;
  R regress = kp_max<R>(curr_pos - prev_pos, Z) + kp_max<R>(curr_ori - prev_ori, Z);
  R ho_regr = (prev_ho || curr_ho) ? -cfg.handover_regression_weight * regress : Z;
  R dwell = (cn && in.dwell >= 2) ? cfg.dwell_bonus : Z;
  int drift_esc = kp_maxi(in.drift_count - cfg.drift_penalty_escalation_start, 0);
  R drift_scale = (R)1 + cfg.drift_penalty_escalation_per_count * (R)drift_esc;
  R drift_w = cfg.drift_penalty_weight * drift_scale;
  R drift_penalty = pn ? -drift_w * kp_max<R>(curr_pos - prev_pos, Z) : Z;
  R leave_penalty = (pn && !cn) ? -cfg.near_goal_leave_penalty : Z;
  const R an = in.action_norm, pan = in.prev_action_norm, dqn = in.dq_norm, pdqn = in.prev_dq_norm;
  bool dc_en = cfg.dock_coarse_ready_pos_threshold_m > Z && cfg.dock_coarse_ready_ori_threshold_rad > Z;
  bool curr_dc_pose = dc_en && curr_pos <= cfg.dock_coarse_ready_pos_threshold_m && curr_ori <= cfg.dock_coarse_ready_ori_threshold_rad;
  bool prev_dc_pose = dc_en && prev_pos <= cfg.dock_coarse_ready_pos_threshold_m && prev_ori <= cfg.dock_coarse_ready_ori_threshold_rad;
  bool curr_dc_motion = (cfg.dock_coarse_ready_action_threshold <= Z || an <= cfg.dock_coarse_ready_action_threshold) &&
                        (cfg.dock_coarse_ready_dq_threshold <= Z || dqn <= cfg.dock_coarse_ready_dq_threshold);
  bool prev_dc_motion = (cfg.dock_coarse_ready_action_threshold <= Z || pan <= cfg.dock_coarse_ready_action_threshold) &&
                        (cfg.dock_coarse_ready_dq_threshold <= Z || pdqn <= cfg.dock_coarse_ready_dq_threshold);
  bool curr_dc = curr_dc_pose && curr_dc_motion, prev_dc = prev_dc_pose && prev_dc_motion;
  bool fr_en = cfg.finisher_ready_pos_threshold_m > Z && cfg.finisher_ready_ori_threshold_rad > Z;
  bool curr_fr_pose = fr_en && curr_pos <= cfg.finisher_ready_pos_threshold_m && curr_ori <= cfg.finisher_ready_ori_threshold_rad;
  bool prev_fr_pose = fr_en && prev_pos <= cfg.finisher_ready_pos_threshold_m && prev_ori <= cfg.finisher_ready_ori_threshold_rad;
  bool curr_fr_motion = (cfg.finisher_ready_action_threshold <= Z || an <= cfg.finisher_ready_action_threshold) &&
                        (cfg.finisher_ready_dq_threshold <= Z || dqn <= cfg.finisher_ready_dq_threshold);
  bool prev_fr_motion = (cfg.finisher_ready_action_threshold <= Z || pan <= cfg.finisher_ready_action_threshold) &&
                        (cfg.finisher_ready_dq_threshold <= Z || pdqn <= cfg.finisher_ready_dq_threshold);
  bool curr_fr = curr_fr_pose && curr_fr_motion, prev_fr = prev_fr_pose && prev_fr_motion;
  bool nh_en = cfg.near_handoff_pos_threshold_m > Z && cfg.near_handoff_ori_threshold_rad > Z;
  bool nh = nh_en && curr_pos <= cfg.near_handoff_pos_threshold_m && curr_ori <= cfg.near_handoff_ori_threshold_rad;
  bool prev_nh = nh_en && prev_pos <= cfg.near_handoff_pos_threshold_m && prev_ori <= cfg.near_handoff_ori_threshold_rad;
  R dc_bonus = (curr_dc && !prev_dc) ? cfg.dock_coarse_ready_bonus : Z;
  R dc_ret = (curr_dc && prev_dc) ? cfg.dock_coarse_ready_retention_bonus : Z;
  R dc_dwell = (curr_dc && in.dwell >= 2) ? cfg.dock_coarse_ready_dwell_bonus : Z;
  R dc_leave = (prev_dc && !curr_dc) ? -cfg.dock_coarse_ready_leave_penalty : Z;
  R dc_regr = (nh || prev_nh || curr_dc_pose || prev_dc_pose) ? -cfg.dock_coarse_ready_regression_weight * regress : Z;
  R fr_bonus = (curr_fr && !prev_fr) ? cfg.finisher_ready_bonus : Z;
  R fr_ret = (curr_fr && prev_fr) ? cfg.finisher_ready_retention_bonus : Z;
  R fr_dwell = (curr_fr && in.dwell >= 2) ? cfg.finisher_ready_dwell_bonus : Z;
  R fr_leave = (prev_fr && !curr_fr) ? -cfg.finisher_ready_leave_penalty : Z;
  R fr_regr = (nh || prev_nh || curr_fr_pose || prev_fr_pose) ? -cfg.finisher_ready_regression_weight * regress : Z;
  bool nh_any = nh || curr_dc_pose || curr_fr_pose;
  R nh_action = nh_any ? -cfg.near_handoff_action_weight * in.action_msq : Z;
  R nh_dq = nh_any ? -cfg.near_handoff_dq_weight * dqn : Z;
  R nh_motion = Z, nh_settle = Z;
  if (nh_any) {
    R at = cfg.finisher_ready_action_threshold != Z ? cfg.finisher_ready_action_threshold : cfg.dock_coarse_ready_action_threshold;
    R dt = cfg.finisher_ready_dq_threshold != Z ? cfg.finisher_ready_dq_threshold : cfg.dock_coarse_ready_dq_threshold;
    R action_clean = at > Z ? kp_max<R>((R)1 - kp_div(an, kp_max<R>(at, (R)1e-9)), Z) : Z;
    R dq_clean = dt > Z ? kp_max<R>((R)1 - kp_div(dqn, kp_max<R>(dt, (R)1e-9)), Z) : Z;
    nh_motion = cfg.near_handoff_motion_bonus_weight * ((R)0.5 * action_clean + (R)0.5 * dq_clean);
    nh_settle = cfg.near_handoff_settle_bonus_weight * ((R)0.5 * kp_max<R>(pan - an, Z) + (R)0.5 * kp_max<R>(pdqn - dqn, Z));
  }
  R same_step = (curr_pos < prev_pos && curr_ori < prev_ori && (pre || nh)) ? cfg.same_step_alignment_bonus : Z;
  R smooth_mult = (curr_ho || prev_ho) ? cfg.handover_smoothness_multiplier : (R)1;
  R smooth = smooth_mult * (-cfg.action_magnitude_weight * in.action_msq - cfg.action_delta_weight * in.action_delta_msq);
  R jl_pen = -cfg.joint_limit_penalty_weight * (kp_max<R>((R)0.25 - in.margin_min, Z) / (R)0.25);
  R success_bonus = in.success ? cfg.success_bonus : Z;

  if constexpr (COMPS) {
    int k = 0;
#define KP1_C(v) comps[(k++) * stride] = (R)(v)
    KP1_C(position_progress); KP1_C(global_ori); KP1_C(nf_ori); KP1_C(orientation_progress); KP1_C(milestone);
    KP1_C(nf_center); KP1_C(pre_near_goal); KP1_C(near_goal); KP1_C(inner); KP1_C((cn && !pn) ? bonus_scale : Z);
    KP1_C(coarse); KP1_C(ho_bonus); KP1_C(ho_ret); KP1_C(ho_dwell); KP1_C(ho_leave); KP1_C(ho_regr); KP1_C(dc_bonus);
    KP1_C(dc_ret); KP1_C(dc_dwell); KP1_C(dc_leave); KP1_C(dc_regr); KP1_C(fr_bonus); KP1_C(fr_ret); KP1_C(fr_dwell);
    KP1_C(fr_leave); KP1_C(fr_regr); KP1_C(nh_action); KP1_C(nh_dq); KP1_C(nh_motion); KP1_C(nh_settle); KP1_C(same_step);
    KP1_C(dwell); KP1_C(drift_penalty); KP1_C(leave_penalty); KP1_C(drift_scale); KP1_C(in.entry_count);
    KP1_C(in.drift_count); KP1_C(smooth); KP1_C(smooth_mult); KP1_C(jl_pen); KP1_C(success_bonus); KP1_C(curr_pos);
    KP1_C(curr_ori); KP1_C(an); KP1_C(dqn); KP1_C(in.dwell); KP1_C(pre ? 1 : 0); KP1_C(cn ? 1 : 0); KP1_C(curr_ho ? 1 : 0);
    KP1_C(curr_dc ? 1 : 0); KP1_C(curr_dc_pose ? 1 : 0); KP1_C(curr_fr ? 1 : 0); KP1_C(curr_fr_pose ? 1 : 0); KP1_C(nh ? 1 : 0);
  }
  // reward_approach.py:334-372 summation order
  R r = Z;
  r += position_progress; r += orientation_progress; r += milestone; r += nf_center; r += pre_near_goal; r += near_goal;
  r += inner; r += coarse; r += ho_bonus; r += ho_ret; r += ho_dwell; r += ho_leave; r += ho_regr; r += dc_bonus;
  r += dc_ret; r += dc_dwell; r += dc_leave; r += dc_regr; r += fr_bonus; r += fr_ret; r += fr_dwell; r += fr_leave;
  r += fr_regr; r += nh_action; r += nh_dq; r += nh_motion; r += nh_settle; r += same_step; r += dwell;
  r += drift_penalty; r += leave_penalty; r += smooth; r += jl_pen; r += success_bonus;
  return r;
}

// KP1/envs/reward_dock.py:105-120
template <typename R>
__device__ __forceinline__ R entry_penalty_scale(R pos, R near_t, R far_t, R near_m, R far_m) {
  if (near_t <= (R)0 || far_t <= near_t) return (R)1;
  if (pos <= near_t) return near_m;
  if (pos >= far_t) return far_m;
  R alpha = kp_div(pos - near_t, kp_max<R>(far_t - near_t, (R)1e-9));
  return near_m + alpha * (far_m - near_m);
}

// KP1/envs/reward_dock.py:123-484
template <typename R, bool COMPS>
__device__ __forceinline__ R dock_reward(const typename DevCfg<R>::DockReward& __restrict__ cfg, const RewardIn<R>& in,
                                         R* comps, int64_t stride) {
  const R Z = (R)0, ONE = (R)1, EPS = (R)1e-9;
  const R prev_pos = in.prev_pos, curr_pos = in.curr_pos, prev_ori = in.prev_ori, curr_ori = in.curr_ori;
  const bool cn = in.cn, pn = in.pn;
  const int dwell_count = in.dwell;
  R position_progress = cfg.position_progress_weight * (prev_pos - curr_pos);
  R orientation_progress = cfg.orientation_progress_weight * (prev_ori - curr_ori);
  R stay = cn ? cfg.stay_in_zone_bonus : Z;
  R dwell_bonus = cn ? cfg.dwell_bonus * (R)kp_maxi(dwell_count - 1, 0) : Z;
  R wr_bonus = cn ? cfg.working_range_bonus : Z;
  R wr_dwell = (cn && dwell_count >= cfg.working_range_dwell_start)
                   ? cfg.working_range_dwell_bonus * (R)kp_maxi(dwell_count - cfg.working_range_dwell_start + 1, 0) : Z;
  bool curr_tight = curr_pos <= cfg.tight_pose_pos_threshold_m && curr_ori <= cfg.tight_pose_ori_threshold_rad;
  bool prev_tight = prev_pos <= cfg.tight_pose_pos_threshold_m && prev_ori <= cfg.tight_pose_ori_threshold_rad;
  R ns_pos_t = cfg.near_strict_pos_threshold_m != Z ? cfg.near_strict_pos_threshold_m : cfg.tight_pose_pos_threshold_m * (R)2;
  R ns_ori_t = cfg.near_strict_ori_threshold_rad != Z ? cfg.near_strict_ori_threshold_rad : cfg.tight_pose_ori_threshold_rad * (R)3;
  bool curr_ns = curr_pos <= ns_pos_t && curr_ori <= ns_ori_t;
  bool prev_ns = prev_pos <= ns_pos_t && prev_ori <= ns_ori_t;
  R tp = kp_max<R>(cfg.tight_pose_pos_threshold_m, EPS), to = kp_max<R>(cfg.tight_pose_ori_threshold_rad, EPS);
  R spc = kp_max<R>(ONE - kp_div(curr_pos, tp), Z);
  R soc = kp_max<R>(ONE - kp_div(curr_ori, to), Z);
  R sc_base = (R)0.8 * spc + (R)0.2 * soc;
  R strict_closeness = sc_base * sc_base;
  R tight_bonus = curr_tight ? cfg.tight_pose_bonus : Z;
  R tight_dwell = curr_tight ? cfg.tight_pose_dwell_bonus * (R)kp_maxi(dwell_count - 1, 0) : Z;
  R strict_leave = (prev_tight && !curr_tight) ? -cfg.strict_pose_leave_penalty : Z;
  R sc_reward = curr_tight ? cfg.strict_center_reward_weight * strict_closeness : Z;
  R rp = kp_div(curr_pos, tp), ro = kp_div(curr_ori, to);
  R sc_pos_pen = cfg.strict_center_position_weight > Z ? -cfg.strict_center_position_weight * (rp * rp) : Z;
  R sc_ori_pen = cfg.strict_center_orientation_weight > Z ? -cfg.strict_center_orientation_weight * (ro * ro) : Z;
  R action_rms = kp_sqrt(in.action_msq);
  R sc_small = Z;
  if (cfg.strict_center_small_action_bonus_weight > Z && cfg.strict_center_small_action_pos_radius_m > Z &&
      cfg.strict_center_small_action_ori_radius_rad > Z && cfg.strict_center_small_action_scale > Z && curr_tight) {
    R cpc = kp_max<R>(ONE - kp_div(curr_pos, cfg.strict_center_small_action_pos_radius_m), Z);
    R coc = kp_max<R>(ONE - kp_div(curr_ori, cfg.strict_center_small_action_ori_radius_rad), Z);
    R cc = kp_pow((R)0.8 * cpc + (R)0.2 * coc, cfg.strict_center_small_action_power);
    R sm = kp_max<R>(ONE - kp_div(action_rms, cfg.strict_center_small_action_scale), Z);
    sc_small = cfg.strict_center_small_action_bonus_weight * cc * sm;
  }
  R sc_dwell = Z;
  if (curr_tight && cfg.strict_center_dwell_bonus_weight > Z && dwell_count >= cfg.strict_center_dwell_start) {
    int esc = kp_maxi(dwell_count - cfg.strict_center_dwell_escalation_start, 0);
    R scale = ONE + cfg.strict_center_dwell_escalation_per_step * (R)esc;
    sc_dwell = cfg.strict_center_dwell_bonus_weight * strict_closeness * scale;
  }
  R tps = cfg.tight_position_shaping_radius_m > Z
              ? cfg.tight_position_shaping_weight * kp_max<R>(ONE - kp_div(curr_pos, kp_max<R>(cfg.tight_position_shaping_radius_m, EPS)), Z) : Z;
  R tos = cfg.tight_orientation_shaping_radius_rad > Z
              ? cfg.tight_orientation_shaping_weight * kp_max<R>(ONE - kp_div(curr_ori, kp_max<R>(cfg.tight_orientation_shaping_radius_rad, EPS)), Z) : Z;
  R conv_pos = (cfg.convergence_position_radius_m > Z && kp_min<R>(prev_pos, curr_pos) <= cfg.convergence_position_radius_m)
                   ? cfg.convergence_position_progress_weight * (prev_pos - curr_pos) : Z;
  R gate_scale = (cfg.position_first_orientation_pos_threshold_m > Z && curr_pos > cfg.position_first_orientation_pos_threshold_m)
                     ? cfg.position_first_orientation_pre_scale : ONE;
  R conv_ori = (cfg.convergence_orientation_radius_rad > Z && kp_min<R>(prev_ori, curr_ori) <= cfg.convergence_orientation_radius_rad)
                   ? gate_scale * cfg.convergence_orientation_progress_weight * (prev_ori - curr_ori) : Z;
  R leave_zone = (pn && !cn) ? -cfg.leave_zone_penalty : Z;
  R wr_exit = (pn && !cn) ? -cfg.working_range_exit_penalty : Z;
  R dpos = kp_max<R>(curr_pos - prev_pos, Z), dori = kp_max<R>(curr_ori - prev_ori, Z);
  R drift = -cfg.drift_penalty_position_weight * dpos;
  drift += -cfg.drift_penalty_orientation_weight * dori;
  if (curr_tight || prev_tight) drift *= cfg.strict_zone_drift_penalty_multiplier;
  const R action_l2 = in.action_norm;
  R eps_scale = entry_penalty_scale<R>(kp_max<R>(prev_pos, curr_pos), cfg.entry_action_penalty_near_pos_threshold_m,
                                       cfg.entry_action_penalty_far_pos_threshold_m, cfg.entry_action_penalty_near_multiplier,
                                       cfg.entry_action_penalty_far_multiplier);
  R smooth = -cfg.action_magnitude_weight * in.action_msq;
  smooth += -cfg.action_delta_weight * in.action_delta_msq;
  if (curr_tight) smooth *= cfg.strict_zone_action_penalty_multiplier;
  smooth *= eps_scale;
  R action_delta_rms = kp_sqrt(in.action_delta_msq);
  R adv = (cfg.action_delta_violation_weight > Z && cfg.action_delta_violation_threshold > Z)
              ? -cfg.action_delta_violation_weight * eps_scale * kp_max<R>(action_delta_rms - cfg.action_delta_violation_threshold, Z) : Z;
  R dqc = (cfg.delta_q_change_penalty_weight > Z && cfg.delta_q_change_penalty_threshold > Z)
              ? -cfg.delta_q_change_penalty_weight * eps_scale * kp_max<R>(in.dq_change_l2 - cfg.delta_q_change_penalty_threshold, Z) : Z;
  const R entry_pos = in.entry_pos, entry_ori = in.entry_ori;
  R preserve = Z;
  if (cfg.preserve_state_bonus > Z && (curr_ns || curr_tight)) {
    bool pos_ok = curr_pos <= entry_pos + cfg.preserve_position_tolerance_m;
    bool ori_ok = curr_ori <= entry_ori + cfg.preserve_orientation_tolerance_rad;
    if (pos_ok && ori_ok) preserve = cfg.preserve_state_bonus;
  }
  R strict_hold = curr_tight ? cfg.strict_hold_bonus * (R)kp_maxi(dwell_count - 1, 0) : Z;
  R low_motion = Z;
  if (cfg.low_motion_bonus > Z && curr_ns && (cfg.low_motion_action_threshold <= Z || action_l2 <= cfg.low_motion_action_threshold) &&
      (cfg.low_motion_dq_threshold <= Z || in.dq_norm <= cfg.low_motion_dq_threshold))
    low_motion = cfg.low_motion_bonus;
  R tiny = Z;
  if (cfg.tiny_correction_bonus > Z && curr_ns && !curr_tight) {
    bool improved = curr_pos <= prev_pos && curr_ori <= prev_ori;
    bool small = cfg.tiny_correction_action_threshold <= Z || action_l2 <= cfg.tiny_correction_action_threshold;
    if (improved && small) tiny = cfg.tiny_correction_bonus;
  }
  R worse = Z;
  worse += -cfg.worse_than_entry_position_weight * kp_max<R>(curr_pos - entry_pos - cfg.worse_than_entry_position_tolerance_m, Z);
  worse += -cfg.worse_than_entry_orientation_weight * kp_max<R>(curr_ori - entry_ori - cfg.worse_than_entry_orientation_tolerance_rad, Z);
  R ns_regr = Z;
  if (curr_ns || prev_ns)
    ns_regr = -cfg.near_strict_regression_multiplier * (cfg.drift_penalty_position_weight * dpos + cfg.drift_penalty_orientation_weight * dori);
  R agg_scale = curr_ns ? cfg.near_strict_action_penalty_multiplier : ONE;
  R aggressive = (cfg.aggressive_action_weight > Z && cfg.aggressive_action_threshold > Z)
                     ? -cfg.aggressive_action_weight * agg_scale * kp_max<R>(action_l2 - cfg.aggressive_action_threshold, Z) : Z;
  R dqp_scale = curr_ns ? cfg.near_strict_dq_penalty_multiplier : ONE;
  R dq_pen = (cfg.dq_penalty_weight > Z && cfg.dq_penalty_threshold > Z)
                 ? -cfg.dq_penalty_weight * dqp_scale * kp_max<R>(in.dq_norm - cfg.dq_penalty_threshold, Z) : Z;
  R jl_pen = -cfg.joint_limit_penalty_weight * (kp_max<R>((R)0.25 - in.margin_min, Z) / (R)0.25);
  R success_bonus = in.success ? cfg.success_bonus : Z;
  R b_outer = Z, b_inner = Z, b_dwell = Z, b_outer_exit = Z, b_inner_exit = Z, b_dwell_break = Z, b_drift = Z;
  int zone = 0;
  if (cfg.basin_outer_radius_m > Z && cfg.basin_inner_radius_m > Z && cfg.basin_dwell_radius_m > Z) {
    R outer_r = kp_max<R>(cfg.basin_outer_radius_m, EPS), inner_r = kp_max<R>(cfg.basin_inner_radius_m, EPS), dwell_r = kp_max<R>(cfg.basin_dwell_radius_m, EPS);
    bool p_o = prev_pos <= outer_r, p_i = prev_pos <= inner_r, p_d = prev_pos <= dwell_r;
    bool c_o = curr_pos <= outer_r, c_i = curr_pos <= inner_r, c_d = curr_pos <= dwell_r;
    zone = c_d ? 3 : (c_i ? 2 : (c_o ? 1 : 0));
    if (c_o) b_outer = cfg.basin_outer_bonus * (ONE + kp_max<R>(ONE - kp_div(curr_pos, outer_r), Z));
    if (c_i) b_inner = cfg.basin_inner_bonus * (ONE + kp_max<R>(ONE - kp_div(curr_pos, inner_r), Z));
    if (c_d) b_dwell = cfg.basin_dwell_bonus * (ONE + kp_max<R>(ONE - kp_div(curr_pos, dwell_r), Z));
    b_outer_exit = (p_o && !c_o) ? -cfg.basin_outer_exit_penalty : Z;
    b_inner_exit = (p_i && !c_i) ? -cfg.basin_inner_exit_penalty : Z;
    b_dwell_break = (p_d && !c_d) ? -cfg.basin_dwell_break_penalty : Z;
    b_drift = (p_o || c_o) ? -cfg.basin_drift_penalty_weight * dpos : Z;
  }
  if constexpr (COMPS) {
    int k = 0;
    // strict_center_small_action_bonus is reported 0 outside the tight pose; identical to the reference (:199-203)
    KP1_C(position_progress); KP1_C(orientation_progress); KP1_C(stay); KP1_C(dwell_bonus); KP1_C(wr_bonus); KP1_C(wr_dwell);
    KP1_C(tight_bonus); KP1_C(tight_dwell); KP1_C(strict_leave); KP1_C(sc_reward); KP1_C(sc_pos_pen); KP1_C(sc_ori_pen);
    KP1_C(sc_small); KP1_C(sc_dwell); KP1_C(tps); KP1_C(tos); KP1_C(conv_pos); KP1_C(conv_ori); KP1_C(gate_scale);
    KP1_C(eps_scale); KP1_C(leave_zone); KP1_C(wr_exit); KP1_C(drift); KP1_C(smooth); KP1_C(adv); KP1_C(dqc); KP1_C(preserve);
    KP1_C(strict_hold); KP1_C(low_motion); KP1_C(tiny); KP1_C(worse); KP1_C(ns_regr); KP1_C(aggressive); KP1_C(dq_pen);
    KP1_C(jl_pen); KP1_C(success_bonus); KP1_C(b_outer); KP1_C(b_inner); KP1_C(b_dwell); KP1_C(b_outer_exit);
    KP1_C(b_inner_exit); KP1_C(b_dwell_break); KP1_C(b_drift); KP1_C(zone); KP1_C(curr_pos); KP1_C(curr_ori); KP1_C(dwell_count);
    KP1_C(curr_tight ? 1 : 0); KP1_C(curr_ns ? 1 : 0); KP1_C(entry_pos); KP1_C(entry_ori); KP1_C(in.entry_action);
    KP1_C(in.entry_dq); KP1_C(curr_pos - entry_pos); KP1_C(curr_ori - entry_ori); KP1_C(action_l2 - in.entry_action);
    KP1_C(in.dq_norm - in.entry_dq); KP1_C(in.entry_count); KP1_C(in.drift_count); KP1_C(cn ? 1 : 0);
#undef KP1_C
  }
  // reward_dock.py:438-483 summation order
  R r = Z;
  r += position_progress; r += orientation_progress; r += stay; r += dwell_bonus; r += wr_bonus; r += wr_dwell;
  r += tight_bonus; r += tight_dwell; r += strict_leave; r += sc_reward; r += sc_pos_pen; r += sc_ori_pen; r += sc_small;
  r += sc_dwell; r += tps; r += tos; r += conv_pos; r += conv_ori; r += leave_zone; r += wr_exit; r += drift; r += smooth;
  r += adv; r += dqc; r += preserve; r += strict_hold; r += low_motion; r += tiny; r += worse; r += ns_regr;
  r += aggressive; r += dq_pen; r += jl_pen; r += success_bonus; r += b_outer; r += b_inner; r += b_dwell;
  r += b_outer_exit; r += b_inner_exit; r += b_dwell_break; r += b_drift;
  return r;
}

// ---- joint utilities (KP1/kinematics/joint_limits.py) ---------------------------------------
// One joint at a time; the step kernel, the observation builder and the kp1_joint_utils entry point (device-side check of the
// reference's joint_utils fixture) all go through these, so what the fixture pins is what the hot path runs.
template <typename R>
__device__ __forceinline__ R joint_span(R lo, R hi) { return kp_max<R>(hi - lo, (R)1e-9); }
// clip_joint_configuration :133-135
template <typename R>
__device__ __forceinline__ R joint_clip(R q, R lo, R hi) { return kp_clip<R>(q, lo, hi); }
// joint_limit_margin :166-174
template <typename R>
__device__ __forceinline__ R joint_limit_margin(R q, R lo, R hi) {
  const R span = joint_span<R>(lo, hi);
  return kp_clip<R>((R)2 * kp_min<R>(kp_div(q - lo, span), kp_div(hi - q, span)), (R)0, (R)1);
}
// normalize_joint_positions :153-158
template <typename R>
__device__ __forceinline__ R joint_normalize_q(R q, R lo, R hi) { return kp_clip<R>((R)2 * kp_div(q - lo, joint_span<R>(lo, hi)) - (R)1, (R)-1, (R)1); }
// normalize_joint_deltas :161-163
template <typename R>
__device__ __forceinline__ R joint_normalize_dq(R dq, R dlim) { return kp_clip<R>(kp_div(dq, kp_max<R>(dlim, (R)1e-9)), (R)-1, (R)1); }

// KP1/envs/observation_builder.py:29-94 -> one row of 56 floats in SB3 key order (kp1.h KP1_OBS_*)
template <typename R>
__device__ __forceinline__ void build_observation(const DevCfg<R>& __restrict__ c, int mode, const R* q, const R* dq, const R* prev_action,
                                                  const R* pos_err, const R* ori_err, int episode_step, int dwell_count, float* o) {
#pragma unroll
  for (int i = 0; i < KP1_OBS_DIM; ++i) o[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    o[KP1_OBS_Q + i] = (float)joint_normalize_q<R>(q[i], c.lower[i], c.upper[i]);
    o[KP1_OBS_DQ + i] = (float)joint_normalize_dq<R>(dq[i], c.dlim[i]);
    o[KP1_OBS_PREV_ACTION + i] = (float)kp_clip<R>(prev_action[i], (R)-1, (R)1);
    o[KP1_OBS_JOINT_LIMIT_MARGIN + i] = (float)joint_limit_margin<R>(q[i], c.lower[i], c.upper[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[KP1_OBS_GOAL_POS_ERR + i] = (float)kp_clip<R>(kp_div(pos_err[i], c.obs.pos_err_scale_m), (R)-1, (R)1);
    o[KP1_OBS_GOAL_ORI_ERR + i] = (float)kp_clip<R>(kp_div(ori_err[i], c.obs.ori_err_scale_rad), (R)-1, (R)1);
  }
  o[KP1_OBS_TASK_TYPE] = 1.0f;
  o[KP1_OBS_MODE_FLAG + (mode == KP1_MODE_APPROACH ? 0 : 1)] = 1.0f;
  R ep = kp_div((R)episode_step, (R)kp_maxi(c.env.episode_length, 1));
  R dp = kp_div((R)dwell_count, (R)kp_maxi(c.env.dwell_steps_target, 1));
  o[KP1_OBS_PROGRESS + 0] = (float)kp_clip<R>(ep, (R)0, (R)1);
  o[KP1_OBS_PROGRESS + 1] = (float)kp_clip<R>(dp, (R)0, (R)1);
}

// ---- numpy Generator(PCG64) on device -------------------------------------------------------
// PCG64 XSL-RR 128/64 (numpy/random/src/pcg64/pcg64.h), buffered next_uint32, next_double,
// uniform and Lemire bounded integers (numpy/random/src/distributions/distributions.c).
struct Pcg {
  unsigned __int128 state, inc;
  uint32_t has_uint32, uinteger;
};
__device__ __forceinline__ uint64_t pcg_next64(Pcg& r) {
  const unsigned __int128 MULT = (((unsigned __int128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;
  r.state = r.state * MULT + r.inc;
  uint64_t hi = (uint64_t)(r.state >> 64), lo = (uint64_t)r.state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
__device__ __forceinline__ uint32_t pcg_next32(Pcg& r) {
  if (r.has_uint32) {
    r.has_uint32 = 0;
    return r.uinteger;
  }
  uint64_t n = pcg_next64(r);
  r.has_uint32 = 1;
  r.uinteger = (uint32_t)(n >> 32);
  return (uint32_t)n;
}
__device__ __forceinline__ double pcg_double(Pcg& r) { return (double)(pcg_next64(r) >> 11) * (1.0 / 9007199254740992.0); }
// Generator.integers(low, high) for the small ranges the samplers use (range < 2^32)
__device__ __forceinline__ int pcg_integers(Pcg& r, int low, int high_exclusive) {
  uint32_t rng = (uint32_t)(high_exclusive - 1 - low);
  if (rng == 0) return low;
  uint32_t rng_excl = rng + 1u;
  uint64_t m = (uint64_t)pcg_next32(r) * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (leftover < rng_excl) {
    uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
    while (leftover < threshold) {
      m = (uint64_t)pcg_next32(r) * rng_excl;
      leftover = (uint32_t)m;
    }
  }
  return low + (int)(m >> 32);
}
// Generator.uniform: low + (high - low) * u with the product rounded BEFORE the add (numpy does not fuse);
// contraction is switched off here so the device compiler cannot turn it into an FMA (HIP's __dmul_rn/__dadd_rn are
// plain operators and would still be contracted): samples stay bit-exact.
__device__ __forceinline__ double uniform_scale(double lo, double range, double u) {
#pragma clang fp contract(off)
  double prod = range * u;
  return lo + prod;
}
__device__ __forceinline__ void pcg_uniform_sym7(Pcg& r, const double* noise, double* out) {
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double lo = -noise[i], range = noise[i] - lo;
    out[i] = uniform_scale(lo, range, pcg_double(r));
  }
}
__device__ __forceinline__ bool any_positive7(const double* v) {
  bool a = false;
#pragma unroll
  for (int i = 0; i < NJ; ++i) a = a || (v[i] > 0.0);
  return a;
}
__device__ __forceinline__ double dclip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ int opt_or(int v, int dflt) { return v == KP1_UNSET ? dflt : v; }

// KP1/envs/curriculum.py:90-101
__device__ __forceinline__ void sample_stage_joint_target(const DevSampler& __restrict__ s, Pcg& r, const double* base, const double* noise, double* out) {
  double d[NJ];
  bool any = any_positive7(noise);
  if (any) pcg_uniform_sym7(r, noise, d);
#pragma unroll
  for (int i = 0; i < NJ; ++i) out[i] = dclip(any ? base[i] + d[i] : base[i], s.lower[i], s.upper[i]);
}
// KP1/kinematics/joint_limits.py:138-150
__device__ __forceinline__ void sample_joint_configuration(const DevSampler& __restrict__ s, Pcg& r, double margin_fraction, double* out) {
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    double span = s.upper[i] - s.lower[i];
    double margin = fmax(span * margin_fraction, 1e-6);
    double lo = s.lower[i] + margin, hi = s.upper[i] - margin;
    out[i] = uniform_scale(lo, hi - lo, pcg_double(r));
  }
}
// KP1/envs/reset_samplers.py:344-390
__device__ __forceinline__ int sample_workspace_stage_index(const DevSampler& __restrict__ s, Pcg& r, int current_stage_index) {
  const kp1_stage_sampling& c = s.ss;
  int current = kp_clipi(current_stage_index, 0, kp_maxi(s.n_stages - 1, 0));
  if (!c.enabled || current <= 0) return current;
  double cr = fmax(c.current_stage_ratio, 0.0), pr = fmax(c.previous_stage_ratio, 0.0);
  double orr = fmax(c.old_workspace_replay_ratio, 0.0), fr = fmax(c.failure_replay_ratio, 0.0);
  double total = cr + pr + orr + fr;
  if (total <= 0.0) return current;
  double draw = pcg_double(r) * total;
  if (draw < cr) return current;
  draw -= cr;
  if (draw < pr && current > 0) {
    int low = kp_maxi(c.previous_stage_min_index, 0);
    int high = kp_maxi(current - 1, low);
    return pcg_integers(r, low, high + 1);
  }
  draw -= pr;
  int old_max = opt_or(c.old_workspace_max_stage_index, kp_mini(5, current));
  old_max = kp_clipi(old_max, 0, kp_mini(s.n_stages - 1, current));
  if (draw < orr && old_max >= 0) return pcg_integers(r, 0, old_max + 1);
  int replay_max = kp_maxi(kp_mini(old_max, current - 1), 0);
  return replay_max > 0 ? pcg_integers(r, 0, replay_max + 1) : current;
}

enum { SRC_HOME = 0, SRC_OLD_SUCCESS, SRC_RANDOM_VALID, SRC_FRONTIER, SRC_FAILURE_RECOVERY, SRC_STRESS };
// KP1/envs/reset_samplers.py:321-341
__device__ __forceinline__ int sample_target_stage_for_source(const DevSampler& __restrict__ s, Pcg& r, int source, int current) {
  const kp1_random_start& c = s.rs;
  const int n = s.n_stages;
  if (source == SRC_HOME || source == SRC_OLD_SUCCESS) {
    int mx = kp_clipi(opt_or(c.known_target_max_stage_index, kp_mini(7, current)), 0, n - 1);
    return pcg_integers(r, 0, mx + 1);
  }
  if (source == SRC_FRONTIER) {
    int mn = kp_clipi(opt_or(c.frontier_target_min_stage_index, kp_mini(8, current)), 0, n - 1);
    int mx = kp_clipi(opt_or(c.frontier_target_max_stage_index, current), mn, n - 1);
    return pcg_integers(r, mn, mx + 1);
  }
  if (source == SRC_STRESS) {
    int mn = kp_clipi(opt_or(c.stress_target_min_stage_index, kp_mini(8, current)), 0, n - 1);
    int mx = kp_clipi(opt_or(c.stress_target_max_stage_index, n - 1), mn, n - 1);
    return pcg_integers(r, mn, mx + 1);
  }
  int mx = kp_clipi(opt_or(c.mixed_target_max_stage_index, current), 0, n - 1);
  return pcg_integers(r, 0, mx + 1);
}

struct ResetSample {
  double initial_q[NJ], goal_q[NJ], goal_pose6[6], initial_dq[NJ], initial_prev_action[NJ];
  int stage;
};

// KP1/envs/reset_samplers.py:213-305
__device__ __forceinline__ void sample_random_start_pair(const DevSampler& __restrict__ s, Pcg& r, int stage_index, ResetSample& o) {
  const kp1_random_start& c = s.rs;
  const int n = s.n_stages;
  int current = kp_clipi(stage_index, 0, n - 1);
  double ratios[6] = {c.home_start_ratio, c.old_successful_start_ratio, c.random_valid_q_start_ratio,
                      c.frontier_pair_ratio, c.failure_recovery_start_ratio, c.stress_start_ratio};
  // _sample_ratio_key :308-318
  int source = SRC_OLD_SUCCESS;
  {
    double total = 0.0;
    for (int i = 0; i < 6; ++i) {
      ratios[i] = fmax(ratios[i], 0.0);
      total += ratios[i];
    }
    if (total > 0.0) {
      double draw = pcg_double(r) * total;
      for (int i = 0; i < 6; ++i) {
        if (draw <= ratios[i]) {
          source = i;
          break;
        }
        draw -= ratios[i];
      }
    }
  }
  int target_stage = sample_target_stage_for_source(s, r, source, current);
  double target_q[NJ], start_q[NJ];
  sample_stage_joint_target(s, r, s.stages[target_stage].goal_q, s.stages[target_stage].goal_noise, target_q);
  if (source == SRC_HOME) {
    int ss = kp_mini(c.home_stage_index, n - 1);
    sample_stage_joint_target(s, r, s.stages[ss].start_q, s.stages[ss].start_noise, start_q);
  } else if (source == SRC_OLD_SUCCESS) {
    int max_old = kp_clipi(opt_or(c.old_success_max_stage_index, kp_mini(7, current)), 0, n - 1);
    int old_idx = pcg_integers(r, 0, max_old + 1);
    sample_stage_joint_target(s, r, s.stages[old_idx].goal_q, s.stages[old_idx].goal_noise, start_q);
  } else if (source == SRC_FRONTIER) {
    int fmin_ = kp_clipi(opt_or(c.frontier_min_stage_index, kp_mini(8, current)), 0, n - 1);
    int fmax_ = kp_clipi(opt_or(c.frontier_max_stage_index, current), fmin_, n - 1);
    int fi = pcg_integers(r, fmin_, fmax_ + 1);
    sample_stage_joint_target(s, r, s.stages[fi].start_q, s.stages[fi].start_noise, start_q);
  } else if (source == SRC_FAILURE_RECOVERY) {
    double d[NJ];
    pcg_uniform_sym7(r, c.failure_recovery_q_noise, d);
    for (int i = 0; i < NJ; ++i) start_q[i] = dclip(target_q[i] + d[i], s.lower[i], s.upper[i]);
  } else if (source == SRC_STRESS) {
    double margin = c.has_stress_start_margin_fraction ? c.stress_start_margin_fraction : s.start_sample_margin_fraction;
    sample_joint_configuration(s, r, margin, start_q);
  } else {
    double margin = c.has_random_valid_start_margin_fraction ? c.random_valid_start_margin_fraction : s.start_sample_margin_fraction;
    sample_joint_configuration(s, r, margin, start_q);
  }
  if (any_positive7(c.initial_dq_noise)) pcg_uniform_sym7(r, c.initial_dq_noise, o.initial_dq);
  else for (int i = 0; i < NJ; ++i) o.initial_dq[i] = 0.0;
  if (any_positive7(c.initial_prev_action_noise)) pcg_uniform_sym7(r, c.initial_prev_action_noise, o.initial_prev_action);
  else for (int i = 0; i < NJ; ++i) o.initial_prev_action[i] = 0.0;
  if (c.min_pair_joint_l2 > 0.0) {
    for (int k = 0; k < 12; ++k) {
      double ss = 0.0;
      for (int i = 0; i < NJ; ++i) ss += (target_q[i] - start_q[i]) * (target_q[i] - start_q[i]);
      if (sqrt(ss) >= c.min_pair_joint_l2) break;
      target_stage = sample_target_stage_for_source(s, r, source, current);
      sample_stage_joint_target(s, r, s.stages[target_stage].goal_q, s.stages[target_stage].goal_noise, target_q);
    }
  }
  for (int i = 0; i < NJ; ++i) {
    o.goal_q[i] = dclip(target_q[i], s.lower[i], s.upper[i]);
    o.initial_q[i] = dclip(start_q[i], s.lower[i], s.upper[i]);
  }
  o.stage = target_stage;
}

}  // namespace kp1
