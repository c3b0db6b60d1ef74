// kp1_env.hip -- kernels + C ABI (include/kp1.h) of the MI355X kinematic_phase1 rollout engine.
//
// Kernels (all: one lane = one env, SoA state, wave-uniform config through scalar loads):
//   kp1_step_kernel   action clip -> delta-q -> joint clip -> FK -> pose error -> zone counters ->
//                     termination -> reward -> state commit -> observation -> fused auto-reset
//                     (reference: KP1/envs/arm_kinematic_env.py:213-365 + callees, VecEnv auto-reset)
//   kp1_reset_kernel  reset(): explicit options or PCG64 sampling (arm_kinematic_env.py:102-211)
//   kp1_observe_kernel / kp1_fk_kernel / kp1_init_kernel / state gather-scatter helpers
//
// There is no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/kp1_ppo.h"
#include "kp1_device.hpp"
#include "kp1_host.hpp"

using namespace kp1;

// ============================================================================================
// device code
// ============================================================================================
namespace {

#include "kp1_env_step.inc"

template <typename R, int MODE, bool COMPS>
__global__ void __launch_bounds__(256, KP1_STEP_MIN_WAVES) kp1_step_kernel(const StepArgs<R> a) {
  extern __shared__ float obs_tiles[];   // OBS_TILE_FLOATS per wave of the workgroup
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#ifdef KP1_ENV_TRACE   // slots 11 / 12: the chip-wide 100 MHz clock at wave entry / exit (the shader clock of slots 0..10 is not comparable across the chip)
  if ((threadIdx.x & 63) == 0 && (i >> 6) < KP1_ENV_TRACE_WAVES) kp1_env_trace_buf[(i >> 6) * KP1_ENV_TRACE_SLOTS + 11] = __builtin_amdgcn_s_memrealtime();
#endif
  const bool live = i < a.st.n;
  float o[KP1_OBS_DIM];
  if (live) step_env_lane<R, MODE, COMPS>(a, i, a.actions + i * NJ, o);
  const int64_t i0 = i - (int64_t)(threadIdx.x & 63u);
  const int64_t left = a.st.n - i0;
  store_obs_tile(a.obs, i0, left >= 64 ? 64 : (left > 0 ? (int)left : 0), o, live, a.obs_stride, obs_tiles + (threadIdx.x >> 6) * OBS_TILE_FLOATS);
  KP1_ETR(8)
#ifdef KP1_ENV_TRACE
  __builtin_amdgcn_s_waitcnt(0);   // every store of this wave acknowledged
  KP1_ETR(9)
  if ((threadIdx.x & 63) == 0 && (i >> 6) < KP1_ENV_TRACE_WAVES) kp1_env_trace_buf[(i >> 6) * KP1_ENV_TRACE_SLOTS + 12] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <typename R, int MODE>
__global__ void __launch_bounds__(256) kp1_reset_kernel(const EnvState<R> st, const DevCfg<R>* cfg, const DevSampler* smp, const kp1_handoff_state* handoff,
                                 const uint8_t* mask, const ResetOptsDev opts, int stage_index, float* obs, int obs_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  if (mask && !mask[i]) return;
  float o[KP1_OBS_DIM];
  reset_env<R, MODE>(st, uniform_block(cfg), uniform_block(smp), handoff, opts, stage_index, i, o);
  if (obs) store_obs_row(obs, i, o, obs_stride);
}

template <typename R>
__global__ void kp1_observe_kernel(const EnvState<R> st, const DevCfg<R>* cfgp, int mode, float* obs, int obs_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  const DevCfg<R>& cfg = *cfgp;
  R q[NJ], dq[NJ], pa[NJ], goal[6], ee[6], pe[3], oe[3], pn, on;
  for (int k = 0; k < NJ; ++k) {
    q[k] = st.r(F_Q + k, i);
    dq[k] = st.r(F_DQ + k, i);
    pa[k] = st.r(F_PREV_ACTION + k, i);
  }
  for (int k = 0; k < 6; ++k) {
    goal[k] = st.r(F_GOAL_POSE + k, i);
    ee[k] = st.r(F_EE_POSE + k, i);
  }
  pose_error_norms<R>(ee, goal, pe, oe, &pn, &on);
  float o[KP1_OBS_DIM];
  build_observation<R>(cfg, mode, q, dq, pa, pe, oe, st.iv(I_STEP, i), st.iv(I_DWELL, i), o);
  store_obs_row(obs, i, o, obs_stride);
}

// __init__: zero state, ee_pose6 = FK(0); arm_kinematic_env.py:80-100
template <typename R>
__global__ void __launch_bounds__(256) kp1_init_kernel(const EnvState<R> st, const DevCfg<R>* cfg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  for (int f = 0; f < F_NUM_REAL; ++f) st.r(f, i) = (R)0;
  for (int f = 0; f < I_NUM_INT; ++f) st.iv(f, i) = 0;
  const double q[NJ] = {0, 0, 0, 0, 0, 0, 0};
  R ee[6];
  fk_pose6_kin<R>(cfg->kin.fk, q, ee);
  for (int k = 0; k < 6; ++k) st.r(F_EE_POSE + k, i) = ee[k];
  st.r(F_MIN_POS, i) = std::numeric_limits<R>::infinity();
}

template <typename R>
__global__ void kp1_fk_kernel(const DevFk<double>* fk, const R* q, R* pose, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double qq[NJ];
  R p[6];
  for (int k = 0; k < NJ; ++k) qq[k] = (double)q[i * NJ + k];
  fk_pose6_kin<R>(*fk, qq, p);
  for (int k = 0; k < 6; ++k) pose[i * 6 + k] = p[k];
}

template <typename R>
__global__ void kp1_pose_error_kernel(const R* curr, const R* goal, R* pos_err, R* ori_err, R* norms, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  R c[6], g[6], pe[3], oe[3], pn, on;
  for (int k = 0; k < 6; ++k) {
    c[k] = curr[i * 6 + k];
    g[k] = goal[i * 6 + k];
  }
  pose_error_norms<R>(c, g, pe, oe, &pn, &on);
  for (int k = 0; k < 3; ++k) {
    if (pos_err) pos_err[i * 3 + k] = pe[k];
    if (ori_err) ori_err[i * 3 + k] = oe[k];
  }
  if (norms) {
    norms[i * 2 + 0] = pn;
    norms[i * 2 + 1] = on;
  }
}

template <typename R>
struct JointLimitsDev { R lower[NJ], upper[NJ], dlim[NJ]; };

template <typename R>
__global__ void kp1_joint_utils_kernel(const JointLimitsDev<R> lim, const R* q, const R* dq, R* clipped, R* margin, R* q_norm, R* dq_norm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < NJ; ++k) {
    const R lo = lim.lower[k], hi = lim.upper[k];
    if (q) {
      const R c = joint_clip<R>(q[i * NJ + k], lo, hi);
      if (clipped) clipped[i * NJ + k] = c;
      if (margin) margin[i * NJ + k] = joint_limit_margin<R>(c, lo, hi);
      if (q_norm) q_norm[i * NJ + k] = joint_normalize_q<R>(q[i * NJ + k], lo, hi);
    }
    if (dq && dq_norm) dq_norm[i * NJ + k] = joint_normalize_dq<R>(dq[i * NJ + k], lim.dlim[k]);
  }
}

// set_state: scatter row-major fp64 host-provided rows into the SoA state, recompute ee = FK(q)
template <typename R>
__global__ void kp1_set_state_kernel(const EnvState<R> st, const DevCfg<R>* cfg, const double* q, const double* dq, const double* pa,
                                     const double* goal_q, const double* goal_pose, int capture_entry) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  if (q) {
    double qq[NJ];
    R ee[6];
    for (int k = 0; k < NJ; ++k) {
      qq[k] = q[i * NJ + k];
      st.q_store(k, i, qq[k]);
    }
    fk_pose6_kin<R>(cfg->kin.fk, qq, ee);
    for (int k = 0; k < 6; ++k) st.r(F_EE_POSE + k, i) = ee[k];
  }
  if (dq) for (int k = 0; k < NJ; ++k) st.r(F_DQ + k, i) = (R)dq[i * NJ + k];
  if (pa) for (int k = 0; k < NJ; ++k) st.r(F_PREV_ACTION + k, i) = (R)pa[i * NJ + k];
  if (goal_q) for (int k = 0; k < NJ; ++k) st.r(F_GOAL_Q + k, i) = (R)goal_q[i * NJ + k];
  if (goal_pose) for (int k = 0; k < 6; ++k) st.r(F_GOAL_POSE + k, i) = (R)goal_pose[i * 6 + k];
  if (capture_entry) {
    R ee[6], goal[6], pe[3], oe[3], pn, on, v[NJ];
    for (int k = 0; k < 6; ++k) {
      ee[k] = st.r(F_EE_POSE + k, i);
      goal[k] = st.r(F_GOAL_POSE + k, i);
    }
    pose_error_norms<R>(ee, goal, pe, oe, &pn, &on);
    st.r(F_ENTRY + 0, i) = pn;
    st.r(F_ENTRY + 1, i) = on;
    for (int k = 0; k < NJ; ++k) v[k] = st.r(F_PREV_ACTION + k, i);
    st.r(F_ENTRY + 2, i) = norm7<R>(v);
    for (int k = 0; k < NJ; ++k) v[k] = st.r(F_DQ + k, i);
    st.r(F_ENTRY + 3, i) = norm7<R>(v);
    st.r(F_POS_ERR, i) = pn;
    st.r(F_ORI_ERR, i) = on;
  }
}

template <typename R>
__global__ void kp1_get_state_kernel(const EnvState<R> st, double* q, double* dq, double* pa, double* goal_q, double* goal_pose) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  for (int k = 0; k < NJ; ++k) {
    if (q) q[i * NJ + k] = st.q_load(k, i);
    if (dq) dq[i * NJ + k] = (double)st.r(F_DQ + k, i);
    if (pa) pa[i * NJ + k] = (double)st.r(F_PREV_ACTION + k, i);
    if (goal_q) goal_q[i * NJ + k] = (double)st.r(F_GOAL_Q + k, i);
  }
  if (goal_pose) for (int k = 0; k < 6; ++k) goal_pose[i * 6 + k] = (double)st.r(F_GOAL_POSE + k, i);
}

}  // namespace

// ============================================================================================
// host side
// ============================================================================================
namespace {

// ---- numpy SeedSequence + PCG64 seeding (numpy/random/bit_generator.pyx, src/pcg64/pcg64.h) ----
uint32_t ss_hashmix(uint32_t value, uint32_t& hash_const) {
  value ^= hash_const;
  hash_const *= 0x931e8875u;
  value *= hash_const;
  value ^= value >> 16;
  return value;
}
uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
  r ^= r >> 16;
  return r;
}
void pcg64_seed(uint64_t seed, kp1_rng_state* out) {
  uint32_t entropy[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  const int n_ent = entropy[1] != 0 ? 2 : 1;
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;
  for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n_ent ? entropy[i] : 0u, hc);
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 4; ++d)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
  uint32_t w[8];
  uint32_t hb = 0x8b51f9ddu;
  for (int i = 0; i < 8; ++i) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= 0x58f38dedu;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t s64[4];
  for (int i = 0; i < 4; ++i) s64[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  const unsigned __int128 MULT = (((unsigned __int128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;
  unsigned __int128 initstate = ((unsigned __int128)s64[0] << 64) | s64[1];
  unsigned __int128 initseq = ((unsigned __int128)s64[2] << 64) | s64[3];
  unsigned __int128 inc = (initseq << 1) | 1u;
  unsigned __int128 state = 0;
  state = state * MULT + inc;
  state += initstate;
  state = state * MULT + inc;
  out->state_hi = (uint64_t)(state >> 64);
  out->state_lo = (uint64_t)state;
  out->inc_hi = (uint64_t)(inc >> 64);
  out->inc_lo = (uint64_t)inc;
  out->has_uint32 = 0;
  out->uinteger = 0;
}

// ---- FK constant folding in fp64 (V51/ee_fk.py:14-95) ----
const double ORIGIN_XYZ[7][3] = {
    {0.00715921043213119, 0.0000809621375843506, -0.0635},
    {-0.021178, 0.0, 0.1868},
    {-0.0633967414837172, 0.000642782425827271, 0.0602000000000009},
    {-0.000134989688424625, 0.425, 0.0133123982251372},
    {-0.0000850456535865796, -0.39225, -0.0083864861805065},
    {0.0475482889721905, -0.000817137634885778, -0.0805958577476871},
    {0.0436977540622506, 0.000443046177049933, -0.0521517110277254}};
const double ORIGIN_RPY[7][3] = {{0, 0, 0},
                                 {0, 0, 0},
                                 {1.5707963267949, 0.0, 1.5707963267949},
                                 {3.14159265358979, 0.0, 0.0},
                                 {3.14159265358979, 0.0, -1.5707963267949},
                                 {3.14159265358979, 1.5707963267949, 0.0},
                                 {-1.5707963267949, 0.0, -1.5707963267949}};
const double AXES_LOCAL[7][3] = {{1.0, 0.0, 0.0},
                                 {0.0, 0.0, 1.0},
                                 {0.0101382310641698, 0.0, -0.999948606814815},
                                 {0.010138231064165, 0.0, 0.999948606814815},
                                 {0.0, -0.0101382310641647, -0.999948606814815},
                                 {0.0, 0.0, -1.0},
                                 {-0.0101384515502096, 0.0, 0.999948604579338}};

void mat3_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j];
      c[3 * i + j] = s;
    }
}
void rpy_to_rot(const double* rpy, double* R) {
  double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]), cp = std::cos(rpy[1]), sp = std::sin(rpy[1]), cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  double rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
  double ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
  double rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
  double t[9];
  mat3_mul(rz, ry, t);
  mat3_mul(t, rx, R);
}
template <typename R>
void fold_fk(DevFk<R>* out) {
  double RA[7][9];
  for (int i = 0; i < 7; ++i) rpy_to_rot(ORIGIN_RPY[i], RA[i]);
  // prismatic joint 0: p = p0 + RA0 (a0 q0); then origin of joint 1: p += RA0 p1
  double a0n = std::sqrt(AXES_LOCAL[0][0] * AXES_LOCAL[0][0] + AXES_LOCAL[0][1] * AXES_LOCAL[0][1] + AXES_LOCAL[0][2] * AXES_LOCAL[0][2]);
  (void)a0n;  // the reference does NOT normalise the prismatic axis (ee_fk.py:113)
  for (int r = 0; r < 3; ++r) {
    double v = 0, pp = ORIGIN_XYZ[0][r];
    for (int k = 0; k < 3; ++k) {
      v += RA[0][3 * r + k] * AXES_LOCAL[0][k];
      pp += RA[0][3 * r + k] * ORIGIN_XYZ[1][k];
    }
    out->v0[r] = (R)v;
    out->p01[r] = (R)pp;
  }
  for (int j = 1; j < 7; ++j) {
    const double* ax = AXES_LOCAL[j];
    double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-12;  // ee_fk.py:76
    double x = ax[0] / nrm, y = ax[1] / nrm, z = ax[2] / nrm;
    double aat[9] = {x * x, x * y, x * z, y * x, y * y, y * z, z * x, z * y, z * z};
    double skew[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    double eye_m[9];
    for (int e = 0; e < 9; ++e) eye_m[e] = ((e % 4 == 0) ? 1.0 : 0.0) - aat[e];
    double base[9];
    if (j == 1) mat3_mul(RA[0], RA[1], base);  // fold the (rotation of the) prismatic stage into joint 1
    else std::memcpy(base, RA[j], sizeof base);
    double k1[9], kc[9], ks[9];
    mat3_mul(base, aat, k1);
    mat3_mul(base, eye_m, kc);
    mat3_mul(base, skew, ks);
    for (int e = 0; e < 9; ++e) {
      out->k1[j - 1][e] = (R)k1[e];
      out->kc[j - 1][e] = (R)kc[e];
      out->ks[j - 1][e] = (R)ks[e];
    }
    if (j >= 2)
      for (int r = 0; r < 3; ++r) out->p[j - 2][r] = (R)ORIGIN_XYZ[j][r];
  }
}

#define KP1_CONV(type, name, dflt) d.name = (decltype(d.name))s.name;
template <typename R>
void make_dev_cfg(const kp1_config& c, DevCfg<R>* out) {
  std::memset(out, 0, sizeof *out);
  { auto& d = out->env; const auto& s = c.env; KP1_ENV_FIELDS(KP1_CONV) }
  { auto& d = out->reward; const auto& s = c.reward; KP1_APPROACH_REWARD_FIELDS(KP1_CONV) }
  { auto& d = out->dock; const auto& s = c.dock_reward; KP1_DOCK_REWARD_FIELDS(KP1_CONV) }
  { auto& d = out->term; const auto& s = c.termination; KP1_TERMINATION_FIELDS(KP1_CONV) }
  { auto& d = out->obs; const auto& s = c.observation; KP1_OBSERVATION_FIELDS(KP1_CONV) }
  for (int i = 0; i < KP1_MAX_MILESTONES; ++i) {
    out->reward.ms_thr[i] = (R)c.reward.orientation_milestone_thresholds_rad[i];
    out->reward.ms_bonus[i] = (R)c.reward.orientation_milestone_bonuses[i];
  }
  for (int i = 0; i < NJ; ++i) {
    out->lower[i] = (R)c.joints.lower[i];
    out->upper[i] = (R)c.joints.upper[i];
    out->dlim[i] = (R)c.joints.delta_limit[i];
    out->kin.lower[i] = c.joints.lower[i];
    out->kin.upper[i] = c.joints.upper[i];
    out->kin.dlim[i] = c.joints.delta_limit[i];
  }
  out->kin.action_delta_scale = c.env.action_delta_scale;
  out->kin.dock_action_delta_scale = c.env.dock_action_delta_scale;
  fold_fk<double>(&out->kin.fk);
}
void make_dev_sampler(const kp1_config& c, int n_handoff, DevSampler* s) {
  std::memset(s, 0, sizeof *s);
  for (int i = 0; i < NJ; ++i) {
    s->lower[i] = c.joints.lower[i];
    s->upper[i] = c.joints.upper[i];
  }
  s->curriculum_enabled = c.curriculum_enabled;
  s->n_stages = c.n_stages;
  std::memcpy(s->stages, c.stages, sizeof c.stages);
  s->ss = c.stage_sampling;
  s->rs = c.random_start;
  s->dr = c.dock_reset;
  s->start_sample_margin_fraction = c.env.start_sample_margin_fraction;
  s->goal_sample_margin_fraction = c.env.goal_sample_margin_fraction;
  s->n_handoff = n_handoff;
  s->handoff_offset = 0;
  fold_fk<double>(&s->fk);
}

#define KP1_SETD(type, name, dflt) s.name = dflt;
void config_default(kp1_config* cfg) {
  std::memset(cfg, 0, sizeof *cfg);
  { auto& s = cfg->env; KP1_ENV_FIELDS(KP1_SETD) }
  { auto& s = cfg->reward; KP1_APPROACH_REWARD_FIELDS(KP1_SETD) }
  { auto& s = cfg->dock_reward; KP1_DOCK_REWARD_FIELDS(KP1_SETD) }
  { auto& s = cfg->termination; KP1_TERMINATION_FIELDS(KP1_SETD) }
  { auto& s = cfg->observation; KP1_OBSERVATION_FIELDS(KP1_SETD) }
  { auto& s = cfg->stage_sampling; KP1_STAGE_SAMPLING_FIELDS(KP1_SETD) }
  { auto& s = cfg->random_start; KP1_RANDOM_START_FIELDS(KP1_SETD) }
  { auto& s = cfg->dock_reset; KP1_DOCK_RESET_FIELDS(KP1_SETD) }
  const double PI = 3.141592653589793;
  const double lim[NJ] = {0.385, PI, PI, PI, PI, PI, PI};              // KP1/kinematics/joint_limits.py:37-47
  const double dl[NJ] = {0.08, 0.30, 0.24, 0.24, 0.30, 0.40, 0.30};
  for (int i = 0; i < NJ; ++i) {
    cfg->joints.lower[i] = -lim[i];
    cfg->joints.upper[i] = lim[i];
    cfg->joints.delta_limit[i] = dl[i];
    cfg->random_start.failure_recovery_q_noise[i] = 0.04;
  }
  const double gn[NJ] = {0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01};   // KP1/envs/reset_samplers.py:50-54
  const double iq[NJ] = {0.01, 0.02, 0.03, 0.02, 0.015, 0.015, 0.01};
  const double cq[NJ] = {0.006, 0.012, 0.018, 0.012, 0.009, 0.009, 0.006};
  std::memcpy(cfg->dock_reset.goal_noise, gn, sizeof gn);
  std::memcpy(cfg->dock_reset.init_q_noise, iq, sizeof iq);
  std::memcpy(cfg->dock_reset.close_init_q_noise, cq, sizeof cq);
  const double goal_noise[6][NJ] = {{0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01}, {0.02, 0.06, 0.08, 0.06, 0.04, 0.04, 0.03},
                                    {0.03, 0.09, 0.12, 0.09, 0.06, 0.05, 0.04}, {0.04, 0.12, 0.16, 0.12, 0.08, 0.06, 0.05},
                                    {0.05, 0.14, 0.18, 0.14, 0.09, 0.07, 0.06}, {0.06, 0.18, 0.22, 0.16, 0.10, 0.08, 0.07}};
  const double start_noise[6] = {0.0, 0.0, 0.0, 0.01, 0.02, 0.03};      // KP1/envs/curriculum.py:36-78
  const double goal4[NJ] = {0.03, -0.04, 0.05, -0.03, 0.02, -0.01, 0.01};
  cfg->curriculum_enabled = 1;
  cfg->n_stages = 6;
  for (int k = 0; k < 6; ++k) {
    std::memcpy(cfg->stages[k].goal_noise, goal_noise[k], sizeof goal_noise[k]);
    for (int i = 1; i < NJ; ++i) cfg->stages[k].start_noise[i] = start_noise[k];
  }
  std::memcpy(cfg->stages[4].goal_q, goal4, sizeof goal4);
}

}  // namespace

struct kp1_env {
  kp1_config cfg;
  int32_t n = 0, device = 0, real_type = 0, stage = 0, mode = 0;
  hipStream_t stream = nullptr;
  void* real = nullptr;        // R[F_NUM_REAL][N]
  int32_t* ints = nullptr;     // [I_NUM_INT][N]
  uint64_t* rng64 = nullptr;   // [4][N]
  uint32_t* rng32 = nullptr;   // [2][N]
  void* dev_cfg = nullptr;     // DevCfg<R>
  DevSampler* dev_smp = nullptr;
  kp1_handoff_state* dev_handoff = nullptr;
  int32_t n_handoff = 0;
  void* snapshot = nullptr;    // kp1_state_snapshot: real | ints | rng64 | rng32, allocated on first use
  kp1_dock_curriculum_state* dock_tracker = nullptr;   // device tracker attached by kp1_dock_curriculum_create (its stage survives upload_cfg)
  double* opt_scratch = nullptr;  // 4*[N][7] + [N][6] doubles for explicit reset options / set_state
  const int32_t* stage_ptr = nullptr;  // kp1_bind_stage_ptr
  int32_t obs_stride = KP1_OBS_DIM;    // kp1_set_obs_stride
  void* comps = nullptr;          // R[64][N] when enabled
  bool comps_enabled = false;
  size_t real_size() const { return real_type == KP1_REAL_F64 ? 8 : 4; }
};

namespace {

// One wave per workgroup at every size: the step kernel holds 500 vector registers, so a CU runs four waves (one per SIMD), and a 256-thread
// workgroup can only be replaced when all four of its waves have finished.  64-thread workgroups let every SIMD take its next wave by itself:
// 117.5 -> 99.1 us at 524 288 envs, 33.8 -> 31.3 us at 131 072 (profiles/r03_ab_env_block_size.log; KP1_BIG_BLOCK is the A/B switch).
#ifndef KP1_BIG_BLOCK
#define KP1_BIG_BLOCK 64
#endif
int block_for(int64_t n) { return n <= 65536 ? 64 : KP1_BIG_BLOCK; }


template <typename R>
EnvState<R> state_of(const kp1_env* e) {
  EnvState<R> st;
  st.real = (R*)e->real;
  st.ints = e->ints;
  st.rng64 = e->rng64;
  st.rng32 = e->rng32;
  st.n = e->n;
  return st;
}

int reapply_dock_stage(kp1_env* env);   // kp1_dock_curriculum.inc

int upload_cfg(kp1_env* e) {
  if (e->real_type == KP1_REAL_F64) {
    DevCfg<double> d;
    make_dev_cfg<double>(e->cfg, &d);
    HIP_TRY(hipMemcpyAsync(e->dev_cfg, &d, sizeof d, hipMemcpyHostToDevice, e->stream));
  } else {
    DevCfg<float> d;
    make_dev_cfg<float>(e->cfg, &d);
    HIP_TRY(hipMemcpyAsync(e->dev_cfg, &d, sizeof d, hipMemcpyHostToDevice, e->stream));
  }
  DevSampler s;
  make_dev_sampler(e->cfg, e->n_handoff, &s);
  HIP_TRY(hipMemcpyAsync(e->dev_smp, &s, sizeof s, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));  // d / s are stack objects
  return reapply_dock_stage(e);
}

int seed_streams(kp1_env* e, uint64_t seed0, uint64_t first_env_id) {
  const int64_t n = e->n;
  std::vector<uint64_t> r64(4 * (size_t)n);
  std::vector<uint32_t> r32(2 * (size_t)n, 0u);
  for (int64_t i = 0; i < n; ++i) {
    kp1_rng_state s;
    pcg64_seed(seed0 + first_env_id + (uint64_t)i, &s);
    r64[0 * n + i] = s.state_hi;
    r64[1 * n + i] = s.state_lo;
    r64[2 * n + i] = s.inc_hi;
    r64[3 * n + i] = s.inc_lo;
  }
  HIP_TRY(hipMemcpyAsync(e->rng64, r64.data(), r64.size() * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->rng32, r32.data(), r32.size() * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return KP1_OK;
}

template <typename R>
StepArgs<R> make_step_args(kp1_env* e, const void* actions, float* obs, void* reward, uint8_t* done, float* terminal_obs, int auto_reset) {
  StepArgs<R> a;
  a.st = state_of<R>(e);
  a.cfg = (const DevCfg<R>*)e->dev_cfg;
  a.smp = e->dev_smp;
  a.handoff = e->dev_handoff;
  a.actions = (const R*)actions;
  a.obs = obs;
  a.reward = (R*)reward;
  a.done = done;
  a.terminal_obs = terminal_obs;
  a.comps = (R*)e->comps;
  a.auto_reset = auto_reset;
  a.stage_index = e->stage;
  a.obs_stride = e->obs_stride;
  a.stage_ptr = e->cfg.curriculum_enabled ? e->stage_ptr : nullptr;
  return a;
}

template <typename R>
int launch_step(kp1_env* e, const void* actions, float* obs, void* reward, uint8_t* done, float* terminal_obs, int auto_reset) {
  const StepArgs<R> a = make_step_args<R>(e, actions, obs, reward, done, terminal_obs, auto_reset);
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  const bool comps = e->comps_enabled && e->comps;
  const size_t lds = sizeof(float) * OBS_TILE_FLOATS * (size_t)(block / 64);   // one observation tile per wave (store_obs_tile)
  if (e->mode == KP1_MODE_DOCK) {
    if (comps) hipLaunchKernelGGL((kp1_step_kernel<R, KP1_MODE_DOCK, true>), grid, dim3(block), lds, e->stream, a);
    else hipLaunchKernelGGL((kp1_step_kernel<R, KP1_MODE_DOCK, false>), grid, dim3(block), lds, e->stream, a);
  } else {
    if (comps) hipLaunchKernelGGL((kp1_step_kernel<R, KP1_MODE_APPROACH, true>), grid, dim3(block), lds, e->stream, a);
    else hipLaunchKernelGGL((kp1_step_kernel<R, KP1_MODE_APPROACH, false>), grid, dim3(block), lds, e->stream, a);
  }
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

template <typename R>
int launch_reset(kp1_env* e, const uint8_t* mask, const ResetOptsDev& opts, int mode, float* obs) {
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  if (mode == KP1_MODE_DOCK)
    hipLaunchKernelGGL((kp1_reset_kernel<R, KP1_MODE_DOCK>), grid, dim3(block), 0, e->stream, state_of<R>(e),
                       (const DevCfg<R>*)e->dev_cfg, e->dev_smp, e->dev_handoff, mask, opts, e->stage, obs, e->obs_stride);
  else
    hipLaunchKernelGGL((kp1_reset_kernel<R, KP1_MODE_APPROACH>), grid, dim3(block), 0, e->stream, state_of<R>(e),
                       (const DevCfg<R>*)e->dev_cfg, e->dev_smp, e->dev_handoff, mask, opts, e->stage, obs, e->obs_stride);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

// ---------------------------------------------------------------------------------------------- evaluator bookkeeping (kp1_eval_accumulate)
// One lane per episode.  STATE_W = the 34 leading real fields of the handle (q, dq, prev_action, goal_q, goal_pose6: F_Q .. F_GOAL_POSE + 5).
constexpr int EVAL_STATE_W = F_GOAL_POSE + 6;
static_assert(EVAL_STATE_W == 34 && F_Q == 0, "kp1_eval_buffers::state is the leading 34 fields of the handle");
template <typename R>
__global__ void __launch_bounds__(256) eval_accumulate_kernel(const R* __restrict__ real, int n, kp1_eval_buffers b, const double* __restrict__ action_norm,
                                                              const uint8_t* __restrict__ done, const uint8_t* __restrict__ active, int step, bool track_ready,
                                                              double thr_pos, double thr_ori, double thr_act, double thr_dq, int confirm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int alive_after = 0;
  if (i < n) {
    double* M = b.metrics;
    int32_t* C = b.counters;
    uint8_t* F = b.flags;
    const double pos = (double)real[(size_t)F_POS_ERR * n + i], ori = (double)real[(size_t)F_ORI_ERR * n + i];
    if (step == 0) {
      M[0 * (size_t)n + i] = pos; M[1 * (size_t)n + i] = ori; M[2 * (size_t)n + i] = pos; M[3 * (size_t)n + i] = ori;
      for (int k = 4; k < 8; ++k) M[k * (size_t)n + i] = 0.0;
      C[0 * (size_t)n + i] = 0; C[1 * (size_t)n + i] = 0; C[2 * (size_t)n + i] = -1; C[3 * (size_t)n + i] = 0;
      const uint8_t a = active ? (active[i] ? 1 : 0) : 1;
      F[0 * (size_t)n + i] = a; F[1 * (size_t)n + i] = 0; F[2 * (size_t)n + i] = 0; F[3 * (size_t)n + i] = 0;
      for (int f = 0; f < EVAL_STATE_W; ++f) b.state[(size_t)i * EVAL_STATE_W + f] = (double)real[(size_t)f * n + i];
      if (b.hand_metrics) {
        for (int k = 0; k < 8; ++k) b.hand_metrics[k * (size_t)n + i] = 0.0;
        b.hand_step[i] = 0;
        b.hand_success[i] = 0;
        for (int f = 0; f < EVAL_STATE_W; ++f) b.hand_state[(size_t)i * EVAL_STATE_W + f] = 0.0;
      }
      alive_after = a;
    } else if (F[i]) {
      const double an = action_norm[i], dqn = (double)real[(size_t)F_EXEC_DQ * n + i];
      const uint8_t d = done[i];
      const uint8_t succ = (d & KP1_DONE_SUCCESS) ? 1 : 0;
      C[i] = step;
      M[6 * (size_t)n + i] += an;
      M[7 * (size_t)n + i] += dqn;
      M[0 * (size_t)n + i] = pos; M[1 * (size_t)n + i] = ori; M[4 * (size_t)n + i] = an; M[5 * (size_t)n + i] = dqn;
      const double mp = fmin(M[2 * (size_t)n + i], pos), mo = fmin(M[3 * (size_t)n + i], ori);
      M[2 * (size_t)n + i] = mp; M[3 * (size_t)n + i] = mo;
      F[1 * (size_t)n + i] = succ;
      double st[EVAL_STATE_W];
      for (int f = 0; f < EVAL_STATE_W; ++f) {
        st[f] = (double)real[(size_t)f * n + i];
        b.state[(size_t)i * EVAL_STATE_W + f] = st[f];
      }
      if (track_ready) {
        bool rdy = thr_pos > 0.0 && thr_ori > 0.0 && pos <= thr_pos && ori <= thr_ori;
        if (thr_act > 0.0) rdy = rdy && an <= thr_act;
        if (thr_dq > 0.0) rdy = rdy && dqn <= thr_dq;
        if (rdy) {
          F[2 * (size_t)n + i] = 1;
          if (C[2 * (size_t)n + i] < 0) C[2 * (size_t)n + i] = step;
        }
        const int streak = rdy ? C[3 * (size_t)n + i] + 1 : 0;
        C[3 * (size_t)n + i] = streak;
        if (streak > C[1 * (size_t)n + i]) C[1 * (size_t)n + i] = streak;
        // first-confirmed handoff snapshot.  `ready_streak >= handoff_confirm_steps` as the reference writes it (eval_pipeline_ablation.py:103):
        // with confirm <= 0 it holds at step 1 whatever the streak.  Whether a snapshot is wanted at all is "hand_metrics given".
        if (b.hand_metrics && !F[3 * (size_t)n + i] && streak >= confirm) {
          F[3 * (size_t)n + i] = 1;
          double* H = b.hand_metrics;
          H[0 * (size_t)n + i] = pos; H[1 * (size_t)n + i] = ori; H[2 * (size_t)n + i] = an; H[3 * (size_t)n + i] = dqn;
          H[4 * (size_t)n + i] = mp; H[5 * (size_t)n + i] = mo;
          H[6 * (size_t)n + i] = M[6 * (size_t)n + i]; H[7 * (size_t)n + i] = M[7 * (size_t)n + i];   // sums up to and including this step (:111-112)
          b.hand_step[i] = step;
          b.hand_success[i] = succ;
          for (int f = 0; f < EVAL_STATE_W; ++f) b.hand_state[(size_t)i * EVAL_STATE_W + f] = st[f];
        }
      }
      const uint8_t still = (d & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) ? 0 : 1;
      F[i] = still;
      alive_after = still;
    }
  }
  // episodes still alive: wave ballot, one atomic per wave (an integer count: order does not matter)
  const unsigned long long bal = __ballot(alive_after != 0);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(b.n_alive, __popcll(bal));
}

int ensure_scratch(kp1_env* e) {
  if (!e->opt_scratch) HIP_TRY(hipMalloc((void**)&e->opt_scratch, sizeof(double) * (size_t)e->n * (4 * NJ + 6)));
  return KP1_OK;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
namespace kp1 {
int env_step_args_f32(kp1_env* e, void* out, size_t out_bytes, const void* actions, float* obs, void* reward, uint8_t* done, float* terminal_obs,
                      int auto_reset, int* mode, int64_t* n_envs, int* device) {
  if (!e || !out) return fail(KP1_ERR_INVALID, "NULL argument");
  if (e->real_type != KP1_REAL_F32) return fail(KP1_ERR_UNSUPPORTED, "the fused policy + env step runs on the fp32 handle");
  if (e->comps_enabled && e->comps) return fail(KP1_ERR_UNSUPPORTED, "the fused policy + env step does not record reward components");
  if (out_bytes != sizeof(StepArgs<float>)) return fail(KP1_ERR_INVALID, "StepArgs<float> layout mismatch between translation units");
  const StepArgs<float> a = make_step_args<float>(e, actions, obs, reward, done, terminal_obs, auto_reset);
  std::memcpy(out, &a, sizeof a);
  *mode = e->mode;
  *n_envs = e->n;
  *device = e->device;
  return KP1_OK;
}
}  // namespace kp1

extern "C" {

const char* kp1_last_error(void) { return g_last_error.c_str(); }
int kp1_abi_version(void) { return 1; }
uint64_t kp1_config_size(void) { return sizeof(kp1_config); }
int kp1_config_default(kp1_config* cfg) {
  if (!cfg) return fail(KP1_ERR_INVALID, "cfg is NULL");
  config_default(cfg);
  return KP1_OK;
}
int kp1_rng_seed_state(uint64_t seed, kp1_rng_state* out) {
  if (!out) return fail(KP1_ERR_INVALID, "out is NULL");
  pcg64_seed(seed, out);
  return KP1_OK;
}

int kp1_create(const kp1_config* cfg, int32_t n_envs, int32_t device, int32_t real_type, uint64_t seed0, uint64_t first_env_id,
               void* stream, kp1_env** out) {
  if (!cfg || !out) return fail(KP1_ERR_INVALID, "cfg/out is NULL");
  if (n_envs <= 0) return fail(KP1_ERR_INVALID, "n_envs must be positive");
  if (n_envs > (1 << 22)) return fail(KP1_ERR_INVALID, "n_envs above 2^22: a lane's byte offset inside the state arrays ((field * n + env) * 8) must fit 32 bits");
  if (real_type != KP1_REAL_F32 && real_type != KP1_REAL_F64) return fail(KP1_ERR_INVALID, "real_type must be KP1_REAL_F32 or KP1_REAL_F64");
  if (cfg->env.mode != KP1_MODE_APPROACH && cfg->env.mode != KP1_MODE_DOCK) return fail(KP1_ERR_UNSUPPORTED, "mode must be approach or dock");
  if (cfg->n_stages < 0 || cfg->n_stages > KP1_MAX_STAGES) return fail(KP1_ERR_INVALID, "n_stages out of range");
  if (cfg->curriculum_enabled && cfg->n_stages == 0) return fail(KP1_ERR_INVALID, "curriculum enabled without stages");
  if (cfg->reward.n_orientation_milestones < 0 || cfg->reward.n_orientation_milestones > KP1_MAX_MILESTONES)
    return fail(KP1_ERR_INVALID, "n_orientation_milestones out of range");
  {   // the fp32 handle's generated FK chain carries the robot constants as literals: they must be what fold_fk computes here, bit for bit
    DevFk<double> fk;
    fold_fk<double>(&fk);
    static_assert(sizeof(DevFk<double>) == sizeof(FKG_CHECK), "FKG_CHECK lists DevFk<double> member by member");
    if (std::memcmp(&fk, FKG_CHECK, sizeof fk) != 0)
      return fail(KP1_ERR_UNSUPPORTED, "csrc/kp1_fk_generated.inc does not match the robot constants of kp1_env.hip: run python3 tools/gen_fk_chain.py and rebuild");
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= count) return fail(KP1_ERR_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  kp1_env* e = new kp1_env();
  e->cfg = *cfg;
  e->n = n_envs;
  e->device = device;
  e->real_type = real_type;
  e->stream = (hipStream_t)stream;
  e->mode = cfg->env.mode;
  e->stage = 0;
  const size_t n = (size_t)n_envs;
  auto cleanup = [&]() { kp1_destroy(e); };
#define KP1_ALLOC(ptr, bytes)                                   \
  if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) {       \
    cleanup();                                                  \
    return fail(KP1_ERR_ALLOC, "hipMalloc failed for " #ptr);   \
  }
  KP1_ALLOC(e->real, e->real_size() * F_NUM_REAL * n);
  KP1_ALLOC(e->ints, sizeof(int32_t) * I_NUM_INT * n);
  KP1_ALLOC(e->rng64, sizeof(uint64_t) * 4 * n);
  KP1_ALLOC(e->rng32, sizeof(uint32_t) * 2 * n);
  KP1_ALLOC(e->dev_cfg, ((real_type == KP1_REAL_F64 ? sizeof(DevCfg<double>) : sizeof(DevCfg<float>)) + KP1_WARM_BYTES - 1) / KP1_WARM_BYTES * KP1_WARM_BYTES);
  KP1_ALLOC(e->dev_smp, sizeof(DevSampler));
#undef KP1_ALLOC
  int rc = upload_cfg(e);
  if (rc == KP1_OK) rc = seed_streams(e, seed0, first_env_id);
  if (rc != KP1_OK) {
    cleanup();
    return rc;
  }
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  if (real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(kp1_init_kernel<double>, grid, dim3(block), 0, e->stream, state_of<double>(e), (const DevCfg<double>*)e->dev_cfg);
  else
    hipLaunchKernelGGL(kp1_init_kernel<float>, grid, dim3(block), 0, e->stream, state_of<float>(e), (const DevCfg<float>*)e->dev_cfg);
  if (hipGetLastError() != hipSuccess) {
    cleanup();
    return fail(KP1_ERR_NO_DEVICE, "init kernel launch failed");
  }
  *out = e;
  return KP1_OK;
}

int kp1_destroy(kp1_env* e) {
  if (!e) return KP1_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  else (void)hipDeviceSynchronize();
  for (void* p : {(void*)e->real, (void*)e->ints, (void*)e->rng64, (void*)e->rng32, e->dev_cfg, (void*)e->dev_smp, (void*)e->dev_handoff,
                  (void*)e->opt_scratch, e->comps, e->snapshot})
    (void)hipFree(p);
  delete e;
  return KP1_OK;
}

int kp1_num_envs(const kp1_env* e) { return e ? e->n : 0; }

int kp1_set_stage(kp1_env* e, int32_t stage_index) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  if (!e->cfg.curriculum_enabled) return KP1_OK;  // arm_kinematic_env.py:447-448
  int hi = e->cfg.n_stages - 1;
  e->stage = stage_index < 0 ? 0 : (stage_index > hi ? hi : stage_index);
  return KP1_OK;
}
int kp1_bind_stage_ptr(kp1_env* e, const int32_t* stage_dev) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  e->stage_ptr = stage_dev;
  return KP1_OK;
}
int kp1_set_stream(kp1_env* e, void* stream) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  e->stream = (hipStream_t)stream;
  return KP1_OK;
}
int kp1_set_obs_stride(kp1_env* e, int32_t stride) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  if (stride != KP1_OBS_DIM && stride != 64) return fail(KP1_ERR_INVALID, "obs stride must be 56 or 64");
  e->obs_stride = stride;
  return KP1_OK;
}
int kp1_get_stage(const kp1_env* e, int32_t* stage_index) {
  if (!e || !stage_index) return fail(KP1_ERR_INVALID, "NULL argument");
  *stage_index = e->stage;
  return KP1_OK;
}
int kp1_set_mode(kp1_env* e, int32_t mode) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  if (mode != KP1_MODE_APPROACH && mode != KP1_MODE_DOCK) return fail(KP1_ERR_INVALID, "Unsupported policy mode");
  e->mode = mode;
  return KP1_OK;
}
int kp1_update_config(kp1_env* e, const kp1_config* cfg) {
  if (!e || !cfg) return fail(KP1_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  e->cfg.env = cfg->env;
  e->cfg.dock_reset = cfg->dock_reset;
  return upload_cfg(e);
}
int kp1_set_handoff_states(kp1_env* e, const kp1_handoff_state* states, int32_t n_states) {
  if (!e || (n_states > 0 && !states) || n_states < 0) return fail(KP1_ERR_INVALID, "bad handoff states");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->dev_handoff);
  e->dev_handoff = nullptr;
  e->n_handoff = n_states;
  if (n_states > 0) {
    HIP_TRY(hipMalloc((void**)&e->dev_handoff, sizeof(kp1_handoff_state) * (size_t)n_states));
    HIP_TRY(hipMemcpy(e->dev_handoff, states, sizeof(kp1_handoff_state) * (size_t)n_states, hipMemcpyHostToDevice));
  }
  return upload_cfg(e);
}
int kp1_seed(kp1_env* e, uint64_t seed0, uint64_t first_env_id) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  return seed_streams(e, seed0, first_env_id);
}

int kp1_reset(kp1_env* e, const uint8_t* mask_dev, const kp1_reset_opts* opts, float* obs_dev) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  ResetOptsDev d = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  int mode = e->cfg.env.mode;  // reset() falls back to config.mode_name; arm_kinematic_env.py:111
  if (opts) {
    if (opts->policy_mode >= 0) {
      if (opts->policy_mode != KP1_MODE_APPROACH && opts->policy_mode != KP1_MODE_DOCK) return fail(KP1_ERR_INVALID, "Unsupported policy mode");
      mode = opts->policy_mode;
    }
    const bool any = opts->initial_q || opts->initial_dq || opts->initial_prev_action || opts->goal_q || opts->goal_pose6;
    if (any) {
      int rc = ensure_scratch(e);
      if (rc != KP1_OK) return rc;
      const size_t n = (size_t)e->n;
      double* base = e->opt_scratch;
      struct Item { const double* src; const double** dst; int w; int flag; } items[5] = {
          {opts->initial_q, &d.initial_q, NJ, OPT_INITIAL_Q}, {opts->initial_dq, &d.initial_dq, NJ, OPT_INITIAL_DQ},
          {opts->initial_prev_action, &d.initial_prev_action, NJ, OPT_INITIAL_PREV_ACTION},
          {opts->goal_q, &d.goal_q, NJ, OPT_GOAL_Q}, {opts->goal_pose6, &d.goal_pose6, 6, OPT_GOAL_POSE6}};
      for (auto& it : items) {
        if (it.src) {
          HIP_TRY(hipMemcpyAsync(base, it.src, sizeof(double) * n * it.w, hipMemcpyHostToDevice, e->stream));
          *it.dst = base;
          d.flags |= it.flag;
        }
        base += n * it.w;
      }
      HIP_TRY(hipStreamSynchronize(e->stream));  // host source buffers are the caller's
    }
  }
  e->mode = mode;
  return e->real_type == KP1_REAL_F64 ? launch_reset<double>(e, mask_dev, d, mode, obs_dev) : launch_reset<float>(e, mask_dev, d, mode, obs_dev);
}

int kp1_step(kp1_env* e, const void* actions_dev, float* obs_dev, void* reward_dev, uint8_t* done_dev, float* terminal_obs_dev,
             int32_t auto_reset) {
  if (!e || !actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(KP1_ERR_INVALID, "NULL buffer passed to kp1_step");
  HIP_TRY(hipSetDevice(e->device));
  return e->real_type == KP1_REAL_F64 ? launch_step<double>(e, actions_dev, obs_dev, reward_dev, done_dev, terminal_obs_dev, auto_reset)
                                      : launch_step<float>(e, actions_dev, obs_dev, reward_dev, done_dev, terminal_obs_dev, auto_reset);
}

int kp1_observe(kp1_env* e, float* obs_dev) {
  if (!e || !obs_dev) return fail(KP1_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  if (e->real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(kp1_observe_kernel<double>, grid, dim3(block), 0, e->stream, state_of<double>(e), (const DevCfg<double>*)e->dev_cfg, e->mode, obs_dev, e->obs_stride);
  else
    hipLaunchKernelGGL(kp1_observe_kernel<float>, grid, dim3(block), 0, e->stream, state_of<float>(e), (const DevCfg<float>*)e->dev_cfg, e->mode, obs_dev, e->obs_stride);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_get_info(kp1_env* e, kp1_info_view* v) {
  if (!e || !v) return fail(KP1_ERR_INVALID, "NULL argument");
  const size_t rs = e->real_size(), n = (size_t)e->n;
  const char* base = (const char*)e->real;
  auto rf = [&](int f) { return (const void*)(base + rs * n * (size_t)f); };
  v->position_error_norm = rf(F_POS_ERR);
  v->orientation_error_norm = rf(F_ORI_ERR);
  v->min_position_error = rf(F_MIN_POS);
  v->executed_delta_q_l2 = rf(F_EXEC_DQ);
  v->action_l2 = rf(F_ACTION_L2);
  v->delta_q_change_l2 = rf(F_DQ_CHANGE);
  v->q = rf(F_Q);
  v->dq = rf(F_DQ);
  v->prev_action = rf(F_PREV_ACTION);
  v->goal_q = rf(F_GOAL_Q);
  v->goal_pose6 = rf(F_GOAL_POSE);
  v->ee_pose6 = rf(F_EE_POSE);
  v->entry_metrics = rf(F_ENTRY);
  v->episode_step = e->ints + n * I_STEP;
  v->dwell_count = e->ints + n * I_DWELL;
  v->near_goal_entry_count = e->ints + n * I_ENTRY;
  v->near_goal_drift_count = e->ints + n * I_DRIFT;
  v->flags = e->ints + n * I_FLAGS;
  v->stage_index = e->ints + n * I_STAGE;
  v->n_envs = e->n;
  v->real_type = e->real_type;
  return KP1_OK;
}

int kp1_eval_accumulate(kp1_env* e, const kp1_eval_buffers* b, const double* action_norm, const uint8_t* done, const uint8_t* active, int32_t step,
                        const double* ready_thresholds, int32_t handoff_confirm_steps, void* stream) {
  if (!e || !b || !b->metrics || !b->counters || !b->flags || !b->state || !b->n_alive) return fail(KP1_ERR_INVALID, "kp1_eval_accumulate: NULL buffer");
  if (step < 0 || (step > 0 && (!action_norm || !done))) return fail(KP1_ERR_INVALID, "kp1_eval_accumulate: a step needs action norms and done bytes");
  if (b->hand_metrics && (!b->hand_step || !b->hand_success || !b->hand_state)) return fail(KP1_ERR_INVALID, "kp1_eval_accumulate: incomplete handoff buffers");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = stream ? (hipStream_t)stream : e->stream;
  HIP_TRY(hipMemsetAsync(b->n_alive, 0, sizeof(int32_t), st));
  const bool track = ready_thresholds != nullptr;
  const double t0 = track ? ready_thresholds[0] : 0.0, t1 = track ? ready_thresholds[1] : 0.0, t2 = track ? ready_thresholds[2] : 0.0,
               t3 = track ? ready_thresholds[3] : 0.0;
  const dim3 grid((unsigned)((e->n + 255) / 256));
  if (e->real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(eval_accumulate_kernel<double>, grid, dim3(256), 0, st, (const double*)e->real, e->n, *b, action_norm, done, active, step, track, t0, t1,
                       t2, t3, handoff_confirm_steps);
  else
    hipLaunchKernelGGL(eval_accumulate_kernel<float>, grid, dim3(256), 0, st, (const float*)e->real, e->n, *b, action_norm, done, active, step, track, t0, t1, t2,
                       t3, handoff_confirm_steps);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_enable_reward_components(kp1_env* e, int32_t enable) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  if (enable && !e->comps) {
    HIP_TRY(hipMalloc(&e->comps, e->real_size() * 64 * (size_t)e->n));
    HIP_TRY(hipMemsetAsync(e->comps, 0, e->real_size() * 64 * (size_t)e->n, e->stream));
  }
  e->comps_enabled = enable != 0;
  return KP1_OK;
}
int kp1_get_reward_components(kp1_env* e, const void** comps_dev, int32_t* n_components) {
  if (!e || !comps_dev || !n_components) return fail(KP1_ERR_INVALID, "NULL argument");
  if (!e->comps_enabled || !e->comps) return fail(KP1_ERR_INVALID, "reward components are not enabled");
  *comps_dev = e->comps;
  *n_components = kp1_num_components(e->mode);
  return KP1_OK;
}

static const char* const APPROACH_COMPONENT_NAMES[KP1_N_APPROACH_COMPONENTS] = {
    "position_progress", "global_orientation_progress", "near_field_orientation_progress", "orientation_progress",
    "orientation_milestone_bonus", "near_field_orientation_center", "pre_near_goal_bonus", "near_goal_bonus",
    "pre_near_to_near_progress", "near_goal_bonus_scale", "coarse_orientation_bonus", "handover_bonus",
    "handover_retention_bonus", "handover_dwell_bonus", "handover_leave_penalty", "handover_regression_penalty",
    "dock_coarse_ready_bonus", "dock_coarse_ready_retention_bonus", "dock_coarse_ready_dwell_bonus",
    "dock_coarse_ready_leave_penalty", "dock_coarse_ready_regression_penalty", "finisher_ready_bonus",
    "finisher_ready_retention_bonus", "finisher_ready_dwell_bonus", "finisher_ready_leave_penalty",
    "finisher_ready_regression_penalty", "near_handoff_action_penalty", "near_handoff_dq_penalty",
    "near_handoff_motion_bonus", "near_handoff_settle_bonus", "same_step_alignment_bonus", "dwell_bonus",
    "drift_penalty", "near_goal_leave_penalty", "drift_penalty_scale", "near_goal_entry_count", "near_goal_drift_count",
    "smoothness_penalty", "smoothness_multiplier", "joint_limit_penalty", "success_bonus", "curr_pos_error",
    "curr_ori_error", "curr_action_norm", "curr_dq_norm", "dwell_count", "in_pre_near_goal", "in_near_goal",
    "in_handover_zone", "in_dock_coarse_ready", "in_dock_coarse_ready_pose", "in_finisher_ready",
    "in_finisher_ready_pose", "in_near_handoff_zone"};
static const char* const DOCK_COMPONENT_NAMES[KP1_N_DOCK_COMPONENTS] = {
    "position_progress", "orientation_progress", "stay_in_zone_bonus", "dwell_bonus", "working_range_bonus",
    "working_range_dwell_bonus", "tight_pose_bonus", "tight_pose_dwell_bonus", "strict_pose_leave_penalty",
    "strict_center_reward", "strict_center_position_penalty", "strict_center_orientation_penalty",
    "strict_center_small_action_bonus", "strict_center_dwell_bonus", "tight_position_shaping",
    "tight_orientation_shaping", "convergence_position_progress", "convergence_orientation_progress",
    "orientation_position_gate_scale", "entry_action_penalty_scale", "leave_zone_penalty", "working_range_exit_penalty",
    "drift_penalty", "smoothness_penalty", "action_delta_violation_penalty", "delta_q_change_penalty",
    "preserve_state_bonus", "strict_hold_bonus", "low_motion_bonus", "tiny_correction_bonus", "worse_than_entry_penalty",
    "near_strict_regression_penalty", "aggressive_action_penalty", "dq_penalty", "joint_limit_penalty", "success_bonus",
    "basin_outer_bonus", "basin_inner_bonus", "basin_dwell_bonus", "basin_outer_exit_penalty", "basin_inner_exit_penalty",
    "basin_dwell_break_penalty", "basin_drift_penalty", "basin_zone_index", "curr_pos_error", "curr_ori_error",
    "dwell_count", "in_tight_pose", "in_near_strict", "entry_pos_error", "entry_ori_error", "entry_action_l2",
    "entry_dq_norm", "entry_to_curr_delta_position_error", "entry_to_curr_delta_orientation_error",
    "entry_to_curr_delta_action_l2", "entry_to_curr_delta_dq_norm", "near_goal_entry_count", "near_goal_drift_count",
    "in_near_goal"};
int kp1_num_components(int32_t mode) { return mode == KP1_MODE_DOCK ? KP1_N_DOCK_COMPONENTS : KP1_N_APPROACH_COMPONENTS; }
const char* kp1_component_name(int32_t mode, int32_t index) {
  if (index < 0 || index >= kp1_num_components(mode)) return "";
  return mode == KP1_MODE_DOCK ? DOCK_COMPONENT_NAMES[index] : APPROACH_COMPONENT_NAMES[index];
}

int kp1_get_state(kp1_env* e, double* q, double* dq, double* prev_action, double* goal_q, double* goal_pose6) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  int rc = ensure_scratch(e);
  if (rc != KP1_OK) return rc;
  const size_t n = (size_t)e->n;
  double* b = e->opt_scratch;
  double *dq_q = b, *dq_dq = b + n * NJ, *dq_pa = b + 2 * n * NJ, *dq_gq = b + 3 * n * NJ, *dq_gp = b + 4 * n * NJ;
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  if (e->real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(kp1_get_state_kernel<double>, grid, dim3(block), 0, e->stream, state_of<double>(e), dq_q, dq_dq, dq_pa, dq_gq, dq_gp);
  else
    hipLaunchKernelGGL(kp1_get_state_kernel<float>, grid, dim3(block), 0, e->stream, state_of<float>(e), dq_q, dq_dq, dq_pa, dq_gq, dq_gp);
  HIP_TRY(kp1::launch_status());
  if (q) HIP_TRY(hipMemcpyAsync(q, dq_q, sizeof(double) * n * NJ, hipMemcpyDeviceToHost, e->stream));
  if (dq) HIP_TRY(hipMemcpyAsync(dq, dq_dq, sizeof(double) * n * NJ, hipMemcpyDeviceToHost, e->stream));
  if (prev_action) HIP_TRY(hipMemcpyAsync(prev_action, dq_pa, sizeof(double) * n * NJ, hipMemcpyDeviceToHost, e->stream));
  if (goal_q) HIP_TRY(hipMemcpyAsync(goal_q, dq_gq, sizeof(double) * n * NJ, hipMemcpyDeviceToHost, e->stream));
  if (goal_pose6) HIP_TRY(hipMemcpyAsync(goal_pose6, dq_gp, sizeof(double) * n * 6, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return KP1_OK;
}

int kp1_set_state(kp1_env* e, const double* q, const double* dq, const double* prev_action, const double* goal_q,
                  const double* goal_pose6, int32_t capture_entry_metrics) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  int rc = ensure_scratch(e);
  if (rc != KP1_OK) return rc;
  const size_t n = (size_t)e->n;
  double* b = e->opt_scratch;
  const double* src[5] = {q, dq, prev_action, goal_q, goal_pose6};
  const double* dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 5; ++k) {
    const int w = k == 4 ? 6 : NJ;
    if (src[k]) {
      HIP_TRY(hipMemcpyAsync(b, src[k], sizeof(double) * n * w, hipMemcpyHostToDevice, e->stream));
      dev[k] = b;
    }
    b += n * NJ;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int block = block_for(e->n);
  const dim3 grid((unsigned)((e->n + block - 1) / block));
  if (e->real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(kp1_set_state_kernel<double>, grid, dim3(block), 0, e->stream, state_of<double>(e), (const DevCfg<double>*)e->dev_cfg,
                       dev[0], dev[1], dev[2], dev[3], dev[4], capture_entry_metrics);
  else
    hipLaunchKernelGGL(kp1_set_state_kernel<float>, grid, dim3(block), 0, e->stream, state_of<float>(e), (const DevCfg<float>*)e->dev_cfg,
                       dev[0], dev[1], dev[2], dev[3], dev[4], capture_entry_metrics);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_rng_get(kp1_env* e, kp1_rng_state* out) {
  if (!e || !out) return fail(KP1_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  const size_t n = (size_t)e->n;
  std::vector<uint64_t> r64(4 * n);
  std::vector<uint32_t> r32(2 * n);
  HIP_TRY(hipMemcpyAsync(r64.data(), e->rng64, r64.size() * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(r32.data(), e->rng32, r32.size() * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < n; ++i) {
    out[i].state_hi = r64[0 * n + i];
    out[i].state_lo = r64[1 * n + i];
    out[i].inc_hi = r64[2 * n + i];
    out[i].inc_lo = r64[3 * n + i];
    out[i].has_uint32 = r32[0 * n + i];
    out[i].uinteger = r32[1 * n + i];
  }
  return KP1_OK;
}
int kp1_rng_set(kp1_env* e, const kp1_rng_state* in) {
  if (!e || !in) return fail(KP1_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  const size_t n = (size_t)e->n;
  std::vector<uint64_t> r64(4 * n);
  std::vector<uint32_t> r32(2 * n);
  for (size_t i = 0; i < n; ++i) {
    r64[0 * n + i] = in[i].state_hi;
    r64[1 * n + i] = in[i].state_lo;
    r64[2 * n + i] = in[i].inc_hi;
    r64[3 * n + i] = in[i].inc_lo;
    r32[0 * n + i] = in[i].has_uint32;
    r32[1 * n + i] = in[i].uinteger;
  }
  HIP_TRY(hipMemcpyAsync(e->rng64, r64.data(), r64.size() * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->rng32, r32.data(), r32.size() * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return KP1_OK;
}

static int state_copy(kp1_env* e, bool save) {
  if (!e) return fail(KP1_ERR_INVALID, "env is NULL");
  HIP_TRY(hipSetDevice(e->device));
  const size_t n = (size_t)e->n;
  const size_t sizes[4] = {e->real_size() * F_NUM_REAL * n, sizeof(int32_t) * I_NUM_INT * n, sizeof(uint64_t) * 4 * n, sizeof(uint32_t) * 2 * n};
  void* live[4] = {e->real, e->ints, e->rng64, e->rng32};
  if (!e->snapshot) {
    if (!save) return fail(KP1_ERR_INVALID, "kp1_state_restore without a kp1_state_snapshot");
    HIP_TRY(hipMalloc(&e->snapshot, sizes[0] + sizes[1] + sizes[2] + sizes[3]));
  }
  char* shadow = (char*)e->snapshot;
  for (int k = 0; k < 4; ++k) {
    HIP_TRY(hipMemcpyAsync(save ? (void*)shadow : live[k], save ? live[k] : (const void*)shadow, sizes[k], hipMemcpyDeviceToDevice, e->stream));
    shadow += sizes[k];
  }
  return KP1_OK;
}
int kp1_state_snapshot(kp1_env* e) { return state_copy(e, true); }
int kp1_state_restore(kp1_env* e) { return state_copy(e, false); }

#ifdef KP1_ENV_TRACE
int kp1_debug_env_trace(unsigned long long* out, int clear) {
  HIP_TRY(hipDeviceSynchronize());
  if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(kp1_env_trace_buf), sizeof(unsigned long long) * KP1_ENV_TRACE_SLOTS * KP1_ENV_TRACE_WAVES));
  if (clear) {
    std::vector<unsigned long long> z(KP1_ENV_TRACE_SLOTS * KP1_ENV_TRACE_WAVES, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(kp1_env_trace_buf), z.data(), sizeof(unsigned long long) * z.size()));
  }
  return KP1_OK;
}
#endif

int kp1_fk_pose6(int32_t device, int32_t real_type, const void* q_dev, void* pose6_dev, int64_t n, void* stream) {
  if (!q_dev || !pose6_dev || n < 0) return fail(KP1_ERR_INVALID, "bad argument");
  if (n == 0) return KP1_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  HIP_TRY(hipSetDevice(device));
  const int block = 256;
  const dim3 grid((unsigned)((n + block - 1) / block));
  void* dfk = nullptr;
  if (real_type != KP1_REAL_F64 && real_type != KP1_REAL_F32) return fail(KP1_ERR_INVALID, "real_type must be KP1_REAL_F32 or KP1_REAL_F64");
  DevFk<double> fk;   // the chain itself is fp64 for both real types; real_type is the type of q and pose6 in memory
  fold_fk<double>(&fk);
  HIP_TRY(hipMalloc(&dfk, sizeof fk));
  HIP_TRY(hipMemcpy(dfk, &fk, sizeof fk, hipMemcpyHostToDevice));
  if (real_type == KP1_REAL_F64)
    hipLaunchKernelGGL(kp1_fk_kernel<double>, grid, dim3(block), 0, (hipStream_t)stream, (const DevFk<double>*)dfk, (const double*)q_dev, (double*)pose6_dev, n);
  else
    hipLaunchKernelGGL(kp1_fk_kernel<float>, grid, dim3(block), 0, (hipStream_t)stream, (const DevFk<double>*)dfk, (const float*)q_dev, (float*)pose6_dev, n);
  hipError_t le = hipGetLastError();
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  (void)hipFree(dfk);
  if (le != hipSuccess) return fail(KP1_ERR_NO_DEVICE, hipGetErrorString(le));
  return KP1_OK;
}

int kp1_pose_error(int32_t device, int32_t real_type, const void* curr_dev, const void* goal_dev, void* pos_err_dev, void* ori_err_dev,
                   void* norms_dev, int64_t n, void* stream) {
  if (!curr_dev || !goal_dev || n < 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_pose_error");
  if (n == 0) return KP1_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  HIP_TRY(hipSetDevice(device));
  const dim3 grid((unsigned)((n + 255) / 256));
  if (real_type == KP1_REAL_F64) {
    hipLaunchKernelGGL(kp1_pose_error_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)curr_dev, (const double*)goal_dev,
                       (double*)pos_err_dev, (double*)ori_err_dev, (double*)norms_dev, n);
  } else if (real_type == KP1_REAL_F32) {
    hipLaunchKernelGGL(kp1_pose_error_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)curr_dev, (const float*)goal_dev,
                       (float*)pos_err_dev, (float*)ori_err_dev, (float*)norms_dev, n);
  } else {
    return fail(KP1_ERR_INVALID, "real_type must be KP1_REAL_F32 or KP1_REAL_F64");
  }
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_joint_utils(int32_t device, int32_t real_type, const kp1_config* cfg, const void* q_dev, const void* dq_dev, void* clipped_dev,
                    void* margin_dev, void* q_norm_dev, void* dq_norm_dev, int64_t n, void* stream) {
  if (!cfg || (!q_dev && !dq_dev) || n < 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_joint_utils");
  if (n == 0) return KP1_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  HIP_TRY(hipSetDevice(device));
  const dim3 grid((unsigned)((n + 255) / 256));
  if (real_type == KP1_REAL_F64) {
    JointLimitsDev<double> lim;
    for (int k = 0; k < NJ; ++k) { lim.lower[k] = cfg->joints.lower[k]; lim.upper[k] = cfg->joints.upper[k]; lim.dlim[k] = cfg->joints.delta_limit[k]; }
    hipLaunchKernelGGL(kp1_joint_utils_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, lim, (const double*)q_dev, (const double*)dq_dev,
                       (double*)clipped_dev, (double*)margin_dev, (double*)q_norm_dev, (double*)dq_norm_dev, n);
  } else if (real_type == KP1_REAL_F32) {
    JointLimitsDev<float> lim;
    for (int k = 0; k < NJ; ++k) { lim.lower[k] = (float)cfg->joints.lower[k]; lim.upper[k] = (float)cfg->joints.upper[k]; lim.dlim[k] = (float)cfg->joints.delta_limit[k]; }
    hipLaunchKernelGGL(kp1_joint_utils_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, lim, (const float*)q_dev, (const float*)dq_dev,
                       (float*)clipped_dev, (float*)margin_dev, (float*)q_norm_dev, (float*)dq_norm_dev, n);
  } else {
    return fail(KP1_ERR_INVALID, "real_type must be KP1_REAL_F32 or KP1_REAL_F64");
  }
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

}  // extern "C"

#include "kp1_route.inc"
#include "kp1_dock_curriculum.inc"
