/*
 * kp1_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See kp1_oracle.h.
 *
 * Reference = /root/reference/hrl_ws/src/hrl_trainer/hrl_trainer/ ; shorthand below:
 *   KP1/ = kinematic_phase1/ , V51/ = v5_1/ .
 * Each function cites the Python it restates.  Arithmetic is fp64 in the reference's
 * operation order wherever the order is visible in the source.
 *
 * The numpy Generator pieces (SeedSequence, PCG64 XSL-RR 128/64, next_double, uniform,
 * Lemire bounded integers) restate numpy's published algorithms (numpy is a third-party
 * dependency of the reference, pinned numpy==2.4.2 in final_codes_docker/Dockerfile.demo:25;
 * 2.2.6 is what is installed here) and are pinned against numpy itself and against the
 * bit_generator.state words captured in tests/golden/resets_*.npz.
 */
#include "kp1_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NJ 7
#define PI 3.141592653589793

/* ------------------------------------------------------------------ V51/ee_fk.py:14-61 */
static const int JOINT_PRISMATIC[NJ] = {1, 0, 0, 0, 0, 0, 0};
static const double ORIGIN_XYZ[NJ][3] = {
    {0.00715921043213119, 0.0000809621375843506, -0.0635},
    {-0.021178, 0.0, 0.1868},
    {-0.0633967414837172, 0.000642782425827271, 0.0602000000000009},
    {-0.000134989688424625, 0.425, 0.0133123982251372},
    {-0.0000850456535865796, -0.39225, -0.0083864861805065},
    {0.0475482889721905, -0.000817137634885778, -0.0805958577476871},
    {0.0436977540622506, 0.000443046177049933, -0.0521517110277254},
};
static const double ORIGIN_RPY[NJ][3] = {
    {0.0, 0.0, 0.0},
    {0.0, 0.0, 0.0},
    {1.5707963267949, 0.0, 1.5707963267949},
    {3.14159265358979, 0.0, 0.0},
    {3.14159265358979, 0.0, -1.5707963267949},
    {3.14159265358979, 1.5707963267949, 0.0},
    {-1.5707963267949, 0.0, -1.5707963267949},
};
static const double AXES_LOCAL[NJ][3] = {
    {1.0, 0.0, 0.0},
    {0.0, 0.0, 1.0},
    {0.0101382310641698, 0.0, -0.999948606814815},
    {0.010138231064165, 0.0, 0.999948606814815},
    {0.0, -0.0101382310641647, -0.999948606814815},
    {0.0, 0.0, -1.0},
    {-0.0101384515502096, 0.0, 0.999948604579338},
};

static void mat3_mul(const double a[9], const double b[9], double c[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j];
      c[3 * i + j] = s;
    }
}
static void mat4_mul(const double a[16], const double b[16], double c[16]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      c[4 * i + j] = s;
    }
}
/* V51/ee_fk.py:64-71 _rpy_to_rot: rz @ ry @ rx */
static void rpy_to_rot(double roll, double pitch, double yaw, double R[9]) {
  double cr = cos(roll), sr = sin(roll), cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw);
  double rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
  double ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
  double rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
  double t[9];
  mat3_mul(rz, ry, t);
  mat3_mul(t, rx, R);
}
/* V51/ee_fk.py:74-88 _rot_axis_local (Rodrigues about axis/(|axis|+1e-12)) */
static void rot_axis_local(const double axis_local[3], double angle, double R[9]) {
  double n = sqrt(axis_local[0] * axis_local[0] + axis_local[1] * axis_local[1] + axis_local[2] * axis_local[2]) + 1e-12;
  double x = axis_local[0] / n, y = axis_local[1] / n, z = axis_local[2] / n;
  double c = cos(angle), s = sin(angle), C = 1.0 - c;
  R[0] = c + x * x * C; R[1] = x * y * C - z * s; R[2] = x * z * C + y * s;
  R[3] = y * x * C + z * s; R[4] = c + y * y * C; R[5] = y * z * C - x * s;
  R[6] = z * x * C - y * s; R[7] = z * y * C + x * s; R[8] = c + z * z * C;
}
/* V51/ee_fk.py:91-95 _make_T */
static void make_T(const double R[9], const double p[3], double T[16]) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
    T[4 * i + 3] = p[i];
  }
  T[12] = T[13] = T[14] = 0.0;
  T[15] = 1.0;
}
/* V51/ee_fk.py:98-117 fk_matrix_from_q7 */
void kp1o_fk_matrix(const double q[7], double T_W[16]) {
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double T[16], Tn[16], M[16], R[9];
  memset(T, 0, sizeof T);
  T[0] = T[5] = T[10] = T[15] = 1.0;
  for (int i = 0; i < NJ; ++i) {
    rpy_to_rot(ORIGIN_RPY[i][0], ORIGIN_RPY[i][1], ORIGIN_RPY[i][2], R);
    make_T(R, ORIGIN_XYZ[i], M);
    mat4_mul(T, M, Tn);
    if (!JOINT_PRISMATIC[i]) {
      static const double Z3[3] = {0, 0, 0};
      rot_axis_local(AXES_LOCAL[i], q[i], R);
      make_T(R, Z3, M);
    } else {
      double p[3] = {AXES_LOCAL[i][0] * q[i], AXES_LOCAL[i][1] * q[i], AXES_LOCAL[i][2] * q[i]};
      make_T(I3, p, M);
    }
    mat4_mul(Tn, M, T);
  }
  memcpy(T_W, T, sizeof T);
}
/* V51/ee_fk.py:120-134 ee_pose6_from_q ; KP1/kinematics/fk_interface.py:21-22 */
void kp1o_fk_pose6(const double q[7], double pose6[6]) {
  double T[16];
  kp1o_fk_matrix(q, T);
  pose6[0] = T[3];
  pose6[1] = T[7];
  pose6[2] = T[11];
  pose6[3] = atan2(T[9], T[10]);                                /* R[2,1], R[2,2] */
  pose6[4] = atan2(-T[8], sqrt(T[0] * T[0] + T[4] * T[4]));      /* -R[2,0], sqrt(R00^2+R10^2) */
  pose6[5] = atan2(T[4], T[0]);                                 /* R[1,0], R[0,0] */
}

/* KP1/kinematics/pose_utils.py:11-12 wrap_to_pi with numpy floor-mod semantics */
double kp1o_wrap_to_pi(double v) {
  double a = v + PI, b = 2.0 * PI;
  double m = fmod(a, b);
  if (m != 0.0) {
    if (m < 0.0) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m - PI;
}
/* KP1/kinematics/pose_utils.py:15-26 */
void kp1o_pose_error(const double curr[6], const double goal[6], double pos_err[3], double ori_err[3]) {
  for (int i = 0; i < 3; ++i) pos_err[i] = goal[i] - curr[i];
  for (int i = 0; i < 3; ++i) ori_err[i] = kp1o_wrap_to_pi(goal[3 + i] - curr[3 + i]);
}
static double norm_n(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double maxd(double a, double b) { return a > b ? a : b; }
static double mind(double a, double b) { return a < b ? a : b; }
static int clipi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static int mini(int a, int b) { return a < b ? a : b; }
static int maxi(int a, int b) { return a > b ? a : b; }

/* KP1/kinematics/joint_limits.py:133-135 */
void kp1o_clip_q(const kp1_joint_specs* js, const double q[7], double out[7]) {
  for (int i = 0; i < NJ; ++i) out[i] = clipd(q[i], js->lower[i], js->upper[i]);
}
/* KP1/kinematics/joint_limits.py:166-174 */
void kp1o_joint_limit_margin(const kp1_joint_specs* js, const double q[7], double out[7]) {
  for (int i = 0; i < NJ; ++i) {
    double span = maxd(js->upper[i] - js->lower[i], 1e-9);
    double left = (q[i] - js->lower[i]) / span, right = (js->upper[i] - q[i]) / span;
    out[i] = clipd(2.0 * mind(left, right), 0.0, 1.0);
  }
}
/* KP1/kinematics/joint_limits.py:153-158 */
void kp1o_normalize_q(const kp1_joint_specs* js, const double q[7], double out[7]) {
  for (int i = 0; i < NJ; ++i) {
    double span = maxd(js->upper[i] - js->lower[i], 1e-9);
    out[i] = clipd(2.0 * ((q[i] - js->lower[i]) / span) - 1.0, -1.0, 1.0);
  }
}
/* KP1/kinematics/joint_limits.py:161-163 */
void kp1o_normalize_dq(const kp1_joint_specs* js, const double dq[7], double out[7]) {
  for (int i = 0; i < NJ; ++i) out[i] = clipd(dq[i] / maxd(js->delta_limit[i], 1e-9), -1.0, 1.0);
}

/* ------------------------------------------------------------------ numpy Generator(PCG64) */
/* numpy/random/bit_generator.pyx SeedSequence (pool_size 4) */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u
static uint32_t ss_hashmix(uint32_t value, uint32_t* hash_const) {
  value ^= *hash_const;
  *hash_const *= SS_MULT_A;
  value *= *hash_const;
  value ^= value >> 16;
  return value;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
  r ^= r >> 16;
  return r;
}
static void seed_sequence_state4x64(uint64_t seed, uint64_t out[4]) {
  uint32_t entropy[2];
  int n_ent = 1;
  entropy[0] = (uint32_t)(seed & 0xffffffffu);
  entropy[1] = (uint32_t)(seed >> 32);
  if (entropy[1] != 0) n_ent = 2;
  uint32_t pool[4];
  uint32_t hc = SS_INIT_A;
  for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n_ent ? entropy[i] : 0u, &hc);
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 4; ++d)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
  /* generate_state(4, uint64) = 8 uint32 words, pairs little-endian */
  uint32_t w[8];
  uint32_t hb = SS_INIT_B;
  for (int i = 0; i < 8; ++i) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= SS_MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  for (int i = 0; i < 4; ++i) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
#define PCG_MULT ((((unsigned __int128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL)
/* numpy/random/src/pcg64/pcg64.h pcg_setseq_128_srandom_r */
void kp1o_rng_seed(kp1o_rng* r, uint64_t seed) {
  uint64_t s[4];
  seed_sequence_state4x64(seed, s);
  unsigned __int128 initstate = ((unsigned __int128)s[0] << 64) | s[1];
  unsigned __int128 initseq = ((unsigned __int128)s[2] << 64) | s[3];
  r->state = 0;
  r->inc = (initseq << 1) | 1u;
  r->state = r->state * PCG_MULT + r->inc;
  r->state += initstate;
  r->state = r->state * PCG_MULT + r->inc;
  r->has_uint32 = 0;
  r->uinteger = 0;
}
uint64_t kp1o_rng_next64(kp1o_rng* r) {
  r->state = r->state * PCG_MULT + r->inc;
  uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
uint32_t kp1o_rng_next32(kp1o_rng* r) {
  if (r->has_uint32) {
    r->has_uint32 = 0;
    return r->uinteger;
  }
  uint64_t n = kp1o_rng_next64(r);
  r->has_uint32 = 1;
  r->uinteger = (uint32_t)(n >> 32);
  return (uint32_t)(n & 0xffffffffu);
}
double kp1o_rng_double(kp1o_rng* r) { return (double)(kp1o_rng_next64(r) >> 11) * (1.0 / 9007199254740992.0); }
/* numpy/random/src/distributions/distributions.c random_bounded_uint64_fill (Lemire, no mask) */
int64_t kp1o_rng_integers(kp1o_rng* r, int64_t low, int64_t high_exclusive) {
  uint64_t rng = (uint64_t)(high_exclusive - 1 - low);
  if (rng == 0) return low;
  if (rng <= 0xFFFFFFFFull) {
    if (rng == 0xFFFFFFFFull) return low + (int64_t)kp1o_rng_next32(r);
    uint32_t rng_excl = (uint32_t)rng + 1u;
    uint64_t m = (uint64_t)kp1o_rng_next32(r) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
      uint32_t threshold = (0xFFFFFFFFu - (uint32_t)rng) % rng_excl;
      while (leftover < threshold) {
        m = (uint64_t)kp1o_rng_next32(r) * rng_excl;
        leftover = (uint32_t)m;
      }
    }
    return low + (int64_t)(m >> 32);
  }
  if (rng == 0xFFFFFFFFFFFFFFFFull) return low + (int64_t)kp1o_rng_next64(r);
  {
    uint64_t rng_excl = rng + 1;
    unsigned __int128 m = (unsigned __int128)kp1o_rng_next64(r) * rng_excl;
    uint64_t leftover = (uint64_t)m;
    if (leftover < rng_excl) {
      uint64_t threshold = (0xFFFFFFFFFFFFFFFFull - rng) % rng_excl;
      while (leftover < threshold) {
        m = (unsigned __int128)kp1o_rng_next64(r) * rng_excl;
        leftover = (uint64_t)m;
      }
    }
    return low + (int64_t)(uint64_t)(m >> 64);
  }
}
void kp1o_rng_get(const kp1o_rng* r, kp1_rng_state* o) {
  o->state_hi = (uint64_t)(r->state >> 64);
  o->state_lo = (uint64_t)r->state;
  o->inc_hi = (uint64_t)(r->inc >> 64);
  o->inc_lo = (uint64_t)r->inc;
  o->has_uint32 = (uint32_t)r->has_uint32;
  o->uinteger = r->uinteger;
}
void kp1o_rng_set(kp1o_rng* r, const kp1_rng_state* i) {
  r->state = ((unsigned __int128)i->state_hi << 64) | i->state_lo;
  r->inc = ((unsigned __int128)i->inc_hi << 64) | i->inc_lo;
  r->has_uint32 = (int)i->has_uint32;
  r->uinteger = i->uinteger;
}
/* Generator.uniform(low, high) with array bounds: low + (high - low) * next_double, one draw per element */
static void rng_uniform7(kp1o_rng* r, const double low[7], const double high[7], double out[7]) {
  for (int i = 0; i < NJ; ++i) {
    double range = high[i] - low[i];
    out[i] = low[i] + range * kp1o_rng_double(r);
  }
}
static void rng_uniform_sym7(kp1o_rng* r, const double noise[7], double out[7]) {
  double lo[7], hi[7];
  for (int i = 0; i < NJ; ++i) {
    lo[i] = -noise[i];
    hi[i] = noise[i];
  }
  rng_uniform7(r, lo, hi, out);
}
static int any_positive7(const double v[7]) {
  for (int i = 0; i < NJ; ++i)
    if (v[i] > 0.0) return 1;
  return 0;
}

/* ------------------------------------------------------------------ defaults */
#define KP1O_SET(type, name, dflt) s->name = dflt;
void kp1o_config_default(kp1_config* cfg) {
  memset(cfg, 0, sizeof *cfg);
  { kp1_env_scalars* s = &cfg->env; KP1_ENV_FIELDS(KP1O_SET) }
  { kp1_approach_reward* s = &cfg->reward; KP1_APPROACH_REWARD_FIELDS(KP1O_SET) }
  { kp1_dock_reward* s = &cfg->dock_reward; KP1_DOCK_REWARD_FIELDS(KP1O_SET) }
  { kp1_termination* s = &cfg->termination; KP1_TERMINATION_FIELDS(KP1O_SET) }
  { kp1_observation* s = &cfg->observation; KP1_OBSERVATION_FIELDS(KP1O_SET) }
  { kp1_stage_sampling* s = &cfg->stage_sampling; KP1_STAGE_SAMPLING_FIELDS(KP1O_SET) }
  { kp1_random_start* s = &cfg->random_start; KP1_RANDOM_START_FIELDS(KP1O_SET) }
  { kp1_dock_reset* s = &cfg->dock_reset; KP1_DOCK_RESET_FIELDS(KP1O_SET) }
  /* KP1/kinematics/joint_limits.py:37-47 */
  static const double lim[NJ] = {0.385, PI, PI, PI, PI, PI, PI};
  static const double dl[NJ] = {0.08, 0.30, 0.24, 0.24, 0.30, 0.40, 0.30};
  for (int i = 0; i < NJ; ++i) {
    cfg->joints.lower[i] = -lim[i];
    cfg->joints.upper[i] = lim[i];
    cfg->joints.delta_limit[i] = dl[i];
    cfg->random_start.failure_recovery_q_noise[i] = 0.04;
  }
  /* KP1/envs/reset_samplers.py:50-54 */
  static const double gn[NJ] = {0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01};
  static const double iq[NJ] = {0.01, 0.02, 0.03, 0.02, 0.015, 0.015, 0.01};
  static const double cq[NJ] = {0.006, 0.012, 0.018, 0.012, 0.009, 0.009, 0.006};
  memcpy(cfg->dock_reset.goal_noise, gn, sizeof gn);
  memcpy(cfg->dock_reset.init_q_noise, iq, sizeof iq);
  memcpy(cfg->dock_reset.close_init_q_noise, cq, sizeof cq);
  /* KP1/envs/curriculum.py:36-78 default_point_curriculum_stages */
  static const double goal_noise[6][NJ] = {
      {0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01}, {0.02, 0.06, 0.08, 0.06, 0.04, 0.04, 0.03},
      {0.03, 0.09, 0.12, 0.09, 0.06, 0.05, 0.04}, {0.04, 0.12, 0.16, 0.12, 0.08, 0.06, 0.05},
      {0.05, 0.14, 0.18, 0.14, 0.09, 0.07, 0.06}, {0.06, 0.18, 0.22, 0.16, 0.10, 0.08, 0.07}};
  static const double start_noise[6] = {0.0, 0.0, 0.0, 0.01, 0.02, 0.03};
  static const double goal4[NJ] = {0.03, -0.04, 0.05, -0.03, 0.02, -0.01, 0.01};
  cfg->curriculum_enabled = 1;
  cfg->n_stages = 6;
  for (int k = 0; k < 6; ++k) {
    memcpy(cfg->stages[k].goal_noise, goal_noise[k], sizeof goal_noise[k]);
    for (int i = 1; i < NJ; ++i) cfg->stages[k].start_noise[i] = start_noise[k];
  }
  memcpy(cfg->stages[4].goal_q, goal4, sizeof goal4);
}

/* ------------------------------------------------------------------ reset samplers */
/* KP1/envs/curriculum.py:90-101 sample_stage_joint_target */
static void sample_stage_joint_target(kp1o_env* e, const double base[7], const double noise[7], double out[7]) {
  double b[7];
  memcpy(b, base, sizeof b);
  if (any_positive7(noise)) {
    double d[7];
    rng_uniform_sym7(&e->rng, noise, d);
    for (int i = 0; i < NJ; ++i) b[i] = b[i] + d[i];
  }
  kp1o_clip_q(&e->cfg.joints, b, out);
}
/* KP1/kinematics/joint_limits.py:138-150 sample_joint_configuration */
static void sample_joint_configuration(kp1o_env* e, double margin_fraction, double out[7]) {
  double lo[7], hi[7];
  for (int i = 0; i < NJ; ++i) {
    double span = e->cfg.joints.upper[i] - e->cfg.joints.lower[i];
    double margin = maxd(span * margin_fraction, 1e-6);
    lo[i] = e->cfg.joints.lower[i] + margin;
    hi[i] = e->cfg.joints.upper[i] - margin;
  }
  rng_uniform7(&e->rng, lo, hi, out);
}
static int opt_or(int v, int dflt) { return v == KP1_UNSET ? dflt : v; }

/* KP1/envs/reset_samplers.py:344-390 _sample_workspace_stage_index */
static int sample_workspace_stage_index(kp1o_env* e, int current_stage_index) {
  const kp1_stage_sampling* c = &e->cfg.stage_sampling;
  int stage_count = e->cfg.n_stages;
  int current = clipi(current_stage_index, 0, maxi(stage_count - 1, 0));
  if (!c->enabled || current <= 0) return current;
  double current_ratio = maxd(c->current_stage_ratio, 0.0);
  double previous_ratio = maxd(c->previous_stage_ratio, 0.0);
  double old_ratio = maxd(c->old_workspace_replay_ratio, 0.0);
  double failure_ratio = maxd(c->failure_replay_ratio, 0.0);
  double total = current_ratio + previous_ratio + old_ratio + failure_ratio;
  if (total <= 0.0) return current;
  double draw = kp1o_rng_double(&e->rng) * total;
  if (draw < current_ratio) return current;
  draw -= current_ratio;
  if (draw < previous_ratio && current > 0) {
    int low = maxi(c->previous_stage_min_index, 0);
    int high = maxi(current - 1, low);
    return (int)kp1o_rng_integers(&e->rng, low, high + 1);
  }
  draw -= previous_ratio;
  int old_max = opt_or(c->old_workspace_max_stage_index, mini(5, current));
  old_max = clipi(old_max, 0, mini(stage_count - 1, current));
  if (draw < old_ratio && old_max >= 0) return (int)kp1o_rng_integers(&e->rng, 0, old_max + 1);
  int replay_max = maxi(mini(old_max, current - 1), 0);
  return replay_max > 0 ? (int)kp1o_rng_integers(&e->rng, 0, replay_max + 1) : current;
}

enum { SRC_HOME = 0, SRC_OLD_SUCCESS, SRC_RANDOM_VALID, SRC_FRONTIER, SRC_FAILURE_RECOVERY, SRC_STRESS };
/* KP1/envs/reset_samplers.py:308-318 _sample_ratio_key (dict insertion order :239-244) */
static int sample_ratio_key(kp1o_env* e, const double ratios[6], int dflt) {
  double clean[6], total = 0.0;
  for (int i = 0; i < 6; ++i) {
    clean[i] = maxd(ratios[i], 0.0);
    total += clean[i];
  }
  if (total <= 0.0) return dflt;
  double draw = kp1o_rng_double(&e->rng) * total;
  for (int i = 0; i < 6; ++i) {
    if (draw <= clean[i]) return i;
    draw -= clean[i];
  }
  return dflt;
}
/* KP1/envs/reset_samplers.py:321-341 _sample_target_stage_for_source */
static int sample_target_stage_for_source(kp1o_env* e, int source, int current) {
  const kp1_random_start* c = &e->cfg.random_start;
  int n = e->cfg.n_stages;
  if (source == SRC_HOME || source == SRC_OLD_SUCCESS) {
    int mx = clipi(opt_or(c->known_target_max_stage_index, mini(7, current)), 0, n - 1);
    return (int)kp1o_rng_integers(&e->rng, 0, mx + 1);
  }
  if (source == SRC_FRONTIER) {
    int mn = clipi(opt_or(c->frontier_target_min_stage_index, mini(8, current)), 0, n - 1);
    int mx = clipi(opt_or(c->frontier_target_max_stage_index, current), mn, n - 1);
    return (int)kp1o_rng_integers(&e->rng, mn, mx + 1);
  }
  if (source == SRC_STRESS) {
    int mn = clipi(opt_or(c->stress_target_min_stage_index, mini(8, current)), 0, n - 1);
    int mx = clipi(opt_or(c->stress_target_max_stage_index, n - 1), mn, n - 1);
    return (int)kp1o_rng_integers(&e->rng, mn, mx + 1);
  }
  int mx = clipi(opt_or(c->mixed_target_max_stage_index, current), 0, n - 1);
  return (int)kp1o_rng_integers(&e->rng, 0, mx + 1);
}

typedef struct reset_sample {
  double initial_q[7], goal_q[7], goal_pose6[6];
  int has_dq, has_prev_action;
  double initial_dq[7], initial_prev_action[7];
  int stage;
} reset_sample;

/* KP1/envs/reset_samplers.py:213-305 sample_random_start_workspace_pair */
static void sample_random_start_workspace_pair(kp1o_env* e, int stage_index, reset_sample* rs) {
  const kp1_random_start* c = &e->cfg.random_start;
  const kp1_config* cfg = &e->cfg;
  int n = cfg->n_stages;
  int current = clipi(stage_index, 0, n - 1);
  double ratios[6] = {c->home_start_ratio, c->old_successful_start_ratio, c->random_valid_q_start_ratio,
                      c->frontier_pair_ratio, c->failure_recovery_start_ratio, c->stress_start_ratio};
  int source = sample_ratio_key(e, ratios, SRC_OLD_SUCCESS);
  int target_stage = sample_target_stage_for_source(e, source, current);
  double target_q[7], start_q[7];
  sample_stage_joint_target(e, cfg->stages[target_stage].goal_q, cfg->stages[target_stage].goal_noise, target_q);
  if (source == SRC_HOME) {
    int ss = mini(c->home_stage_index, n - 1);
    sample_stage_joint_target(e, cfg->stages[ss].start_q, cfg->stages[ss].start_noise, start_q);
  } else if (source == SRC_OLD_SUCCESS) {
    int max_old = clipi(opt_or(c->old_success_max_stage_index, mini(7, current)), 0, n - 1);
    int old_idx = (int)kp1o_rng_integers(&e->rng, 0, max_old + 1);
    sample_stage_joint_target(e, cfg->stages[old_idx].goal_q, cfg->stages[old_idx].goal_noise, start_q);
  } else if (source == SRC_FRONTIER) {
    int fmin = clipi(opt_or(c->frontier_min_stage_index, mini(8, current)), 0, n - 1);
    int fmax = clipi(opt_or(c->frontier_max_stage_index, current), fmin, n - 1);
    int fi = (int)kp1o_rng_integers(&e->rng, fmin, fmax + 1);
    sample_stage_joint_target(e, cfg->stages[fi].start_q, cfg->stages[fi].start_noise, start_q);
  } else if (source == SRC_FAILURE_RECOVERY) {
    double d[7], t[7];
    rng_uniform_sym7(&e->rng, c->failure_recovery_q_noise, d);
    for (int i = 0; i < NJ; ++i) t[i] = target_q[i] + d[i];
    kp1o_clip_q(&cfg->joints, t, start_q);
  } else if (source == SRC_STRESS) {
    double margin = c->has_stress_start_margin_fraction ? c->stress_start_margin_fraction : cfg->env.start_sample_margin_fraction;
    sample_joint_configuration(e, margin, start_q);
  } else {
    double margin = c->has_random_valid_start_margin_fraction ? c->random_valid_start_margin_fraction : cfg->env.start_sample_margin_fraction;
    sample_joint_configuration(e, margin, start_q);
  }
  rs->has_dq = rs->has_prev_action = 1; /* the sampler always returns arrays (zeros when no noise) */
  if (any_positive7(c->initial_dq_noise)) rng_uniform_sym7(&e->rng, c->initial_dq_noise, rs->initial_dq);
  else memset(rs->initial_dq, 0, sizeof rs->initial_dq);
  if (any_positive7(c->initial_prev_action_noise)) rng_uniform_sym7(&e->rng, c->initial_prev_action_noise, rs->initial_prev_action);
  else memset(rs->initial_prev_action, 0, sizeof rs->initial_prev_action);
  if (c->min_pair_joint_l2 > 0.0) {
    for (int k = 0; k < 12; ++k) {
      double d[7];
      for (int i = 0; i < NJ; ++i) d[i] = target_q[i] - start_q[i];
      if (norm_n(d, NJ) >= c->min_pair_joint_l2) break;
      target_stage = sample_target_stage_for_source(e, source, current);
      sample_stage_joint_target(e, cfg->stages[target_stage].goal_q, cfg->stages[target_stage].goal_noise, target_q);
    }
  }
  kp1o_clip_q(&cfg->joints, target_q, rs->goal_q);
  kp1o_clip_q(&cfg->joints, start_q, rs->initial_q);
  kp1o_fk_pose6(rs->goal_q, rs->goal_pose6);
  rs->stage = target_stage;
}

/* KP1/envs/reset_samplers.py:168-210 sample_approach_reset (route reset: out of scope) */
static void sample_approach_reset(kp1o_env* e, int stage_index, reset_sample* rs) {
  const kp1_config* cfg = &e->cfg;
  rs->has_dq = rs->has_prev_action = 0;
  if (cfg->random_start.enabled && cfg->curriculum_enabled && cfg->n_stages > 0) {
    sample_random_start_workspace_pair(e, stage_index, rs);
    return;
  }
  if (cfg->curriculum_enabled && cfg->n_stages > 0) {
    int idx = sample_workspace_stage_index(e, stage_index);
    sample_stage_joint_target(e, cfg->stages[idx].start_q, cfg->stages[idx].start_noise, rs->initial_q);
    sample_stage_joint_target(e, cfg->stages[idx].goal_q, cfg->stages[idx].goal_noise, rs->goal_q);
    rs->stage = idx;
  } else {
    sample_joint_configuration(e, cfg->env.start_sample_margin_fraction, rs->initial_q);
    sample_joint_configuration(e, cfg->env.goal_sample_margin_fraction, rs->goal_q);
    rs->stage = 0;
  }
  kp1o_fk_pose6(rs->goal_q, rs->goal_pose6);
}

/* KP1/envs/reset_samplers.py:474-515 _sample_close_bucket_initial_q */
static void sample_close_bucket_initial_q(kp1o_env* e, const double goal_q[7], const double goal_pose6[6], double out[7]) {
  const kp1_dock_reset* c = &e->cfg.dock_reset;
  double best_q[7];
  int have_best = 0;
  double best_dist = INFINITY;
  int attempts = maxi(c->close_bucket_max_attempts, 1);
  for (int a = 0; a < attempts; ++a) {
    double d[7], t[7], cand[7], pose[6], pe[3], oe[3];
    rng_uniform_sym7(&e->rng, c->close_init_q_noise, d);
    for (int i = 0; i < NJ; ++i) t[i] = goal_q[i] + d[i];
    kp1o_clip_q(&e->cfg.joints, t, cand);
    kp1o_fk_pose6(cand, pose);
    kp1o_pose_error(pose, goal_pose6, pe, oe);
    double pn = norm_n(pe, 3), on = norm_n(oe, 3);
    if (c->close_bucket_min_pos_error_m <= pn && pn <= c->close_bucket_max_pos_error_m &&
        on >= c->close_bucket_min_ori_error_rad && on <= c->close_bucket_max_ori_error_rad) {
      memcpy(out, cand, sizeof cand);
      return;
    }
    double bd;
    if (pn < c->close_bucket_min_pos_error_m) bd = c->close_bucket_min_pos_error_m - pn;
    else if (pn > c->close_bucket_max_pos_error_m) bd = pn - c->close_bucket_max_pos_error_m;
    else bd = maxd(maxd(c->close_bucket_min_ori_error_rad - on, on - c->close_bucket_max_ori_error_rad), 0.0);
    if (bd < best_dist) {
      memcpy(best_q, cand, sizeof cand);
      have_best = 1;
      best_dist = bd;
    }
  }
  if (have_best) memcpy(out, best_q, sizeof best_q);
  else kp1o_clip_q(&e->cfg.joints, goal_q, out);
}

/* KP1/envs/reset_samplers.py:426-471 sample_dock_reset */
static void sample_dock_reset(kp1o_env* e, int stage_index, reset_sample* rs) {
  const kp1_config* cfg = &e->cfg;
  const kp1_dock_reset* c = &cfg->dock_reset;
  rs->has_dq = rs->has_prev_action = 0;
  rs->stage = 0;
  if (c->handoff_state_probability > 0.0 && e->n_handoff > 0 && kp1o_rng_double(&e->rng) < c->handoff_state_probability) {
    const kp1_handoff_state* s = &e->handoff[kp1o_rng_integers(&e->rng, 0, e->n_handoff)];
    memcpy(rs->initial_q, s->initial_q, sizeof rs->initial_q);
    memcpy(rs->goal_q, s->goal_q, sizeof rs->goal_q);
    memcpy(rs->goal_pose6, s->goal_pose6, sizeof rs->goal_pose6);
    memcpy(rs->initial_dq, s->initial_dq, sizeof rs->initial_dq);
    memcpy(rs->initial_prev_action, s->initial_prev_action, sizeof rs->initial_prev_action);
    rs->has_dq = rs->has_prev_action = 1;
    return;
  }
  if (cfg->curriculum_enabled && cfg->n_stages > 0) {
    int idx = clipi(stage_index, 0, cfg->n_stages - 1);
    sample_stage_joint_target(e, cfg->stages[idx].goal_q, cfg->stages[idx].goal_noise, rs->goal_q);
    rs->stage = idx;
  } else {
    sample_stage_joint_target(e, c->goal_q, c->goal_noise, rs->goal_q);
  }
  kp1o_fk_pose6(rs->goal_q, rs->goal_pose6);
  if (c->close_bucket_probability > 0.0 && kp1o_rng_double(&e->rng) < c->close_bucket_probability) {
    sample_close_bucket_initial_q(e, rs->goal_q, rs->goal_pose6, rs->initial_q);
    return;
  }
  double d[7], t[7];
  rng_uniform_sym7(&e->rng, c->init_q_noise, d);
  for (int i = 0; i < NJ; ++i) t[i] = rs->goal_q[i] + d[i];
  kp1o_clip_q(&cfg->joints, t, rs->initial_q);
}

/* ------------------------------------------------------------------ env */
/* KP1/envs/arm_kinematic_env.py:74-100 __init__ */
void kp1o_env_init(kp1o_env* e, const kp1_config* cfg) {
  memset(e, 0, sizeof *e);
  e->cfg = *cfg;
  kp1o_rng_seed(&e->rng, 0); /* default_rng(0) :80 */
  e->min_pos_error = INFINITY;
  kp1o_fk_pose6(e->q, e->ee_pose6);
  e->curriculum_stage_index = 0;
  e->policy_mode = cfg->env.mode;
}
void kp1o_env_set_handoff(kp1o_env* e, const kp1_handoff_state* states, int n) {
  e->handoff = states;
  e->n_handoff = n;
}
void kp1o_env_seed(kp1o_env* e, uint64_t seed) { kp1o_rng_seed(&e->rng, seed); }
/* KP1/envs/arm_kinematic_env.py:446-449 */
void kp1o_env_set_stage(kp1o_env* e, int stage) {
  if (!e->cfg.curriculum_enabled) return;
  e->curriculum_stage_index = clipi(stage, 0, e->cfg.n_stages - 1);
}
/* KP1/envs/arm_kinematic_env.py:432-444 */
static int is_near_goal(const kp1o_env* e, double pos, double ori) {
  const kp1_approach_reward* r = &e->cfg.reward;
  if (pos > r->near_goal_pos_threshold_m) return 0;
  if (r->use_orientation_gate && ori > r->near_goal_ori_threshold_rad) return 0;
  return 1;
}
static int is_pre_near_goal(const kp1o_env* e, double pos, double ori) {
  const kp1_approach_reward* r = &e->cfg.reward;
  if (pos > r->pre_near_goal_pos_threshold_m) return 0;
  if (r->use_orientation_gate && ori > r->near_goal_ori_threshold_rad) return 0;
  return 1;
}
/* KP1/envs/arm_kinematic_env.py:425-430 */
void kp1o_env_capture_entry_metrics(kp1o_env* e) {
  double pe[3], oe[3];
  kp1o_pose_error(e->ee_pose6, e->goal_pose6, pe, oe);
  e->entry_position_error_norm = norm_n(pe, 3);
  e->entry_orientation_error_norm = norm_n(oe, 3);
  e->entry_action_l2 = norm_n(e->prev_action, NJ);
  e->entry_dq_norm = norm_n(e->dq, NJ);
}
/* KP1/envs/observation_builder.py:29-94 build_observation, packed in SB3 key order (kp1.h) */
void kp1o_env_observe(const kp1o_env* e, float obs[KP1_OBS_DIM]) {
  const kp1_config* cfg = &e->cfg;
  double pe[3], oe[3], v[7];
  memset(obs, 0, sizeof(float) * KP1_OBS_DIM);
  kp1o_pose_error(e->ee_pose6, e->goal_pose6, pe, oe);
  kp1o_normalize_q(&cfg->joints, e->q, v);
  for (int i = 0; i < NJ; ++i) obs[KP1_OBS_Q + i] = (float)v[i];
  kp1o_normalize_dq(&cfg->joints, e->dq, v);
  for (int i = 0; i < NJ; ++i) obs[KP1_OBS_DQ + i] = (float)v[i];
  for (int i = 0; i < NJ; ++i) obs[KP1_OBS_PREV_ACTION + i] = (float)clipd(e->prev_action[i], -1.0, 1.0);
  for (int i = 0; i < 3; ++i) obs[KP1_OBS_GOAL_POS_ERR + i] = (float)clipd(pe[i] / cfg->observation.pos_err_scale_m, -1.0, 1.0);
  for (int i = 0; i < 3; ++i) obs[KP1_OBS_GOAL_ORI_ERR + i] = (float)clipd(oe[i] / cfg->observation.ori_err_scale_rad, -1.0, 1.0);
  obs[KP1_OBS_TASK_TYPE] = 1.0f;
  int mode_index = e->policy_mode == KP1_MODE_APPROACH ? 0 : (e->policy_mode == KP1_MODE_DOCK ? 1 : 2);
  obs[KP1_OBS_MODE_FLAG + clipi(mode_index, 0, 3)] = 1.0f;
  double ep = (double)e->episode_step / (double)maxi(cfg->env.episode_length, 1);
  double dp = (double)e->dwell_count / (double)maxi(cfg->env.dwell_steps_target, 1);
  obs[KP1_OBS_PROGRESS + 0] = (float)clipd(ep, 0.0, 1.0);
  obs[KP1_OBS_PROGRESS + 1] = (float)clipd(dp, 0.0, 1.0);
  kp1o_joint_limit_margin(&cfg->joints, e->q, v);
  for (int i = 0; i < NJ; ++i) obs[KP1_OBS_JOINT_LIMIT_MARGIN + i] = (float)v[i];
}

/* KP1/envs/arm_kinematic_env.py:102-211 reset (bridge branch out of scope) */
void kp1o_env_reset(kp1o_env* e, const kp1o_reset_opts* opts, float obs[KP1_OBS_DIM]) {
  static const kp1o_reset_opts none = {0, 0, 0, 0, 0, -1};
  const kp1o_reset_opts* o = opts ? opts : &none;
  const kp1_config* cfg = &e->cfg;
  reset_sample rs;
  int have_sample = 0;
  e->episode_step = 0;
  e->dwell_count = 0;
  e->near_goal_entry_count = 0;
  e->near_goal_drift_count = 0;
  e->pre_near_goal_hit = 0;
  e->near_goal_hit = 0;
  e->min_pos_error = INFINITY;
  e->policy_mode = o->policy_mode >= 0 ? o->policy_mode : cfg->env.mode;
  if (o->initial_q) {
    kp1o_clip_q(&cfg->joints, o->initial_q, e->q);
  } else {
    if (e->policy_mode == KP1_MODE_DOCK) sample_dock_reset(e, e->curriculum_stage_index, &rs);
    else sample_approach_reset(e, e->curriculum_stage_index, &rs);
    have_sample = 1;
    e->last_reset_stage = rs.stage;
    memcpy(e->q, rs.initial_q, sizeof e->q);
  }
  if (o->initial_dq) memcpy(e->dq, o->initial_dq, sizeof e->dq);
  else if (have_sample && rs.has_dq) memcpy(e->dq, rs.initial_dq, sizeof e->dq);
  else memset(e->dq, 0, sizeof e->dq);
  if (o->initial_prev_action) memcpy(e->prev_action, o->initial_prev_action, sizeof e->prev_action);
  else if (have_sample && rs.has_prev_action) memcpy(e->prev_action, rs.initial_prev_action, sizeof e->prev_action);
  else memset(e->prev_action, 0, sizeof e->prev_action);
  kp1o_fk_pose6(e->q, e->ee_pose6);
  if (o->goal_pose6) {
    memcpy(e->goal_pose6, o->goal_pose6, sizeof e->goal_pose6);
    if (o->goal_q) memcpy(e->goal_q, o->goal_q, sizeof e->goal_q);
    else memset(e->goal_q, 0, sizeof e->goal_q);
  } else if (o->goal_q) {
    kp1o_clip_q(&cfg->joints, o->goal_q, e->goal_q);
    kp1o_fk_pose6(e->goal_q, e->goal_pose6);
  } else if (!o->initial_q) {
    memcpy(e->goal_q, rs.goal_q, sizeof e->goal_q);
    memcpy(e->goal_pose6, rs.goal_pose6, sizeof e->goal_pose6);
  } else {
    /* fk_interface.py:25-32 sample_reachable_target */
    sample_joint_configuration(e, cfg->env.goal_sample_margin_fraction, e->goal_q);
    kp1o_fk_pose6(e->goal_q, e->goal_pose6);
  }
  kp1o_env_capture_entry_metrics(e);
  if (obs) kp1o_env_observe(e, obs);
}

/* KP1/envs/arm_kinematic_env.py:489-506 _interpolate_dock_control_value */
static double interpolate_control(double pos, double near_t, double far_t, double near_v, double far_v, double fallback) {
  if (near_t <= 0.0 || far_t <= near_t) return fallback;
  if (pos <= near_t) return near_v;
  if (pos >= far_t) return far_v;
  double alpha = (pos - near_t) / maxd(far_t - near_t, 1e-9);
  return near_v + alpha * (far_v - near_v);
}

/* ---- reward component name tables (dict insertion order of the reference) ---- */
static const char* APPROACH_COMPONENTS[] = {
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
static const char* DOCK_COMPONENTS[] = {
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
#define N_APPROACH_COMPONENTS ((int)(sizeof(APPROACH_COMPONENTS) / sizeof(APPROACH_COMPONENTS[0])))
#define N_DOCK_COMPONENTS ((int)(sizeof(DOCK_COMPONENTS) / sizeof(DOCK_COMPONENTS[0])))
int kp1o_num_components(int mode) { return mode == KP1_MODE_DOCK ? N_DOCK_COMPONENTS : N_APPROACH_COMPONENTS; }
const char* kp1o_component_name(int mode, int i) {
  if (i < 0 || i >= kp1o_num_components(mode)) return "";
  return mode == KP1_MODE_DOCK ? DOCK_COMPONENTS[i] : APPROACH_COMPONENTS[i];
}

typedef struct reward_in {
  const double *prev_pose6, *curr_pose6, *goal_pose6, *action, *prev_action;
  int curr_in_pre_near_goal, prev_in_near_goal, curr_in_near_goal;
  int dwell_count, near_goal_entry_count, near_goal_drift_count;
  double joint_limit_margin_min;
  int success;
  double dq_norm, prev_dq_norm, delta_q_change_l2;
  double entry_pos, entry_ori, entry_action, entry_dq;
} reward_in;

static double mean_sq7(const double* a) {
  double s = 0.0;
  for (int i = 0; i < NJ; ++i) s += a[i] * a[i];
  return s / 7.0;
}
static double mean_sq_diff7(const double* a, const double* b) {
  double s = 0.0;
  for (int i = 0; i < NJ; ++i) s += (a[i] - b[i]) * (a[i] - b[i]);
  return s / 7.0;
}

/* KP1/envs/reward_approach.py:75-373 compute_approach_reward */
static double compute_approach_reward(const kp1_approach_reward* cfg, const reward_in* in, double* c) {
  double ppe[3], poe[3], cpe[3], coe[3];
  kp1o_pose_error(in->prev_pose6, in->goal_pose6, ppe, poe);
  kp1o_pose_error(in->curr_pose6, in->goal_pose6, cpe, coe);
  double prev_pos = norm_n(ppe, 3), curr_pos = norm_n(cpe, 3), prev_ori = norm_n(poe, 3), curr_ori = norm_n(coe, 3);
  int pre = in->curr_in_pre_near_goal, cn = in->curr_in_near_goal, pn = in->prev_in_near_goal;

  double position_progress = cfg->position_progress_weight * (prev_pos - curr_pos);
  double global_ori_progress = cfg->orientation_progress_weight * (prev_ori - curr_ori);
  double nf_ori_progress = pre ? cfg->near_field_orientation_progress_weight * (prev_ori - curr_ori) : 0.0;
  double orientation_progress = global_ori_progress + nf_ori_progress;
  double milestone = 0.0;
  if (pre)
    for (int i = 0; i < cfg->n_orientation_milestones; ++i)
      if (curr_ori <= cfg->orientation_milestone_thresholds_rad[i]) milestone += cfg->orientation_milestone_bonuses[i];
  double nf_center = pre ? -cfg->near_field_orientation_center_weight * curr_ori : 0.0;
  double pre_near_goal = (pre && !cn) ? cfg->pre_near_goal_bonus : 0.0;
  double near_goal_bonus_scale = pow(cfg->near_goal_bonus_decay, (double)maxi(in->near_goal_entry_count - 1, 0));
  double near_goal = (cn && !pn) ? cfg->near_goal_bonus * near_goal_bonus_scale : 0.0;
  double inner_progress = (pre && !cn) ? cfg->pre_near_to_near_progress_weight * maxd(prev_pos - curr_pos, 0.0) : 0.0;
  double coarse_bonus = (pre && curr_ori <= cfg->coarse_orientation_bonus_threshold_rad) ? cfg->coarse_orientation_bonus : 0.0;
  int curr_ho = cfg->handover_pos_threshold_m > 0.0 && curr_pos <= cfg->handover_pos_threshold_m &&
                (cfg->handover_ori_threshold_rad <= 0.0 || curr_ori <= cfg->handover_ori_threshold_rad);
  int prev_ho = cfg->handover_pos_threshold_m > 0.0 && prev_pos <= cfg->handover_pos_threshold_m &&
                (cfg->handover_ori_threshold_rad <= 0.0 || prev_ori <= cfg->handover_ori_threshold_rad);
  double ho_bonus = (curr_ho && !prev_ho) ? cfg->handover_bonus : 0.0;
  double ho_ret = (curr_ho && prev_ho) ? cfg->handover_retention_bonus : 0.0;
  double ho_dwell = (curr_ho && in->dwell_count >= 2) ? cfg->handover_dwell_bonus : 0.0;
  double ho_leave = (prev_ho && !curr_ho) ? -cfg->handover_leave_penalty : 0.0;
  double regress = maxd(curr_pos - prev_pos, 0.0) + maxd(curr_ori - prev_ori, 0.0);
  double ho_regr = (prev_ho || curr_ho) ? -cfg->handover_regression_weight * regress : 0.0;
  double dwell = (cn && in->dwell_count >= 2) ? cfg->dwell_bonus : 0.0;
  int drift_esc = maxi(in->near_goal_drift_count - cfg->drift_penalty_escalation_start, 0);
  double drift_scale = 1.0 + cfg->drift_penalty_escalation_per_count * (double)drift_esc;
  double drift_w = cfg->drift_penalty_weight * drift_scale;
  double drift_penalty = pn ? -drift_w * maxd(curr_pos - prev_pos, 0.0) : 0.0;
  double leave_penalty = (pn && !cn) ? -cfg->near_goal_leave_penalty : 0.0;
  double action_norm = norm_n(in->action, NJ), prev_action_norm = norm_n(in->prev_action, NJ);
  double dqn = in->dq_norm, pdqn = in->prev_dq_norm;
  int dc_enabled = cfg->dock_coarse_ready_pos_threshold_m > 0.0 && cfg->dock_coarse_ready_ori_threshold_rad > 0.0;
  int curr_dc_pose = dc_enabled && curr_pos <= cfg->dock_coarse_ready_pos_threshold_m && curr_ori <= cfg->dock_coarse_ready_ori_threshold_rad;
  int prev_dc_pose = dc_enabled && prev_pos <= cfg->dock_coarse_ready_pos_threshold_m && prev_ori <= cfg->dock_coarse_ready_ori_threshold_rad;
  int curr_dc_motion = (cfg->dock_coarse_ready_action_threshold <= 0.0 || action_norm <= cfg->dock_coarse_ready_action_threshold) &&
                       (cfg->dock_coarse_ready_dq_threshold <= 0.0 || dqn <= cfg->dock_coarse_ready_dq_threshold);
  int prev_dc_motion = (cfg->dock_coarse_ready_action_threshold <= 0.0 || prev_action_norm <= cfg->dock_coarse_ready_action_threshold) &&
                       (cfg->dock_coarse_ready_dq_threshold <= 0.0 || pdqn <= cfg->dock_coarse_ready_dq_threshold);
  int curr_dc = curr_dc_pose && curr_dc_motion, prev_dc = prev_dc_pose && prev_dc_motion;
  int fr_enabled = cfg->finisher_ready_pos_threshold_m > 0.0 && cfg->finisher_ready_ori_threshold_rad > 0.0;
  int curr_fr_pose = fr_enabled && curr_pos <= cfg->finisher_ready_pos_threshold_m && curr_ori <= cfg->finisher_ready_ori_threshold_rad;
  int prev_fr_pose = fr_enabled && prev_pos <= cfg->finisher_ready_pos_threshold_m && prev_ori <= cfg->finisher_ready_ori_threshold_rad;
  int curr_fr_motion = (cfg->finisher_ready_action_threshold <= 0.0 || action_norm <= cfg->finisher_ready_action_threshold) &&
                       (cfg->finisher_ready_dq_threshold <= 0.0 || dqn <= cfg->finisher_ready_dq_threshold);
  int prev_fr_motion = (cfg->finisher_ready_action_threshold <= 0.0 || prev_action_norm <= cfg->finisher_ready_action_threshold) &&
                       (cfg->finisher_ready_dq_threshold <= 0.0 || pdqn <= cfg->finisher_ready_dq_threshold);
  int curr_fr = curr_fr_pose && curr_fr_motion, prev_fr = prev_fr_pose && prev_fr_motion;
  int nh = cfg->near_handoff_pos_threshold_m > 0.0 && cfg->near_handoff_ori_threshold_rad > 0.0 &&
           curr_pos <= cfg->near_handoff_pos_threshold_m && curr_ori <= cfg->near_handoff_ori_threshold_rad;
  int prev_nh = cfg->near_handoff_pos_threshold_m > 0.0 && cfg->near_handoff_ori_threshold_rad > 0.0 &&
                prev_pos <= cfg->near_handoff_pos_threshold_m && prev_ori <= cfg->near_handoff_ori_threshold_rad;
  double dc_bonus = (curr_dc && !prev_dc) ? cfg->dock_coarse_ready_bonus : 0.0;
  double dc_ret = (curr_dc && prev_dc) ? cfg->dock_coarse_ready_retention_bonus : 0.0;
  double dc_dwell = (curr_dc && in->dwell_count >= 2) ? cfg->dock_coarse_ready_dwell_bonus : 0.0;
  double dc_leave = (prev_dc && !curr_dc) ? -cfg->dock_coarse_ready_leave_penalty : 0.0;
  double dc_regr = (nh || prev_nh || curr_dc_pose || prev_dc_pose) ? -cfg->dock_coarse_ready_regression_weight * regress : 0.0;
  double fr_bonus = (curr_fr && !prev_fr) ? cfg->finisher_ready_bonus : 0.0;
  double fr_ret = (curr_fr && prev_fr) ? cfg->finisher_ready_retention_bonus : 0.0;
  double fr_dwell = (curr_fr && in->dwell_count >= 2) ? cfg->finisher_ready_dwell_bonus : 0.0;
  double fr_leave = (prev_fr && !curr_fr) ? -cfg->finisher_ready_leave_penalty : 0.0;
  double fr_regr = (nh || prev_nh || curr_fr_pose || prev_fr_pose) ? -cfg->finisher_ready_regression_weight * regress : 0.0;
  int nh_any = nh || curr_dc_pose || curr_fr_pose;
  double msq = mean_sq7(in->action);
  double nh_action = nh_any ? -cfg->near_handoff_action_weight * msq : 0.0;
  double nh_dq = nh_any ? -cfg->near_handoff_dq_weight * dqn : 0.0;
  double nh_motion = 0.0, nh_settle = 0.0;
  if (nh_any) {
    double at = cfg->finisher_ready_action_threshold != 0.0 ? cfg->finisher_ready_action_threshold : cfg->dock_coarse_ready_action_threshold;
    double dt = cfg->finisher_ready_dq_threshold != 0.0 ? cfg->finisher_ready_dq_threshold : cfg->dock_coarse_ready_dq_threshold;
    double as = maxd(at, 1e-9), ds = maxd(dt, 1e-9);
    double action_clean = at > 0 ? maxd(1.0 - action_norm / as, 0.0) : 0.0;
    double dq_clean = dt > 0 ? maxd(1.0 - dqn / ds, 0.0) : 0.0;
    nh_motion = cfg->near_handoff_motion_bonus_weight * (0.5 * action_clean + 0.5 * dq_clean);
    nh_settle = cfg->near_handoff_settle_bonus_weight * (0.5 * maxd(prev_action_norm - action_norm, 0.0) + 0.5 * maxd(pdqn - dqn, 0.0));
  }
  double same_step = (curr_pos < prev_pos && curr_ori < prev_ori && (pre || nh)) ? cfg->same_step_alignment_bonus : 0.0;
  double smooth_mult = (curr_ho || prev_ho) ? cfg->handover_smoothness_multiplier : 1.0;
  double smooth = smooth_mult * (-cfg->action_magnitude_weight * msq - cfg->action_delta_weight * mean_sq_diff7(in->action, in->prev_action));
  double jl_pen = -cfg->joint_limit_penalty_weight * (maxd(0.25 - in->joint_limit_margin_min, 0.0) / 0.25);
  double success_bonus = in->success ? cfg->success_bonus : 0.0;

  int k = 0;
  c[k++] = position_progress; c[k++] = global_ori_progress; c[k++] = nf_ori_progress; c[k++] = orientation_progress;
  c[k++] = milestone; c[k++] = nf_center; c[k++] = pre_near_goal; c[k++] = near_goal; c[k++] = inner_progress;
  c[k++] = (cn && !pn) ? near_goal_bonus_scale : 0.0; c[k++] = coarse_bonus; c[k++] = ho_bonus; c[k++] = ho_ret;
  c[k++] = ho_dwell; c[k++] = ho_leave; c[k++] = ho_regr; c[k++] = dc_bonus; c[k++] = dc_ret; c[k++] = dc_dwell;
  c[k++] = dc_leave; c[k++] = dc_regr; c[k++] = fr_bonus; c[k++] = fr_ret; c[k++] = fr_dwell; c[k++] = fr_leave;
  c[k++] = fr_regr; c[k++] = nh_action; c[k++] = nh_dq; c[k++] = nh_motion; c[k++] = nh_settle; c[k++] = same_step;
  c[k++] = dwell; c[k++] = drift_penalty; c[k++] = leave_penalty; c[k++] = drift_scale;
  c[k++] = (double)in->near_goal_entry_count; c[k++] = (double)in->near_goal_drift_count; c[k++] = smooth;
  c[k++] = smooth_mult; c[k++] = jl_pen; c[k++] = success_bonus; c[k++] = curr_pos; c[k++] = curr_ori;
  c[k++] = action_norm; c[k++] = dqn; c[k++] = (double)in->dwell_count; c[k++] = (double)pre; c[k++] = (double)cn;
  c[k++] = (double)curr_ho; c[k++] = (double)curr_dc; c[k++] = (double)curr_dc_pose; c[k++] = (double)curr_fr;
  c[k++] = (double)curr_fr_pose; c[k++] = (double)nh;
  /* reward_approach.py:334-372, summed left to right from 0 */
  double r = 0.0;
  r += position_progress; r += orientation_progress; r += milestone; r += nf_center; r += pre_near_goal; r += near_goal;
  r += inner_progress; r += coarse_bonus; r += ho_bonus; r += ho_ret; r += ho_dwell; r += ho_leave; r += ho_regr;
  r += dc_bonus; r += dc_ret; r += dc_dwell; r += dc_leave; r += dc_regr; r += fr_bonus; r += fr_ret; r += fr_dwell;
  r += fr_leave; r += fr_regr; r += nh_action; r += nh_dq; r += nh_motion; r += nh_settle; r += same_step; r += dwell;
  r += drift_penalty; r += leave_penalty; r += smooth; r += jl_pen; r += success_bonus;
  return r;
}

/* KP1/envs/reward_dock.py:105-120 */
static double interpolate_entry_penalty_scale(double pos, double near_t, double far_t, double near_m, double far_m) {
  if (near_t <= 0.0 || far_t <= near_t) return 1.0;
  if (pos <= near_t) return near_m;
  if (pos >= far_t) return far_m;
  double alpha = (pos - near_t) / maxd(far_t - near_t, 1e-9);
  return near_m + alpha * (far_m - near_m);
}

/* KP1/envs/reward_dock.py:123-484 compute_dock_reward */
static double compute_dock_reward(const kp1_dock_reward* cfg, const reward_in* in, double* c) {
  double ppe[3], poe[3], cpe[3], coe[3];
  kp1o_pose_error(in->prev_pose6, in->goal_pose6, ppe, poe);
  kp1o_pose_error(in->curr_pose6, in->goal_pose6, cpe, coe);
  double prev_pos = norm_n(ppe, 3), curr_pos = norm_n(cpe, 3), prev_ori = norm_n(poe, 3), curr_ori = norm_n(coe, 3);
  int cn = in->curr_in_near_goal, pn = in->prev_in_near_goal, dwell_count = in->dwell_count;

  double position_progress = cfg->position_progress_weight * (prev_pos - curr_pos);
  double orientation_progress = cfg->orientation_progress_weight * (prev_ori - curr_ori);
  double stay = cn ? cfg->stay_in_zone_bonus : 0.0;
  double dwell_bonus = cn ? cfg->dwell_bonus * (double)maxi(dwell_count - 1, 0) : 0.0;
  double wr_bonus = cn ? cfg->working_range_bonus : 0.0;
  double wr_dwell = (cn && dwell_count >= cfg->working_range_dwell_start)
                        ? cfg->working_range_dwell_bonus * (double)maxi(dwell_count - cfg->working_range_dwell_start + 1, 0) : 0.0;
  int curr_tight = curr_pos <= cfg->tight_pose_pos_threshold_m && curr_ori <= cfg->tight_pose_ori_threshold_rad;
  int prev_tight = prev_pos <= cfg->tight_pose_pos_threshold_m && prev_ori <= cfg->tight_pose_ori_threshold_rad;
  double ns_pos_t = cfg->near_strict_pos_threshold_m != 0.0 ? cfg->near_strict_pos_threshold_m : cfg->tight_pose_pos_threshold_m * 2.0;
  double ns_ori_t = cfg->near_strict_ori_threshold_rad != 0.0 ? cfg->near_strict_ori_threshold_rad : cfg->tight_pose_ori_threshold_rad * 3.0;
  int curr_ns = curr_pos <= ns_pos_t && curr_ori <= ns_ori_t;
  int prev_ns = prev_pos <= ns_pos_t && prev_ori <= ns_ori_t;
  double spc = maxd(1.0 - curr_pos / maxd(cfg->tight_pose_pos_threshold_m, 1e-9), 0.0);
  double soc = maxd(1.0 - curr_ori / maxd(cfg->tight_pose_ori_threshold_rad, 1e-9), 0.0);
  double strict_closeness = pow(0.8 * spc + 0.2 * soc, 2.0);
  double tight_bonus = curr_tight ? cfg->tight_pose_bonus : 0.0;
  double tight_dwell = curr_tight ? cfg->tight_pose_dwell_bonus * (double)maxi(dwell_count - 1, 0) : 0.0;
  double strict_leave = (prev_tight && !curr_tight) ? -cfg->strict_pose_leave_penalty : 0.0;
  double sc_reward = curr_tight ? cfg->strict_center_reward_weight * strict_closeness : 0.0;
  double sc_pos_pen = cfg->strict_center_position_weight > 0.0
                          ? -cfg->strict_center_position_weight * pow(curr_pos / maxd(cfg->tight_pose_pos_threshold_m, 1e-9), 2.0) : 0.0;
  double sc_ori_pen = cfg->strict_center_orientation_weight > 0.0
                          ? -cfg->strict_center_orientation_weight * pow(curr_ori / maxd(cfg->tight_pose_ori_threshold_rad, 1e-9), 2.0) : 0.0;
  double action_rms = sqrt(mean_sq7(in->action));
  double sc_small = 0.0;
  if (cfg->strict_center_small_action_bonus_weight > 0.0 && cfg->strict_center_small_action_pos_radius_m > 0.0 &&
      cfg->strict_center_small_action_ori_radius_rad > 0.0 && cfg->strict_center_small_action_scale > 0.0) {
    double cpc = maxd(1.0 - curr_pos / cfg->strict_center_small_action_pos_radius_m, 0.0);
    double coc = maxd(1.0 - curr_ori / cfg->strict_center_small_action_ori_radius_rad, 0.0);
    double cc = pow(0.8 * cpc + 0.2 * coc, cfg->strict_center_small_action_power);
    double sm = maxd(1.0 - action_rms / cfg->strict_center_small_action_scale, 0.0);
    sc_small = curr_tight ? cfg->strict_center_small_action_bonus_weight * cc * sm : 0.0;
  }
  double sc_dwell = 0.0;
  if (curr_tight && cfg->strict_center_dwell_bonus_weight > 0.0 && dwell_count >= cfg->strict_center_dwell_start) {
    int esc = maxi(dwell_count - cfg->strict_center_dwell_escalation_start, 0);
    double scale = 1.0 + cfg->strict_center_dwell_escalation_per_step * (double)esc;
    sc_dwell = cfg->strict_center_dwell_bonus_weight * strict_closeness * scale;
  }
  double tps = cfg->tight_position_shaping_radius_m > 0.0
                   ? cfg->tight_position_shaping_weight * maxd(1.0 - curr_pos / maxd(cfg->tight_position_shaping_radius_m, 1e-9), 0.0) : 0.0;
  double tos = cfg->tight_orientation_shaping_radius_rad > 0.0
                   ? cfg->tight_orientation_shaping_weight * maxd(1.0 - curr_ori / maxd(cfg->tight_orientation_shaping_radius_rad, 1e-9), 0.0) : 0.0;
  double conv_pos = (cfg->convergence_position_radius_m > 0.0 && mind(prev_pos, curr_pos) <= cfg->convergence_position_radius_m)
                        ? cfg->convergence_position_progress_weight * (prev_pos - curr_pos) : 0.0;
  double gate_scale = (cfg->position_first_orientation_pos_threshold_m > 0.0 && curr_pos > cfg->position_first_orientation_pos_threshold_m)
                          ? cfg->position_first_orientation_pre_scale : 1.0;
  double conv_ori = (cfg->convergence_orientation_radius_rad > 0.0 && mind(prev_ori, curr_ori) <= cfg->convergence_orientation_radius_rad)
                        ? gate_scale * cfg->convergence_orientation_progress_weight * (prev_ori - curr_ori) : 0.0;
  double leave_zone = (pn && !cn) ? -cfg->leave_zone_penalty : 0.0;
  double wr_exit = (pn && !cn) ? -cfg->working_range_exit_penalty : 0.0;
  double drift = -cfg->drift_penalty_position_weight * maxd(curr_pos - prev_pos, 0.0);
  drift += -cfg->drift_penalty_orientation_weight * maxd(curr_ori - prev_ori, 0.0);
  if (curr_tight || prev_tight) drift *= cfg->strict_zone_drift_penalty_multiplier;
  double action_l2 = norm_n(in->action, NJ);
  double eps_scale = interpolate_entry_penalty_scale(maxd(prev_pos, curr_pos), cfg->entry_action_penalty_near_pos_threshold_m,
                                                     cfg->entry_action_penalty_far_pos_threshold_m,
                                                     cfg->entry_action_penalty_near_multiplier, cfg->entry_action_penalty_far_multiplier);
  double msd = mean_sq_diff7(in->action, in->prev_action);
  double smooth = -cfg->action_magnitude_weight * mean_sq7(in->action);
  smooth += -cfg->action_delta_weight * msd;
  if (curr_tight) smooth *= cfg->strict_zone_action_penalty_multiplier;
  smooth *= eps_scale;
  double action_delta_rms = sqrt(msd);
  double adv = (cfg->action_delta_violation_weight > 0.0 && cfg->action_delta_violation_threshold > 0.0)
                   ? -cfg->action_delta_violation_weight * eps_scale * maxd(action_delta_rms - cfg->action_delta_violation_threshold, 0.0) : 0.0;
  double dqc = (cfg->delta_q_change_penalty_weight > 0.0 && cfg->delta_q_change_penalty_threshold > 0.0)
                   ? -cfg->delta_q_change_penalty_weight * eps_scale * maxd(in->delta_q_change_l2 - cfg->delta_q_change_penalty_threshold, 0.0) : 0.0;
  double entry_pos = in->entry_pos, entry_ori = in->entry_ori, entry_action = in->entry_action, entry_dq = in->entry_dq;
  double preserve = 0.0;
  if (cfg->preserve_state_bonus > 0.0 && (curr_ns || curr_tight)) {
    int pos_ok = curr_pos <= entry_pos + cfg->preserve_position_tolerance_m;
    int ori_ok = curr_ori <= entry_ori + cfg->preserve_orientation_tolerance_rad;
    if (pos_ok && ori_ok) preserve = cfg->preserve_state_bonus;
  }
  double strict_hold = curr_tight ? cfg->strict_hold_bonus * (double)maxi(dwell_count - 1, 0) : 0.0;
  double low_motion = 0.0;
  if (cfg->low_motion_bonus > 0.0 && curr_ns && (cfg->low_motion_action_threshold <= 0.0 || action_l2 <= cfg->low_motion_action_threshold) &&
      (cfg->low_motion_dq_threshold <= 0.0 || in->dq_norm <= cfg->low_motion_dq_threshold))
    low_motion = cfg->low_motion_bonus;
  double tiny = 0.0;
  if (cfg->tiny_correction_bonus > 0.0 && curr_ns && !curr_tight) {
    int improved = curr_pos <= prev_pos && curr_ori <= prev_ori;
    int small = cfg->tiny_correction_action_threshold <= 0.0 || action_l2 <= cfg->tiny_correction_action_threshold;
    if (improved && small) tiny = cfg->tiny_correction_bonus;
  }
  double worse = 0.0;
  worse += -cfg->worse_than_entry_position_weight * maxd(curr_pos - entry_pos - cfg->worse_than_entry_position_tolerance_m, 0.0);
  worse += -cfg->worse_than_entry_orientation_weight * maxd(curr_ori - entry_ori - cfg->worse_than_entry_orientation_tolerance_rad, 0.0);
  double ns_regr = 0.0;
  if (curr_ns || prev_ns)
    ns_regr = -cfg->near_strict_regression_multiplier * (cfg->drift_penalty_position_weight * maxd(curr_pos - prev_pos, 0.0) +
                                                         cfg->drift_penalty_orientation_weight * maxd(curr_ori - prev_ori, 0.0));
  double agg_scale = curr_ns ? cfg->near_strict_action_penalty_multiplier : 1.0;
  double aggressive = (cfg->aggressive_action_weight > 0.0 && cfg->aggressive_action_threshold > 0.0)
                          ? -cfg->aggressive_action_weight * agg_scale * maxd(action_l2 - cfg->aggressive_action_threshold, 0.0) : 0.0;
  double dqp_scale = curr_ns ? cfg->near_strict_dq_penalty_multiplier : 1.0;
  double dq_pen = (cfg->dq_penalty_weight > 0.0 && cfg->dq_penalty_threshold > 0.0)
                      ? -cfg->dq_penalty_weight * dqp_scale * maxd(in->dq_norm - cfg->dq_penalty_threshold, 0.0) : 0.0;
  double jl_pen = -cfg->joint_limit_penalty_weight * (maxd(0.25 - in->joint_limit_margin_min, 0.0) / 0.25);
  double success_bonus = in->success ? cfg->success_bonus : 0.0;
  double b_outer = 0, b_inner = 0, b_dwell = 0, b_outer_exit = 0, b_inner_exit = 0, b_dwell_break = 0, b_drift = 0;
  int zone = 0;
  if (cfg->basin_outer_radius_m > 0.0 && cfg->basin_inner_radius_m > 0.0 && cfg->basin_dwell_radius_m > 0.0) {
    double outer_r = maxd(cfg->basin_outer_radius_m, 1e-9), inner_r = maxd(cfg->basin_inner_radius_m, 1e-9), dwell_r = maxd(cfg->basin_dwell_radius_m, 1e-9);
    int p_o = prev_pos <= outer_r, p_i = prev_pos <= inner_r, p_d = prev_pos <= dwell_r;
    int c_o = curr_pos <= outer_r, c_i = curr_pos <= inner_r, c_d = curr_pos <= dwell_r;
    zone = c_d ? 3 : (c_i ? 2 : (c_o ? 1 : 0));
    if (c_o) b_outer = cfg->basin_outer_bonus * (1.0 + maxd(1.0 - curr_pos / outer_r, 0.0));
    if (c_i) b_inner = cfg->basin_inner_bonus * (1.0 + maxd(1.0 - curr_pos / inner_r, 0.0));
    if (c_d) b_dwell = cfg->basin_dwell_bonus * (1.0 + maxd(1.0 - curr_pos / dwell_r, 0.0));
    b_outer_exit = (p_o && !c_o) ? -cfg->basin_outer_exit_penalty : 0.0;
    b_inner_exit = (p_i && !c_i) ? -cfg->basin_inner_exit_penalty : 0.0;
    b_dwell_break = (p_d && !c_d) ? -cfg->basin_dwell_break_penalty : 0.0;
    b_drift = (p_o || c_o) ? -cfg->basin_drift_penalty_weight * maxd(curr_pos - prev_pos, 0.0) : 0.0;
  }
  int k = 0;
  c[k++] = position_progress; c[k++] = orientation_progress; c[k++] = stay; c[k++] = dwell_bonus; c[k++] = wr_bonus;
  c[k++] = wr_dwell; c[k++] = tight_bonus; c[k++] = tight_dwell; c[k++] = strict_leave; c[k++] = sc_reward;
  c[k++] = sc_pos_pen; c[k++] = sc_ori_pen; c[k++] = sc_small; c[k++] = sc_dwell; c[k++] = tps; c[k++] = tos;
  c[k++] = conv_pos; c[k++] = conv_ori; c[k++] = gate_scale; c[k++] = eps_scale; c[k++] = leave_zone; c[k++] = wr_exit;
  c[k++] = drift; c[k++] = smooth; c[k++] = adv; c[k++] = dqc; c[k++] = preserve; c[k++] = strict_hold; c[k++] = low_motion;
  c[k++] = tiny; c[k++] = worse; c[k++] = ns_regr; c[k++] = aggressive; c[k++] = dq_pen; c[k++] = jl_pen;
  c[k++] = success_bonus; c[k++] = b_outer; c[k++] = b_inner; c[k++] = b_dwell; c[k++] = b_outer_exit; c[k++] = b_inner_exit;
  c[k++] = b_dwell_break; c[k++] = b_drift; c[k++] = (double)zone; c[k++] = curr_pos; c[k++] = curr_ori;
  c[k++] = (double)dwell_count; c[k++] = (double)curr_tight; c[k++] = (double)curr_ns; c[k++] = entry_pos; c[k++] = entry_ori;
  c[k++] = entry_action; c[k++] = entry_dq; c[k++] = curr_pos - entry_pos; c[k++] = curr_ori - entry_ori;
  c[k++] = action_l2 - entry_action; c[k++] = in->dq_norm - entry_dq; c[k++] = (double)in->near_goal_entry_count;
  c[k++] = (double)in->near_goal_drift_count; c[k++] = (double)cn;
  /* reward_dock.py:438-483 */
  double r = 0.0;
  r += position_progress; r += orientation_progress; r += stay; r += dwell_bonus; r += wr_bonus; r += wr_dwell;
  r += tight_bonus; r += tight_dwell; r += strict_leave; r += sc_reward; r += sc_pos_pen; r += sc_ori_pen; r += sc_small;
  r += sc_dwell; r += tps; r += tos; r += conv_pos; r += conv_ori; r += leave_zone; r += wr_exit; r += drift; r += smooth;
  r += adv; r += dqc; r += preserve; r += strict_hold; r += low_motion; r += tiny; r += worse; r += ns_regr;
  r += aggressive; r += dq_pen; r += jl_pen; r += success_bonus; r += b_outer; r += b_inner; r += b_dwell;
  r += b_outer_exit; r += b_inner_exit; r += b_dwell_break; r += b_drift;
  return r;
}

/* KP1/envs/arm_kinematic_env.py:213-365 step */
void kp1o_env_step(kp1o_env* e, const double action_in[7], float obs[KP1_OBS_DIM], kp1o_step_out* out) {
  const kp1_config* cfg = &e->cfg;
  const kp1_env_scalars* ec = &cfg->env;
  double action[7], prev_pose6[6], prev_action[7], pe[3], oe[3];
  for (int i = 0; i < NJ; ++i) action[i] = clipd(action_in[i], -1.0, 1.0);
  memcpy(prev_pose6, e->ee_pose6, sizeof prev_pose6);
  memcpy(prev_action, e->prev_action, sizeof prev_action);
  kp1o_pose_error(prev_pose6, e->goal_pose6, pe, oe);
  double prev_pos_norm = norm_n(pe, 3), prev_ori_norm = norm_n(oe, 3);
  int dock = e->policy_mode == KP1_MODE_DOCK;
  double dyn_limit = clipd(ec->dock_residual_action_limit, 0.0, 1.0);
  double dyn_dqc = maxd(ec->dock_delta_q_change_limit_scale, 0.0);
  if (dock) {
    /* :508-528 */
    dyn_limit = clipd(interpolate_control(prev_pos_norm, ec->dock_dynamic_action_limit_near_pos_threshold_m,
                                          ec->dock_dynamic_action_limit_far_pos_threshold_m, ec->dock_dynamic_residual_action_limit_near,
                                          ec->dock_dynamic_residual_action_limit_far, ec->dock_residual_action_limit), 0.0, 1.0);
    dyn_dqc = maxd(interpolate_control(prev_pos_norm, ec->dock_dynamic_action_limit_near_pos_threshold_m,
                                       ec->dock_dynamic_action_limit_far_pos_threshold_m, ec->dock_dynamic_delta_q_change_limit_scale_near,
                                       ec->dock_dynamic_delta_q_change_limit_scale_far, ec->dock_delta_q_change_limit_scale), 0.0);
    for (int i = 0; i < NJ; ++i) action[i] = clipd(action[i], -dyn_limit, dyn_limit);
  }
  int prev_in_near = is_near_goal(e, prev_pos_norm, prev_ori_norm);
  double scale = ec->action_delta_scale;
  if (dock && ec->dock_action_delta_scale > 0.0) {
    scale = ec->dock_action_delta_scale;
  } else if (!dock) {
    /* :530-542 */
    scale = ec->action_delta_scale;
    if (ec->dynamic_action_delta_scale_enabled) {
      double mult = interpolate_control(prev_pos_norm, ec->dynamic_action_delta_scale_near_pos_threshold_m,
                                        ec->dynamic_action_delta_scale_far_pos_threshold_m, ec->dynamic_action_delta_scale_near_multiplier,
                                        ec->dynamic_action_delta_scale_far_multiplier, 1.0);
      scale = ec->action_delta_scale * maxd(mult, 0.0);
    }
  }
  double max_dq[7], dq_cmd[7], q_next[7], dq_next[7], t[7];
  for (int i = 0; i < NJ; ++i) max_dq[i] = cfg->joints.delta_limit[i] * scale;
  for (int i = 0; i < NJ; ++i) dq_cmd[i] = action[i] * max_dq[i];
  if (dock && dyn_dqc > 0.0) {
    for (int i = 0; i < NJ; ++i) {
      double lim = max_dq[i] * dyn_dqc;
      dq_cmd[i] = e->dq[i] + clipd(dq_cmd[i] - e->dq[i], -lim, lim);
      dq_cmd[i] = clipd(dq_cmd[i], -max_dq[i], max_dq[i]);
    }
  }
  for (int i = 0; i < NJ; ++i) t[i] = e->q[i] + dq_cmd[i];
  kp1o_clip_q(&cfg->joints, t, q_next);
  for (int i = 0; i < NJ; ++i) dq_next[i] = q_next[i] - e->q[i];
  for (int i = 0; i < NJ; ++i) t[i] = dq_next[i] - e->dq[i];
  double dq_change_l2 = norm_n(t, NJ);
  double ee_next[6];
  kp1o_fk_pose6(q_next, ee_next);
  kp1o_pose_error(ee_next, e->goal_pose6, pe, oe);
  double curr_pos_norm = norm_n(pe, 3), curr_ori_norm = norm_n(oe, 3);
  int curr_pre = is_pre_near_goal(e, curr_pos_norm, curr_ori_norm);
  int curr_near = is_near_goal(e, curr_pos_norm, curr_ori_norm);
  e->min_pos_error = fmin(e->min_pos_error, curr_pos_norm);
  if (curr_pre) e->pre_near_goal_hit = 1;
  if (curr_near && !prev_in_near) e->near_goal_entry_count += 1;
  if (curr_near) e->dwell_count += 1;
  else e->dwell_count = 0;
  if (prev_in_near && curr_pos_norm > prev_pos_norm) e->near_goal_drift_count += 1;
  /* KP1/envs/termination.py:20-57 */
  const kp1_termination* tc = &cfg->termination;
  int step_count = e->episode_step + 1;
  int terminated = 0, truncated = 0, success = 0, invalid = 0;
  int criteria = curr_pos_norm <= tc->success_pos_threshold_m &&
                 (!tc->require_orientation || curr_ori_norm <= tc->success_ori_threshold_rad) &&
                 e->dwell_count >= tc->success_dwell_steps;
  if (!isfinite(curr_pos_norm) || !isfinite(curr_ori_norm)) {
    terminated = 1;
    invalid = 1;
  } else if (criteria) {
    success = 1;
    if (tc->terminate_on_success) terminated = 1;
  }
  if (!terminated && step_count >= tc->max_episode_steps) truncated = 1;

  double margin[7], margin_min;
  kp1o_joint_limit_margin(&cfg->joints, q_next, margin);
  margin_min = margin[0];
  for (int i = 1; i < NJ; ++i) margin_min = mind(margin_min, margin[i]);
  reward_in ri;
  ri.prev_pose6 = prev_pose6; ri.curr_pose6 = ee_next; ri.goal_pose6 = e->goal_pose6;
  ri.action = action; ri.prev_action = prev_action;
  ri.curr_in_pre_near_goal = curr_pre; ri.prev_in_near_goal = prev_in_near; ri.curr_in_near_goal = curr_near;
  ri.dwell_count = e->dwell_count; ri.near_goal_entry_count = e->near_goal_entry_count;
  ri.near_goal_drift_count = e->near_goal_drift_count; ri.joint_limit_margin_min = margin_min; ri.success = success;
  ri.dq_norm = norm_n(dq_next, NJ); ri.prev_dq_norm = norm_n(e->dq, NJ); ri.delta_q_change_l2 = dq_change_l2;
  ri.entry_pos = e->entry_position_error_norm; ri.entry_ori = e->entry_orientation_error_norm;
  ri.entry_action = e->entry_action_l2; ri.entry_dq = e->entry_dq_norm;
  memset(out->components, 0, sizeof out->components);
  if (dock) {
    out->reward = compute_dock_reward(&cfg->dock_reward, &ri, out->components);
    out->n_components = N_DOCK_COMPONENTS;
  } else {
    out->reward = compute_approach_reward(&cfg->reward, &ri, out->components);
    out->n_components = N_APPROACH_COMPONENTS;
  }
  /* :344-350 commit */
  e->episode_step += 1;
  memcpy(e->q, q_next, sizeof e->q);
  memcpy(e->dq, dq_next, sizeof e->dq);
  memcpy(e->prev_action, action, sizeof e->prev_action);
  memcpy(e->ee_pose6, ee_next, sizeof e->ee_pose6);
  if (curr_near) e->near_goal_hit = 1;
  if (obs) kp1o_env_observe(e, obs);
  out->terminated = terminated;
  out->truncated = truncated;
  out->success = success;
  out->invalid = invalid;
  out->position_error_norm = curr_pos_norm;
  out->orientation_error_norm = curr_ori_norm;
  out->executed_delta_q_l2 = ri.dq_norm;
  out->action_l2 = norm_n(action, NJ);
  out->delta_q_change_l2 = dq_change_l2;
  out->dock_action_limit = dyn_limit;
  out->dock_delta_q_change_limit_scale = dyn_dqc;
  out->joint_limit_margin_min = margin_min;
}

/* ------------------------------------------------------------------ curriculum tracker */
/* KP1/envs/curriculum.py:104-154 PointCurriculumTracker == KP1/training/callbacks.py:71-92 per-episode rule */
void kp1o_tracker_init(kp1o_tracker* t, double threshold, int window, int min_episodes, int max_stage_index, int initial_stage) {
  memset(t, 0, sizeof *t);
  t->threshold = threshold;
  t->window = maxi(window, 1);
  if (t->window > 1024) t->window = 1024;
  t->min_episodes = min_episodes;
  t->max_stage_index = maxi(max_stage_index, 0);
  t->stage_index = clipi(initial_stage, 0, t->max_stage_index);
}
int kp1o_tracker_record(kp1o_tracker* t, int success) {
  t->stage_episode_count += 1;
  if (t->ring_len < t->window) {
    t->ring[(t->ring_head + t->ring_len) % t->window] = success ? 1 : 0;
    t->ring_len += 1;
  } else {
    t->ring[t->ring_head] = success ? 1 : 0;
    t->ring_head = (t->ring_head + 1) % t->window;
  }
  if (t->stage_index >= t->max_stage_index) return 0;
  if (t->stage_episode_count < t->min_episodes) return 0;
  if (t->ring_len < t->window) return 0;
  int s = 0;
  for (int i = 0; i < t->ring_len; ++i) s += t->ring[i];
  double rate = (double)s / (double)t->ring_len;
  if (rate < t->threshold) return 0;
  t->stage_index += 1;
  t->stage_episode_count = 0;
  t->ring_len = 0;
  t->ring_head = 0;
  t->last_trigger_rate = rate;
  return 1;
}

/* ------------------------------------------------------------------ batched CPU baseline */
int kp1o_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void kp1o_batch_step(kp1o_env* envs, int n, const double* actions, float* obs, double* reward, uint8_t* done,
                     int auto_reset, int n_threads) {
  kp1o_batch_step_components(envs, n, actions, obs, reward, done, auto_reset, n_threads, 0);
}

/* same, also returning every env's reward components [n][KP1O_MAX_COMPONENTS] (NULL = skip) */
void kp1o_batch_step_components(kp1o_env* envs, int n, const double* actions, float* obs, double* reward, uint8_t* done,
                                int auto_reset, int n_threads, double* components) {
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
  for (int i = 0; i < n; ++i) {
    kp1o_step_out so;
    kp1o_env_step(&envs[i], actions + 7 * (size_t)i, obs + KP1_OBS_DIM * (size_t)i, &so);
    reward[i] = so.reward;
    uint8_t d = (uint8_t)((so.terminated ? KP1_DONE_TERMINATED : 0) | (so.truncated ? KP1_DONE_TRUNCATED : 0) |
                          (so.success ? KP1_DONE_SUCCESS : 0) | (so.invalid ? KP1_DONE_INVALID : 0));
    done[i] = d;
    if (components) memcpy(components + (size_t)i * KP1O_MAX_COMPONENTS, so.components, sizeof so.components);
    if (auto_reset && (so.terminated || so.truncated)) kp1o_env_reset(&envs[i], 0, obs + KP1_OBS_DIM * (size_t)i);
  }
}

/* layout probes for the ctypes mirror in oracle/oracle.py */
size_t kp1o_sizeof_env(void) { return sizeof(kp1o_env); }
size_t kp1o_offsetof_env(int which) {
  switch (which) {
    case 0: return offsetof(kp1o_env, rng);
    case 1: return offsetof(kp1o_env, handoff);
    case 2: return offsetof(kp1o_env, min_pos_error);
    case 3: return offsetof(kp1o_env, goal_pose6);
    default: return offsetof(kp1o_env, last_reset_stage);
  }
}


/* Stand-alone evaluation of the two reward functions with caller-supplied arguments (the way the reference's own unit tests call
 * compute_approach_reward / compute_dock_reward: tests/test_kinematic_phase1_approach_reward.py, test_kinematic_phase1_split.py).
 * flags = {curr_in_pre_near_goal, prev_in_near_goal, curr_in_near_goal, dwell_count, near_goal_entry_count, near_goal_drift_count, success};
 * scalars = {joint_limit_margin_min, dq_norm, prev_dq_norm, delta_q_change_l2, entry_pos, entry_ori, entry_action_l2, entry_dq_norm}. */
double kp1o_reward_eval(const kp1_config* cfg, int mode, const double prev_pose6[6], const double curr_pose6[6], const double goal_pose6[6],
                        const double action[7], const double prev_action[7], const int32_t flags[7], const double scalars[8],
                        double* components, int32_t* n_components) {
  reward_in ri;
  ri.prev_pose6 = prev_pose6; ri.curr_pose6 = curr_pose6; ri.goal_pose6 = goal_pose6; ri.action = action; ri.prev_action = prev_action;
  ri.curr_in_pre_near_goal = flags[0]; ri.prev_in_near_goal = flags[1]; ri.curr_in_near_goal = flags[2];
  ri.dwell_count = flags[3]; ri.near_goal_entry_count = flags[4]; ri.near_goal_drift_count = flags[5]; ri.success = flags[6];
  ri.joint_limit_margin_min = scalars[0]; ri.dq_norm = scalars[1]; ri.prev_dq_norm = scalars[2]; ri.delta_q_change_l2 = scalars[3];
  ri.entry_pos = scalars[4]; ri.entry_ori = scalars[5]; ri.entry_action = scalars[6]; ri.entry_dq = scalars[7];
  if (mode == KP1_MODE_DOCK) {
    *n_components = N_DOCK_COMPONENTS;
    return compute_dock_reward(&cfg->dock_reward, &ri, components);
  }
  *n_components = N_APPROACH_COMPONENTS;
  return compute_approach_reward(&cfg->reward, &ri, components);
}
