/*
 * kp1_route_oracle.c -- CPU ORACLE of the route-curriculum environments (test infrastructure, NOT product code).
 * Reference = /root/reference/hrl_ws/src/hrl_trainer/hrl_trainer/kinematic_phase1/route/ ; each function cites what it restates.
 * The numpy pieces (ziggurat normal, choice with p) restate numpy's published algorithms; their tables come from
 * include/kp1_ziggurat_tables.h and the samplers are pinned draw for draw by tests/golden/route_rng_streams.npz.
 */
#include "kp1_route_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/kp1_ziggurat_tables.h"

#define NJ 7
static const uint64_t KI[256] = KP1_ZIGGURAT_KI;
static const double WI[256] = KP1_ZIGGURAT_WI;
static const double FI[256] = KP1_ZIGGURAT_FI;

static double norm7(const double* v) {
  double s = 0.0;
  for (int i = 0; i < NJ; ++i) s += v[i] * v[i];
  return sqrt(s);
}
static double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double maxd(double a, double b) { return a > b ? a : b; }

/* numpy/random/src/distributions/distributions.c random_standard_normal */
double kp1o_rng_standard_normal(kp1o_rng* g) {
  for (;;) {
    uint64_t r = kp1o_rng_next64(g);
    int idx = (int)(r & 0xff);
    r >>= 8;
    int sign = (int)(r & 0x1);
    uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * WI[idx];
    if (sign) x = -x;
    if (rabs < KI[idx]) return x;
    if (idx == 0) {
      for (;;) {
        double xx = -KP1_ZIGGURAT_NOR_INV_R * log1p(-kp1o_rng_double(g));
        double yy = -log1p(-kp1o_rng_double(g));
        if (yy + yy > xx * xx) return ((rabs >> 8) & 0x1) ? -(KP1_ZIGGURAT_NOR_R + xx) : KP1_ZIGGURAT_NOR_R + xx;
      }
    } else {
      if (((FI[idx - 1] - FI[idx]) * kp1o_rng_double(g) + FI[idx]) < exp(-0.5 * x * x)) return x;
    }
  }
}

/* Generator.choice(a, p=p) for one element: cdf = cumsum(p) / cdf[-1]; idx = searchsorted(cdf, random(), side="right") */
int kp1o_rng_choice_p(kp1o_rng* g, const double* p, int n) {
  double cdf[16];
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    acc += p[i];
    cdf[i] = acc;
  }
  for (int i = 0; i < n; ++i) cdf[i] /= acc;
  double u = kp1o_rng_double(g);
  int idx = 0;
  while (idx < n && cdf[idx] <= u) ++idx;
  return idx;
}

/* route_dataset.py:73-99 */
kp1o_route* kp1o_route_load(const double* route_q, int n) {
  kp1o_route* r = (kp1o_route*)calloc(1, sizeof *r);
  r->n = n;
  r->q = (double*)malloc(sizeof(double) * NJ * n);
  r->pose = (double*)malloc(sizeof(double) * 6 * n);
  r->next_dq = (double*)malloc(sizeof(double) * NJ * n);
  r->progress = (double*)malloc(sizeof(double) * n);
  r->chunk = (int*)malloc(sizeof(int) * n);
  memcpy(r->q, route_q, sizeof(double) * NJ * n);
  for (int i = 0; i < n; ++i) kp1o_fk_pose6(r->q + NJ * i, r->pose + 6 * i);
  double acc = 0.0;
  r->progress[0] = 0.0;
  for (int i = 1; i < n; ++i) {
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = r->pose[6 * i + k] - r->pose[6 * (i - 1) + k];
    acc += norm3(d);  /* np.cumsum: sequential */
    r->progress[i] = acc;
  }
  /* default_chunk_bounds(max_index) :59-68 */
  const int mx = n - 1;
  const int lo[7] = {1, 41, 81, 121, 181, 261, 361};
  const int hi[7] = {40, 80, 120, 180, 260, 360, mx};
  for (int i = 0; i < n; ++i) {
    int nx = i + 1 < n ? i + 1 : n - 1;
    for (int k = 0; k < NJ; ++k) r->next_dq[NJ * i + k] = r->q[NJ * nx + k] - r->q[NJ * i + k];
    int c = 6;
    for (int b = 0; b < 7; ++b) {
      int h = b < 6 ? (hi[b] < mx ? hi[b] : mx) : mx;
      if (lo[b] <= i && i <= h) {
        c = b;
        break;
      }
    }
    r->chunk[i] = c;
  }
  return r;
}
void kp1o_route_free(kp1o_route* r) {
  if (!r) return;
  free(r->q); free(r->pose); free(r->next_dq); free(r->progress); free(r->chunk); free(r);
}
static const double* wp_q(const kp1o_route* r, int i) { return r->q + NJ * clipi(i, 0, r->n - 1); }
static const double* wp_pose(const kp1o_route* r, int i) { return r->pose + 6 * clipi(i, 0, r->n - 1); }
static const double* wp_next(const kp1o_route* r, int i) { return r->next_dq + NJ * clipi(i, 0, r->n - 1); }

static void normal_noise7(kp1o_rng* g, double std, double out[NJ]) {  /* _normal_noise :43-44 */
  for (int i = 0; i < NJ; ++i) out[i] = std > 0.0 ? 0.0 + std * kp1o_rng_standard_normal(g) : 0.0;
}

/* route_reset_samplers.py:47-117 */
void kp1o_route_sample_reset(kp1o_rng* rng, const kp1o_route* route, const kp1_joint_specs* js, const kp1_route_reset_cfg* c, kp1o_route_sample* out) {
  const int max_index = route->n - 1;
  const int lo = clipi(c->min_route_index, 1, max_index);
  const int hi = clipi(c->max_route_index, lo, max_index);
  double ratios[5] = {maxd(c->prefix_start_reset_ratio, 0.0), maxd(c->random_prefix_reset_ratio, 0.0), maxd(c->segment_reset_ratio, 0.0),
                      maxd(c->replay_reset_ratio, 0.0), maxd(c->recovery_reset_ratio, 0.0)};
  double total = 0.0;
  for (int i = 0; i < 5; ++i) total += ratios[i];
  if (total > 0.0) {
    for (int i = 0; i < 5; ++i) ratios[i] /= total;
  } else {
    const double fallback[5] = {0.0, 1.0, 0.0, 0.0, 0.0};
    memcpy(ratios, fallback, sizeof ratios);
  }
  int mode = kp1o_rng_choice_p(rng, ratios, 5);
  if (c->mode >= 1 && c->mode <= 5) mode = c->mode - 1;
  int route_index, start_index;
  if (mode == KP1_ROUTE_MODE_PREFIX_START) {
    route_index = (int)kp1o_rng_integers(rng, lo, hi + 1);
    start_index = 0;
  } else if (mode == KP1_ROUTE_MODE_SEGMENT) {
    int seg_lo = clipi(c->segment_start_index, 1, max_index);
    int seg_hi = clipi(c->segment_end_index, seg_lo, max_index);
    route_index = (int)kp1o_rng_integers(rng, seg_lo, (seg_hi < hi ? seg_hi : hi) + 1);
    start_index = route_index - 1 > 0 ? route_index - 1 : 0;
  } else if (mode == KP1_ROUTE_MODE_REPLAY) {
    int rlo = clipi(c->replay_start_index, 1, max_index);
    int rhi = clipi(c->replay_end_index, rlo, max_index);
    route_index = (int)kp1o_rng_integers(rng, rlo, (rhi < hi ? rhi : hi) + 1);
    start_index = route_index - 1 > 0 ? route_index - 1 : 0;
  } else {
    route_index = (int)kp1o_rng_integers(rng, lo, hi + 1);
    start_index = route_index - 1 > 0 ? route_index - 1 : 0;
  }
  memcpy(out->goal_q, wp_q(route, route_index), sizeof out->goal_q);
  double q0[NJ], noise[NJ];
  memcpy(q0, wp_q(route, mode == KP1_ROUTE_MODE_RECOVERY ? route_index : start_index), sizeof q0);
  normal_noise7(rng, c->q_noise_std, noise);
  for (int i = 0; i < NJ; ++i) q0[i] = q0[i] + noise[i];
  kp1o_clip_q(js, q0, out->initial_q);
  normal_noise7(rng, c->dq_noise_std, out->initial_dq);
  normal_noise7(rng, c->prev_action_noise_std, noise);
  for (int i = 0; i < NJ; ++i) out->initial_prev_action[i] = noise[i] < -1.0 ? -1.0 : (noise[i] > 1.0 ? 1.0 : noise[i]);
  out->route_index = route_index;
  out->start_index = start_index;
  out->mode = mode;
}

/* reward_route.py:54-143; comps in the dict's insertion order */
double kp1o_route_reward(const kp1_route_reward* cfg, const double prev_q[7], const double curr_q[7], const double goal_q[7], const double prev_pose6[6],
                         const double curr_pose6[6], const double goal_pose6[6], const double tangent[7], const double action[7],
                         const double prev_action[7], const double prev_dq[7], const double curr_dq[7], int ready_streak, double nearest,
                         double comps[KP1_ROUTE_N_COMPONENTS]) {
  (void)prev_dq;
  double d[NJ];
  for (int i = 0; i < NJ; ++i) d[i] = goal_q[i] - prev_q[i];
  const double prev_q_err = norm7(d);
  for (int i = 0; i < NJ; ++i) d[i] = goal_q[i] - curr_q[i];
  const double curr_q_err = norm7(d);
  double pe[3], oe[3];
  kp1o_pose_error(prev_pose6, goal_pose6, pe, oe);
  const double prev_pos = norm3(pe), prev_ori = norm3(oe);
  kp1o_pose_error(curr_pose6, goal_pose6, pe, oe);
  const double curr_pos = norm3(pe), curr_ori = norm3(oe);
  const double action_norm = norm7(action), dq_norm = norm7(curr_dq), tangent_norm = norm7(tangent);
  double dot = 0.0;
  for (int i = 0; i < NJ; ++i) dot += (curr_q[i] - prev_q[i]) * tangent[i];
  const double tangent_progress = tangent_norm > 0.0 ? dot / maxd(tangent_norm, 1e-9) : 0.0;
  const int ready_now = curr_q_err <= cfg->route_ready_q_threshold && curr_pos <= cfg->route_ready_pos_threshold_m &&
                        curr_ori <= cfg->route_ready_ori_threshold_rad && action_norm <= cfg->route_ready_action_threshold &&
                        dq_norm <= cfg->route_ready_dq_threshold;
  double low_motion = 0.0;
  if (curr_pos <= 2.0 * cfg->route_ready_pos_threshold_m && curr_ori <= 2.0 * cfg->route_ready_ori_threshold_rad) {
    double action_clean = maxd(1.0 - action_norm / maxd(cfg->route_ready_action_threshold, 1e-9), 0.0);
    double dq_clean = maxd(1.0 - dq_norm / maxd(cfg->route_ready_dq_threshold, 1e-9), 0.0);
    low_motion = cfg->low_motion_near_waypoint_bonus * 0.5 * (action_clean + dq_clean);
  }
  double a2 = 0.0, da2 = 0.0;
  for (int i = 0; i < NJ; ++i) {
    a2 += action[i] * action[i];
    da2 += (action[i] - prev_action[i]) * (action[i] - prev_action[i]);
  }
  double smooth = -cfg->action_magnitude_weight * (a2 / 7.0);
  smooth += -cfg->action_delta_weight * (da2 / 7.0);
  comps[0] = cfg->q_goal_progress_weight * (prev_q_err - curr_q_err);
  comps[1] = cfg->ee_position_progress_weight * (prev_pos - curr_pos);
  comps[2] = cfg->ee_orientation_progress_weight * (prev_ori - curr_ori);
  comps[3] = cfg->route_tangent_progress_weight * maxd(tangent_progress, 0.0);
  comps[4] = ready_now ? cfg->same_step_route_ready_bonus : 0.0;
  comps[5] = (ready_now && ready_streak >= 1) ? cfg->route_ready_dwell_bonus : 0.0;
  comps[6] = low_motion;
  comps[7] = -cfg->orientation_regression_penalty_weight * maxd(curr_ori - prev_ori, 0.0);
  comps[8] = -cfg->q_route_regression_penalty_weight * maxd(curr_q_err - prev_q_err, 0.0);
  comps[9] = -cfg->off_route_penalty_weight * maxd(nearest, 0.0);
  comps[10] = smooth;
  comps[11] = -cfg->dq_penalty_weight * dq_norm;
  comps[12] = (curr_q_err >= prev_q_err && curr_pos >= prev_pos && curr_ori >= prev_ori) ? -cfg->no_progress_penalty : 0.0;
  comps[13] = curr_q_err;
  comps[14] = curr_pos;
  comps[15] = curr_ori;
  comps[16] = (double)ready_now;
  double reward = 0.0;  /* python sum(): left to right from 0 */
  for (int i = 0; i < 13; ++i) reward += comps[i];
  return reward;
}

/* ------------------------------------------------------------------ env wrappers */
void kp1o_route_env_init(kp1o_route_env* e, const kp1_config* base_cfg, const kp1_route_config* cfg, const kp1o_route* route) {
  memset(e, 0, sizeof *e);
  kp1o_env_init(&e->base, base_cfg);
  e->cfg = *cfg;
  e->route = route;
  kp1o_rng_seed(&e->rng, 0);  /* default_rng(None) in the reference is entropy-seeded; tests always seed */
  e->current_route_index = 1;
}
void kp1o_route_env_seed(kp1o_route_env* e, uint64_t seed) {
  kp1o_rng_seed(&e->rng, seed);
  kp1o_env_seed(&e->base, seed);
}
int kp1o_route_obs_dim(const kp1o_route_env* e) { return e->cfg.include_route_keys ? KP1_ROUTE_OBS_DIM : KP1_OBS_DIM; }
size_t kp1o_sizeof_route_env(void) { return sizeof(kp1o_route_env); }

/* route_observation.py:31-61 on the flattened, key-sorted layout */
static void augment(const kp1o_route_env* e, const float base[KP1_OBS_DIM], float* obs) {
  if (!e->cfg.include_route_keys) {
    memcpy(obs, base, sizeof(float) * KP1_OBS_DIM);
    return;
  }
  const kp1_joint_specs* js = &e->base.cfg.joints;
  const kp1o_route* r = e->route;
  const double* goal = wp_q(r, e->current_route_index);
  const double* tangent = wp_next(r, e->current_route_index - 1 > 0 ? e->current_route_index - 1 : 0);
  double gq[NJ], err[NJ], en[NJ], tn[NJ];
  kp1o_normalize_q(js, goal, gq);
  for (int i = 0; i < NJ; ++i) err[i] = goal[i] - e->base.q[i];
  kp1o_normalize_dq(js, err, en);
  kp1o_normalize_dq(js, tangent, tn);
  memcpy(obs, base, sizeof(float) * 47);                 /* dq .. q */
  for (int i = 0; i < NJ; ++i) obs[47 + i] = (float)en[i];   /* route_q_error */
  for (int i = 0; i < NJ; ++i) obs[54 + i] = (float)gq[i];   /* route_q_goal */
  const int max_route_index = r->n - 1;
  double s0 = (double)e->current_route_index / (double)(max_route_index > 1 ? max_route_index : 1);
  double s1 = r->progress[clipi(e->current_route_index, 0, r->n - 1)] / maxd(r->progress[r->n - 1], 1e-9);
  obs[61] = (float)(s0 < 0.0 ? 0.0 : (s0 > 1.0 ? 1.0 : s0));
  obs[62] = (float)(s1 < 0.0 ? 0.0 : (s1 > 1.0 ? 1.0 : s1));
  obs[63] = 0.0f;
  for (int i = 0; i < NJ; ++i) obs[64 + i] = (float)tn[i];   /* route_tangent */
  memcpy(obs + 71, base + 47, sizeof(float) * 9);          /* task_type, wp_ori_err, wp_pos_err */
}

/* route_env.py:48-99 / route_sequence_env.py:98-148 */
void kp1o_route_env_reset(kp1o_route_env* e, int route_index, int start_index, const double* initial_q, const double* initial_dq,
                          const double* initial_prev_action, float* obs) {
  const kp1o_route* r = e->route;
  double q0[NJ], dq0[NJ], pa0[NJ], goal[NJ];
  int first_target, start;
  if (route_index >= 0) {
    first_target = route_index;
    start = start_index >= 0 ? start_index : (route_index - 1 > 0 ? route_index - 1 : 0);
    const int seq = e->cfg.sequence_enabled;
    memcpy(q0, (seq && initial_q) ? initial_q : wp_q(r, start), sizeof q0);
    if (seq && initial_dq) memcpy(dq0, initial_dq, sizeof dq0); else memset(dq0, 0, sizeof dq0);
    if (seq && initial_prev_action) memcpy(pa0, initial_prev_action, sizeof pa0); else memset(pa0, 0, sizeof pa0);
    memcpy(goal, wp_q(r, first_target), sizeof goal);
    e->reset_mode = KP1_ROUTE_MODE_EXPLICIT;
  } else {
    kp1o_route_sample s;
    kp1o_route_sample_reset(&e->rng, r, &e->base.cfg.joints, &e->cfg.reset, &s);
    first_target = s.route_index;
    start = s.start_index;
    memcpy(q0, s.initial_q, sizeof q0);
    memcpy(dq0, s.initial_dq, sizeof dq0);
    memcpy(pa0, s.initial_prev_action, sizeof pa0);
    memcpy(goal, s.goal_q, sizeof goal);
    e->reset_mode = s.mode;
  }
  if (e->cfg.sequence_enabled) {
    int max_index = e->cfg.reset.max_route_index < r->n - 1 ? e->cfg.reset.max_route_index : r->n - 1;
    int seq_len = e->cfg.sequence_length > 1 ? e->cfg.sequence_length : 1;
    e->current_route_index = clipi(first_target, 1, max_index);
    e->last_route_index = max_index < e->current_route_index + seq_len - 1 ? max_index : e->current_route_index + seq_len - 1;
    memcpy(goal, wp_q(r, e->current_route_index), sizeof goal);
  } else {
    e->current_route_index = first_target;
    e->last_route_index = first_target;
  }
  e->start_route_index = start;
  e->ready_streak = 0;
  e->completed_waypoints = 0;
  kp1o_reset_opts o = {q0, dq0, pa0, goal, 0, KP1_MODE_APPROACH};
  float base_obs[KP1_OBS_DIM];
  kp1o_env_reset(&e->base, &o, base_obs);
  memcpy(e->prev_q, e->base.q, sizeof e->prev_q);
  memcpy(e->prev_dq, e->base.dq, sizeof e->prev_dq);
  if (obs) augment(e, base_obs, obs);
}

/* route_env.py:124-192 / route_sequence_env.py:150-236 */
void kp1o_route_env_step(kp1o_route_env* e, const double action[7], float* obs, kp1o_route_step_out* out) {
  const kp1o_route* r = e->route;
  const kp1_route_reward* rc = &e->cfg.reward;
  double prev_q[NJ], prev_dq[NJ], prev_action[NJ], prev_pose6[6];
  memcpy(prev_q, e->prev_q, sizeof prev_q);
  memcpy(prev_dq, e->prev_dq, sizeof prev_dq);
  kp1o_fk_pose6(prev_q, prev_pose6);
  memcpy(prev_action, e->base.prev_action, sizeof prev_action);
  const int target_index = e->current_route_index;
  const double* goal_q = wp_q(r, target_index);
  const double* goal_pose6 = wp_pose(r, target_index);
  const double* tangent = wp_next(r, target_index - 1 > 0 ? target_index - 1 : 0);

  float base_obs[KP1_OBS_DIM];
  kp1o_step_out so;
  kp1o_env_step(&e->base, action, base_obs, &so);
  double curr_q[NJ], curr_dq[NJ], curr_pose6[6], d[NJ];
  memcpy(curr_q, e->base.q, sizeof curr_q);
  memcpy(curr_dq, e->base.dq, sizeof curr_dq);
  kp1o_fk_pose6(curr_q, curr_pose6);
  for (int i = 0; i < NJ; ++i) d[i] = goal_q[i] - curr_q[i];
  const double q_error_norm = norm7(d);
  for (int i = 0; i < NJ; ++i) d[i] = goal_q[i] - prev_q[i];
  const double prev_q_error_norm = norm7(d);
  const double action_norm = norm7(action), dq_norm = norm7(curr_dq);
  double nearest = INFINITY;
  for (int w = 0; w < r->n; ++w) {
    for (int i = 0; i < NJ; ++i) d[i] = r->q[NJ * w + i] - curr_q[i];
    double dist = norm7(d);
    if (dist < nearest) nearest = dist;
  }
  const int ready_now = q_error_norm <= rc->route_ready_q_threshold && so.position_error_norm <= rc->route_ready_pos_threshold_m &&
                        so.orientation_error_norm <= rc->route_ready_ori_threshold_rad && action_norm <= rc->route_ready_action_threshold &&
                        dq_norm <= rc->route_ready_dq_threshold;
  e->ready_streak = ready_now ? e->ready_streak + 1 : 0;
  out->reward = kp1o_route_reward(rc, prev_q, curr_q, goal_q, prev_pose6, curr_pose6, goal_pose6, tangent, action, prev_action, prev_dq, curr_dq,
                                  e->ready_streak, nearest, out->components);
  const int reached = ready_now && e->ready_streak >= e->base.cfg.termination.success_dwell_steps;
  int terminated = 0, success = 0, retarget = 0;
  out->ready_streak = e->ready_streak;  /* the streak reported is the value before a reset-on-advance?  no: after (info is built later) */
  if (e->cfg.sequence_enabled) {
    if (reached) {
      e->completed_waypoints += 1;
      if (target_index >= e->last_route_index) {
        success = 1;
        terminated = 1;
      } else {
        e->current_route_index = target_index + 1;  /* _advance_target :253-257 */
        memcpy(e->base.goal_q, wp_q(r, e->current_route_index), sizeof e->base.goal_q);
        memcpy(e->base.goal_pose6, wp_pose(r, e->current_route_index), sizeof e->base.goal_pose6);
        kp1o_env_capture_entry_metrics(&e->base);
        if (e->cfg.reset_ready_streak_on_advance) e->ready_streak = 0;
        retarget = 1;
      }
    }
    if (so.terminated && !terminated && !so.success) terminated = 1;  /* base reason != "success" (invalid_state) */
    out->ready_streak = e->ready_streak;
  } else {
    success = reached;
    terminated = so.terminated;
    if (so.terminated && so.success && !so.invalid && !success) terminated = 0;
    if (success && e->base.cfg.termination.terminate_on_success) terminated = 1;
  }
  if (retarget) kp1o_env_observe(&e->base, base_obs);
  out->terminated = terminated;
  out->truncated = so.truncated;
  out->success = success;
  out->route_ready = ready_now;
  out->waypoint_success = reached;
  out->route_regression = q_error_norm > prev_q_error_norm;
  out->orientation_hit = so.orientation_error_norm <= rc->route_ready_ori_threshold_rad;
  out->route_index = e->current_route_index;
  out->completed_waypoints = e->completed_waypoints;
  out->q_error_norm = q_error_norm;
  out->nearest_route_q_distance = nearest;
  memcpy(e->prev_q, curr_q, sizeof e->prev_q);
  memcpy(e->prev_dq, curr_dq, sizeof e->prev_dq);
  if (obs) augment(e, base_obs, obs);
}

/* accessors for the ctypes wrapper (oracle/route_oracle.py treats kp1o_route_env as opaque) */
kp1o_rng* kp1o_route_env_rng(kp1o_route_env* e) { return &e->rng; }
kp1o_env* kp1o_route_env_base(kp1o_route_env* e) { return &e->base; }
int kp1o_route_env_field(const kp1o_route_env* e, int which) {
  switch (which) {
    case 0: return e->current_route_index;
    case 1: return e->start_route_index;
    case 2: return e->last_route_index;
    case 3: return e->ready_streak;
    case 4: return e->completed_waypoints;
    default: return e->reset_mode;
  }
}
void kp1o_route_env_set_window(kp1o_route_env* e, int min_route_index, int max_route_index) {
  e->cfg.reset.min_route_index = min_route_index;
  e->cfg.reset.max_route_index = max_route_index;
}
