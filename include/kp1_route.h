/*
 * kp1_route.h -- C ABI of the route-curriculum environments of the MI355X kinematic_phase1 engine.
 *
 * Replaces, for N environments at once, the reference's
 *   kinematic_phase1/route/route_dataset.py:73-99          load_route_dataset   (FK per waypoint, path length, tangents, chunks)
 *   kinematic_phase1/route/route_reset_samplers.py:47-117  sample_route_reset
 *   kinematic_phase1/route/reward_route.py:36-143          route_ready, compute_route_reward
 *   kinematic_phase1/route/route_observation.py:31-61      augment_route_observation
 *   kinematic_phase1/route/route_env.py:29-212             RouteKinematicEnv          (sequence_enabled = 0)
 *   kinematic_phase1/route/route_sequence_env.py:29-278    RouteSequenceKinematicEnv  (sequence_enabled = 1)
 * Both wrappers drive an approach-mode base env (kp1.h) whose reward is discarded; a kp1_route handle borrows a kp1_env the
 * caller created with the route config's env block and owns the per-env route state and a second PCG64 stream per env
 * (the wrapper's own `self._rng`; env i is seeded default_rng(seed + first_env_id + i) like make_vec_env does).
 *
 * All pointers are HIP device pointers unless named *_host.  Returns KP1_OK or a negative kp1_status (kp1.h).
 */
#ifndef KP1_ROUTE_H
#define KP1_ROUTE_H

#include <stdint.h>

#include "kp1.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reward_route.py:13-33 RouteRewardConfig, declaration order */
#define KP1_ROUTE_REWARD_FIELDS(X)                                                                                          \
  X(q_goal_progress_weight, 2.0) X(ee_position_progress_weight, 6.0) X(ee_orientation_progress_weight, 5.0)                 \
  X(route_tangent_progress_weight, 0.25) X(same_step_route_ready_bonus, 1.5) X(route_ready_dwell_bonus, 0.8)                \
  X(low_motion_near_waypoint_bonus, 0.4) X(orientation_regression_penalty_weight, 4.0) X(q_route_regression_penalty_weight, 1.0) \
  X(off_route_penalty_weight, 0.25) X(action_magnitude_weight, 0.02) X(action_delta_weight, 0.03) X(dq_penalty_weight, 0.8)   \
  X(no_progress_penalty, 0.02) X(route_ready_pos_threshold_m, 0.010) X(route_ready_ori_threshold_rad, 0.150)                 \
  X(route_ready_q_threshold, 0.080) X(route_ready_action_threshold, 0.25) X(route_ready_dq_threshold, 0.010)
typedef struct kp1_route_reward {
#define X(name, dflt) double name;
  KP1_ROUTE_REWARD_FIELDS(X)
#undef X
} kp1_route_reward;

/* route_reset_samplers.py:14-30 RouteResetSamplerConfig.  mode: 0 "mixed_prefix_segment" (or any other string: the drawn
 * mode stands), 1 prefix_start_reset, 2 random_prefix_reset, 3 segment_reset, 4 replay_reset, 5 recovery_reset */
typedef struct kp1_route_reset_cfg {
  int32_t mode, min_route_index, max_route_index, segment_start_index, segment_end_index, replay_start_index, replay_end_index, pad_;
  double prefix_start_reset_ratio, random_prefix_reset_ratio, segment_reset_ratio, replay_reset_ratio, recovery_reset_ratio;
  double q_noise_std, dq_noise_std, prev_action_noise_std;
} kp1_route_reset_cfg;

typedef struct kp1_route_config {
  kp1_route_reward reward;
  kp1_route_reset_cfg reset;
  int32_t include_route_keys;             /* route_observation.py RouteObservationConfig */
  int32_t sequence_enabled, sequence_length, reset_ready_streak_on_advance; /* route_sequence_env.py:20-26 */
} kp1_route_config;

/* drawn / explicit reset modes as reported in info["route_reset_mode"] */
enum { KP1_ROUTE_MODE_PREFIX_START = 0, KP1_ROUTE_MODE_RANDOM_PREFIX, KP1_ROUTE_MODE_SEGMENT, KP1_ROUTE_MODE_REPLAY, KP1_ROUTE_MODE_RECOVERY,
       KP1_ROUTE_MODE_EXPLICIT };

/* Observation with the route keys: the 17 Dict keys in SB3's sorted order -- the 56-float layout up to and including `q`, then
 * route_q_error 7, route_q_goal 7, route_scalar 3, route_tangent 7, then task_type 3, wp_ori_err 3, wp_pos_err 3. */
#define KP1_ROUTE_OBS_DIM 80
#define KP1_ROUTE_N_COMPONENTS 17  /* compute_route_reward's components dict, insertion order (reward_route.py:122-140) */

typedef struct kp1_route kp1_route;

/* route_q_host: [n_waypoints][7] f64 joint goals (the JSON's "route_q").  FK per waypoint runs on the device in fp64. */
int kp1_route_create(kp1_env* base_env, const kp1_route_config* cfg, const double* route_q_host, int32_t n_waypoints, uint64_t seed,
                     uint64_t first_env_id, kp1_route** out);
int kp1_route_destroy(kp1_route* r);
/* host copies of the dataset as load_route_dataset computes it: poses6 [W][6], route_progress_m [W], next_q_delta [W][7], chunk_id [W] */
int kp1_route_get_dataset(kp1_route* r, double* poses6_host, double* progress_host, double* next_q_delta_host, int32_t* chunk_id_host);
/* set_route_window (route_env.py:101-122): replaces reset.min/max_route_index */
int kp1_route_set_window(kp1_route* r, int32_t min_route_index, int32_t max_route_index);
int kp1_route_seed(kp1_route* r, uint64_t seed, uint64_t first_env_id);

/* explicit reset (options{"route_index", "start_route_index", "initial_q", ...}); NULL members fall back as the wrappers do */
typedef struct kp1_route_reset_opts {
  const int32_t* route_index;       /* [N] or NULL = sample with the route stream */
  const int32_t* start_route_index; /* [N] or NULL = max(route_index - 1, 0) */
  const double* initial_q;          /* [N][7] (sequence env only) or NULL = waypoint(start).q_goal */
  const double* initial_dq;         /* [N][7] or NULL = 0 */
  const double* initial_prev_action;
  int32_t evaluator_state;          /* != 0: honour initial_q / initial_dq / initial_prev_action in the single-waypoint env too -- what
                                       eval_route_curriculum.py:62-83 does by resetting base_env and poking _route_index / _prev_info */
  int32_t pad_;
} kp1_route_reset_opts;
int kp1_route_reset(kp1_route* r, const uint8_t* mask, const kp1_route_reset_opts* opts, float* obs);
/* actions [N][7] (f32 or f64 like the base env); obs [N][obs_dim()]; reward [N]; done [N] KP1_DONE_* bits (SUCCESS = route /
 * sequence success); auto_reset != 0 resets finished envs in place after writing terminal_obs (may be NULL) */
int kp1_route_step(kp1_route* r, const void* actions, float* obs, void* reward, uint8_t* done, float* terminal_obs, int32_t auto_reset);
int kp1_route_obs_dim(const kp1_route* r);
/* row pitch of the observation buffers passed to reset / step (default obs_dim; the PPO loop uses the MFMA kernels' padded
 * widths 64 / 128 -- only the first obs_dim floats of a row are written) */
int kp1_route_set_obs_stride(kp1_route* r, int32_t stride);

typedef struct kp1_route_info_view {  /* device arrays [N], valid until the next call */
  const int32_t* route_index; const int32_t* start_route_index; const int32_t* last_route_index; const int32_t* reset_mode;
  const int32_t* ready_streak; const int32_t* completed_waypoints;
  const uint8_t* route_ready; const uint8_t* waypoint_success; const uint8_t* route_regression; const uint8_t* orientation_hit;
  const void* q_error_norm; const void* nearest_route_q_distance;  /* real type of the base env */
} kp1_route_info_view;
int kp1_route_get_info(kp1_route* r, kp1_route_info_view* out);
int kp1_route_enable_reward_components(kp1_route* r, int32_t enable);
int kp1_route_get_reward_components(kp1_route* r, const void** comps /* [17][N] */);
const char* kp1_route_component_name(int32_t index);
int kp1_route_rng_get(kp1_route* r, kp1_rng_state* out_host);
int kp1_route_rng_set(kp1_route* r, const kp1_rng_state* in_host);
void kp1_route_config_default(kp1_route_config* cfg);

/* ---- prefix curriculum on the device (route/route_curriculum.py:23-132, RoutePrefixCurriculumCallback) ----------------
 * The callback scans the finished episodes of every VecEnv step in env order, appends (success, route_ready, orientation hit, regression)
 * to four windows and promotes to the next prefix when the four window rates pass; promotion calls set_route_window(max = prefix, min = 1)
 * on all envs.  kp1_route_curriculum_observe does that scan on the device after each kp1_route_step (same stream): it reads the done
 * bytes the caller passes and the wrapper's own per-env flags, and on promotion rewrites the reset window in the device-resident route
 * config, so the rollout needs no host synchronisation and can be replayed from a hipGraph. */
#define KP1_ROUTE_CURRICULUM_MAX_STAGES 16
#define KP1_ROUTE_CURRICULUM_MAX_WINDOW 1024
#define KP1_ROUTE_CURRICULUM_MAX_HISTORY 32
typedef struct kp1_route_curriculum_event {
  int64_t total_timesteps;
  int32_t from_stage, to_stage, from_prefix_end_index, to_prefix_end_index;
  double recent_success_rate, recent_route_ready_hit_rate, recent_orientation_hit_rate, recent_regression_rate;
} kp1_route_curriculum_event;
typedef struct kp1_route_curriculum_state {
  int32_t stage_index, stage_episode_count, ring_len, ring_head;
  int32_t window_episodes, min_episodes_per_stage, n_stages, n_events;
  int32_t prefix_end_index[KP1_ROUTE_CURRICULUM_MAX_STAGES];
  int32_t ring_sums[4]; /* running sums of the four windows (the callback recomputes the means per finished episode) */
  double promotion_success_rate, promotion_route_ready_hit_rate, promotion_orientation_hit_rate, promotion_max_regression_rate;
  int64_t num_timesteps;
  uint8_t ring[4][KP1_ROUTE_CURRICULUM_MAX_WINDOW]; /* successes, ready_hits, orientation_hits, regressions */
  kp1_route_curriculum_event events[KP1_ROUTE_CURRICULUM_MAX_HISTORY];
} kp1_route_curriculum_state;

/* allocate the tracker in device memory and apply the first stage's window (_on_training_start) */
int kp1_route_curriculum_create(kp1_route* r, const int32_t* prefix_end_index, int32_t n_stages, double promotion_success_rate,
                                double promotion_route_ready_hit_rate, double promotion_orientation_hit_rate, double promotion_max_regression_rate,
                                int32_t window_episodes, int32_t min_episodes_per_stage, kp1_route_curriculum_state** out_dev);
int kp1_route_curriculum_destroy(kp1_route* r, kp1_route_curriculum_state* st_dev);
/* _on_step: dones[0..N) = the KP1_DONE_* bytes kp1_route_step just wrote; steps_per_call = env steps this call stands for */
int kp1_route_curriculum_observe(kp1_route* r, kp1_route_curriculum_state* st_dev, const uint8_t* dones, int32_t steps_per_call, void* stream);
/* copy the tracker to the host (synchronises the stream) and bring the host copy of the reset window up to date */
int kp1_route_curriculum_read(kp1_route* r, const kp1_route_curriculum_state* st_dev, kp1_route_curriculum_state* out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KP1_ROUTE_H */
