/*
 * kp1_route_oracle.h -- CPU ORACLE of the route-curriculum environments (test infrastructure, NOT product code).
 * Restates kinematic_phase1/route/{route_dataset,route_reset_samplers,reward_route,route_observation,route_env,
 * route_sequence_env}.py in fp64; pinned by tests/golden/route_*.npz (written by tests/golden/make_golden_route.py from the
 * imported reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under oracle/.
 */
#ifndef KP1_ROUTE_ORACLE_H
#define KP1_ROUTE_ORACLE_H

#include "../include/kp1_route.h"
#include "kp1_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kp1o_route {
  int n;
  double* q;        /* [n][7] */
  double* pose;     /* [n][6] */
  double* next_dq;  /* [n][7] */
  double* progress; /* [n] */
  int* chunk;       /* [n] */
} kp1o_route;

kp1o_route* kp1o_route_load(const double* route_q, int n);
void kp1o_route_free(kp1o_route* r);

/* numpy Generator.normal (256-layer ziggurat) and Generator.choice(p=...) of one element */
double kp1o_rng_standard_normal(kp1o_rng* r);
int kp1o_rng_choice_p(kp1o_rng* r, const double* p, int n);

typedef struct kp1o_route_sample {
  double initial_q[7], initial_dq[7], initial_prev_action[7], goal_q[7];
  int route_index, start_index, mode;
} kp1o_route_sample;
void kp1o_route_sample_reset(kp1o_rng* rng, const kp1o_route* route, const kp1_joint_specs* js, const kp1_route_reset_cfg* cfg, kp1o_route_sample* out);

double kp1o_route_reward(const kp1_route_reward* cfg, const double prev_q[7], const double curr_q[7], const double goal_q[7], const double prev_pose6[6],
                         const double curr_pose6[6], const double goal_pose6[6], const double tangent[7], const double action[7],
                         const double prev_action[7], const double prev_dq[7], const double curr_dq[7], int ready_streak, double nearest,
                         double comps[KP1_ROUTE_N_COMPONENTS]);

typedef struct kp1o_route_env {
  kp1o_env base;
  kp1o_rng rng;
  const kp1o_route* route; /* borrowed */
  kp1_route_config cfg;
  int current_route_index, start_route_index, last_route_index, ready_streak, completed_waypoints, reset_mode;
  double prev_q[7], prev_dq[7];
} kp1o_route_env;

typedef struct kp1o_route_step_out {
  double reward;
  int terminated, truncated, success, route_ready, ready_streak, waypoint_success, route_regression, orientation_hit, route_index, completed_waypoints;
  double q_error_norm, nearest_route_q_distance;
  double components[KP1_ROUTE_N_COMPONENTS];
} kp1o_route_step_out;

void kp1o_route_env_init(kp1o_route_env* e, const kp1_config* base_cfg, const kp1_route_config* cfg, const kp1o_route* route);
void kp1o_route_env_seed(kp1o_route_env* e, uint64_t seed);
/* explicit: route_index >= 0 (start_index < 0 = max(route_index-1, 0)); initial_* may be NULL */
void kp1o_route_env_reset(kp1o_route_env* e, int route_index, int start_index, const double* initial_q, const double* initial_dq,
                          const double* initial_prev_action, float* obs);
void kp1o_route_env_step(kp1o_route_env* e, const double action[7], float* obs, kp1o_route_step_out* out);
int kp1o_route_obs_dim(const kp1o_route_env* e);
size_t kp1o_sizeof_route_env(void);
kp1o_rng* kp1o_route_env_rng(kp1o_route_env* e);
kp1o_env* kp1o_route_env_base(kp1o_route_env* e);
int kp1o_route_env_field(const kp1o_route_env* e, int which);
void kp1o_route_env_set_window(kp1o_route_env* e, int min_route_index, int max_route_index);

#ifdef __cplusplus
}
#endif
#endif
