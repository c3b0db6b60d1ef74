/*
 * kp1_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, fp64, one-env-at-a-time restatement of the reference's
 * hrl_trainer.kinematic_phase1 environment path, written function by function after
 * the Python (file:line cited at each function in kp1_oracle.c).  It is pinned against
 * the golden vectors in tests/golden/ that tests/golden/make_golden.py captured by
 * importing the reference in the build container (tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (rl_brain_trainer_amd/, libkp1.so) never links or calls it.
 * It shares only the kp1_config struct declaration with the product header.
 */
#ifndef KP1_ORACLE_H
#define KP1_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/kp1.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KP1O_MAX_COMPONENTS 64

typedef struct kp1o_rng {
  unsigned __int128 state;
  unsigned __int128 inc;
  int has_uint32;
  uint32_t uinteger;
} kp1o_rng;

typedef struct kp1o_env {
  kp1_config cfg;
  kp1o_rng rng;
  const kp1_handoff_state* handoff; /* borrowed */
  int n_handoff;
  /* ArmKinematicEnv fields; arm_kinematic_env.py:80-100 */
  int episode_step, dwell_count, near_goal_entry_count, near_goal_drift_count;
  int pre_near_goal_hit, near_goal_hit;
  double min_pos_error;
  double q[7], dq[7], prev_action[7];
  double entry_position_error_norm, entry_orientation_error_norm, entry_action_l2, entry_dq_norm;
  double goal_q[7], goal_pose6[6], ee_pose6[6];
  int curriculum_stage_index;
  int policy_mode;
  int last_reset_stage; /* stage index the last sampling reset drew from (diagnostic) */
} kp1o_env;

typedef struct kp1o_reset_opts {
  const double* initial_q;           /* [7] or NULL */
  const double* initial_dq;
  const double* initial_prev_action;
  const double* goal_q;
  const double* goal_pose6;          /* [6] or NULL */
  int policy_mode;                   /* -1 = config mode */
} kp1o_reset_opts;

typedef struct kp1o_step_out {
  double reward;
  int terminated, truncated, success, invalid;
  double position_error_norm, orientation_error_norm;
  double executed_delta_q_l2, action_l2, delta_q_change_l2;
  double dock_action_limit, dock_delta_q_change_limit_scale;
  double joint_limit_margin_min;
  int n_components;
  double components[KP1O_MAX_COMPONENTS];
} kp1o_step_out;

/* kinematics */
void kp1o_fk_matrix(const double q[7], double T[16]);
void kp1o_fk_pose6(const double q[7], double pose6[6]);
void kp1o_pose_error(const double curr[6], const double goal[6], double pos_err[3], double ori_err[3]);
double kp1o_wrap_to_pi(double v);
void kp1o_clip_q(const kp1_joint_specs* js, const double q[7], double out[7]);
void kp1o_joint_limit_margin(const kp1_joint_specs* js, const double q[7], double out[7]);
void kp1o_normalize_q(const kp1_joint_specs* js, const double q[7], double out[7]);
void kp1o_normalize_dq(const kp1_joint_specs* js, const double dq[7], double out[7]);

/* numpy Generator(PCG64) restatement */
void kp1o_rng_seed(kp1o_rng* r, uint64_t seed);
uint64_t kp1o_rng_next64(kp1o_rng* r);
uint32_t kp1o_rng_next32(kp1o_rng* r);
double kp1o_rng_double(kp1o_rng* r);
int64_t kp1o_rng_integers(kp1o_rng* r, int64_t low, int64_t high_exclusive);
void kp1o_rng_get(const kp1o_rng* r, kp1_rng_state* out);
void kp1o_rng_set(kp1o_rng* r, const kp1_rng_state* in);

/* env */
void kp1o_config_default(kp1_config* cfg);
void kp1o_env_init(kp1o_env* e, const kp1_config* cfg);
void kp1o_env_set_handoff(kp1o_env* e, const kp1_handoff_state* states, int n);
void kp1o_env_seed(kp1o_env* e, uint64_t seed);
void kp1o_env_set_stage(kp1o_env* e, int stage);
void kp1o_env_reset(kp1o_env* e, const kp1o_reset_opts* opts, float obs[KP1_OBS_DIM]);
void kp1o_env_step(kp1o_env* e, const double action[7], float obs[KP1_OBS_DIM], kp1o_step_out* out);
void kp1o_env_observe(const kp1o_env* e, float obs[KP1_OBS_DIM]);
void kp1o_env_capture_entry_metrics(kp1o_env* e);
const char* kp1o_component_name(int mode, int index);
/* compute_approach_reward / compute_dock_reward with caller-supplied arguments (reward_approach.py:75-373, reward_dock.py:123-484) */
double kp1o_reward_eval(const kp1_config* cfg, int mode, const double prev_pose6[6], const double curr_pose6[6], const double goal_pose6[6],
                        const double action[7], const double prev_action[7], const int32_t flags[7], const double scalars[8],
                        double* components, int32_t* n_components);
int kp1o_num_components(int mode);

/* curriculum tracker; curriculum.py:104-154 / callbacks.py:32-101 */
typedef struct kp1o_tracker {
  double threshold;
  int window, min_episodes, max_stage_index;
  int stage_index, stage_episode_count;
  int ring[1024];
  int ring_len, ring_head;
  double last_trigger_rate;
} kp1o_tracker;
void kp1o_tracker_init(kp1o_tracker* t, double threshold, int window, int min_episodes, int max_stage_index, int initial_stage);
int kp1o_tracker_record(kp1o_tracker* t, int success);

/* batched helpers for the CPU baseline (OpenMP over envs when built with -fopenmp) */
size_t kp1o_sizeof_env(void);
size_t kp1o_offsetof_env(int which);
int kp1o_max_threads(void);
void kp1o_batch_step(kp1o_env* envs, int n, const double* actions /*[n][7]*/, float* obs /*[n][56]*/,
                     double* reward, uint8_t* done, int auto_reset, int n_threads);
void kp1o_batch_step_components(kp1o_env* envs, int n, const double* actions, float* obs, double* reward, uint8_t* done, int auto_reset,
                                int n_threads, double* components /*[n][KP1O_MAX_COMPONENTS] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif
