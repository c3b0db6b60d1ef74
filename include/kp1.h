/*
 * kp1.h -- C ABI of the MI355X-native kinematic_phase1 rollout engine.
 *
 * This is the drop-in boundary for the Approach -> Finisher hot path of
 * jerry102102102/RL_brain_trainer (hrl_trainer.kinematic_phase1).  The reference is
 * pure Python and has no FFI; the seam it offers is the Gymnasium-style env API
 * (reference: kinematic_phase1/envs/arm_kinematic_env.py:69-560) that trainers,
 * callbacks and evaluators call.  Every entry point below is the batched
 * (N environments per call) counterpart of one method of that class and cites it.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no exceptions: every function returns
 *     KP1_OK (0) or a negative kp1_status; kp1_last_error() gives the text.
 *   - "dev" pointers are HIP device pointers (hipMalloc / torch.cuda tensors);
 *     "host" pointers are ordinary host memory.  Sizes are in elements.
 *   - one handle owns one HIP stream-ordered set of device buffers; calls on one
 *     handle are stream ordered on the stream given at creation (0 = null stream).
 *   - the library refuses to run without a HIP device: there is no CPU fallback.
 *
 * The struct layouts are generated from the X-macro field lists below; the Python
 * host side (rl_brain_trainer_amd/config.py) parses the same lists, so this header
 * is the single source of truth for field order and defaults.  Defaults are the
 * reference dataclass defaults (file:line given per block).
 */
#ifndef KP1_H
#define KP1_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KP1_NJ 7              /* joints: rack (prismatic) + 6 revolute; joint_limits.py:13-21 */
#define KP1_OBS_DIM 56        /* flattened Dict observation; spaces.py:73-90 */
#define KP1_MAX_STAGES 16     /* curriculum stages (reference configs use 6, 10 or 12) */
#define KP1_MAX_MILESTONES 8  /* reward_approach.py:22-23 tuple fields */

typedef enum kp1_status {
  KP1_OK = 0,
  KP1_ERR_INVALID = -1,   /* bad argument / shape (the reference raises ValueError) */
  KP1_ERR_NO_DEVICE = -2, /* no HIP device (hipErrorNoDevice / hipErrorInvalidDevice): this library has no CPU path */
  KP1_ERR_ALLOC = -3,     /* device or host memory exhausted (hipErrorOutOfMemory) */
  KP1_ERR_UNSUPPORTED = -4, /* mode outside {approach, dock} */
  KP1_ERR_LAUNCH = -5,    /* a kernel launch was rejected (bad configuration, LDS / register budget) */
  KP1_ERR_RUNTIME = -6    /* any other HIP runtime failure, incl. an asynchronous fault of an earlier kernel surfacing at this call;
                             kp1_last_error() names the HIP call and hipGetErrorString */
} kp1_status;

/* policy modes; arm_kinematic_env.py:544-551 (_mode_index). bridge/dock_coarse are out of scope. */
enum { KP1_MODE_APPROACH = 0, KP1_MODE_DOCK = 1 };

/* arithmetic the kernels run in; state/IO buffers are this type except obs (always f32). */
enum { KP1_REAL_F32 = 0, KP1_REAL_F64 = 1 };

/* Observation layout (KP1_OBS_DIM floats per env, row-major [N][56]).  The order is the
 * one SB3's CombinedExtractor feeds the MLP: gymnasium.spaces.Dict sorts plain-dict keys,
 * so checkpoints' first layer assumes alphabetical key order. */
enum {
  KP1_OBS_DQ = 0,                /* 7  observation_builder.py:79 */
  KP1_OBS_GOAL_ORI_ERR = 7,      /* 3  :82 */
  KP1_OBS_GOAL_POS_ERR = 10,     /* 3  :81 */
  KP1_OBS_JOINT_LIMIT_MARGIN = 13, /* 7  :92 */
  KP1_OBS_MODE_FLAG = 20,        /* 4  :88 */
  KP1_OBS_NEXT_WP_ORI_ERR = 24,  /* 3  zeros :71 */
  KP1_OBS_NEXT_WP_POS_ERR = 27,  /* 3  zeros :70 */
  KP1_OBS_PREV_ACTION = 30,      /* 7  :80 */
  KP1_OBS_PROGRESS = 37,         /* 3  :89-91 */
  KP1_OBS_Q = 40,                /* 7  :78 */
  KP1_OBS_TASK_TYPE = 47,        /* 3  :87 */
  KP1_OBS_WP_ORI_ERR = 50,       /* 3  zeros :63 */
  KP1_OBS_WP_POS_ERR = 53        /* 3  zeros :62 */
};

typedef double kp1_f64;
typedef int32_t kp1_i32;

/* ---- Phase1EnvConfig scalars; arm_kinematic_env.py:32-66 ------------------------------ */
#define KP1_ENV_FIELDS(X)                                              \
  X(i32, mode, 0)                                                      \
  X(i32, episode_length, 75)                                           \
  X(i32, dwell_steps_target, 3)                                        \
  X(i32, dynamic_action_delta_scale_enabled, 0)                        \
  X(f64, goal_sample_margin_fraction, 0.10)                            \
  X(f64, start_sample_margin_fraction, 0.20)                           \
  X(f64, action_delta_scale, 1.0)                                      \
  X(f64, dynamic_action_delta_scale_near_pos_threshold_m, 0.0)         \
  X(f64, dynamic_action_delta_scale_far_pos_threshold_m, 0.0)          \
  X(f64, dynamic_action_delta_scale_near_multiplier, 1.0)              \
  X(f64, dynamic_action_delta_scale_far_multiplier, 1.0)               \
  X(f64, dock_action_delta_scale, 0.0)                                 \
  X(f64, dock_residual_action_limit, 1.0)                              \
  X(f64, dock_delta_q_change_limit_scale, 0.0)                         \
  X(f64, dock_dynamic_action_limit_near_pos_threshold_m, 0.0)          \
  X(f64, dock_dynamic_action_limit_far_pos_threshold_m, 0.0)           \
  X(f64, dock_dynamic_residual_action_limit_near, 1.0)                 \
  X(f64, dock_dynamic_residual_action_limit_far, 1.0)                  \
  X(f64, dock_dynamic_delta_q_change_limit_scale_near, 0.0)            \
  X(f64, dock_dynamic_delta_q_change_limit_scale_far, 0.0)

/* ---- ApproachRewardConfig; reward_approach.py:13-72 (milestone tuples handled apart) --- */
#define KP1_APPROACH_REWARD_FIELDS(X)                   \
  X(f64, position_progress_weight, 8.0)                 \
  X(f64, orientation_progress_weight, 1.0)              \
  X(f64, near_field_orientation_progress_weight, 2.0)   \
  X(f64, pre_near_goal_pos_threshold_m, 0.12)           \
  X(f64, near_goal_pos_threshold_m, 0.05)               \
  X(f64, near_goal_ori_threshold_rad, 0.35)             \
  X(f64, coarse_orientation_bonus_threshold_rad, 0.35)  \
  X(f64, near_field_orientation_center_weight, 0.0)     \
  X(i32, use_orientation_gate, 0)                       \
  X(i32, n_orientation_milestones, 0)                   \
  X(f64, pre_near_goal_bonus, 0.03)                     \
  X(f64, near_goal_bonus, 0.10)                         \
  X(f64, near_goal_bonus_decay, 0.5)                    \
  X(f64, pre_near_to_near_progress_weight, 0.0)         \
  X(f64, coarse_orientation_bonus, 0.04)                \
  X(f64, handover_pos_threshold_m, 0.0)                 \
  X(f64, handover_ori_threshold_rad, 0.0)               \
  X(f64, handover_bonus, 0.0)                           \
  X(f64, handover_retention_bonus, 0.0)                 \
  X(f64, handover_dwell_bonus, 0.0)                     \
  X(f64, handover_leave_penalty, 0.0)                   \
  X(f64, handover_regression_weight, 0.0)               \
  X(f64, handover_smoothness_multiplier, 1.0)           \
  X(f64, dock_coarse_ready_pos_threshold_m, 0.0)        \
  X(f64, dock_coarse_ready_ori_threshold_rad, 0.0)      \
  X(f64, dock_coarse_ready_action_threshold, 0.0)       \
  X(f64, dock_coarse_ready_dq_threshold, 0.0)           \
  X(f64, dock_coarse_ready_bonus, 0.0)                  \
  X(f64, dock_coarse_ready_retention_bonus, 0.0)        \
  X(f64, dock_coarse_ready_dwell_bonus, 0.0)            \
  X(f64, dock_coarse_ready_leave_penalty, 0.0)          \
  X(f64, dock_coarse_ready_regression_weight, 0.0)      \
  X(f64, finisher_ready_pos_threshold_m, 0.0)           \
  X(f64, finisher_ready_ori_threshold_rad, 0.0)         \
  X(f64, finisher_ready_action_threshold, 0.0)          \
  X(f64, finisher_ready_dq_threshold, 0.0)              \
  X(f64, finisher_ready_bonus, 0.0)                     \
  X(f64, finisher_ready_retention_bonus, 0.0)           \
  X(f64, finisher_ready_dwell_bonus, 0.0)               \
  X(f64, finisher_ready_leave_penalty, 0.0)             \
  X(f64, finisher_ready_regression_weight, 0.0)         \
  X(f64, near_handoff_pos_threshold_m, 0.0)             \
  X(f64, near_handoff_ori_threshold_rad, 0.0)           \
  X(f64, near_handoff_action_weight, 0.0)               \
  X(f64, near_handoff_dq_weight, 0.0)                   \
  X(f64, near_handoff_motion_bonus_weight, 0.0)         \
  X(f64, near_handoff_settle_bonus_weight, 0.0)         \
  X(f64, same_step_alignment_bonus, 0.0)                \
  X(f64, dwell_bonus, 0.12)                             \
  X(f64, drift_penalty_weight, 3.0)                     \
  X(i32, drift_penalty_escalation_start, 2)             \
  X(i32, reserved0, 0)                                  \
  X(f64, drift_penalty_escalation_per_count, 0.5)       \
  X(f64, near_goal_leave_penalty, 0.0)                  \
  X(f64, action_magnitude_weight, 0.002)                \
  X(f64, action_delta_weight, 0.004)                    \
  X(f64, joint_limit_penalty_weight, 0.05)              \
  X(f64, success_bonus, 1.0)

/* ---- DockRewardConfig; reward_dock.py:13-102 ------------------------------------------ */
#define KP1_DOCK_REWARD_FIELDS(X)                        \
  X(f64, position_progress_weight, 6.0)                  \
  X(f64, orientation_progress_weight, 5.0)               \
  X(f64, stay_in_zone_bonus, 0.08)                       \
  X(f64, dwell_bonus, 0.18)                              \
  X(f64, leave_zone_penalty, 0.25)                       \
  X(f64, working_range_bonus, 0.0)                       \
  X(f64, working_range_dwell_bonus, 0.0)                 \
  X(i32, working_range_dwell_start, 2)                   \
  X(i32, strict_center_dwell_start, 2)                   \
  X(f64, working_range_exit_penalty, 0.0)                \
  X(f64, drift_penalty_position_weight, 4.0)             \
  X(f64, drift_penalty_orientation_weight, 2.0)          \
  X(f64, action_magnitude_weight, 0.006)                 \
  X(f64, action_delta_weight, 0.012)                     \
  X(f64, joint_limit_penalty_weight, 0.05)               \
  X(f64, success_bonus, 2.0)                             \
  X(f64, tight_pose_pos_threshold_m, 0.005)              \
  X(f64, tight_pose_ori_threshold_rad, 0.05)             \
  X(f64, tight_pose_bonus, 0.0)                          \
  X(f64, tight_pose_dwell_bonus, 0.0)                    \
  X(f64, strict_pose_leave_penalty, 0.0)                 \
  X(f64, strict_center_reward_weight, 0.0)               \
  X(f64, strict_center_position_weight, 0.0)             \
  X(f64, strict_center_orientation_weight, 0.0)          \
  X(f64, strict_center_small_action_bonus_weight, 0.0)   \
  X(f64, strict_center_small_action_pos_radius_m, 0.0)   \
  X(f64, strict_center_small_action_ori_radius_rad, 0.0) \
  X(f64, strict_center_small_action_scale, 0.0)          \
  X(f64, strict_center_small_action_power, 2.0)          \
  X(f64, strict_center_dwell_bonus_weight, 0.0)          \
  X(i32, strict_center_dwell_escalation_start, 5)        \
  X(i32, reserved0, 0)                                   \
  X(f64, strict_center_dwell_escalation_per_step, 0.0)   \
  X(f64, strict_zone_drift_penalty_multiplier, 1.0)      \
  X(f64, strict_zone_action_penalty_multiplier, 1.0)     \
  X(f64, tight_position_shaping_radius_m, 0.0)           \
  X(f64, tight_position_shaping_weight, 0.0)             \
  X(f64, tight_orientation_shaping_radius_rad, 0.0)      \
  X(f64, tight_orientation_shaping_weight, 0.0)          \
  X(f64, convergence_position_radius_m, 0.0)             \
  X(f64, convergence_position_progress_weight, 0.0)      \
  X(f64, convergence_orientation_radius_rad, 0.0)        \
  X(f64, convergence_orientation_progress_weight, 0.0)   \
  X(f64, position_first_orientation_pos_threshold_m, 0.0)\
  X(f64, position_first_orientation_pre_scale, 1.0)      \
  X(f64, action_delta_violation_threshold, 0.0)          \
  X(f64, action_delta_violation_weight, 0.0)             \
  X(f64, delta_q_change_penalty_threshold, 0.0)          \
  X(f64, delta_q_change_penalty_weight, 0.0)             \
  X(f64, entry_action_penalty_near_pos_threshold_m, 0.0) \
  X(f64, entry_action_penalty_far_pos_threshold_m, 0.0)  \
  X(f64, entry_action_penalty_near_multiplier, 1.0)      \
  X(f64, entry_action_penalty_far_multiplier, 1.0)       \
  X(f64, basin_outer_radius_m, 0.0)                      \
  X(f64, basin_inner_radius_m, 0.0)                      \
  X(f64, basin_dwell_radius_m, 0.0)                      \
  X(f64, basin_outer_bonus, 0.0)                         \
  X(f64, basin_inner_bonus, 0.0)                         \
  X(f64, basin_dwell_bonus, 0.0)                         \
  X(f64, basin_outer_exit_penalty, 0.0)                  \
  X(f64, basin_inner_exit_penalty, 0.0)                  \
  X(f64, basin_dwell_break_penalty, 0.0)                 \
  X(f64, basin_drift_penalty_weight, 0.0)                \
  X(f64, near_strict_pos_threshold_m, 0.0)               \
  X(f64, near_strict_ori_threshold_rad, 0.0)             \
  X(f64, preserve_state_bonus, 0.0)                      \
  X(f64, preserve_position_tolerance_m, 0.0)             \
  X(f64, preserve_orientation_tolerance_rad, 0.0)        \
  X(f64, strict_hold_bonus, 0.0)                         \
  X(f64, low_motion_bonus, 0.0)                          \
  X(f64, low_motion_action_threshold, 0.0)               \
  X(f64, low_motion_dq_threshold, 0.0)                   \
  X(f64, tiny_correction_bonus, 0.0)                     \
  X(f64, tiny_correction_action_threshold, 0.0)          \
  X(f64, worse_than_entry_position_weight, 0.0)          \
  X(f64, worse_than_entry_orientation_weight, 0.0)       \
  X(f64, worse_than_entry_position_tolerance_m, 0.0)     \
  X(f64, worse_than_entry_orientation_tolerance_rad, 0.0)\
  X(f64, near_strict_regression_multiplier, 1.0)         \
  X(f64, aggressive_action_weight, 0.0)                  \
  X(f64, aggressive_action_threshold, 0.0)               \
  X(f64, dq_penalty_weight, 0.0)                         \
  X(f64, dq_penalty_threshold, 0.0)                      \
  X(f64, near_strict_action_penalty_multiplier, 1.0)     \
  X(f64, near_strict_dq_penalty_multiplier, 1.0)

/* ---- TerminationConfig; termination.py:10-17 ------------------------------------------ */
#define KP1_TERMINATION_FIELDS(X)          \
  X(i32, max_episode_steps, 75)            \
  X(i32, success_dwell_steps, 2)           \
  X(i32, require_orientation, 0)           \
  X(i32, terminate_on_success, 1)          \
  X(f64, success_pos_threshold_m, 0.06)    \
  X(f64, success_ori_threshold_rad, 0.15)

/* ---- ObservationBuilderConfig; observation_builder.py:17-20 --------------------------- */
#define KP1_OBSERVATION_FIELDS(X) \
  X(f64, pos_err_scale_m, 0.5)    \
  X(f64, ori_err_scale_rad, 3.141592653589793)

/* ---- workspace_stage_sampling dict; reset_samplers.py:344-390.
 *      KP1_UNSET marks "key absent": the sampler then uses the reference's
 *      current-stage-dependent default (e.g. min(5, current)). */
#define KP1_UNSET (-2147483647 - 1)
#define KP1_STAGE_SAMPLING_FIELDS(X)          \
  X(i32, enabled, 0)                          \
  X(i32, previous_stage_min_index, 0)         \
  X(i32, old_workspace_max_stage_index, KP1_UNSET) \
  X(i32, reserved0, 0)                        \
  X(f64, current_stage_ratio, 0.50)           \
  X(f64, previous_stage_ratio, 0.25)          \
  X(f64, old_workspace_replay_ratio, 0.20)    \
  X(f64, failure_replay_ratio, 0.05)

/* ---- workspace_stage_sampling.random_start_pair_sampling; reset_samplers.py:213-341 ---- */
#define KP1_RANDOM_START_FIELDS(X)                     \
  X(i32, enabled, 0)                                   \
  X(i32, home_stage_index, 0)                          \
  X(i32, old_success_max_stage_index, KP1_UNSET)       \
  X(i32, frontier_min_stage_index, KP1_UNSET)          \
  X(i32, frontier_max_stage_index, KP1_UNSET)          \
  X(i32, known_target_max_stage_index, KP1_UNSET)      \
  X(i32, frontier_target_min_stage_index, KP1_UNSET)   \
  X(i32, frontier_target_max_stage_index, KP1_UNSET)   \
  X(i32, stress_target_min_stage_index, KP1_UNSET)     \
  X(i32, stress_target_max_stage_index, KP1_UNSET)     \
  X(i32, mixed_target_max_stage_index, KP1_UNSET)      \
  X(i32, has_stress_start_margin_fraction, 0)          \
  X(i32, has_random_valid_start_margin_fraction, 0)    \
  X(i32, reserved0, 0)                                 \
  X(f64, home_start_ratio, 0.15)                       \
  X(f64, old_successful_start_ratio, 0.25)             \
  X(f64, random_valid_q_start_ratio, 0.25)             \
  X(f64, frontier_pair_ratio, 0.20)                    \
  X(f64, failure_recovery_start_ratio, 0.10)           \
  X(f64, stress_start_ratio, 0.05)                     \
  X(f64, stress_start_margin_fraction, 0.0)            \
  X(f64, random_valid_start_margin_fraction, 0.0)      \
  X(f64, min_pair_joint_l2, 0.0)

/* ---- DockResetConfig scalars; reset_samplers.py:48-65 (vectors below) ------------------ */
#define KP1_DOCK_RESET_FIELDS(X)                      \
  X(i32, close_bucket_max_attempts, 128)              \
  X(i32, reserved0, 0)                                \
  X(f64, close_bucket_probability, 0.0)               \
  X(f64, close_bucket_min_pos_error_m, 0.005)         \
  X(f64, close_bucket_max_pos_error_m, 0.020)         \
  X(f64, close_bucket_min_ori_error_rad, 0.0)         \
  X(f64, close_bucket_max_ori_error_rad, 0.12)        \
  X(f64, handoff_state_probability, 0.0)

#define KP1_DECL_f64(name) kp1_f64 name;
#define KP1_DECL_i32(name) kp1_i32 name;
#define KP1_DECL(type, name, dflt) KP1_DECL_##type(name)

typedef struct kp1_env_scalars { KP1_ENV_FIELDS(KP1_DECL) } kp1_env_scalars;
typedef struct kp1_approach_reward {
  KP1_APPROACH_REWARD_FIELDS(KP1_DECL)
  kp1_f64 orientation_milestone_thresholds_rad[KP1_MAX_MILESTONES];
  kp1_f64 orientation_milestone_bonuses[KP1_MAX_MILESTONES];
} kp1_approach_reward;
typedef struct kp1_dock_reward { KP1_DOCK_REWARD_FIELDS(KP1_DECL) } kp1_dock_reward;
typedef struct kp1_termination { KP1_TERMINATION_FIELDS(KP1_DECL) } kp1_termination;
typedef struct kp1_observation { KP1_OBSERVATION_FIELDS(KP1_DECL) } kp1_observation;
typedef struct kp1_stage_sampling { KP1_STAGE_SAMPLING_FIELDS(KP1_DECL) } kp1_stage_sampling;
typedef struct kp1_random_start {
  KP1_RANDOM_START_FIELDS(KP1_DECL)
  kp1_f64 failure_recovery_q_noise[KP1_NJ];   /* default 0.04 each; reset_samplers.py:274 */
  kp1_f64 initial_dq_noise[KP1_NJ];           /* default 0; :283 */
  kp1_f64 initial_prev_action_noise[KP1_NJ];  /* default 0; :284 */
} kp1_random_start;
typedef struct kp1_dock_reset {
  KP1_DOCK_RESET_FIELDS(KP1_DECL)
  kp1_f64 goal_q[KP1_NJ];
  kp1_f64 goal_noise[KP1_NJ];
  kp1_f64 init_q_noise[KP1_NJ];
  kp1_f64 close_init_q_noise[KP1_NJ];
} kp1_dock_reset;

/* JointSpec table; joint_limits.py:24-47 */
typedef struct kp1_joint_specs {
  kp1_f64 lower[KP1_NJ];
  kp1_f64 upper[KP1_NJ];
  kp1_f64 delta_limit[KP1_NJ];
} kp1_joint_specs;

/* CurriculumStageConfig; curriculum.py:21-33 */
typedef struct kp1_stage {
  kp1_f64 start_q[KP1_NJ];
  kp1_f64 goal_q[KP1_NJ];
  kp1_f64 start_noise[KP1_NJ];
  kp1_f64 goal_noise[KP1_NJ];
} kp1_stage;

/* Phase1EnvConfig; arm_kinematic_env.py:32-66 */
typedef struct kp1_config {
  kp1_env_scalars env;
  kp1_i32 curriculum_enabled;      /* PointCurriculumConfig.enabled; curriculum.py:83 */
  kp1_i32 n_stages;
  kp1_joint_specs joints;
  kp1_stage stages[KP1_MAX_STAGES];
  kp1_stage_sampling stage_sampling;
  kp1_random_start random_start;
  kp1_approach_reward reward;
  kp1_dock_reward dock_reward;
  kp1_dock_reset dock_reset;
  kp1_termination termination;
  kp1_observation observation;
} kp1_config;

/* One stored finisher handoff state; reset_samplers.py:32-45 (HandoffResetState). */
typedef struct kp1_handoff_state {
  kp1_f64 initial_q[KP1_NJ];
  kp1_f64 goal_q[KP1_NJ];
  kp1_f64 goal_pose6[6];
  kp1_f64 initial_dq[KP1_NJ];
  kp1_f64 initial_prev_action[KP1_NJ];
} kp1_handoff_state;

/* numpy Generator(PCG64) stream state of one env (bit_generator.state); the env owns one
 * stream, seeded like np.random.default_rng(seed) at reset(seed=...); arm_kinematic_env.py:80,103-104. */
typedef struct kp1_rng_state {
  uint64_t state_hi, state_lo;
  uint64_t inc_hi, inc_lo;
  uint32_t has_uint32;
  uint32_t uinteger;
} kp1_rng_state;

/* Explicit reset options, SoA over the envs being reset (any pointer may be NULL = "not given");
 * arm_kinematic_env.py:113-117 (opts.get("initial_q") ...).  Host pointers, f64, [n][7] / [n][6]. */
typedef struct kp1_reset_opts {
  const double* initial_q;
  const double* initial_dq;
  const double* initial_prev_action;
  const double* goal_q;
  const double* goal_pose6;
  int32_t policy_mode; /* -1 = keep config.mode; else KP1_MODE_* (opts["policy_mode"], :111) */
} kp1_reset_opts;

/* Per-env info, device SoA [field][N] in the handle's real type unless noted; written by
 * kp1_step / kp1_reset.  Mirrors the info keys consumers read; arm_kinematic_env.py:384-423,359-364. */
typedef struct kp1_info_view {
  const void* position_error_norm;    /* real[N] */
  const void* orientation_error_norm; /* real[N] */
  const void* min_position_error;     /* real[N] */
  const void* executed_delta_q_l2;    /* real[N]  info["executed_delta_q_l2"] */
  const void* action_l2;              /* real[N]  norm of the clipped action */
  const void* delta_q_change_l2;      /* real[N] */
  const void* q;                      /* real[7][N] */
  const void* dq;                     /* real[7][N] */
  const void* prev_action;            /* real[7][N]  env._prev_action */
  const void* goal_q;                 /* real[7][N] */
  const void* goal_pose6;             /* real[6][N] */
  const void* ee_pose6;               /* real[6][N] */
  const void* entry_metrics;          /* real[4][N]  pos, ori, action_l2, dq_norm at reset (:425-430) */
  const int32_t* episode_step;        /* i32[N] */
  const int32_t* dwell_count;
  const int32_t* near_goal_entry_count;
  const int32_t* near_goal_drift_count;
  const int32_t* flags;               /* bit0 pre_near_goal_hit, bit1 near_goal_hit, bit2 success(last step) */
  const int32_t* stage_index;         /* stage the last reset sampled from */
  int32_t n_envs;
  int32_t real_type;
} kp1_info_view;

/* bits of the per-step done byte written by kp1_step */
enum { KP1_DONE_TERMINATED = 1, KP1_DONE_TRUNCATED = 2, KP1_DONE_SUCCESS = 4, KP1_DONE_INVALID = 8 };

typedef struct kp1_env kp1_env; /* opaque handle: N environments on one device */

const char* kp1_last_error(void);
int kp1_abi_version(void);
/* sizeof(kp1_config) as compiled, so a binding can check its mirror of the struct. */
uint64_t kp1_config_size(void);
/* Fill *cfg with the reference defaults (Phase1EnvConfig() with default_joint_specs()). */
int kp1_config_default(kp1_config* cfg);

/* ArmKinematicEnv.__init__ x N; arm_kinematic_env.py:74-100.  Env i owns PCG64 stream
 * default_rng(seed0 + first_env_id + i) (SB3 make_vec_env seeds env i with seed + i).
 * real_type KP1_REAL_F32 (production) or KP1_REAL_F64 (strict parity). stream: hipStream_t or NULL. */
int kp1_create(const kp1_config* cfg, int32_t n_envs, int32_t device, int32_t real_type,
               uint64_t seed0, uint64_t first_env_id, void* stream, kp1_env** out);
int kp1_destroy(kp1_env* env);
/* change the HIP stream later launches of this handle are ordered on (e.g. a stream under hipGraph capture) */
int kp1_set_stream(kp1_env* env, void* stream);
int kp1_num_envs(const kp1_env* env);

/* set_curriculum_stage / get_curriculum_stage (all envs, like VecEnv.env_method); :446-452 */
int kp1_set_stage(kp1_env* env, int32_t stage_index);
int kp1_get_stage(const kp1_env* env, int32_t* stage_index);
/* set_policy_mode; :454-457 */
int kp1_set_mode(kp1_env* env, int32_t mode);
/* apply_dock_training_stage / config replacement (env scalars + dock_reset block only); :459-487 */
int kp1_update_config(kp1_env* env, const kp1_config* cfg);
/* DockResetConfig._handoff_states (already filtered by the caller); reset_samplers.py:131-165 */
int kp1_set_handoff_states(kp1_env* env, const kp1_handoff_state* states_host, int32_t n_states);
/* re-seed env streams: env i <- default_rng(seed0 + first_env_id + i); reset(seed=...) :103-104 */
int kp1_seed(kp1_env* env, uint64_t seed0, uint64_t first_env_id);

/* reset(); :102-211.  mask_dev: u8[N] device (NULL = all envs).  opts: explicit options for ALL N envs
 * (rows of masked-out envs ignored) or NULL to sample.  obs_dev: f32[N][56] device (NULL = skip). */
int kp1_reset(kp1_env* env, const uint8_t* mask_dev, const kp1_reset_opts* opts, float* obs_dev);

/* step(); :213-365 with VecEnv auto-reset fused: envs whose episode ends are reset in the same
 * launch (sampling path) and obs_dev holds the NEW episode's first observation, while
 * terminal_obs_dev (may be NULL) receives the last observation of the finished episode for
 * those envs (SB3 "terminal_observation", used for time-limit bootstrapping).
 *   actions_dev  real[N][7]   (unclipped policy output; clipped to [-1,1] inside, :214)
 *   obs_dev      f32[N][56]
 *   reward_dev   real[N]
 *   done_dev     u8[N]  KP1_DONE_* bits
 * auto_reset = 0 leaves finished envs un-reset (evaluators). */
int kp1_step(kp1_env* env, const void* actions_dev, float* obs_dev, void* reward_dev,
             uint8_t* done_dev, float* terminal_obs_dev, int32_t auto_reset);

/* current_observation(); :381 */
int kp1_observe(kp1_env* env, float* obs_dev);
/* info dict as device SoA views valid until the next call on the handle */
int kp1_get_info(kp1_env* env, kp1_info_view* view);

/* ---- batched deterministic evaluator: the per-step bookkeeping of eval_workspace_expansion.py:86-211 (_run_policy /
 * _run_approach_with_handoff) for all episodes of a vectorised run in ONE launch -------------------------------------------------
 * Every episode is one env of `env` (no auto-reset).  After each kp1_step the caller hands over the norm of the action it applied and the
 * done bytes; the kernel reads position / orientation error, executed |dq| and the env state from the handle and updates, for the episodes
 * still alive: step count, sums and finals of |action| and |dq|, final and minimum errors, success, the state snapshot
 * (q, dq, prev_action, goal_q, goal_pose6: what a handoff continues from), the finisher-ready hit / streak bookkeeping and -- the first
 * time the streak reaches handoff_confirm_steps -- the first-confirmed handoff snapshot; then alive &= not done.  All buffers are device
 * memory owned by the caller, layouts as noted (n = kp1_num_envs). */
typedef struct kp1_eval_buffers {
  double* metrics;        /* [8][n]: final pos err, final ori err, min pos err, min ori err, final |action|, final |dq|, sum |action|, sum |dq| */
  int32_t* counters;      /* [4][n]: step_count, max_ready_streak, first_ready_step (-1 = never), current streak */
  uint8_t* flags;         /* [4][n]: alive, success, ready_hit, handoff taken */
  double* state;          /* [n][34]: q 7, dq 7, prev_action 7, goal_q 7, goal_pose6 6 of the last step the episode was alive */
  double* hand_metrics;   /* [8][n]: pos, ori, |action|, |dq|, min pos, min ori, sum |action|, sum |dq| at the handoff step; NULL = no handoff
                             snapshot wanted (_run_policy).  Given: snapshot at the first step with ready_streak >= handoff_confirm_steps
                             (eval_pipeline_ablation.py:103; confirm <= 0 hands over at step 1) */
  int32_t* hand_step;     /* [n] */
  uint8_t* hand_success;  /* [n] */
  double* hand_state;     /* [n][34] */
  int32_t* n_alive;       /* [1]: episodes still alive after this call (lets the caller stop early without reading back [n] flags) */
} kp1_eval_buffers;
/* step == 0: initialise from the freshly reset env (finals = mins = current errors, everything else zero, alive = active[i] or 1 when
 * active is NULL); step >= 1: account env step number `step`.  ready_thresholds = {pos, ori, |action|, |dq|} of the readiness predicate
 * (reward config dock_coarse_ready_*; pos or ori <= 0 disables it, |action| / |dq| <= 0 skip that clause) or NULL for no ready tracking. */
int kp1_eval_accumulate(kp1_env* env, const kp1_eval_buffers* buffers, const double* action_norm, const uint8_t* done, const uint8_t* active,
                        int32_t step, const double* ready_thresholds, int32_t handoff_confirm_steps, void* stream);
/* reward_components of the last step, device real[n_components][N]; names via kp1_component_name */
int kp1_get_reward_components(kp1_env* env, const void** comps_dev, int32_t* n_components);
int kp1_enable_reward_components(kp1_env* env, int32_t enable);
const char* kp1_component_name(int32_t mode, int32_t index);
int kp1_num_components(int32_t mode);

/* direct state access for handoff / route re-target (callers poke _q, _dq, _prev_action, _goal_q,
 * _goal_pose6 then _capture_entry_metrics(); eval_three_stage.py:122, route_sequence_env.py:255-257).
 * Host f64 arrays [N][7] / [N][6]; NULL pointers are skipped.  set recomputes ee_pose6 = FK(q). */
int kp1_get_state(kp1_env* env, double* q, double* dq, double* prev_action, double* goal_q, double* goal_pose6);
int kp1_set_state(kp1_env* env, const double* q, const double* dq, const double* prev_action,
                  const double* goal_q, const double* goal_pose6, int32_t capture_entry_metrics);
int kp1_rng_get(kp1_env* env, kp1_rng_state* states_host /* [N] */);
int kp1_rng_set(kp1_env* env, const kp1_rng_state* states_host /* [N] */);
/* Whole-state snapshot on the device (every real / integer field and the PCG64 streams of all envs), ordered on the env's stream.
   No reference counterpart: the reference never needs one (its envs are Python objects).  Here a hipGraph capture must be preceded by
   one eager execution of every kernel it records; kp1_state_snapshot before that warm-up and kp1_state_restore after it leave the
   episodes and the random streams exactly where they were, so a captured rollout continues bit-identically to an eager one. */
int kp1_state_snapshot(kp1_env* env);
int kp1_state_restore(kp1_env* env);

/* stand-alone batched kinematics (fk_interface.py:21-22, pose_utils.py:21-26): q real[n][7] -> pose6 real[n][6] */
int kp1_fk_pose6(int32_t device, int32_t real_type, const void* q_dev, void* pose6_dev, int64_t n, void* stream);

/* pose_error_components (KP1/kinematics/pose_utils.py:21-30, wrap_to_pi :11-12) through the device function the step and reset
 * kernels call: curr / goal real[n][6] -> pos_err real[n][3], ori_err real[n][3] (each component wrapped to [-pi, pi)),
 * norms real[n][2] = (|pos_err|, |ori_err|).  Any output may be NULL. */
int kp1_pose_error(int32_t device, int32_t real_type, const void* curr_dev, const void* goal_dev, void* pos_err_dev, void* ori_err_dev,
                   void* norms_dev, int64_t n, void* stream);
/* joint utilities of KP1/kinematics/joint_limits.py through the device functions of the hot path: q, dq real[n][7] ->
 * clipped = clip_joint_configuration(q) :133-135, margin = joint_limit_margin(clipped) :166-174, q_norm = normalize_joint_positions(q)
 * :153-158, dq_norm = normalize_joint_deltas(dq) :161-163, all real[n][7].  Limits come from `cfg->joints`.  Outputs may be NULL. */
int kp1_joint_utils(int32_t device, int32_t real_type, const kp1_config* cfg, const void* q_dev, const void* dq_dev, void* clipped_dev,
                    void* margin_dev, void* q_norm_dev, void* dq_norm_dev, int64_t n, void* stream);

/* numpy-compatible seeding helper: PCG64 state of np.random.default_rng(seed) (host side) */
int kp1_rng_seed_state(uint64_t seed, kp1_rng_state* out);

#ifdef __cplusplus
}
#endif
#endif /* KP1_H */
