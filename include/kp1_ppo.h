/*
 * kp1_ppo.h -- C ABI of the PPO-side device kernels of the MI355X kinematic_phase1 engine.
 *
 * The reference delegates PPO to stable_baselines3.PPO("MultiInputPolicy") (call sites:
 * kinematic_phase1/train_workspace_expansion.py:199,232; training/train_dock_policy.py:99,102).
 * SB3 2.8.0 (final_codes_docker/Dockerfile.demo:32) is absent from the reference tree and from this
 * image, so these kernels restate SB3's published semantics ("parity unpinned")
 * and are checked against plain PyTorch fp32 references of the same ops (tests/test_ppo_kernels_gpu.py).
 *
 * All pointers are HIP device pointers unless named *_host.  Layout [T][N] = time-major.
 * Every function returns KP1_OK or a negative kp1_status (kp1.h); text via kp1_last_error().
 */
#ifndef KP1_PPO_H
#define KP1_PPO_H

#include <stdint.h>

#include "kp1.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RolloutBuffer.compute_returns_and_advantage (SB3 common/buffers.py): reverse GAE(lambda) scan, one lane per env.
 *   delta_t = r_t + gamma * V_{t+1} * (1 - done_t) - V_t ;  A_t = delta_t + gamma*lambda*(1 - done_t) * A_{t+1}
 *   V_T = last_values ; returns = A + V.   done_t = (done bits of step t) & (TERMINATED|TRUNCATED).
 * rewards/values/advantages/returns f32 [T][N]; dones u8 [T][N] (KP1_DONE_* bits as written by kp1_step). */
int kp1_gae_scan(int32_t device, const float* rewards, const float* values, const uint8_t* dones, const float* last_values,
                 float gamma, float gae_lambda, float* advantages, float* returns, int32_t T, int32_t N, void* stream);

/* Time-limit bootstrap (SB3 on_policy_algorithm.collect_rollouts): rewards[t][i] += gamma * terminal_values[t][i]
 * where step t of env i was truncated and not terminated. */
int kp1_bootstrap_truncated(int32_t device, float* rewards, const float* terminal_values, const uint8_t* dones, float gamma,
                            int64_t count, void* stream);

/* Advantage statistics of every minibatch of an epoch in ONE launch (SB3 normalises per minibatch: ppo.py train(),
 * advantages = (adv - adv.mean()) / (adv.std() + 1e-8)).  Minibatch b = idx[b*minibatch .. min((b+1)*minibatch, total)) (idx NULL =
 * identity).  out_sums f64 [n_minibatches][3] = (sum, sum of squares, count): the host side (or, data parallel, ONE all-reduce per
 * epoch instead of one per minibatch) turns them into (mean, 1/(std+1e-8)) for kp1_mlp_loss_grad's adv_stats_dev. */
int kp1_adv_minibatch_sums(int32_t device, const float* advantages, const int64_t* idx, int64_t total, int64_t minibatch, double* out_sums,
                           void* stream);
/* Minibatch shuffle of one epoch (SB3 RolloutBuffer.get: indices = np.random.permutation(buffer_size * n_envs)): out[i] = P(i), i in [0, n),
 * with P a keyed pseudo-random permutation of [0, n) evaluated per element -- four rounds of (odd multiply + add, xorshift) modulo
 * 2^ceil(log2 n), each invertible, walked until the value lands below n (cycle walking; the index-shuffle construction of input
 * pipelines).  One elementwise launch instead of the radix sort + merge passes of a sort-based randperm (0.16 ms for 524288 elements,
 * eight times per PPO iteration).  keys: 8 x u32 (host memory) drawn by the caller per epoch; n <= 2^31. */
int kp1_random_permutation(int32_t device, int64_t n, const uint32_t* keys, int64_t* out, void* stream);

/* (sum, sum of squares, count) f64 [n_minibatches][3] (after the data-parallel all-reduce, if any) -> f32 [n_minibatches][2] =
 * (mean, 1 / (std + 1e-8)) with torch's unbiased std, in ONE launch (the same arithmetic in f64 as the tensor expressions it replaces:
 * thirteen elementwise launches per epoch inside the update graph). */
int kp1_adv_minibatch_stats(int32_t device, const double* sums, int64_t n_minibatches, float* out_stats, void* stream);

/* ---- device-resident PointCurriculumCallback (kinematic_phase1/training/callbacks.py:32-101) --------------------
 * The callback scans (done, info["success"]) in env order after every VecEnv step and may promote the stage for ALL
 * envs; resets inside that step already happened with the old stage.  Keeping the tracker on the device removes the
 * per-step host sync: kp1_curriculum_observe runs after each kp1_step on the same stream and publishes the stage in
 * device memory, from which the next kp1_step reads it (kp1_bind_stage_ptr). */
#define KP1_CURRICULUM_MAX_WINDOW 1024
#define KP1_CURRICULUM_MAX_HISTORY 64
typedef struct kp1_curriculum_event {
  int64_t total_timesteps; /* env steps taken (all ranks) when the promotion fired, like history["total_timesteps"] */
  int32_t from_stage, to_stage;
  double trigger_success_rate;
} kp1_curriculum_event;
typedef struct kp1_curriculum_state {
  int32_t stage_index;          /* current_stage_index; first word so an env kernel can read it as int32 */
  int32_t stage_episode_count;
  int32_t ring_len, ring_head;
  int32_t window_episodes, min_episodes_per_stage, max_stage_index, n_events;
  double success_rate_threshold;
  int64_t num_timesteps;
  int32_t ring[KP1_CURRICULUM_MAX_WINDOW];
  kp1_curriculum_event events[KP1_CURRICULUM_MAX_HISTORY];
} kp1_curriculum_state;

/* allocate + initialise a tracker in device memory (callbacks.py:33-51) */
int kp1_curriculum_create(int32_t device, double success_rate_threshold, int32_t window_episodes, int32_t min_episodes_per_stage,
                          int32_t max_stage_index, int32_t initial_stage_index, kp1_curriculum_state** out_dev);
int kp1_curriculum_destroy(int32_t device, kp1_curriculum_state* st_dev);
/* _on_step (callbacks.py:71-92): consume dones[0..n) (KP1_DONE_* bits) in index order; steps_per_call = env steps this
 * call represents (n_envs * world_size) for the num_timesteps clock. */
int kp1_curriculum_observe(int32_t device, kp1_curriculum_state* st_dev, const uint8_t* dones, int32_t n, int32_t steps_per_call, void* stream);
/* Data-parallel form: dones = the all-gathered [world][chunk_steps][n_local] done bytes of a chunk of env steps (rank-major).  Replayed
 * step by step and, inside a step, rank by rank = global env id order, i.e. exactly the (done, info) sequence the reference callback
 * would see on one VecEnv of world * n_local envs; the clock advances by world * n_local per env step.  A promotion takes effect for
 * the resets of the NEXT chunk (every rank runs this on identical bytes, so all ranks switch together). */
int kp1_curriculum_observe_chunk(int32_t device, kp1_curriculum_state* st_dev, const uint8_t* dones, int32_t n_local, int32_t chunk_steps,
                                 int32_t world, void* stream);
int kp1_curriculum_read(int32_t device, const kp1_curriculum_state* st_dev, kp1_curriculum_state* out_host, void* stream);
/* make kp1_step take its curriculum stage from *stage_dev (e.g. &tracker->stage_index) instead of kp1_set_stage */
int kp1_bind_stage_ptr(kp1_env* env, const int32_t* stage_dev);

/* ---- device-resident DockReverseCurriculumCallback (kinematic_phase1/training/callbacks.py:104-212) -------------
 * The Finisher's reverse curriculum: after every VecEnv step the callback scans (done, info["success"]) in env order, keeps the last
 * `window_episodes` success bits and, when the current stage has seen `min_episodes` episodes and the newest `stage window` of them reach
 * `success_rate_threshold`, applies the next stage's overrides to all envs (apply_dock_training_stage, arm_kinematic_env.py:459-487).
 * A stage here carries the RESOLVED values (base config overlaid by the payloads of stages 0..k in order, which is what the env holds after
 * k promotions); the tracker writes them into the device config the step / reset kernels read. */
#define KP1_DOCK_CURRICULUM_MAX_STAGES 16
#define KP1_DOCK_CURRICULUM_MAX_WINDOW 1024
#define KP1_DOCK_CURRICULUM_MAX_HISTORY 32
typedef struct kp1_dock_curriculum_stage {
  double action_delta_scale, dock_residual_action_limit, dock_delta_q_change_limit_scale;             /* env scalars */
  double close_bucket_probability, close_bucket_min_pos_error_m, close_bucket_max_pos_error_m, close_bucket_max_ori_error_rad,
         handoff_state_probability;                                                                    /* dock_reset scalars */
  double close_init_q_noise[KP1_NJ], init_q_noise[KP1_NJ];
  double success_rate_threshold;     /* stage.get("success_rate_threshold", 1.0) */
  int32_t min_episodes;              /* stage.get("min_episodes", window_episodes) */
  int32_t window_episodes;           /* stage.get("window_episodes", window_episodes) */
  /* handoff-state buffer of this stage (stages may override handoff_state_buffer_path and the three handoff_state_max_* filters,
   * reset_samplers.py:131-165): a slice [offset, offset + count) of the buffer given to kp1_set_handoff_states, which then holds the
   * filtered lists of all stages back to back.  count < 0: the stage does not manage the buffer (the env's whole buffer stays in use). */
  int32_t handoff_offset, handoff_count;
} kp1_dock_curriculum_stage;
typedef struct kp1_dock_curriculum_event {
  int64_t total_timesteps;
  int32_t from_stage, to_stage, stage_episode_count, reserved0;
  double trigger_success_rate;
} kp1_dock_curriculum_event;
typedef struct kp1_dock_curriculum_state {
  int32_t stage_index, stage_episode_count, ring_len, ring_head, window_episodes, n_stages, n_events, reserved0;
  int64_t num_timesteps;
  kp1_dock_curriculum_stage stages[KP1_DOCK_CURRICULUM_MAX_STAGES];
  kp1_dock_curriculum_event events[KP1_DOCK_CURRICULUM_MAX_HISTORY];
  uint8_t ring[KP1_DOCK_CURRICULUM_MAX_WINDOW];
} kp1_dock_curriculum_state;
/* allocate the tracker next to a dock-mode env and apply stage 0 (_on_training_start) */
int kp1_dock_curriculum_create(kp1_env* env, const kp1_dock_curriculum_stage* stages_host, int32_t n_stages, int32_t window_episodes,
                               kp1_dock_curriculum_state** out_dev);
int kp1_dock_curriculum_destroy(kp1_env* env, kp1_dock_curriculum_state* st_dev);
/* _on_step: dones = [world][chunk_steps][n_local] done bytes (one process, one env step: world = chunk_steps = 1), replayed step by step
 * and rank by rank = global env order; the clock advances by world * n_local per env step */
int kp1_dock_curriculum_observe(kp1_env* env, kp1_dock_curriculum_state* st_dev, const uint8_t* dones, int32_t n_local, int32_t chunk_steps, int32_t world,
                                void* stream);
/* copy the tracker to the host; also brings the env handle's host-side config mirror in step with the stage the device applied */
int kp1_dock_curriculum_read(kp1_env* env, const kp1_dock_curriculum_state* st_dev, kp1_dock_curriculum_state* out_host, void* stream);

/* ---- actor-critic MLP on the matrix cores (fp32-in / fp32-accumulate MFMA, exact f32) ---------------------------
 * SB3 MultiInputActorCriticPolicy with net_arch pi = vf = [H, H], tanh:
 *   h1 = tanh(x W1^T + b1); h2 = tanh(h1 W2^T + b2); mean = h2p Wa^T + ba (7); value = h2v Wv^T + bv (1).
 * Both nets (index 0 = policy, 1 = value) run in one launch (grid.z).  H must be a multiple of 128 or equal to 64.
 * Kernel-format weights ("kw", see kp1_mlp_pack_weights) hold W1 zero-padded to 64 input columns, W2 and W2^T. */
#define KP1_MLP_IN 56
#define KP1_MLP_IN_PAD 64
#define KP1_MLP_IN_ROUTE 80      /* with the route observation keys (route/route_observation.py:14-61); padded to 128 */
#define KP1_MLP_IN_ROUTE_PAD 128
#define KP1_MLP_ACT 7

typedef struct kp1_mlp kp1_mlp; /* workspace: packed weights, activations, gradients for up to max_batch rows */

int kp1_mlp_create(int32_t device, int32_t hidden, int32_t max_batch, kp1_mlp** out);
/* same, for an observation of obs_dim floats: 56 (ArmKinematicEnv, observation_builder.py:29-94) or 80 (the route wrappers with
 * include_route_keys, route/route_env.py:186-207).  Rows are read with pitch obs_dim or the padded width (64 / 128). */
int kp1_mlp_create_ex(int32_t device, int32_t hidden, int32_t obs_dim, int32_t max_batch, kp1_mlp** out);
int kp1_mlp_destroy(kp1_mlp* m);
/* number of f32 parameters in SB3 state_dict order (log_std, pi.0.w, pi.0.b, pi.2.w, pi.2.b, vf.0.w, ..., action_net.w/b, value_net.w/b) */
int64_t kp1_mlp_num_params(int32_t hidden);
int64_t kp1_mlp_num_params_ex(int32_t hidden, int32_t obs_dim);
/* repack the flat SB3-order parameter vector into kernel-format weights (call after every optimiser step) */
int kp1_mlp_pack_weights(kp1_mlp* m, const float* params, void* stream);

/* policy.forward for a rollout step: obs f32 [n][obs_stride] (first 56 columns used; obs_stride 56 or 64 with zero pad),
 * noise f32 [n][7] ~ N(0,1) or NULL (deterministic).  Outputs (any may be NULL):
 *   mean[n][7], value[n], action[n][7] = mean + exp(log_std) * noise, clipped_action[n][7] = clip(action, -1, 1),
 *   log_prob[n] = sum_d N(action_d; mean_d, std_d). */
int kp1_mlp_forward(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, const float* noise, float* mean, float* value,
                    float* action, float* clipped_action, float* log_prob, void* stream);

/* One rollout step in ONE launch: policy.forward on the current observations of ALL envs of `env` (row m = env m; obs f32 [N][obs_stride]),
 * Gaussian sampling, then VecEnv.step(clip(action)) of every env with the auto-reset -- SB3's collect_rollouts body
 * (on_policy_algorithm.py: policy(obs) -> clip -> env.step) without the action round trip through HBM and without a launch for the env.
 * Same results as kp1_mlp_forward followed by kp1_step (tests/test_ppo_kernels_gpu.py).  fp32 handles, hidden 256, 56-float observations;
 * reward components must be off.  value / log_prob may be NULL (value NULL skips the value net); terminal_obs may be NULL. */
int kp1_mlp_forward_env_step(kp1_mlp* m, kp1_env* env, const float* obs, int32_t obs_stride, const float* noise, float* value, float* action,
                             float* log_prob, float* next_obs, float* reward, uint8_t* done, float* terminal_obs, void* stream);

/* one PPO minibatch: forward + loss + full backward.  Rows are gathered through idx (int64 [n] into the [total] axis, or
 * NULL = rows 0..n).  grad_out f32 [num_params] receives d loss / d params in SB3 order (overwritten), with
 *   loss = sum_i[-min(r_i A_i, clip(r_i, 1-c, 1+c) A_i)] * inv_count + vf_coef * sum_i (R_i - V_i)^2 * inv_count - ent_coef * H,
 * r = exp(logp - old_logp), H = entropy of the diagonal Gaussian (state independent).
 * stats_out f32[4] += (policy_loss, value_loss, entropy, approx_kl) of this minibatch.
 * Every producer kernel writes per-workgroup partials with plain stores and one finalize kernel sums them in a fixed order:
 * no float atomics touch a gradient, results are bitwise reproducible.  grad_is_zero is accepted for ABI stability and ignored. */
int kp1_mlp_loss_grad(kp1_mlp* m, const float* obs, int32_t obs_stride, const int64_t* idx, int32_t n, const float* actions,
                      const float* old_log_prob, const float* advantages, const float* returns, float adv_mean, float adv_inv_std,
                      const float* adv_stats_dev, float clip_range, float ent_coef, float vf_coef, float inv_count, float* grad_out,
                      float* stats_out, int32_t grad_is_zero, void* stream);

/* Options.  KP1_MLP_OPT_FUSED (default 1): for hidden = 256, kp1_mlp_loss_grad runs layer 1, layer 2, the heads, the PPO loss
 * and the activation backward of a 32-row minibatch tile in one workgroup (mlp_tile_kernel<true>) instead of four
 * chip-synchronous launches, and kp1_mlp_forward runs layer 1, layer 2, the heads and the Gaussian sampling in one launch
 * (mlp_tile_kernel<false>) instead of three; 0 selects the layer-wise kernels (always used for hidden = 128).  Same results up
 * to fp32 summation order of the per-tile partials. */
#define KP1_MLP_OPT_FUSED 1
/* KP1_MLP_OPT_ACTOR_EXTRA_STEPS (default 0): optimiser steps the actor tensors (mlp_extractor.policy_net.*, action_net.*) have taken
 * on top of the common count -- torch.optim.Adam counts per tensor, and the route trainer's teacher-anchor callback
 * (route/teacher_anchor.py:68-87) steps only the tensors its imitation loss reaches.  kp1_mlp_adam_step uses common + extra in
 * the bias corrections of those tensors. */
#define KP1_MLP_OPT_ACTOR_EXTRA_STEPS 2
/* KP1_MLP_OPT_STEP_COUNT: set the device-resident optimiser step count (PPO.load restores torch.optim.Adam's state, whose per-tensor
 * `step` feeds the bias corrections); kp1_mlp_loss_grad increments it, kp1_mlp_adam_step(step = 0) reads it. */
#define KP1_MLP_OPT_STEP_COUNT 3
/* KP1_MLP_OPT_PROFILE (default 0): attach a HIP-event pair to every kernel launch of the optimiser step (the dispatch's own begin and end),
 * on the launch stream and in the real launch sequence, so that a kernel's average duration is measured in situ (caches in the state the previous kernel of the
 * step left them) rather than in a back-to-back micro loop.  kp1_mlp_profile_read waits for the recorded launches and returns, per
 * slot, the average duration in microseconds and the launch count since the last read.  Not usable while a hipGraph is being captured. */
#define KP1_MLP_OPT_PROFILE 4
/* KP1_MLP_OPT_BF16X3_WGRAD (default 0) -- EXPERIMENT, not the measured product path: the weight-gradient GEMMs dW2 = dZ2^T h1, dW1 = dZ1^T X on
 * operands the tile kernel pre-splits into three bf16 pieces (x = hi + mid + lo, 24 bits) and six bf16 MFMAs per product block, fp32 accumulation
 * (gemm_tn_bf16x3_kernel) instead of the exact fp32 MFMA kernel.  Not bit-identical to the exact path: error table in DESIGN.md / bench.py's
 * `experiment_bf16x3_wgrad` block.  bench.py's `value` and `dtype` never use it. */
#define KP1_MLP_OPT_BF16X3_WGRAD 5
#define KP1_MLP_PROFILE_SLOTS 4
#define KP1_MLP_PROFILE_TILE 0      /* mlp_tile_kernel<true, .>: forward + loss + activation backward of both nets */
#define KP1_MLP_PROFILE_WGRAD 1     /* gemm_tn_frag_kernel: dW2 + dW1 of both nets */
#define KP1_MLP_PROFILE_FINALIZE 2  /* grad_finalize_kernel */
#define KP1_MLP_PROFILE_ADAM 3      /* (sum of squares +) adam_kernel */
int kp1_mlp_set_option(kp1_mlp* m, int32_t option, int32_t value);
int kp1_mlp_profile_read(kp1_mlp* m, float* out_us /* [KP1_MLP_PROFILE_SLOTS] */, int32_t* out_launches /* [KP1_MLP_PROFILE_SLOTS] */);

/* clip_grad_norm_(max_norm) + Adam(beta 0.9/0.999, eps) step on the flat vectors; the same pass repacks the kernel-format
 * weights.  step = 1-based Adam step count, or <= 0 to use the device-resident counter that every kp1_mlp_loss_grad call
 * increments (needed when the call sequence is replayed from a hipGraph).  flags bit 1 (value 2): grad is untouched since the kp1_mlp_loss_grad call that
 * produced it (no all-reduce in between), so the norm partials that call left are reused instead of a reduction launch. */
int kp1_mlp_adam_step(kp1_mlp* m, float* params, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float eps,
                      float max_grad_norm, int32_t step, int32_t flags, void* stream);

/* HIP-event timing of the MFMA kernels at minibatch size n (for bench.py's roofline block): runs each kernel `iters` times back
 * to back on `stream` between hipEventRecord pairs and returns the mean duration in milliseconds (out_ms / out_flops hold 6):
 *   out_ms[0] gemm_nt fwd layer 2 (H x H, bias+tanh)    out_ms[1] gemm_nt bwd dZ1 (H x H, dtanh)
 *   out_ms[2] gemm_tn dW2 (H x H, split over the batch)  out_ms[3] gemm_nt fwd layer 1 (64 -> H)
 *   out_ms[4] mlp_tile_kernel<true> (hidden 256: layer 1 + layer 2 + heads + loss + activation backward of both nets)
 *   out_ms[5] gemm_tn_frag_kernel (hidden 256: dW2 + dW1 of both nets)          [4], [5] = 0 for other widths
 * out_flops[k] = algorithmic FLOPs of one launch of kernel k (2*M*N*K summed over its GEMMs and both nets). */
/* Placement self-check of the two SPEED assumptions of the update kernels (never correctness): launches a probe with the training tile's launch
 * shape for an n_rows minibatch and reports where the hardware put the workgroups.  out[8] (host): [0] workgroups, [1] pairs (k, k + #CUs) probed,
 * [2] of them on the same CU (what the tile kernel's 12 us stagger assumes), [3] workgroups that run on the XCD of workgroup (linear index % 8)
 * and [7] distinct XCDs among workgroups 0..7 (what the weight-gradient kernel's chunk-major block order assumes: [3] = all, [7] = 8),
 * [4] #CUs, [5] distinct CUs used, [6] workgroups that arrived while the probe waited. */
int kp1_mlp_placement_check(int32_t device, int32_t n_rows, int32_t* out, void* stream);
int kp1_mlp_time_kernels(kp1_mlp* m, const float* obs, int32_t obs_stride, int32_t n, int32_t iters, float* out_ms, double* out_flops,
                         void* stream);

/* make kp1_step / kp1_reset write observation rows with this stride (56 default, 64 = MFMA-friendly, zero padded) */
int kp1_set_obs_stride(kp1_env* env, int32_t stride);

#ifdef __cplusplus
}
#endif
#endif /* KP1_PPO_H */
