// kp1_ppo.hip -- PPO-side device kernels (include/kp1_ppo.h): GAE scan, time-limit bootstrap,
// device-resident curriculum tracker.  The actor-critic MLP kernels live in kp1_mlp.hip.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/kp1_ppo.h"
#include "kp1_host.hpp"

using kp1::fail;

namespace {

// One lane per env; every [t] row access is a fully coalesced wave instruction (4 B/lane f32, 1 B/lane done).
// Algorithmic traffic: read r, V, done (9 B) + write A, R (8 B) per sample.
__global__ void __launch_bounds__(256) gae_scan_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                       const uint8_t* __restrict__ dones, const float* __restrict__ last_values,
                                                       float gamma, float lam, float* __restrict__ adv, float* __restrict__ ret, int T, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float next_value = last_values[i];
  float last_gae = 0.0f;
  // prefetch-friendly reverse walk; T is small (<= a few thousand)
  for (int t = T - 1; t >= 0; --t) {
    const int64_t k = (int64_t)t * N + i;
    const float nonterm = (dones[k] & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) ? 0.0f : 1.0f;
    const float v = values[k];
    const float delta = rewards[k] + gamma * next_value * nonterm - v;
    last_gae = delta + gamma * lam * nonterm * last_gae;
    adv[k] = last_gae;
    ret[k] = last_gae + v;
    next_value = v;
  }
}

__global__ void __launch_bounds__(256) bootstrap_kernel(float* __restrict__ rewards, const float* __restrict__ tv,
                                                        const uint8_t* __restrict__ dones, float gamma, int64_t count) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint8_t d = dones[k];
  if ((d & KP1_DONE_TRUNCATED) && !(d & KP1_DONE_TERMINATED)) rewards[k] += gamma * tv[k];
}

// PointCurriculumCallback._on_step; callbacks.py:71-92.  One wave: 64-env chunks are skipped with a ballot when no
// episode ended (the common case: 95 of 96 steps); finished episodes are replayed in env order by lane 0.
__global__ void __launch_bounds__(64) curriculum_kernel(kp1_curriculum_state* __restrict__ st, const uint8_t* __restrict__ dones, int n,
                                                        int steps_per_call) {
  const int lane = threadIdx.x;
  if (lane == 0) st->num_timesteps += steps_per_call;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint8_t d = i < n ? dones[i] : 0;
    const bool done = (d & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) != 0;
    const unsigned long long done_mask = __ballot(done);
    if (done_mask == 0ull) continue;
    const unsigned long long succ_mask = __ballot(done && (d & KP1_DONE_SUCCESS));
    if (lane == 0) {
      unsigned long long m = done_mask;
      int stage = st->stage_index, count = st->stage_episode_count, len = st->ring_len, head = st->ring_head;
      const int window = st->window_episodes;
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int success = (int)((succ_mask >> b) & 1ull);
        count += 1;
        if (len < window) {
          st->ring[(head + len) % window] = success;
          len += 1;
        } else {
          st->ring[head] = success;
          head = (head + 1) % window;
        }
        if (stage >= st->max_stage_index) continue;
        if (count < st->min_episodes_per_stage) continue;
        if (len < window) continue;
        int s = 0;
        for (int k = 0; k < len; ++k) s += st->ring[k];
        const double rate = (double)s / (double)len;
        if (rate >= st->success_rate_threshold) {
          if (st->n_events < KP1_CURRICULUM_MAX_HISTORY) {
            kp1_curriculum_event& ev = st->events[st->n_events];
            ev.total_timesteps = st->num_timesteps;
            ev.from_stage = stage;
            ev.to_stage = stage + 1;
            ev.trigger_success_rate = rate;
          }
          st->n_events += 1;
          stage += 1;
          count = 0;
          len = 0;
          head = 0;
        }
      }
      st->stage_index = stage;
      st->stage_episode_count = count;
      st->ring_len = len;
      st->ring_head = head;
    }
  }
}

int check_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(KP1_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= count) return fail(KP1_ERR_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  return KP1_OK;
}

}  // namespace

extern "C" {

int kp1_gae_scan(int32_t device, const float* rewards, const float* values, const uint8_t* dones, const float* last_values, float gamma,
                 float gae_lambda, float* advantages, float* returns, int32_t T, int32_t N, void* stream) {
  if (!rewards || !values || !dones || !last_values || !advantages || !returns || T <= 0 || N <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_gae_scan");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  const int block = N <= 16384 ? 64 : 256;
  hipLaunchKernelGGL(gae_scan_kernel, dim3((N + block - 1) / block), dim3(block), 0, (hipStream_t)stream, rewards, values, dones, last_values,
                     gamma, gae_lambda, advantages, returns, T, N);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

int kp1_bootstrap_truncated(int32_t device, float* rewards, const float* terminal_values, const uint8_t* dones, float gamma, int64_t count,
                            void* stream) {
  if (!rewards || !terminal_values || !dones || count <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_bootstrap_truncated");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(bootstrap_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, terminal_values, dones,
                     gamma, count);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}

int kp1_curriculum_create(int32_t device, double success_rate_threshold, int32_t window_episodes, int32_t min_episodes_per_stage,
                          int32_t max_stage_index, int32_t initial_stage_index, kp1_curriculum_state** out_dev) {
  if (!out_dev) return fail(KP1_ERR_INVALID, "out_dev is NULL");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  kp1_curriculum_state h;
  std::memset(&h, 0, sizeof h);
  h.success_rate_threshold = success_rate_threshold;
  h.window_episodes = window_episodes < 1 ? 1 : window_episodes;  // callbacks.py:45-47 max(..., 1)
  if (h.window_episodes > KP1_CURRICULUM_MAX_WINDOW) return fail(KP1_ERR_INVALID, "window_episodes exceeds KP1_CURRICULUM_MAX_WINDOW");
  h.min_episodes_per_stage = min_episodes_per_stage < 1 ? 1 : min_episodes_per_stage;
  h.max_stage_index = max_stage_index < 0 ? 0 : max_stage_index;
  int init = initial_stage_index < h.max_stage_index ? initial_stage_index : h.max_stage_index;
  h.stage_index = init < 0 ? 0 : init;  // callbacks.py:48
  kp1_curriculum_state* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, sizeof h));
  HIP_TRY(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
  *out_dev = d;
  return KP1_OK;
}
int kp1_curriculum_destroy(int32_t device, kp1_curriculum_state* st_dev) {
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  HIP_TRY(hipFree(st_dev));
  return KP1_OK;
}
int kp1_curriculum_observe(int32_t device, kp1_curriculum_state* st_dev, const uint8_t* dones, int32_t n, int32_t steps_per_call, void* stream) {
  if (!st_dev || !dones || n <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_curriculum_observe");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(curriculum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st_dev, dones, n, steps_per_call);
  HIP_TRY(hipGetLastError());
  return KP1_OK;
}
int kp1_curriculum_read(int32_t device, const kp1_curriculum_state* st_dev, kp1_curriculum_state* out_host, void* stream) {
  if (!st_dev || !out_host) return fail(KP1_ERR_INVALID, "NULL argument");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  HIP_TRY(hipMemcpyAsync(out_host, st_dev, sizeof *out_host, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return KP1_OK;
}

}  // extern "C"
