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

// PointCurriculumCallback._on_step; callbacks.py:71-92.  One wave.  Each lane loads 64 consecutive done bytes (4 x 16 B)
// and condenses them into a 64-bit done mask and a success mask, so 4096 envs cost one round of loads; a ballot skips
// the common case where no episode ended.  Finished episodes are then replayed strictly in env order by lane 0 (the masks
// of lane k are fetched with a shuffle), which is what the reference's sequential scan does.
//
// [r2] The tracker's state lives in registers and its success ring in LDS for the whole launch (TrackerCtx): round 2's profile had this
// one-wave kernel at 13.2 us per env step (3 % of the PPO iteration) because lane 0 replayed ~45 finished episodes per step with every
// field of *st re-read from global memory after every ring store (the compiler must assume the ring aliases them) -- a chain of
// dependent memory round trips -- and, below the last stage, re-summed the whole ring per episode.  The window sum is now carried along
// (sum += new - overwritten: the same integer the reference's sum(window) gives), so an episode costs a few LDS / scalar instructions.
struct TrackerCtx {
  int stage, count, len, head, window, min_episodes, max_stage, n_events, sum;
  double threshold;
  long long timesteps;
  bool ring_dirty;
};

__device__ __forceinline__ void tracker_load(const kp1_curriculum_state* __restrict__ st, int* __restrict__ ring, TrackerCtx& c) {
  const int lane = threadIdx.x;
  c.stage = st->stage_index; c.count = st->stage_episode_count; c.len = st->ring_len; c.head = st->ring_head;
  c.window = st->window_episodes; c.min_episodes = st->min_episodes_per_stage; c.max_stage = st->max_stage_index; c.n_events = st->n_events;
  c.threshold = st->success_rate_threshold; c.timesteps = st->num_timesteps;
  c.ring_dirty = false;
  // live entries: the len positions from head on (while the ring is filling head is 0; once full every entry is live)
  int part = 0;
  for (int k = lane; k < c.window; k += 64) {
    const int v = st->ring[k];
    ring[k] = v;
    const int age = k >= c.head ? k - c.head : k - c.head + c.window;
    if (age < c.len) part += v;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  c.sum = part;
  __syncthreads();
}

__device__ __forceinline__ void tracker_store(kp1_curriculum_state* __restrict__ st, const int* __restrict__ ring, const TrackerCtx& c) {
  const int lane = threadIdx.x;
  __syncthreads();
  if (c.ring_dirty)
    for (int k = lane; k < c.window; k += 64) st->ring[k] = ring[k];
  if (lane == 0) {
    st->stage_index = c.stage; st->stage_episode_count = c.count; st->ring_len = c.len; st->ring_head = c.head;
    st->n_events = c.n_events; st->num_timesteps = c.timesteps;
  }
}

// [r3] Many episodes in one block of 4096 envs (every env of a handle truncates in the same step when the episodes started together: 4096
// finished episodes at once, which lane 0 replayed in ~600 us -- 1.4 % of the benchmark's iteration hidden in a "4.7 us" kernel).  Same result,
// computed by the whole wave: the success bits in episode order go to LDS, a prefix count over (window contents ++ new bits) gives the window
// sum after EVERY episode at once, the first episode that satisfies the promotion rule is a wave-min, and the ring / length / head / sum after
// appending a run of episodes follow in closed form (entry j of a run lands in slot (head + len + j) mod window, full or not).  A promotion
// empties the window and the search continues behind it.  Wave-uniform in, wave-uniform out; LDS scratch: 4096 bits-as-bytes + prefix counts.
#ifndef KP1_TRK_PARALLEL_MIN
#define KP1_TRK_PARALLEL_MIN 192          // finished episodes per 4096-env block from which the whole-wave path is taken (A/B: 1 << 30 = never)
#endif
constexpr int TRK_BLOCK = 64 * 64, TRK_PARALLEL_MIN = KP1_TRK_PARALLEL_MIN;
struct TrackerScratch { uint8_t* sbit; uint16_t* pfx; };

// append episodes [pos, pos + R) of sbit to the window; no promotion check
__device__ __forceinline__ void tracker_append_run(int* __restrict__ ring, TrackerCtx& c, const uint8_t* __restrict__ sbit, int pos, int R) {
  const int lane = threadIdx.x, W = c.window;
  const int j0 = R > W ? R - W : 0;                       // earlier entries of the run are overwritten by later ones
  __syncthreads();
  for (int j = j0 + lane; j < R; j += 64) ring[(c.head + c.len + j) % W] = sbit[pos + j];
  __syncthreads();
  const int over = c.len + R - W;
  c.head = over > 0 ? (c.head + over) % W : c.head;
  c.len = c.len + R < W ? c.len + R : W;
  c.count += R;
  int part = 0;
  for (int k = lane; k < W; k += 64) {
    const int age = k >= c.head ? k - c.head : k - c.head + W;
    if (age < c.len) part += ring[k];
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  c.sum = part;
}

__device__ __forceinline__ void tracker_block_parallel(kp1_curriculum_state* __restrict__ st, int* __restrict__ ring, TrackerCtx& c, const TrackerScratch& ws,
                                                       unsigned long long dmask, unsigned long long smask, int E) {
  const int lane = threadIdx.x, W = c.window;
  // success bits in episode (= env) order
  int mine = __popcll(dmask), off = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(off, d);
    if (lane >= d) off += up;
  }
  off -= mine;                                            // exclusive prefix: index of this lane's first finished episode
  __syncthreads();
  for (unsigned long long m = dmask; m; m &= m - 1) {
    const int b = __ffsll((long long)m) - 1;
    ws.sbit[off++] = (uint8_t)((smask >> b) & 1ull);
  }
  __syncthreads();
  int pos = 0;
  while (pos < E) {
    const int R = E - pos;
    if (c.stage >= c.max_stage) {                          // last stage: the window only records
      tracker_append_run(ring, c, ws.sbit, pos, R);
      break;
    }
    // prefix counts Q[0 .. L + R] over Y = (window contents, oldest first) ++ sbit[pos ..): Q[j] = successes among the first j entries of Y
    const int L = c.len, T = L + R;
    const int seg = (T + 63) / 64, j_lo = lane * seg < T ? lane * seg : T, j_hi = j_lo + seg < T ? j_lo + seg : T;
    auto y = [&](int j) -> int { return j < L ? ring[(c.head + j) % W] : (int)ws.sbit[pos + (j - L)]; };
    int tot = 0;
    for (int j = j_lo; j < j_hi; ++j) tot += y(j);
    int run = tot;
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(run, d);
      if (lane >= d) run += up;
    }
    run -= tot;
    __syncthreads();
    if (lane == 0) ws.pfx[0] = 0;
    for (int j = j_lo; j < j_hi; ++j) {
      run += y(j);
      ws.pfx[j + 1] = (uint16_t)run;
    }
    __syncthreads();
    // first episode r of the run after whose append the rule holds: count + r + 1 >= min_episodes, window full, rate >= threshold
    int best = R;
    for (int r = lane; r < R; r += 64) {
      const int filled = L + r + 1;
      if (filled < W || c.count + r + 1 < c.min_episodes) continue;
      const int wsum = (int)ws.pfx[filled] - (int)ws.pfx[filled - W];
      if ((double)wsum / (double)W >= c.threshold) {
        best = r;
        break;                                            // r only grows along this lane's stride
      }
    }
    for (int d = 32; d > 0; d >>= 1) {
      const int o = __shfl_xor(best, d);
      best = o < best ? o : best;
    }
    if (best >= R) {                                       // no promotion inside this run
      tracker_append_run(ring, c, ws.sbit, pos, R);
      break;
    }
    const int filled = L + best + 1;
    const double rate = (double)((int)ws.pfx[filled] - (int)ws.pfx[filled - W]) / (double)W;
    if (lane == 0 && c.n_events < KP1_CURRICULUM_MAX_HISTORY) {
      kp1_curriculum_event& ev = st->events[c.n_events];
      ev.total_timesteps = c.timesteps;
      ev.from_stage = c.stage;
      ev.to_stage = c.stage + 1;
      ev.trigger_success_rate = rate;
    }
    c.n_events += 1;
    c.stage += 1;
    c.count = 0;
    c.len = 0;
    c.head = 0;
    c.sum = 0;
    pos += best + 1;
  }
}

// consume dones[0 .. n) in index order.  c is wave-uniform on entry and on exit (lane 0's values are broadcast at the end).
__device__ __forceinline__ void curriculum_scan(kp1_curriculum_state* __restrict__ st, int* __restrict__ ring, TrackerCtx& c, const TrackerScratch& ws,
                                                const uint8_t* __restrict__ dones, int n, int steps_per_call) {
  const int lane = threadIdx.x;
  c.timesteps += steps_per_call;
  for (int base = 0; base < n; base += 64 * 64) {
    const int first = base + lane * 64;
    unsigned long long dmask = 0ull, smask = 0ull;
    if (first + 64 <= n && (reinterpret_cast<uintptr_t>(dones + first) & 15) == 0) {
      const uint4* p = reinterpret_cast<const uint4*>(dones + first);
      uint4 w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = p[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int words[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const unsigned int d = (words[k] >> (8 * b)) & 0xffu;
            const int bit = q * 16 + k * 4 + b;
            if (d & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) {
              dmask |= 1ull << bit;
              if (d & KP1_DONE_SUCCESS) smask |= 1ull << bit;
            }
          }
      }
    } else {
      for (int b = 0; b < 64 && first + b < n; ++b) {
        const uint8_t d = dones[first + b];
        if (d & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) {
          dmask |= 1ull << b;
          if (d & KP1_DONE_SUCCESS) smask |= 1ull << b;
        }
      }
    }
    unsigned long long busy = __ballot(dmask != 0ull);
    if (busy == 0ull) continue;
    c.ring_dirty = true;
    {
      int ended = __popcll(dmask);
      for (int off = 32; off > 0; off >>= 1) ended += __shfl_xor(ended, off);
      if (ended >= TRK_PARALLEL_MIN) {                     // wave-uniform
        tracker_block_parallel(st, ring, c, ws, dmask, smask, ended);
        continue;
      }
    }
    int stage = c.stage, count = c.count, len = c.len, head = c.head, sum = c.sum, n_events = c.n_events;
    const int window = c.window;
    while (busy) {   // lanes that saw a finished episode, in lane (= env) order
      const int src = __ffsll((long long)busy) - 1;
      busy &= busy - 1;
      unsigned long long m = __shfl(dmask, src);
      const unsigned long long sm = __shfl(smask, src);
      if (lane != 0) continue;
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int success = (int)((sm >> b) & 1ull);
        count += 1;
        if (len < window) {
          ring[head + len < window ? head + len : head + len - window] = success;
          len += 1;
          sum += success;
        } else {
          sum += success - ring[head];
          ring[head] = success;
          head = head + 1 < window ? head + 1 : 0;
        }
        if (stage >= c.max_stage) continue;
        if (count < c.min_episodes) continue;
        if (len < window) continue;
        const double rate = (double)sum / (double)len;
        if (rate >= c.threshold) {
          if (n_events < KP1_CURRICULUM_MAX_HISTORY) {
            kp1_curriculum_event& ev = st->events[n_events];
            ev.total_timesteps = c.timesteps;
            ev.from_stage = stage;
            ev.to_stage = stage + 1;
            ev.trigger_success_rate = rate;
          }
          n_events += 1;
          stage += 1;
          count = 0;
          len = 0;
          head = 0;
          sum = 0;
        }
      }
    }
    c.stage = __shfl(stage, 0); c.count = __shfl(count, 0); c.len = __shfl(len, 0); c.head = __shfl(head, 0);
    c.sum = __shfl(sum, 0); c.n_events = __shfl(n_events, 0);
  }
}

__global__ void __launch_bounds__(64) curriculum_kernel(kp1_curriculum_state* __restrict__ st, const uint8_t* __restrict__ dones, int n,
                                                        int steps_per_call) {
  __shared__ int ring[KP1_CURRICULUM_MAX_WINDOW];
  __shared__ uint8_t sbit[TRK_BLOCK];
  __shared__ uint16_t pfx[TRK_BLOCK + KP1_CURRICULUM_MAX_WINDOW + 1];
  const TrackerScratch ws = {sbit, pfx};
  // [r3] the common VecEnv step ends no episode: then the only state that changes is num_timesteps, and the kernel is ONE round of loads (the
  // done bytes) + a fire-and-forget add instead of three dependent round trips (state -> window -> done bytes) + the state stores
#ifndef KP1_TRK_FAST_EMPTY
#define KP1_TRK_FAST_EMPTY 1
#endif
  if (KP1_TRK_FAST_EMPTY) {
    const int lane = threadIdx.x;
    bool any = false;
    for (int base = 0; base < n; base += 64 * 64) {
      const int first = base + lane * 64;
      if (first + 64 <= n && (reinterpret_cast<uintptr_t>(dones + first) & 15) == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(dones + first);
        unsigned int acc = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 w = p[q];
          acc |= w.x | w.y | w.z | w.w;
        }
        any |= (acc & (0x01010101u * (unsigned)(KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED))) != 0u;
      } else {
        for (int b = 0; b < 64 && first + b < n; ++b) any |= (dones[first + b] & (KP1_DONE_TERMINATED | KP1_DONE_TRUNCATED)) != 0;
      }
    }
    if (__ballot(any) == 0ull) {
      if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&st->num_timesteps), (unsigned long long)(long long)steps_per_call);
      return;
    }
  }
  TrackerCtx c;
  tracker_load(st, ring, c);
  curriculum_scan(st, ring, c, ws, dones, n, steps_per_call);
  tracker_store(st, ring, c);
}

// Data-parallel rollouts exchange the done bytes once per CHUNK of env steps instead of once per step: `dones` is the all-gathered
// [world][chunk_steps][n_local] block (rank-major, as all_gather_into_tensor lays it out).  The tracker replays it in the order the
// reference callback would have seen a single VecEnv of world * n_local envs: step by step, and inside a step rank by rank = global
// env id order (callbacks.py:78-91).  One wave; the state stays in registers / LDS across the whole chunk.
__global__ void __launch_bounds__(64) curriculum_chunk_kernel(kp1_curriculum_state* __restrict__ st, const uint8_t* __restrict__ dones, int n_local,
                                                              int chunk_steps, int world, int steps_per_env_step) {
  __shared__ int ring[KP1_CURRICULUM_MAX_WINDOW];
  __shared__ uint8_t sbit[TRK_BLOCK];
  __shared__ uint16_t pfx[TRK_BLOCK + KP1_CURRICULUM_MAX_WINDOW + 1];
  const TrackerScratch ws = {sbit, pfx};
  TrackerCtx c;
  tracker_load(st, ring, c);
  for (int t = 0; t < chunk_steps; ++t)
    for (int r = 0; r < world; ++r)
      curriculum_scan(st, ring, c, ws, dones + ((int64_t)r * chunk_steps + t) * n_local, n_local, r == 0 ? steps_per_env_step : 0);
  tracker_store(st, ring, c);
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
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

}  // extern "C"

namespace {
// (sum, sum of squares) of the advantages of every minibatch of an epoch in one launch: block b covers
// idx[b * mb .. min((b + 1) * mb, total)).  fp64, fixed summation order => deterministic.
__global__ void __launch_bounds__(256) adv_minibatch_sums_kernel(const float* __restrict__ adv, const int64_t* __restrict__ idx, int64_t total, int64_t mb,
                                                                 double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  const int64_t begin = (int64_t)blockIdx.x * mb, end = begin + mb < total ? begin + mb : total;
  double a = 0.0, b = 0.0;
  for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
    const double v = (double)adv[idx ? idx[i] : i];
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      s1[threadIdx.x] += s1[threadIdx.x + k];
      s2[threadIdx.x] += s2[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = s1[0];
    out[3 * blockIdx.x + 1] = s2[0];
    out[3 * blockIdx.x + 2] = (double)(end - begin);
  }
}
// keyed bijection on [0, 2^bits): every round is invertible modulo 2^bits (odd multiplier; x ^= x >> s with s >= 1), so the composition is
// a permutation; multiplication carries low bits upward, the xorshift carries high bits downward.  Cycle walking restricts it to [0, n).
struct PermKeys { uint32_t mul[4], add[4]; };
__global__ void __launch_bounds__(256) permutation_kernel(int64_t* __restrict__ out, int64_t n, int bits, const PermKeys k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
  const int sh = bits > 1 ? bits / 2 : 1;
  uint32_t x = (uint32_t)i;
  do {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x = (x * k.mul[r] + k.add[r]) & mask;
      x ^= x >> sh;
    }
  } while ((int64_t)x >= n);
  out[i] = (int64_t)x;
}

// (sum, sum of squares, count) -> (mean, 1 / (unbiased std + 1e-8)); f64 arithmetic in the order of the torch expressions it replaces
__global__ void __launch_bounds__(64) adv_minibatch_stats_kernel(const double* __restrict__ sums, int64_t n_mb, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= n_mb) return;
  const double s0 = sums[3 * b], s1 = sums[3 * b + 1], cnt = sums[3 * b + 2];
  const double mean = s0 / cnt;
  const double dof = cnt - 1.0 < 1.0 ? 1.0 : cnt - 1.0;
  double var = (s1 - cnt * mean * mean) / dof;
  var = var < 0.0 ? 0.0 : var;
  out[2 * b] = (float)mean;
  out[2 * b + 1] = (float)(1.0 / (sqrt(var) + 1e-8));
}
}  // namespace

extern "C" {

int kp1_random_permutation(int32_t device, int64_t n, const uint32_t* keys, int64_t* out, void* stream) {
  if (!keys || !out || n <= 0 || n > ((int64_t)1 << 31)) return fail(KP1_ERR_INVALID, "bad argument to kp1_random_permutation");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  int bits = 1;
  while (((int64_t)1 << bits) < n) ++bits;
  PermKeys k;
  for (int r = 0; r < 4; ++r) {
    k.mul[r] = keys[r] | 1u;     // odd: invertible modulo 2^bits
    k.add[r] = keys[4 + r];
  }
  hipLaunchKernelGGL(permutation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, bits, k);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_adv_minibatch_stats(int32_t device, const double* sums, int64_t n_minibatches, float* out_stats, void* stream) {
  if (!sums || !out_stats || n_minibatches <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_adv_minibatch_stats");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(adv_minibatch_stats_kernel, dim3((unsigned)((n_minibatches + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sums, n_minibatches,
                     out_stats);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_adv_minibatch_sums(int32_t device, const float* advantages, const int64_t* idx, int64_t total, int64_t minibatch, double* out_sums,
                           void* stream) {
  if (!advantages || !out_sums || total <= 0 || minibatch <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_adv_minibatch_sums");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  const int64_t n_mb = (total + minibatch - 1) / minibatch;
  hipLaunchKernelGGL(adv_minibatch_sums_kernel, dim3((unsigned)n_mb), dim3(256), 0, (hipStream_t)stream, advantages, idx, total, minibatch, out_sums);
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}

int kp1_bootstrap_truncated(int32_t device, float* rewards, const float* terminal_values, const uint8_t* dones, float gamma, int64_t count,
                            void* stream) {
  if (!rewards || !terminal_values || !dones || count <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_bootstrap_truncated");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(bootstrap_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, terminal_values, dones,
                     gamma, count);
  HIP_TRY(kp1::launch_status());
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
  HIP_TRY(kp1::launch_status());
  return KP1_OK;
}
int kp1_curriculum_observe_chunk(int32_t device, kp1_curriculum_state* st_dev, const uint8_t* dones, int32_t n_local, int32_t chunk_steps,
                                 int32_t world, void* stream) {
  if (!st_dev || !dones || n_local <= 0 || chunk_steps <= 0 || world <= 0) return fail(KP1_ERR_INVALID, "bad argument to kp1_curriculum_observe_chunk");
  int rc = check_device(device);
  if (rc != KP1_OK) return rc;
  hipLaunchKernelGGL(curriculum_chunk_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st_dev, dones, n_local, chunk_steps, world, n_local * world);
  HIP_TRY(kp1::launch_status());
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
