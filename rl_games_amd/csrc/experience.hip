// Rollout-buffer writes and per-step bookkeeping for gfx950 (MI355X).
//
// Replaces, per environment step n of A2CBase.play_steps (rl_games/common/a2c_common.py:
// 985-1069):
//   * ExperienceBuffer.update_data x7-8 (rl_games/common/experience.py:433-456; call sites
//     a2c_common.py:1000-1011)                      -> rollout_store_kernel   (one launch)
//   * reward shaping (rl_games/common/tr_helpers.py:33-42), time-out bootstrap
//     (a2c_common.py:1021-1023), update_data('rewards') (:1025), episode accumulators and
//     done handling (:1027-1051)                    -> rollout_post_step_kernel (one launch)
//   * AverageMeter.update x3 per step (rl_games/algos_torch/torch_ext.py:333-342), replayed
//     once per rollout from per-step partial sums   -> episode_meters_kernel
//
// Storage is ENV-MAJOR: a field with per-env row of `row` elements lives as [N][H][row], so
// the flat index of (env, t) is env*H + t - exactly the order swap_and_flatten01
// (a2c_common.py:33-40) produces.  The reference's [H, N, ...] tensors are exposed to Python
// as strided views of this storage and the epoch-end transpose copy disappears.
//
// All kernels are pure data movement / element-wise fp32 (one rounding per op, same op order
// as the reference), HBM-bound: the store kernel moves (O + 3A + 3)*4 + 1 bytes per env-step
// in and the same out.

#include "rlg_device.hpp"

namespace rlg {

constexpr int kMaxSegments = 12;

// One "field" of the step: src is [N][row_bytes] contiguous, dst is env-major storage.
struct StoreSegments {
  const void* src[kMaxSegments];
  void* dst[kMaxSegments];
  int row_bytes[kMaxSegments];   // bytes per env row
  int unit[kMaxSegments];        // copy granularity: 16, 4 or 1 bytes
  int block_begin[kMaxSegments + 1];
  int count;
};

// kStream: non-temporal stores - the observation rows (28 MB per step at 65,536 x 108) are not read
// again before the update phase, so they should not displace the rewards / values / done flags of
// the rollout (the GAE kernel's inputs) from the 256 MB Infinity Cache.
template <typename T, bool kStream>
__device__ __forceinline__ void copy_units(const void* __restrict__ src, void* __restrict__ dst,
                                           long long units_total, int units_per_row,
                                           long long dst_row_stride_units, long long dst_off_units,
                                           long long first, int stride) {
  const T* s = static_cast<const T*>(src);
  T* d = static_cast<T*>(dst);
  for (long long i = first; i < units_total; i += stride) {
    const long long env = i / units_per_row;
    const long long c = i - env * units_per_row;
    if (kStream) __builtin_nontemporal_store(s[i], d + env * dst_row_stride_units + dst_off_units + c);
    else d[env * dst_row_stride_units + dst_off_units + c] = s[i];
  }
}

constexpr int kStoreBlock = 256;
constexpr int kStoreUnitsPerThread = 4;

__global__ __launch_bounds__(kStoreBlock) void rollout_store_kernel(StoreSegments seg, int N, int H,
                                                                    int step, int stream_wide_rows) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < kMaxSegments; ++k) {
    if (k < seg.count && static_cast<int>(blockIdx.x) >= seg.block_begin[k]) s = k;
  }
  const int local_block = blockIdx.x - seg.block_begin[s];
  const int nblocks = seg.block_begin[s + 1] - seg.block_begin[s];
  const int unit = seg.unit[s];
  const int upr = seg.row_bytes[s] / unit;                 // units per env row
  const long long total = static_cast<long long>(N) * upr;
  const long long first = static_cast<long long>(local_block) * kStoreBlock + threadIdx.x;
  const int stride = nblocks * kStoreBlock;
  const long long row_stride = static_cast<long long>(H) * upr;
  const long long off = static_cast<long long>(step) * upr;
  if (unit == 16) {
    if (stream_wide_rows && seg.row_bytes[s] >= 128)
      copy_units<u32x4, true>(seg.src[s], seg.dst[s], total, upr, row_stride, off, first, stride);
    else
      copy_units<u32x4, false>(seg.src[s], seg.dst[s], total, upr, row_stride, off, first, stride);
  } else if (unit == 4) {
    copy_units<uint32_t, false>(seg.src[s], seg.dst[s], total, upr, row_stride, off, first, stride);
  } else {
    copy_units<uint8_t, false>(seg.src[s], seg.dst[s], total, upr, row_stride, off, first, stride);
  }
}

// ---------------------------------------------------------------------------------
// Post-step: rewards, accumulators, finished-episode partial sums
// ---------------------------------------------------------------------------------

struct PostStepArgs {
  const float* rewards;        // [N, V] raw env rewards of this step
  const uint8_t* dones;        // [N] done flags produced by this step
  const void* time_outs;       // [N] or nullptr   (infos['time_outs'])
  int time_outs_kind;          // 0 none, 1 uint8/bool, 2 float32
  const float* values;         // [N, V] critic values of this step (res_dict['values'])
  const float* live_rows;      // [N] or nullptr: 1 - prev_dones (next_step autoreset mask)
  float* rewards_buf;          // env-major [N][H][V]
  float* cur_rewards;          // [N, V]
  float* cur_shaped;           // [N, V]
  float* cur_lengths;          // [N]
  double* ep_partials;         // [H][nblocks][2V+2]: sum rew[V], sum shaped[V], sum len, count
  float shift, scale, rmin, rmax;
  int clamp_rewards;           // bit 0: clamp to [rmin, rmax] (0: the bounds are -inf/+inf); bit 1: log_val (log of the shaped reward)
  int bootstrap;               // value_bootstrap and 'time_outs' in infos
  float gamma;
  int N, H, V, step;
  int num_agents;              // meters take one row per env: rows with row % num_agents == 0
};

constexpr int kPostBlock = 256;
constexpr int kMaxV = 8;

__global__ __launch_bounds__(kPostBlock) void rollout_post_step_kernel(PostStepArgs a) {
  __shared__ double scratch[(2 * kMaxV + 2) * (kPostBlock / kWave)];
  const int env = blockIdx.x * kPostBlock + threadIdx.x;
  const int V = a.V;
  double acc[2 * kMaxV + 2];
#pragma unroll
  for (int k = 0; k < 2 * kMaxV + 2; ++k) acc[k] = 0.0;

  if (env < a.N) {
    const float done = static_cast<float>(a.dones[env]);
    const bool is_done = a.dones[env] != 0;
    float to = 0.0f;
    if (a.bootstrap) {
      to = (a.time_outs_kind == 2) ? static_cast<const float*>(a.time_outs)[env]
                                   : static_cast<float>(static_cast<const uint8_t*>(a.time_outs)[env]);
    }
    const float live = a.live_rows ? a.live_rows[env] : 1.0f;
    const float alive = 1.0f - done;
    // game_rewards / game_shaped_rewards / game_lengths are fed from all_done_indices[::num_agents]
    // (a2c_common.py:1040-1044): the first agent row of every finished env (the agents of one env
    // finish together, which is also what the reference's stride over the done list assumes).
    const bool metered = is_done && (a.num_agents <= 1 || env % a.num_agents == 0);
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      if (k < V) {
        const float rew = a.rewards[env * V + k];
        // DefaultRewardsShaper.__call__: (r + shift) * scale, clamp           tr_helpers.py:35-39
        float shaped = (rew + a.shift) * a.scale;
        if (a.clamp_rewards & 1) shaped = clamp_nan(shaped, a.rmin, a.rmax);
        if (a.clamp_rewards & 2) shaped = logf(shaped);                      // log_val       tr_helpers.py:40-41
        // shaped += gamma * values * time_outs                               a2c_common.py:1022-1023
        if (a.bootstrap) shaped = shaped + (a.gamma * a.values[env * V + k]) * to;
        a.rewards_buf[(static_cast<long long>(env) * a.H + a.step) * V + k] = shaped;   // :1025
        // episode accumulators                                               :1027-1037
        float cr, cs;
        if (a.live_rows) {
          cr = a.cur_rewards[env * V + k] + rew * live;
          cs = a.cur_shaped[env * V + k] + shaped * live;
        } else {
          cr = a.cur_rewards[env * V + k] + rew;
          cs = a.cur_shaped[env * V + k] + shaped;
        }
        if (metered) {
          acc[k] = cr;
          acc[kMaxV + k] = cs;
        }
        a.cur_rewards[env * V + k] = cr * alive;                                        // :1049
        a.cur_shaped[env * V + k] = cs * alive;                                         // :1050
      }
    }
    const float cl = a.cur_lengths[env] + (a.live_rows ? live : 1.0f);
    if (metered) {
      acc[2 * kMaxV] = cl;
      acc[2 * kMaxV + 1] = 1.0;
    }
    a.cur_lengths[env] = cl * alive;                                                    // :1051
  }
  block_sum<2 * kMaxV + 2, kPostBlock>(acc, scratch);
  if (threadIdx.x == 0) {
    double* out = a.ep_partials + (static_cast<long long>(a.step) * gridDim.x + blockIdx.x) * (2 * V + 2);
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      if (k < V) {
        out[k] = acc[k];
        out[V + k] = acc[kMaxV + k];
      }
    }
    out[2 * V] = acc[2 * kMaxV];
    out[2 * V + 1] = acc[2 * kMaxV + 1];
  }
}

// Replays AverageMeter.update for steps 0..H-1 (game_rewards, game_shaped_rewards,
// game_lengths) from the per-step partial sums.  One block: per step a parallel sum over the
// post-step blocks, then thread 0 applies the meter recurrence.
//   size = clip(count, 0, max); old = min(max - size, cur); mean = (mean*old + new*size)/(old+size)
__global__ __launch_bounds__(256) void episode_meters_kernel(
    const double* __restrict__ ep_partials, int H, int nblocks, int V, int max_size,
    float* __restrict__ mean_rewards, float* __restrict__ mean_shaped,
    float* __restrict__ mean_lengths, int* __restrict__ current_sizes /* [3] */,
    long long* __restrict__ finished_total) {
  __shared__ double scratch[(2 * kMaxV + 2) * 4];
  const int W = 2 * V + 2;
  for (int t = 0; t < H; ++t) {
    double s[2 * kMaxV + 2];
#pragma unroll
    for (int k = 0; k < 2 * kMaxV + 2; ++k) s[k] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
      const double* p = ep_partials + (static_cast<long long>(t) * nblocks + b) * W;
#pragma unroll
      for (int k = 0; k < 2 * kMaxV + 2; ++k) {
        if (k < W) s[k] += p[k];
      }
    }
    block_sum<2 * kMaxV + 2, 256>(s, scratch);
    if (threadIdx.x == 0) {
      double sv[2 * kMaxV + 2];
#pragma unroll
      for (int k = 0; k < 2 * kMaxV + 2; ++k) sv[k] = s[k];
      double cnt_d = 0.0, len_d = 0.0;
#pragma unroll
      for (int k = 0; k < 2 * kMaxV + 2; ++k) {
        if (k == 2 * V + 1) cnt_d = sv[k];
        if (k == 2 * V) len_d = sv[k];
      }
      const long long count = static_cast<long long>(cnt_d);
      if (count != 0) {                                           // torch_ext.py:334-336
        *finished_total += count;
        const int size = static_cast<int>(count < max_size ? count : max_size);
        for (int m = 0; m < 3; ++m) {
          const int old_size = min(max_size - size, current_sizes[m]);
          const int size_sum = old_size + size;
          current_sizes[m] = size_sum;
          if (m < 2) {
            float* mean = (m == 0) ? mean_rewards : mean_shaped;
#pragma unroll
            for (int k = 0; k < kMaxV; ++k) {
              if (k < V) {
                double tot = 0.0;
#pragma unroll
                for (int q = 0; q < 2 * kMaxV; ++q) {
                  if (q == m * V + k) tot = sv[q];
                }
                const float new_mean = static_cast<float>(tot / static_cast<double>(count));
                mean[k] = (mean[k] * static_cast<float>(old_size) + new_mean * static_cast<float>(size)) /
                          static_cast<float>(size_sum);
              }
            }
          } else {
            const float new_mean = static_cast<float>(len_d / static_cast<double>(count));
            mean_lengths[0] = (mean_lengths[0] * static_cast<float>(old_size) +
                               new_mean * static_cast<float>(size)) /
                              static_cast<float>(size_sum);
          }
        }
      }
    }
    __syncthreads();   // scratch is reused by the next step's block_sum
  }
}

// ---------------------------------------------------------------------------------
// Policy head of a rollout step (is_train = False branch of ModelA2CContinuousLogStd.forward,
// rl_games/algos_torch/models.py:348-359 + neglogp :361-364 + denorm_value :58-60), fused with
// the buffer writes of its outputs (update_data for actions/mus/sigmas/neglogpacs/values,
// a2c_common.py:1008-1009):
//   sigma = exp(logstd);  action = mu + sigma * noise   (= Normal(mu, sigma).sample())
//   neglogp = 0.5*sum(((action-mu)/sigma)^2) + 0.5*log(2pi)*A + sum(logstd)
//   value   = sqrt(var+eps)*clamp(v,-5,5) + mean        (running_mean_std.py:106-107)
// heads: [N, 1+A] fused (value | mu) output of the MLP.  One thread per env.
struct PolicyHeadArgs {
  const float* heads;      // [N, ld]
  int ld;
  const float* logstd;     // [A]
  const float* noise;      // [N, A] standard normal
  const double* v_mean;    // value RunningMeanStd (fp64) or nullptr when normalize_value is off
  const double* v_var;
  float eps;
  float* actions_out;      // [N, A] contiguous (goes to the env)
  float* values_out;       // [N]    de-normalised values (time-out bootstrap needs them)
  float* buf_actions;      // env-major [N][H][A]
  float* buf_mus;
  float* buf_sigmas;
  float* buf_neglogp;      // [N][H]
  float* buf_values;       // [N][H]
  // optional: what goes to the env when clip_actions is set (a2c_common.py:725-733 preprocess_actions:
  // rescale_actions(low, high, clamp(actions, -1, 1)), torch_ext.py rescale_actions)
  float* env_actions_out;  // [N, A] or nullptr
  const float* act_low;    // [A]
  const float* act_high;   // [A]
  int N, H, A, step;
};

// Two phases per 256-env block.  A: one thread per env reduces sum z^2 / sum logstd over the
// actions and writes the per-env scalars.  B: the block's 256*A (env, action) elements are walked
// with consecutive threads on consecutive addresses, so the four [.., A] outputs are written as
// contiguous 4*A-byte runs instead of one scalar per thread per iteration at a 4*A*H-byte stride
// (90 -> ~25 us at 65,536 x 21).  Phase B recomputes act = mu + sigma*noise with the same
// expression, so both phases see bit-identical values.
__global__ __launch_bounds__(256) void rollout_policy_head_kernel(PolicyHeadArgs p) {
  const int env0 = blockIdx.x * blockDim.x;
  const int env = env0 + threadIdx.x;
  if (env < p.N) {
    const float* h = p.heads + static_cast<long long>(env) * p.ld;
    const float* nz = p.noise + static_cast<long long>(env) * p.A;
    const long long slot = static_cast<long long>(env) * p.H + p.step;
    float s_z2 = 0.0f, s_ls = 0.0f;
    for (int a = 0; a < p.A; ++a) {
      const float mu = h[1 + a];
      const float ls = p.logstd[a];
      const float sg = expf(ls);
      const float act = mu + sg * nz[a];
      const float z = (act - mu) / sg;
      s_z2 += z * z;
      s_ls += ls;
    }
    const float nlp = (0.5f * s_z2 + static_cast<float>(0.9189385332046727 * p.A)) + s_ls;
    p.buf_neglogp[slot] = nlp;
    float v = h[0];
    if (p.v_mean) {
      const float m = static_cast<float>(p.v_mean[0]);
      const float d = sqrt_rn(static_cast<float>(p.v_var[0]) + p.eps);
      v = d * clamp_nan(v, -5.0f, 5.0f) + m;
    }
    p.values_out[env] = v;
    p.buf_values[slot] = v;
  }
  const int rows = min(static_cast<int>(blockDim.x), p.N - env0);
  const int total = rows * p.A;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int el = idx / p.A;
    const int a = idx - el * p.A;
    const long long e = env0 + el;
    const float mu = p.heads[e * p.ld + 1 + a];
    const float sg = expf(p.logstd[a]);
    const float act = mu + sg * p.noise[e * p.A + a];
    const long long o = (e * p.H + p.step) * p.A + a;
    p.actions_out[e * p.A + a] = act;
    if (p.env_actions_out) {
      // d = (high - low) / 2, m = (high + low) / 2, clamp(act, -1, 1) * d + m - each op rounded like the
      // reference's separate torch kernels
      const float lo = p.act_low[a], hi = p.act_high[a];
      const float d = (hi - lo) / 2.0f, m = (hi + lo) / 2.0f;
      p.env_actions_out[e * p.A + a] = clamp_nan(act, -1.0f, 1.0f) * d + m;
    }
    p.buf_actions[o] = act;
    p.buf_mus[o] = mu;
    p.buf_sigmas[o] = sg;
  }
}

// The same two phases with the block's heads and noise tiles staged in LDS: both are read from global memory ONCE, with
// consecutive threads on consecutive addresses (phase A's one-thread-per-env walk over a 4*ld-byte row stride, and phase
// B's second pass over both arrays, then run from LDS; row strides ld and A words - odd for the usual 1 + A heads - spread
// the rows over the banks).  Same expressions in the same order: bit-identical outputs.  lds: kB * (ld + A) floats.
template <int kB>
__global__ __launch_bounds__(kB) void rollout_policy_head_lds_kernel(PolicyHeadArgs p) {
  extern __shared__ float head_lds[];
  const int env0 = blockIdx.x * kB;
  const int rows = min(kB, p.N - env0);
  float* const sh = head_lds;                  // [rows][ld]
  float* const sn = head_lds + kB * p.ld;      // [rows][A]
  const float* gh = p.heads + static_cast<long long>(env0) * p.ld;
  const float* gn = p.noise + static_cast<long long>(env0) * p.A;
  for (int i = threadIdx.x; i < rows * p.ld; i += kB) sh[i] = gh[i];
  for (int i = threadIdx.x; i < rows * p.A; i += kB) sn[i] = gn[i];
  __syncthreads();
  const int env = env0 + threadIdx.x;
  if (env < p.N) {
    const float* h = sh + threadIdx.x * p.ld;
    const float* nz = sn + threadIdx.x * p.A;
    const long long slot = static_cast<long long>(env) * p.H + p.step;
    float s_z2 = 0.0f, s_ls = 0.0f;
    for (int a = 0; a < p.A; ++a) {
      const float mu = h[1 + a];
      const float ls = p.logstd[a];
      const float sg = expf(ls);
      const float act = mu + sg * nz[a];
      const float z = (act - mu) / sg;
      s_z2 += z * z;
      s_ls += ls;
    }
    const float nlp = (0.5f * s_z2 + static_cast<float>(0.9189385332046727 * p.A)) + s_ls;
    p.buf_neglogp[slot] = nlp;
    float v = h[0];
    if (p.v_mean) {
      const float m = static_cast<float>(p.v_mean[0]);
      const float d = sqrt_rn(static_cast<float>(p.v_var[0]) + p.eps);
      v = d * clamp_nan(v, -5.0f, 5.0f) + m;
    }
    p.values_out[env] = v;
    p.buf_values[slot] = v;
  }
  const int total = rows * p.A;
  for (int idx = threadIdx.x; idx < total; idx += kB) {
    const int el = idx / p.A;
    const int a = idx - el * p.A;
    const long long e = env0 + el;
    const float mu = sh[el * p.ld + 1 + a];
    const float sg = expf(p.logstd[a]);
    const float act = mu + sg * sn[idx];
    const long long o = (e * p.H + p.step) * p.A + a;
    p.actions_out[e * p.A + a] = act;
    if (p.env_actions_out) {
      const float lo = p.act_low[a], hi = p.act_high[a];
      const float d = (hi - lo) / 2.0f, m = (hi + lo) / 2.0f;
      p.env_actions_out[e * p.A + a] = clamp_nan(act, -1.0f, 1.0f) * d + m;
    }
    p.buf_actions[o] = act;
    p.buf_mus[o] = mu;
    p.buf_sigmas[o] = sg;
  }
}

// RNN rollout helpers (a2c_common.py:1081-1083 snapshot, :1150-1153 zero-on-done).
// states: [L, N, U] contiguous.  snapshot dst: [num_seqs, L, N, U] slice `seq`.
__global__ __launch_bounds__(256) void rnn_zero_done_kernel(float* __restrict__ states,
                                                            const uint8_t* __restrict__ dones,
                                                            int L, int N, int U) {
  const long long total = static_cast<long long>(L) * N * U;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long e = (i / U) % N;
    if (dones[e]) states[i] = 0.0f;
  }
}

}  // namespace rlg

extern "C" {

// 1 (default): rows of >= 128 bytes (the observations) are written with non-temporal stores.
static int g_stream_wide_rows = 1;
int rlg_rollout_store_streaming(int enable) {
  const int prev = g_stream_wide_rows;
  if (enable >= 0) g_stream_wide_rows = enable ? 1 : 0;
  return prev;
}

int rlg_rollout_store_step(int count, const void* const* srcs, void* const* dsts,
                           const int* row_bytes, int num_envs, int horizon, int step,
                           void* stream) {
  using namespace rlg;
  if (count <= 0 || num_envs <= 0) return 0;
  if (count > kMaxSegments || step < 0 || step >= horizon) return static_cast<int>(hipErrorInvalidValue);
  StoreSegments seg;
  seg.count = count;
  int blocks = 0;
  for (int k = 0; k < count; ++k) {
    seg.src[k] = srcs[k];
    seg.dst[k] = dsts[k];
    seg.row_bytes[k] = row_bytes[k];
    const uintptr_t a = reinterpret_cast<uintptr_t>(srcs[k]) | reinterpret_cast<uintptr_t>(dsts[k]) |
                        static_cast<uintptr_t>(row_bytes[k]);
    seg.unit[k] = (a % 16 == 0) ? 16 : ((a % 4 == 0) ? 4 : 1);
    const long long units = static_cast<long long>(num_envs) * (row_bytes[k] / seg.unit[k]);
    long long nb = (units + static_cast<long long>(kStoreBlock) * kStoreUnitsPerThread - 1) /
                   (static_cast<long long>(kStoreBlock) * kStoreUnitsPerThread);
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    seg.block_begin[k] = blocks;
    blocks += static_cast<int>(nb);
  }
  for (int k = count; k <= kMaxSegments; ++k) seg.block_begin[k] = blocks;
  hipLaunchKernelGGL(rollout_store_kernel, dim3(blocks), dim3(kStoreBlock), 0,
                     static_cast<hipStream_t>(stream), seg, num_envs, horizon, step, g_stream_wide_rows);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_rollout_post_step_num_blocks(int num_envs) {
  return (num_envs + rlg::kPostBlock - 1) / rlg::kPostBlock;
}

int rlg_rollout_post_step(const float* rewards, const uint8_t* dones, const void* time_outs,
                          int time_outs_kind, const float* values, const float* live_rows,
                          float* rewards_buf, float* cur_rewards, float* cur_shaped,
                          float* cur_lengths, double* ep_partials, float shift, float scale,
                          float rmin, float rmax, int clamp_rewards, int bootstrap, float gamma,
                          int num_envs, int horizon, int value_size, int step, int num_agents,
                          void* stream) {
  using namespace rlg;
  if (num_envs <= 0) return 0;
  if (value_size < 1 || value_size > kMaxV || step < 0 || step >= horizon)
    return static_cast<int>(hipErrorInvalidValue);
  PostStepArgs a;
  a.rewards = rewards;
  a.dones = dones;
  a.time_outs = time_outs;
  a.time_outs_kind = time_outs ? time_outs_kind : 0;
  a.values = values;
  a.live_rows = live_rows;
  a.rewards_buf = rewards_buf;
  a.cur_rewards = cur_rewards;
  a.cur_shaped = cur_shaped;
  a.cur_lengths = cur_lengths;
  a.ep_partials = ep_partials;
  a.shift = shift;
  a.scale = scale;
  a.rmin = rmin;
  a.rmax = rmax;
  a.clamp_rewards = clamp_rewards;
  a.bootstrap = (bootstrap && time_outs && a.time_outs_kind != 0) ? 1 : 0;
  a.gamma = gamma;
  a.N = num_envs;
  a.H = horizon;
  a.V = value_size;
  a.step = step;
  a.num_agents = num_agents < 1 ? 1 : num_agents;
  const int grid = rlg_rollout_post_step_num_blocks(num_envs);
  hipLaunchKernelGGL(rollout_post_step_kernel, dim3(grid), dim3(kPostBlock), 0,
                     static_cast<hipStream_t>(stream), a);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_episode_meters_update(const double* ep_partials, int horizon, int num_blocks,
                              int value_size, int max_size, float* mean_rewards,
                              float* mean_shaped, float* mean_lengths, int* current_sizes,
                              long long* finished_total, void* stream) {
  if (value_size < 1 || value_size > rlg::kMaxV) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::episode_meters_kernel, dim3(1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ep_partials, horizon, num_blocks, value_size,
                     max_size, mean_rewards, mean_shaped, mean_lengths, current_sizes,
                     finished_total);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_rollout_policy_head(const float* heads, int ld_heads, const float* logstd, const float* noise,
                            const double* value_mean_or_null, const double* value_var_or_null,
                            float eps, float* actions_out, float* values_out, float* buf_actions,
                            float* buf_mus, float* buf_sigmas, float* buf_neglogp, float* buf_values,
                            float* env_actions_out, const float* act_low, const float* act_high,
                            int num_envs, int horizon, int actions_num, int step, void* stream) {
  if (num_envs <= 0) return 0;
  if (step < 0 || step >= horizon || actions_num <= 0) return static_cast<int>(hipErrorInvalidValue);
  if (env_actions_out && (!act_low || !act_high)) return static_cast<int>(hipErrorInvalidValue);
  rlg::PolicyHeadArgs p;
  p.heads = heads;
  p.ld = ld_heads;
  p.logstd = logstd;
  p.noise = noise;
  p.v_mean = value_mean_or_null;
  p.v_var = value_var_or_null;
  p.eps = eps;
  p.actions_out = actions_out;
  p.values_out = values_out;
  p.buf_actions = buf_actions;
  p.buf_mus = buf_mus;
  p.buf_sigmas = buf_sigmas;
  p.buf_neglogp = buf_neglogp;
  p.buf_values = buf_values;
  p.N = num_envs;
  p.H = horizon;
  p.A = actions_num;
  p.step = step;
  p.env_actions_out = env_actions_out;
  p.act_low = act_low;
  p.act_high = act_high;
  // one wave per workgroup: 8,192 envs (a rank of 8) are 128 workgroups instead of 32 - the kernel is a chain of dependent
  // memory round trips per workgroup, so it wants every CU (16 -> 8 us there); the layout of the work inside a
  // workgroup (phase A one thread per env, phase B consecutive threads on consecutive addresses) is unchanged
  constexpr int kHeadBlock = 64;
  // the LDS-staged form where a block's tiles fit 32 KiB (any realistic head width); RLG_ROLLOUT_HEAD_LDS=0: the direct form
  static const bool staged = [] {
    const char* e = getenv("RLG_ROLLOUT_HEAD_LDS");
    return !(e && e[0] == '0');
  }();
  const size_t lds = static_cast<size_t>(kHeadBlock) * (static_cast<size_t>(ld_heads) + actions_num) * sizeof(float);
  if (staged && ld_heads >= 1 + actions_num && lds <= 32 * 1024) {
    hipLaunchKernelGGL(rlg::rollout_policy_head_lds_kernel<kHeadBlock>, dim3((num_envs + kHeadBlock - 1) / kHeadBlock),
                       dim3(kHeadBlock), lds, static_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(rlg::rollout_policy_head_kernel, dim3((num_envs + kHeadBlock - 1) / kHeadBlock), dim3(kHeadBlock), 0,
                       static_cast<hipStream_t>(stream), p);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_rnn_zero_done_states(float* states, const uint8_t* dones, int layers, int num_envs,
                             int units, void* stream) {
  const long long total = static_cast<long long>(layers) * num_envs * units;
  if (total <= 0) return 0;
  int grid = static_cast<int>((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rlg::rnn_zero_done_kernel, dim3(grid), dim3(256), 0,
                     static_cast<hipStream_t>(stream), states, dones, layers, num_envs, units);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
