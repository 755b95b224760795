// Gradient truncation + Adam + KL-adaptive learning rate on a flat parameter arena, gfx950.
//
// Replaces A2CBase.trancate_gradients_and_step (rl_games/common/a2c_common.py:493-514):
//   * multi-GPU averaging of the all-reduced gradients ( / world_size, :505-507);
//   * torch.nn.utils.clip_grad_norm_(params, grad_norm) (:510-511);
//   * optim.Adam(..., eps=1e-8).step() (a2c_continuous.py:44-48, a2c_common.py:513);
// and the per-minibatch learning-rate control: AdaptiveScheduler.update
// (rl_games/common/schedulers.py:27-33) + update_lr (a2c_common.py:564-576, :1557-1563)
// without the `.item()` host sync: the learning rate lives in device memory (two fp64 slots,
// ping-pong by step parity) and is read lazily by the host; the Adam step counter is a device
// word too, so a (norm, Adam) launch pair has no per-call host arguments and can be replayed from a
// captured HIP graph.
//
// All parameters (and their gradients / Adam moments) are views of contiguous fp32 arenas, so
// one launch covers the whole model (~0.2 M parameters) instead of ~10 foreach kernels.
// Every block recomputes the (tiny) global-norm reduction from the per-block partial sums, so
// no grid barrier or atomic is needed and results are bit-reproducible.

#include "optim_common.hpp"

namespace rlg {

#ifdef RLG_ADAM_TRACE
unsigned long long* g_adam_trace_rows = nullptr;
int g_adam_trace_cap = 0, g_adam_trace_flags = 0;
#endif

constexpr int kOptBlock = 256;

// partial sum of squares of (grad * grad_scale), fp64, one value per block.  Also advances the
// device-resident optimiser step counter (read by adam_step_kernel, which always follows on the
// same stream): keeping the counter on the device makes the pair replayable from a HIP graph.
__global__ __launch_bounds__(kOptBlock) void grad_sumsq_kernel(const float* __restrict__ grads,
                                                               long long n, float grad_scale,
                                                               double* __restrict__ partials,
                                                               long long* __restrict__ step_counter) {
  __shared__ double scratch[kOptBlock / kWave];
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;
  double s[1] = {0.0};
  for (long long i = static_cast<long long>(blockIdx.x) * kOptBlock + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * kOptBlock) {
    const float g = grads[i] * grad_scale;
    s[0] = fma(static_cast<double>(g), static_cast<double>(g), s[0]);
  }
  block_sum<1, kOptBlock>(s, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = s[0];
}

// One thread = one aligned group of 4 parameters (the grid covers n / 4 groups + the < 4 elements behind them): all of a
// thread's loads are issued first and overlap with the gradient-norm reduction (two dependent memory round trips and two
// barriers) instead of following it - the launch is latency, not bandwidth (8 -> 5 us for 145 k parameters).
__global__ __launch_bounds__(kOptBlock) void adam_step_kernel(AdamArgs a) {
  __shared__ float sh_clip;
  __shared__ float sh_norm;
  __shared__ double scratch[kOptBlock / kWave];
  const long long n4 = a.n >> 2;
  const long long t = static_cast<long long>(blockIdx.x) * kOptBlock + threadIdx.x;
  const bool vec = t < n4;
  const long long tail = (n4 << 2) + (t - n4);                 // threads behind the groups take one tail element each
  const bool one = !vec && tail < a.n;
  f32x4 g4 = {0.0f, 0.0f, 0.0f, 0.0f}, p4 = g4, m4 = g4, v4 = g4;
  if (vec) {
    g4 = reinterpret_cast<const f32x4*>(a.grads)[t];
    p4 = reinterpret_cast<const f32x4*>(a.params)[t];
    m4 = reinterpret_cast<const f32x4*>(a.exp_avg)[t];
    v4 = reinterpret_cast<const f32x4*>(a.exp_avg_sq)[t];
  } else if (one) {
    g4[0] = a.grads[tail];
    p4[0] = a.params[tail];
    m4[0] = a.exp_avg[tail];
    v4[0] = a.exp_avg_sq[tail];
  }
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0u;
  const long long step = *a.step_counter;
  const int cur = static_cast<int>((step - 1) & 1);
  const double lr = a.lr_slots[cur];

  double sq[1] = {0.0};
  if (a.norm_partials) {
    // (a few thousand entries when the weight-gradient finalise launch produced them: 4 loads in flight per
    //  thread, fixed order - every block computes the same sum)
    int b = threadIdx.x;
    for (; b + 3 * kOptBlock < a.norm_blocks; b += 4 * kOptBlock) {
      const double v0 = a.norm_partials[b], v1 = a.norm_partials[b + kOptBlock];
      const double v2 = a.norm_partials[b + 2 * kOptBlock], v3 = a.norm_partials[b + 3 * kOptBlock];
      sq[0] += v0;
      sq[0] += v1;
      sq[0] += v2;
      sq[0] += v3;
    }
    for (; b < a.norm_blocks; b += kOptBlock) sq[0] += a.norm_partials[b];
    block_sum<1, kOptBlock>(sq, scratch);
  }
  if (threadIdx.x == 0) {
    float coef = 1.0f, total_norm = 0.0f;
    if (a.norm_partials) {
      const double s = sq[0];
      total_norm = static_cast<float>(sqrt(s));
      coef = adam_clip_coef(a.max_norm, total_norm);
    }
    sh_clip = coef;
    sh_norm = total_norm;
  }
  __syncthreads();
  const float clip = sh_clip;
  const AdamScalars k = adam_scalars(a, step, lr);
#ifdef RLG_ADAM_TRACE
  AdamTraceAcc tr;
  if ((vec || one) && !skip) {
    for (int e = 0; e < (vec ? 4 : 1); ++e) adam_trace_in(tr, a, step, (vec ? 4 * t : tail) + e, g4[e], p4[e], m4[e], v4[e]);
  }
#endif

  if ((vec || one) && !skip) {
    f32x4 gc;
    const int cnt = vec ? 4 : 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < cnt) {
        float g = (g4[e] * a.grad_scale) * clip;                     // (the same operations as adam_update)
        gc[e] = g;
        float p = p4[e];
        if (k.wd != 0.0f) g = g + k.wd * p;
        float m = m4[e];
        m = m + k.w1 * (g - m);
        float v = v4[e];
        v = v * k.b2 + (k.w2 * g) * g;
        const float denom = sqrt_rn(v) / k.bc2_sqrt + k.eps;
        p = p - k.step_size * (m / denom);
        m4[e] = m;
        v4[e] = v;
        p4[e] = p;
      }
    }
    if (vec) {
      reinterpret_cast<f32x4*>(a.grads)[t] = gc;
      reinterpret_cast<f32x4*>(a.exp_avg)[t] = m4;
      reinterpret_cast<f32x4*>(a.exp_avg_sq)[t] = v4;
      reinterpret_cast<f32x4*>(a.params)[t] = p4;
    } else {
      a.grads[tail] = gc[0];
      a.exp_avg[tail] = m4[0];
      a.exp_avg_sq[tail] = v4[0];
      a.params[tail] = p4[0];
    }
#ifdef RLG_ADAM_TRACE
    for (int e = 0; e < cnt; ++e) adam_trace_out(tr, (vec ? 4 * t : tail) + e, gc[e], p4[e], m4[e], v4[e]);
#endif
  }
#ifdef RLG_ADAM_TRACE
  adam_trace_flush(a, step, tr);
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_trace_scalars(a, step, clip, sh_norm, lr);
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_finish(a, cur, lr, skip, sh_norm, clip);
}

}  // namespace rlg

extern "C" {

#ifdef RLG_ADAM_TRACE
// diagnostic builds only: rows = device buffer of cap x 32 u64 (zeroed by the caller), flags bit 0 = coherent re-reads
int rlg_debug_adam_trace(unsigned long long* rows, int cap, int flags) {
  rlg::g_adam_trace_rows = rows;
  rlg::g_adam_trace_cap = cap;
  rlg::g_adam_trace_flags = flags;
  return 0;
}
#endif

int rlg_grad_norm_num_blocks(long long n) {
  long long b = (n + rlg::kOptBlock * 8 - 1) / (rlg::kOptBlock * 8);
  if (b < 1) b = 1;
  if (b > 256) b = 256;
  return static_cast<int>(b);
}

int rlg_grad_sumsq(const float* grads, long long n, float grad_scale, double* partials,
                   int num_blocks, long long* step_counter_or_null, void* stream) {
  if (n <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::grad_sumsq_kernel, dim3(num_blocks), dim3(rlg::kOptBlock), 0,
                     static_cast<hipStream_t>(stream), grads, n, grad_scale, partials,
                     step_counter_or_null);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                  const double* norm_partials_or_null, int norm_blocks, float grad_scale,
                  float max_norm, double* lr_slots, const long long* step_counter, double beta1,
                  double beta2, double eps, double weight_decay, int schedule_kind,
                  const float* kl_or_null, float kl_scale, double kl_threshold, double min_lr,
                  double max_lr, double lr_multiplier, float* stats_out_or_null,
                  const unsigned* skip_flag_or_null, void* stream) {
  using namespace rlg;
  if (n <= 0 || !step_counter) return static_cast<int>(hipErrorInvalidValue);
  if (schedule_kind == 1 && !kl_or_null) return static_cast<int>(hipErrorInvalidValue);
  AdamArgs a;
  a.params = params;
  a.grads = grads;
  a.exp_avg = exp_avg;
  a.exp_avg_sq = exp_avg_sq;
  a.n = n;
  a.norm_partials = norm_partials_or_null;
  a.norm_blocks = norm_blocks;
  a.grad_scale = grad_scale;
  a.max_norm = max_norm;
  a.lr_slots = lr_slots;
  a.step_counter = step_counter;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.schedule_kind = schedule_kind;
  a.kl = kl_or_null;
  a.kl_scale = kl_scale;
  a.kl_threshold = kl_threshold;
  a.min_lr = min_lr;
  a.max_lr = max_lr;
  a.lr_multiplier = lr_multiplier;
  a.stats_out = stats_out_or_null;
  a.skip_flag = skip_flag_or_null;
  RLG_ADAM_TRACE_FILL(a);
  // one thread per group of 4 parameters + one per tail element (the arenas are 16-byte aligned: torch allocations)
  if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  const long long threads = (n >> 2) + (n & 3);
  const long long grid = (threads + kOptBlock - 1) / kOptBlock;
  hipLaunchKernelGGL(adam_step_kernel, dim3(static_cast<int>(grid < 1 ? 1 : grid)), dim3(kOptBlock), 0,
                     static_cast<hipStream_t>(stream), a);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
