// Pieces of the optimiser step shared by adam_step_kernel (optim.hip) and the fused finalise + Adam launch (mlp_dw.hip).
#pragma once

#include "rlg_device.hpp"

namespace rlg {

struct AdamArgs {
  float* params;
  float* grads;            // overwritten with the scaled/clipped gradient (like clip_grad_norm_)
  float* exp_avg;
  float* exp_avg_sq;
  long long n;
  const double* norm_partials;  // [norm_blocks] from grad_sumsq_kernel, or nullptr (no truncation)
  int norm_blocks;
  float grad_scale;        // 1/world_size (multi-GPU average), 1 otherwise
  float max_norm;          // grad_norm
  double* lr_slots;        // [2] fp64; slot (step-1)&1 is read, the other receives the next lr
  const long long* step_counter;  // device: Adam step count AFTER this update (1-based)
  double beta1, beta2, eps, weight_decay;
  // adaptive schedule (schedule_kind 1) driven by the KL of THIS minibatch
  int schedule_kind;       // 0 keep lr, 1 adaptive on *kl
  const float* kl;         // device scalar (already averaged over ranks)
  float kl_scale;          // 1/world_size when kl holds a cross-rank SUM
  double kl_threshold, min_lr, max_lr, lr_multiplier;
  float* stats_out;        // [4]: total_norm, clip_coef, lr used, lr next
  const unsigned* skip_flag;  // device word or nullptr; non-zero: the gradients are invalid (a failed in-graph
                              // all-reduce) - nothing is updated, the learning rate is carried over unchanged
#ifdef RLG_ADAM_TRACE
  unsigned long long* trace_rows;   // diagnostic builds only (tools/exp/build_trace_libs.sh): see adam_trace.hpp
  int trace_cap, trace_flags;
#endif
};

// torch.optim.Adam (single-tensor path) scalar prologue, evaluated in double like Python
struct AdamScalars {
  float step_size, bc2_sqrt, w1, b2, w2, eps, wd;
};
__device__ __forceinline__ AdamScalars adam_scalars(const AdamArgs& a, long long step, double lr) {
  AdamScalars k;
  const double bc1 = 1.0 - pow(a.beta1, static_cast<double>(step));
  const double bc2 = 1.0 - pow(a.beta2, static_cast<double>(step));
  k.step_size = static_cast<float>(lr / bc1);
  k.bc2_sqrt = static_cast<float>(sqrt(bc2));
  k.w1 = static_cast<float>(1.0 - a.beta1);
  k.b2 = static_cast<float>(a.beta2);
  k.w2 = static_cast<float>(1.0 - a.beta2);
  k.eps = static_cast<float>(a.eps);
  k.wd = static_cast<float>(a.weight_decay);
  return k;
}
// one parameter: g = (grad * grad_scale) * clip is written back (like clip_grad_norm_), then the Adam update
__device__ __forceinline__ void adam_update(const AdamArgs& a, const AdamScalars& k, long long i, float clip) {
  float g = (a.grads[i] * a.grad_scale) * clip;
  a.grads[i] = g;
  float p = a.params[i];
  if (k.wd != 0.0f) g = g + k.wd * p;                          // grad.add(param, alpha=wd)
  float m = a.exp_avg[i];
  m = m + k.w1 * (g - m);                                      // exp_avg.lerp_(grad, 1-beta1)
  float v = a.exp_avg_sq[i];
  v = v * k.b2 + (k.w2 * g) * g;                               // mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = sqrt_rn(v) / k.bc2_sqrt + k.eps;
  p = p - k.step_size * (m / denom);                           // addcdiv_(exp_avg, denom, -step_size)
  a.exp_avg[i] = m;
  a.exp_avg_sq[i] = v;
  a.params[i] = p;
}
// clip_coef = max_norm / (total_norm + 1e-6); clamp(max=1.0)       torch clip_grad_norm_
__device__ __forceinline__ float adam_clip_coef(float max_norm, float total_norm) {
  return fminf(max_norm / (total_norm + 1e-6f), 1.0f);
}
// AdaptiveScheduler.update in python-float arithmetic (schedulers.py:27-33) + the statistics row; one thread
__device__ __forceinline__ void adam_finish(const AdamArgs& a, int cur, double lr, bool skip, float total_norm, float clip) {
  double next = lr;
  if (a.schedule_kind == 1 && !skip) {
    const double kl = static_cast<double>(*a.kl * a.kl_scale);
    if (kl > 2.0 * a.kl_threshold) next = fmax(lr / a.lr_multiplier, a.min_lr);
    if (kl < 0.5 * a.kl_threshold) next = fmin(lr * a.lr_multiplier, a.max_lr);
  }
  a.lr_slots[cur ^ 1] = next;
  if (a.stats_out) {
    a.stats_out[0] = total_norm;
    a.stats_out[1] = clip;
    a.stats_out[2] = static_cast<float>(lr);
    a.stats_out[3] = static_cast<float>(next);
  }
}

}  // namespace rlg

#include "adam_trace.hpp"
