// Diagnostic instrumentation of the optimiser launches (adam_step_kernel, adam_pack_kernel, adam_frags_kernel), compiled in
// ONLY with -DRLG_ADAM_TRACE (tools/exp/build_trace_libs.sh builds such libraries next to the product one; the product
// library contains none of this).  Every optimiser step leaves one row of 32 u64 words in a device buffer:
//   [0] step   [1] clip bits | total-norm bits << 32   [2] lr (fp64 bits)   [3] kl bits | grad_scale bits << 32
//   [4..7]   position-weighted checksums  sum(bits(x[i]) * (2 i + 1)) mod 2^64  of grads, params, exp_avg, exp_avg_sq AS LOADED
//   [8..11]  the same of the four arrays AS STORED (clipped gradient, new parameters, new moments)
//   [12..15] (flag bit 0) the same of the four inputs read a second time with system-scope atomic loads (cache-bypassing)
//   [16] number of elements whose plain load and coherent load differ, [17..30] the first 7 of them as
//        (index | array << 48, plain bits | coherent bits << 32)
// Checksums are sums of per-element terms (atomicAdd, order-independent), so two ranks - or one rank's row s "as stored" and
// row s + 1 "as loaded" - can be compared word by word: which quantity of which step differs first names the mechanism
// (a different reduced gradient, a different clip coefficient, a stale read, different arithmetic).
#pragma once

#ifdef RLG_ADAM_TRACE
namespace rlg {

constexpr int kAdamTraceWords = 32;
extern unsigned long long* g_adam_trace_rows;
extern int g_adam_trace_cap, g_adam_trace_flags;

struct AdamTraceAcc {
  unsigned long long s[12];
  __device__ __forceinline__ AdamTraceAcc() {
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0ull;
  }
};

__device__ __forceinline__ unsigned long long adam_trace_term(float x, long long i) {
  return static_cast<unsigned long long>(__float_as_uint(x)) * (2ull * static_cast<unsigned long long>(i) + 1ull);
}
__device__ __forceinline__ unsigned long long* adam_trace_row(const AdamArgs& a, long long step) {
  if (a.trace_rows == nullptr || a.trace_cap <= 0) return nullptr;
  return a.trace_rows + static_cast<long long>((step - 1) % a.trace_cap) * kAdamTraceWords;
}
// one element as loaded (+ the coherent re-read)
__device__ __forceinline__ void adam_trace_in(AdamTraceAcc& acc, const AdamArgs& a, long long step, long long i, float g,
                                              float p, float m, float v) {
  unsigned long long* row = adam_trace_row(a, step);
  if (row == nullptr) return;
  acc.s[0] += adam_trace_term(g, i);
  acc.s[1] += adam_trace_term(p, i);
  acc.s[2] += adam_trace_term(m, i);
  acc.s[3] += adam_trace_term(v, i);
  if (a.trace_flags & 1) {
    const float* arr[4] = {a.grads, a.params, a.exp_avg, a.exp_avg_sq};
    const float plain[4] = {g, p, m, v};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned c = __hip_atomic_load(reinterpret_cast<const unsigned*>(arr[k]) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      acc.s[8 + k] += static_cast<unsigned long long>(c) * (2ull * static_cast<unsigned long long>(i) + 1ull);
      if (c != __float_as_uint(plain[k])) {
        const unsigned long long slot = atomicAdd(row + 16, 1ull);
        if (slot < 7ull) {
          row[17 + 2 * slot] = static_cast<unsigned long long>(i) | (static_cast<unsigned long long>(k) << 48);
          row[18 + 2 * slot] = static_cast<unsigned long long>(__float_as_uint(plain[k])) | (static_cast<unsigned long long>(c) << 32);
        }
      }
    }
  }
}
__device__ __forceinline__ void adam_trace_out(AdamTraceAcc& acc, long long i, float g, float p, float m, float v) {
  acc.s[4] += adam_trace_term(g, i);
  acc.s[5] += adam_trace_term(p, i);
  acc.s[6] += adam_trace_term(m, i);
  acc.s[7] += adam_trace_term(v, i);
}
// EVERY thread of the block, once, in uniform control flow (wave shuffles)
__device__ __forceinline__ void adam_trace_flush(const AdamArgs& a, long long step, const AdamTraceAcc& acc) {
  unsigned long long* row = adam_trace_row(a, step);
  if (row == nullptr) return;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    unsigned long long x = acc.s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    if (lane_id() == 0 && x != 0ull) atomicAdd(row + 4 + k, x);
  }
}
__device__ __forceinline__ void adam_trace_scalars(const AdamArgs& a, long long step, float clip, float norm, double lr) {
  unsigned long long* row = adam_trace_row(a, step);
  if (row == nullptr) return;
  row[0] = static_cast<unsigned long long>(step);
  row[1] = static_cast<unsigned long long>(__float_as_uint(clip)) | (static_cast<unsigned long long>(__float_as_uint(norm)) << 32);
  row[2] = static_cast<unsigned long long>(__double_as_longlong(lr));
  const float kl = a.kl != nullptr ? *a.kl : 0.0f;
  row[3] = static_cast<unsigned long long>(__float_as_uint(kl)) | (static_cast<unsigned long long>(__float_as_uint(a.grad_scale)) << 32);
}

// the flat-range elements (adam_update): loads, traces, updates, traces what it stored
__device__ __forceinline__ void adam_update_traced(AdamTraceAcc& acc, const AdamArgs& a, const AdamScalars& k, long long step,
                                                   long long i, float clip) {
  adam_trace_in(acc, a, step, i, a.grads[i], a.params[i], a.exp_avg[i], a.exp_avg_sq[i]);
  adam_update(a, k, i, clip);
  adam_trace_out(acc, i, a.grads[i], a.params[i], a.exp_avg[i], a.exp_avg_sq[i]);
}

}  // namespace rlg
#define RLG_ADAM_TRACE_FILL(a)                \
  do {                                        \
    (a).trace_rows = rlg::g_adam_trace_rows;  \
    (a).trace_cap = rlg::g_adam_trace_cap;    \
    (a).trace_flags = rlg::g_adam_trace_flags; \
  } while (0)
#else
#define RLG_ADAM_TRACE_FILL(a) \
  do {                         \
  } while (0)
#endif
