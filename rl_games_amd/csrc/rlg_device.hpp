// Shared device-side helpers for the rl_games_amd HIP kernels (gfx950 / CDNA4 only).
//
// Wavefronts are 64 lanes wide on CDNA4; every reduction below hard-codes that.
// The whole library is compiled with -ffp-contract=off so that every fp32
// multiply/add written in the kernels is rounded individually, exactly like the
// chain of eager PyTorch ops the reference executes; fused multiply-adds are only
// used where they are spelled out (fmaf / fma).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlg {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// the same value as a scalar (SGPR): loops and branches on it stay on the scalar unit
__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
  return x;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
  return x;
}

// Sum `K` doubles per thread across a block of `BLOCK` threads (BLOCK % 64 == 0).
// Result valid in thread 0 only.  `scratch` must hold K * BLOCK/64 doubles.
// The combination order is fixed (wave tree, then waves in index order), so the
// result is bit-reproducible run to run.
template <int K, int BLOCK>
__device__ __forceinline__ void block_sum(double (&v)[K], double* scratch) {
  constexpr int kWaves = BLOCK / kWave;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (kWaves == 1) return;
  if (lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[wave_id() * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = scratch[k];
      for (int w = 1; w < kWaves; ++w) s += scratch[w * K + k];
      v[k] = s;
    }
  }
}

// 16-byte vector helpers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Correctly rounded fp32 square root.  hipcc lowers sqrtf/__fsqrt_rn to the 1-ulp v_sqrt_f32;
// the reference's CPU ops are IEEE.  sqrt in fp64 then one rounding to fp32 is exact for
// sqrt (53 >= 2*24+2 bits), so this matches the CPU bit for bit.
__device__ __forceinline__ float sqrt_rn(float x) {
  return static_cast<float>(sqrt(static_cast<double>(x)));
}

// torch.clamp(x, lo, hi): min(max(x, lo), hi) that PROPAGATES NaN.  fmaxf / fminf (v_max_f32 / v_min_f32) return the
// non-NaN operand, i.e. a NaN observation / reward / ratio would silently become a bound of the clamp - a finite wrong
// value where the reference's op chain yields NaN (and a visibly broken run).
__device__ __forceinline__ float clamp_nan(float x, float lo, float hi) {
  const float c = fminf(fmaxf(x, lo), hi);
  return x != x ? x : c;
}

__device__ __forceinline__ bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// running_mean_std.py:55-67, operands promoted exactly as torch does: batch_mean / batch_var
// are fp32 tensors (input.mean / input.var of an fp32 input), the state is fp64.
__device__ __forceinline__ void chan_merge(double& mean, double& var, double count_f,
                                           double batch_mean, double batch_var, double batch_count) {
  const double tot = count_f + batch_count;
  const double delta = batch_mean - mean;
  const double new_mean = mean + delta * batch_count / tot;
  const double m_a = var * count_f;
  const double m_b = batch_var * batch_count;
  const double M2 = m_a + m_b + delta * delta * count_f * batch_count / tot;
  mean = new_mean;
  var = M2 / tot;
}


}  // namespace rlg

// Host-side launch check used by every extern "C" entry point: launchers never
// synchronise and never allocate; they return the hipError_t of the launch.
#define RLG_RETURN_LAUNCH_STATUS()          \
  do {                                      \
    hipError_t e__ = hipGetLastError();     \
    return static_cast<int>(e__);           \
  } while (0)
