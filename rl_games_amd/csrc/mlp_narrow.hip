// Two products of the policy network that are too narrow for the MFMA kernels (and used to fall back to the library
// GEMM): the input gradient through a head with very few outputs, and the weight gradient of a first layer over very
// few observations - BASELINE config #5 (Pendulum-shaped: obs 3, act 1) has both.
//
//   narrow_dx   : dX[r][j]  = sum_k dZ[r][k] W[k][j],  K <= 8      autograd's grad_output.mm(weight) of the fused
//                 (value | mu) head, rl_games/algos_torch/network_builder.py:295-311, :506-512
//   narrow_dw   : dW[o][i] = sum_r dZ[r][o] X[r][i],  Mi <= 8     grad_output.t().mm(input) of actor_mlp's first
//                 nn.Linear, network_builder.py:118-147
// Both are HBM-bound streaming passes (no reuse to exploit): coalesced 16-byte accesses, fp32 products and fp32 sums
// inside a thread's run of rows, fp64 partial sums across workgroups combined in a fixed order (deterministic).

#include "rlg_device.hpp"
#include "rlg_hip.h"

namespace rlg {

constexpr int kNarrowMax = 8;

// one thread = one row x 4 consecutive output columns
__global__ __launch_bounds__(256) void narrow_dx_kernel(const float* __restrict__ dz, long long lddz,
                                                        const float* __restrict__ w, float* __restrict__ dx, long long lddx,
                                                        long long rows, int K, int M) {
  const int m4 = (M + 3) >> 2;
  const long long total = rows * m4;
  for (long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; t < total; t += static_cast<long long>(gridDim.x) * 256) {
    const long long r = t / m4;
    const int j = static_cast<int>(t - r * m4) * 4;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < K; ++k) {
      const float d = dz[r * lddz + k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j + e < M) acc[e] = acc[e] + d * w[k * M + j + e];     // k ascending, one rounding per op (like a k-loop GEMM)
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (j + e < M) dx[r * lddx + j + e] = acc[e];
    }
  }
}

// workgroup b sums rows [b * chunk, (b + 1) * chunk): thread (o, s) - s = one of 256 / No row sub-sequences.
// Latency bound (a few dozen rows per thread): four rows' loads are in flight at a time.
__global__ __launch_bounds__(256) void narrow_dw_partial_kernel(const float* __restrict__ dz, long long lddz,
                                                                const float* __restrict__ x, long long ldx,
                                                                double* __restrict__ partials, long long rows, int chunk,
                                                                int No, int Mi) {
  __shared__ float sh[256 * kNarrowMax];
  const int subs = 256 / No;
  const int o = threadIdx.x % No, s = threadIdx.x / No;
  float acc[kNarrowMax];
#pragma unroll
  for (int i = 0; i < kNarrowMax; ++i) acc[i] = 0.0f;
  const long long r0 = static_cast<long long>(blockIdx.x) * chunk;
  const long long r1 = r0 + chunk < rows ? r0 + chunk : rows;
  if (s < subs) {
    long long r = r0 + s;
    for (; r + 3LL * subs < r1; r += 4LL * subs) {
      float d[4], xv[4][kNarrowMax];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = dz[(r + static_cast<long long>(u) * subs) * lddz + o];
#pragma unroll
        for (int i = 0; i < kNarrowMax; ++i) xv[u][i] = i < Mi ? x[(r + static_cast<long long>(u) * subs) * ldx + i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < kNarrowMax; ++i) {
          if (i < Mi) acc[i] = acc[i] + d[u] * xv[u][i];
        }
      }
    }
    for (; r < r1; r += subs) {
      const float d = dz[r * lddz + o];
#pragma unroll
      for (int i = 0; i < kNarrowMax; ++i) {
        if (i < Mi) acc[i] = acc[i] + d * x[r * ldx + i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kNarrowMax; ++i) sh[i * 256 + threadIdx.x] = acc[i];
  __syncthreads();
  if (s == 0) {
    for (int i = 0; i < Mi; ++i) {
      double t = 0.0;
      for (int q = 0; q < subs; ++q) t += static_cast<double>(sh[i * 256 + q * No + o]);
      partials[(static_cast<long long>(blockIdx.x) * No + o) * Mi + i] = t;
    }
  }
}

// one wave per gradient element: lane l adds the partials of workgroups l, l + 64, ..., then a fixed shuffle tree
__global__ __launch_bounds__(256) void narrow_dw_finalize_kernel(const double* __restrict__ partials, int blocks, int n,
                                                                 float* __restrict__ grad) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n) return;
  double t = 0.0;
  for (int b = lane_id(); b < blocks; b += kWave) t += partials[static_cast<long long>(b) * n + e];
  t = wave_sum(t);
  if (lane_id() == 0) grad[e] = static_cast<float>(t);
}

}  // namespace rlg

extern "C" {

int rlg_narrow_dx(const float* dz, long long lddz, const float* w, float* dx, long long lddx, long long rows,
                  int out_features, int in_features, void* stream) {
  if (rows <= 0) return 0;
  if (out_features < 1 || out_features > rlg::kNarrowMax || in_features < 1) return static_cast<int>(hipErrorInvalidValue);
  const long long total = rows * ((in_features + 3) / 4);
  long long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rlg::narrow_dx_kernel, dim3(static_cast<int>(grid)), dim3(256), 0, static_cast<hipStream_t>(stream), dz,
                     lddz, w, dx, lddx, rows, out_features, in_features);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_narrow_dw_blocks(long long rows) {
  long long b = (rows + 63) / 64;         // >= 64 rows per workgroup, at most two workgroups per CU
  if (b > 512) b = 512;
  return static_cast<int>(b < 1 ? 1 : b);
}

int rlg_narrow_dw(const float* dz, long long lddz, const float* x, long long ldx, float* grad, double* partials,
                  long long rows, int out_features, int in_features, void* stream) {
  if (rows <= 0 || out_features < 1 || out_features > 256 || in_features < 1 || in_features > rlg::kNarrowMax || !partials)
    return static_cast<int>(hipErrorInvalidValue);
  const int blocks = rlg_narrow_dw_blocks(rows);
  const int chunk = static_cast<int>((rows + blocks - 1) / blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rlg::narrow_dw_partial_kernel, dim3(blocks), dim3(256), 0, st, dz, lddz, x, ldx, partials, rows, chunk,
                     out_features, in_features);
  const int n = out_features * in_features;
  hipLaunchKernelGGL(rlg::narrow_dw_finalize_kernel, dim3((n + 3) / 4), dim3(256), 0, st, partials, blocks, n, grad);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
