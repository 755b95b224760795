// Element-wise pieces of the actor-critic MLP's manual backward pass, gfx950.
//
// The MLP GEMMs stay on rocBLAS/hipBLASLt (torch.mm / addmm); what torch autograd adds around
// them for every hidden layer is an activation-backward kernel plus a separate column
// reduction for the bias gradient (16 % of the epoch in round-1 profiles).  Here both are one
// pass:      dZ = dH * act'(Z)      and      db = sum_rows dZ
// replacing torch's `elu_backward` + `sum(0)` of nn.Linear's backward for the layers built by
// `A2CBuilder._build_sequential_mlp` (rl_games/algos_torch/network_builder.py:118-147,
// forward :498).  act'(z) for ELU(alpha=1): 1 for z > 0, exp(z) otherwise - evaluated from the
// pre-activation exactly like aten's elu_backward (is_result = false); the *Out kinds take the
// derivative from the layer output instead (is_result = true) for in-place activations.
//
// Memory-bound: reads dH and Z, writes dZ (12 B per element); the column sums ride along in
// registers (fp64 per lane, combined through LDS, per-block partials, no atomics).

#include "rlg_device.hpp"

namespace rlg {

constexpr int kActBlock = 256;
constexpr int kActWaves = kActBlock / kWave;
constexpr int kActUnroll = 4;

enum ActKind { kActIdentity = 0, kActElu = 1, kActRelu = 2, kActTanh = 3,
               // the same derivatives evaluated from the layer OUTPUT h = act(z) (in-place activations:
               // only h is kept), like aten's elu_backward(is_result = true): elu' = h > 0 ? 1 : h + 1
               kActEluOut = 17, kActReluOut = 18, kActTanhOut = 19 };

template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
  if (ACT == kActElu) return z > 0.0f ? 1.0f : expf(z);
  if (ACT == kActRelu) return z > 0.0f ? 1.0f : 0.0f;
  if (ACT == kActTanh) {
    const float t = tanhf(z);
    return 1.0f - t * t;
  }
  if (ACT == kActEluOut) return z > 0.0f ? 1.0f : z + 1.0f;
  if (ACT == kActReluOut) return z > 0.0f ? 1.0f : 0.0f;
  if (ACT == kActTanhOut) return 1.0f - z * z;
  return 1.0f;
}

// partials layout: [gridDim.x][C] fp64 column sums of dZ.
// Lane mapping as in column_moments_kernel: a row is C/4 16-byte units; a wave covers
// rpp = 64 / units consecutive rows per pass so its lanes read one contiguous span.
template <int ACT>
__global__ __launch_bounds__(kActBlock) void act_bwd_colsum_kernel(
    const float* dH, const float* __restrict__ Z, float* dZ,   // dZ may alias dH (in-place call)
    long long rows, int C, long long ld, double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];   // [kActWaves][64][4]
  const int upr = C / 4;
  const int lane = lane_id();
  const int wave = wave_id();
  const long long gwave = static_cast<long long>(blockIdx.x) * kActWaves + wave;
  const long long nwaves = static_cast<long long>(gridDim.x) * kActWaves;
  double* out = partials + static_cast<long long>(blockIdx.x) * C;

  for (int cb = 0; cb < upr; cb += kWave) {
    const int w_upr = min(kWave, upr - cb);
    const int rpp = kWave / w_upr;
    const int sub = lane / w_upr;
    const int cu = lane - sub * w_upr;
    const bool active = sub < rpp;
    const long long col = static_cast<long long>(cb + cu) * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const long long groups = (rows + rpp - 1) / rpp;
    for (long long g0 = gwave; g0 < groups; g0 += nwaves * kActUnroll) {
      f32x4 a[kActUnroll], z[kActUnroll];
      bool ok[kActUnroll];
#pragma unroll
      for (int u = 0; u < kActUnroll; ++u) {
        const long long row = (g0 + u * nwaves) * rpp + sub;
        ok[u] = active && (g0 + u * nwaves) < groups && row < rows;
        const long long off = ok[u] ? row * ld + col : 0;
        a[u] = *reinterpret_cast<const f32x4*>(dH + off);
        if (ACT != kActIdentity) z[u] = *reinterpret_cast<const f32x4*>(Z + off);
      }
#pragma unroll
      for (int u = 0; u < kActUnroll; ++u) {
        if (!ok[u]) continue;
        const long long row = (g0 + u * nwaves) * rpp + sub;
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d[k] = (ACT == kActIdentity) ? a[u][k] : a[u][k] * act_grad<ACT>(z[u][k]);
          s[k] += static_cast<double>(d[k]);
        }
        if (ACT != kActIdentity || dZ != dH) *reinterpret_cast<f32x4*>(dZ + row * ld + col) = d;
      }
    }
    double* mine = smem_d + (wave * kWave + lane) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[k] = s[k];
    __syncthreads();
    for (int j = threadIdx.x; j < w_upr * 4; j += kActBlock) {
      const int u_idx = j >> 2, k = j & 3;
      double acc = 0.0;
      for (int w = 0; w < kActWaves; ++w) {
        for (int sb = 0; sb < rpp; ++sb) acc += smem_d[(w * kWave + sb * w_upr + u_idx) * 4 + k];
      }
      out[(cb + u_idx) * 4 + k] = acc;
    }
    __syncthreads();
  }
}

// db[c] (+)= sum over blocks of partials[b][c]
// 1024 threads = 32 row-slices x 32 columns: each thread sums <= nblocks/32 partial rows, so the
// dependent-load chain stays short (this kernel is pure latency).
constexpr int kFinSlices = 32;
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const double* __restrict__ partials,
                                                               int nblocks, int C,
                                                               float* __restrict__ out, int accumulate) {
  __shared__ double part[kFinSlices][33];
  const int col_in_pass = threadIdx.x & 31;
  const int slice = threadIdx.x >> 5;
  for (int c0 = blockIdx.x * 32; c0 < C; c0 += gridDim.x * 32) {
    const int c = c0 + col_in_pass;
    double s = 0.0;
    if (c < C) {
      for (int b = slice; b < nblocks; b += kFinSlices) s += partials[static_cast<long long>(b) * C + c];
    }
    part[slice][col_in_pass] = s;
    __syncthreads();
    if (slice == 0 && c < C) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < kFinSlices; ++k) t += part[k][col_in_pass];
      const float v = static_cast<float>(t);
      out[c] = accumulate ? out[c] + v : v;
    }
    __syncthreads();
  }
}

}  // namespace rlg

extern "C" {

int rlg_act_bwd_num_blocks(long long rows, int cols) {
  // >= 16 elements per thread; small (per-rank, multi-GPU) minibatches still spread over the
  // whole chip instead of a few dozen CUs (4,096 x 400: 100 -> 256 blocks).
  long long need = (rows * cols + 256LL * 16 - 1) / (256LL * 16);
  if (need < 1) need = 1;
  if (need > 256) need = 256;      // one block per CU; fewer partial rows for the finalise pass
  return static_cast<int>(need);
}

int rlg_act_bwd_colsum(const float* d_out, const float* pre_act, float* d_pre, long long rows,
                       int cols, long long ld, int act_kind, double* partials, int num_blocks,
                       void* stream) {
  using namespace rlg;
  if (rows <= 0 || cols <= 0 || cols % 4 != 0 || ld % 4 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  if ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_pre) |
       (act_kind != kActIdentity ? reinterpret_cast<uintptr_t>(pre_act) : 0)) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t shm = static_cast<size_t>(kActWaves) * kWave * 4 * sizeof(double);
  const dim3 grid(num_blocks), block(kActBlock);
  switch (act_kind) {
    case kActIdentity:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActIdentity>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    case kActElu:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActElu>), grid, block, shm, st, d_out, pre_act, d_pre,
                         rows, cols, ld, partials);
      break;
    case kActRelu:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActRelu>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    case kActTanh:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActTanh>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    case kActEluOut:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActEluOut>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    case kActReluOut:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActReluOut>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    case kActTanhOut:
      hipLaunchKernelGGL((act_bwd_colsum_kernel<kActTanhOut>), grid, block, shm, st, d_out, pre_act,
                         d_pre, rows, cols, ld, partials);
      break;
    default:
      return static_cast<int>(hipErrorInvalidValue);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_colsum_finalize(const double* partials, int num_blocks, int cols, float* out, int accumulate,
                        void* stream) {
  int grid = (cols + 31) / 32;
  if (grid > 64) grid = 64;
  hipLaunchKernelGGL(rlg::colsum_finalize_kernel, dim3(grid), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), partials, num_blocks, cols, out, accumulate);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
