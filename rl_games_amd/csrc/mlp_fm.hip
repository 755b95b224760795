// Feature-major MLP forward / input-gradient GEMMs on f32 MFMA, LDS-free, fused epilogues (gfx950).
//
// With activations stored FEATURE-MAJOR (H^T [features, samples]) both remaining GEMMs of the
// actor-critic MLP become sums of outer products of contiguous rows - the shape that feeds
// v_mfma_f32_32x32x2_f32 straight from coalesced global loads (see mlp_dw.hip):
//   forward  Z_l^T [N, M] = sum_k  W_l^T[k, :] (x) H_{l-1}^T[k, :]  (+ b),  H_l^T = act(Z_l^T)
//   dX       dZ_{l-1}^T [Mi, M] = ( sum_o  W_l[o, :] (x) dZ_l^T[o, :] ) * act'(Z_{l-1}^T)
// replacing nn.Linear + activation and their autograd (rl_games/algos_torch/network_builder.py:
// 118-147 `_build_sequential_mlp`, heads :295-311, forward :498-512) - the library GEMMs AND the
// element-wise passes torch runs around them (bias add, ELU, elu_backward).
//   A operand = a [K, N] row-major matrix (forward: W^T, kept next to W; dX: W as stored),
//   B operand = a [K, M] feature-major activation / gradient.  Wave tile 64 (2 interleaved blocks
//   along N) x 128 (4 interleaved blocks along M): lane j loads 2 resp. 4 consecutive floats per
//   k, 8 MFMAs per k-pair, K is short (22..400) so there is no split-K.
// The weight-gradient kernel needs SAMPLE-major operands, so the epilogue also writes a
// sample-major copy (H resp. dZ) - 8 consecutive floats per lane and row, 32 rows per store.
// Numerics: exact fp32 products, fp32 accumulation; summation order differs from the library.

#include "rlg_device.hpp"

namespace rlg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFmUnroll = 4;

enum { kFmNone = 0, kFmElu = 1, kFmRelu = 2, kFmTanh = 3 };

template <int ACT>
__device__ __forceinline__ float fm_act(float z) {
  if (ACT == kFmElu) return z > 0.0f ? z : expm1f(z);
  if (ACT == kFmRelu) return fmaxf(z, 0.0f);
  if (ACT == kFmTanh) return tanhf(z);
  return z;
}

template <int ACT>
__device__ __forceinline__ float fm_act_grad(float z) {
  if (ACT == kFmElu) return z > 0.0f ? 1.0f : expf(z);
  if (ACT == kFmRelu) return z > 0.0f ? 1.0f : 0.0f;
  if (ACT == kFmTanh) {
    const float t = tanhf(z);
    return 1.0f - t * t;
  }
  return 1.0f;
}

struct FmArgs {
  const float* a;       // [K, N] (ld lda): forward W^T, dX W
  const float* b;       // [K, M] (ld ldm): H_prev^T resp. dZ^T
  const float* bias;    // forward: [N] or nullptr
  const float* zprev;   // dX: Z_prev^T [N, M] (ld ldm) or nullptr
  float* out_z;         // forward: Z^T [N, M] or nullptr;   dX: unused
  float* out_fm;        // forward: H^T [N, M];              dX: dZ_prev^T [N, M]   (nullptr: skip)
  float* out_sm;        // sample-major copy [M, N] (ld ld_sm) of out_fm's values, or nullptr
  int K, N, M;
  int lda, ldm, ld_sm;
};

// kDx = false: forward epilogue (bias, activation); true: dX epilogue (act' multiply).
template <int ACT, bool kDx>
__global__ __launch_bounds__(256, 2) void mlp_fm_kernel(FmArgs p) {
  constexpr int BO = 2, BI = 4;
  const int lane = lane_id();
  const int h = lane >> 5;
  const int j = lane & 31;
  const int n_tiles = (p.N + 63) >> 6;
  const int n_tile = blockIdx.x % n_tiles;          // consecutive blocks share the same samples (B rows)
  const int m_tile = blockIdx.x / n_tiles;
  const int n0 = n_tile * 64;
  const int m0 = m_tile * 512 + wave_id() * 128;
  if (m0 >= p.M) return;                            // no barriers below
  const int n_col = min(n0 + BO * j, p.N - BO);      // clamp (N even); out-of-range rows never stored
  const int m_col = min(m0 + BI * j, p.M - BI);      // M % 4 == 0
  const float* pa = p.a + n_col;
  const float* pb = p.b + m_col;

  f32x16 acc[BO][BI];
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b = 0; b < BI; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
    }
  }
  const int pairs = (p.K + 1) >> 1;
  const int full = p.K >> 1;
  f32x2 a_cur[kFmUnroll];
  f32x4 b_cur[kFmUnroll];
  auto load_batch = [&](f32x2 (&av)[kFmUnroll], f32x4 (&bv)[kFmUnroll], int p0) {
#pragma unroll
    for (int u = 0; u < kFmUnroll; ++u) {
      const long long k = 2LL * (p0 + u) + h;
      av[u] = *reinterpret_cast<const f32x2*>(pa + k * p.lda);
      bv[u] = *reinterpret_cast<const f32x4*>(pb + k * p.ldm);
    }
  };
  auto mfma_batch = [&](const f32x2 (&av)[kFmUnroll], const f32x4 (&bv)[kFmUnroll]) {
#pragma unroll
    for (int u = 0; u < kFmUnroll; ++u) {
#pragma unroll
      for (int a = 0; a < BO; ++a) {
#pragma unroll
        for (int b = 0; b < BI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
      }
    }
  };
  int pp = 0;
  if (pp + kFmUnroll <= full) {
    load_batch(a_cur, b_cur, pp);
    pp += kFmUnroll;
    while (pp + kFmUnroll <= full) {
      f32x2 a_nxt[kFmUnroll];
      f32x4 b_nxt[kFmUnroll];
      load_batch(a_nxt, b_nxt, pp);
      mfma_batch(a_cur, b_cur);
#pragma unroll
      for (int u = 0; u < kFmUnroll; ++u) { a_cur[u] = a_nxt[u]; b_cur[u] = b_nxt[u]; }
      pp += kFmUnroll;
    }
    mfma_batch(a_cur, b_cur);
  }
  for (; pp < pairs; ++pp) {
    const long long k = 2LL * pp + h;
    f32x2 av = {0.0f, 0.0f};
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k < p.K) {
      av = *reinterpret_cast<const f32x2*>(pa + k * p.lda);
      bv = *reinterpret_cast<const f32x4*>(pb + k * p.ldm);
    }
#pragma unroll
    for (int a = 0; a < BO; ++a) {
#pragma unroll
      for (int b = 0; b < BI; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue.  acc[a][b][q] = C[n = n0 + 2*row + a][m = m0 + 4*j + b], row = (q&3) + 8*(q>>2) + 4h.
  const int m = m0 + BI * j;
  const bool m_ok = m + BI <= p.M;
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
      const int n = n0 + BO * row + a;
      const bool ok = m_ok && n < p.N;
      const long long o = static_cast<long long>(ok ? n : 0) * p.ldm + (ok ? m : 0);
      f32x4 v = {acc[a][0][q], acc[a][1][q], acc[a][2][q], acc[a][3][q]};
      if (!kDx) {
        const float bias = (p.bias && ok) ? p.bias[n] : 0.0f;
        v += bias;
        if (ok && p.out_z) *reinterpret_cast<f32x4*>(p.out_z + o) = v;
#pragma unroll
        for (int b = 0; b < BI; ++b) v[b] = fm_act<ACT>(v[b]);
      } else if (ACT != kFmNone) {
        if (p.zprev) {
          const f32x4 z = *reinterpret_cast<const f32x4*>(p.zprev + o);
#pragma unroll
          for (int b = 0; b < BI; ++b) v[b] *= fm_act_grad<ACT>(z[b]);
        }
      }
      if (ok && p.out_fm) *reinterpret_cast<f32x4*>(p.out_fm + o) = v;
#pragma unroll
      for (int b = 0; b < BI; ++b) acc[a][b][q] = v[b];      // keep the final values for the copy below
    }
  }
  if (p.out_sm && m_ok) {
    // sample-major copy: for sample m+b, this lane holds n = n0 + 2*(8g + 4h) + {0..7}: (q&3, a) pairs
#pragma unroll
    for (int b = 0; b < BI; ++b) {
      float* dst_row = p.out_sm + static_cast<long long>(m + b) * p.ld_sm;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = n0 + BO * (8 * g + 4 * h);
        const f32x4 lo = {acc[0][b][4 * g + 0], acc[1][b][4 * g + 0], acc[0][b][4 * g + 1], acc[1][b][4 * g + 1]};
        const f32x4 hi = {acc[0][b][4 * g + 2], acc[1][b][4 * g + 2], acc[0][b][4 * g + 3], acc[1][b][4 * g + 3]};
        if (nb + 8 <= p.N) {
          *reinterpret_cast<f32x4*>(dst_row + nb) = lo;
          *reinterpret_cast<f32x4*>(dst_row + nb + 4) = hi;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (nb + e < p.N) dst_row[nb + e] = lo[e];
            if (nb + 4 + e < p.N) dst_row[nb + 4 + e] = hi[e];
          }
        }
      }
    }
  }
}

// [R, C] -> [C, R] through a padded LDS tile (weights W -> W^T, obs -> obs^T, d_heads -> d_heads^T).
__global__ __launch_bounds__(256) void fm_transpose_kernel(const float* __restrict__ src, long long lds_,
                                                           float* __restrict__ dst, long long ldd, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[static_cast<long long>(r) * lds_ + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) dst[static_cast<long long>(c) * ldd + r] = tile[tx][ty + 8 * i];
  }
}

// Row sums of a feature-major gradient dZ^T [N, M] -> bias gradient [N].  One block per row,
// fp64 accumulation in a fixed order (deterministic).
__global__ __launch_bounds__(256) void fm_row_sum_kernel(const float* __restrict__ dzt, long long ldm, int M,
                                                         float* __restrict__ out) {
  __shared__ double red[4];
  const float* row = dzt + static_cast<long long>(blockIdx.x) * ldm;
  double s[1] = {0.0};
  for (int i = threadIdx.x * 4; i < M; i += 256 * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    s[0] += (static_cast<double>(v[0]) + static_cast<double>(v[1])) +
            (static_cast<double>(v[2]) + static_cast<double>(v[3]));
  }
  block_sum<1, 256>(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = static_cast<float>(s[0]);
}

template <bool kDx>
static int launch_fm(const FmArgs& p, int act, hipStream_t st) {
  const int n_tiles = (p.N + 63) / 64;
  const int m_tiles = (p.M + 511) / 512;
  const dim3 grid(static_cast<unsigned>(n_tiles) * m_tiles), block(256);
  switch (act) {
    case kFmNone: hipLaunchKernelGGL((mlp_fm_kernel<kFmNone, kDx>), grid, block, 0, st, p); break;
    case kFmElu: hipLaunchKernelGGL((mlp_fm_kernel<kFmElu, kDx>), grid, block, 0, st, p); break;
    case kFmRelu: hipLaunchKernelGGL((mlp_fm_kernel<kFmRelu, kDx>), grid, block, 0, st, p); break;
    case kFmTanh: hipLaunchKernelGGL((mlp_fm_kernel<kFmTanh, kDx>), grid, block, 0, st, p); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

}  // namespace rlg

extern "C" {

// Z^T = W X^T + b, H^T = act(Z^T).  wt = W^T [K, N]; xt = X^T [K, M] (ld ldm); outputs feature-major
// [N, M] (ld ldm) and, optionally, sample-major H [M, N] (ld ld_sm).  N even, M % 4 == 0, ldm % 4 == 0.
int rlg_mlp_fm_forward(const float* wt, const float* xt, const float* bias_or_null, float* zt_or_null,
                       float* ht_or_null, float* h_sm_or_null, long long ld_sm, int in_features,
                       int out_features, int samples, long long ldm, int act_kind, void* stream) {
  using namespace rlg;
  if (in_features <= 0 || out_features < 2 || out_features % 2 != 0 || samples < 4 || samples % 4 != 0 ||
      ldm % 4 != 0 || (reinterpret_cast<uintptr_t>(wt) % 8) != 0 || (reinterpret_cast<uintptr_t>(xt) % 16) != 0)
    return static_cast<int>(hipErrorInvalidValue);
  FmArgs p;
  p.a = wt;
  p.b = xt;
  p.bias = bias_or_null;
  p.zprev = nullptr;
  p.out_z = zt_or_null;
  p.out_fm = ht_or_null;
  p.out_sm = h_sm_or_null;
  p.K = in_features;
  p.N = out_features;
  p.M = samples;
  p.lda = out_features;
  p.ldm = static_cast<int>(ldm);
  p.ld_sm = static_cast<int>(ld_sm);
  return launch_fm<false>(p, act_kind, static_cast<hipStream_t>(stream));
}

// dZ_prev^T = (W^T dZ^T) * act'(Z_prev^T).  w = W [No, Mi] as stored; dzt [No, M]; zt_prev / outputs [Mi, M].
int rlg_mlp_fm_backward(const float* w, const float* dzt, const float* zt_prev_or_null, float* dzt_prev_or_null,
                        float* dz_prev_sm_or_null, long long ld_sm, int out_features, int in_features,
                        int samples, long long ldm, int act_kind, void* stream) {
  using namespace rlg;
  if (out_features <= 0 || in_features < 2 || in_features % 2 != 0 || samples < 4 || samples % 4 != 0 ||
      ldm % 4 != 0 || (reinterpret_cast<uintptr_t>(w) % 8) != 0 || (reinterpret_cast<uintptr_t>(dzt) % 16) != 0)
    return static_cast<int>(hipErrorInvalidValue);
  FmArgs p;
  p.a = w;
  p.b = dzt;
  p.bias = nullptr;
  p.zprev = zt_prev_or_null;
  p.out_z = nullptr;
  p.out_fm = dzt_prev_or_null;
  p.out_sm = dz_prev_sm_or_null;
  p.K = out_features;
  p.N = in_features;
  p.M = samples;
  p.lda = in_features;
  p.ldm = static_cast<int>(ldm);
  p.ld_sm = static_cast<int>(ld_sm);
  return launch_fm<true>(p, zt_prev_or_null ? act_kind : 0, static_cast<hipStream_t>(stream));
}

// dst [C, R] (ld ldd) = src [R, C] (ld lds)^T
int rlg_fm_transpose(const float* src, long long lds, float* dst, long long ldd, int rows, int cols,
                     void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(rlg::fm_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, lds, dst, ldd, rows, cols);
  RLG_RETURN_LAUNCH_STATUS();
}

// out[n] = sum_m dzt[n, m]   (bias gradient from a feature-major gradient)
int rlg_fm_row_sum(const float* dzt, long long ldm, int num_rows, int samples, float* out, void* stream) {
  if (num_rows <= 0 || samples <= 0 || samples % 4 != 0) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::fm_row_sum_kernel, dim3(num_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dzt, ldm, samples, out);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
