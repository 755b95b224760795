// Forward and input-gradient GEMMs of the actor-critic MLP on f32 MFMA with fused epilogues and
// NO LDS staging (gfx950).
//
//   forward   Z = X W^T + b,  H = act(Z)          X [M, K], W [N, K]   (nn.Linear + activation,
//             rl_games/algos_torch/network_builder.py:118-147 `_build_sequential_mlp`, heads :295-311)
//   dX        dZ_prev = (dZ W) * act'(Z_prev)      dZ [M, K=No], W [K=No, N=Mi]   (autograd of the same
//             layers: grad_input = grad_output.mm(weight), then elu_backward)
//
// Operand mapping for v_mfma_f32_32x32x2_f32 (A[m][k]: lane = 32*k + m; B[k][n]: lane = 32*k + n).
// The sum over k is order-free, so MFMA step s of an 8-wide k-panel is given k = s to the lower
// half-wave and k = 4 + s to the upper one: lane (h, m) then needs X[m][k0 + 4h + s], s = 0..3 - ONE
// 16-byte load of 4 consecutive k from its own row, straight from global memory into the MFMA
// source registers.  Each element of X is loaded exactly once per wave; neighbouring panels hit the
// same cache lines (32 B per row per panel), so HBM sees X once.  The W operand is L2/L1 resident
// (<= 320 KB per layer) and is read the same way (forward: W[n][k0+4h..]; dX: W[k0+4h+s][n]).
//   * block = 4 waves stacked in M (128 rows) sharing the same W tile; a wave owns 32 rows x up to
//     4 column blocks of 32 (64 accumulator registers), column blocks are spread evenly over the
//     column tiles (N = 400 -> 13 blocks -> 4+3+3+3), so padding is at the 32-granule only;
//   * next panel's operands are loaded while the current panel's MFMAs run; 2-3 waves per SIMD
//     overlap one wave's epilogue stores with the others' MFMAs;
//   * epilogues: bias + activation (writes Z and H), or act'(Z_prev) multiply (writes dZ_prev) -
//     the passes torch runs as separate kernels around every GEMM.
// Numerics: exact fp32 products, fp32 accumulation; only the summation order differs from the
// rocBLAS/hipBLASLt kernels this replaces.

#include "rlg_device.hpp"

namespace rlg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRgMaxNB = 4;

enum { kRgNone = 0, kRgElu = 1, kRgRelu = 2, kRgTanh = 3 };

template <int ACT>
__device__ __forceinline__ float rg_act(float z) {
  if (ACT == kRgElu) return z > 0.0f ? z : expm1f(z);
  if (ACT == kRgRelu) return z != z ? z : fmaxf(z, 0.0f);       // torch.relu(NaN) = NaN
  if (ACT == kRgTanh) return tanhf(z);
  return z;
}

template <int ACT>
__device__ __forceinline__ float rg_act_grad(float z) {
  if (ACT == kRgElu) return z > 0.0f ? 1.0f : expf(z);
  if (ACT == kRgRelu) return z > 0.0f ? 1.0f : 0.0f;
  if (ACT == kRgTanh) {
    const float t = tanhf(z);
    return 1.0f - t * t;
  }
  return 1.0f;
}

struct RowGemmArgs {
  const float* a;        // [M, K] row operand (X or dZ), leading dim lda
  const float* w;        // forward: W [N, K]; dX: W [K, N]
  const float* bias;     // forward: [N] or nullptr
  const float* zprev;    // dX: pre-activations [M, N] (ld ldo) or nullptr (no activation backward)
  float* out0;           // forward: Z (or nullptr); dX: dZ_prev
  float* out1;           // forward: H; dX: unused
  int M, N, K;
  int lda, ldw, ldo;
  int col_tiles;         // column tiles per row of blocks
};

// kDx = false: forward (B[k][n] = W[n][k]);  kDx = true: dX (B[k][n] = W[k][n]).
template <int ACT, bool kDx>
__global__ __launch_bounds__(256, 2) void mlp_rowgemm_kernel(RowGemmArgs p) {
  const int lane = lane_id();
  const int h = lane >> 5;
  const int j = lane & 31;
  const int row_tile = blockIdx.x / p.col_tiles;
  const int col_tile = blockIdx.x - row_tile * p.col_tiles;
  const int nblk = (p.N + 31) >> 5;
  // column blocks [b_begin, b_end) of this tile: even split
  const int b_begin = (nblk * col_tile) / p.col_tiles;
  const int b_end = (nblk * (col_tile + 1)) / p.col_tiles;
  const int nb = b_end - b_begin;                       // 1..kRgMaxNB (block-uniform)
  const int m0 = row_tile * 128 + wave_id() * 32;
  if (m0 >= p.M) return;                                // no barriers in this kernel
  const int row = min(m0 + j, p.M - 1);                 // clamp: tail rows are computed, never stored
  const float* arow = p.a + static_cast<long long>(row) * p.lda + 4 * h;
  const int n0 = b_begin * 32;

  f32x16 acc[kRgMaxNB];
#pragma unroll
  for (int b = 0; b < kRgMaxNB; ++b) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[b][q] = 0.0f;
  }

  // B operand addressing.  forward: lane (h, n) reads 4 consecutive k of row n of W.
  //                         dX    : lane (h, n) reads column n of rows k0+4h+s of W.
  int bcol[kRgMaxNB];
#pragma unroll
  for (int b = 0; b < kRgMaxNB; ++b) bcol[b] = min(n0 + 32 * b + j, p.N - 1);

  auto load_panel = [&](f32x4& av, f32x4 (&bv)[kRgMaxNB], int k0) {
    const int kk = k0 + 4 * h;                           // this half-wave's 4 k values: kk .. kk+3
    const bool kin = kk < p.K;                           // K % 4 == 0: all four in or all out
    const int ks = kin ? kk : 0;
    av = *reinterpret_cast<const f32x4*>(arow + (ks - 4 * h));
    if (!kin) av = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < kRgMaxNB; ++b) {
      if (b < nb) {
        if (!kDx) {
          bv[b] = *reinterpret_cast<const f32x4*>(p.w + static_cast<long long>(bcol[b]) * p.ldw + ks);
        } else {
          const float* wp = p.w + static_cast<long long>(ks) * p.ldw + bcol[b];
          bv[b] = f32x4{wp[0], wp[p.ldw], wp[2 * p.ldw], wp[3 * p.ldw]};
        }
        // (a zeroed A chunk already nulls the products of an out-of-range k chunk)
      }
    }
  };
  auto mfma_panel = [&](const f32x4& av, const f32x4 (&bv)[kRgMaxNB]) {
#pragma unroll
    for (int b = 0; b < kRgMaxNB; ++b) {
      if (b < nb) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[b][s], acc[b], 0, 0, 0);
      }
    }
  };

  f32x4 a_cur, b_cur[kRgMaxNB];
  load_panel(a_cur, b_cur, 0);
  for (int k0 = 8; k0 < p.K; k0 += 8) {
    f32x4 a_nxt, b_nxt[kRgMaxNB];
    load_panel(a_nxt, b_nxt, k0);
    mfma_panel(a_cur, b_cur);
    a_cur = a_nxt;
#pragma unroll
    for (int b = 0; b < kRgMaxNB; ++b) b_cur[b] = b_nxt[b];
  }
  mfma_panel(a_cur, b_cur);
  // One more issue slot in front of the first read of an accumulator: the MFMA result is not interlocked
  // against VALU reads (18 wait states for this shape) and hipcc counts one short across the branches of
  // the last panel (tools/audit_mfma.py; csrc/mlp_chain.hip has the measurements).
  asm volatile("s_nop 0" : "+v"(acc[0]));
#pragma unroll
  for (int b = 1; b < kRgMaxNB; ++b) asm volatile("" : "+v"(acc[b]));

  // ---- epilogue: acc[b][q] is C[m0 + (q&3) + 8*(q>>2) + 4h][n0 + 32b + j]
#pragma unroll
  for (int b = 0; b < kRgMaxNB; ++b) {
    if (b >= nb) continue;
    const int col = n0 + 32 * b + j;
    if (col >= p.N) continue;
    const float bias = (!kDx && p.bias) ? p.bias[col] : 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = m0 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (r >= p.M) continue;
      const long long o = static_cast<long long>(r) * p.ldo + col;
      if (!kDx) {
        const float z = acc[b][q] + bias;
        if (p.out0) p.out0[o] = z;
        p.out1[o] = rg_act<ACT>(z);
      } else {
        const float g = (ACT != kRgNone && p.zprev) ? rg_act_grad<ACT>(p.zprev[o]) : 1.0f;
        p.out0[o] = acc[b][q] * g;
      }
    }
  }
}

template <bool kDx>
static int launch_rowgemm(const RowGemmArgs& p, int act, hipStream_t st) {
  const int nblk = (p.N + 31) / 32;
  RowGemmArgs q = p;
  q.col_tiles = (nblk + kRgMaxNB - 1) / kRgMaxNB;
  const int row_tiles = (p.M + 127) / 128;
  const dim3 grid(static_cast<unsigned>(row_tiles) * q.col_tiles), block(256);
  switch (act) {
    case kRgNone: hipLaunchKernelGGL((mlp_rowgemm_kernel<kRgNone, kDx>), grid, block, 0, st, q); break;
    case kRgElu: hipLaunchKernelGGL((mlp_rowgemm_kernel<kRgElu, kDx>), grid, block, 0, st, q); break;
    case kRgRelu: hipLaunchKernelGGL((mlp_rowgemm_kernel<kRgRelu, kDx>), grid, block, 0, st, q); break;
    case kRgTanh: hipLaunchKernelGGL((mlp_rowgemm_kernel<kRgTanh, kDx>), grid, block, 0, st, q); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

}  // namespace rlg

extern "C" {

int rlg_mlp_rowgemm_supported(int in_features, long long lda) {
  return (in_features >= 4 && in_features % 4 == 0 && lda % 4 == 0) ? 1 : 0;
}

// H = act(X W^T + b) (and Z = X W^T + b when pre_act != NULL).  X [rows, K] (ld ldx), W [N, K],
// outputs [rows, N] (ld ldo).  K % 4 == 0, 16-byte aligned X / W.
int rlg_mlp_linear_act_forward(const float* x, long long ldx, const float* w, const float* bias_or_null,
                               float* pre_act_or_null, float* out, long long ldo, int rows, int out_features,
                               int in_features, int act_kind, void* stream) {
  using namespace rlg;
  if (rows <= 0 || out_features <= 0 || !rlg_mlp_rowgemm_supported(in_features, ldx) ||
      (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  RowGemmArgs p;
  p.a = x;
  p.w = w;
  p.bias = bias_or_null;
  p.zprev = nullptr;
  p.out0 = pre_act_or_null;
  p.out1 = out;
  p.M = rows;
  p.N = out_features;
  p.K = in_features;
  p.lda = static_cast<int>(ldx);
  p.ldw = in_features;
  p.ldo = static_cast<int>(ldo);
  p.col_tiles = 1;
  return launch_rowgemm<false>(p, act_kind, static_cast<hipStream_t>(stream));
}

// dZ_prev = (dZ W) * act'(Z_prev).  dZ [rows, No] (ld lddz), W [No, Mi], Z_prev / dZ_prev [rows, Mi]
// (ld ldo).  No % 4 == 0, 16-byte aligned dZ.  z_prev NULL (or act_kind 0): plain dZ W.
int rlg_mlp_linear_act_backward(const float* dz, long long lddz, const float* w, const float* z_prev_or_null,
                                float* dz_prev, long long ldo, int rows, int out_features, int in_features,
                                int act_kind, void* stream) {
  using namespace rlg;
  if (rows <= 0 || in_features <= 0 || !rlg_mlp_rowgemm_supported(out_features, lddz) ||
      reinterpret_cast<uintptr_t>(dz) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  RowGemmArgs p;
  p.a = dz;
  p.w = w;
  p.bias = nullptr;
  p.zprev = z_prev_or_null;
  p.out0 = dz_prev;
  p.out1 = nullptr;
  p.M = rows;
  p.N = in_features;
  p.K = out_features;
  p.lda = static_cast<int>(lddz);
  p.ldw = in_features;
  p.ldo = static_cast<int>(ldo);
  p.col_tiles = 1;
  return launch_rowgemm<true>(p, z_prev_or_null ? act_kind : 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
