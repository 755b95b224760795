// fp32 MFMA GEMMs with fused epilogues for the actor-critic MLP, gfx950 (MI355X).
//
// "MFMA only if the actor/critic MLP GEMM is actually fused in" (north star): these kernels fuse
// what torch runs as separate launches around each nn.Linear of A2CBuilder's MLP
// (rl_games/algos_torch/network_builder.py:118-147, forward :498):
//   forward   Y = act(X W^T + b)              (addmm + activation; also keeps Z = X W^T + b)
//   backward  dX = (dY W) * act'(Z_prev)      (mm + activation backward + bias-grad column sums)
//             dW = dY^T X                      (mm with a 32,768-long reduction -> split-K)
// Arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (an fmaf chain in k
// order); gfx950 has no TF32, so this is the same numerics class as the rocBLAS/hipBLASLt fp32
// kernels it replaces; only the summation order differs.
//
// Tiling (all three): a block is 4 waves; wave w owns a 32-row strip and NT 32x32 accumulator
// tiles (NT*16 AGPRs).  Operands are staged through LDS in [rows][BK=16] panels with a 20-float
// row stride (16-byte aligned, conflict-free ds_read_b128), k-major so that both MFMA operands
// are read with the same pattern: lane l gets 4 consecutive k of row (l & 31) from chunk
// (l >> 5), i.e. MFMA step s multiplies k = s (lanes 0-31) and k = 4+s (lanes 32-63).
// Global loads of panel p+1 are issued into registers before the MFMAs of panel p (register
// prefetch), one LDS buffer, two barriers per panel; NT*8 MFMAs (>= 512 cycles) per barrier pair.

#include "rlg_device.hpp"
#include <cstdlib>

namespace rlg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;         // k-panel depth
constexpr int kLdk = 20;        // LDS row stride in floats (16 + 4 pad)
constexpr int kEpiLd = 36;      // row stride of the per-wave 32x32 epilogue staging tile

enum { kEpiNone = 0, kEpiElu = 1, kEpiRelu = 2, kEpiTanh = 3 };

template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if (ACT == kEpiElu) return z > 0.0f ? z : expm1f(z);
  if (ACT == kEpiRelu) return fmaxf(z, 0.0f);
  if (ACT == kEpiTanh) return tanhf(z);
  return z;
}

// Stage a [ROWS x 16] k-major panel: global (row-major, leading dim ld, k contiguous) -> registers.
// Rows >= rows_total and k >= K are zero-filled.  ROWS*4 16-byte chunks are spread over THREADS.
template <int ROWS, int THREADS>
struct Panel {
  static constexpr int kChunks = ROWS * 4;
  static constexpr int kPerThread = (kChunks + THREADS - 1) / THREADS;
  f32x4 r[kPerThread];
  int k_limit;   // chunks whose first k >= k_limit are zeroed at store() time (k tail of the last panel)

  // vec_ok (wave-uniform): 16-byte aligned base, ld % 4 == 0 and K % 4 == 0.  Then every chunk is
  // either entirely inside or entirely outside [0, K), rows/k are CLAMPED instead of branched on
  // (out-of-range rows only feed output rows/columns that are never stored) and the k tail is
  // zeroed with a select: the hot loop has no divergent control flow.
  __device__ __forceinline__ void load(const float* __restrict__ g, long long ld, int row0, int rows_total,
                                       int k0, int K, bool vec_ok) {
    k_limit = 16;
    if (vec_ok) {
      // issue only: nothing here consumes the loaded registers, so the loads stay in flight across
      // the MFMAs of the current panel; the k-tail select happens in store().
      k_limit = K - k0;
#pragma unroll
      for (int i = 0; i < kPerThread; ++i) {
        const int c = i * THREADS + threadIdx.x;
        const int row = min(c >> 2, ROWS - 1), kc = (c & 3) * 4;
        const int rr = min(row0 + row, rows_total - 1);
        const int kk = min(k0 + kc, K - 4);
        r[i] = *reinterpret_cast<const f32x4*>(g + static_cast<long long>(rr) * ld + kk);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int c = i * THREADS + threadIdx.x;
      const int row = c >> 2, kc = (c & 3) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < kChunks && row0 + row < rows_total) {
        const float* p = g + static_cast<long long>(row0 + row) * ld + k0 + kc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (k0 + kc + q < K) v[q] = p[q];
        }
      }
      r[i] = v;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int c = i * THREADS + threadIdx.x;
      if (c < kChunks) {
        const int row = c >> 2, kc = (c & 3) * 4;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + row * kLdk + kc) = (kc < k_limit) ? r[i] : z;
      }
    }
  }
};

// One k-panel of MFMAs for a wave: A strip rows [a_row0, +32), NT B tiles.  The B fragments of
// tile t+1 are read from LDS before the MFMAs of tile t are issued (register double buffer), so
// the LDS latency hides behind 8 x 64-cycle MFMAs.
template <int NT>
__device__ __forceinline__ void mfma_panel(const float* __restrict__ As, const float* __restrict__ Bs,
                                           int a_row0, f32x16 (&acc)[NT]) {
  const int lane = lane_id();
  const int r = lane & 31, h = lane >> 5;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + (a_row0 + r) * kLdk + h * 4);
  const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + (a_row0 + r) * kLdk + 8 + h * 4);
  const float* bp = Bs + r * kLdk + h * 4;
  f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
  f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 8);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4 n0 = b0, n1 = b1;
    if (t + 1 < NT) {
      n0 = *reinterpret_cast<const f32x4*>(bp + (t + 1) * 32 * kLdk);
      n1 = *reinterpret_cast<const f32x4*>(bp + (t + 1) * 32 * kLdk + 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[t], 0, 0, 0);
    b0 = n0;
    b1 = n1;
  }
}

// Epilogue helper: the wave's 32x32 accumulator tile (C/D layout col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) goes through a wave-private LDS tile so that
// each lane ends up with 4 consecutive columns of a row: 16-byte global stores, 8 rows x 128 B per
// wave instruction.  `emit(row_in_tile, col_in_tile, float4 value)` is called 4 times per lane.
template <typename F>
__device__ __forceinline__ void tile_transposed(const f32x16& acc, float* __restrict__ stage, F emit) {
  const int lane = lane_id();
  const int cl = lane & 31, rh = (lane >> 5) * 4;
#pragma unroll
  for (int q = 0; q < 16; ++q) stage[((q & 3) + 8 * (q >> 2) + rh) * kEpiLd + cl] = acc[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int c4 = (lane & 7) * 4, r8 = lane >> 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = r8 + 8 * j;
    emit(row, c4, *reinterpret_cast<const f32x4*>(stage + row * kEpiLd + c4));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------
// Forward: Z = X W^T + b ; H = act(Z).   X [M,K] (ldx), W [N,K] (ldw), both k-contiguous.
// ---------------------------------------------------------------------------------
template <int WPB, int NT, int ACT>
__global__ __launch_bounds__(WPB * 64) void mlp_fwd_kernel(
    const float* __restrict__ X, long long ldx, const float* __restrict__ W, long long ldw,
    const float* __restrict__ bias, float* __restrict__ Z, long long ldz, float* __restrict__ Hout,
    long long ldh, int M, int N, int K) {
  constexpr int BM = WPB * 32, THREADS = WPB * 64;
  constexpr int kBsFloats = NT * 32 * kLdk;
  constexpr int kStageFloats = WPB * 32 * kEpiLd;
  __shared__ __attribute__((aligned(16))) float As[BM * kLdk];
  __shared__ __attribute__((aligned(16))) float Bs[kBsFloats > kStageFloats ? kBsFloats : kStageFloats];
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * (NT * 32);
  const int wave = wave_id();
  const bool vec_x = (ldx % 4 == 0) && (K % 4 == 0) && aligned16(X);
  const bool vec_w = (ldw % 4 == 0) && (K % 4 == 0) && aligned16(W);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
  }
  Panel<BM, THREADS> pa;
  Panel<NT * 32, THREADS> pb;
  const int nk = (K + kBK - 1) / kBK;
  pa.load(X, ldx, row0, M, 0, K, vec_x);
  pb.load(W, ldw, col0, N, 0, K, vec_w);
  for (int kt = 0; kt < nk; ++kt) {
    pa.store(As);
    pb.store(Bs);
    __syncthreads();
    if (kt + 1 < nk) {
      pa.load(X, ldx, row0, M, (kt + 1) * kBK, K, vec_x);
      pb.load(W, ldw, col0, N, (kt + 1) * kBK, K, vec_w);
    }
    mfma_panel<NT>(As, Bs, wave * 32, acc);
    __syncthreads();
  }
  // epilogue (Bs is free after the last barrier: reuse it as the staging area)
  float* stage = Bs + wave * 32 * kEpiLd;
  const bool vec_out = (ldh % 4 == 0) && aligned16(Hout) && (!Z || ((ldz % 4 == 0) && aligned16(Z)));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tcol0 = col0 + t * 32;
    if (tcol0 >= N) continue;
    tile_transposed(acc[t], stage, [&](int r, int c, f32x4 v) {
      const int row = row0 + wave * 32 + r, col = tcol0 + c;
      if (row >= M || col >= N) return;
      f32x4 z, hval;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float b = (bias && col + q < N) ? bias[col + q] : 0.0f;
        z[q] = v[q] + b;
        hval[q] = act_fwd<ACT>(z[q]);
      }
      if (vec_out && col + 3 < N) {
        if (Z) *reinterpret_cast<f32x4*>(Z + static_cast<long long>(row) * ldz + col) = z;
        *reinterpret_cast<f32x4*>(Hout + static_cast<long long>(row) * ldh + col) = hval;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (col + q < N) {
            if (Z) Z[static_cast<long long>(row) * ldz + col + q] = z[q];
            Hout[static_cast<long long>(row) * ldh + col + q] = hval[q];
          }
        }
      }
    });
  }
}

template <int WPB, int NT>
static int launch_fwd(int act, const float* X, long long ldx, const float* W, long long ldw,
                      const float* bias, float* Z, long long ldz, float* Hout, long long ldh, int M, int N,
                      int K, hipStream_t st) {
  const dim3 grid((M + WPB * 32 - 1) / (WPB * 32), (N + NT * 32 - 1) / (NT * 32)), block(WPB * 64);
#define RLG_FWD(ACT)                                                                                       \
  hipLaunchKernelGGL((mlp_fwd_kernel<WPB, NT, ACT>), grid, block, 0, st, X, ldx, W, ldw, bias, Z, ldz, Hout, \
                     ldh, M, N, K)
  switch (act) {
    case kEpiNone: RLG_FWD(kEpiNone); break;
    case kEpiElu: RLG_FWD(kEpiElu); break;
    case kEpiRelu: RLG_FWD(kEpiRelu); break;
    case kEpiTanh: RLG_FWD(kEpiTanh); break;
    default: return static_cast<int>(hipErrorInvalidValue);
  }
#undef RLG_FWD
  RLG_RETURN_LAUNCH_STATUS();
}

}  // namespace rlg

extern "C" {

int rlg_mlp_forward_layer(const float* x, long long ldx, const float* weight, long long ldw,
                          const float* bias_or_null, float* pre_act_or_null, long long ldz, float* out,
                          long long ldh, int rows, int out_features, int in_features, int act_kind,
                          void* stream) {
  using namespace rlg;
  if (rows <= 0 || out_features <= 0 || in_features <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles = (out_features + 31) / 32;
  static const int force_nt = getenv("RLG_GEMM_NT") ? atoi(getenv("RLG_GEMM_NT")) : 0;
  static const int force_wpb = getenv("RLG_GEMM_WPB") ? atoi(getenv("RLG_GEMM_WPB")) : 0;
#define RLG_GO(WPB, NT) \
  return launch_fwd<WPB, NT>(act_kind, x, ldx, weight, ldw, bias_or_null, pre_act_or_null, ldz, out, ldh, \
                             rows, out_features, in_features, st)
  // NT <= 7 keeps the kernel at <= 256 registers -> 2 waves per SIMD, so the epilogue / panel loads
  // of one block overlap the MFMAs of another; wider layers use several column blocks.
  int nt = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : tiles <= 7 ? 7 : (tiles <= 10 ? 5 : 7);
  if (force_nt) nt = force_nt;
  const int wpb = force_wpb ? force_wpb : 2;
  if (wpb == 4) {
    switch (nt) {
      case 1: RLG_GO(4, 1);
      case 2: RLG_GO(4, 2);
      case 4: RLG_GO(4, 4);
      case 5: RLG_GO(4, 5);
      case 7: RLG_GO(4, 7);
      default: RLG_GO(4, 13);
    }
  }
  switch (nt) {
    case 1: RLG_GO(2, 1);
    case 2: RLG_GO(2, 2);
    case 4: RLG_GO(2, 4);
    case 5: RLG_GO(2, 5);
    case 7: RLG_GO(2, 7);
    default: RLG_GO(2, 13);
  }
#undef RLG_GO
}

}  // extern "C"
