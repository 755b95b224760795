// Weight-gradient GEMMs of the actor-critic MLP on f32 MFMA, all layers in ONE launch (gfx950).
//
//   G_l [No, Mi] = dZ_l^T X_l = sum over the minibatch rows r of  dZ_l[r, :]^T (x) X_l[r, :]
//
// replaces autograd's `grad_weight = grad_output.t().mm(input)` of every nn.Linear built by
// A2CBuilder (rl_games/algos_torch/network_builder.py:118-147, heads :295-311).  These are the
// worst-shaped GEMMs of the path - a 32,768-long reduction into a 400x108 ... 22x100 output -
// where the tuned rocBLAS/hipBLASLt solutions reach 6-54 TFLOP/s (250 us per optimiser step, a
// third of it; tools/bench_dw.py).
//
// A sum of outer products needs NO operand staging for v_mfma_f32_32x32x2_f32: the A operand of
// one MFMA is A[m][k] = dZ[r0+k][o0+m] (k = lane/32, m = lane%32) and the B operand is
// B[k][n] = X[r0+k][i0+n] - both are 32 consecutive floats of one row per half-wave, i.e. plain
// coalesced global loads straight into the MFMA source registers.  No LDS, no transposes.
//   * a wave owns BO x 4 accumulator blocks (32x32 each).  The BO (4) blocks are INTERLEAVED:
//     lane j loads BO (4) consecutive floats at column BO*j (4*j), element b of that vector
//     belongs to block b.  One 8/16-byte load per lane feeds BO (4) MFMA operands, and the
//     epilogue stores 4 consecutive floats per lane (16-byte, coalesced) for free;
//   * split-K over blocks (gridDim) and over the 4 waves of a block (reduced through LDS, one
//     32x32 block at a time), per-block partial tiles to a workspace, one finalise kernel sums the
//     K-slices in a fixed order (deterministic, no atomics) and writes W.grad;
//   * every layer of the MLP is a work item of the same launch (descriptor table), so the four
//     GEMM launches + their split-K post-kernels of the library path become two launches.
// Numerics: exact fp32 products, fp32 accumulation - same class as the library kernels it
// replaces; only the summation order differs.

#include "rlg_device.hpp"

namespace rlg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kDwMaxLayers = 8;
constexpr int kDwUnroll = 4;       // k-pairs per prefetch batch

struct DwLayer {
  const float* dz;     // [rows, No]
  const float* x;      // [rows, Mi]
  float* partial;      // [ksplit][No][Mi] workspace
  float* grad;         // [No, Mi]
  int No, Mi;
  int bo;              // 1 or 2 interleaved output blocks per wave (No % bo == 0); 4 would need 512 registers
  int tiles_o, tiles_i, ksplit;
  int block_begin;     // first blockIdx.x of this layer
};

struct DwArgs {
  DwLayer layer[kDwMaxLayers];
  int num_layers;
  int rows;
};

// Bias gradients ride along in the finalise launch: out[c] = sum_b partials[b][c] over the
// per-block fp64 column sums that rlg_act_bwd_colsum left behind (one item per hidden layer).
struct ColsumItems {
  const double* partials[kDwMaxLayers];
  float* out[kDwMaxLayers];
  int nblocks[kDwMaxLayers];
  int cols[kDwMaxLayers];
  int count;
  int first_block;     // blockIdx.x of the first column-sum block (after the dW finalise blocks)
};

template <int BO> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = f32x2; };
template <> struct VecOf<4> { using type = f32x4; };

template <int BO> __device__ __forceinline__ float vec_get(const typename VecOf<BO>::type& v, int b) { return v[b]; }
template <> __device__ __forceinline__ float vec_get<1>(const float& v, int) { return v; }

template <int BO>
__device__ __forceinline__ void dw_tile(const DwLayer& L, int rows, int local_block, float* lds) {
  using VA = typename VecOf<BO>::type;
  constexpr int BI = 4;
  const int lane = lane_id();
  const int wave = wave_id();
  const int half = lane >> 5;
  const int j = lane & 31;
  const int tiles = L.tiles_o * L.tiles_i;
  const int z = local_block / tiles;
  const int t = local_block - z * tiles;
  const int to = t / L.tiles_i;
  const int ti = t - to * L.tiles_i;
  const int o0 = to * 32 * BO, i0 = ti * 32 * BI;
  const int o_col = min(o0 + BO * j, L.No - BO);      // clamp: out-of-range lanes feed rows never stored
  const int i_col = min(i0 + BI * j, L.Mi - BI);

  // rows of this block / wave (pairs of rows per MFMA)
  const int pairs_total = (rows + 1) >> 1;
  const int pairs_per_block = (pairs_total + L.ksplit - 1) / L.ksplit;
  const int pairs_per_wave = (pairs_per_block + 3) >> 2;
  const int p_begin = z * pairs_per_block + wave * pairs_per_wave;
  const int p_end = min(min(p_begin + pairs_per_wave, (z + 1) * pairs_per_block), pairs_total);

  f32x16 acc[BO][BI];
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b = 0; b < BI; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
    }
  }
  const float* pa = L.dz + o_col;
  const float* pb = L.x + i_col;
  const long long lda = L.No, ldb = L.Mi;

  int p = p_begin;
  // full batches: rows 2p+half .. are all < rows when 2*(p + U) <= rows
  const int p_full_end = min(p_end, rows >> 1);
  VA a_cur[kDwUnroll];
  f32x4 b_cur[kDwUnroll];
  auto load_batch = [&](VA (&av)[kDwUnroll], f32x4 (&bv)[kDwUnroll], int p0) {
#pragma unroll
    for (int u = 0; u < kDwUnroll; ++u) {
      const long long r = 2LL * (p0 + u) + half;
      av[u] = *reinterpret_cast<const VA*>(pa + r * lda);
      bv[u] = *reinterpret_cast<const f32x4*>(pb + r * ldb);
    }
  };
  auto mfma_batch = [&](const VA (&av)[kDwUnroll], const f32x4 (&bv)[kDwUnroll]) {
#pragma unroll
    for (int u = 0; u < kDwUnroll; ++u) {
#pragma unroll
      for (int a = 0; a < BO; ++a) {
#pragma unroll
        for (int b = 0; b < BI; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_get<BO>(av[u], a), bv[u][b], acc[a][b], 0, 0, 0);
        }
      }
    }
  };
  if (p + kDwUnroll <= p_full_end) {
    load_batch(a_cur, b_cur, p);
    p += kDwUnroll;
    while (p + kDwUnroll <= p_full_end) {
      VA a_nxt[kDwUnroll];
      f32x4 b_nxt[kDwUnroll];
      load_batch(a_nxt, b_nxt, p);          // in flight while the MFMAs of the current batch run
      mfma_batch(a_cur, b_cur);
#pragma unroll
      for (int u = 0; u < kDwUnroll; ++u) { a_cur[u] = a_nxt[u]; b_cur[u] = b_nxt[u]; }
      p += kDwUnroll;
    }
    mfma_batch(a_cur, b_cur);
  }
  // tail pairs (and the odd last row): predicated, zero-filled
  for (; p < p_end; ++p) {
    const long long r = 2LL * p + half;
    VA av;
    f32x4 bv;
    if (r < rows) {
      av = *reinterpret_cast<const VA*>(pa + r * lda);
      bv = *reinterpret_cast<const f32x4*>(pb + r * ldb);
    } else {
      av = VA(0.0f);
      bv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int a = 0; a < BO; ++a) {
#pragma unroll
      for (int b = 0; b < BI; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_get<BO>(av, a), bv[b], acc[a][b], 0, 0, 0);
    }
  }

  // ---- reduce the 4 waves' K-slices through LDS (one 32x32 block = 4 KB per wave at a time) and
  //      store: acc[a][b][q] is G[o0 + BO*row + a][i0 + 4*j + b], row = (q&3) + 8*(q>>2) + 4*half.
  float* out = L.partial + static_cast<long long>(z) * L.No * L.Mi;
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b = 0; b < BI; ++b) {
      if (wave != 0) {
        float* dst = lds + ((wave - 1) * 64 + lane) * 16;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<f32x4*>(dst + 4 * q4) =
              f32x4{acc[a][b][4 * q4], acc[a][b][4 * q4 + 1], acc[a][b][4 * q4 + 2], acc[a][b][4 * q4 + 3]};
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float* src = lds + (w * 64 + lane) * 16;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q4);
            acc[a][b][4 * q4] += v[0];
            acc[a][b][4 * q4 + 1] += v[1];
            acc[a][b][4 * q4 + 2] += v[2];
            acc[a][b][4 * q4 + 3] += v[3];
          }
        }
      }
      __syncthreads();
    }
  }
  if (wave == 0) {
    const int i = i0 + BI * j;
    if (i + BI <= L.Mi) {
#pragma unroll
      for (int a = 0; a < BO; ++a) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * half;
          const int o = o0 + BO * row + a;
          if (o < L.No) {
            const f32x4 v = {acc[a][0][q], acc[a][1][q], acc[a][2][q], acc[a][3][q]};
            *reinterpret_cast<f32x4*>(out + static_cast<long long>(o) * L.Mi + i) = v;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void mlp_dw_kernel(DwArgs args) {
  __shared__ __attribute__((aligned(16))) float lds[3 * 64 * 16];
  int l = 0;
#pragma unroll 1
  for (int k = 1; k < args.num_layers; ++k) {
    if (static_cast<int>(blockIdx.x) >= args.layer[k].block_begin) l = k;
  }
  const DwLayer& L = args.layer[l];
  const int local = blockIdx.x - L.block_begin;
  if (L.bo == 2) dw_tile<2>(L, args.rows, local, lds);
  else dw_tile<1>(L, args.rows, local, lds);
}

// grad[e] = sum_z partial[z][e].  64 float4 elements x 4 z-groups per block: group g sums the
// slices z = g, g+4, ... (4 independent loads in flight), the groups are combined through LDS in a
// fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void mlp_dw_finalize_kernel(DwArgs args, ColsumItems cs) {
  __shared__ f32x4 part[4][64];
  if (static_cast<int>(blockIdx.x) >= cs.first_block) {
    // ---- bias-gradient blocks: 32 columns x 8 row-slices per block, slices combined in order
    __shared__ double cpart[8][33];
    int b = blockIdx.x - cs.first_block;
    int item = 0;
#pragma unroll 1
    for (int k = 0; k < cs.count; ++k) {
      const int nb = (cs.cols[k] + 31) / 32;
      if (b < nb) { item = k; break; }
      b -= nb;
    }
    const int C = cs.cols[item];
    const int col = b * 32 + (threadIdx.x & 31);
    const int slice = threadIdx.x >> 5;
    double s = 0.0;
    if (col < C) {
      const double* src = cs.partials[item] + col;
      for (int r = slice; r < cs.nblocks[item]; r += 8) s += src[static_cast<long long>(r) * C];
    }
    cpart[slice][threadIdx.x & 31] = s;
    __syncthreads();
    if (slice == 0 && col < C) {
      double t = cpart[0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += cpart[k][threadIdx.x];
      cs.out[item][col] = static_cast<float>(t);
    }
    return;
  }
  int l = 0;
  int base = 0;
#pragma unroll 1
  for (int k = 0; k < args.num_layers; ++k) {
    const int n4 = (args.layer[k].No * args.layer[k].Mi) >> 2;
    const int blocks = (n4 + 63) / 64;
    if (static_cast<int>(blockIdx.x) < base + blocks) { l = k; break; }
    base += blocks;
  }
  const DwLayer& L = args.layer[l];
  const int n4 = (L.No * L.Mi) >> 2;
  const int el = threadIdx.x & 63;
  const int g = threadIdx.x >> 6;
  const int e = (blockIdx.x - base) * 64 + el;
  f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
  if (e < n4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(L.partial) + e;
    int z = g;
    for (; z + 12 < L.ksplit; z += 16) {
      const f32x4 v0 = src[static_cast<long long>(z) * n4];
      const f32x4 v1 = src[static_cast<long long>(z + 4) * n4];
      const f32x4 v2 = src[static_cast<long long>(z + 8) * n4];
      const f32x4 v3 = src[static_cast<long long>(z + 12) * n4];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; z < L.ksplit; z += 4) s += src[static_cast<long long>(z) * n4];
  }
  part[g][el] = s;
  __syncthreads();
  if (g == 0 && e < n4) {
    f32x4 t = part[0][el];
    t += part[1][el];
    t += part[2][el];
    t += part[3][el];
    reinterpret_cast<f32x4*>(L.grad)[e] = t;
  }
}

}  // namespace rlg

extern "C" {

// Plans one layer: writes {bo, tiles_o, tiles_i, ksplit} and returns the workspace floats needed,
// or -1 when the shape is not supported by the MFMA path (caller falls back to the library GEMM).
long long rlg_mlp_dw_plan(int rows, int out_features, int in_features, int target_blocks, int* plan4) {
  if (rows <= 0 || out_features <= 0 || in_features < 4 || in_features % 4 != 0) return -1;
  if ((static_cast<long long>(out_features) * in_features) % 4 != 0) return -1;
  int bo = 1;
  if (out_features % 2 == 0 && out_features > 32) bo = 2;
  const int tiles_o = (out_features + 32 * bo - 1) / (32 * bo);
  const int tiles_i = (in_features + 127) / 128;
  int ksplit = target_blocks / (tiles_o * tiles_i);
  const int max_split = (rows / 2 + 4 * rlg::kDwUnroll * 2 - 1) / (4 * rlg::kDwUnroll * 2);   // >= 2 batches per wave
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit > 128) ksplit = 128;
  if (ksplit < 1) ksplit = 1;
  plan4[0] = bo;
  plan4[1] = tiles_o;
  plan4[2] = tiles_i;
  plan4[3] = ksplit;
  return static_cast<long long>(ksplit) * out_features * in_features;
}

// All layers in one launch.  Arrays are indexed by layer; plans from rlg_mlp_dw_plan.
int rlg_mlp_dw_launch(int num_layers, const float* const* dz, const float* const* x, float* const* partial,
                      float* const* grad, const int* out_features, const int* in_features,
                      const int* plans4, int rows, int num_colsums, const double* const* colsum_partials,
                      const int* colsum_blocks, const int* colsum_cols, float* const* colsum_out,
                      void* stream) {
  using namespace rlg;
  if (num_layers <= 0 || num_layers > kDwMaxLayers || rows <= 0 || num_colsums < 0 ||
      num_colsums > kDwMaxLayers)
    return static_cast<int>(hipErrorInvalidValue);
  DwArgs args;
  args.num_layers = num_layers;
  args.rows = rows;
  int blocks = 0, fin_blocks = 0;
  for (int l = 0; l < num_layers; ++l) {
    DwLayer& L = args.layer[l];
    L.dz = dz[l];
    L.x = x[l];
    L.partial = partial[l];
    L.grad = grad[l];
    L.No = out_features[l];
    L.Mi = in_features[l];
    L.bo = plans4[4 * l + 0];
    L.tiles_o = plans4[4 * l + 1];
    L.tiles_i = plans4[4 * l + 2];
    L.ksplit = plans4[4 * l + 3];
    L.block_begin = blocks;
    if ((reinterpret_cast<uintptr_t>(L.dz) | reinterpret_cast<uintptr_t>(L.x) |
         reinterpret_cast<uintptr_t>(L.partial) | reinterpret_cast<uintptr_t>(L.grad)) % 16 != 0 ||
        L.Mi % 4 != 0 || L.No % L.bo != 0 || (L.bo != 1 && L.bo != 2))
      return static_cast<int>(hipErrorInvalidValue);
    blocks += L.tiles_o * L.tiles_i * L.ksplit;
    fin_blocks += ((L.No * L.Mi) / 4 + 63) / 64;
  }
  ColsumItems cs;
  cs.count = num_colsums;
  cs.first_block = fin_blocks;
  int cs_blocks = 0;
  for (int k = 0; k < num_colsums; ++k) {
    if (colsum_cols[k] <= 0 || colsum_blocks[k] <= 0) return static_cast<int>(hipErrorInvalidValue);
    cs.partials[k] = colsum_partials[k];
    cs.out[k] = colsum_out[k];
    cs.nblocks[k] = colsum_blocks[k];
    cs.cols[k] = colsum_cols[k];
    cs_blocks += (colsum_cols[k] + 31) / 32;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mlp_dw_kernel, dim3(blocks), dim3(256), 0, st, args);
  hipLaunchKernelGGL(mlp_dw_finalize_kernel, dim3(fin_blocks + cs_blocks), dim3(256), 0, st, args, cs);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
