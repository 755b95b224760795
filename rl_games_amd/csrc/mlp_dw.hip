// Weight-gradient GEMMs of the actor-critic MLP on the matrix cores, all layers in ONE launch (gfx950).
//
//   G_l [No, Mi] = dZ_l^T X_l = sum over the minibatch rows r of  dZ_l[r, :]^T (x) X_l[r, :]
//
// replaces autograd's `grad_weight = grad_output.t().mm(input)` of every nn.Linear built by
// A2CBuilder (rl_games/algos_torch/network_builder.py:118-147, heads :295-311).  These are the
// worst-shaped GEMMs of the path - a 32,768-long reduction into a 400x108 ... 22x100 output.
//
// A sum of outer products needs NO operand staging for v_mfma_f32_16x16x4_f32
// (D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[l&15][l>>4], B[l>>4][l&15]): with i = output
// feature, j = input feature and k = minibatch row, the A operand is dZ[r0 + (l>>4)][o0 + (l&15)] and
// the B operand X[r0 + (l>>4)][i0 + (l&15)] - plain row-major global loads straight into the MFMA
// source registers.  No LDS, no transposes.
//   * a wave owns a (16*BO) x (16*BI) output tile, BO, BI in {1, 2, 4}.  The BO (BI) blocks are
//     INTERLEAVED: lane l&15 loads BO consecutive floats at column o0 + BO*(l&15), element e of that
//     vector belongs to block e.  One 4/8/16-byte load per lane feeds BO MFMA operands, a wave load
//     reads 4 rows x 64*BO contiguous bytes, and the epilogue stores BI consecutive floats per lane;
//   * 16-wide blocks and the {64, 32, 16} tile widths keep the tile padding at 5 % (32x32 blocks: 36 %);
//   * split-K over workgroups (ordered so that a K-slice of rows is only ever touched by one XCD, which
//     takes all tiles of a slice before the next slice - the slice stays in that XCD's L2) and over the 4 waves
//     of a workgroup (combined through LDS, each wave finishing a quarter of the tile); per-slice
//     partial tiles go to a workspace, one finalise kernel sums them in a fixed order (deterministic,
//     no atomics) and writes W.grad;
//   * loads are software-pipelined in batches of 4 k-steps (16 rows): the next batch is in flight
//     while the 4*BO*BI MFMAs of the current one issue;
//   * the default form (split-bf16 products) keeps exactly this data movement: 8 k-steps (32 rows) make the
//     K = 32 of one v_mfma_f32_16x16x32_bf16 - a lane's 8 k values of a block are element (block) of its 8
//     row vectors, A and B agree on that order and a sum needs no more - split into three bf16 planes per
//     operand in registers, 6*BO*BI MFMAs per batch;
//   * every layer of the MLP is a work item of the same launch (descriptor table).
// Numerics: fp32 accumulation of products that are exact (f32 form) or exact up to 3 * 2^-24 |x||y| (split-bf16
// form, the default: six bf16 plane products per fp32 product on the 16x16x32 bf16 MFMA, see dw_split8) -
// the same class as the library kernels it replaces; only the summation order differs.

#include "rlg_device.hpp"
#include <cmath>
#include "split_bf16.hpp"
#include "split_f16.hpp"
#include "rlg_hip.h"

namespace rlg {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kDwMaxLayers = 8;
constexpr int kDwMaxTiles = 16;    // tiles per dimension of one layer (<= 1024 features in 64-wide tiles)
constexpr int kDwBatch = 4;        // k-steps (of 4 rows) per prefetch batch
#ifndef RLG_DW_SETS
#define RLG_DW_SETS 3
#endif
constexpr int kDwSets = RLG_DW_SETS;   // register sets: kDwSets-1 batches of loads in flight
#ifndef RLG_DW_PK_A
#define RLG_DW_PK_A 1
#endif
#ifndef RLG_DW_PK_B
#define RLG_DW_PK_B 1
#endif
constexpr int kDwSplitBatch = 8;       // k-steps per batch of the split-bf16 form (K = 32 of one bf16 MFMA)

// Product forms of the launch
constexpr int kDwExact = 0;     // exact f32 products on v_mfma_f32_16x16x4_f32 (the round-1 kernel: 1.3x slower than kDwBf16)
constexpr int kDwBf16 = 1;      // six exact bf16 plane products per fp32 product (split_bf16.hpp)
constexpr int kDwF16 = 2;       // three exact fp16 plane products per fp32 product, operands scaled (split_f16.hpp)

// RLG_DW_BF16=0 selects exact f32 products; RLG_DW_F16=1 the fp16 form (the layers then carry their operands' maxima)
static bool dw_split_products() {
  static const bool on = [] {
    const char* e = std::getenv("RLG_DW_BF16");
    return e == nullptr || std::atoi(e) != 0;
  }();
  return on;
}
// host-side twin of f16_scale_for
static float f16_scale_host(float amax) {
  if (!(amax > 0.0f)) return std::ldexp(1.0f, 100);
  int e;
  std::frexp(amax, &e);                       // amax in [2^(e-1), 2^e)
  int s = 13 - (e - 1);
  s = s < -100 ? -100 : (s > 100 ? 100 : s);
  return std::ldexp(1.0f, s);
}

static bool dw_f16_products() {
  static const bool on = [] {
    const char* e = std::getenv("RLG_DW_F16");
    return e == nullptr || std::atoi(e) != 0;
  }();
  return on && dw_split_products();
}

// rlg_mlp_dw_gradient_maxima: one-shot, consumed by the next rlg_mlp_dw_launch
static const float* g_dw_amax = nullptr;
static int g_dw_amax_stride = 0, g_dw_amax_dz[kDwMaxLayers], g_dw_amax_n = 0, g_dw_amax_rows = 64;
static float g_dw_scale_x[kDwMaxLayers];

struct DwLayer {
  const float* dz;     // [rows, lda]
  const float* x;      // [rows, ldb]
  float* partial;      // [ksplit][No][Mi] workspace
  float* grad;         // [No, Mi]
  long long lda, ldb;
  int No, Mi;
  int tiles_o, tiles_i, ksplit;
  int block_begin;     // first blockIdx.x of this layer
  // fp16 form: dz is scaled by the power of two that the largest entry over a wave's rows asks for - amax_dz_entries[e] =
  // the largest |dz| of rows [64 e, 64 e + 64) (left by the split-fp16 backward launch, csrc/bx_form.hpp), or nullptr:
  // then the host-side bound amax_dz; x by the fixed scale the forward splits the same tensor with (scale_x)
  const float* amax_dz_entries;
  float amax_dz, scale_x;
  // tile t of a dimension covers columns [start[t], start[t] + 16 * b[t]), b in {1, 2, 4}
  short o_start[kDwMaxTiles], i_start[kDwMaxTiles];
  signed char o_b[kDwMaxTiles], i_b[kDwMaxTiles];
};

struct DwArgs {
  DwLayer layer[kDwMaxLayers];
  int num_layers;
  int rows;
  int amax_shift;      // fp16 form: log2 of the rows one gradient-maxima entry covers (6: the 64-row kernels, 4: the lean ones)
};

// Bias gradients ride along in the finalise launch: out[c] = sum_b partials[b][c] over the
// per-block fp64 column sums that the backward kernels left behind (one item per hidden layer).
struct ColsumItems {
  const double* partials[kDwMaxLayers];
  float* out[kDwMaxLayers];
  int nblocks[kDwMaxLayers];
  int cols[kDwMaxLayers];
  int count;
  int num_blocks;      // column-sum blocks: the first blocks of the finalise launch (longest latency chain)
};

template <int B> struct DwVec;
template <> struct DwVec<1> { using type = float; };
template <> struct DwVec<2> { using type = f32x2; };
template <> struct DwVec<4> { using type = f32x4; };
template <int B> __device__ __forceinline__ float dw_get(const typename DwVec<B>::type& v, int e) { return v[e]; }
template <> __device__ __forceinline__ float dw_get<1>(const float& v, int) { return v; }
template <int B> __device__ __forceinline__ typename DwVec<B>::type dw_zero() { return typename DwVec<B>::type(0.0f); }

#define RLG_DW_PIN() __builtin_amdgcn_sched_barrier(0)

// ---- split-bf16 products: x = x0 + x1 + x2 with bf16 planes (exact: 3 x 8 significant bits); a product is
// the sum of the 6 plane products of weight >= 2^-16 - each exact in fp32 - on v_mfma_f32_16x16x32_bf16
// (12.8x the rate of the f32 MFMA, 2.1x for six of them).  The truncated terms are <= 3 * 2^-24 |x||y|, one
// fp32 rounding: against fp64 the launch is as accurate as with exact f32 products and ~10x more accurate
// than the library's fp32 GEMM (tools/exp/split_bf16_numerics.py, tools/exp/dw_bf16_check.py,
// profiles/r2_dw_bf16x6.txt).
// kSplit = false: exact f32 products (v_mfma_f32_16x16x4_f32).  kSplit = true: split-bf16 products, two
// register sets of 32 rows.
template <int BO, int BI, int kMode>
__device__ __forceinline__ void dw_tile(const DwLayer& L, int rows, int o0, int i0, int z, float* lds, int amax_shift) {
  using VA = typename DwVec<BO>::type;
  using VB = typename DwVec<BI>::type;
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int kq = lane >> 4;           // row within a k-step
  const int j = lane & 15;
  // out-of-range columns are clamped: they only feed output elements that are never stored
  const int o_col = min(o0 + BO * j, L.No - BO);
  const int i_col = min(i0 + BI * j, L.Mi - BI);
  const float* pa = L.dz + o_col;
  const float* pb = L.x + i_col;

  // k-steps (4 rows each) of this workgroup's slice, dealt to the 4 waves as contiguous runs
  const int steps_total = (rows + 3) >> 2;
  const int steps_per_block = (steps_total + L.ksplit - 1) / L.ksplit;
  const int steps_per_wave = (steps_per_block + 3) >> 2;
  const int s_begin = z * steps_per_block + wave * steps_per_wave;
  const int s_end = min(min(s_begin + steps_per_wave, (z + 1) * steps_per_block), steps_total);
  const int s_full_end = min(s_end, rows >> 2);      // steps whose 4 rows all exist

  float scale_a = 1.0f, scale_b = 1.0f;          // fp16 form: this wave's operand scales
  f32x4 acc[BO][BI];
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b = 0; b < BI; ++b) {
      acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      // f32 form: accumulators pinned to AGPRs (keeps every MFMA in place and the 114 VGPRs for the three
      // load sets).  Split form: NO AGPRs - hipcc halves the register budget of a kernel that uses any
      // (128 + 128 at two workgroups per CU), and the two 64-register load sets + planes need ~244 VGPRs.
      if constexpr (kMode == kDwExact) asm volatile("" : "+a"(acc[a][b]));
    }
  }
  // row pointers of this lane, advanced by one batch (16 rows) at a time
  const long long step_a = 4 * L.lda, step_b = 4 * L.ldb;
  const float* ra = pa + (4LL * s_begin + kq) * L.lda;
  const float* rb = pb + (4LL * s_begin + kq) * L.ldb;
  auto load_batch = [&](VA (&av)[kDwBatch], VB (&bv)[kDwBatch]) {
#pragma unroll
    for (int u = 0; u < kDwBatch; ++u) {
      av[u] = *reinterpret_cast<const VA*>(ra + u * step_a);
      bv[u] = *reinterpret_cast<const VB*>(rb + u * step_b);
    }
    ra += kDwBatch * step_a;
    rb += kDwBatch * step_b;
  };
  auto mfma_step = [&](const VA& av, const VB& bv) {
#pragma unroll
    for (int a = 0; a < BO; ++a) {
#pragma unroll
      for (int b = 0; b < BI; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(dw_get<BO>(av, a), dw_get<BI>(bv, b), acc[a][b], 0, 0, 0);
    }
  };
  if constexpr (kMode == kDwExact) {
  // Full batches: kDwSets register sets, kDwSets-1 batches of loads in flight while one issues its
  // MFMAs.  All tiles of a K-slice stream their band of rows at the same time, so every load of a
  // workgroup sees first-touch (HBM / Infinity Cache) latency even when the line counts as an L2 hit:
  // with one batch of look-ahead (~1-1.7 us of MFMAs at two waves per SIMD) the waves spent 74 % of
  // their cycles in s_waitcnt (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).  Straight-line structure with
  // statically named sets and unconditional loads, so hipcc's vmcnt bookkeeping stays exact.
  int s = s_begin;
  const int nb = (s_full_end > s_begin) ? (s_full_end - s_begin) / kDwBatch : 0;
  auto mfma_batch = [&](const VA (&av)[kDwBatch], const VB (&bv)[kDwBatch]) {
    RLG_DW_PIN();
#pragma unroll
    for (int u = 0; u < kDwBatch; ++u) mfma_step(av[u], bv[u]);
    RLG_DW_PIN();
  };
  {
    VA av[kDwSets][kDwBatch];
    VB bv[kDwSets][kDwBatch];
    // the pipelined section takes (kDwSets-1) + kDwSets*m batches (ONE way in and out: with several
    // exits hipcc stops accumulating in place and spills); the batches that do not fit are issued the
    // plain way first.  Set indices are compile-time constants after unrolling.
    const int plain = (nb >= kDwSets - 1) ? (nb - (kDwSets - 1)) % kDwSets : nb;
    const bool piped = nb - plain >= kDwSets - 1;
    if (piped) {
#pragma unroll
      for (int j = 0; j < kDwSets - 1; ++j) load_batch(av[j], bv[j]);
    }
    // (requested right behind the pipeline's first batches: one cold round trip covers them all)
#pragma unroll 1
    for (int e = 0; e < plain; ++e) {
      load_batch(av[kDwSets - 1], bv[kDwSets - 1]);
      mfma_batch(av[kDwSets - 1], bv[kDwSets - 1]);
    }
    if (piped) {
#pragma unroll 1
      for (int t = plain + kDwSets - 1; t < nb; t += kDwSets) {
#pragma unroll
        for (int j = 0; j < kDwSets; ++j) {
          load_batch(av[(j + kDwSets - 1) % kDwSets], bv[(j + kDwSets - 1) % kDwSets]);
          mfma_batch(av[j], bv[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < kDwSets - 1; ++j) mfma_batch(av[j], bv[j]);
    }
  }
  s += nb * kDwBatch;
  // tail steps (and the ragged last rows): predicated, zero-filled
  for (; s < s_end; ++s) {
    const long long r = 4LL * s + kq;
    VA av = dw_zero<BO>();
    VB bv = dw_zero<BI>();
    if (r < rows) {
      av = *reinterpret_cast<const VA*>(pa + r * L.lda);
      bv = *reinterpret_cast<const VB*>(pb + r * L.ldb);
    }
    mfma_step(av, bv);
  }
  } else {
    // split-bf16: a batch is 8 k-steps = the K = 32 of one bf16 MFMA; the lane's 8 k values of a block
    // are element (block) of its 8 row vectors - A and B agree on the row order, which is all a sum needs
    constexpr int S = 2;
    constexpr int KB = kDwSplitBatch;
    auto load8 = [&](VA (&av)[KB], VB (&bv)[KB]) {
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        av[u] = *reinterpret_cast<const VA*>(ra + u * step_a);
        bv[u] = *reinterpret_cast<const VB*>(rb + u * step_b);
      }
      ra += KB * step_a;
      rb += KB * step_b;
    };
    if constexpr (kMode == kDwF16) {
      float amax = L.amax_dz;
      if (L.amax_dz_entries != nullptr) {
        // (wave-uniform addresses: scalar loads; a wave's rows span one to three 64-row entries)
        amax = 0.0f;
        const int e0 = (4 * s_begin) >> amax_shift, e1 = (min(4 * s_end, rows) - 1) >> amax_shift;
        for (int e = e0; e <= e1; ++e) amax = __builtin_fmaxf(amax, L.amax_dz_entries[e]);
      }
      scale_a = f16_scale_for(amax);
      scale_b = L.scale_x;
    }
    auto compute8_f16 = [&](const VA (&av)[KB], const VB (&bv)[KB]) {
      RLG_DW_PIN();
      u32x4 pb[BI][2];
#pragma unroll
      for (int b = 0; b < BI; ++b) {
        float x[8];
#pragma unroll
        for (int u = 0; u < KB; ++u) x[u] = dw_get<BI>(bv[u], b);
        f16_split8(x, scale_b, pb[b]);
      }
#pragma unroll
      for (int a = 0; a < BO; ++a) {
        float x[8];
#pragma unroll
        for (int u = 0; u < KB; ++u) x[u] = dw_get<BO>(av[u], a);
        u32x4 pa[2];
        f16_split8(x, scale_a, pa);
        // small terms first; consecutive MFMAs go to different accumulators
        constexpr int kPa[3] = {1, 0, 0};
        constexpr int kPb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
          for (int b = 0; b < BI; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, pa[kPa[t]]),
                                                               __builtin_bit_cast(f16x8, pb[b][kPb[t]]),
                                                               acc[a][b], 0, 0, 0);
        }
      }
      RLG_DW_PIN();
    };
    auto compute8_bf16 = [&](const VA (&av)[KB], const VB (&bv)[KB]) {
      RLG_DW_PIN();
      // blocks are split two at a time where a lane's row vector holds two (dw_split8x2: packed residuals)
      u32x4 pb[BI][3];
      if constexpr (BI >= 2 && RLG_DW_PK_B) {
#pragma unroll
        for (int b = 0; b < BI; b += 2) {
          split_f32x2 x[8];
#pragma unroll
          for (int u = 0; u < KB; ++u) x[u] = split_f32x2{dw_get<BI>(bv[u], b), dw_get<BI>(bv[u], b + 1)};
          dw_split8x2(x, pb[b], pb[b + 1]);
        }
      } else {
#pragma unroll
        for (int b = 0; b < BI; ++b) {
          float x[8];
#pragma unroll
          for (int u = 0; u < KB; ++u) x[u] = dw_get<BI>(bv[u], b);
          dw_split8(x, pb[b]);
        }
      }
      // small terms first; consecutive MFMAs go to different accumulators
      auto products = [&](int a, const u32x4 (&pa)[3]) {
        constexpr int kPa[6] = {2, 0, 1, 1, 0, 0};
        constexpr int kPb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
          for (int b = 0; b < BI; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa[kPa[t]]),
                                                                __builtin_bit_cast(bf16x8, pb[b][kPb[t]]),
                                                                acc[a][b], 0, 0, 0);
        }
      };
      if constexpr (BO >= 2 && RLG_DW_PK_A) {
#pragma unroll
        for (int a = 0; a < BO; a += 2) {
          split_f32x2 x[8];
#pragma unroll
          for (int u = 0; u < KB; ++u) x[u] = split_f32x2{dw_get<BO>(av[u], a), dw_get<BO>(av[u], a + 1)};
          u32x4 pa0[3], pa1[3];
          dw_split8x2(x, pa0, pa1);
          products(a, pa0);
          products(a + 1, pa1);
        }
      } else {
#pragma unroll
        for (int a = 0; a < BO; ++a) {
          float x[8];
#pragma unroll
          for (int u = 0; u < KB; ++u) x[u] = dw_get<BO>(av[u], a);
          u32x4 pa[3];
          dw_split8(x, pa);
          products(a, pa);
        }
      }
      RLG_DW_PIN();
    };
    auto compute8 = [&](const VA (&av)[KB], const VB (&bv)[KB]) {
      if constexpr (kMode == kDwF16) compute8_f16(av, bv);
      else compute8_bf16(av, bv);
    };
    int s = s_begin;
    const int nb = (s_full_end > s_begin) ? (s_full_end - s_begin) / KB : 0;
    {
      VA av[S][KB];
      VB bv[S][KB];
      const int plain = (nb >= S - 1) ? (nb - (S - 1)) % S : nb;
      const bool piped = nb - plain >= S - 1;
      if (piped) {
#pragma unroll
        for (int j = 0; j < S - 1; ++j) load8(av[j], bv[j]);
      }
#pragma unroll 1
      for (int e = 0; e < plain; ++e) {
        load8(av[S - 1], bv[S - 1]);
        compute8(av[S - 1], bv[S - 1]);
      }
      if (piped) {
#pragma unroll 1
        for (int t = plain + S - 1; t < nb; t += S) {
#pragma unroll
          for (int j = 0; j < S; ++j) {
            load8(av[(j + S - 1) % S], bv[(j + S - 1) % S]);
            compute8(av[j], bv[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < S - 1; ++j) compute8(av[j], bv[j]);
      }
    }
    s += nb * KB;
    // tail batches (and the ragged last rows): predicated, zero-filled
#pragma unroll 1
    for (; s < s_end; s += KB) {
      VA av[KB];
      VB bv[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const long long r = 4LL * (s + u) + kq;
        av[u] = dw_zero<BO>();
        bv[u] = dw_zero<BI>();
        if (s + u < s_end && r < rows) {
          av[u] = *reinterpret_cast<const VA*>(pa + r * L.lda);
          bv[u] = *reinterpret_cast<const VB*>(pb + r * L.ldb);
        }
      }
      compute8(av, bv);
    }
  }

  if constexpr (kMode == kDwF16) {
    // the accumulators hold S_dz S_x times the sums: un-scale (a power of two - exact) before they meet other slices
    const float inv = 1.0f / (scale_a * scale_b);
#pragma unroll
    for (int a = 0; a < BO; ++a) {
#pragma unroll
      for (int b = 0; b < BI; ++b) acc[a][b] *= inv;
    }
  }

  // ---- combine the 4 waves' K-slices through LDS.  Fragment q = a*BI + b; wave w finishes the PER
  //      consecutive fragments [w*PER, (w+1)*PER) - consecutive b of one a, so it stores PER
  //      consecutive floats per lane.  LDS: [4 src waves][BO*BI fragments][64 lanes] float4.
  constexpr int NF = BO * BI;
  constexpr int PER = (NF >= 4) ? NF / 4 : 1;
  f32x4* slots = reinterpret_cast<f32x4*>(lds);
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b = 0; b < BI; ++b) {
      const int q = a * BI + b;
      if (q / PER != wave) slots[(wave * NF + q) * 64 + lane] = acc[a][b];
    }
  }
  __syncthreads();
  float* out = L.partial + static_cast<long long>(z) * L.No * L.Mi;
#pragma unroll
  for (int a = 0; a < BO; ++a) {
#pragma unroll
    for (int b0 = 0; b0 < BI; b0 += PER) {
      const int q0 = a * BI + b0;
      if (q0 / PER == wave) {
        f32x4 v[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          v[p] = acc[a][b0 + p];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (w != wave) v[p] += slots[(w * NF + q0 + p) * 64 + lane];
          }
        }
        // v[p][reg] = G[o0 + BO*(4*kq + reg) + a][i0 + BI*j + b0 + p]
        const int i = i0 + BI * j + b0;
        if (i + PER <= L.Mi) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int o = o0 + BO * (4 * kq + reg) + a;
            if (o < L.No) {
              float* dst = out + static_cast<long long>(o) * L.Mi + i;
              if constexpr (PER == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0][reg], v[1][reg], v[2][reg], v[3][reg]};
              } else if constexpr (PER == 2) {
                *reinterpret_cast<f32x2*>(dst) = f32x2{v[0][reg], v[1][reg]};
              } else {
                dst[0] = v[0][reg];
              }
            }
          }
        }
      }
    }
  }
}

template <int kMode>
__device__ __forceinline__ void mlp_dw_body(const DwArgs& args) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 16 * 64 * 4];     // 64 KiB
  int l = 0;
#pragma unroll 1
  for (int k = 1; k < args.num_layers; ++k) {
    if (static_cast<int>(blockIdx.x) >= args.layer[k].block_begin) l = k;
  }
  const DwLayer& L = args.layer[l];
  // Workgroup order inside a layer: id = ((z / 8) * tiles + tile) * 8 + (z % 8).  Workgroup b runs on
  // XCD b % 8, so K-slice z (a band of rows of dZ and X) is only ever touched by XCD z % 8, and that
  // XCD takes ALL tiles of one slice before the next one: the band (1.2 MB for the 200x400 layer at
  // 64 slices) stays in its 4 MB L2 while the tiles re-read it, instead of 8 bands competing for it.
  const int local = blockIdx.x - L.block_begin;
  const int tiles = L.tiles_o * L.tiles_i;
  int t, z;
  if ((L.ksplit & 7) == 0) {
    const int zlo = local & 7;
    const int q = local >> 3;
    const int zhi = q / tiles;
    t = q - zhi * tiles;
    z = zhi * 8 + zlo;
  } else {
    t = local / L.ksplit;
    z = local - t * L.ksplit;
  }
  const int to = t / L.tiles_i;
  const int ti = t - to * L.tiles_i;
  const int o0 = L.o_start[to], i0 = L.i_start[ti];
  const int bo = L.o_b[to], bi = L.i_b[ti];
#define RLG_DW_CASE(BO_, BI_) \
  if (bo == BO_ && bi == BI_) { dw_tile<BO_, BI_, kMode>(L, args.rows, o0, i0, z, lds, args.amax_shift); return; }
  RLG_DW_CASE(4, 4)
  RLG_DW_CASE(4, 2)
  RLG_DW_CASE(4, 1)
  RLG_DW_CASE(2, 4)
  RLG_DW_CASE(2, 2)
  RLG_DW_CASE(2, 1)
  RLG_DW_CASE(1, 4)
  RLG_DW_CASE(1, 2)
  RLG_DW_CASE(1, 1)
#undef RLG_DW_CASE
}

__global__ __launch_bounds__(256, 2) void mlp_dw_kernel(DwArgs args) { mlp_dw_body<kDwExact>(args); }
__global__ __launch_bounds__(256, 2) void mlp_dw_bf16x6_kernel(DwArgs args) { mlp_dw_body<kDwBf16>(args); }
__global__ __launch_bounds__(256, 2) void mlp_dw_f16x3_kernel(DwArgs args) { mlp_dw_body<kDwF16>(args); }

// The PPO loss partials ride along too (what ppo_loss_finalize_kernel does in a launch of its own,
// csrc/ppo_loss.hip): block 0 of the item folds the 7 scalar columns (losses, KL, sum of d values),
// every further block 8 of the 2A vector columns (d logstd terms, column sums of d mu) plus the
// mask-sum column it needs.  16 column slots x 16 row slices per block, 8 loads in flight per thread,
// slices combined in a fixed order.
constexpr int kLfScalars = 7;      // = kLossScalars of ppo_loss.hip (layout of a partial row)
constexpr int kLfCols = 8;         // vector columns per block
struct LossFinalizeItem {
  rlg_loss_finalize_desc d;
  int num_blocks;                  // 0: no item
};

// Returns this thread's share of sum (g * grad_scale)^2 over the gradient elements it wrote.
__device__ __forceinline__ double loss_finalize_block(const rlg_loss_finalize_desc& d, int b, double (*part)[17],
                                                      float grad_scale) {
  const int A = d.actions_num;
  const int W = kLfScalars + 2 * A;
  const int slot = threadIdx.x & 15;
  const int slice = threadIdx.x >> 4;
  // column of this slot: block 0 = the scalars; block b > 0 = vector columns 8(b-1).. and, in slot 8, the mask sum
  int c = -1;
  if (b == 0) {
    if (slot < kLfScalars) c = slot;
  } else if (slot < kLfCols) {
    c = kLfScalars + kLfCols * (b - 1) + slot;
    if (c >= W) c = -1;
  } else if (slot == kLfCols) {
    c = 5;
  }
  double s = 0.0;
  if (c >= 0) {
    const double* src = d.partials + c;
    int r = slice;
    for (; r + 7 * 16 < d.num_blocks; r += 8 * 16) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[static_cast<long long>(r + u * 16) * W];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < d.num_blocks; r += 16) s += src[static_cast<long long>(r) * W];
  }
  part[slice][slot] = s;
  __syncthreads();
  if (slice == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][slot];
    part[16][slot] = t;                   // column totals of this block
  }
  __syncthreads();
  const double msum = part[16][b == 0 ? 5 : kLfCols];
  const double denom = d.masked ? fmax(msum, 1.0) : static_cast<double>(d.minibatch);
  if (b > 0) {
    // d loss / d logstd_a = sum_i g_i (1 - z^2)  -  entropy_coef * sum_i w_i   (d ent/d logstd = 1)
    if (slice == 0 && slot < kLfCols && c >= 0) {
      const float w_total = static_cast<float>(msum / denom);
      const int a = c - kLfScalars;
      float g = 0.0f;
      if (a < A) {
        g = static_cast<float>(part[16][slot]) - d.entropy_coef * w_total;
        d.d_logstd[a] = g;
      } else if (d.d_mu_bias_or_null) {
        g = static_cast<float>(part[16][slot]);
        d.d_mu_bias_or_null[a - A] = g;                                     // bias grad of the mu head
      }
      g *= grad_scale;
      return static_cast<double>(g) * static_cast<double>(g);
    }
    return 0.0;
  }
  if (threadIdx.x == 0) {
    const float a_loss = static_cast<float>(part[16][0] / denom);
    const float c_loss = static_cast<float>(part[16][1] / denom);
    const float ent = static_cast<float>(part[16][2] / denom);
    const float b_loss = static_cast<float>(part[16][3] / denom);
    const float kl = static_cast<float>(part[16][4] / denom);
    // loss = a + 0.5*c*critic_coef - entropy*entropy_coef + b*bounds_coef   a2c_continuous.py:133
    const float loss = ((a_loss + (0.5f * c_loss) * d.critic_coef) - ent * d.entropy_coef) + b_loss * d.bounds_coef;
    d.scalars8[0] = a_loss;
    d.scalars8[1] = c_loss;
    d.scalars8[2] = ent;
    d.scalars8[3] = b_loss;
    d.scalars8[4] = kl;
    d.scalars8[5] = loss;
    d.scalars8[6] = static_cast<float>(msum);
    d.scalars8[7] = 0.0f;
    if (d.kl_slot_or_null) *d.kl_slot_or_null = kl;
    if (d.d_value_bias_or_null) {
      const float g = static_cast<float>(part[16][6]);                      // bias grad of the value head
      *d.d_value_bias_or_null = g;
      return static_cast<double>(g * grad_scale) * static_cast<double>(g * grad_scale);
    }
  }
  return 0.0;
}

// Optional by-product of the finalise launch: per-block sums of (g * grad_scale)^2 over every gradient
// element the launch writes - what grad_sumsq_kernel (csrc/optim.hip) computes in a launch of its own for
// clip_grad_norm_ (a2c_common.py:510-512).  Valid when this launch produces ALL gradients of the arena
// and nothing modifies them before the Adam launch (single GPU).  Also advances the Adam step counter.
struct NormItem {
  double* partials;          // [gridDim.x] or nullptr
  long long* step_counter;   // or nullptr
  float grad_scale;
};

// grad[e] = sum_z partial[z][e].  A block covers kFinElems consecutive float4 elements (a 256-byte
// span per slice) with kFinGroups z-groups: group g sums the slices z = g, g+16, ... (up to 4 loads
// in flight per thread), the groups are combined through LDS in a fixed order, so the result does
// not depend on scheduling.  Many small blocks: the launch is latency bound, not bandwidth bound.
constexpr int kFinElems = 16;
constexpr int kFinGroups = 16;
constexpr int kCsCols = 16;       // bias-gradient blocks: columns x row-slices of the per-block partials
constexpr int kCsSlices = 16;
struct FinWhere {         // the decoded work item of a (virtual) finalise block
  int kind;               // 0 loss item, 1 bias column sums, 2 weight-gradient elements
  int item;               // kind 1: colsum item; kind 2: layer
  int local;              // block index inside its kind / item / layer
};
__device__ __forceinline__ FinWhere fin_where(int vb, const DwArgs& args, const ColsumItems& cs, const LossFinalizeItem& lf) {
  FinWhere w;
  if (vb < lf.num_blocks) {
    w.kind = 0;
    w.item = 0;
    w.local = vb;
    return w;
  }
  int b = vb - lf.num_blocks;
  if (b < cs.num_blocks) {
    w.kind = 1;
    w.item = 0;
#pragma unroll 1
    for (int k = 0; k < cs.count; ++k) {
      const int nb = (cs.cols[k] + kCsCols - 1) / kCsCols;
      if (b < nb) { w.item = k; break; }
      b -= nb;
    }
    w.local = b;
    return w;
  }
  w.kind = 2;
  w.item = 0;
  int base = 0;
  const int fin_block = b - cs.num_blocks;
#pragma unroll 1
  for (int k = 0; k < args.num_layers; ++k) {
    const int n4 = (args.layer[k].No * args.layer[k].Mi) >> 2;
    const int blocks = (n4 + kFinElems - 1) / kFinElems;
    if (fin_block < base + blocks) { w.item = k; break; }
    base += blocks;
  }
  w.local = fin_block - base;
  return w;
}
// The work of finalise block `vb`; returns this thread's share of sum (g * grad_scale)^2 over what it wrote.
__device__ __forceinline__ double fin_vblock(int vb, const DwArgs& args, const ColsumItems& cs, const LossFinalizeItem& lf,
                                             float grad_scale) {
  __shared__ f32x4 part[kFinGroups][kFinElems];
  __shared__ double lpart[17][17];
  __shared__ double cpart[kCsSlices][kCsCols + 1];
  double sq = 0.0;                                 // this thread's share of sum (g * grad_scale)^2
  const FinWhere w = fin_where(vb, args, cs, lf);
  if (w.kind == 0) {
    sq = loss_finalize_block(lf.d, w.local, lpart, grad_scale);
  } else if (w.kind == 1) {
    // ---- bias-gradient blocks: kCsCols columns x kCsSlices row-slices per block; a thread sums the
    //      rows slice, slice + 16, ... with 8 independent loads in flight (the per-block partials of
    //      the backward launch are a few hundred rows - one dependent load per row would dominate
    //      this whole launch: 29 us -> 13 us in the update step), slices combined in a fixed order
    const int item = w.item;
    const int C = cs.cols[item];
    const int cl = threadIdx.x & (kCsCols - 1);
    const int col = w.local * kCsCols + cl;
    const int slice = threadIdx.x / kCsCols;
    double s = 0.0;
    if (col < C) {
      const double* src = cs.partials[item] + col;
      const int nblk = cs.nblocks[item];
      int r = slice;
      for (; r + 7 * kCsSlices < nblk; r += 8 * kCsSlices) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[static_cast<long long>(r + u * kCsSlices) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; r < nblk; r += kCsSlices) s += src[static_cast<long long>(r) * C];
    }
    cpart[slice][cl] = s;
    __syncthreads();
    if (slice == 0 && col < C) {
      double t = cpart[0][cl];
#pragma unroll
      for (int k = 1; k < kCsSlices; ++k) t += cpart[k][cl];
      const float gv = static_cast<float>(t);
      cs.out[item][col] = gv;
      sq = static_cast<double>(gv * grad_scale) * static_cast<double>(gv * grad_scale);
    }
  } else {
    const DwLayer& L = args.layer[w.item];
    const int n4 = (L.No * L.Mi) >> 2;
    const int el = threadIdx.x & (kFinElems - 1);
    const int g = threadIdx.x / kFinElems;
    const int e = w.local * kFinElems + el;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (e < n4) {
      const f32x4* src = reinterpret_cast<const f32x4*>(L.partial) + e;
      int z = g;
      for (; z + 3 * kFinGroups < L.ksplit; z += 4 * kFinGroups) {      // 4 independent loads in flight
        const f32x4 v0 = src[static_cast<long long>(z) * n4];
        const f32x4 v1 = src[static_cast<long long>(z + kFinGroups) * n4];
        const f32x4 v2 = src[static_cast<long long>(z + 2 * kFinGroups) * n4];
        const f32x4 v3 = src[static_cast<long long>(z + 3 * kFinGroups) * n4];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
      }
      for (; z < L.ksplit; z += kFinGroups) s += src[static_cast<long long>(z) * n4];
    }
    part[g][el] = s;
    __syncthreads();
    if (g == 0 && e < n4) {
      f32x4 t = part[0][el];
#pragma unroll
      for (int k = 1; k < kFinGroups; ++k) t += part[k][el];
      reinterpret_cast<f32x4*>(L.grad)[e] = t;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gv = t[k] * grad_scale;
        sq = fma(static_cast<double>(gv), static_cast<double>(gv), sq);
      }
    }
  }
  return sq;
}

__global__ __launch_bounds__(256) void mlp_dw_finalize_kernel(DwArgs args, ColsumItems cs, LossFinalizeItem lf,
                                                              NormItem nrm) {
  const double sq = fin_vblock(blockIdx.x, args, cs, lf, nrm.grad_scale);
  if (nrm.partials) {                                // (uniform: every thread of the block gets here)
    __shared__ double nscratch[256 / kWave];
    double one[1] = {sq};
    __syncthreads();
    block_sum<1, 256>(one, nscratch);
    if (threadIdx.x == 0) {
      nrm.partials[blockIdx.x] = one[0];
      if (blockIdx.x == 0 && nrm.step_counter) *nrm.step_counter += 1;
    }
  }
}

// Splits `width` columns into tiles of 64 / 32 / 16 (16*b, b = 4 / 2 / 1); `max_b` limits b by the
// alignment of the operand rows (a b-float vector load per lane).  Returns the tile count.
static int dw_split(int width, int max_b, short* start, signed char* b) {
  int n = 0, c = 0;
  while (c < width && n < kDwMaxTiles) {
    const int left = width - c;
    int bb = 1;
    if (left > 48) bb = 4;
    else if (left > 32) bb = (max_b >= 2) ? 2 : 1;      // 32 now, 16 next
    else if (left > 16) bb = 2;
    if (bb > max_b) bb = max_b;
    start[n] = static_cast<short>(c);
    b[n] = static_cast<signed char>(bb);
    c += 16 * bb;
    ++n;
  }
  return (c >= width) ? n : -1;
}

static int dw_max_b(const void* p, long long ld, int width) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  int b = 1;
  if (a % 8 == 0 && ld % 2 == 0 && width >= 2) b = 2;
  if (a % 16 == 0 && ld % 4 == 0 && width >= 4) b = 4;
  return b;
}

}  // namespace rlg

extern "C" {

// Plans one layer (target_blocks <= 0: the default): writes {0, tiles_o, tiles_i, ksplit} and returns the workspace floats needed,
// or -1 when the shape is not supported (caller falls back to the library GEMM).  The tile split
// itself is recomputed at launch from the operand alignment; the counts here assume 16-byte aligned
// operands with ld % 4 == 0 unless in/out features say otherwise.
long long rlg_mlp_dw_plan(int rows, int out_features, int in_features, int target_blocks, int* plan4) {
  using namespace rlg;
  if (rows <= 0 || out_features <= 0 || in_features <= 0) return -1;
  if ((static_cast<long long>(out_features) * in_features) % 4 != 0) return -1;   // finalise works on float4
  short st[kDwMaxTiles];
  signed char bb[kDwMaxTiles];
  const int mo = (out_features % 4 == 0) ? 4 : (out_features % 2 == 0 ? 2 : 1);
  const int mi = (in_features % 4 == 0) ? 4 : (in_features % 2 == 0 ? 2 : 1);
  const int tiles_o = dw_split(out_features, mo, st, bb);
  const int tiles_i = dw_split(in_features, mi, st, bb);
  if (tiles_o < 0 || tiles_i < 0) return -1;
  // k-slices: a multiple of 8 (one XCD per slice of rows) with >= 2 prefetch batches per wave
  const int steps = (rows + 3) / 4;
  const int batch = dw_split_products() ? kDwSplitBatch : kDwBatch;
  // default workgroup count (32,768 rows: 80 us at 256 / 89 us at 1,024 in the split form, 171 / 119 us with
  // f32 products, whose workgroups are 2.5x longer; a rank's 4,096-row minibatch wants <= 16 slices either way)
  // (<= 4,096 rows - one rank of 8: 8 K-slices per layer, 416 workgroups of the humanoid network = ONE round of the 512
  //  resident slots instead of 830 in 1.6 rounds: 31.1 -> 30.5 ms per rank epoch, profiles/r4_rank_shapes.txt)
  if (target_blocks <= 0) target_blocks = rows <= 4096 ? 100 : ((dw_split_products() || rows <= 8192) ? 256 : 1024);
  int ksplit = 8;
  while (ksplit * 2 <= 64 && steps / (ksplit * 2 * 4) >= 2 * batch && tiles_o * tiles_i * ksplit < target_blocks)
    ksplit *= 2;
  while (ksplit > 1 && steps / (ksplit * 4) < batch) ksplit /= 2;
  plan4[0] = 0;
  plan4[1] = tiles_o;
  plan4[2] = tiles_i;
  plan4[3] = ksplit;
  return static_cast<long long>(ksplit) * out_features * in_features;
}

// All layers in one launch.  Arrays are indexed by layer; plans from rlg_mlp_dw_plan.  dz [rows, No]
// and x [rows, Mi] are contiguous (row stride = width).
static int dw_launch_impl(int num_layers, const float* const* dz, const float* const* x, float* const* partial,
                          float* const* grad, const int* out_features, const int* in_features,
                          const int* plans4, int rows, int num_colsums, const double* const* colsum_partials,
                          const int* colsum_blocks, const int* colsum_cols, float* const* colsum_out,
                          const rlg_loss_finalize_desc* loss_finalize, double* norm_partials, float grad_scale,
                          long long* step_counter, int* finalize_blocks_out, void* stream) {
  using namespace rlg;
  if (num_layers <= 0 || num_layers > kDwMaxLayers || rows <= 0 || num_colsums < 0 ||
      num_colsums > kDwMaxLayers)
    return static_cast<int>(hipErrorInvalidValue);
  DwArgs args;
  args.num_layers = num_layers;
  args.rows = rows;
  // fp16 form: the gradients' largest magnitudes per 64 rows - left by the split-fp16 backward (rlg_mlp_dw_gradient_maxima) -
  // and the other operand's fixed scale; or, for the tools, host-side bounds from the environment; neither: the bf16 form
  const float* amax = (g_dw_amax_n == num_layers && g_dw_amax_stride >= (rows + g_dw_amax_rows - 1) / g_dw_amax_rows) ? g_dw_amax : nullptr;
  args.amax_shift = g_dw_amax_rows == 16 ? 4 : 6;
  const int amax_stride = g_dw_amax_stride;
  float scale_x[kDwMaxLayers];
  for (int l = 0; l < kDwMaxLayers; ++l) scale_x[l] = g_dw_scale_x[l];
  g_dw_amax = nullptr;
  g_dw_amax_n = 0;
  const char* ea = std::getenv("RLG_DW_F16_AMAX_DZ");
  const char* eb = std::getenv("RLG_DW_F16_AMAX_X");
  const float host_amax_dz = ea ? static_cast<float>(std::atof(ea)) : 0.0f;
  const float host_amax_x = eb ? static_cast<float>(std::atof(eb)) : 0.0f;
  const bool f16 = dw_f16_products() && (amax != nullptr || (ea != nullptr && eb != nullptr));
  int blocks = 0, fin_blocks = 0;
  for (int l = 0; l < num_layers; ++l) {
    DwLayer& L = args.layer[l];
    L.dz = dz[l];
    L.x = x[l];
    L.partial = partial[l];
    L.grad = grad[l];
    L.No = out_features[l];
    L.Mi = in_features[l];
    L.lda = L.No;
    L.ldb = L.Mi;
    L.ksplit = plans4[4 * l + 3];
    L.amax_dz_entries = amax ? amax + static_cast<long long>(g_dw_amax_dz[l]) * amax_stride : nullptr;
    L.amax_dz = host_amax_dz;
    L.scale_x = amax ? scale_x[l] : f16_scale_host(host_amax_x);
    if (L.ksplit < 1 || (reinterpret_cast<uintptr_t>(L.partial) | reinterpret_cast<uintptr_t>(L.grad)) % 16 != 0 ||
        (static_cast<long long>(L.No) * L.Mi) % 4 != 0)
      return static_cast<int>(hipErrorInvalidValue);
    L.tiles_o = dw_split(L.No, dw_max_b(L.dz, L.lda, L.No), L.o_start, L.o_b);
    L.tiles_i = dw_split(L.Mi, dw_max_b(L.x, L.ldb, L.Mi), L.i_start, L.i_b);
    if (L.tiles_o < 0 || L.tiles_i < 0) return static_cast<int>(hipErrorInvalidValue);
    L.block_begin = blocks;
    blocks += L.tiles_o * L.tiles_i * L.ksplit;
    fin_blocks += ((L.No * L.Mi) / 4 + kFinElems - 1) / kFinElems;
  }
  ColsumItems cs;
  cs.count = num_colsums;
  int cs_blocks = 0;
  for (int k = 0; k < num_colsums; ++k) {
    if (colsum_cols[k] <= 0 || colsum_blocks[k] <= 0) return static_cast<int>(hipErrorInvalidValue);
    cs.partials[k] = colsum_partials[k];
    cs.out[k] = colsum_out[k];
    cs.nblocks[k] = colsum_blocks[k];
    cs.cols[k] = colsum_cols[k];
    cs_blocks += (colsum_cols[k] + kCsCols - 1) / kCsCols;
  }
  cs.num_blocks = cs_blocks;
  LossFinalizeItem lf = {};
  if (loss_finalize) {
    lf.d = *loss_finalize;
    if (!lf.d.partials || lf.d.num_blocks <= 0 || lf.d.actions_num < 0 || !lf.d.scalars8 ||
        (lf.d.actions_num > 0 && !lf.d.d_logstd))
      return static_cast<int>(hipErrorInvalidValue);
    lf.num_blocks = 1 + (2 * lf.d.actions_num + kLfCols - 1) / kLfCols;
  }
  const int total_vb = lf.num_blocks + cs_blocks + fin_blocks;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (f16) hipLaunchKernelGGL(mlp_dw_f16x3_kernel, dim3(blocks), dim3(256), 0, st, args);
  else if (dw_split_products()) hipLaunchKernelGGL(mlp_dw_bf16x6_kernel, dim3(blocks), dim3(256), 0, st, args);
  else hipLaunchKernelGGL(mlp_dw_kernel, dim3(blocks), dim3(256), 0, st, args);
  if (finalize_blocks_out) *finalize_blocks_out = total_vb;
  NormItem nrm = {norm_partials, norm_partials ? step_counter : nullptr, grad_scale};
  hipLaunchKernelGGL(mlp_dw_finalize_kernel, dim3(total_vb), dim3(256), 0, st, args, cs, lf, nrm);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_mlp_dw_gradient_maxima(const float* entries, int stride, int rows_per_entry, const int* dz_slot, const float* x_scale,
                               int num_layers) {
  using namespace rlg;
  g_dw_amax = nullptr;
  g_dw_amax_n = 0;
  if (entries == nullptr) return 0;
  if (num_layers <= 0 || num_layers > kDwMaxLayers || stride <= 0 || (rows_per_entry != 16 && rows_per_entry != 64))
    return static_cast<int>(hipErrorInvalidValue);
  g_dw_amax_rows = rows_per_entry;
  for (int l = 0; l < num_layers; ++l) {
    if (dz_slot[l] < 0 || dz_slot[l] >= 8 || !(x_scale[l] > 0.0f)) return static_cast<int>(hipErrorInvalidValue);
    g_dw_amax_dz[l] = dz_slot[l];
    g_dw_scale_x[l] = x_scale[l];
  }
  g_dw_amax = entries;
  g_dw_amax_stride = stride;
  g_dw_amax_n = num_layers;
  return 0;
}

int rlg_mlp_dw_launch(int num_layers, const float* const* dz, const float* const* x, float* const* partial,
                      float* const* grad, const int* out_features, const int* in_features,
                      const int* plans4, int rows, int num_colsums, const double* const* colsum_partials,
                      const int* colsum_blocks, const int* colsum_cols, float* const* colsum_out,
                      const rlg_loss_finalize_desc* loss_finalize, double* norm_partials, float grad_scale,
                      long long* step_counter, int* finalize_blocks_out, void* stream) {
  return dw_launch_impl(num_layers, dz, x, partial, grad, out_features, in_features, plans4, rows, num_colsums,
                        colsum_partials, colsum_blocks, colsum_cols, colsum_out, loss_finalize, norm_partials, grad_scale,
                        step_counter, finalize_blocks_out, stream);
}

}  // extern "C"
