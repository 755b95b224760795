// The fused MLP chain on split-bf16 products (gfx950): the same vertical fusion as mlp_chain.hip - every layer of a
// 64-row tile in ONE launch, activations resident in LDS, weights streamed from L2 - but every fp32 product is the
// sum of six exact bf16 plane products on v_mfma_f32_16x16x32_bf16 (the scheme of mlp_dw.hip, split_bf16.hpp):
// 6 x 16 cycles per 16x16x32 tile instead of 8 x 32 on the f32 MFMA.
//
// The f32 MFMA kernels are bound by everything that is NOT an MFMA (VALU and MFMA never co-execute on a CDNA4 SIMD,
// profiles/r3_coexec_and_launch_probes.txt), so the split must not cost inner-loop VALU:
//   * the WEIGHTS are split once per optimizer step by rlg_mlp_chain_pack_planes into fragment order: fragment
//     (block ib, chunk c, plane p) = 64 lanes x 8 bf16 = 1 KiB, lane l holds A[16 ib + (l & 15)][k] for its 8 k slots
//     of chunk c.  A wave's A operands of a chunk are three 16-byte buffer loads at SCALAR addresses (no address VALU),
//     1 KiB contiguous each; out-of-range rows / k are zero in the fragments, so the kernels need no masks;
//   * the ACTIVATIONS are split once, by the epilogue that produces them, and live in LDS as planes: fragment
//     (chunk c, row group g, plane p) = 1 KiB, read back with one ds_read_b128 per lane.
// k slots: a chunk is 32 input features = two 16-feature blocks; lane l (q = l >> 4) takes features 4q .. 4q+3 of
// block 2c (elements 0..3) and of block 2c+1 (elements 4..7).  The MFMA output D[4q + r][row] of a 16-feature block
// is exactly elements 4*(block & 1) .. +3 of the SAME lane's B fragment for the next layer: the epilogue writes 8
// bytes per plane, no transposes (a sum does not care which k sits in which slot as long as A and B agree).
// Numerics: products exact up to 3 * 2^-24 |x||w| (the three dropped plane products), fp32 accumulation; see
// tests/test_mlp_chain_gpu.py for the bounds against fp64.
//
// Backward (this file, round 3):  d heads -> ((dZ W) * act'(H)) x L with the PPO loss tile in front, like
// mlp_chain_bwd_kernel<4, 4>; replaces the autograd dX / activation-backward / bias-sum nodes behind
// rl_games/algos_torch/network_builder.py:447-512.

#include "mlp_chain_bx.hpp"
#include "optim_common.hpp"

namespace rlg {

// direction 0: forward products of layer L (i = out, k = in); 1: backward (i = in, k = out; layer 0 needs no dX)
long long chain_bx_plane_offsets(int num_layers, const int* in_features, const int* out_features, int direction,
                                 unsigned* offsets) {
  long long total = 0;
  for (int L = 0; L < num_layers; ++L) {
    if (offsets) offsets[L] = static_cast<unsigned>(total);
    if (direction == 1 && L == 0) continue;
    const int I = direction == 0 ? out_features[L] : in_features[L];
    const int K = direction == 0 ? in_features[L] : out_features[L];
    total += static_cast<long long>(bx_nb(I)) * bx_kc(K) * kBxChunk;
  }
  return total;
}

// ------------------------------------------------------------------------------------------------
// weights -> plane fragments
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_pack_planes_kernel(PackArgs a) {
  chain_pack_planes_block(a, blockIdx.x, threadIdx.x);
}

int chain_bx_pack_blocks(const PackArgs& a) { return (a.total_pairs * 64 + 255) / 256; }
int chain_bx_pack_launch(const PackArgs& a, hipStream_t st) {
  if (a.total_pairs <= 0) return 0;
  hipLaunchKernelGGL(chain_pack_planes_kernel, dim3(chain_bx_pack_blocks(a)), dim3(256), 0, st, a);
  RLG_RETURN_LAUNCH_STATUS();
}

// direction 2: both directions into one buffer - the forward fragments at 0, the backward ones at
// chain_bx_both_offset (one launch splits the weights for the forward AND the backward launch of a training step)
long long chain_bx_both_offset(int num_layers, const int* in_features, const int* out_features) {
  const long long fwd = chain_bx_plane_offsets(num_layers, in_features, out_features, 0, nullptr);
  return (fwd + 255) & ~255LL;
}

bool chain_bx_fill_pack(PackArgs& args, int num_layers, const float* const* weights, const int* in_features,
                        const int* out_features, int direction, void* planes) {
  args.njobs = 0;
  args.total_pairs = 0;
  args.dst = static_cast<unsigned char*>(planes);
  if (num_layers < 1 || num_layers > kChainMaxLayers || direction < 0 || direction > 2 || planes == nullptr) return false;
  for (int dir = 0; dir < 2; ++dir) {
    if (direction != 2 && direction != dir) continue;
    unsigned off[kChainMaxLayers];
    const long long total = chain_bx_plane_offsets(num_layers, in_features, out_features, dir, off);
    const long long base = (direction == 2 && dir == 1) ? chain_bx_both_offset(num_layers, in_features, out_features) : 0;
    if (base + total >= static_cast<long long>(kOob)) return false;
    for (int L = (dir == 1 ? 1 : 0); L < num_layers; ++L) {
      if (args.njobs >= kChainMaxLayers) return false;          // (deep networks: one launch per direction)
      PackJob& J = args.job[args.njobs++];
      J.w = weights[L];
      J.in = in_features[L];
      J.I = dir == 0 ? out_features[L] : in_features[L];
      J.K = dir == 0 ? in_features[L] : out_features[L];
      J.KC = bx_kc(J.K);
      J.transposed = dir;
      J.pair_begin = args.total_pairs;
      J.dst_off = static_cast<unsigned>(base + off[L]);
      args.total_pairs += bx_nb(J.I) * J.KC;
    }
  }
  return args.total_pairs > 0;
}

// PACT: the activation of every hidden layer when the launch knows it (the usual network), else kChAny: per layer
template <int G, int PACT>
__global__ __launch_bounds__(64 * kBxW) void mlp_chain_bwd_bx_kernel(ChainArgs a, LossArgs loss) {
  constexpr int W = kBxW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4);
  const long long row0 = static_cast<long long>(blockIdx.x) * (16 * G);
  char* tile_a = ldsb;
  char* tile_b = ldsb + static_cast<long long>(a.lds_b_floats) * 4;

  int stamp = 0;
  chain_stamp(a.dbg, wave, stamp);                                   // tools/exp/bx_phases.py: start
  bool via_lds = false;        // d heads handed over by the loss tile in LDS
  if (a.with_loss) {
    float* handoff = a.bx_handoff_off >= 0 ? reinterpret_cast<float*>(ldsb + a.bx_handoff_off) : nullptr;
    via_lds = ppo_loss_tile<16 * G, 64 * W>(loss, lds, blockIdx.x, handoff, a.bx_handoff_ld);
    if (!via_lds) {
      // the prologue reads d heads that OTHER waves of this workgroup have just stored (see mlp_chain_bwd_kernel)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // (via LDS: the tile ends with a barrier behind its last LDS write - nothing to wait for; the stores of d mu /
    //  d values to global memory - the weight-gradient launch reads them - complete with the kernel)
  }
  chain_stamp(a.dbg, wave, stamp);                                   // loss tile done
  const rsrc_t pr = make_rsrc(a.planes, a.planes_bytes);
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);

  // H fragments of the units: two register sets, "next" is requested one unit ahead (see bx_units) - also across the
  // calls and the layers: H does not depend on the barrier between two layers
  f32x4 hval[2][G], hnext[2][G];
  auto wave_blocks = [&](int nob) -> int { return nob / W + (wave < nob % W ? 1 : 0); };
  auto wave_first = [&](int nob) -> int { return wave * (nob / W) + (wave < nob % W ? wave : nob % W); };
  // blocks ob .. ob + nf - 1 (nf <= 2) of H_{L-1}, the layer whose dZ step L produces; every global access is a
  // buffer instruction with the tile's row range as the bound: ragged tiles need no masks, out of range reads 0
  auto request_h = [&](int L, int ob, int nf) {
    const float* ph = pin_s(a.layer[L - 1].h);
    const long long ld = pin_s(a.layer[L - 1].ldh);
    const int width = pin_s(a.layer[L].in);
    const rsrc_t hr = make_rsrc(ph + row0 * ld, tile_bytes(n_rows - row0, 16 * G, ld));
    const unsigned h_lane = static_cast<unsigned>(((lane & 15) * static_cast<int>(ld) + q4) * 4);
    const unsigned h_group = static_cast<unsigned>(16 * static_cast<int>(ld) * 4);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const bool ok = !(kAbl & 4) && f < nf && (ob + f) * 16 + q4 < width;
#pragma unroll
      for (int g = 0; g < G; ++g)
        hnext[f][g] = buf_load4(hr, ok ? h_lane + static_cast<unsigned>(g) * h_group + static_cast<unsigned>(ob + f) * 64u : kOob);
    }
  };
  // (the first request goes out in front of the prologue, which covers part of its round trip)
  {
    const int nob = (pin_s(a.layer[num_layers - 1].in) + 15) >> 4;
    request_h(num_layers - 1, wave_first(nob), wave_blocks(nob) >= 2 ? 2 : wave_blocks(nob));
  }

  // ---- prologue: d heads tile -> planes in LDS -----------------------------------------------------
  {
    const int w = a.layer[num_layers - 1].out;
    const int KC0 = (w + 31) >> 5;
    const bool xv = vec4_ok(a.x, a.ldx);
    for (int u = wave; u < KC0 * G; u += W) {
      const int c = u / G;
      const int g = u - c * G;
      const long long row = row0 + g * 16 + (lane & 15);
      const int f = c * 32 + q4;
      f32x4 lo = {0.0f, 0.0f, 0.0f, 0.0f}, hi = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < n_rows) {
        if (via_lds) {
          const float* d = reinterpret_cast<const float*>(ldsb + a.bx_handoff_off) + (g * 16 + (lane & 15)) * a.bx_handoff_ld;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (f + e < w) lo[e] = d[f + e];
            if (f + 16 + e < w) hi[e] = d[f + 16 + e];
          }
        } else {
          lo = load_row4(a.x, a.ldx, row, f, w, xv);
          hi = load_row4(a.x, a.ldx, row, f + 16, w, xv);
        }
      }
      const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      u32x4 plane[3];
      dw_split8(x, plane);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(tile_a + (u * 3 + p) * kBxFrag + lane * 16) = plane[p];
    }
    __syncthreads();
  }
  chain_stamp(a.dbg, wave, stamp);                                   // prologue + barrier

  char* tin = tile_a;
  char* tout = tile_b;
  for (int L = num_layers - 1; L >= 1; --L) {
    const int l_in = pin_s(a.layer[L].in), l_out = pin_s(a.layer[L].out), p_act = pin_s(a.layer[L - 1].act);
    float* p_dz = pin_s(a.layer[L - 1].dz);
    const long long p_lddz = pin_s(a.layer[L - 1].lddz);
    const unsigned l_off = static_cast<unsigned>(pin_s(static_cast<int>(a.p_off[L])));
    const int width = l_in;                     // == layer[L-1].out
    const int KC = (l_out + 31) >> 5;
    const int NOB = (width + 15) >> 4;
    const bool keep_tile = (L - 1 >= 1);        // dZ_0 feeds nothing further down
    double* bpart = pin_s(a.layer[L - 1].bias_partials);
    if (bpart != nullptr) bpart += static_cast<long long>(blockIdx.x) * width;
    // every global access is a buffer instruction: the row range of the tile is the bound, ragged tiles need no masks
    const rsrc_t dr = make_rsrc(p_dz + row0 * p_lddz, tile_bytes(n_rows - row0, 16 * G, p_lddz));
    const unsigned d_lane = static_cast<unsigned>(((lane & 15) * static_cast<int>(p_lddz) + q4) * 4);
    const unsigned d_group = static_cast<unsigned>(16 * static_cast<int>(p_lddz) * 4);

    // dZ = acc * act'(h): fp32 to global, planes to the output tile; returns the lane's 4 feature values (rows past
    // the end are exact zeros: their d heads are, and out-of-range H reads 0)
    auto epilogue = [&](int ob, int g, const f32x4& accv, const f32x4& hval) -> f32x4 {
      const int f = ob * 16 + q4;
      f32x4 v;
      if constexpr (PACT == kChElu) {
        // h > 0 ? 1 : h + 1  ==  min(h, 0) + 1, the same bits with one VALU instruction less per element
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = accv[e] * (__builtin_fminf(hval[e], 0.0f) + 1.0f);
      } else {
        v = chain_act_grad4(accv, hval, p_act);
      }
      if (!(kAbl & 2)) buf_store4(dr, f < width ? d_lane + static_cast<unsigned>(g) * d_group + static_cast<unsigned>(ob) * 64u : kOob, v);
      if (keep_tile) {
        unsigned plane[3][2];
        split4_planes(v, plane);
        char* dst = tout + (((ob >> 1) * G + g) * 3) * kBxFrag + lane * 16 + (ob & 1) * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * kBxFrag) = make_uint2(plane[p][0], plane[p][1]);
      }
      return v;
    };
    // column sums over the workgroup's rows of one 16-feature block (fixed butterfly order: deterministic)
    auto colsum_store = [&](int ob, f32x4 s) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = row16_sum(s[e]);
      if (bpart != nullptr && (lane & 15) == 0) {
        const int f = ob * 16 + q4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (f + e < width) as_global(bpart)[f + e] = static_cast<double>(s[e]);
        }
      }
    };

    // This wave's blocks of the layer: NOB / W each, the first NOB % W waves one more (a whole extra block on some
    // waves costs less than the row-group-split remainder units of mlp_chain.hip: one 16-wide block has 6 x 4
    // MFMAs per chunk here, a unit of one row group would expose every load's latency behind 6 of them).
    // Two blocks per unit, then one.
    const int nb_w = wave_blocks(NOB), first_ob = wave_first(NOB);
    const int units2 = nb_w >> 1, left = nb_w & 1;
    // what follows a call's last unit: the single-block unit of this layer, else the next layer's first unit
    auto request_after = [&](bool after_pairs) {
      if (after_pairs && left) {
        request_h(L, first_ob + 2 * units2, 1);
      } else if (L >= 2) {
        const int nob_n = (pin_s(a.layer[L - 1].in) + 15) >> 4;
        request_h(L - 1, wave_first(nob_n), wave_blocks(nob_n) >= 2 ? 2 : wave_blocks(nob_n));
      }
    };
    auto whole = [&](auto nf_tag, int first, int nunits) {
      constexpr int NF = decltype(nf_tag)::value;
      bx_units<G, G, NF>(
          pr, l_off, KC, tin + lane * 16, nunits, [&](int j) { return first + j * NF; }, [&](int) { return 0; },
          [&](int j) {
            if (j < nunits) request_h(L, first + j * NF, NF);
            else request_after(NF == 2);
          },
          [&]() {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
              for (int g = 0; g < G; ++g) {
                hval[f][g] = hnext[f][g];
                asm volatile("" : "+v"(hval[f][g]));      // claimed here (one exact wait), not at its first use
              }
            }
          },
          [&](int j, const f32x4 (&acc)[NF][G]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const int ob = first + j * NF + f;
              f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
              for (int g = 0; g < G; ++g) s += epilogue(ob, g, acc[f][g], hval[f][g]);
              colsum_store(ob, s);
            }
          },
          true, (NF == 2 && L == 1) ? a.dbg : nullptr, wave, &stamp);
    };
    whole(std::integral_constant<int, 2>{}, first_ob, units2);
    whole(std::integral_constant<int, 1>{}, first_ob + 2 * units2, left);
    if (nb_w == 0) request_after(false);        // a wave without a block here still owes itself the next layer's first H
    chain_stamp(a.dbg, wave, stamp);                                 // per layer: units done
    // an odd number of blocks leaves half a chunk of the output tile unwritten: zero it (the weights there are zero,
    // but 0 x stale bits may be NaN)
    if (keep_tile && (NOB & 1)) {
      for (int u = wave; u < G * 3; u += W)
        *reinterpret_cast<uint2*>(tout + (((NOB >> 1) * G) * 3 + u) * kBxFrag + lane * 16 + 8) = make_uint2(0u, 0u);
    }
    __syncthreads();
    chain_stamp(a.dbg, wave, stamp);                                 // barrier
    char* t = tin;
    tin = tout;
    tout = t;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
int chain_bx_bwd_lds(ChainArgs& args, int G) {
  const int n = args.num_layers;
  // tiles: d heads (region A), then dZ_{L-1} for L = n-1 .. 2 alternating, starting with B
  long long a_chunks = bx_kc(args.layer[n - 1].out), b_chunks = 0;
  int flip = 0;
  for (int L = n - 1; L >= 2; --L, flip ^= 1) {
    const long long kc = bx_kc(args.layer[L].in);
    if (flip == 0) b_chunks = kc > b_chunks ? kc : b_chunks;
    else a_chunks = kc > a_chunks ? kc : a_chunks;
  }
  const long long a_bytes = a_chunks * G * kBxChunk, b_bytes = b_chunks * G * kBxChunk;
  const long long bytes = a_bytes + b_bytes;
  args.lds_b_floats = static_cast<int>(a_bytes / 4);
  return bytes <= 160 * 1024 ? static_cast<int>(bytes) : -1;
}

bool chain_bx_bwd_eligible(const ChainArgs& args) {
  if (args.planes == nullptr || args.num_layers < 2) return false;
  for (int L = 0; L + 1 < args.num_layers; ++L) {
    const ChainLayer& ly = args.layer[L];
    if ((reinterpret_cast<uintptr_t>(ly.h) & 15u) || (reinterpret_cast<uintptr_t>(ly.dz) & 15u) || (ly.ldh & 3) || (ly.lddz & 3) || (ly.out & 3)) return false;
    if (ly.ldh * 64 * 4 >= static_cast<long long>(kOob) || ly.lddz * 64 * 4 >= static_cast<long long>(kOob)) return false;
  }
  return true;
}

template <int G, int PACT>
static int chain_bx_launch_bwd_as(const ChainArgs& args, int lds_bytes, hipStream_t st, const LossArgs* loss, hipEvent_t ev0,
                                  hipEvent_t ev1) {
  const int grid = static_cast<int>((args.rows + 16 * G - 1) / (16 * G));
  LossArgs none = {};
  if (ev0 != nullptr)
    hipExtLaunchKernelGGL((mlp_chain_bwd_bx_kernel<G, PACT>), dim3(grid), dim3(64 * kBxW), static_cast<size_t>(lds_bytes), st, ev0,
                          ev1, 0, args, loss ? *loss : none);
  else
    hipLaunchKernelGGL((mlp_chain_bwd_bx_kernel<G, PACT>), dim3(grid), dim3(64 * kBxW), static_cast<size_t>(lds_bytes), st, args,
                       loss ? *loss : none);
  RLG_RETURN_LAUNCH_STATUS();
}
// raises the dynamic-LDS limit of the kernels once, outside any stream capture (rlg_mlp_chain_prepare)
int chain_bx_prepare() {
  static bool raised = false;
  if (raised) return 0;
  for (const void* k : {reinterpret_cast<const void*>(mlp_chain_bwd_bx_kernel<4, kChElu>),
                        reinterpret_cast<const void*>(mlp_chain_bwd_bx_kernel<4, kChAny>)}) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  raised = true;
  return 0;
}
int chain_bx_launch_bwd(const ChainArgs& args, int G, int lds_bytes, hipStream_t st, const LossArgs* loss, hipEvent_t ev0,
                        hipEvent_t ev1) {
  if (G != 4) return static_cast<int>(hipErrorInvalidValue);
  if (const int e = chain_bx_prepare()) return e;
  bool elu_only = true;       // every layer whose derivative is taken (all but the head)
  for (int L = 0; L + 1 < args.num_layers; ++L) elu_only = elu_only && args.layer[L].act == kChElu;
  return elu_only ? chain_bx_launch_bwd_as<4, kChElu>(args, lds_bytes, st, loss, ev0, ev1)
                  : chain_bx_launch_bwd_as<4, kChAny>(args, lds_bytes, st, loss, ev0, ev1);
}

// ---- optimiser step that leaves the weight planes behind (round 4) -------------------------------------------------
// adam_step_kernel (optim.hip) + chain_pack_planes_kernel as ONE launch: the thread that updates a 4 x 4 block of a
// weight matrix W [O][I] holds, afterwards, 4 consecutive k of 4 rows (forward operand A = W: one 8-byte store per
// row and plane) and 4 consecutive k of 4 columns (backward operand A = W^T: one 8-byte store per column and plane) of
// the NEW weights.  Same Adam arithmetic per element as adam_update (optim_common.hpp); the planes are the same bytes
// chain_pack_planes_kernel writes (zero padding outside the matrices is never touched: the buffer is packed once in
// full before the first step).  Everything of the arena that is not one of the chain's matrices (biases, sigma) is
// updated by the flat ranges at the end of the grid.
// (Round 4 also had a form with one thread per (row, 4 columns) - 9.0 us instead of 12.8 us - and round 4's lean kernels an
//  Adam + fp32-fragments launch of the same build, adam_frags_kernel.  Both are gone (round 5): two ranks sharing one GPU ended
//  epochs with a 16-lane group's exp_avg_sq one update apart in 20 % / 80 - 100 % of the runs, while this form (0 of 49) and
//  the plain rlg_adam_step (0 of 40) never did; the defect follows the compiled code of that kernel family, not any of its
//  parts - profiles/r5_two_rank_sync.txt.)
constexpr int kApMaxRanges = 2 * kChainMaxLayers + 2;
struct AdamPackArgs {
  AdamArgs adam;
  // per weight matrix: element offset in the arena, shape, byte offsets of its forward / backward fragments (-1: none)
  long long w_off[kChainMaxLayers];
  int O[kChainMaxLayers], I[kChainMaxLayers];
  long long fwd_off[kChainMaxLayers], bwd_off[kChainMaxLayers];
  int item_begin[kChainMaxLayers + 1];     // first 4 x 4 block of matrix L; [num] = total
  int num;
  unsigned char* planes;
  // the rest of the arena as flat ranges [begin, end)
  long long r_begin[kApMaxRanges], r_end[kApMaxRanges];
  int nranges;
  int matrix_blocks;                       // workgroups that walk 4 x 4 blocks; the flat ranges take the others
};

__device__ __forceinline__ void ap_store8(unsigned char* p, const unsigned (&w)[2]) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2*>(p) = u32x2{w[0], w[1]};
}

__global__ __launch_bounds__(256) void adam_pack_kernel(AdamPackArgs ap) {
  const AdamArgs& a = ap.adam;
  __shared__ float sh_clip;
  __shared__ float sh_norm;
  __shared__ double scratch[256 / kWave];
  // ---- everything this thread will need is requested FIRST (the gradient-norm reduction below is a chain of two memory
  //      round trips and two barriers: the loads of the block overlap with it instead of following it)
  const bool matrix_block = static_cast<int>(blockIdx.x) < ap.matrix_blocks;
  const int item = static_cast<int>(blockIdx.x) * 256 + threadIdx.x;
  const bool has_item = matrix_block && item < ap.item_begin[ap.num];
  int L = 0, O = 0, I = 0, o0 = 0, i0 = 0;
  f32x4 pn[4], g4[4], p4[4], m4[4], v4[4];
  long long idx[4] = {0, 0, 0, 0};
  if (has_item) {
    for (int j = 1; j < ap.num; ++j) L = (item >= ap.item_begin[j]) ? j : L;
    O = ap.O[L];
    I = ap.I[L];
    const int niq = I >> 2;
    const int local = item - ap.item_begin[L];
    const int oq = local / niq, iq = local - oq * niq;
    o0 = 4 * oq;
    i0 = 4 * iq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool in = o0 + r < O;
      idx[r] = ap.w_off[L] + static_cast<long long>(in ? o0 + r : o0) * I + i0;
      g4[r] = *reinterpret_cast<const f32x4*>(a.grads + idx[r]);
      p4[r] = *reinterpret_cast<const f32x4*>(a.params + idx[r]);
      m4[r] = *reinterpret_cast<const f32x4*>(a.exp_avg + idx[r]);
      v4[r] = *reinterpret_cast<const f32x4*>(a.exp_avg_sq + idx[r]);
    }
  }
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0u;
  const long long step = *a.step_counter;
  const int cur = static_cast<int>((step - 1) & 1);
  const double lr = a.lr_slots[cur];

  double sq[1] = {0.0};
  if (a.norm_partials) {           // (as adam_step_kernel: every workgroup computes the same sum in the same order)
    int b = threadIdx.x;
    for (; b + 3 * 256 < a.norm_blocks; b += 4 * 256) {
      const double v0 = a.norm_partials[b], v1 = a.norm_partials[b + 256];
      const double v2 = a.norm_partials[b + 2 * 256], v3 = a.norm_partials[b + 3 * 256];
      sq[0] += v0;
      sq[0] += v1;
      sq[0] += v2;
      sq[0] += v3;
    }
    for (; b < a.norm_blocks; b += 256) sq[0] += a.norm_partials[b];
    block_sum<1, 256>(sq, scratch);
  }
  if (threadIdx.x == 0) {
    float coef = 1.0f, total_norm = 0.0f;
    if (a.norm_partials) {
      total_norm = static_cast<float>(sqrt(sq[0]));
      coef = adam_clip_coef(a.max_norm, total_norm);
    }
    sh_clip = coef;
    sh_norm = total_norm;
  }
  __syncthreads();
  const float clip = sh_clip;
  const AdamScalars k = adam_scalars(a, step, lr);
#ifdef RLG_ADAM_TRACE
  AdamTraceAcc tr;
  if (matrix_block && has_item && !skip) {
    for (int r = 0; r < 4; ++r)
      if (o0 + r < O)
        for (int e = 0; e < 4; ++e) adam_trace_in(tr, a, step, idx[r] + e, g4[r][e], p4[r][e], m4[r][e], v4[r][e]);
  }
#endif

  if (matrix_block) {
    if (has_item && !skip) {
      f32x4 gc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g = (g4[r][e] * a.grad_scale) * clip;
          gc[r][e] = g;
          float p = p4[r][e];
          if (k.wd != 0.0f) g = g + k.wd * p;
          float m = m4[r][e];
          m = m + k.w1 * (g - m);
          float v = v4[r][e];
          v = v * k.b2 + (k.w2 * g) * g;
          const float denom = sqrt_rn(v) / k.bc2_sqrt + k.eps;
          p = p - k.step_size * (m / denom);
          m4[r][e] = m;
          v4[r][e] = v;
          p4[r][e] = p;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pn[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (o0 + r < O) {
          *reinterpret_cast<f32x4*>(a.grads + idx[r]) = gc[r];
          *reinterpret_cast<f32x4*>(a.exp_avg + idx[r]) = m4[r];
          *reinterpret_cast<f32x4*>(a.exp_avg_sq + idx[r]) = v4[r];
          *reinterpret_cast<f32x4*>(a.params + idx[r]) = p4[r];
          pn[r] = p4[r];
#ifdef RLG_ADAM_TRACE
          for (int e = 0; e < 4; ++e) adam_trace_out(tr, idx[r] + e, gc[r][e], p4[r][e], m4[r][e], v4[r][e]);
#endif
        }
      }
      // element slot of feature k inside its 32-feature chunk: lane group q = (k % 16) / 4, half = (k % 32) / 16
      if (ap.fwd_off[L] >= 0) {
        // A = W: block of 16 rows o, chunks over k = i
        const int KC = ((I + 31) >> 5);
        const int c = i0 >> 5, rr = i0 & 31, q = (rr & 15) >> 2, half = rr >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + r;
          if (o < O) {
            unsigned pl[3][2];
            split4_planes(pn[r], pl);
            unsigned char* dst = ap.planes + ap.fwd_off[L] + (static_cast<long long>(o >> 4) * KC + c) * kBxChunk +
                                 ((o & 15) + 16 * q) * 16 + half * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) ap_store8(dst + p * kBxFrag, pl[p]);
          }
        }
      }
      if (ap.bwd_off[L] >= 0) {
        // A = W^T: block of 16 rows i, chunks over k = o; the 4 consecutive k are rows o0 .. o0 + 3 (zero past O)
        const int KC = ((O + 31) >> 5);
        const int c = o0 >> 5, rr = o0 & 31, q = (rr & 15) >> 2, half = rr >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          const f32x4 col = {pn[0][e], pn[1][e], pn[2][e], pn[3][e]};
          unsigned pl[3][2];
          split4_planes(col, pl);
          unsigned char* dst = ap.planes + ap.bwd_off[L] + (static_cast<long long>(i >> 4) * KC + c) * kBxChunk +
                               ((i & 15) + 16 * q) * 16 + half * 8;
#pragma unroll
          for (int p = 0; p < 3; ++p) ap_store8(dst + p * kBxFrag, pl[p]);
        }
      }
    }
  } else if (!skip) {
    // the flat ranges: one thread per element, ranges concatenated
    long long t = (static_cast<long long>(blockIdx.x) - ap.matrix_blocks) * 256 + threadIdx.x;
    for (int r = 0; r < ap.nranges; ++r) {
      const long long len = ap.r_end[r] - ap.r_begin[r];
      if (t < len) {
#ifdef RLG_ADAM_TRACE
        adam_update_traced(tr, a, k, step, ap.r_begin[r] + t, clip);
#else
        adam_update(a, k, ap.r_begin[r] + t, clip);
#endif
        break;
      }
      t -= len;
    }
  }
#ifdef RLG_ADAM_TRACE
  adam_trace_flush(a, step, tr);
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_trace_scalars(a, step, clip, sh_norm, lr);
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_finish(a, cur, lr, skip, sh_norm, clip);
}


}  // namespace rlg

// ---------------------------------------------------------------------------------
// C ABI (declared in include/rlg_hip.h)
// ---------------------------------------------------------------------------------
extern "C" {

long long rlg_mlp_chain_planes_bytes(int num_layers, const int* in_features, const int* out_features, int direction) {
  if (num_layers < 1 || num_layers > rlg::kChainMaxLayers || direction < 0 || direction > 2) return -1;
  if (direction == 2)
    return rlg::chain_bx_both_offset(num_layers, in_features, out_features) +
           rlg::chain_bx_plane_offsets(num_layers, in_features, out_features, 1, nullptr);
  return rlg::chain_bx_plane_offsets(num_layers, in_features, out_features, direction, nullptr);
}

long long rlg_mlp_chain_planes_offset(int num_layers, const int* in_features, const int* out_features, int direction) {
  if (num_layers < 1 || num_layers > rlg::kChainMaxLayers || (direction != 0 && direction != 1)) return -1;
  return direction == 0 ? 0 : rlg::chain_bx_both_offset(num_layers, in_features, out_features);
}

int rlg_mlp_chain_pack_planes(int num_layers, const float* const* weights, const int* in_features,
                              const int* out_features, int direction, void* planes, void* stream) {
  using namespace rlg;
  if (num_layers < 1 || num_layers > kChainMaxLayers || direction < 0 || direction > 2 || planes == nullptr)
    return static_cast<int>(hipErrorInvalidValue);
  if (direction == 1 && num_layers == 1) return 0;        // a single layer has no dX chain: nothing to pack
  PackArgs args;
  if (!chain_bx_fill_pack(args, num_layers, weights, in_features, out_features, direction, planes))
    return static_cast<int>(hipErrorInvalidValue);         // (too large for 32-bit offsets / too many matrices for one launch)
  return chain_bx_pack_launch(args, static_cast<hipStream_t>(stream));
}


// rlg_adam_step that also leaves the chain's weight planes (both directions, the layout of rlg_mlp_chain_pack_planes
// direction 2) for the NEW weights - see adam_pack_kernel.  Requirements (else hipErrorInvalidValue, and the caller
// uses rlg_adam_step + rlg_mlp_chain_pack_planes): every weight matrix lies inside [params, params + n), 16-byte
// aligned relative to it, with in_features % 4 == 0; `planes` has been packed in full once (the zero padding).
int rlg_adam_step_pack(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                       const double* norm_partials_or_null, int norm_blocks, float grad_scale, float max_norm,
                       double* lr_slots, const long long* step_counter, double beta1, double beta2, double eps,
                       double weight_decay, int schedule_kind, const float* kl_or_null, float kl_scale,
                       double kl_threshold, double min_lr, double max_lr, double lr_multiplier, float* stats_out_or_null,
                       const unsigned* skip_flag_or_null, int num_layers, const float* const* weights,
                       const int* in_features, const int* out_features, void* planes, void* stream) {
  using namespace rlg;
  if (n <= 0 || !step_counter || num_layers < 1 || num_layers > kChainMaxLayers || planes == nullptr)
    return static_cast<int>(hipErrorInvalidValue);
  if (schedule_kind == 1 && !kl_or_null) return static_cast<int>(hipErrorInvalidValue);
  AdamPackArgs ap;
  AdamArgs& a = ap.adam;
  a.params = params;
  a.grads = grads;
  a.exp_avg = exp_avg;
  a.exp_avg_sq = exp_avg_sq;
  a.n = n;
  a.norm_partials = norm_partials_or_null;
  a.norm_blocks = norm_blocks;
  a.grad_scale = grad_scale;
  a.max_norm = max_norm;
  a.lr_slots = lr_slots;
  a.step_counter = step_counter;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.schedule_kind = schedule_kind;
  a.kl = kl_or_null;
  a.kl_scale = kl_scale;
  a.kl_threshold = kl_threshold;
  a.min_lr = min_lr;
  a.max_lr = max_lr;
  a.lr_multiplier = lr_multiplier;
  a.stats_out = stats_out_or_null;
  a.skip_flag = skip_flag_or_null;
  RLG_ADAM_TRACE_FILL(a);
  unsigned foff[kChainMaxLayers], boff[kChainMaxLayers];
  const long long ftotal = chain_bx_plane_offsets(num_layers, in_features, out_features, 0, foff);
  const long long btotal = chain_bx_plane_offsets(num_layers, in_features, out_features, 1, boff);
  const long long bbase = chain_bx_both_offset(num_layers, in_features, out_features);
  if (bbase + btotal >= static_cast<long long>(kOob) || ftotal >= static_cast<long long>(kOob)) return static_cast<int>(hipErrorInvalidValue);
  ap.num = num_layers;
  ap.planes = static_cast<unsigned char*>(planes);
  int items = 0;
  // matrices sorted by arena offset -> the gaps between them are the flat ranges
  long long begin[kChainMaxLayers], end[kChainMaxLayers];
  for (int L = 0; L < num_layers; ++L) {
    const long long off = weights[L] - params;
    const long long cnt = static_cast<long long>(in_features[L]) * out_features[L];
    if (off < 0 || off + cnt > n || (off & 3) != 0 || (in_features[L] & 3) != 0) return static_cast<int>(hipErrorInvalidValue);
    ap.w_off[L] = off;
    ap.O[L] = out_features[L];
    ap.I[L] = in_features[L];
    ap.fwd_off[L] = foff[L];
    ap.bwd_off[L] = (L >= 1) ? bbase + boff[L] : -1;
    ap.item_begin[L] = items;
    items += ((out_features[L] + 3) >> 2) * (in_features[L] >> 2);
    begin[L] = off;
    end[L] = off + cnt;
  }
  ap.item_begin[num_layers] = items;
  for (int x = 0; x < num_layers; ++x)            // (insertion sort of <= 8 intervals)
    for (int y = x + 1; y < num_layers; ++y)
      if (begin[y] < begin[x]) { std::swap(begin[x], begin[y]); std::swap(end[x], end[y]); }
  ap.nranges = 0;
  long long pos = 0, flat = 0;
  for (int x = 0; x <= num_layers; ++x) {
    const long long stop = (x < num_layers) ? begin[x] : n;
    if (stop < pos) return static_cast<int>(hipErrorInvalidValue);      // overlapping matrices
    if (stop > pos) {
      if (ap.nranges >= kApMaxRanges) return static_cast<int>(hipErrorInvalidValue);
      ap.r_begin[ap.nranges] = pos;
      ap.r_end[ap.nranges] = stop;
      ++ap.nranges;
      flat += stop - pos;
    }
    if (x < num_layers) pos = end[x];
  }
  ap.matrix_blocks = (items + 255) / 256;
  const int grid = ap.matrix_blocks + static_cast<int>((flat + 255) / 256);
  hipLaunchKernelGGL(adam_pack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), ap);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
