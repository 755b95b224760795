// The fused MLP chain on split-bf16 products (gfx950): the same vertical fusion as mlp_chain.hip - every layer of a
// 64-row tile in ONE launch, activations resident in LDS, weights streamed from L2 - but every fp32 product is the
// sum of six exact bf16 plane products on v_mfma_f32_16x16x32_bf16 (the scheme of mlp_dw.hip, split_bf16.hpp):
// 6 x 16 cycles per 16x16x32 tile instead of 8 x 32 on the f32 MFMA.
//
// The f32 MFMA kernels are bound by everything that is NOT an MFMA (v_mfma_f32_16x16x4_f32 and VALU instructions do not
// co-execute, profiles/r3_coexec_and_launch_probes.txt; beside the bf16 MFMA of this file one or two PLAIN VALU instructions
// per MFMA do, v_pk_add_f32 does not - profiles/r6_coexec_bf16.txt), so the split must not cost inner-loop VALU:
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

// waves per workgroup of the backward (see mlp_chain_bx_fwd.hip: 8 = two per SIMD, 256 registers each, accumulators in
// VGPRs, one block per unit)
#ifndef RLG_BX_BWD_W
#define RLG_BX_BWD_W 8
#endif
#if RLG_BX_BWD_W == 8 && !defined(RLG_ACC_CLASS)
#define RLG_ACC_CLASS "+v"
#endif

#include "mlp_chain_bx.hpp"
#include "optim_common.hpp"

#ifndef RLG_BX_TRACK
#define RLG_BX_TRACK 1          // tools: 0 = no gradient maxima, 2 = tracked but not published (timing experiments)
#endif

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
constexpr int kBwW = RLG_BX_BWD_W;
constexpr int kBwNF = (kBwW == 8) ? 1 : 2;      // blocks of the widest unit

template <int G, int PACT>
__global__ __launch_bounds__(64 * kBwW) void mlp_chain_bwd_bx_kernel(ChainArgs a, LossArgs loss) {
  constexpr int W = kBwW, NFM = kBwNF;
  static_assert(W % G == 0, "the prologue deals row group (wave % G) to a wave");
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
  f32x4 hval[NFM][G], hnext[NFM][G];
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
    for (int f = 0; f < NFM; ++f) {
      const bool ok = !(kAbl & 4) && f < nf && (ob + f) * 16 + q4 < width;
#pragma unroll
      for (int g = 0; g < G; ++g)
        hnext[f][g] = buf_load4(hr, ok ? h_lane + static_cast<unsigned>(g) * h_group + static_cast<unsigned>(ob + f) * 64u : kOob);
    }
  };
  // (the first request goes out in front of the prologue, which covers part of its round trip)
  {
    const int nob = (pin_s(a.layer[num_layers - 1].in) + 15) >> 4;
    request_h(num_layers - 1, wave_first(nob), wave_blocks(nob) >= NFM ? NFM : wave_blocks(nob));
  }

  // ---- prologue: d heads tile -> planes in LDS -----------------------------------------------------
  // scales of the rows (lane & 15 of every row group) of the tile the current step reads (fp16 form: from each row's
  // largest magnitude; 1 in the bf16 form)
  float scale_in[G];
#pragma unroll
  for (int g = 0; g < G; ++g) scale_in[g] = 1.0f;
  {
    const int w = a.layer[num_layers - 1].out;
    const int KC0 = (w + 31) >> 5;
    const bool xv = vec4_ok(a.x, a.ldx);
    float scale_mine = 1.0f;                    // of row (lane & 15) of row group wave % G: a wave splits ITS group's rows
    const int group = wave % G;
    float* row_scales = reinterpret_cast<float*>(ldsb + a.bx_scales_off);
    float* wg_max = row_scales + 16 * G + (num_layers - 1) * W;      // [layers][W]: the waves' maxima of every dZ tensor
    if (RLG_BX_F16) {
      float mine = 0.0f;
      const long long row = row0 + group * 16 + (lane & 15);
      if (row < n_rows) {
        for (int c = 0; c < KC0; ++c) {
          const int f = c * 32 + q4;
          if (via_lds) {
            const float* d = reinterpret_cast<const float*>(ldsb + a.bx_handoff_off) + (group * 16 + (lane & 15)) * a.bx_handoff_ld;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (f + e < w) mine = __builtin_fmaxf(mine, bx_finite_abs(d[f + e]));
              if (f + 16 + e < w) mine = __builtin_fmaxf(mine, bx_finite_abs(d[f + 16 + e]));
            }
          } else {
            const f32x4 lo = load_row4(a.x, a.ldx, row, f, w, xv), hi = load_row4(a.x, a.ldx, row, f + 16, w, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) mine = __builtin_fmaxf(mine, __builtin_fmaxf(bx_finite_abs(lo[e]), bx_finite_abs(hi[e])));
          }
        }
      }
      scale_mine = bx_row_scale(mine);
      if (lane < 16 && wave < G) row_scales[group * 16 + lane] = scale_mine;
      const float wmax = bx_wave_max(mine);
      if (lane == 0) wg_max[wave] = wmax;
    }
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
      u32x4 plane[kBxPlanes];
      bx_split8(x, scale_mine, plane);
#pragma unroll
      for (int p = 0; p < kBxPlanes; ++p) *reinterpret_cast<u32x4*>(tile_a + (u * kBxPlanes + p) * kBxFrag + lane * 16) = plane[p];
    }
    __syncthreads();
    if (RLG_BX_F16) {
#pragma unroll
      for (int g = 0; g < G; ++g) scale_in[g] = row_scales[g * 16 + (lane & 15)];
    }
  }
  float* const wg_max_all = reinterpret_cast<float*>(ldsb + a.bx_scales_off) + 16 * G;
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

    // fp16 form: the accumulators hold (weight scale x input-tile scale) x the sums; the tile this step writes is scaled
    // one step below the one it reads (a layer's dZ may exceed the dZ above it)
    float inv[G], scale_out[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      inv[g] = 1.0f / (kBxScaleW * scale_in[g]);
      scale_out[g] = scale_in[g] * kBxScaleStepBwd;
    }
    // dZ = acc * act'(h): fp32 to global, planes to the output tile; returns the lane's 4 feature values (rows past
    // the end are exact zeros: their d heads are, and out-of-range H reads 0)
    // largest |dZ_{L-1}| this lane produced (fp16 form; one v_max_f32 with |.| per element: a NaN is dropped by the maximum
    // and poisons everything computed from its row whatever the scales)
    float dz_max = 0.0f;
    auto epilogue = [&](int ob, int g, const f32x4& acc_scaled, const f32x4& hval) -> f32x4 {
      const int f = ob * 16 + q4;
      f32x4 v;
      f32x4 accv = acc_scaled;
      if constexpr (RLG_BX_F16) accv = acc_scaled * inv[g];
      if constexpr (PACT == kChElu) {
        // h > 0 ? 1 : h + 1  ==  min(h, 0) + 1, the same bits with one VALU instruction less per element
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = accv[e] * (__builtin_fminf(hval[e], 0.0f) + 1.0f);
      } else {
        v = chain_act_grad4(accv, hval, p_act);
      }
      if constexpr (RLG_BX_F16 && RLG_BX_TRACK != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dz_max = __builtin_fmaxf(dz_max, __builtin_fabsf(v[e]));
      }
      if (!(kAbl & 2)) buf_store4(dr, f < width ? d_lane + static_cast<unsigned>(g) * d_group + static_cast<unsigned>(ob) * 64u : kOob, v);
      if (keep_tile) {
        unsigned plane[kBxPlanes][2];
        bx_split4(v, scale_out[g], plane);
        char* dst = tout + (((ob >> 1) * G + g) * kBxPlanes) * kBxFrag + lane * 16 + (ob & 1) * 8;
#pragma unroll
        for (int p = 0; p < kBxPlanes; ++p) *reinterpret_cast<uint2*>(dst + p * kBxFrag) = make_uint2(plane[p][0], plane[p][1]);
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
    const int units2 = (NFM == 2) ? nb_w >> 1 : 0, left = nb_w - 2 * units2;
    // what follows a call's last unit: the single-block unit of this layer, else the next layer's first unit
    auto request_after = [&](bool after_pairs) {
      if (after_pairs && left) {
        request_h(L, first_ob + 2 * units2, 1);
      } else if (L >= 2) {
        const int nob_n = (pin_s(a.layer[L - 1].in) + 15) >> 4;
        request_h(L - 1, wave_first(nob_n), wave_blocks(nob_n) >= NFM ? NFM : wave_blocks(nob_n));
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
    if constexpr (NFM == 2) whole(std::integral_constant<int, 2>{}, first_ob, units2);
    whole(std::integral_constant<int, 1>{}, first_ob + 2 * units2, left);
    float* const wg_max = wg_max_all + (L - 1) * W;
    if (RLG_BX_F16 && RLG_BX_TRACK == 1 && a.amax != nullptr) {
      const float wmax = bx_wave_max(dz_max);
      if (lane == 0) wg_max[wave] = wmax;
    }
    if (RLG_BX_TRACK == 2) asm volatile("" :: "v"(dz_max));
    if (nb_w == 0) request_after(false);        // a wave without a block here still owes itself the next layer's first H
    chain_stamp(a.dbg, wave, stamp);                                 // per layer: units done
    // an odd number of blocks leaves half a chunk of the output tile unwritten: zero it (the weights there are zero,
    // but 0 x stale bits may be NaN)
    if (keep_tile && (NOB & 1)) {
      for (int u = wave; u < G * kBxPlanes; u += W)
        *reinterpret_cast<uint2*>(tout + (((NOB >> 1) * G) * kBxPlanes + u) * kBxFrag + lane * 16 + 8) = make_uint2(0u, 0u);
    }
    __syncthreads();
    chain_stamp(a.dbg, wave, stamp);                                 // barrier
#pragma unroll
    for (int g = 0; g < G; ++g) scale_in[g] = scale_out[g];
    char* t = tin;
    tin = tout;
    tout = t;
  }
  // The workgroup's gradient maxima (csrc/bx_form.hpp), stored behind the last layer's barrier: every wave has left the
  // largest |dZ_l| it produced in LDS, thread l combines layer l's.  (Stored BETWEEN the layers by a thread that kept
  // them in registers, the launch lost 7 us: 8 more live registers across the unit engine cost 260 accumulator moves and
  // turned 22 exact vmcnt waits into vmcnt(0).)
  if (RLG_BX_F16 && a.amax != nullptr && static_cast<int>(threadIdx.x) < num_layers) {
    const float* m = wg_max_all + threadIdx.x * W;
    float top = m[0];
#pragma unroll
    for (int w = 1; w < W; ++w) top = __builtin_fmaxf(top, m[w]);
    a.amax[static_cast<long long>(kBxAmaxDz + threadIdx.x) * a.amax_stride + blockIdx.x] = top;
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
  // (room for the fp16 form's row scales and the waves' gradient maxima behind the tiles: kBxBwdScratch)
  return bytes + kBxBwdScratch <= 160 * 1024 ? static_cast<int>(bytes) : -1;
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
    hipExtLaunchKernelGGL((mlp_chain_bwd_bx_kernel<G, PACT>), dim3(grid), dim3(64 * kBwW), static_cast<size_t>(lds_bytes), st, ev0,
                          ev1, 0, args, loss ? *loss : none);
  else
    hipLaunchKernelGGL((mlp_chain_bwd_bx_kernel<G, PACT>), dim3(grid), dim3(64 * kBwW), static_cast<size_t>(lds_bytes), st, args,
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

}  // namespace rlg

// ---------------------------------------------------------------------------------
// C ABI (declared in include/rlg_hip.h)
// ---------------------------------------------------------------------------------
extern "C" {

/* plane products per fp32 product of the chain's split-product kernels: 3 = two fp16 planes per operand (round 6), 6 = three
 * bf16 planes (a build with -DRLG_BX_F16=0) */
int rlg_mlp_chain_split_products(void) { return rlg::kBxProducts; }

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


}  // extern "C"
