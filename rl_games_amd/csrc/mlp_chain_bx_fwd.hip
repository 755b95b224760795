// Forward of the fused MLP chain on split-bf16 products (gfx950); scheme, fragment layouts and the unit engine:
// mlp_chain_bx.hip / mlp_chain_bx.hpp.
//
//   obs -> [RunningMeanStd normalise, statistics folded first in training] -> (Linear + act) x L -> fused (value | mu) head
//
// replaces `A2CBuilder.Network.forward` (rl_games/algos_torch/network_builder.py:447-512) with `norm_obs`
// (rl_games/algos_torch/models.py:54-56) in front of it, like mlp_chain_fwd_kernel; same interface and outputs
// (activations fp32 in global memory for the backward / weight-gradient launches, heads fp32).
//
// 64-row workgroups (G = 4), 8 waves (RLG_BX_FWD_W below), activations as planes in LDS: 12 KiB per 32-feature chunk (bf16
// form; 8 KiB in the fp16 form).  A
// 400-wide layer (156 KiB) does not fit next to its neighbours, so ONE tile of the chain may be WINDOWED: its
// producer layer p-1 and its consumer layer p run interleaved in passes over windows of the tile's chunks -
//   [units of layer p-1 for the blocks of the window]  barrier  [layer p accumulates the window's chunks]  barrier
// - with layer p's accumulators (all of a wave's blocks, at most 4) kept in registers across the passes; layer p's
// epilogue follows the last pass.  The host picks the window (chain_bx_fwd_plan).

// waves per workgroup: 8 = two per SIMD with 256 registers each - a wave's epilogue (VALU: bias, activation, the split
// into planes, stores) issues under the other wave's MFMAs - every wave with half the column blocks of a layer; 4 = one per
// SIMD with all 512 registers (rounds 3 - 6 until profiles/r6_fwd_two_waves.txt).  Two waves per SIMD need accumulators
// in VGPRs: with an AGPR constraint anywhere in the kernel hipcc splits the 256 registers 128 + 128 and spills 163 of them,
// without one it allocates one file of 256 and issues the MFMAs on VGPR accumulators.
#ifndef RLG_BX_FWD_W
#define RLG_BX_FWD_W 8
#endif
#if RLG_BX_FWD_W == 8 && !defined(RLG_ACC_CLASS)
#define RLG_ACC_CLASS "+v"
#endif
// blocks of the widest unit (a B fragment read from LDS feeds that many MFMA chains).  One at two waves per SIMD: pairs
// need 90 registers more than the 256 there are (scratch: + 5 % instead of - 11 %)
#ifndef RLG_BX_FWD_NF
#define RLG_BX_FWD_NF (RLG_BX_FWD_W == 8 ? 1 : 2)
#endif

#include "mlp_chain_bx.hpp"

namespace rlg {

constexpr int kFwG = 4;
constexpr int kFwUnitBlocks = RLG_BX_FWD_NF;
constexpr int kFwW = RLG_BX_FWD_W;
static_assert(kFwW % kFwG == 0, "the prologue deals row group (wave % G) to a wave");
constexpr int kFwMaxPersist = 16 / kFwW;   // blocks of the windowed tile's consumer per wave (accumulators across passes)

// Chunks [c0, c1) of the reduction of NF blocks (block f: ob_first + min(f, nb_valid - 1); the surplus ones repeat the
// last valid block and are dropped by the caller) for all G row groups, accumulated into acc (first: from zero).
// tile_lane: lane's base of the B tile, already shifted so that ABSOLUTE chunk indices address it.
template <int G, int NF>
__device__ __forceinline__ void bx_span(rsrc_t pr, unsigned layer_off, int KC, const char* tile_lane, int ob_first, int nb_valid,
                                        int c0, int c1, f32x4 (&acc)[NF][G], bool first) {
  const unsigned lane16 = static_cast<unsigned>(lane_id()) * 16u;
  int boff[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int ob = ob_first + (f < nb_valid ? f : nb_valid - 1);
    boff[f] = __builtin_amdgcn_readfirstlane(static_cast<int>(layer_off) + ob * KC * kBxChunk);
  }
  u32x4 a0[NF][kBxPlanes], a1[NF][kBxPlanes], b0[G][kBxPlanes], b1[G][kBxPlanes];
  auto load_a = [&](u32x4 (&av)[NF][kBxPlanes], int c) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int p = 0; p < kBxPlanes; ++p)
        av[f][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, lane16 + static_cast<unsigned>(p * kBxFrag),
                                                                                  boff[f] + c * kBxChunk, 0));
    }
  };
  auto load_b = [&](u32x4 (&bv)[G][kBxPlanes], int c) {
    const char* p = tile_lane + c * G * kBxChunk;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int pl = 0; pl < kBxPlanes; ++pl) bv[g][pl] = *reinterpret_cast<const u32x4*>(p + (g * kBxPlanes + pl) * kBxFrag);
    }
  };
  auto mfmas = [&](auto first_tag, const u32x4 (&av)[NF][kBxPlanes], const u32x4 (&bv)[G][kBxPlanes]) {
    constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int t = 0; t < kBxProducts; ++t)
          acc[f][g] = bx_mfma(av[f][kBxPa[t]], bv[g][kBxPb[t]], (kFirst && t == 0) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[f][g]);
        asm volatile("" : RLG_ACC_REG(acc[f][g]));      // accumulators live in AGPRs
      }
    }
  };
  auto step = [&](auto cur_tag, auto pf_tag, int c, auto first_tag) {
    constexpr bool kCur1 = decltype(cur_tag)::value;
    constexpr bool kPf = decltype(pf_tag)::value;
    if constexpr (kPf) {
      load_a(kCur1 ? a0 : a1, c + 1);
      load_b(kCur1 ? b0 : b1, c + 1);
    }
    mfmas(first_tag, kCur1 ? a1 : a0, kCur1 ? b1 : b0);
    if constexpr (kPf) {
#pragma unroll
      for (int i = 0; i < kBxPlanes * NF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < kBxPlanes * G; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (kBxProducts * NF * G - kBxPlanes * (NF + G) > 0)
        __builtin_amdgcn_sched_group_barrier(0x008, kBxProducts * NF * G - kBxPlanes * (NF + G), 0);
    }
    RLG_PIN();
  };
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  load_a(a0, c0);
  load_b(b0, c0);
  int c = c0;
  if (c + 1 < c1) {
    if (first) step(F, T, c, T);
    else step(F, T, c, F);
  } else {
    if (first) step(F, F, c, T);
    else step(F, F, c, F);
    return;
  }
  ++c;                                             // chunk c sits in bank 1
  for (; c + 2 < c1; c += 2) {
    step(T, T, c, F);
    step(F, T, c + 1, F);
  }
  if (c1 - c == 2) {
    step(T, T, c, F);
    step(F, F, c + 1, F);
  } else {
    step(T, F, c, F);
  }
}

template <int HACT>
__global__ __launch_bounds__(64 * kFwW) void mlp_chain_fwd_bx_kernel(ChainArgs a) {
  constexpr int G = kFwG, W = kFwW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (static_cast<int>(blockIdx.x) >= a.fwd_blocks) {      // the workgroups behind the row tiles: the backward's planes
    if (threadIdx.x < 256) chain_pack_planes_block(a.pack, static_cast<int>(blockIdx.x) - a.fwd_blocks, threadIdx.x);
    return;
  }
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4);
  const long long row0 = static_cast<long long>(blockIdx.x) * (16 * G);
  const rsrc_t pr = make_rsrc(a.planes, a.planes_bytes);
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);
  int stamp = 0;
  chain_stamp(a.dbg, wave, stamp);

  // scales of the rows (lane & 15 of every row group) of the tile the current layer reads (fp16 form; 1 in the bf16 form)
  float scale_in[G];
#pragma unroll
  for (int g = 0; g < G; ++g) scale_in[g] = kBxScaleObsNorm;
  // ---- prologue: observation tile -> planes in LDS, normalised on the way ------------------------------------------
  {
    const int in0 = a.layer[0].in;
    const int in0p = (in0 + 3) & ~3;
    const bool norm = a.rms_mean != nullptr;
    const int KC0 = (in0 + 31) >> 5;
    const int nfrag = KC0 * G;
    const bool xv = vec4_ok(a.x, a.ldx);
    const bool xnv = a.xn != nullptr && vec4_ok(a.xn, in0);
    float* stats = reinterpret_cast<float*>(ldsb + a.bx_stats_off);
    char* t0 = ldsb + a.bx_tile_off[0];
    constexpr int kProBatch = 4;        // fragments per wave at a time: every load is issued before the first is used
    f32x4 xlo[kProBatch], xhi[kProBatch];
    // fast path (rows of 4-float groups at 16-byte aligned addresses - every BASELINE shape): buffer loads over the
    // tile with 32-bit lane offsets (rows past the end and columns past the width read zero), the normaliser's
    // mean / denominator as two 16-byte LDS reads per half fragment
    const bool fast = pin_s(static_cast<int>(xv && (in0 & 3) == 0 && a.ldx * 64 * 4 < static_cast<long long>(kOob))) != 0;
    const rsrc_t xr = make_rsrc(a.x + row0 * a.ldx, fast ? tile_bytes(n_rows - row0, 16 * G, a.ldx) : 0u);
    const unsigned x_lane = static_cast<unsigned>(((lane & 15) * static_cast<int>(a.ldx) + q4) * 4);
    const unsigned x_group = static_cast<unsigned>(16 * static_cast<int>(a.ldx) * 4);
    auto load_frags = [&](int u0) {
#pragma unroll
      for (int k = 0; k < kProBatch; ++k) {
        const int u = u0 + k * W;
        const int c = u / G, g = u - c * G;
        const long long row = row0 + g * 16 + (lane & 15);
        xlo[k] = xhi[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (fast) {
          const unsigned off = x_lane + static_cast<unsigned>(g) * x_group + static_cast<unsigned>(c) * 128u;
          xlo[k] = buf_load4(xr, (u < nfrag && c * 32 + q4 < in0) ? off : kOob);
          xhi[k] = buf_load4(xr, (u < nfrag && c * 32 + 16 + q4 < in0) ? off + 64u : kOob);
        } else if (u < nfrag && row < n_rows) {
          xlo[k] = load_row4(a.x, a.ldx, row, c * 32 + q4, in0, xv);
          xhi[k] = load_row4(a.x, a.ldx, row, c * 32 + 16 + q4, in0, xv);
        }
      }
    };
    auto put_frags = [&](int u0, float scale) {
#pragma unroll
      for (int k = 0; k < kProBatch; ++k) {
        const int u = u0 + k * W;
        if (u < nfrag) {
          const int c = u / G, g = u - c * G;
          const long long row = row0 + g * 16 + (lane & 15);
          f32x4 v[2] = {xlo[k], xhi[k]};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int f = c * 32 + 16 * h + q4;
            if (fast) {
              if (norm && f < in0) {
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(stats + f);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(stats + in0p + f);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = clamp_nan((v[h][e] - m4[e]) / d4[e], -5.0f, 5.0f);
                if (row >= n_rows) v[h] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};       // (rows past the end stay zero)
              }
              if (a.xn && f < in0 && row < n_rows) store_row4(a.xn, in0, row, f, in0, v[h], xnv);
            } else if (row < n_rows) {
              if (norm) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  if (f + e < in0) v[h][e] = clamp_nan((v[h][e] - stats[f + e]) / stats[in0p + f + e], -5.0f, 5.0f);
                }
              }
              if (a.xn && f < in0) store_row4(a.xn, in0, row, f, in0, v[h], xnv);
            }
          }
          const float x[8] = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
          u32x4 plane[kBxPlanes];
          bx_split8(x, scale, plane);
#pragma unroll
          for (int p = 0; p < kBxPlanes; ++p) *reinterpret_cast<u32x4*>(t0 + (u * kBxPlanes + p) * kBxFrag + lane * 16) = plane[p];
        }
      }
    };
    load_frags(wave);
    chain_stamp(a.dbg, wave, stamp);                                 // (tools: observation loads requested)
    if (norm) {
      chain_norm_stats<W>(a, stats, in0, in0p);
      chain_stamp(a.dbg, wave, stamp);                               // (tools: statistics written)
      __syncthreads();
    }
    chain_stamp(a.dbg, wave, stamp);                                 // (tools: statistics visible)
    float scale_mine = kBxScaleObsNorm;         // of row (lane & 15) of row group wave % G: a wave splits ITS group's rows
    if (RLG_BX_F16 && !norm) {
      // raw observations have no bound: every row gets its scale from its largest magnitude (one more pass over the
      // rows, which the loads behind it find in the cache)
      float mine = 0.0f;
      const int group = wave % G;
      const long long row = row0 + group * 16 + (lane & 15);
      if (row < n_rows) {
        for (int c = 0; c < KC0; ++c) {
          const f32x4 lo = load_row4(a.x, a.ldx, row, c * 32 + q4, in0, xv), hi = load_row4(a.x, a.ldx, row, c * 32 + 16 + q4, in0, xv);
#pragma unroll
          for (int e = 0; e < 4; ++e) mine = __builtin_fmaxf(mine, __builtin_fmaxf(bx_finite_abs(lo[e]), bx_finite_abs(hi[e])));
        }
      }
      scale_mine = bx_row_scale(mine);
      if (lane < 16 && wave < G) stats[group * 16 + lane] = scale_mine;   // (the normaliser scratch is free without a normaliser)
    }
    put_frags(wave, scale_mine);
    for (int u0 = wave + W * kProBatch; u0 < nfrag; u0 += W * kProBatch) {
      load_frags(u0);
      put_frags(u0, scale_mine);
    }
    __syncthreads();
    if (RLG_BX_F16 && !norm) {
#pragma unroll
      for (int g = 0; g < G; ++g) scale_in[g] = stats[g * 16 + (lane & 15)];
      __syncthreads();                          // (the scratch shares its bytes with a later tile)
    }
  }
  chain_stamp(a.dbg, wave, stamp);                                   // prologue + barrier

  auto wave_blocks = [&](int nob) -> int { return nob / W + (wave < nob % W ? 1 : 0); };
  auto wave_first = [&](int nob) -> int { return wave * (nob / W) + (wave < nob % W ? wave : nob % W); };
  const int pass_layer = pin_s(a.bx_pass_layer), win = pin_s(a.bx_pass_chunks);
  f32x4 bval[kFwUnitBlocks], bnext[kFwUnitBlocks];

  for (int L = 0; L < num_layers; ++L) {
    const bool last = (L == num_layers - 1);
    const int l_in = pin_s(a.layer[L].in), l_out = pin_s(a.layer[L].out);
    const int KC = (l_in + 31) >> 5, NOB = (l_out + 15) >> 4;
    const unsigned l_off = static_cast<unsigned>(pin_s(static_cast<int>(a.p_off[L])));
    const char* tin = ldsb + pin_s(a.bx_tile_off[L]);
    char* tout = last ? nullptr : ldsb + pin_s(a.bx_tile_off[L + 1]);

    // bias + activation of a fragment of layer `layer`: fp32 to global memory (training / heads), planes to `dst_tile`
    // at chunk (ob >> 1) - chunk_base.  Padded features come out as act(0 + 0) = 0.
    auto make_epilogue = [&](int layer, char* dst_tile, int chunk_base, const float (&in_scale)[G]) {
      const int width = pin_s(a.layer[layer].out), act = pin_s(a.layer[layer].act);
      float* ph = pin_s(a.layer[layer].h);
      const long long ld = pin_s(a.layer[layer].ldh);
      const bool h_on = ph != nullptr;
      const rsrc_t hr = make_rsrc(h_on ? ph + row0 * ld : nullptr, h_on ? tile_bytes(n_rows - row0, 16 * G, ld) : 0u);
      const unsigned h_lane = static_cast<unsigned>(((lane & 15) * static_cast<int>(ld) + q4) * 4);
      const unsigned h_group = static_cast<unsigned>(16 * static_cast<int>(ld) * 4);
      // (rows of 4-float groups at 16-byte aligned addresses: one store per fragment; else - the 22-wide head - four)
      const bool h_fast = pin_s(static_cast<int>(h_on && vec4_ok(ph, ld) && (width & 3) == 0)) != 0;
      // the accumulators hold (weight scale x input-tile scale) x the sums: un-scaled by a power of two on the way to the bias
      float inv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) inv[g] = 1.0f / (kBxScaleW * in_scale[g]);
      return [=](int ob, int g, const f32x4& accv, const f32x4& bias) {
        const int f = ob * 16 + q4;
        f32x4 z;
        if constexpr (RLG_BX_F16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = __builtin_fmaf(accv[e], inv[g], bias[e]);
        } else {
          z = accv + bias;
        }
        const f32x4 v = chain_act4<HACT>(z, act);
        if (h_on) {
          const unsigned off = h_lane + static_cast<unsigned>(g) * h_group + static_cast<unsigned>(ob) * 64u;
          if (h_fast) {
            buf_store4(hr, f < width ? off : kOob, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) buf_store1(hr, f + e < width ? off + 4u * e : kOob, v[e]);
          }
        }
        if (dst_tile != nullptr) {
          unsigned plane[kBxPlanes][2];
          bx_split4(v, kBxScaleH, plane);
          char* dst = dst_tile + ((((ob >> 1) - chunk_base) * G + g) * kBxPlanes) * kBxFrag + lane * 16 + (ob & 1) * 8;
#pragma unroll
          for (int p = 0; p < kBxPlanes; ++p) *reinterpret_cast<uint2*>(dst + p * kBxFrag) = make_uint2(plane[p][0], plane[p][1]);
        }
      };
    };
    auto bias_rsrc = [&](int layer) -> rsrc_t {
      return make_rsrc(pin_s(a.layer[layer].bias), static_cast<unsigned>(pin_s(a.layer[layer].out)) * 4u);
    };
    // the 4 bias values of a lane's features fo .. fo+3 of a layer `width` wide (zero beyond it: the resource's bound)
    auto load_bias = [&](rsrc_t br, int fo, bool fast) -> f32x4 {
      if (fast) return buf_load4(br, static_cast<unsigned>(fo) * 4u);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = buf_load1(br, static_cast<unsigned>(fo + e) * 4u);
      return v;
    };
    auto bias_fast = [&](int layer) -> bool {
      return pin_s(static_cast<int>(aligned16(a.layer[layer].bias) && (a.layer[layer].out & 3) == 0)) != 0;
    };
    // blocks [b0, b1) of layer L for all row groups: this wave's share, four blocks per unit, then two, then one
    auto run_blocks = [&](int b0, int b1, char* dst_tile, int chunk_base) {
      const int nb = b1 - b0;
      const int nb_w = wave_blocks(nb), first_ob = b0 + wave_first(nb);
      // widest units first: kFwUnitBlocks blocks each, then (unless the widest is 3: pairs would be a third instantiation of
      // the engine and its registers) pairs, then single blocks
      const int unitsw = (kFwUnitBlocks > 2) ? nb_w / kFwUnitBlocks : 0;
      const int rest = nb_w - kFwUnitBlocks * unitsw;
      const int units2 = (kFwUnitBlocks == 3 || kFwUnitBlocks == 1) ? 0 : rest >> 1, left = rest - 2 * units2;
      const rsrc_t br = bias_rsrc(L);
      const bool bfast = bias_fast(L);
      auto epilogue = make_epilogue(L, dst_tile, chunk_base, scale_in);
      auto whole = [&](auto nf_tag, int first, int nunits) {
        constexpr int NF = decltype(nf_tag)::value;
        bx_units<G, G, NF>(
            pr, l_off, KC, tin + lane * 16, nunits, [&](int j) { return first + j * NF; }, [&](int) { return 0; },
            [&](int j) {
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                const int fo = (first + (j < nunits ? j : nunits - 1) * NF + f) * 16 + q4;
                bnext[f] = load_bias(br, fo, bfast);
              }
            },
            [&]() {
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                bval[f] = bnext[f];
                asm volatile("" : "+v"(bval[f]));
              }
            },
            [&](int j, const f32x4 (&acc)[NF][G]) {
#pragma unroll
              for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int g = 0; g < G; ++g) epilogue(first + j * NF + f, g, acc[f][g], bval[f]);
              }
            },
            false);
      };
      if constexpr (kFwUnitBlocks > 2) whole(std::integral_constant<int, kFwUnitBlocks>{}, first_ob, unitsw);
      if constexpr (kFwUnitBlocks != 3 && kFwUnitBlocks != 1)
        whole(std::integral_constant<int, 2>{}, first_ob + kFwUnitBlocks * unitsw, units2);
      whole(std::integral_constant<int, 1>{}, first_ob + kFwUnitBlocks * unitsw + 2 * units2, left);
    };
    // an odd number of blocks leaves half a chunk of a tile unwritten: zero it (the weights there are zero, but
    // 0 x stale bits may be NaN)
    auto zero_pad = [&](char* tile, int nob, int chunk_base) {
      if (tile != nullptr && (nob & 1)) {
        for (int u = wave; u < G * kBxPlanes; u += W)
          *reinterpret_cast<uint2*>(tile + ((((nob >> 1) - chunk_base) * G) * kBxPlanes + u) * kBxFrag + lane * 16 + 8) = make_uint2(0u, 0u);
      }
    };

    if (L + 1 != pass_layer) {
      run_blocks(0, NOB, tout, 0);
      zero_pad(tout, NOB, 0);
      chain_stamp(a.dbg, wave, stamp);                               // per layer: units done
      __syncthreads();
      chain_stamp(a.dbg, wave, stamp);                               // barrier
#pragma unroll
      for (int g = 0; g < G; ++g) scale_in[g] = kBxScaleH;
      continue;
    }

    // ---- layer L produces the windowed tile, layer P = L + 1 consumes it ----------------------------------------
    const int P = L + 1;
    const int p_in = pin_s(a.layer[P].in), p_out = pin_s(a.layer[P].out);
    const int KCP = (p_in + 31) >> 5, NOBP = (p_out + 15) >> 4;
    const unsigned p_off = static_cast<unsigned>(pin_s(static_cast<int>(a.p_off[P])));
    const int pnb = wave_blocks(NOBP), pfirst = wave_first(NOBP);
    const bool p_last = (P == num_layers - 1);
    char* ptile = p_last ? nullptr : ldsb + pin_s(a.bx_tile_off[P + 1]);
    f32x4 pacc[kFwMaxPersist][G];
    for (int c0 = 0; c0 < KCP; c0 += win) {
      const int c1 = (c0 + win < KCP) ? c0 + win : KCP;
      const int b0 = 2 * c0, b1 = (2 * c1 < NOB) ? 2 * c1 : NOB;
      run_blocks(b0, b1, tout, c0);
      if (b1 == NOB) zero_pad(tout, NOB, c0);
      __syncthreads();
      if (pnb > 0) bx_span<G, kFwMaxPersist>(pr, p_off, KCP, tout + lane * 16 - c0 * G * kBxChunk, pfirst, pnb, c0, c1, pacc, c0 == 0);
      __syncthreads();
    }
    chain_stamp(a.dbg, wave, stamp);                                 // passes done
    if (pnb > 0) {
      const rsrc_t br = bias_rsrc(P);
      const bool bfast = bias_fast(P);
      float hidden_scale[G];
#pragma unroll
      for (int g = 0; g < G; ++g) hidden_scale[g] = kBxScaleH;
      auto epilogue = make_epilogue(P, ptile, 0, hidden_scale);
      f32x4 pb[kFwMaxPersist];
#pragma unroll
      for (int f = 0; f < kFwMaxPersist; ++f) {
        const int fo = (pfirst + f) * 16 + q4;
        pb[f] = load_bias(br, f < pnb ? fo : p_out, bfast);
      }
      asm volatile("s_nop 7" : RLG_ACC_REG(pacc[0][0]));
#pragma unroll
      for (int f = 0; f < kFwMaxPersist; ++f) {
        if (f < pnb) {
#pragma unroll
          for (int g = 0; g < G; ++g) epilogue(pfirst + f, g, pacc[f][g], pb[f]);
        }
      }
    }
    zero_pad(ptile, NOBP, 0);
    chain_stamp(a.dbg, wave, stamp);                                 // consumer epilogue
    __syncthreads();
    chain_stamp(a.dbg, wave, stamp);                                 // barrier
#pragma unroll
    for (int g = 0; g < G; ++g) scale_in[g] = kBxScaleH;
    ++L;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
// Tiles: T_L = the input of layer L (T_0: the observations).  T_L is written during layer L-1 and read during layer L,
// so only CONSECUTIVE tiles are alive together: even tiles sit at the bottom of the LDS, odd tiles at the top, and a pair
// fits when the two sizes add up to the budget at most.  At most one tile (T_p, p >= 1) is windowed when a pair does not
// fit: the window gets what T_{p-1} leaves free (T_{p+1}, written behind the last pass, may overlap both).
int chain_bx_fwd_plan(ChainArgs& args) {
  const int n = args.num_layers;
  long long kc[kChainMaxLayers];
  for (int L = 0; L < n; ++L) kc[L] = bx_kc(args.layer[L].in);
  const long long chunk = kFwG * kBxChunk;
  long long stats = ((2LL * ((args.layer[0].in + 3) & ~3) * 4) + 15) & ~15LL;     // prologue only: shares the top
  if (stats < 16 * kFwG * 4) stats = 16 * kFwG * 4;                                // (... or the row scales of raw observations)
  // the smallest LDS that holds every consecutive pair (small networks then run several workgroups per CU), else all of it
  long long need = kc[0] * chunk + stats;
  for (int L = 0; L < n; ++L) {
    const long long pair = (kc[L] + (L + 1 < n ? kc[L + 1] : 0)) * chunk;
    need = pair > need ? pair : need;
  }
  const long long budget = need <= 160 * 1024 ? need : 160 * 1024;
  auto place = [&](int p, long long window) -> bool {
    for (int L = 0; L < n; ++L) {
      const long long c = (L == p) ? window : kc[L];
      if (c * chunk > budget) return false;
      args.bx_tile_off[L] = static_cast<int>((L & 1) ? budget - c * chunk : 0);
    }
    args.bx_tile_off[n] = 0;
    for (int L = 0; L + 1 < n; ++L) {
      if (L == p) continue;        // T_{p+1} is written behind the last pass, when the window is dead: they may overlap
      const long long c0 = kc[L], c1 = (L + 1 == p) ? window : kc[L + 1];
      if ((c0 + c1) * chunk > budget) return false;
    }
    // the normaliser scratch lives during the prologue, next to T_0 only
    if (kc[0] * chunk + stats > budget) return false;
    args.bx_stats_off = static_cast<int>(budget - stats);
    return true;
  };
  args.bx_pass_layer = -1;
  args.bx_pass_chunks = 0;
  if (place(-1, 0)) return static_cast<int>(budget);
  // window the largest tile
  int p = -1;
  for (int L = 1; L < n; ++L) {
    if (p < 0 || kc[L] > kc[p]) p = L;
  }
  if (p < 1) return -1;
  if (bx_nb(args.layer[p].out) > kFwMaxPersist * kFwW) return -1;     // the consumer's accumulators stay in registers
  for (long long passes = 2; passes <= kc[p]; ++passes) {
    const long long window = (kc[p] + passes - 1) / passes;
    if (place(p, window)) {
      args.bx_pass_layer = p;
      args.bx_pass_chunks = static_cast<int>(window);
      return static_cast<int>(budget);
    }
  }
  return -1;
}

bool chain_bx_fwd_eligible(const ChainArgs& args) {
  if (args.planes == nullptr) return false;
  for (int L = 0; L < args.num_layers; ++L) {
    const ChainLayer& ly = args.layer[L];
    if (reinterpret_cast<uintptr_t>(ly.bias) & 3u) return false;
    if (ly.h != nullptr && ly.ldh * 64 * 4 >= static_cast<long long>(kOob)) return false;
  }
  return true;
}

template <int HACT>
static int chain_bx_launch_fwd_as(const ChainArgs& args_in, int lds_bytes, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  ChainArgs args = args_in;
  int grid = static_cast<int>((args.rows + 16 * kFwG - 1) / (16 * kFwG));
  args.fwd_blocks = grid;
  if (args.pack.total_pairs > 0) grid += chain_bx_pack_blocks(args.pack);
  if (ev0 != nullptr)
    hipExtLaunchKernelGGL((mlp_chain_fwd_bx_kernel<HACT>), dim3(grid), dim3(64 * kFwW), static_cast<size_t>(lds_bytes), st, ev0,
                          ev1, 0, args);
  else
    hipLaunchKernelGGL((mlp_chain_fwd_bx_kernel<HACT>), dim3(grid), dim3(64 * kFwW), static_cast<size_t>(lds_bytes), st, args);
  RLG_RETURN_LAUNCH_STATUS();
}
int chain_bx_fwd_prepare() {
  static bool raised = false;
  if (raised) return 0;
  for (const void* k : {reinterpret_cast<const void*>(mlp_chain_fwd_bx_kernel<kChElu>),
                        reinterpret_cast<const void*>(mlp_chain_fwd_bx_kernel<kChAny>)}) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  raised = true;
  return 0;
}
int chain_bx_launch_fwd(const ChainArgs& args, int lds_bytes, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  if (const int e = chain_bx_fwd_prepare()) return e;
  bool elu_only = true;
  for (int L = 0; L < args.num_layers; ++L)
    elu_only = elu_only && (args.layer[L].act == kChElu || args.layer[L].act == kChIdentity);
  return elu_only ? chain_bx_launch_fwd_as<kChElu>(args, lds_bytes, st, ev0, ev1)
                  : chain_bx_launch_fwd_as<kChAny>(args, lds_bytes, st, ev0, ev1);
}

}  // namespace rlg
