// The actor-critic MLP as a vertically fused chain on f32 MFMA (gfx950): every layer of a row tile
// in ONE launch, activations resident in LDS, weights streamed from L2.
//
//   forward :  obs -> [RunningMeanStd normalise] -> (Linear + act) x L -> fused (value | mu) head
//   backward:  d heads -> ((dZ W) * act'(H)) x L   (+ per-layer bias-gradient partial sums)
//
// replaces `A2CBuilder.Network.forward` (rl_games/algos_torch/network_builder.py:447-512: actor_mlp,
// value / mu heads), `norm_obs` (rl_games/algos_torch/models.py:54-56) in front of it, and the
// autograd dX / activation-backward / bias-sum nodes behind it.  The weight gradients are the
// separate launch of mlp_dw16.hip, which reads the dZ / H arrays this file writes.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies
// A[l&15][l>>4], B[l>>4][l&15] and receives D[4*(l>>4)+reg][l&15], reg = 0..3):
//   * every product is computed TRANSPOSED: i = output feature, j = batch row, k = input feature.
//     The weights are the A operand, the activations the B operand, and the result comes out with
//     "lane = (row, 4 consecutive features)" - exactly the shape the next layer needs as ITS B
//     operand.  A layer's output therefore goes to LDS with one ds_write_b128 per lane and comes
//     back with one ds_read_b128 per lane, no transposes, no bank conflicts (lane-consecutive 16 B).
//   * k is assigned so that a lane's 4 MFMA steps take 4 CONSECUTIVE input features
//     (step s, lane l: feature 16c + 4*(l>>4) + s).  Forward: the lane's A operands of a k-chunk are
//     one 16-byte load from the row-major weight W[out][in]; backward (A = W^T): four 4-byte loads,
//     each 4 x 64 contiguous bytes per wave.  Nothing is packed or transposed on the host.
//   * a workgroup (4 waves, one per SIMD) owns 16*G rows.  Per layer the output blocks of 16
//     features are dealt to the waves; a wave computes a block for all G row groups at once (the A
//     operand is loaded once per block), the remainder blocks are split by row group so that all
//     four waves do the same number of MFMAs.
//   * out-of-range weight rows are fetched through a bounds-checked buffer descriptor (zero), the
//     K padding of a tile is zero in LDS, so padded features stay exactly zero through the chain.
// Numerics: exact fp32 products, fp32 accumulation in k order (bitwise a chain of fmaf), bias
// added after the sum like the library epilogue; ELU = x > 0 ? x : exp(x) - 1.
// LDS tile of a layer with `nb` 16-feature blocks: [nb][G][64 lanes][4] floats (nb*G KiB).

#include "mlp_chain_shared.hpp"
#include "optim_common.hpp"

namespace rlg {


// A wave's share of one layer: `nunits` output units, unit j = the NF consecutive 16-feature blocks
// ob_of(j) .. ob_of(j)+NF-1 for the NG row groups g_of(j) .. g_of(j)+NG-1 (NF > 1 shares the B fragments
// and spreads the per-unit costs - epilogue, K tail, hand-over to the next unit - over NF times the MFMAs):
//   [request the first batch of unit j+1];  pre(j);  acc[f][g] = sum over the KC k-chunks of  A(ob+f, chunk) x B(chunk, group);  epi(j, acc)
// (pre: loads the epilogue will need, issued before the unit's MFMAs)
// kTransposedA = false: A[i][k] = W[ob*16 + i][k]        (forward, W row-major [out=i][in=k], ld = K)
// kTransposedA = true : A[i][k] = W[k][ob*16 + i]        (backward, W row-major [out=k][in=i], ld = I)
// Software pipeline: k-chunks are issued in batches of 4 (16*NG MFMAs); the A operands of the NEXT
// batch - which may be the first batch of the next unit - are in flight while the current batch
// issues, the B fragments (LDS) are double-buffered one chunk ahead, also across unit boundaries,
// so neither the L2 latency at the start of a unit nor the epilogue of the previous one is exposed.
// All register arrays are indexed statically (a rotation through moves would make every step wait
// for the load it has just issued); the K remainder (KC % 4 chunks) is a switch over straight-line
// tails, so there is no branch between a fragment load and the MFMAs in front of which it is issued.
// hipcc's scheduler otherwise sinks the prefetch loads down to their first use (and the fragment
// reads to two MFMAs before theirs): pin the order issue-loads / MFMA group / issue-loads / ...
#define RLG_PIN() __builtin_amdgcn_sched_barrier(0)

template <int NG, int NF, bool kTransposedA, class ObOf, class GOf, class Pre, class Epi>
__device__ __forceinline__ void chain_units(rsrc_t wr, int I, int K, int ld, const float* in_tile, int G,
                                            int nunits, ObOf ob_of, GOf g_of, Pre pre, Epi epi,
                                            long long* dbg = nullptr, int dbg_wave = 0, int* dbg_slot = nullptr,
                                            int kc_begin = 0, int kc_end = -1) {
  if (nunits <= 0) return;
  const int lane = lane_id();
  // k-chunks [kc_begin, kc_end) of the reduction (default: all of them; a K-split unit takes a part)
  const int kc_all = (K + 15) >> 4;
  const int KC = (kc_end < 0 ? kc_all : kc_end) - kc_begin;
  const int nfull = KC >> 2;
  const int rem = KC & 3;
  const unsigned astep = kTransposedA ? static_cast<unsigned>(16 * ld * 4) : 64u;
  const int bstride = G * 256;
  auto a_base = [&](int j, int f) -> unsigned {
    if (j >= nunits || (kAbl & 64)) return kOob;
    const int i = (ob_of(j) + f) * 16 + (lane & 15);
    if (i >= I) return kOob;
    const unsigned first = static_cast<unsigned>(kc_begin) * astep;
    return first + (kTransposedA ? static_cast<unsigned>(((4 * (lane >> 4)) * ld + i) * 4)
                                 : static_cast<unsigned>((i * ld + 4 * (lane >> 4)) * 4));
  };
  auto b_base = [&](int j) -> const float* {
    const int jj = (j < nunits) ? j : nunits - 1;
    return in_tile + kc_begin * bstride + (g_of(jj) * 64 + lane) * 4;
  };
  auto load_a = [&](unsigned base, int c) -> f32x4 {
    const unsigned off = (c < KC) ? base + static_cast<unsigned>(c) * astep : kOob;
    if (!kTransposedA) return buf_load4(wr, off);
    f32x4 v;
    v[0] = buf_load1(wr, off);
    v[1] = buf_load1(wr, off + static_cast<unsigned>(ld * 4));
    v[2] = buf_load1(wr, off + static_cast<unsigned>(ld * 8));
    v[3] = buf_load1(wr, off + static_cast<unsigned>(ld * 12));
    return v;
  };
  auto load_b = [&](f32x4 (&bv)[NG], const float* p) {
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p + g * 256);
  };

  unsigned abase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) abase[f] = a_base(0, f);
  const float* bp = b_base(0);
  f32x4 cur[4][NF], nxt[4][NF];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[u][f] = load_a(abase[f], u);
  }
  f32x4 b0[NG], b1[NG];
  load_b(b0, bp);

  for (int j = 0; j < nunits; ++j) {
    unsigned abase_n[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) abase_n[f] = a_base(j + 1, f);
    const float* bp_n = b_base(j + 1);
    // the first batch of the NEXT unit is requested now: it has this whole unit to arrive
    f32x4 nn[4][NF];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int f = 0; f < NF; ++f) nn[u][f] = load_a(abase_n[f], u);
    }
    pre(j);
    f32x4 acc[NF][NG];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[f][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // NG == NF == 1: a second accumulator takes the odd steps (40-cycle dependent-issue latency vs
    // 32-cycle issue), folded in before the epilogue - a fixed order, so still deterministic
    f32x4 acc_odd = {0.0f, 0.0f, 0.0f, 0.0f};
    // Accumulators live in AGPRs: every MFMA then accumulates in place and the VGPRs stay free for the
    // operand fragments.  (History: the 8-wave backward instance once produced wrong fragment registers 2, 3
    // for K <= 32 in VGPR form.  The cause was NOT the destination / source overlap suspected first - every
    // overlap form is exact on gfx950, tools/exp/mfma_overlap_probe.hip - but a missing wait state between the
    // last MFMA of a tail case and the first VALU read of its result, see the s_nop in front of the epilogue.)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < NG; ++g) asm volatile("" : "+a"(acc[f][g]));
    }
    asm volatile("" : "+a"(acc_odd));
    // one k-chunk = 4 MFMA steps x NF blocks x NG groups; the fragment reads of the following chunk are
    // issued after step 0, so that a wait for THIS chunk's fragments never includes them
    auto mfma_head = [&](const f32x4 (&av)[NF], const f32x4 (&bv)[NG]) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][0], bv[g][0], acc[f][g], 0, 0, 0);
      }
    };
    auto mfma_rest = [&](const f32x4 (&av)[NF], const f32x4 (&bv)[NG]) {
      if constexpr (NG == 1 && NF == 1) {
        acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][1], bv[0][1], acc_odd, 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][2], bv[0][2], acc[0][0], 0, 0, 0);
        acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][3], bv[0][3], acc_odd, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 1; s < 4; ++s) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
              acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][s], bv[g][s], acc[f][g], 0, 0, 0);
          }
        }
      }
    };
    for (int bi = 0; bi < nfull; ++bi) {
      const int c = bi * 4;
      const bool wraps = (c + 4 >= KC);                 // the following chunk belongs to the next unit
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[u][f] = load_a(wraps ? kOob : abase[f], c + 4 + u);
      }
      mfma_head(cur[0], b0);
      RLG_PIN();
      load_b(b1, bp + (c + 1) * bstride);
      RLG_PIN();
      mfma_rest(cur[0], b0);
      RLG_PIN();
      mfma_head(cur[1], b1);
      RLG_PIN();
      load_b(b0, bp + (c + 2) * bstride);
      RLG_PIN();
      mfma_rest(cur[1], b1);
      RLG_PIN();
      mfma_head(cur[2], b0);
      RLG_PIN();
      load_b(b1, bp + (c + 3) * bstride);
      RLG_PIN();
      mfma_rest(cur[2], b0);
      RLG_PIN();
      mfma_head(cur[3], b1);
      RLG_PIN();
      load_b(b0, wraps ? bp_n : bp + (c + 4) * bstride);
      RLG_PIN();
      mfma_rest(cur[3], b1);
      RLG_PIN();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[u][f] = nxt[u][f];
      }
    }
    if (dbg != nullptr && j < 2) chain_stamp(dbg, dbg_wave, *dbg_slot);     // after the full batches
    if (rem != 0) {
      // tail chunks nfull*4 .. KC-1 are in cur[0..rem-1]; next comes the first batch of the next unit
      const float* bt = bp + (nfull * 4) * bstride;
      if (rem == 1) {
        mfma_head(cur[0], b0);
        RLG_PIN();
        load_b(b1, bp_n);
        RLG_PIN();
        mfma_rest(cur[0], b0);
        RLG_PIN();
#pragma unroll
        for (int g = 0; g < NG; ++g) b0[g] = b1[g];
      } else if (rem == 2) {
        mfma_head(cur[0], b0);
        RLG_PIN();
        load_b(b1, bt + bstride);
        RLG_PIN();
        mfma_rest(cur[0], b0);
        RLG_PIN();
        mfma_head(cur[1], b1);
        RLG_PIN();
        load_b(b0, bp_n);
        RLG_PIN();
        mfma_rest(cur[1], b1);
        RLG_PIN();
      } else {
        mfma_head(cur[0], b0);
        RLG_PIN();
        load_b(b1, bt + bstride);
        RLG_PIN();
        mfma_rest(cur[0], b0);
        RLG_PIN();
        mfma_head(cur[1], b1);
        RLG_PIN();
        load_b(b0, bt + 2 * bstride);
        RLG_PIN();
        mfma_rest(cur[1], b1);
        RLG_PIN();
        mfma_head(cur[2], b0);
        RLG_PIN();
        load_b(b1, bp_n);
        RLG_PIN();
        mfma_rest(cur[2], b0);
        RLG_PIN();
#pragma unroll
        for (int g = 0; g < NG; ++g) b0[g] = b1[g];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int f = 0; f < NF; ++f) cur[u][f] = nn[u][f];
    }
    // The next unit's first fragments (requested at the start of this unit) are claimed HERE, in front of the
    // epilogue: hipcc would otherwise place the copy - and its s_waitcnt vmcnt(0) - at the top of
    // the next unit, behind the epilogue's global stores, and wait for those as well.
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(cur[u][f]));
    }
    // One more issue slot in front of the first read of an accumulator: gfx950 needs 10 wait states between
    // v_mfma_f32_16x16x4_f32 and a VALU read of its result (40 cycles, NOT interlocked - a too-early read
    // returns the old register contents), hipcc (ROCm 7.2) counts one short across the branch from a tail
    // case to this merge point (tools/audit_mfma.py, tools/exp/mfma_valu_read_probe.hip).
    // (The accumulators pass through the asm statements, which hipcc keeps in order: every AGPR read of the
    // epilogue is issued behind the s_nop.)
    asm volatile("s_nop 0" : "+a"(acc[0][0]));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (f + g > 0) asm volatile("" : "+a"(acc[f][g]));
      }
    }
    asm volatile("" : "+a"(acc_odd));
    if constexpr (NG == 1 && NF == 1) acc[0][0] += acc_odd;
    if (dbg != nullptr && j < 2) chain_stamp(dbg, dbg_wave, *dbg_slot);     // after the tail + claim of the next fragments
    epi(j, acc);
    if (dbg != nullptr && j < 2) chain_stamp(dbg, dbg_wave, *dbg_slot);     // after the epilogue
#pragma unroll
    for (int f = 0; f < NF; ++f) abase[f] = abase_n[f];
    bp = bp_n;
  }
}


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int G, int HACT, int W>
__global__ __launch_bounds__(64 * W) void mlp_chain_fwd_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (static_cast<int>(blockIdx.x) >= a.fwd_blocks) {      // the workgroups behind the row tiles: weight planes
    for (int t = threadIdx.x; t < 256; t += 64 * W) chain_pack_planes_block(a.pack, static_cast<int>(blockIdx.x) - a.fwd_blocks, t);
    return;
  }
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const long long row0 = static_cast<long long>(blockIdx.x) * (16 * G);
  float* tile_a = lds;
  float* tile_b = lds + a.lds_b_floats;
  int stamp = 0;
  chain_stamp(a.dbg, wave, stamp);

  // ---- prologue: observation tile -> LDS (fragment layout), normalised on the way -----------------
  chain_fwd_prologue<G, W>(a, tile_a, tile_b, row0, lane, wave, stamp);
  chain_stamp(a.dbg, wave, stamp);

  // ---- the layers ---------------------------------------------------------------------------------
  float* tin = tile_a;
  float* tout = tile_b;
  for (int L = 0; L < a.num_layers; ++L) {
    const bool last = (L == a.num_layers - 1);
    const int l_in = pin_s(a.layer[L].in), l_out = pin_s(a.layer[L].out), l_act = pin_s(a.layer[L].act);
    const float* l_bias = pin_s(a.layer[L].bias);
    float* l_h = pin_s(a.layer[L].h);
    const long long l_ldh = pin_s(a.layer[L].ldh);
    const long long n_rows = pin_s(a.rows);
    const rsrc_t wr = make_rsrc(a.layer[L].w, static_cast<unsigned>(l_in) * l_out * 4u);
    const int NOB = (l_out + 15) >> 4;
    const int full = NOB / W;
    // fast path: every fragment inside the matrix is one aligned 16-byte store
    const bool h_on = l_h != nullptr;
    const bool h_fast = pin_s(static_cast<int>(h_on && vec4_ok(l_h, l_ldh) && (l_out & 3) == 0)) != 0;
    // bias of the unit's 4 features: requested before the unit's MFMAs (pre), used in its epilogue
    // blocks per unit (whole blocks), the rest as one smaller unit.  G = 2 runs two workgroups per CU (two
    // waves per SIMD): 3 blocks per unit would push it past 256 registers and halve the occupancy.
    constexpr int kNF = (G == 2) ? 2 : 3;
    f32x4 biasv[kNF];
    const bool bias_fast = pin_s(static_cast<int>(aligned16(l_bias) && (l_out & 3) == 0)) != 0;
    auto load_bias = [&](int ob, int slot) {
      const int f = ob * 16 + 4 * (lane >> 4);
      biasv[slot] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (bias_fast) {
        if (f < l_out) biasv[slot] = *(const glob_t<f32x4>*)as_global(l_bias + f);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) biasv[slot][e] = (f + e < l_out) ? as_global(l_bias)[f + e] : 0.0f;
      }
    };
    auto epilogue = [&](int ob, int g, const f32x4& accv, const f32x4& bias) {
      const int f = ob * 16 + 4 * (lane >> 4);
      if ((kAbl & 4) && n_rows >= 0) return;
      const f32x4 v = (kAbl & 1) ? accv : chain_act4<HACT>(accv + bias, l_act);
      if (!last) *reinterpret_cast<f32x4*>(tout + ((ob * G + g) * 64 + lane) * 4) = v;
      const long long row = row0 + g * 16 + (lane & 15);
      if ((kAbl & 2) && !last) return;
      if (h_fast) {
        if (row < n_rows && f < l_out) *(glob_t<f32x4>*)as_global(l_h + row * l_ldh + f) = v;
      } else if (h_on && row < n_rows) {
        store_row4(l_h, l_ldh, row, f, l_out, v, false);
      }
    };
    // whole blocks: all G row groups, one weight stream per block; kNF blocks per unit
    auto whole = [&](auto nf_tag, int first_ob, int nunits, bool stamps) {
      constexpr int NF = decltype(nf_tag)::value;
      chain_units<G, NF, false>(
          wr, l_out, l_in, l_in, tin, G, nunits, [&](int j) { return first_ob + j * NF; }, [&](int) { return 0; },
          [&](int j) {
#pragma unroll
            for (int f = 0; f < NF; ++f) load_bias(first_ob + j * NF + f, f);
          },
          [&](int j, const f32x4 (&acc)[NF][G]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
              for (int g = 0; g < G; ++g) epilogue(first_ob + j * NF + f, g, acc[f][g], biasv[f]);
            }
          },
          stamps ? a.dbg : nullptr, wave, &stamp);
    };
    {
      const int units = full / kNF, left = full - units * kNF;
      whole(std::integral_constant<int, kNF>{}, wave * full, units, L < 2);
      if (kNF == 3 && left == 2) whole(std::integral_constant<int, 2>{}, wave * full + units * kNF, 1, false);
      else if (left >= 1) whole(std::integral_constant<int, 1>{}, wave * full + units * kNF, 1, false);
    }
    chain_stamp(a.dbg, wave, stamp);
    // remainder blocks: dealt out per (block, row group) so that every wave gets the same share
    const int rem_first = full * W;
    const int rem_units = (kAbl & 8) ? 0 : (NOB - rem_first) * G;
    // Fewer (block, row group) units than waves (the one remainder block of the 400- and 200-wide layers):
    // the idle waves take a share of the reduction instead - `ksplit` waves per unit, each over a part
    // of the k-chunks, partial fragments combined through LDS in a fixed order by the unit's first wave.
    const int ksplit = (a.no_ksplit || rem_units <= 0) ? 1 : ((4 * rem_units <= W) ? 4 : ((2 * rem_units <= W) ? 2 : 1));
    if (ksplit > 1) {
      const bool active = wave < rem_units * ksplit;
      const int bu = active ? wave % rem_units : 0;           // unit, part of its reduction
      const int part = active ? wave / rem_units : 0;
      const int ob = rem_first + bu / G, g = bu % G;
      const int KCall = (l_in + 15) >> 4;
      f32x4 held = {0.0f, 0.0f, 0.0f, 0.0f};
      chain_units<1, 1, false>(
          wr, l_out, l_in, l_in, tin, G, active ? 1 : 0, [&](int) { return ob; }, [&](int) { return g; },
          [&](int) { if (part == 0) load_bias(ob, 0); },
          [&](int, const f32x4 (&acc)[1][1]) { held = acc[0][0]; },
          nullptr, 0, nullptr, (KCall * part) / ksplit, (KCall * (part + 1)) / ksplit);
      float* scratch = lds + a.lds_split_floats;
      if (active && part > 0) *reinterpret_cast<f32x4*>(scratch + ((bu * (ksplit - 1) + part - 1) * 64 + lane) * 4) = held;
      __syncthreads();
      if (active && part == 0) {
        for (int q = 1; q < ksplit; ++q) held += *reinterpret_cast<const f32x4*>(scratch + ((bu * (ksplit - 1) + q - 1) * 64 + lane) * 4);
        epilogue(ob, g, held, biasv[0]);
      }
    } else {
    const int my_rem = (rem_units > wave) ? (rem_units - wave + W - 1) / W : 0;
    chain_units<1, 1, false>(
        wr, l_out, l_in, l_in, tin, G, my_rem, [&](int j) { return rem_first + (wave + j * W) / G; },
        [&](int j) { return (wave + j * W) % G; },
        [&](int j) { load_bias(rem_first + (wave + j * W) / G, 0); },
        [&](int j, const f32x4 (&acc)[1][1]) {
          const int u = wave + j * W;
          epilogue(rem_first + u / G, u % G, acc[0][0], biasv[0]);
        });
    }
    chain_stamp(a.dbg, wave, stamp);
    if (!(kAbl & 16)) __syncthreads();
    chain_stamp(a.dbg, wave, stamp);
    float* t = tin;
    tin = tout;
    tout = t;
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined forward for large minibatches: 64-row workgroups (G = 4), 4 waves = one per SIMD.
//
// What the ablation builds of the unit-structured kernel above showed (tools/ablate_chain.sh,
// profiles/r3_chain_ablation.txt; 32,768 rows): of its 131 us, 20 are epilogues that no MFMA overlaps at one
// wave per SIMD, 10 the inefficiency of the remainder units (a cold weight fetch and a barrier behind a handful of
// MFMAs each), 12 weight-fetch latency at the head of every chain_units call.  This kernel keeps ONE stream of
// chunk steps going per wave and layer instead:
//   * a unit is ONE 16-feature output block for all four row groups (4 accumulators), or - the last NOB mod 4
//     blocks of a layer - one block for ONE row group (every wave takes a different group: same MFMA count for all);
//     both kinds are steps of the same stream, so the remainder units inherit the weight prefetch of their
//     predecessors.  Four chunk slots of A fragments are in flight; a slot is refilled right behind the MFMAs that
//     consumed it with the chunk four steps ahead - of this unit, else of the NEXT unit, else of the next LAYER'S
//     first unit (weights do not depend on the barrier), so no weight latency is ever exposed after the first unit;
//   * a finished unit's accumulators are copied aside (16 moves) and its epilogue - bias, activation, LDS write,
//     global store - runs as four pieces inside the first four chunk steps of the following unit;
//   * every global access is a buffer instruction: one resource over all weights and biases (the flat parameter
//     arena), one per activation output with the row range as its bound, so the ragged last tile needs no masks
//     and the addresses are 32-bit lane offsets.
// Same products in the same order as mlp_chain_fwd_kernel<4>: the results are bit-identical.
// ------------------------------------------------------------------------------------------------

struct PipeGeo {
  int in, out, KC, full, nrem, rem_first, nunits;
  unsigned w_off;
};

template <int G, int HACT, int W>
__device__ __forceinline__ void chain_fwd_pipe_body(const ChainArgs& a, float* lds) {
  using WholeTag = std::integral_constant<int, G>;
  using OneTag = std::integral_constant<int, 1>;
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4);
  const long long row0 = static_cast<long long>(blockIdx.x) * (16 * G);
  float* tile_a = lds;
  float* tile_b = lds + a.lds_b_floats;
  const rsrc_t wr = make_rsrc(a.w_base, a.w_bytes);
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);

  auto geo = [&](int L) -> PipeGeo {
    PipeGeo g;
    if (L >= num_layers) {
      g.in = g.out = g.KC = g.full = g.nrem = g.rem_first = g.nunits = 0;
      g.w_off = kOob;
      return g;
    }
    g.in = pin_s(a.layer[L].in);
    g.out = pin_s(a.layer[L].out);
    g.w_off = static_cast<unsigned>(pin_s(static_cast<int>(a.w_off[L])));
    g.KC = (g.in + 15) >> 4;
    const int NOB = (g.out + 15) >> 4;
    g.full = NOB / W;
    g.rem_first = g.full * W;
    // the remaining blocks are dealt out per (block, row group): unit u = wave + 4 j -> block u / G, group u % G
    const int rem_units = (NOB - g.rem_first) * G;
    g.nrem = rem_units > wave ? (rem_units - wave + W - 1) / W : 0;
    g.nunits = g.full + g.nrem;
    return g;
  };
  auto unit_ob = [&](const PipeGeo& g, int idx) -> int {
    return idx < g.full ? wave * g.full + idx : g.rem_first + (wave + W * (idx - g.full)) / G;
  };
  auto rem_group = [&](int r) -> int { return (wave + W * r) % G; };
  // per-lane byte offset of chunk 0 of block ob's weight rows; out-of-range rows read zero
  auto a_base = [&](const PipeGeo& g, int ob) -> unsigned {
    const int i = ob * 16 + (lane & 15);
    return (!(kAbl & 64) && g.nunits > 0 && i < g.out) ? g.w_off + static_cast<unsigned>((i * g.in + q4) * 4) : kOob;
  };
  // ---- the first unit's weights are requested before anything else
  PipeGeo cur = geo(0);
  f32x4 aq[4];
  auto first_batch = [&](const PipeGeo& g) {
    const unsigned base0 = a_base(g, unit_ob(g, 0));
#pragma unroll
    for (int u = 0; u < 4; ++u) aq[u] = buf_load4(wr, u < g.KC ? base0 + static_cast<unsigned>(u) * 64u : kOob);
  };
  first_batch(cur);
  bool have_aq = cur.nunits > 0;      // aq holds the first batch of this wave's first unit of the layer
  int stamp = 0;
  chain_stamp(a.dbg, wave, stamp);                                   // tools/exp/pipe_phases.py: start
  chain_fwd_prologue<G, W>(a, tile_a, tile_b, row0, lane, wave, stamp);    // tile in LDS (stamped inside)
  chain_stamp(a.dbg, wave, stamp);                                   // behind the prologue's barrier

  f32x4 acc[4], accP[4], b0[4], b1[4], an[4], ar[4], biasC, biasP;
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = accP[g] = an[g] = ar[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  biasC = biasP = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  float* tin = tile_a;
  float* tout = tile_b;
  for (int L = 0; L < num_layers; ++L) {
    const bool last = (L == num_layers - 1);
    const PipeGeo nxt = geo(L + 1);
    const int l_out = cur.out, l_act = pin_s(a.layer[L].act);
    const unsigned b_off = static_cast<unsigned>(pin_s(static_cast<int>(a.b_off[L])));
    float* l_h = pin_s(a.layer[L].h);
    const long long l_ldh = pin_s(a.layer[L].ldh);
    const bool h_on = l_h != nullptr;
    const bool h_fast = pin_s(static_cast<int>(h_on && vec4_ok(l_h, l_ldh) && (l_out & 3) == 0)) != 0;
    const bool bias_fast = pin_s(static_cast<int>(((reinterpret_cast<uintptr_t>(a.w_base) + b_off) & 15u) == 0 && (l_out & 3) == 0)) != 0;
    const rsrc_t hr = make_rsrc(h_on ? l_h + row0 * l_ldh : nullptr, h_on ? tile_bytes(n_rows - row0, 16 * G, l_ldh) : 0u);
    const unsigned h_lane = static_cast<unsigned>(((lane & 15) * static_cast<int>(l_ldh) + q4) * 4);
    const unsigned h_group = static_cast<unsigned>(16 * static_cast<int>(l_ldh) * 4);       // bytes between row groups
    const float* bp = tin + lane * 4;
    const int KC = cur.KC;

    auto load_bias = [&](int ob) -> f32x4 {
      const int f = ob * 16 + q4;
      if (bias_fast) return buf_load4(wr, f < l_out ? b_off + static_cast<unsigned>(f) * 4u : kOob);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = buf_load1(wr, f + e < l_out ? b_off + static_cast<unsigned>(f + e) * 4u : kOob);
      return v;
    };
    // B fragments of chunk c: all G row groups, or the one group g1 of a remainder unit
    auto load_b = [&](auto ng_tag, f32x4 (&bv)[4], int c, int g1) {
      constexpr int NG = decltype(ng_tag)::value;
      if (kAbl & 128) return;
      if constexpr (NG == G) {
#pragma unroll
        for (int g = 0; g < G; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bp + (c * G + g) * 256);
      } else {
        bv[0] = *reinterpret_cast<const f32x4*>(bp + (c * G + g1) * 256);
      }
    };
    int pend_ng = 0, pend_ob = 0, pend_g = 0;
    // epilogue piece U of the pending unit: fragment (pend_ob, row group U | pend_g)
    auto piece = [&](auto u_tag) {
      constexpr int U = decltype(u_tag)::value;
      const int g = pend_ng == G ? U : pend_g;
      const int f = pend_ob * 16 + q4;
      if ((kAbl & 4) && n_rows >= 0) return;
      const f32x4 v = (kAbl & 1) ? accP[U] : chain_act4<HACT>(accP[U] + biasP, l_act);
      if (!last) *reinterpret_cast<f32x4*>(tout + ((pend_ob * G + g) * 64 + lane) * 4) = v;
      if ((kAbl & 2) && !last) return;
      if (h_on) {
        const unsigned off = h_lane + static_cast<unsigned>(g) * h_group + static_cast<unsigned>(pend_ob) * 64u;
        if (h_fast) {
          buf_store4(hr, f < l_out ? off : kOob, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) buf_store1(hr, f + e < l_out ? off + 4u * e : kOob, v[e]);
        }
      }
    };
    // The MFMAs of chunk c of a unit with NG row groups, the B fragments of chunk c + 1 (double buffer b0 / b1 by
    // chunk parity; past the reduction's end the read is a harmless one of LDS behind the tile) and the refill of
    // A slot U with chunk c + 4 (zero beyond the reduction: an out-of-range load) - ONE scheduling region, in which
    // sched_group_barrier deals the LDS / VMEM / VALU instructions out between the MFMAs: at one wave per SIMD a
    // clump of more than ~6 other instructions between two MFMAs leaves the matrix core idle (tools/exp/mfma_probe).
    auto chunk = [&](auto ng_tag, auto bank_tag, auto u_tag, int c, int g1, unsigned base_cur) {
      constexpr int NG = decltype(ng_tag)::value;
      constexpr int U = decltype(u_tag)::value;
      constexpr int BANK = decltype(bank_tag)::value;
      f32x4 (&bc)[4] = (U & 1) ? b1 : b0;
      f32x4 (&bn)[4] = (U & 1) ? b0 : b1;
      // A slots come in two banks that alternate by batch of four chunks: the refill for chunk c + 4 goes into the
      // OTHER bank's slot U while this bank's slot is still being read - straight into its final register
      const f32x4 av = (BANK == 0) ? aq[U] : ar[U];
      load_b(ng_tag, bn, c + 1, g1);
      ((BANK == 0) ? ar[U] : aq[U]) = buf_load4(wr, c + 4 < KC ? base_cur + static_cast<unsigned>(c + 4) * 64u : kOob);
      if constexpr (NG == G && G > 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bc[g][s], acc[g], 0, 0, 0);
        }
        // issue order: MFMA, LDS read, MFMA, LDS read, ... then the refill's address arithmetic and load
#pragma unroll
        for (int g = 0; g < G; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * G - G - 2, 0);
      } else {
        // a second accumulator takes the odd steps (40-cycle dependent-issue latency vs 32-cycle issue)
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bc[0][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bc[0][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bc[0][2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bc[0][3], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      RLG_PIN();
    };
    // One unit: [request the next unit's first batch + this unit's bias] [first batch: chunks 0..3, each followed
    // by a piece of the pending epilogue] [the remaining batches] [hand-over].
    auto run_unit = [&](auto ng_tag, int ob, int g1, unsigned base_cur, unsigned base_nxt, int KC_nxt) {
      constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) an[u] = buf_load4(wr, u < KC_nxt ? base_nxt + static_cast<unsigned>(u) * 64u : kOob);
      biasC = load_bias(ob);
      RLG_PIN();
      {
        constexpr std::integral_constant<int, 0> U0{};
        constexpr std::integral_constant<int, 1> U1{};
        constexpr std::integral_constant<int, 2> U2{};
        constexpr std::integral_constant<int, 3> U3{};
        constexpr std::integral_constant<int, 0> BA{};
        constexpr std::integral_constant<int, 1> BB{};
        chunk(ng_tag, BA, U0, 0, g1, base_cur);
        if (pend_ng > 0) piece(U0);
        RLG_PIN();
        if (1 < KC) chunk(ng_tag, BA, U1, 1, g1, base_cur);
        if (pend_ng > 1) piece(U1);
        RLG_PIN();
        if (2 < KC) chunk(ng_tag, BA, U2, 2, g1, base_cur);
        if constexpr (G > 2) { if (pend_ng > 2) piece(U2); }
        RLG_PIN();
        if (3 < KC) chunk(ng_tag, BA, U3, 3, g1, base_cur);
        if constexpr (G > 2) { if (pend_ng > 3) piece(U3); }
        RLG_PIN();
        // the body of the reduction: eight chunks per iteration, no branch between them (hipcc falls back to
        // vmcnt(0) behind a conditional chunk, i.e. it would wait for the refill it has just issued)
        int c0 = 4;
        for (; c0 + 8 <= KC; c0 += 8) {
          chunk(ng_tag, BB, U0, c0, g1, base_cur);
          chunk(ng_tag, BB, U1, c0 + 1, g1, base_cur);
          chunk(ng_tag, BB, U2, c0 + 2, g1, base_cur);
          chunk(ng_tag, BB, U3, c0 + 3, g1, base_cur);
          chunk(ng_tag, BA, U0, c0 + 4, g1, base_cur);
          chunk(ng_tag, BA, U1, c0 + 5, g1, base_cur);
          chunk(ng_tag, BA, U2, c0 + 6, g1, base_cur);
          chunk(ng_tag, BA, U3, c0 + 7, g1, base_cur);
        }
        // its last 0 .. 7 chunks
        if (c0 < KC) {
          chunk(ng_tag, BB, U0, c0, g1, base_cur);
          if (c0 + 1 < KC) chunk(ng_tag, BB, U1, c0 + 1, g1, base_cur);
          if (c0 + 2 < KC) chunk(ng_tag, BB, U2, c0 + 2, g1, base_cur);
          if (c0 + 3 < KC) chunk(ng_tag, BB, U3, c0 + 3, g1, base_cur);
          if (c0 + 4 < KC) chunk(ng_tag, BA, U0, c0 + 4, g1, base_cur);
          if (c0 + 5 < KC) chunk(ng_tag, BA, U1, c0 + 5, g1, base_cur);
          if (c0 + 6 < KC) chunk(ng_tag, BA, U2, c0 + 6, g1, base_cur);
        }
      }
      // ---- hand-over: the accumulators move aside (their epilogue rides along with the next unit's first
      //      steps), the next unit's first weight fragments - requested a whole unit ago - become current
      if constexpr (NG == G && G > 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) accP[g] = acc[g];
      } else {
        accP[0] = acc[0] + acc[1];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) aq[u] = an[u];
      biasP = biasC;
      pend_ng = NG;
      pend_ob = ob;
      pend_g = g1;
    };

    // a wave without a unit in the previous layer (few blocks, G < 4) has nothing prefetched
    if (!have_aq && cur.nunits > 0) first_batch(cur);
    // the layer's first B fragments (the input tile is complete: a barrier precedes every layer)
    if (cur.full > 0) load_b(WholeTag{}, b0, 0, 0);
    else if (cur.nrem > 0) load_b(OneTag{}, b0, 0, rem_group(0));
    unsigned base_cur = a_base(cur, unit_ob(cur, 0));
    // whole blocks: all G row groups
    for (int idx = 0; idx < cur.full; ++idx) {
      const bool more = idx + 1 < cur.nunits;
      const unsigned base_nxt = more ? a_base(cur, unit_ob(cur, idx + 1)) : a_base(nxt, unit_ob(nxt, 0));
      run_unit(WholeTag{}, unit_ob(cur, idx), 0, base_cur, base_nxt, more ? KC : nxt.KC);
      if (idx + 1 < cur.full) load_b(WholeTag{}, b0, 0, 0);
      else if (more) load_b(OneTag{}, b0, 0, rem_group(0));
      base_cur = base_nxt;
    }
    chain_stamp(a.dbg, wave, stamp);                                 // per layer: whole units done
    // remainder blocks, one row group at a time: every wave the same number of MFMAs (+- one unit)
    for (int r = 0; r < ((kAbl & 8) ? 0 : cur.nrem); ++r) {
      const int idx = cur.full + r;
      const bool more = r + 1 < cur.nrem;
      const unsigned base_nxt = more ? a_base(cur, unit_ob(cur, idx + 1)) : a_base(nxt, unit_ob(nxt, 0));
      run_unit(OneTag{}, unit_ob(cur, idx), rem_group(r), base_cur, base_nxt, more ? KC : nxt.KC);
      if (more) load_b(OneTag{}, b0, 0, rem_group(r + 1));
      base_cur = base_nxt;
    }
    have_aq = cur.nunits > 0 && nxt.nunits > 0;
    chain_stamp(a.dbg, wave, stamp);                                 // remainder units done
    // the layer's last epilogue has no successor to ride with
    if (pend_ng > 0) piece(std::integral_constant<int, 0>{});
    if (pend_ng > 1) {
      piece(std::integral_constant<int, 1>{});
      if constexpr (G > 2) {
        piece(std::integral_constant<int, 2>{});
        piece(std::integral_constant<int, 3>{});
      }
    }
    chain_stamp(a.dbg, wave, stamp);                                 // last epilogue flushed
    if (!(kAbl & 16)) __syncthreads();
    chain_stamp(a.dbg, wave, stamp);                                 // behind the barrier
    float* t = tin;
    tin = tout;
    tout = t;
    cur = nxt;
  }
}

template <int G, int HACT, int W = 4>
__global__ __launch_bounds__(64 * W) void mlp_chain_fwd_pipe_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (static_cast<int>(blockIdx.x) >= a.fwd_blocks) {      // the workgroups behind the row tiles: weight planes
    if (threadIdx.x < 256) chain_pack_planes_block(a.pack, static_cast<int>(blockIdx.x) - a.fwd_blocks, threadIdx.x);
    return;
  }
  chain_fwd_pipe_body<G, HACT, W>(a, lds);
}

// ------------------------------------------------------------------------------------------------
// backward (dX chain): layer[num_layers-1] is the head, its dZ is `x` (d heads) itself.
// For L = num_layers-1 .. 1:  dZ_{L-1} = (dZ_L W_L) * act'_{L-1}(H_{L-1});  layer 0 needs no dX.
// ------------------------------------------------------------------------------------------------
template <int G, int W>
__global__ __launch_bounds__(64 * W) void mlp_chain_bwd_kernel(ChainArgs a, LossArgs loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const long long row0 = static_cast<long long>(blockIdx.x) * (16 * G);
  float* tile_a = lds;
  float* tile_b = lds + a.lds_b_floats;

  // ---- the PPO loss of this row tile first (training steps): d heads = d loss / d (value, mu) of the rows,
  //      the mu / sigma write-back and the tile's partial sums; the tile regions are still free
  if (a.with_loss) {
    ppo_loss_tile<16 * G, 64 * W>(loss, lds, blockIdx.x);
    // The prologue below reads d heads that OTHER waves of this workgroup have just stored.  A workgroup
    // barrier alone does not wait for global stores on gfx950 (hipcc emits no vmcnt(0) in front of
    // s_barrier at workgroup scope) and the loads did overtake them (wrong dZ with 8 waves): every wave
    // waits for its stores to be acknowledged by the L2 first.  (An agent-scope fence here - L2
    // write-back + invalidate in every workgroup - cost 15 ms per epoch.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- prologue: d heads tile -> LDS -----------------------------------------------------------
  {
    const int w = a.layer[a.num_layers - 1].out;
    const int KC0 = (w + 15) >> 4;
    const bool xv = vec4_ok(a.x, a.ldx);
    for (int u = wave; u < KC0 * G; u += W) {
      const int c = u / G;
      const int g = u - c * G;
      const long long row = row0 + g * 16 + (lane & 15);
      const int f = c * 16 + 4 * (lane >> 4);
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < a.rows) v = load_row4(a.x, a.ldx, row, f, w, xv);
      *reinterpret_cast<f32x4*>(tile_a + (u * 64 + lane) * 4) = v;
    }
    __syncthreads();
  }

  float* tin = tile_a;
  float* tout = tile_b;
  for (int L = a.num_layers - 1; L >= 1; --L) {
    // weights of layer L (K = its outputs, the produced features = its inputs); H / act / dZ / bias
    // partials of layer L-1, the layer whose dZ is produced
    const int l_in = pin_s(a.layer[L].in), l_out = pin_s(a.layer[L].out), p_act = pin_s(a.layer[L - 1].act);
    const float* p_h = pin_s(a.layer[L - 1].h);
    float* p_dz = pin_s(a.layer[L - 1].dz);
    const long long p_ldh = pin_s(a.layer[L - 1].ldh), p_lddz = pin_s(a.layer[L - 1].lddz);
    const long long n_rows = pin_s(a.rows);
    const rsrc_t wr = make_rsrc(a.layer[L].w, static_cast<unsigned>(l_in) * l_out * 4u);
    const int width = l_in;                     // == layer[L-1].out
    const int NOB = (width + 15) >> 4;
    const int full = NOB / W;
    const bool keep_tile = (L - 1 >= 1);        // dZ_0 feeds nothing further down
    const bool fast = pin_s(static_cast<int>(vec4_ok(p_h, p_ldh) && vec4_ok(p_dz, p_lddz) && (width & 3) == 0)) != 0;
    double* bpart = pin_s(a.layer[L - 1].bias_partials);
    if (bpart != nullptr) bpart += static_cast<long long>(blockIdx.x) * width;

    // dZ = acc * act'(h); stores; returns the lane's 4 feature values (zero for rows past the end).
    // `slot`: position of the (block, group) fragment in the output tile.
    auto epilogue = [&](int ob, int g, int slot, bool to_lds, const f32x4& accv, const f32x4& hval) -> f32x4 {
      const int f = ob * 16 + 4 * (lane >> 4);
      const long long row = row0 + g * 16 + (lane & 15);
      if ((kAbl & 4) && n_rows >= 0) return accv;
      f32x4 v = (kAbl & 1) ? accv : chain_act_grad4(accv, hval, p_act);
      if (row >= n_rows) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (to_lds) *reinterpret_cast<f32x4*>(tout + (slot * 64 + lane) * 4) = v;
      if (kAbl & 2) return v;
      if (fast) {
        if (row < n_rows && f < width) *(glob_t<f32x4>*)as_global(p_dz + row * p_lddz + f) = v;
      } else if (row < n_rows) {
        store_row4(p_dz, p_lddz, row, f, width, v, false);
      }
      return v;
    };
    auto load_h = [&](int ob, int g) -> f32x4 {
      if (kAbl & 4) return f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      const int f = ob * 16 + 4 * (lane >> 4);
      const long long row = row0 + g * 16 + (lane & 15);
      if (fast) {
        if (row < n_rows && f < width) return *(const glob_t<f32x4>*)as_global(p_h + row * p_ldh + f);
        return f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      if (row < n_rows) return load_row4(p_h, p_ldh, row, f, width, false);
      return f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    // column sums over the block's rows of one 16-feature block: the 16 lanes that share l>>4 hold
    // the 16 rows of a group for the same 4 features (fixed butterfly order: deterministic)
    auto colsum_store = [&](int ob, f32x4 s) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = s[e];
        t += __shfl_xor(t, 8, kWave);
        t += __shfl_xor(t, 4, kWave);
        t += __shfl_xor(t, 2, kWave);
        t += __shfl_xor(t, 1, kWave);
        s[e] = t;
      }
      if (bpart != nullptr && (lane & 15) == 0) {
        const int f = ob * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (f + e < width) as_global(bpart)[f + e] = static_cast<double>(s[e]);
        }
      }
    };

    // blocks per unit (whole blocks); the H fragments of a unit live in registers.  G = 4 runs two
    // workgroups per CU in this direction and must stay under 256 registers: one block per unit.
    constexpr int kNF = (G == 4) ? 1 : 2;
    f32x4 hval[kNF][G];
    auto whole = [&](auto nf_tag, int first_ob, int nunits) {
      constexpr int NF = decltype(nf_tag)::value;
      chain_units<G, NF, true>(
          wr, width, l_out, l_in, tin, G, nunits, [&](int j) { return first_ob + j * NF; }, [&](int) { return 0; },
          [&](int j) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
              for (int g = 0; g < G; ++g) hval[f][g] = load_h(first_ob + j * NF + f, g);
            }
          },
          [&](int j, const f32x4 (&acc)[NF][G]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const int ob = first_ob + j * NF + f;
              f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
              for (int g = 0; g < G; ++g) s += epilogue(ob, g, ob * G + g, keep_tile, acc[f][g], hval[f][g]);
              colsum_store(ob, s);
            }
          });
    };
    {
      const int units = full / kNF, left = full - units * kNF;
      whole(std::integral_constant<int, kNF>{}, wave * full, units);
      if (left == 1) whole(std::integral_constant<int, 1>{}, wave * full + units * kNF, 1);
    }
    // remainder blocks, one row group at a time.  The groups of such a block belong to different
    // waves, so every unit leaves its fragment in the output tile (at its normal place when the tile
    // feeds the next step, else in the first slots) and, after the barrier, one wave per block adds
    // the G groups in order.
    const int rem_first = full * W;
    const int rem_blocks = (kAbl & 8) ? 0 : NOB - rem_first;
    const int rem_units = rem_blocks * G;
    const int my_rem = (rem_units > wave) ? (rem_units - wave + W - 1) / W : 0;
    chain_units<1, 1, true>(
        wr, width, l_out, l_in, tin, G, my_rem, [&](int j) { return rem_first + (wave + j * W) / G; },
        [&](int j) { return (wave + j * W) % G; },
        [&](int j) {
          const int u = wave + j * W;
          hval[0][0] = load_h(rem_first + u / G, u % G);
        },
        [&](int j, const f32x4 (&acc)[1][1]) {
          const int u = wave + j * W;
          const int ob = rem_first + u / G, g = u % G;
          epilogue(ob, g, keep_tile ? ob * G + g : u, true, acc[0][0], hval[0][0]);
        });
    if (!(kAbl & 16)) __syncthreads();
    if (bpart != nullptr) {
      for (int rb = wave; rb < rem_blocks; rb += W) {
        const int slot0 = keep_tile ? (rem_first + rb) * G : rb * G;
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int g = 0; g < G; ++g) s += *reinterpret_cast<const f32x4*>(tout + ((slot0 + g) * 64 + lane) * 4);
        colsum_store(rem_first + rb, s);
      }
    }
    float* t = tin;
    tin = tout;
    tout = t;
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined backward for small minibatches (round 4): 16-row workgroups, 4 waves = one per SIMD.
//
// A data-parallel rank's minibatch (4,096 - 8,192 rows) is 256 - 512 workgroups of 16 rows, i.e. one or two per
// CU: the launch time is the serial chain of ONE workgroup, and mlp_chain_bwd_kernel<1, W> pays a cold weight fetch
// at the head of each of its chain_units calls there (two per layer step) with a handful of MFMAs behind it.
// This kernel is the backward counterpart of mlp_chain_fwd_pipe_kernel<1>: per wave and layer step ONE stream of
// chunk steps over the wave's output blocks (a unit = one 16-feature block of dZ_{L-1}, two accumulators taking
// the even / odd MFMA steps), the A operand (W_L transposed: four 4-byte loads per chunk and lane) refilled four
// chunks ahead into two register banks - across units and across LAYERS (weights and H do not depend on the
// barrier) -, the H fragment of the next unit requested a unit ahead, a finished unit's epilogue (act', LDS
// write, dZ store, bias column sums over the row group on the DPP path) riding behind the next unit's first chunk.
// Same exact fp32 products as mlp_chain_bwd_kernel<1, W>; every block accumulates its even k-steps and its odd
// k-steps in two accumulators that are added at the end (that kernel does so only for its single-block units).
// Requirements (host-checked, else the unit-structured kernel runs): H / dZ rows 16-byte aligned, widths % 4 == 0.
// ------------------------------------------------------------------------------------------------
struct BwdPipeGeo {
  int K, I, ld, KC, full, nunits, rem_first;
  const float* w;
};

template <int W, bool kPreloaded>
__device__ __forceinline__ void chain_bwd_pipe_body(const ChainArgs& a, const LossArgs& loss, float* lds,
                                                    const LossQuadInputs& preloaded) {
  constexpr std::integral_constant<int, 0> U0{};
  constexpr std::integral_constant<int, 1> U1{};
  constexpr std::integral_constant<int, 2> U2{};
  constexpr std::integral_constant<int, 3> U3{};
  constexpr std::integral_constant<int, 0> BA{};
  constexpr std::integral_constant<int, 1> BB{};
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4), r16 = lane & 15;
  const long long row0 = static_cast<long long>(blockIdx.x) * 16;
  float* tile_a = lds;
  float* tile_b = lds + a.lds_b_floats;
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);

  // geometry of the step that consumes layer L's weights (L >= 1) and produces dZ_{L-1}
  auto geo = [&](int L) -> BwdPipeGeo {
    BwdPipeGeo g;
    if (L < 1) {
      g.K = g.I = g.ld = g.KC = g.full = g.nunits = g.rem_first = 0;
      g.w = nullptr;
      return g;
    }
    g.K = pin_s(a.layer[L].out);
    g.I = pin_s(a.layer[L].in);
    g.ld = g.I;
    g.w = pin_s(a.layer[L].w);
    g.KC = (g.K + 15) >> 4;
    const int NOB = (g.I + 15) >> 4;
    g.full = NOB / W;
    g.rem_first = g.full * W;
    g.nunits = g.full + ((NOB - g.rem_first > wave) ? 1 : 0);
    return g;
  };
  auto unit_ob = [&](const BwdPipeGeo& g, int idx) -> int { return idx < g.full ? wave * g.full + idx : g.rem_first + wave; };
  // per-lane byte offset of (k = 4 q, i) in W_L [K][I]; out-of-range columns read zero
  auto a_base = [&](const BwdPipeGeo& g, int ob) -> unsigned {
    const int i = ob * 16 + r16;
    return (g.nunits > 0 && i < g.I) ? static_cast<unsigned>((q4 * g.ld + i) * 4) : kOob;
  };
  // chunk c of a block: k = 16 c + 4 q + s, s = 0 .. 3 (rows of W past K lie behind the resource: zero)
  auto load_a = [&](rsrc_t wr, unsigned base, int c, int KC, int ld) -> f32x4 {
    const unsigned step = static_cast<unsigned>(ld) * 4u;
    const unsigned off = (c < KC) ? base + static_cast<unsigned>(c) * 16u * step : kOob;
    f32x4 v;
    v[0] = buf_load1(wr, off);
    v[1] = buf_load1(wr, off + step);
    v[2] = buf_load1(wr, off + 2u * step);
    v[3] = buf_load1(wr, off + 3u * step);
    return v;
  };

  // ---- the first unit's weights are requested before anything else (they arrive during the loss tile)
  BwdPipeGeo cur = geo(num_layers - 1);
  rsrc_t wr_cur = make_rsrc(cur.w, static_cast<unsigned>(cur.K) * cur.I * 4u);
  f32x4 aq[4], ar[4], an[4];
  {
    const unsigned base0 = a_base(cur, unit_ob(cur, 0));
#pragma unroll
    for (int u = 0; u < 4; ++u) aq[u] = load_a(wr_cur, base0, u, cur.KC, cur.ld);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) ar[u] = an[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // ---- the PPO loss of this row tile (training steps), as in mlp_chain_bwd_kernel
  if (a.with_loss) {
    if constexpr (kPreloaded) ppo_loss_quad_run<16, 64 * W>(loss, lds, blockIdx.x, preloaded);   // (inputs requested long ago)
    else ppo_loss_tile<16, 64 * W>(loss, lds, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- prologue: d heads tile -> LDS (fragment layout)
  {
    const int w = a.layer[num_layers - 1].out;
    const int KC0 = (w + 15) >> 4;
    const bool xv = vec4_ok(a.x, a.ldx);
    for (int u = wave; u < KC0; u += W) {
      const long long row = row0 + r16;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < n_rows) v = load_row4(a.x, a.ldx, row, u * 16 + q4, w, xv);
      *reinterpret_cast<f32x4*>(tile_a + (u * 64 + lane) * 4) = v;
    }
    __syncthreads();
  }

  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f}, accP = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 b0 = {0.0f, 0.0f, 0.0f, 0.0f}, b1 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 hC = {0.0f, 0.0f, 0.0f, 0.0f}, hN = {0.0f, 0.0f, 0.0f, 0.0f}, hP = {0.0f, 0.0f, 0.0f, 0.0f};
  float* tin = tile_a;
  float* tout = tile_b;

  // H fragment of block ob of the layer whose dZ a step produces; rows past the end read zero
  auto h_rsrc = [&](int L) -> rsrc_t {
    if (L < 1) return make_rsrc(nullptr, 0u);
    const float* p_h = pin_s(a.layer[L - 1].h);
    const long long p_ldh = pin_s(a.layer[L - 1].ldh);
    return make_rsrc(p_h + row0 * p_ldh, tile_bytes(n_rows - row0, 16, p_ldh));
  };
  auto load_h = [&](rsrc_t hr, long long ldh, int width, int ob) -> f32x4 {
    const int f = ob * 16 + q4;
    return buf_load4(hr, f < width ? static_cast<unsigned>((r16 * static_cast<int>(ldh) + f) * 4) : kOob);
  };
  rsrc_t hr_cur = h_rsrc(num_layers - 1);
  if (cur.nunits > 0) hC = load_h(hr_cur, pin_s(a.layer[num_layers - 2].ldh), cur.I, unit_ob(cur, 0));

  for (int L = num_layers - 1; L >= 1; --L) {
    const BwdPipeGeo nxt = geo(L - 1);
    const rsrc_t wr_nxt = make_rsrc(nxt.w, static_cast<unsigned>(nxt.K) * nxt.I * 4u);
    const rsrc_t hr_nxt = h_rsrc(L - 1);
    const long long ldh_nxt = (L - 1 >= 1) ? pin_s(a.layer[L - 2].ldh) : 0;
    const int p_act = pin_s(a.layer[L - 1].act);
    float* p_dz = pin_s(a.layer[L - 1].dz);
    const long long p_ldh = pin_s(a.layer[L - 1].ldh), p_lddz = pin_s(a.layer[L - 1].lddz);
    const int width = cur.I;
    const bool keep_tile = (L - 1 >= 1);
    double* bpart = pin_s(a.layer[L - 1].bias_partials);
    if (bpart != nullptr) bpart += static_cast<long long>(blockIdx.x) * width;
    const rsrc_t dzr = make_rsrc(p_dz + row0 * p_lddz, tile_bytes(n_rows - row0, 16, p_lddz));
    const unsigned dz_lane = static_cast<unsigned>((r16 * static_cast<int>(p_lddz) + q4) * 4);
    const float* bp = tin + lane * 4;
    const int KC = cur.KC;
    const int ld = cur.ld;
    const bool row_ok = row0 + r16 < n_rows;
    int pend = -1;                                  // block whose epilogue is pending, or -1

    auto piece = [&]() {
      const int ob = pend;
      const int f = ob * 16 + q4;
      f32x4 v = chain_act_grad4(accP, hP, p_act);
      if (!row_ok) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (keep_tile) *reinterpret_cast<f32x4*>(tout + (ob * 64 + lane) * 4) = v;
      buf_store4(dzr, f < width ? dz_lane + static_cast<unsigned>(ob) * 64u : kOob, v);
      if (bpart != nullptr) {
        f32x4 s;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = row16_sum(v[e]);
        if (r16 == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (f + e < width) as_global(bpart)[f + e] = static_cast<double>(s[e]);
          }
        }
      }
    };
    // the MFMAs of chunk c (slot U of bank BANK), the B fragment of chunk c + 1 and the refill of the OTHER bank's
    // slot U with chunk c + 4 of this unit - one scheduling region
    auto chunk = [&](auto bank_tag, auto u_tag, int c, unsigned base_cur) {
      constexpr int U = decltype(u_tag)::value;
      constexpr int BANK = decltype(bank_tag)::value;
      f32x4& bc = (U & 1) ? b1 : b0;
      f32x4& bn = (U & 1) ? b0 : b1;
      const f32x4 av = (BANK == 0) ? aq[U] : ar[U];
      bn = *reinterpret_cast<const f32x4*>(bp + (c + 1) * 256);
      ((BANK == 0) ? ar[U] : aq[U]) = load_a(wr_cur, base_cur, c + 4, KC, ld);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bc[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bc[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bc[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bc[3], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      RLG_PIN();
    };
    // One unit: [request the next unit's first batch + its H] [chunk 0, the pending epilogue] [the other chunks] [hand-over]
    auto run_unit = [&](int ob, unsigned base_cur, rsrc_t wr_n, unsigned base_nxt, int KC_nxt, int ld_nxt, rsrc_t hr_n,
                        long long ldh_n, int width_n, int ob_nxt, bool has_nxt) {
#pragma unroll
      for (int u = 0; u < 4; ++u) an[u] = load_a(wr_n, base_nxt, u, KC_nxt, ld_nxt);
      hN = has_nxt ? load_h(hr_n, ldh_n, width_n, ob_nxt) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      RLG_PIN();
      chunk(BA, U0, 0, base_cur);
      if (pend >= 0) piece();
      RLG_PIN();
      if (1 < KC) chunk(BA, U1, 1, base_cur);
      if (2 < KC) chunk(BA, U2, 2, base_cur);
      if (3 < KC) chunk(BA, U3, 3, base_cur);
      int c0 = 4;
      for (; c0 + 8 <= KC; c0 += 8) {
        chunk(BB, U0, c0, base_cur);
        chunk(BB, U1, c0 + 1, base_cur);
        chunk(BB, U2, c0 + 2, base_cur);
        chunk(BB, U3, c0 + 3, base_cur);
        chunk(BA, U0, c0 + 4, base_cur);
        chunk(BA, U1, c0 + 5, base_cur);
        chunk(BA, U2, c0 + 6, base_cur);
        chunk(BA, U3, c0 + 7, base_cur);
      }
      if (c0 < KC) {
        chunk(BB, U0, c0, base_cur);
        if (c0 + 1 < KC) chunk(BB, U1, c0 + 1, base_cur);
        if (c0 + 2 < KC) chunk(BB, U2, c0 + 2, base_cur);
        if (c0 + 3 < KC) chunk(BB, U3, c0 + 3, base_cur);
        if (c0 + 4 < KC) chunk(BA, U0, c0 + 4, base_cur);
        if (c0 + 5 < KC) chunk(BA, U1, c0 + 5, base_cur);
        if (c0 + 6 < KC) chunk(BA, U2, c0 + 6, base_cur);
      }
      // hand-over (the s_nop: 10 wait states between the last MFMA and the first VALU read of its result on every
      // path into this point, tools/audit_mfma.py)
      asm volatile("s_nop 7\n\ts_nop 1" : "+a"(acc0), "+a"(acc1));
      accP = acc0 + acc1;
      acc0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) aq[u] = an[u];
      hP = hC;
      hC = hN;
      pend = ob;
    };

    if (cur.nunits > 0) b0 = *reinterpret_cast<const f32x4*>(bp);
    unsigned base_cur = a_base(cur, unit_ob(cur, 0));
    for (int idx = 0; idx < cur.nunits; ++idx) {
      const bool more = idx + 1 < cur.nunits;
      const int ob_nxt = more ? unit_ob(cur, idx + 1) : unit_ob(nxt, 0);
      const unsigned base_nxt = more ? a_base(cur, ob_nxt) : a_base(nxt, ob_nxt);
      if (more) run_unit(unit_ob(cur, idx), base_cur, wr_cur, base_nxt, KC, ld, hr_cur, p_ldh, width, ob_nxt, true);
      else run_unit(unit_ob(cur, idx), base_cur, wr_nxt, base_nxt, nxt.KC, nxt.ld, hr_nxt, ldh_nxt, nxt.I, ob_nxt, nxt.nunits > 0);
      if (more) b0 = *reinterpret_cast<const f32x4*>(bp);
      base_cur = base_nxt;
    }
    if (pend >= 0) piece();                         // the layer's last epilogue has no successor to ride with
    __syncthreads();
    float* t = tin;
    tin = tout;
    tout = t;
    // a wave without a unit in this step has nothing prefetched for the next one
    if (cur.nunits == 0 && nxt.nunits > 0) {
      const unsigned base0 = a_base(nxt, unit_ob(nxt, 0));
#pragma unroll
      for (int u = 0; u < 4; ++u) aq[u] = load_a(wr_nxt, base0, u, nxt.KC, nxt.ld);
      hC = load_h(hr_nxt, ldh_nxt, nxt.I, unit_ob(nxt, 0));
    }
    cur = nxt;
    wr_cur = wr_nxt;
    hr_cur = hr_nxt;
  }
}

template <int W>
__global__ __launch_bounds__(64 * W) void mlp_chain_bwd_pipe_kernel(ChainArgs a, LossArgs loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LossQuadInputs none;
  chain_bwd_pipe_body<W, false>(a, loss, lds, none);
}

// ------------------------------------------------------------------------------------------------
// One launch for forward + PPO loss + backward of a 16-row tile (round 4; minibatches < 16,384 rows).
//
// A data-parallel rank's optimiser step is five launches of 6 - 32 us, and two of the things it pays for exist only
// because forward and backward are separate launches: the launch boundary itself (drain of the forward, ramp of the
// backward: ~3 us), and the loss tile's input loads (actions, old mu / sigma, advantages, ... of the tile's 16 rows:
// an HBM round trip with nothing to overlap with at the head of the backward launch, ~9 us at 16 rows per workgroup).
// Here the loss inputs are requested FIRST and arrive during the forward (wave 0 keeps them in 37 registers), the
// forward's heads are read back from the L2 they were just written to, and the backward's first weights and H fragment
// are requested before the loss arithmetic like in its own launch.  Same device code as the two kernels, same results.
// ------------------------------------------------------------------------------------------------
template <int HACT, int W>
__global__ __launch_bounds__(64 * W) void mlp_chain_step_pipe_kernel(ChainArgs fa, ChainArgs ba, LossArgs loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LossQuadInputs pre;
  ppo_loss_quad_load<16, 64 * W>(loss, blockIdx.x, pre);
  chain_fwd_pipe_body<1, HACT, W>(fa, lds);
  // the heads (mu, value) this workgroup has just stored are what its loss tile reads: stores acknowledged by the L2
  // first (a workgroup barrier alone does not wait for global stores on gfx950), every wave past its last LDS access
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  chain_bwd_pipe_body<W, true>(ba, loss, lds, pre);
}


static bool chain_pipe_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("RLG_CHAIN_PIPE");       // tools: A/B against the unit-structured kernels
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}

// round 4: the pipelined forward for 16-row workgroups as well (minibatches < 16,384 rows: a data-parallel rank's
// shapes) - the unit-structured kernel pays a cold weight fetch at the head of each of its 8 chain_units calls there
// (profiles/r4_rank_chain_phases.txt: 58k cycles per tile for 21k of MFMA issue)
static bool chain_pipe1_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("RLG_CHAIN_PIPE1");      // tools: A/B against the unit-structured 16-row kernels
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}

static bool chain_bx_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("RLG_CHAIN_BX");         // tools: A/B against the exact-f32-product kernels
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}

// Launch kinds ("direction" of the C ABI): 0 = inference forward (rollouts, get_values), 1 = backward, 2 = training forward
// (activations kept).  Launches of at least chain_bx_min_rows(kind) rows run the 64-row split-product kernels: 16,384 for
// inference forwards, 8,192 for the training pair (round 6: a rank of 4's 8,192-row update is faster on the fp16 kernels on
// half the CUs than on the 16-row exact-product kernels - 46.1 -> 43.8 ms per rank epoch - while 8,192-row ROLLOUT forwards
// are not: a rank of 8 went 31.0 -> 32.7 ms with them).  RLG_CHAIN_BX_MIN_ROWS: one threshold for all kinds (tools, A/B).
static long long chain_bx_min_rows(int kind) {
  static const long long forced = [] {
    const char* e = std::getenv("RLG_CHAIN_BX_MIN_ROWS");
    return (e && std::atoll(e) > 0) ? std::atoll(e) : 0LL;
  }();
  if (forced > 0) return forced;
  return kind == 0 ? 16384LL : 8192LL;
}

static bool chain_bx_fwd_wanted(long long rows, int groups, bool training) {
  static const bool on = [] {
    const char* e = std::getenv("RLG_CHAIN_BX_FWD");     // tools: A/B against the exact-product forward kernels
    return !(e && std::atoi(e) == 0);
  }();
  // (2: what pick_groups resolves the automatic choice to at these sizes - the callers pass the resolved value)
  return on && chain_bx_enabled() && rows >= chain_bx_min_rows(training ? 2 : 0) && (groups == 0 || groups == 2 || groups == 4);
}

// Row groups per workgroup when the caller does not ask for one.  Measured on MI355X (humanoid MLP,
// profiles/r2_mlp_chain_microbench_*.txt): forward - two 32-row workgroups per CU (G = 2, two waves
// per SIMD) beat one 64-row workgroup (G = 4, one wave per SIMD): the second wave covers the epilogue
// and unit-boundary stalls of the first; below 16,384 rows G = 1 keeps every CU busy.
static int pick_groups(long long rows, int requested, int direction = 0) {
  if (requested == 1 || requested == 2 || requested == 4) return requested;
  {
    static const int forced_fwd = [] { const char* e = std::getenv("RLG_CHAIN_FWD_GROUPS"); return e ? std::atoi(e) : 0; }();
    static const int forced_bwd = [] { const char* e = std::getenv("RLG_CHAIN_BWD_GROUPS"); return e ? std::atoi(e) : 0; }();
    const int f = direction == 1 ? forced_bwd : forced_fwd;       // tools: A/B measurements inside bench.py
    if (rows >= chain_bx_min_rows(direction) && (f == 1 || f == 2 || f == 4)) return f;
  }
  // backward: its LDS footprint is half the forward's, so G = 4 already runs two workgroups per CU.
  // forward: two 32-row workgroups per CU (two waves per SIMD) - in the epoch that beats one 64-row workgroup for
  // both forward kernels (bench.py roofline_fwd via tools/bench_ab.sh: 122.8 / 132.6 us pipelined, 125 / 130.4 us
  // unit-structured), although a back-to-back microbenchmark says the opposite for the pipelined one
  if (rows >= chain_bx_min_rows(direction)) return direction == 1 ? 4 : 2;
  return 1;
}

// LDS bytes of one workgroup, forward (direction 0) or backward (direction 1); also writes the
// offset (in floats) of the second region.  -1: the chain does not fit the 160 KiB LDS.
static int chain_lds(int num_layers, const int* in_features, const int* out_features, int G, int direction,
                     int* b_floats_out) {
  long long a_kb = 0, b_kb = 0;      // region sizes in units of G KiB = 256*G floats
  if (direction == 0) {
    // tiles: T0 = input of layer 0 (region A), T_{L+1} = output of layer L (alternating), last layer -> global
    a_kb = (in_features[0] + 15) / 16;
    for (int L = 0; L + 1 < num_layers; ++L) {
      const long long nb = (out_features[L] + 15) / 16;
      if (L % 2 == 0) b_kb = nb > b_kb ? nb : b_kb;
      else a_kb = nb > a_kb ? nb : a_kb;
    }
  } else {
    // tiles: d heads (region A), then dZ_{L-1} for L = n-1 .. 2 alternating starting with B
    a_kb = (out_features[num_layers - 1] + 15) / 16;
    int flip = 0;
    for (int L = num_layers - 1; L >= 1; --L, flip ^= 1) {
      // dZ_{L-1} tile; the last step (dZ_0 feeds nothing) only stages its remainder blocks: at most 3 with
      // 4 waves per workgroup, at most 7 with the 8 waves of the 16-row workgroups (G == 1)
      const long long all = (in_features[L] + 15) / 16;
      const long long rem_max = (G == 1) ? 7 : 3;
      const long long nb = (L >= 2) ? all : (all < rem_max ? all : rem_max);
      if (flip == 0) b_kb = nb > b_kb ? nb : b_kb;
      else a_kb = nb > a_kb ? nb : a_kb;
    }
  }
  long long a_floats = a_kb * 256 * G, b_floats = b_kb * 256 * G;
  if (direction == 0) {
    const long long stats = 2LL * ((in_features[0] + 3) & ~3);   // mean32 / denom scratch in region B
    if (b_floats < stats) b_floats = (stats + 3) & ~3LL;
  }
  if (b_floats == 0) b_floats = 4;
  *b_floats_out = static_cast<int>(a_floats);
  long long bytes = (a_floats + b_floats) * 4;
  // forward: the K-split scratch behind the tile regions
  if (direction == 0) bytes += chain_split_floats(G) * 4;
  return bytes <= 160 * 1024 ? static_cast<int>(bytes) : -1;
}

int chain_fill(ChainArgs& args, int num_layers, const float* const* weights, const int* in_features,
                      const int* out_features, const int* acts) {
  if (num_layers < 1 || num_layers > kChainMaxLayers) return 1;
  for (int L = 0; L < num_layers; ++L) {
    ChainLayer& ly = args.layer[L];
    ly.w = weights[L];
    ly.in = in_features[L];
    ly.out = out_features[L];
    ly.act = acts[L];
    ly.bias = nullptr;
    ly.h = nullptr;
    ly.dz = nullptr;
    ly.bias_partials = nullptr;
    ly.ldh = ly.lddz = 0;
    if (ly.in <= 0 || ly.out <= 0 || (L > 0 && ly.in != out_features[L - 1])) return 1;
    if (static_cast<long long>(ly.in) * ly.out * 4 >= static_cast<long long>(kOob)) return 1;
    if (reinterpret_cast<uintptr_t>(ly.w) % 4 != 0) return 1;
  }
  args.num_layers = num_layers;
  args.dbg = nullptr;
  args.with_loss = 0;
  args.fwd_blocks = 0;
  args.pack.njobs = args.pack.total_pairs = 0;
  args.pack.dst = nullptr;
  args.planes = nullptr;
  return 0;
}

static bool g_chain_prepared = false;     // rlg_mlp_chain_prepare raised the LDS limit of every kernel
// rlg_mlp_chain_time_next: HIP events bound to the NEXT chain dispatch (its begin / end timestamps, what
// rocprofv3 --kernel-trace reports); one-shot, bench.py's roofline_fwd / roofline_bwd
static hipEvent_t g_chain_ev_start = nullptr, g_chain_ev_stop = nullptr;

template <int G, bool kBackward, int HACT, int W>
static int chain_launch_as(const ChainArgs& args_in, int lds_bytes, hipStream_t st, const LossArgs* loss) {
  ChainArgs args = args_in;
  int grid = static_cast<int>((args.rows + 16 * G - 1) / (16 * G));
  args.fwd_blocks = grid;
  if (!kBackward && args.pack.total_pairs > 0) grid += chain_bx_pack_blocks(args.pack);
  const void* kern;
  if constexpr (kBackward) kern = reinterpret_cast<const void*>(mlp_chain_bwd_kernel<G, W>);
  else kern = reinterpret_cast<const void*>(mlp_chain_fwd_kernel<G, HACT, W>);
  if (lds_bytes > 64 * 1024 && !g_chain_prepared) {
    static bool raised = false;          // one flag per instantiation = per kernel
    if (!raised) {
      const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return static_cast<int>(e);
      raised = true;
    }
  }
  hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
  g_chain_ev_start = g_chain_ev_stop = nullptr;
  if constexpr (kBackward) {
    LossArgs none = {};
    if (ev0 != nullptr)
      hipExtLaunchKernelGGL((mlp_chain_bwd_kernel<G, W>), dim3(grid), dim3(64 * W), static_cast<size_t>(lds_bytes), st,
                            ev0, ev1, 0, args, loss ? *loss : none);
    else
      hipLaunchKernelGGL((mlp_chain_bwd_kernel<G, W>), dim3(grid), dim3(64 * W), static_cast<size_t>(lds_bytes), st,
                         args, loss ? *loss : none);
  } else {
    if (ev0 != nullptr)
      hipExtLaunchKernelGGL((mlp_chain_fwd_kernel<G, HACT, W>), dim3(grid), dim3(64 * W),
                            static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, args);
    else
      hipLaunchKernelGGL((mlp_chain_fwd_kernel<G, HACT, W>), dim3(grid), dim3(64 * W),
                         static_cast<size_t>(lds_bytes), st, args);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

// The pipelined kernels address every weight matrix and bias vector through ONE buffer resource: fills w_base /
// w_bytes / w_off / b_off, false when the arrays are too far apart (separate allocations more than 1 GiB apart)
// or not 16-byte aligned rows (in % 4 != 0) - the unit-structured kernels take those.
static bool chain_pipe_fill(ChainArgs& args, bool with_bias) {
  uintptr_t lo = ~static_cast<uintptr_t>(0), hi = 0;
  // A lane's 16 bytes of a row's last chunk reach past the row when `in` is not a multiple of 16: into the next row
  // (finite values that meet zero activations) and, behind the LAST row, up to 48 bytes past the matrix.  Under one
  // resource over all arrays that read is not bounds-checked per matrix, so it must land in another array of this
  // network (the layout of the flat parameter arena: every weight matrix is followed by its bias) - not in whatever
  // follows a separately allocated tensor, which may be NaN bits or an unmapped page.
  {
    uintptr_t begin[2 * kChainMaxLayers], end[2 * kChainMaxLayers];
    int n = 0;
    for (int L = 0; L < args.num_layers; ++L) {
      const ChainLayer& ly = args.layer[L];
      begin[n] = reinterpret_cast<uintptr_t>(ly.w);
      end[n] = begin[n] + static_cast<uintptr_t>(ly.in) * ly.out * 4;
      ++n;
      if (with_bias && ly.bias) {
        begin[n] = reinterpret_cast<uintptr_t>(ly.bias);
        end[n] = begin[n] + static_cast<uintptr_t>(ly.out) * 4;
        ++n;
      }
    }
    for (int L = 0; L < args.num_layers; ++L) {
      const ChainLayer& ly = args.layer[L];
      if ((ly.in & 15) == 0) continue;
      uintptr_t covered = reinterpret_cast<uintptr_t>(ly.w) + static_cast<uintptr_t>(ly.in) * ly.out * 4;
      const uintptr_t need = covered + 48;
      for (bool grew = true; grew && covered < need;) {
        grew = false;
        for (int k = 0; k < n; ++k) {
          if (begin[k] <= covered + 16 && end[k] > covered) {     // (gaps of up to 16 bytes: the arena's alignment padding)
            covered = end[k];
            grew = true;
          }
        }
      }
      if (covered < need) return false;
    }
  }
  for (int L = 0; L < args.num_layers; ++L) {
    const ChainLayer& ly = args.layer[L];
    const uintptr_t w = reinterpret_cast<uintptr_t>(ly.w);
    if ((w & 15u) != 0 || (ly.in & 3) != 0) return false;
    lo = w < lo ? w : lo;
    hi = w + static_cast<uintptr_t>(ly.in) * ly.out * 4 > hi ? w + static_cast<uintptr_t>(ly.in) * ly.out * 4 : hi;
    if (with_bias) {
      const uintptr_t b = reinterpret_cast<uintptr_t>(ly.bias);
      if (b == 0 || (b & 3u) != 0) return false;
      lo = b < lo ? b : lo;
      hi = b + static_cast<uintptr_t>(ly.out) * 4 > hi ? b + static_cast<uintptr_t>(ly.out) * 4 : hi;
    }
  }
  lo &= ~static_cast<uintptr_t>(15);
  if (hi - lo >= static_cast<uintptr_t>(kOob)) return false;
  args.w_base = reinterpret_cast<const float*>(lo);
  args.w_bytes = static_cast<unsigned>(hi - lo);
  for (int L = 0; L < args.num_layers; ++L) {
    args.w_off[L] = static_cast<unsigned>(reinterpret_cast<uintptr_t>(args.layer[L].w) - lo);
    args.b_off[L] = with_bias ? static_cast<unsigned>(reinterpret_cast<uintptr_t>(args.layer[L].bias) - lo) : 0u;
  }
  return true;
}
template <int G, int HACT, int W = 4>
static int chain_launch_fwd_pipe(const ChainArgs& args_in, int lds_bytes, hipStream_t st) {
  ChainArgs args = args_in;
  int grid = static_cast<int>((args.rows + 16 * G - 1) / (16 * G));
  args.fwd_blocks = grid;
  if (args.pack.total_pairs > 0) grid += chain_bx_pack_blocks(args.pack);
  hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
  g_chain_ev_start = g_chain_ev_stop = nullptr;
  if (ev0 != nullptr)
    hipExtLaunchKernelGGL((mlp_chain_fwd_pipe_kernel<G, HACT, W>), dim3(grid), dim3(64 * W), static_cast<size_t>(lds_bytes), st,
                          ev0, ev1, 0, args);
  else
    hipLaunchKernelGGL((mlp_chain_fwd_pipe_kernel<G, HACT, W>), dim3(grid), dim3(64 * W), static_cast<size_t>(lds_bytes), st, args);
  RLG_RETURN_LAUNCH_STATUS();
}
// 16-row pipelined kernels: waves per workgroup (tools: RLG_PIPE1_WAVES=4|8)
static int chain_pipe1_waves() {
  // default 8: two waves per SIMD - at one wave the four-chunk look-ahead of the weight stream (512 MFMA cycles at one
  // 16-row group) is shorter than the loaded L2 latency (~900 cycles) and every chunk waits; 16 waves were slower
  // again (profiles/r4_rank_chain_probe.txt: 4,096 rows forward 30.6 / 26.5 / 26.0 us, backward 28.6 / 22.8 / 27.6 us)
  static const int w = [] { const char* e = std::getenv("RLG_PIPE1_WAVES"); const int v = e ? std::atoi(e) : 0; return (v == 4 || v == 16) ? v : 8; }();
  return w;
}

// Waves per workgroup.  Small minibatches (one data-parallel rank's 4,096 rows: 256 workgroups of 16
// rows, at most one or two per CU) are bound by the serial chain of a workgroup, not by MFMA
// throughput: 8 waves halve each wave's share of every layer and put two waves on every SIMD.
static int chain_waves(int G, long long rows) {
  static const int forced = [] {
    const char* e = std::getenv("RLG_CHAIN_WAVES");      // tools: A/B measurements
    return e ? std::atoi(e) : 0;
  }();
  if (G != 1) return 4;
  if (forced == 4 || forced == 8) return forced;
  const long long grid = (rows + 15) / 16;
  return grid <= kChainWideBlocks ? 8 : 4;
}

template <int G, bool kBackward, int W>
static int chain_launch_w(const ChainArgs& args, int lds_bytes, hipStream_t st, const LossArgs* loss) {
  if constexpr (!kBackward) {
    // forward: the ELU-or-identity network (every BASELINE configuration) gets its own instance
    bool elu_only = true;
    for (int L = 0; L < args.num_layers; ++L)
      elu_only = elu_only && (args.layer[L].act == kChElu || args.layer[L].act == kChIdentity);
    if (elu_only) return chain_launch_as<G, false, kChElu, W>(args, lds_bytes, st, loss);
  }
  return chain_launch_as<G, kBackward, kChAny, W>(args, lds_bytes, st, loss);
}

template <int G, bool kBackward>
static int chain_launch(const ChainArgs& args, int lds_bytes, hipStream_t st, const LossArgs* loss = nullptr) {
  if constexpr (G == 1) {
    if (chain_waves(1, args.rows) == 8) return chain_launch_w<1, kBackward, 8>(args, lds_bytes, st, loss);
  }
  return chain_launch_w<G, kBackward, 4>(args, lds_bytes, st, loss);
}


}  // namespace rlg

// ---------------------------------------------------------------------------------
// C ABI (declared in include/rlg_hip.h)
// ---------------------------------------------------------------------------------
extern "C" {

int rlg_mlp_chain_groups(long long rows, int requested, int direction) {
  return rlg::pick_groups(rows, requested, direction);
}

int rlg_mlp_chain_num_blocks(long long rows, int groups) {
  const int G = (groups == 1 || groups == 2 || groups == 4) ? groups : rlg::pick_groups(rows, 0);
  return static_cast<int>((rows + 16 * G - 1) / (16 * G));
}

// Raises the dynamic-LDS limit of every chain kernel once, outside any stream capture (the launchers
// would otherwise do it lazily on the first launch that needs more than 64 KiB).
int rlg_mlp_chain_prepare(void) {
  using namespace rlg;
  if (g_chain_prepared) return 0;
  const void* kernels[] = {
      reinterpret_cast<const void*>(mlp_chain_fwd_kernel<1, kChElu, 4>), reinterpret_cast<const void*>(mlp_chain_fwd_kernel<2, kChElu, 4>),
      reinterpret_cast<const void*>(mlp_chain_fwd_kernel<4, kChElu, 4>), reinterpret_cast<const void*>(mlp_chain_fwd_kernel<1, kChAny, 4>),
      reinterpret_cast<const void*>(mlp_chain_fwd_kernel<2, kChAny, 4>), reinterpret_cast<const void*>(mlp_chain_fwd_kernel<4, kChAny, 4>),
      reinterpret_cast<const void*>(mlp_chain_fwd_kernel<1, kChElu, 8>), reinterpret_cast<const void*>(mlp_chain_fwd_kernel<1, kChAny, 8>),
      reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<4, kChElu>), reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<4, kChAny>),
      reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<2, kChElu>), reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<2, kChAny>),
      reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChElu>), reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChAny>),
      reinterpret_cast<const void*>(mlp_chain_bwd_kernel<1, 4>), reinterpret_cast<const void*>(mlp_chain_bwd_kernel<2, 4>),
      reinterpret_cast<const void*>(mlp_chain_bwd_kernel<4, 4>), reinterpret_cast<const void*>(mlp_chain_bwd_kernel<1, 8>),
      reinterpret_cast<const void*>(mlp_chain_bwd_pipe_kernel<4>), reinterpret_cast<const void*>(mlp_chain_bwd_pipe_kernel<8>),
      reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChElu, 8>), reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChAny, 8>),
      reinterpret_cast<const void*>(mlp_chain_bwd_pipe_kernel<16>),
      reinterpret_cast<const void*>(mlp_chain_step_pipe_kernel<kChElu, 8>), reinterpret_cast<const void*>(mlp_chain_step_pipe_kernel<kChAny, 8>),
      reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChElu, 16>), reinterpret_cast<const void*>(mlp_chain_fwd_pipe_kernel<1, kChAny, 16>)};
  for (const void* k : kernels) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  if (const int e = chain_bx_prepare()) return e;
  if (const int e = chain_bx_fwd_prepare()) return e;
  g_chain_prepared = true;
  return 0;
}

// 1: rlg_mlp_chain_backward (direction 1) runs the split-bf16 kernel for this network and minibatch when it is given
// weight planes (and its arrays are 16-byte aligned); the caller packs planes only then.
int rlg_mlp_chain_bx_supported(int num_layers, const int* in_features, const int* out_features, long long rows,
                               int groups, int direction) {
  using namespace rlg;
  if (num_layers < 1 || num_layers > kChainMaxLayers || !chain_bx_enabled()) return 0;
  ChainArgs probe;
  probe.num_layers = num_layers;
  for (int L = 0; L < num_layers; ++L) {
    probe.layer[L].in = in_features[L];
    probe.layer[L].out = out_features[L];
  }
  if (direction == 0 || direction == 2) {
    if (!chain_bx_fwd_wanted(rows, groups, direction == 2)) return 0;
    if (chain_bx_plane_offsets(num_layers, in_features, out_features, 0, nullptr) >= static_cast<long long>(kOob)) return 0;
    return chain_bx_fwd_plan(probe) >= 0 ? 1 : 0;
  }
  if (direction != 1 || num_layers < 2) return 0;
  const int G = pick_groups(rows, groups, 1);
  if (G != 4) return 0;
  if (chain_bx_plane_offsets(num_layers, in_features, out_features, 1, nullptr) >= static_cast<long long>(kOob)) return 0;
  return chain_bx_bwd_lds(probe, G) >= 0 ? 1 : 0;
}

int rlg_mlp_chain_lds_bytes(int num_layers, const int* in_features, const int* out_features, int groups,
                            int direction) {
  int b = 0;
  if (num_layers < 1 || num_layers > rlg::kChainMaxLayers) return -1;
  return rlg::chain_lds(num_layers, in_features, out_features, groups, direction, &b);
}

// The next rlg_mlp_chain_forward / _backward launch (not inside a graph capture) is bracketed by these two HIP
// events (rlg_event_create); one-shot.  bench.py: in-epoch launch durations of the two dominant kernels.
int rlg_mlp_chain_time_next(void* ev_start, void* ev_stop) {
  rlg::g_chain_ev_start = static_cast<hipEvent_t>(ev_start);
  rlg::g_chain_ev_stop = static_cast<hipEvent_t>(ev_stop);
  return 0;
}

// The next rlg_mlp_chain_backward launch, if it runs the split-fp16 kernel, leaves per 64-row workgroup the largest magnitude
// of every dZ tensor it produces (and of the d heads it reads) in entries[(layer) * stride + workgroup] (csrc/bx_form.hpp) -
// what rlg_mlp_dw_gradient_maxima hands to the weight-gradient launch.  One-shot, like the timing events.
static float* g_chain_amax = nullptr;
static int g_chain_amax_stride = 0;
int rlg_mlp_chain_gradient_maxima(float* entries, int stride) {
  g_chain_amax = entries;
  g_chain_amax_stride = stride;
  return 0;
}

static long long* g_chain_dbg = nullptr;
// tools only: phase stamps of the next forward launches ([blocks][4][32] int64), nullptr to stop
int rlg_mlp_chain_debug_stamps(long long* buffer) {
  g_chain_dbg = buffer;
  return 0;
}

int rlg_mlp_chain_forward(int num_layers, const float* const* weights, const float* const* biases,
                          const int* in_features, const int* out_features, const int* acts,
                          float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                          const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                          const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                          double* rms_var_out, long long* rms_count_out,
                          long long rows, int groups, void* pack_backward_planes_or_null,
                          const void* weight_planes_or_null, void* stream) {
  using namespace rlg;
  if (rows <= 0) return 0;
  ChainArgs args;
  if (chain_fill(args, num_layers, weights, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  args.amax = nullptr;
  args.amax_stride = 0;
  // the backward launch's weight planes ride along as extra workgroups (with the tools' phase stamps, which index
  // their buffer by workgroup, as a launch of their own)
  if (pack_backward_planes_or_null != nullptr &&
      chain_bx_fill_pack(args.pack, num_layers, weights, in_features, out_features, 1, pack_backward_planes_or_null) &&
      g_chain_dbg != nullptr) {
    if (const int e = chain_bx_pack_launch(args.pack, static_cast<hipStream_t>(stream))) return e;
    args.pack.total_pairs = 0;
  }
  for (int L = 0; L < num_layers; ++L) {
    args.layer[L].bias = biases[L];
    args.layer[L].h = act_out[L];
    args.layer[L].ldh = act_ld[L];
  }
  if (act_out[num_layers - 1] == nullptr) return static_cast<int>(hipErrorInvalidValue);
  args.x = x;
  args.ldx = ldx;
  args.rms_mean = rms_mean;
  args.rms_var = rms_mean ? rms_var : nullptr;
  args.rms_eps = rms_eps;
  args.rms_batch = rms_mean ? rms_batch : nullptr;
  if (args.rms_batch) {
    if (!rms_count || !rms_mean_out || !rms_var_out || !rms_count_out || rms_mean_out == rms_mean ||
        rms_var_out == rms_var || rms_count_out == rms_count)
      return static_cast<int>(hipErrorInvalidValue);      // the fold publishes into a second buffer set
  }
  args.rms_count = rms_count;
  args.rms_mean_out = rms_mean_out;
  args.rms_var_out = rms_var_out;
  args.rms_count_out = rms_count_out;
  args.xn = xn_out;
  args.rows = rows;
  args.dbg = g_chain_dbg;
  bool training = false;
  for (int L = 0; L + 1 < num_layers; ++L) training = training || act_out[L] != nullptr;
  const int G = pick_groups(rows, groups, training ? 2 : 0);
  int b_floats = 0;
  const int lds_bytes = chain_lds(num_layers, in_features, out_features, G, 0, &b_floats);
  if (lds_bytes < 0) return static_cast<int>(hipErrorInvalidValue);
  args.lds_b_floats = b_floats;
  args.lds_split_floats = lds_bytes / 4 - chain_split_floats(G);
  {
    static const int off = [] { const char* e = std::getenv("RLG_CHAIN_KSPLIT"); return (e && std::atoi(e) == 0) ? 1 : 0; }();
    args.no_ksplit = off;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // split-bf16 products on pre-split weight planes (mlp_chain_bx_fwd.hip): 64-row workgroups
  if (weight_planes_or_null != nullptr && chain_bx_fwd_wanted(rows, groups, training)) {
    ChainArgs bx = args;
    const long long total = chain_bx_plane_offsets(num_layers, in_features, out_features, 0, bx.p_off);
    bx.planes = weight_planes_or_null;
    bx.planes_bytes = static_cast<unsigned>(total);
    const int bx_lds = chain_bx_fwd_plan(bx);
    if (bx_lds >= 0 && total < static_cast<long long>(kOob) && chain_bx_fwd_eligible(bx)) {
      hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
      g_chain_ev_start = g_chain_ev_stop = nullptr;
      return chain_bx_launch_fwd(bx, bx_lds, st, ev0, ev1);
    }
  }
  if ((G >= 2 || chain_pipe1_enabled()) && chain_pipe_enabled() && chain_pipe_fill(args, true)) {
    bool elu_only = true;
    for (int L = 0; L < num_layers; ++L) elu_only = elu_only && (acts[L] == kChElu || acts[L] == kChIdentity);
    if (G == 1 && chain_pipe1_waves() == 8)
      return elu_only ? chain_launch_fwd_pipe<1, kChElu, 8>(args, lds_bytes, st) : chain_launch_fwd_pipe<1, kChAny, 8>(args, lds_bytes, st);
    if (G == 1 && chain_pipe1_waves() == 16)
      return elu_only ? chain_launch_fwd_pipe<1, kChElu, 16>(args, lds_bytes, st) : chain_launch_fwd_pipe<1, kChAny, 16>(args, lds_bytes, st);
    if (G == 1) return elu_only ? chain_launch_fwd_pipe<1, kChElu>(args, lds_bytes, st) : chain_launch_fwd_pipe<1, kChAny>(args, lds_bytes, st);
    if (G == 4) return elu_only ? chain_launch_fwd_pipe<4, kChElu>(args, lds_bytes, st) : chain_launch_fwd_pipe<4, kChAny>(args, lds_bytes, st);
    return elu_only ? chain_launch_fwd_pipe<2, kChElu>(args, lds_bytes, st) : chain_launch_fwd_pipe<2, kChAny>(args, lds_bytes, st);
  }
  if (G == 4) return chain_launch<4, false>(args, lds_bytes, st);
  if (G == 2) return chain_launch<2, false>(args, lds_bytes, st);
  return chain_launch<1, false>(args, lds_bytes, st);
}

int rlg_mlp_chain_backward(int num_layers, const float* const* weights, const int* in_features,
                           const int* out_features, const int* acts, const float* const* act_in,
                           const long long* act_ld, const float* d_out, long long ld_dout,
                           float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                           const rlg_ppo_loss_desc* ppo_loss, long long rows, int groups,
                           const void* weight_planes_or_null, void* stream) {
  using namespace rlg;
  float* const amax = g_chain_amax;
  const int amax_stride = g_chain_amax_stride;
  g_chain_amax = nullptr;
  if (rows <= 0) return 0;
  if (num_layers < 2) return static_cast<int>(hipErrorInvalidValue);
  ChainArgs args;
  if (chain_fill(args, num_layers, weights, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  args.amax = nullptr;
  args.amax_stride = 0;
  for (int L = 0; L + 1 < num_layers; ++L) {
    args.layer[L].h = const_cast<float*>(act_in[L]);
    args.layer[L].ldh = act_ld[L];
    args.layer[L].dz = dz_out[L];
    args.layer[L].lddz = dz_ld[L];
    args.layer[L].bias_partials = bias_partials ? bias_partials[L] : nullptr;
    if (act_in[L] == nullptr || dz_out[L] == nullptr) return static_cast<int>(hipErrorInvalidValue);
  }
  args.x = d_out;
  args.ldx = ld_dout;
  args.rms_mean = args.rms_var = nullptr;
  args.rms_eps = 0.0f;
  args.rms_batch = nullptr;
  args.rms_count = nullptr;
  args.rms_mean_out = args.rms_var_out = nullptr;
  args.rms_count_out = nullptr;
  args.xn = nullptr;
  args.rows = rows;
  const int G = pick_groups(rows, groups, 1);
  int b_floats = 0;
  int lds_bytes = chain_lds(num_layers, in_features, out_features, G, 1, &b_floats);
  if (lds_bytes < 0) return static_cast<int>(hipErrorInvalidValue);
  args.lds_b_floats = b_floats;
  LossArgs loss = {};
  args.with_loss = ppo_loss ? 1 : 0;
  if (ppo_loss) {
    const rlg_ppo_loss_desc& d = *ppo_loss;
    if (d.minibatch != rows || d.actions_num <= 0 || (d.mask_or_null && !d.mask_sum_or_null) || !d.partials ||
        !d.mu || !d.values || !d.d_mu || !d.d_values)
      return static_cast<int>(hipErrorInvalidValue);
    loss.mu = d.mu;
    loss.logstd = d.logstd;
    loss.values = d.values;
    loss.actions = d.actions;
    loss.old_neglogp = d.old_neglogp;
    loss.advantages = d.advantages;
    loss.old_values = d.old_values;
    loss.returns = d.returns;
    loss.old_mu = d.old_mu;
    loss.old_sigma = d.old_sigma;
    loss.mask = d.mask_or_null;
    loss.mask_sum = d.mask_sum_or_null;
    loss.d_mu = d.d_mu;
    loss.d_values = d.d_values;
    loss.partials = d.partials;
    loss.mb = d.minibatch;
    loss.A = d.actions_num;
    loss.ld_mu = d.ld_mu;
    loss.ld_val = d.ld_values;
    loss.ld_dmu = d.ld_d_mu;
    loss.ld_dval = d.ld_d_values;
    loss.e_clip = d.e_clip;
    loss.critic_coef = d.critic_coef;
    loss.bounds_coef = d.bounds_coef;
    loss.clip_value = d.clip_value;
    loss.smooth = d.use_smooth_clamp;
    loss.bound_kind = d.bound_kind;
    loss.write_back = d.write_back;
    const int need = static_cast<int>(ppo_loss_lds_bytes(16 * G, d.actions_num, 512));
    if (need > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    if (need > lds_bytes) lds_bytes = need;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const LossArgs* lp = ppo_loss ? &loss : nullptr;
  // split-bf16 products on pre-split weight planes (mlp_chain_bx.hip): 64-row workgroups, aligned activations
  args.planes = nullptr;
  if (weight_planes_or_null != nullptr && G == 4 && chain_bx_enabled()) {
    const long long total = chain_bx_plane_offsets(num_layers, in_features, out_features, 1, args.p_off);
    ChainArgs bx = args;
    bx.dbg = g_chain_dbg;
    bx.planes = weight_planes_or_null;
    bx.planes_bytes = static_cast<unsigned>(total);
    if (amax != nullptr && amax_stride >= (rows + 16 * G - 1) / (16 * G)) {
      bx.amax = amax;
      bx.amax_stride = amax_stride;
    }
    int bx_lds = chain_bx_bwd_lds(bx, G);
    if (bx_lds >= 0 && total < static_cast<long long>(kOob) && chain_bx_bwd_eligible(bx)) {
      bx.bx_handoff_off = -1;
      bx.bx_handoff_ld = 0;
      if (ppo_loss) {
        const int need = static_cast<int>(ppo_loss_lds_bytes(16 * G, ppo_loss->actions_num, 512));
        if (need > bx_lds) bx_lds = need;
        // the loss tile writes exactly the d heads array this launch reads (column 0: d value, 1 .. A: d mu): hand
        // them over in LDS, behind everything else
        const rlg_ppo_loss_desc& d = *ppo_loss;
        const int w = out_features[num_layers - 1];
        const int hld = (w + 1) | 1;
        const int hbytes = 16 * G * hld * 4;
        if (w == 1 + d.actions_num && d.d_values == d_out && d.d_mu == d_out + 1 && d.ld_d_values == ld_dout &&
            d.ld_d_mu == ld_dout && d.actions_num <= 32 && bx_lds + hbytes + kBxBwdScratch <= 160 * 1024) {
          bx.bx_handoff_off = (bx_lds + 15) & ~15;
          bx.bx_handoff_ld = hld;
          bx_lds = bx.bx_handoff_off + hbytes;
        }
      }
      bx.bx_scales_off = (bx_lds + 15) & ~15;                  // (row scales of the d heads tile, fp16 form)
      bx_lds = bx.bx_scales_off + 16 * G * 4 + kChainMaxLayers * 4 * 4;      // (+ the waves' gradient maxima of every layer)
      hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
      g_chain_ev_start = g_chain_ev_stop = nullptr;
      if (bx_lds <= 160 * 1024) return chain_bx_launch_bwd(bx, G, bx_lds, st, lp, ev0, ev1);
      g_chain_ev_start = ev0;
      g_chain_ev_stop = ev1;
    }
  }
  if (G == 4) return chain_launch<4, true>(args, lds_bytes, st, lp);
  if (G == 2) return chain_launch<2, true>(args, lds_bytes, st, lp);
  // 16-row workgroups: the pipelined kernel when every H / dZ array takes 16-byte row accesses
  if (chain_pipe1_enabled() && chain_pipe_enabled()) {
    bool ok = g_chain_dbg == nullptr;
    for (int L = 0; L + 1 < num_layers && ok; ++L) {
      const ChainLayer& ly = args.layer[L];
      ok = vec4_ok_host(ly.h, ly.ldh) && vec4_ok_host(ly.dz, ly.lddz) && (ly.out & 3) == 0 && ly.ldh < (1 << 20) && ly.lddz < (1 << 20);
    }
    if (ok) {
      const int grid = static_cast<int>((rows + 15) / 16);
      hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
      g_chain_ev_start = g_chain_ev_stop = nullptr;
      LossArgs none = {};
      if (chain_pipe1_waves() == 16) {
        if (ev0 != nullptr)
          hipExtLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<16>, dim3(grid), dim3(1024), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0,
                                args, lp ? *lp : none);
        else
          hipLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<16>, dim3(grid), dim3(1024), static_cast<size_t>(lds_bytes), st, args,
                             lp ? *lp : none);
      } else if (chain_pipe1_waves() == 8) {
        if (ev0 != nullptr)
          hipExtLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<8>, dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0,
                                args, lp ? *lp : none);
        else
          hipLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<8>, dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, args,
                             lp ? *lp : none);
      } else {
        if (ev0 != nullptr)
          hipExtLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<4>, dim3(grid), dim3(256), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0,
                                args, lp ? *lp : none);
        else
          hipLaunchKernelGGL(mlp_chain_bwd_pipe_kernel<4>, dim3(grid), dim3(256), static_cast<size_t>(lds_bytes), st, args,
                             lp ? *lp : none);
      }
      RLG_RETURN_LAUNCH_STATUS();
    }
  }
  return chain_launch<1, true>(args, lds_bytes, st, lp);
}


// Forward + PPO loss + backward of a minibatch as ONE launch (mlp_chain_step_pipe_kernel): the arguments of
// rlg_mlp_chain_forward (training form: every act_out given) and of rlg_mlp_chain_backward with a loss descriptor.
// hipErrorNotSupported when the shape is outside the kernel's envelope (the caller then issues the two launches):
// minibatches of >= 16,384 rows (they run the split-bf16 kernels), weights not in one arena, H / dZ rows not 16-byte
// aligned, more than 32 actions, RLG_CHAIN_PIPE1=0 / RLG_CHAIN_STEP1=0.
int rlg_mlp_chain_step(int num_layers, const float* const* weights, const float* const* biases,
                       const int* in_features, const int* out_features, const int* acts,
                       float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                       const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                       const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                       double* rms_var_out, long long* rms_count_out, float* d_out, long long ld_dout,
                       float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                       const rlg_ppo_loss_desc* ppo_loss, long long rows, void* stream) {
  using namespace rlg;
  static const bool enabled = [] { const char* e = std::getenv("RLG_CHAIN_STEP1"); return !(e && std::atoi(e) == 0); }();
  if (rows <= 0) return 0;
  if (!enabled || !chain_pipe1_enabled() || !chain_pipe_enabled() || chain_pipe1_waves() != 8 || g_chain_dbg != nullptr ||
      num_layers < 2 || ppo_loss == nullptr || pick_groups(rows, 0, 2) != 1 || pick_groups(rows, 0, 1) != 1)
    return static_cast<int>(hipErrorNotSupported);
  // one 8-wave workgroup per CU (200 registers per wave): beyond one round of workgroups the two separate launches,
  // which run two workgroups per CU, are faster (8,192 rows: 52.3 vs 50.0 ms per rank epoch, profiles/r4_rank_shapes.txt)
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return static_cast<int>(hipErrorNotSupported);
    if ((rows + 15) / 16 > cus) return static_cast<int>(hipErrorNotSupported);
  }
  // ---- forward arguments (as rlg_mlp_chain_forward)
  ChainArgs fa;
  if (chain_fill(fa, num_layers, weights, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  for (int L = 0; L < num_layers; ++L) {
    fa.layer[L].bias = biases[L];
    fa.layer[L].h = act_out[L];
    fa.layer[L].ldh = act_ld[L];
    if (act_out[L] == nullptr) return static_cast<int>(hipErrorNotSupported);       // training form only
  }
  fa.x = x;
  fa.ldx = ldx;
  fa.rms_mean = rms_mean;
  fa.rms_var = rms_mean ? rms_var : nullptr;
  fa.rms_eps = rms_eps;
  fa.rms_batch = rms_mean ? rms_batch : nullptr;
  if (fa.rms_batch) {
    if (!rms_count || !rms_mean_out || !rms_var_out || !rms_count_out || rms_mean_out == rms_mean ||
        rms_var_out == rms_var || rms_count_out == rms_count)
      return static_cast<int>(hipErrorInvalidValue);
  }
  fa.rms_count = rms_count;
  fa.rms_mean_out = rms_mean_out;
  fa.rms_var_out = rms_var_out;
  fa.rms_count_out = rms_count_out;
  fa.xn = xn_out;
  fa.rows = rows;
  fa.dbg = nullptr;
  int fb = 0;
  const int fwd_lds = chain_lds(num_layers, in_features, out_features, 1, 0, &fb);
  if (fwd_lds < 0) return static_cast<int>(hipErrorNotSupported);
  fa.lds_b_floats = fb;
  fa.lds_split_floats = fwd_lds / 4 - chain_split_floats(1);
  fa.no_ksplit = 0;
  if (!chain_pipe_fill(fa, true)) return static_cast<int>(hipErrorNotSupported);
  // ---- backward arguments (as rlg_mlp_chain_backward); H of the hidden layers = what the forward half writes
  ChainArgs ba;
  if (chain_fill(ba, num_layers, weights, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  for (int L = 0; L + 1 < num_layers; ++L) {
    ba.layer[L].h = act_out[L];
    ba.layer[L].ldh = act_ld[L];
    ba.layer[L].dz = dz_out[L];
    ba.layer[L].lddz = dz_ld[L];
    ba.layer[L].bias_partials = bias_partials ? bias_partials[L] : nullptr;
    if (dz_out[L] == nullptr) return static_cast<int>(hipErrorInvalidValue);
    const ChainLayer& ly = ba.layer[L];
    if (!(vec4_ok_host(ly.h, ly.ldh) && vec4_ok_host(ly.dz, ly.lddz) && (ly.out & 3) == 0 && ly.ldh < (1 << 20) && ly.lddz < (1 << 20)))
      return static_cast<int>(hipErrorNotSupported);
  }
  ba.x = d_out;
  ba.ldx = ld_dout;
  ba.rms_mean = ba.rms_var = nullptr;
  ba.rms_eps = 0.0f;
  ba.rms_batch = nullptr;
  ba.rms_count = nullptr;
  ba.rms_mean_out = ba.rms_var_out = nullptr;
  ba.rms_count_out = nullptr;
  ba.xn = nullptr;
  ba.rows = rows;
  ba.with_loss = 1;
  int bb = 0;
  int bwd_lds = chain_lds(num_layers, in_features, out_features, 1, 1, &bb);
  if (bwd_lds < 0) return static_cast<int>(hipErrorNotSupported);
  ba.lds_b_floats = bb;
  const rlg_ppo_loss_desc& d = *ppo_loss;
  if (d.minibatch != rows || d.actions_num <= 0 || d.actions_num > 4 * kQuadK || (d.mask_or_null && !d.mask_sum_or_null) ||
      !d.partials || !d.mu || !d.values || !d.d_mu || !d.d_values)
    return d.actions_num > 4 * kQuadK ? static_cast<int>(hipErrorNotSupported) : static_cast<int>(hipErrorInvalidValue);
  LossArgs loss = {};
  loss.mu = d.mu;
  loss.logstd = d.logstd;
  loss.values = d.values;
  loss.actions = d.actions;
  loss.old_neglogp = d.old_neglogp;
  loss.advantages = d.advantages;
  loss.old_values = d.old_values;
  loss.returns = d.returns;
  loss.old_mu = d.old_mu;
  loss.old_sigma = d.old_sigma;
  loss.mask = d.mask_or_null;
  loss.mask_sum = d.mask_sum_or_null;
  loss.d_mu = d.d_mu;
  loss.d_values = d.d_values;
  loss.partials = d.partials;
  loss.mb = d.minibatch;
  loss.A = d.actions_num;
  loss.ld_mu = d.ld_mu;
  loss.ld_val = d.ld_values;
  loss.ld_dmu = d.ld_d_mu;
  loss.ld_dval = d.ld_d_values;
  loss.e_clip = d.e_clip;
  loss.critic_coef = d.critic_coef;
  loss.bounds_coef = d.bounds_coef;
  loss.clip_value = d.clip_value;
  loss.smooth = d.use_smooth_clamp;
  loss.bound_kind = d.bound_kind;
  loss.write_back = d.write_back;
  const int need = static_cast<int>(ppo_loss_lds_bytes(16, d.actions_num, 512));
  int lds_bytes = fwd_lds > bwd_lds ? fwd_lds : bwd_lds;
  if (need > lds_bytes) lds_bytes = need;
  if (lds_bytes > 160 * 1024) return static_cast<int>(hipErrorNotSupported);
  bool elu_only = true;
  for (int L = 0; L < num_layers; ++L) elu_only = elu_only && (acts[L] == kChElu || acts[L] == kChIdentity);
  const int grid = static_cast<int>((rows + 15) / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev0 = g_chain_ev_start, ev1 = g_chain_ev_stop;
  g_chain_ev_start = g_chain_ev_stop = nullptr;
  if (elu_only) {
    if (ev0 != nullptr)
      hipExtLaunchKernelGGL((mlp_chain_step_pipe_kernel<kChElu, 8>), dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, fa, ba, loss);
    else
      hipLaunchKernelGGL((mlp_chain_step_pipe_kernel<kChElu, 8>), dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, fa, ba, loss);
  } else {
    if (ev0 != nullptr)
      hipExtLaunchKernelGGL((mlp_chain_step_pipe_kernel<kChAny, 8>), dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, fa, ba, loss);
    else
      hipLaunchKernelGGL((mlp_chain_step_pipe_kernel<kChAny, 8>), dim3(grid), dim3(512), static_cast<size_t>(lds_bytes), st, fa, ba, loss);
  }
  RLG_RETURN_LAUNCH_STATUS();
}


}  // extern "C"

// ---- what mlp_chain_lean.hip needs of this file's launch state (mlp_chain_shared.hpp)
namespace rlg {
long long* chain_debug_stamps() { return g_chain_dbg; }
void chain_take_events(hipEvent_t* ev_start, hipEvent_t* ev_stop) {
  *ev_start = g_chain_ev_start;
  *ev_stop = g_chain_ev_stop;
  g_chain_ev_start = g_chain_ev_stop = nullptr;
}
void chain_take_gradient_maxima(float** entries, int* stride) {
  *entries = ::g_chain_amax;
  *stride = ::g_chain_amax_stride;
  ::g_chain_amax = nullptr;
}
}  // namespace rlg
