// The lean 16-row kernels of the fused MLP chain (round 4; a file of their own since round 5): forward, backward and the
// one-launch forward + PPO loss + backward step for minibatches / rollouts of < 16,384 rows on exact fp32 products
// (v_mfma_f32_16x16x4_f32) - a data-parallel rank's 4,096 - 8,192-row minibatches, BASELINE configs[1] and [4] - with the
// weights as fp32 FRAGMENTS in the order each wave consumes them.  Same products in the same order as the pipelined
// kernels of mlp_chain.hip (bit-identical results, tests/test_mlp_chain_gpu.py); MFMA mapping, tile layout and what the
// launches replace in the reference: see the head of mlp_chain.hip.

#include "mlp_chain_shared.hpp"
#include "optim_common.hpp"

// Round 6 (second half): the lean kernels on the split-fp16 form of csrc/bx_form.hpp - same streams, same tiles, same loop:
// a GROUP (32 k values) was two 1-KiB fp32 chunks of 16 k values and 8 x v_mfma_f32_16x16x4_f32; it is now the two fp16
// planes of the same 32 values (plane 0 where the even chunk was, plane 1 where the odd one was - byte for byte the same
// buffers and LDS tiles) and 3 x v_mfma_f32_16x16x32_f16.  Lane l holds, per plane, 8 values: features 4q .. 4q + 3 of the
// group's first 16-feature block (elements 0 - 3) and of its second (4 - 7), q = l >> 4 - what the fp32 chunks held as
// their 4 floats each, so the pack launch splits exactly the values it used to copy, and an epilogue writes 8 bytes per
// plane where it wrote 16 bytes of fp32.  Scales as in the 64-row kernels: weights 2^6, hidden activations 2^4, normalised
// observations 2^12, raw observations and gradient rows by their own maxima (a 16-row tile: a lane's row is lane & 15).
// RLG_LEAN_F16=0 (or a -DRLG_BX_F16=0 build): exact fp32 products, bit-identical to the pipelined kernels of mlp_chain.hip.
#ifndef RLG_LEAN_F16
#define RLG_LEAN_F16 RLG_BX_F16
#endif

namespace rlg {

#if RLG_LEAN_F16
// block `ob` (16 features) of a tile in the group layout: this lane's 4 features -> 8 bytes of each plane
__device__ __forceinline__ void lean_put_planes(float* tile, int ob, int lane, const f32x4& v, float scale) {
  unsigned pl[kBxPlanes][2];
  bx_split4(v, scale, pl);
  float* base = tile + ((ob & ~1) * 64 + lane) * 4 + (ob & 1) * 2;
  *reinterpret_cast<uint2*>(base) = make_uint2(pl[0][0], pl[0][1]);
  *reinterpret_cast<uint2*>(base + 256) = make_uint2(pl[1][0], pl[1][1]);
}
__device__ __forceinline__ void lean_zero_block(float* tile, int ob, int lane) {
  float* base = tile + ((ob & ~1) * 64 + lane) * 4 + (ob & 1) * 2;
  *reinterpret_cast<uint2*>(base) = make_uint2(0u, 0u);
  *reinterpret_cast<uint2*>(base + 256) = make_uint2(0u, 0u);
}
// one group of the reduction: three plane products (small terms into acc1)
__device__ __forceinline__ void lean_group_mfma(const f32x4 (&aq)[2], const f32x4 (&bq)[2], f32x4& acc0, f32x4& acc1) {
  const u32x4 a0 = __builtin_bit_cast(u32x4, aq[0]), a1 = __builtin_bit_cast(u32x4, aq[1]);
  const u32x4 b0 = __builtin_bit_cast(u32x4, bq[0]), b1 = __builtin_bit_cast(u32x4, bq[1]);
  acc1 = bx_mfma(a1, b0, acc1);
  acc1 = bx_mfma(a0, b1, acc1);
  acc0 = bx_mfma(a0, b0, acc0);
}

// Forward prologue of the fp16 form (cf. chain_fwd_prologue<1, W>): the observation tile of 16 rows -> planes in tile_a,
// normalised on the way; returns the scale of this lane's row (lane & 15).
template <int W>
__device__ __forceinline__ float lean_fwd_prologue_f16(const ChainArgs& a, float* tile_a, float* tile_b, long long row0, int lane,
                                                       int wave, int& stamp) {
  const int in0 = a.layer[0].in;
  const int in0p = (in0 + 3) & ~3;
  const bool norm = a.rms_mean != nullptr;
  const int KC0 = (in0 + 15) >> 4;
  const int ng = (KC0 + 1) >> 1;                     // groups of two 16-feature blocks
  const bool xv = vec4_ok(a.x, a.ldx);
  const bool xnv = a.xn != nullptr && vec4_ok(a.xn, in0);
  const int q4 = 4 * (lane >> 4);
  const long long row = row0 + (lane & 15);
  constexpr int kProBatch = 4;
  f32x4 xlo[kProBatch], xhi[kProBatch];
  auto load_groups = [&](int u0) {
#pragma unroll
    for (int k = 0; k < kProBatch; ++k) {
      const int u = u0 + k * W;
      xlo[k] = xhi[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (u < ng && row < a.rows) {
        xlo[k] = load_row4(a.x, a.ldx, row, u * 32 + q4, in0, xv);
        xhi[k] = load_row4(a.x, a.ldx, row, u * 32 + 16 + q4, in0, xv);
      }
    }
  };
  auto put_groups = [&](int u0, float scale) {
#pragma unroll
    for (int k = 0; k < kProBatch; ++k) {
      const int u = u0 + k * W;
      if (u < ng) {
        f32x4 v[2] = {xlo[k], xhi[k]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f = u * 32 + 16 * h + q4;
          if (row < a.rows) {
            if (norm) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (f + e < in0) v[h][e] = clamp_nan((v[h][e] - tile_b[f + e]) / tile_b[in0p + f + e], -5.0f, 5.0f);
              }
            }
            if (a.xn && f < in0) store_row4(a.xn, in0, row, f, in0, v[h], xnv);
          }
        }
        const float x[8] = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
        u32x4 plane[kBxPlanes];
        bx_split8(x, scale, plane);
        *reinterpret_cast<u32x4*>(tile_a + ((2 * u) * 64 + lane) * 4) = plane[0];
        *reinterpret_cast<u32x4*>(tile_a + ((2 * u + 1) * 64 + lane) * 4) = plane[1];
      }
    }
  };
  load_groups(wave);
  float scale = kBxScaleObsNorm;
  if (norm) {
    chain_norm_stats<W>(a, tile_b, in0, in0p);
    __syncthreads();
  } else {
    // raw observations have no bound: the row's scale from its largest magnitude (every wave reads the whole row - the
    // loads of the groups it splits find it in the cache)
    float mine = 0.0f;
    if (row < a.rows) {
      for (int c = 0; c < KC0; ++c) {
        const f32x4 t = load_row4(a.x, a.ldx, row, c * 16 + q4, in0, xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) mine = __builtin_fmaxf(mine, bx_finite_abs(t[e]));
      }
    }
    scale = bx_row_scale(mine);
  }
  put_groups(wave, scale);
  for (int u0 = wave + W * kProBatch; u0 < ng; u0 += W * kProBatch) {
    load_groups(u0);
    put_groups(u0, scale);
  }
  chain_stamp(a.dbg, wave, stamp);
  __syncthreads();
  return scale;
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Lean 16-row forward (round 4, experimental - the C entry rlg_mlp_chain_forward_lean; tools/exp/lean_probe.py).
// profiles/r4_rank_chain_ablation.txt: at a data-parallel rank's 4,096-row minibatch the pipelined 16-row forward is bound
// by its own skeleton (per-unit geometry, out-of-range selects, conditional tail chunks), not by memory.  This form has
// none of that:
//   * the weights come as fp32 FRAGMENTS in the order each of the 8 waves consumes them (rlg_mlp_chain_pack_frags): wave w's
//     stream = layer 0's units, layer 1's units, ... ; a unit = one 16-feature block = KC2 chunks of 1 KiB (lane l: W[16 ob +
//     (l & 15)][16 c + 4 (l >> 4) .. + 3]), zero padded to an even number KC2 of chunks (the activation tiles carry zero
//     chunks to match);
//   * a group = 2 chunks = 8 MFMAs; every (wave, layer) segment is padded with zero fragments to a multiple of 4 groups,
//     so ONE scalar offset walks the whole stream (+ 2 KiB per group), the 4 fragment slots rotate with the unrolled loop,
//     requests run 3 groups ahead and cross unit and layer boundaries by themselves;
//   * B fragments (the activations, LDS) one group ahead; a unit's bias fragment is requested when the unit starts; an
//     epilogue only when a unit's last group has issued.  Same products in the same order as the pipelined kernel
//     (even steps of a chunk into one accumulator, odd steps into the other, then their sum, then the bias): the
//     results are bit-identical to it.
constexpr int kLeanW = 8;
struct LeanArgs {
  const float* wf;
  unsigned wf_bytes;
  unsigned stream_off[kLeanW];                        // byte offset of wave w's stream
  unsigned short groups4[kLeanW][kChainMaxLayers];    // groups of the (wave, layer) segment incl. its padding
  unsigned char nunits[kLeanW][kChainMaxLayers];
  unsigned char kc2[kChainMaxLayers];                 // chunks per unit
  unsigned char full[kChainMaxLayers];                // whole blocks per wave; the rest: one more unit for waves < rem
  int tile_b_floats;
};

__device__ __forceinline__ f32x4 buf_load4_s(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int HACT>
__device__ __forceinline__ void chain_fwd_lean_body(const ChainArgs& a, const LeanArgs& la, float* lds) {
  constexpr int W = kLeanW;
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4);
  const long long row0 = static_cast<long long>(blockIdx.x) * 16;
  float* tile_a = lds;
  float* tile_b = lds + la.tile_b_floats;
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);
  const rsrc_t wr = make_rsrc(la.wf, la.wf_bytes);
  const unsigned voff = static_cast<unsigned>(lane) * 16u;
  unsigned soff = pin_s(static_cast<int>(la.stream_off[wave]));
  f32x4 aq[4][2], bq[2][2];
  auto request = [&](int slot) {
    aq[slot][0] = buf_load4_s(wr, voff, soff);
    aq[slot][1] = buf_load4_s(wr, voff, soff + 1024u);
    soff += 2048u;
  };
  request(0);
  request(1);
  request(2);
#if RLG_LEAN_F16
  auto zero_chunk = [&](float* tile, int c) { lean_zero_block(tile, c, lane); };        // (block c of the group layout)
#else
  auto zero_chunk = [&](float* tile, int c) { *reinterpret_cast<f32x4*>(tile + (c * 64 + lane) * 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
#endif
  int stamp = 0;
  chain_stamp(a.dbg, wave, stamp);
  float scale_in = 1.0f;          // fp16 form: scale of this lane's row in the tile the current layer reads
  {
    const int in0 = pin_s(a.layer[0].in);
    const int KC0 = (in0 + 15) >> 4;
#if RLG_LEAN_F16
    if (wave == W - 1) {
      // (behind the last group the prologue writes: its second block when KC0 is odd is written - as zeros - by the prologue)
      for (int c = (KC0 + 1) & ~1; c < la.kc2[0]; ++c) zero_chunk(tile_a, c);
    }
    scale_in = lean_fwd_prologue_f16<W>(a, tile_a, tile_b, row0, lane, wave, stamp);     // (ends with a barrier)
#else
    if (wave == W - 1) {
      for (int c = KC0; c < la.kc2[0]; ++c) zero_chunk(tile_a, c);
    }
    chain_fwd_prologue<1, W>(a, tile_a, tile_b, row0, lane, wave, stamp);     // (ends with a barrier)
#endif
  }
  chain_stamp(a.dbg, wave, stamp);

  float* tin = tile_a;
  float* tout = tile_b;
  for (int L = 0; L < num_layers; ++L) {
    const bool last = (L == num_layers - 1);
    const int l_out = pin_s(a.layer[L].out), l_act = pin_s(a.layer[L].act);
    float* l_h = pin_s(a.layer[L].h);
    const long long l_ldh = pin_s(a.layer[L].ldh);
    const bool h_on = l_h != nullptr;
    const bool h_vec = pin_s(static_cast<int>(h_on && vec4_ok(l_h, l_ldh))) != 0;
    const int KC2 = pin_s(static_cast<int>(la.kc2[L]));
    const int gpu = KC2 >> 1;                                        // groups per unit
    const int nun = pin_s(static_cast<int>(la.nunits[wave][L]));
    const int T4 = pin_s(static_cast<int>(la.groups4[wave][L]));
    const int full = pin_s(static_cast<int>(la.full[L]));
    if (!last && wave == W - 1) {
      // the padding of the tile this layer produces: chunks behind its last block
      const int NOB = (l_out + 15) >> 4;
      for (int c = NOB; c < la.kc2[L + 1]; ++c) zero_chunk(tout, c);
    }
    const float* l_bias = pin_s(a.layer[L].bias);
    const rsrc_t br = make_rsrc(l_bias, static_cast<unsigned>(l_out) * 4u);
    const bool bias_fast = pin_s(static_cast<int>(aligned16(l_bias) && (l_out & 3) == 0)) != 0;
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    auto unit_block = [&](int u) -> int { return u < full ? wave * full + u : W * full + wave; };
    // the 4 bias values of a lane's features (zero beyond the layer's width: the resource's bound)
    auto load_bias = [&](int u) {
      const int f = unit_block(u) * 16 + q4;
      if (bias_fast) {
        bv = buf_load4(br, (u < nun && l_bias != nullptr) ? static_cast<unsigned>(f) * 4u : kOob);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = buf_load1(br, (u < nun && l_bias != nullptr) ? static_cast<unsigned>(f + e) * 4u : kOob);
      }
    };
    const float* bp = tin + lane * 4;
    int cw = 0;                        // chunk (inside its unit) of the next group whose B fragments are read
    auto read_b = [&](int slot) {
      bq[slot][0] = *reinterpret_cast<const f32x4*>(bp + cw * 256);
      bq[slot][1] = *reinterpret_cast<const f32x4*>(bp + (cw + 1) * 256);
      cw += 2;
      if (cw == KC2) cw = 0;
    };
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
    int gu = 0, unit = 0;
#if RLG_LEAN_F16
    const float inv = 1.0f / (kBxScaleW * scale_in);
#endif
    auto epilogue = [&]() {
      const int ob = unit_block(unit);
      const int f = ob * 16 + q4;
#if RLG_LEAN_F16
      f32x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = __builtin_fmaf(acc0[e] + acc1[e], inv, bv[e]);
      const f32x4 v = chain_act4<HACT>(z, l_act);
      if (!last) lean_put_planes(tout, ob, lane, v, kBxScaleH);
#else
      const f32x4 v = chain_act4<HACT>((acc0 + acc1) + bv, l_act);
      if (!last) *reinterpret_cast<f32x4*>(tout + (ob * 64 + lane) * 4) = v;
#endif
      const long long row = row0 + (lane & 15);
      if (h_on && row < n_rows) store_row4(l_h, l_ldh, row, f, l_out, v, h_vec);
    };
    if (T4 > 0) {
      read_b(0);
      load_bias(0);
    }
#pragma unroll 1
    for (int g = 0; g < T4; g += 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        request((s + 3) & 3);
        read_b((s + 1) & 1);
#if RLG_LEAN_F16
        lean_group_mfma(aq[s], bq[s & 1], acc0, acc1);
#else
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][0], bq[s & 1][ch][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][1], bq[s & 1][ch][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][2], bq[s & 1][ch][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][3], bq[s & 1][ch][3], acc1, 0, 0, 0);
        }
#endif
        ++gu;
        if (gu == gpu) {
          gu = 0;
          if (unit < nun) epilogue();
          ++unit;
          load_bias(unit);
          acc0 = acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
    }
    chain_stamp(a.dbg, wave, stamp);
    __syncthreads();
    chain_stamp(a.dbg, wave, stamp);
    scale_in = kBxScaleH;
    float* t = tin;
    tin = tout;
    tout = t;
  }
}

template <int HACT>
__global__ __launch_bounds__(64 * kLeanW) void mlp_chain_fwd_lean_kernel(ChainArgs a, LeanArgs la) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  chain_fwd_lean_body<HACT>(a, la, lds);
}

// The backward of the same form: step t multiplies the dZ tile of layer L = n - 1 - t (LDS) with W_L^T - fragments over the
// layer's input features, packed transposed, so a chunk is ONE 16-byte load per lane where the pipelined kernel issues
// four strided dword loads - and applies act'(H_{L-1}); the H fragment of a unit is requested when the unit starts.
template <bool kPreloaded>
__device__ __forceinline__ void chain_bwd_lean_body(const ChainArgs& a, const LeanArgs& la, const LossArgs& loss, float* lds,
                                                    const LossQuadInputs& preloaded) {
  constexpr int W = kLeanW;
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int q4 = 4 * (lane >> 4), r16 = lane & 15;
  const long long row0 = static_cast<long long>(blockIdx.x) * 16;
  float* tile_a = lds;
  float* tile_b = lds + la.tile_b_floats;
  const int num_layers = pin_s(a.num_layers);
  const long long n_rows = pin_s(a.rows);
  const rsrc_t wr = make_rsrc(la.wf, la.wf_bytes);
  const unsigned voff = static_cast<unsigned>(lane) * 16u;
  unsigned soff = pin_s(static_cast<int>(la.stream_off[wave]));
  f32x4 aq[4][2], bq[2][2];
  auto request = [&](int slot) {
    aq[slot][0] = buf_load4_s(wr, voff, soff);
    aq[slot][1] = buf_load4_s(wr, voff, soff + 1024u);
    soff += 2048u;
  };
  request(0);
  request(1);
  request(2);
#if RLG_LEAN_F16
  auto zero_chunk = [&](float* tile, int c) { lean_zero_block(tile, c, lane); };        // (block c of the group layout)
#else
  auto zero_chunk = [&](float* tile, int c) { *reinterpret_cast<f32x4*>(tile + (c * 64 + lane) * 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; };
#endif
  // ---- the PPO loss of this row tile (training steps), as in mlp_chain_bwd_pipe_kernel
  if (a.with_loss) {
    if constexpr (kPreloaded) ppo_loss_quad_run<16, 64 * W>(loss, lds, blockIdx.x, preloaded);   // (inputs requested long ago)
    else ppo_loss_tile<16, 64 * W>(loss, lds, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- prologue: d heads tile -> LDS (fragment layout), zero chunks up to the first step's chunk count
  float scale_in = 1.0f;          // fp16 form: scale of this lane's row in the tile the current step reads
  {
    const int w = a.layer[num_layers - 1].out;
    const int KC0 = (w + 15) >> 4;
    const bool xv = vec4_ok(a.x, a.ldx);
#if RLG_LEAN_F16
    // the row's scale from its largest magnitude (every wave reads the whole row of d heads: a few loads), then the groups
    // of two 16-feature blocks, dealt to the waves, as planes
    const long long row = row0 + r16;
    float mine = 0.0f;
    if (row < n_rows) {
      for (int c = 0; c < KC0; ++c) {
        const f32x4 t = load_row4(a.x, a.ldx, row, c * 16 + q4, w, xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) mine = __builtin_fmaxf(mine, bx_finite_abs(t[e]));
      }
    }
    scale_in = bx_row_scale(mine);
    if (a.amax != nullptr) {
      // gradient maxima for the weight-gradient launch (csrc/bx_form.hpp), one entry per 16-row workgroup here: every wave
      // leaves the largest magnitude it produced per tensor in LDS, thread l combines tensor l's behind the last barrier
      const float m = bx_wave_max(mine);
      if (lane == 0) (lds + (a.bx_scales_off >> 2))[(num_layers - 1) * W + wave] = m;
    }
    for (int u = wave; u < (la.kc2[0] >> 1); u += W) {
      f32x4 lo = {0.0f, 0.0f, 0.0f, 0.0f}, hi = lo;
      if (row < n_rows) {
        if (2 * u < KC0) lo = load_row4(a.x, a.ldx, row, u * 32 + q4, w, xv);
        if (2 * u + 1 < KC0) hi = load_row4(a.x, a.ldx, row, u * 32 + 16 + q4, w, xv);
      }
      const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      u32x4 plane[kBxPlanes];
      bx_split8(x, scale_in, plane);
      *reinterpret_cast<u32x4*>(tile_a + ((2 * u) * 64 + lane) * 4) = plane[0];
      *reinterpret_cast<u32x4*>(tile_a + ((2 * u + 1) * 64 + lane) * 4) = plane[1];
    }
#else
    for (int u = wave; u < la.kc2[0]; u += W) {
      const long long row = row0 + r16;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (u < KC0 && row < n_rows) v = load_row4(a.x, a.ldx, row, u * 16 + q4, w, xv);
      *reinterpret_cast<f32x4*>(tile_a + (u * 64 + lane) * 4) = v;
    }
#endif
    __syncthreads();
  }
  float* tin = tile_a;
  float* tout = tile_b;
  const bool row_ok = row0 + r16 < n_rows;
  for (int t = 0; t + 1 < num_layers; ++t) {
    const int L = num_layers - 1 - t;                       // the layer whose weights this step uses; it produces dZ_{L-1}
    const int width = pin_s(a.layer[L].in);
    const int p_act = pin_s(a.layer[L - 1].act);
    float* p_dz = pin_s(a.layer[L - 1].dz);
    const float* p_h = pin_s(a.layer[L - 1].h);
    const long long p_ldh = pin_s(a.layer[L - 1].ldh), p_lddz = pin_s(a.layer[L - 1].lddz);
    const bool keep_tile = (L - 1 >= 1);
    double* bpart = pin_s(a.layer[L - 1].bias_partials);
    if (bpart != nullptr) bpart += static_cast<long long>(blockIdx.x) * width;
    const rsrc_t dzr = make_rsrc(p_dz + row0 * p_lddz, tile_bytes(n_rows - row0, 16, p_lddz));
    const rsrc_t hr = make_rsrc(p_h + row0 * p_ldh, tile_bytes(n_rows - row0, 16, p_ldh));
    const unsigned dz_lane = static_cast<unsigned>((r16 * static_cast<int>(p_lddz) + q4) * 4);
    const unsigned h_lane = static_cast<unsigned>((r16 * static_cast<int>(p_ldh) + q4) * 4);
    const int KC2 = pin_s(static_cast<int>(la.kc2[t]));
    const int gpu = KC2 >> 1;
    const int nun = pin_s(static_cast<int>(la.nunits[wave][t]));
    const int T4 = pin_s(static_cast<int>(la.groups4[wave][t]));
    const int full = pin_s(static_cast<int>(la.full[t]));
    if (keep_tile && wave == W - 1) {
      const int NOB = (width + 15) >> 4;
      for (int c = NOB; c < la.kc2[t + 1]; ++c) zero_chunk(tout, c);
    }
    const float* bp = tin + lane * 4;
    int cw = 0;
    auto read_b = [&](int slot) {
      bq[slot][0] = *reinterpret_cast<const f32x4*>(bp + cw * 256);
      bq[slot][1] = *reinterpret_cast<const f32x4*>(bp + (cw + 1) * 256);
      cw += 2;
      if (cw == KC2) cw = 0;
    };
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0, hv = acc0;
    int gu = 0, unit = 0;
    auto unit_block = [&](int u) -> int { return u < full ? wave * full + u : W * full + wave; };
    auto load_h = [&](int u) {
      const int f = unit_block(u) * 16 + q4;
      hv = buf_load4(hr, (u < nun && f < width) ? h_lane + static_cast<unsigned>(unit_block(u)) * 64u : kOob);
    };
#if RLG_LEAN_F16
    // the accumulators hold (weight scale x row scale) x the sums; the tile this step writes is split one step below the one
    // it reads (csrc/bx_form.hpp)
    const float inv = 1.0f / (kBxScaleW * scale_in);
    const float scale_out = scale_in * kBxScaleStepBwd;
    float dz_max = 0.0f;
#endif
    auto epilogue = [&]() {
      const int ob = unit_block(unit);
      const int f = ob * 16 + q4;
#if RLG_LEAN_F16
      f32x4 v = chain_act_grad4((acc0 + acc1) * inv, hv, p_act);
      if (!row_ok) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int e = 0; e < 4; ++e) dz_max = __builtin_fmaxf(dz_max, __builtin_fabsf(v[e]));
      if (keep_tile) lean_put_planes(tout, ob, lane, v, scale_out);
#else
      f32x4 v = chain_act_grad4(acc0 + acc1, hv, p_act);
      if (!row_ok) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (keep_tile) *reinterpret_cast<f32x4*>(tout + (ob * 64 + lane) * 4) = v;
#endif
      buf_store4(dzr, f < width ? dz_lane + static_cast<unsigned>(ob) * 64u : kOob, v);
      if (bpart != nullptr) {
        f32x4 sm;
#pragma unroll
        for (int e = 0; e < 4; ++e) sm[e] = row16_sum(v[e]);
        if (r16 == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (f + e < width) as_global(bpart)[f + e] = static_cast<double>(sm[e]);
          }
        }
      }
    };
    if (T4 > 0) {
      read_b(0);
      load_h(0);
    }
#pragma unroll 1
    for (int g = 0; g < T4; g += 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        request((s + 3) & 3);
        read_b((s + 1) & 1);
#if RLG_LEAN_F16
        lean_group_mfma(aq[s], bq[s & 1], acc0, acc1);
#else
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][0], bq[s & 1][ch][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][1], bq[s & 1][ch][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][2], bq[s & 1][ch][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][ch][3], bq[s & 1][ch][3], acc1, 0, 0, 0);
        }
#endif
        ++gu;
        if (gu == gpu) {
          gu = 0;
          if (unit < nun) epilogue();
          ++unit;
          load_h(unit);
          acc0 = acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
    }
#if RLG_LEAN_F16
    if (a.amax != nullptr) {
      const float m = bx_wave_max(dz_max);
      if (lane == 0) (lds + (a.bx_scales_off >> 2))[(L - 1) * W + wave] = m;
    }
#endif
    __syncthreads();
#if RLG_LEAN_F16
    scale_in = scale_out;
#endif
    float* tt = tin;
    tin = tout;
    tout = tt;
  }
#if RLG_LEAN_F16
  if (a.amax != nullptr && static_cast<int>(threadIdx.x) < num_layers) {
    const float* m = lds + (a.bx_scales_off >> 2) + threadIdx.x * W;
    float t = m[0];
#pragma unroll
    for (int w = 1; w < W; ++w) t = __builtin_fmaxf(t, m[w]);
    a.amax[static_cast<long long>(kBxAmaxDz + threadIdx.x) * a.amax_stride + blockIdx.x] = t;
  }
#endif
}

__global__ __launch_bounds__(64 * kLeanW) void mlp_chain_bwd_lean_kernel(ChainArgs a, LeanArgs la, LossArgs loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LossQuadInputs none = {};
  chain_bwd_lean_body<false>(a, la, loss, lds, none);
}

// forward + PPO loss + backward of a minibatch as ONE launch in the lean form (cf. mlp_chain_step_pipe_kernel): the loss
// tile's inputs are requested before the forward and arrive during it; no launch boundary between the halves
template <int HACT>
__global__ __launch_bounds__(64 * kLeanW) void mlp_chain_step_lean_kernel(ChainArgs fa, LeanArgs fla, ChainArgs ba, LeanArgs bla,
                                                                          LossArgs loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LossQuadInputs pre;
  ppo_loss_quad_load<16, 64 * kLeanW>(loss, blockIdx.x, pre);
  chain_fwd_lean_body<HACT>(fa, fla, lds);
  // the loss tile reads the heads this workgroup has just stored
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  chain_bwd_lean_body<true>(ba, bla, loss, lds, pre);
}

// fragments of the lean kernels: one thread per (1 KiB fragment, lane).  Direction 0: step t = layer t, A = W_t;
// direction 1 (backward): step t = the product with W_L^T, L = num_layers - 1 - t - blocks over the
// layer's INPUT features, chunks over its outputs.
struct LeanPackArgs {
  const float* w[kChainMaxLayers];
  const float* bias[kChainMaxLayers];
  int in[kChainMaxLayers], out[kChainMaxLayers];
  LeanArgs la;
  int num_layers, num_steps, dir;
  unsigned seg_begin[kLeanW][kChainMaxLayers];       // first fragment of the (wave, step) segment
  unsigned total_frags;
  float* dst;
};
__device__ __forceinline__ void chain_pack_frags_block(const LeanPackArgs& p);
__global__ __launch_bounds__(256) void chain_pack_frags_kernel(LeanPackArgs p) { chain_pack_frags_block(p); }
// both directions in one launch: blockIdx.y = direction
__global__ __launch_bounds__(256) void chain_pack_frags2_kernel(LeanPackArgs p0, LeanPackArgs p1) {
  if (blockIdx.y == 0) chain_pack_frags_block(p0);
  else chain_pack_frags_block(p1);
}
__device__ __forceinline__ void chain_pack_frags_block(const LeanPackArgs& p) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
#if RLG_LEAN_F16
  const unsigned frag = 2u * (t >> 6);           // one thread per (group = fragment pair, lane)
#else
  const unsigned frag = t >> 6;
#endif
  const int lane = static_cast<int>(t & 63u);
  if (frag >= p.total_frags) return;
  int w = 0, st = 0;
  for (int ww = 0; ww < kLeanW; ++ww) {
    for (int ss = 0; ss < p.num_steps; ++ss) {
      if (frag >= p.seg_begin[ww][ss]) { w = ww; st = ss; }
    }
  }
  const int q = static_cast<int>(frag - p.seg_begin[w][st]);
  const int KC2 = p.la.kc2[st];
  const int j = q / KC2, c = q - j * KC2;
  // the 4 weights of this lane in chunk cc (16 k values) of unit j: A[16 ob + (lane & 15)][16 cc + 4 (lane >> 4) .. + 3]
  auto chunk_values = [&](int cc) -> f32x4 {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (j < p.la.nunits[w][st]) {
      const int full = p.la.full[st];
      const int ob = j < full ? w * full + j : kLeanW * full + w;
      const int i = ob * 16 + (lane & 15);
      const int k0 = cc * 16 + 4 * (lane >> 4);
      if (p.dir == 0) {
        const int L = st;
        if (i < p.out[L]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            if (k < p.in[L]) v[e] = p.w[L][static_cast<long long>(i) * p.in[L] + k];
          }
        }
      } else {
        const int L = p.num_layers - 1 - st;
        if (i < p.in[L]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            if (k < p.out[L]) v[e] = p.w[L][static_cast<long long>(k) * p.in[L] + i];
          }
        }
      }
    }
    return v;
  };
#if RLG_LEAN_F16
  // a group = the two chunks (c, c + 1), c even: their 8 values per lane as two fp16 planes - plane 0 where chunk c was,
  // plane 1 where chunk c + 1 was (the slack fragments behind the last stream stay zero: q runs past every unit there)
  const f32x4 lo = chunk_values(c), hi = chunk_values(c + 1);       // (c is even: segments and units hold whole groups)
  const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  u32x4 plane[kBxPlanes];
  bx_split8(x, kBxScaleW, plane);
  *reinterpret_cast<u32x4*>(p.dst + (static_cast<long long>(frag) * 64 + lane) * 4) = plane[0];
  *reinterpret_cast<u32x4*>(p.dst + (static_cast<long long>(frag + 1) * 64 + lane) * 4) = plane[1];
#else
  *reinterpret_cast<f32x4*>(p.dst + (static_cast<long long>(frag) * 64 + lane) * 4) = chunk_values(c);
#endif
}

// host: the stream layout of one direction.  Returns the buffer size in bytes (incl. the slack the 3-groups-ahead requests
// of the last step run into), -1 if a shape does not fit the format.
static long long chain_lean_plan(int num_layers, const int* in_features, const int* out_features, int dir, LeanPackArgs* pk) {
  LeanArgs& la = pk->la;
  if (num_layers < 1 || num_layers > kChainMaxLayers || (dir != 0 && num_layers < 2)) return -1;
  const int steps = dir == 0 ? num_layers : num_layers - 1;
  int tile_even = 0, tile_odd = 0;
  int nobs[kChainMaxLayers];
  for (int t = 0; t < steps; ++t) {
    const int L = dir == 0 ? t : num_layers - 1 - t;
    const int K = dir == 0 ? in_features[L] : out_features[L];
    const int I = dir == 0 ? out_features[L] : in_features[L];
    const int kc2 = (((K + 15) >> 4) + 1) & ~1;
    nobs[t] = (I + 15) >> 4;
    if (kc2 > 255 || nobs[t] / kLeanW + 1 > 255) return -1;
    la.kc2[t] = static_cast<unsigned char>(kc2);
    la.full[t] = static_cast<unsigned char>(nobs[t] / kLeanW);
    int& tile = (t & 1) ? tile_odd : tile_even;                            // the input tile of step t: kc2 chunks
    tile = kc2 * 256 > tile ? kc2 * 256 : tile;
  }
  if (dir == 0) {
    // the prologue's statistics scratch lives in the odd tile (2 x padded width floats)
    const int in0p = (in_features[0] + 3) & ~3;
    if (tile_odd < 4 * in0p + 64) tile_odd = 4 * in0p + 64;
  }
  la.tile_b_floats = tile_even;
  pk->dir = dir;
  pk->num_steps = steps;
  unsigned frag = 0;
  for (int w = 0; w < kLeanW; ++w) {
    la.stream_off[w] = frag * 1024u;
    for (int t = 0; t < steps; ++t) {
      const int full = nobs[t] / kLeanW, rem = nobs[t] - full * kLeanW;
      const int nun = full + (w < rem ? 1 : 0);
      const int groups = nun * (la.kc2[t] >> 1);
      const int g4 = (groups + 3) & ~3;
      if (g4 > 65535) return -1;
      la.nunits[w][t] = static_cast<unsigned char>(nun);
      la.groups4[w][t] = static_cast<unsigned short>(g4);
      pk->seg_begin[w][t] = frag;
      frag += static_cast<unsigned>(g4) * 2u;
    }
  }
  pk->total_frags = frag + 8;                     // 3 groups of slack behind the last wave's stream (zero fragments)
  pk->num_layers = num_layers;
  const long long bytes = static_cast<long long>(pk->total_frags) * 1024;
  if (bytes >= static_cast<long long>(kOob)) return -1;
  la.wf_bytes = static_cast<unsigned>(bytes);
  for (int L = 0; L < num_layers; ++L) {
    pk->in[L] = in_features[L];
    pk->out[L] = out_features[L];
  }
  // LDS floats of both tiles (the caller adds what else lives there)
  return bytes + (static_cast<long long>(tile_even + tile_odd) << 40);
}
static long long chain_lean_bytes(long long plan) { return plan < 0 ? plan : (plan & ((1LL << 40) - 1)); }
static int chain_lean_tile_floats(long long plan) { return static_cast<int>(plan >> 40); }


}  // namespace rlg

// ---------------------------------------------------------------------------------
// C ABI (declared in include/rlg_hip.h)
// ---------------------------------------------------------------------------------
extern "C" {

// ---- experimental: the lean 16-row forward (see mlp_chain_fwd_lean_kernel) -----------------------------------------
long long rlg_mlp_chain_frags_bytes(int num_layers, const int* in_features, const int* out_features, int direction) {
  rlg::LeanPackArgs pk = {};
  return rlg::chain_lean_bytes(rlg::chain_lean_plan(num_layers, in_features, out_features, direction, &pk));
}

int rlg_mlp_chain_pack_frags(int num_layers, const float* const* weights, const float* const* biases_or_null,
                             const int* in_features, const int* out_features, int direction, void* frags, void* stream) {
  using namespace rlg;
  LeanPackArgs pk = {};
  if (chain_lean_plan(num_layers, in_features, out_features, direction, &pk) < 0 || frags == nullptr)
    return static_cast<int>(hipErrorInvalidValue);
  for (int L = 0; L < num_layers; ++L) {
    pk.w[L] = weights[L];
    pk.bias[L] = biases_or_null ? biases_or_null[L] : nullptr;
  }
  pk.dst = static_cast<float*>(frags);
  const unsigned threads = pk.total_frags * 64u / (RLG_LEAN_F16 ? 2u : 1u);
  hipLaunchKernelGGL(chain_pack_frags_kernel, dim3((threads + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), pk);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_mlp_chain_forward_lean(int num_layers, const float* const* biases, const int* in_features, const int* out_features,
                               const int* acts, float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                               const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                               const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                               double* rms_var_out, long long* rms_count_out, long long rows, const void* frags,
                               void* stream) {
  using namespace rlg;
  if (rows <= 0) return 0;
  if (frags == nullptr) return static_cast<int>(hipErrorInvalidValue);
  ChainArgs args;
  const float* none[kChainMaxLayers] = {};
  {
    // chain_fill wants weight pointers for its alignment check only: the fragments stand in
    for (int L = 0; L < num_layers && L < kChainMaxLayers; ++L) none[L] = static_cast<const float*>(frags);
  }
  if (chain_fill(args, num_layers, none, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  LeanPackArgs pk = {};
  const long long plan = chain_lean_plan(num_layers, in_features, out_features, 0, &pk);
  if (plan < 0) return static_cast<int>(hipErrorNotSupported);
  for (int L = 0; L < num_layers; ++L) {
    args.layer[L].bias = biases ? biases[L] : nullptr;
    args.layer[L].h = act_out[L];
    args.layer[L].ldh = act_ld[L];
    if (reinterpret_cast<uintptr_t>(args.layer[L].bias) % 4 != 0) return static_cast<int>(hipErrorInvalidValue);
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
      return static_cast<int>(hipErrorInvalidValue);
  }
  args.rms_count = rms_count;
  args.rms_mean_out = rms_mean_out;
  args.rms_var_out = rms_var_out;
  args.rms_count_out = rms_count_out;
  args.xn = xn_out;
  args.rows = rows;
  args.dbg = chain_debug_stamps();
  args.lds_b_floats = pk.la.tile_b_floats;
  args.lds_split_floats = 0;
  args.no_ksplit = 1;
  LeanArgs la = pk.la;
  la.wf = static_cast<const float*>(frags);
  const int lds_bytes = chain_lean_tile_floats(plan) * 4;
  if (lds_bytes > 64 * 1024) return static_cast<int>(hipErrorNotSupported);
  bool elu_only = true;
  for (int L = 0; L < num_layers; ++L) elu_only = elu_only && (acts[L] == kChElu || acts[L] == kChIdentity);
  const int grid = static_cast<int>((rows + 15) / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (rlg_mlp_chain_time_next: the events ride on this dispatch)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  chain_take_events(&ev0, &ev1);
  const auto kern = elu_only ? mlp_chain_fwd_lean_kernel<kChElu> : mlp_chain_fwd_lean_kernel<kChAny>;
  if (ev0 != nullptr)       // (the plain launch otherwise: the one that stream capture takes)
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, args, la);
  else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, args, la);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_mlp_chain_backward_lean(int num_layers, const int* in_features, const int* out_features, const int* acts,
                                const float* const* act_in, const long long* act_ld, const float* d_out, long long ld_dout,
                                float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                                const rlg_ppo_loss_desc* ppo_loss, long long rows, const void* frags, void* stream) {
  using namespace rlg;
  if (rows <= 0) return 0;
  if (num_layers < 2 || frags == nullptr) return static_cast<int>(hipErrorInvalidValue);
  ChainArgs args;
  const float* none[kChainMaxLayers] = {};
  for (int L = 0; L < num_layers && L < kChainMaxLayers; ++L) none[L] = static_cast<const float*>(frags);
  if (chain_fill(args, num_layers, none, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  LeanPackArgs pk = {};
  const long long plan = chain_lean_plan(num_layers, in_features, out_features, 1, &pk);
  if (plan < 0) return static_cast<int>(hipErrorNotSupported);
  for (int L = 0; L + 1 < num_layers; ++L) {
    ChainLayer& ly = args.layer[L];
    ly.h = const_cast<float*>(act_in[L]);
    ly.ldh = act_ld[L];
    ly.dz = dz_out[L];
    ly.lddz = dz_ld[L];
    ly.bias_partials = bias_partials ? bias_partials[L] : nullptr;
    if (act_in[L] == nullptr || dz_out[L] == nullptr) return static_cast<int>(hipErrorInvalidValue);
    if (!(vec4_ok_host(ly.h, ly.ldh) && vec4_ok_host(ly.dz, ly.lddz) && (ly.out & 3) == 0 && ly.ldh < (1 << 20) && ly.lddz < (1 << 20)))
      return static_cast<int>(hipErrorNotSupported);
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
  args.lds_b_floats = pk.la.tile_b_floats;
  int lds_bytes = chain_lean_tile_floats(plan) * 4;
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
    const int need = static_cast<int>(ppo_loss_lds_bytes(16, d.actions_num, 512));
    if (need > lds_bytes) lds_bytes = need;
  }
  // gradient maxima for the weight-gradient launch (rlg_mlp_chain_gradient_maxima; one entry per 16-row workgroup here): the
  // waves' maxima sit behind everything else in LDS
  args.amax = nullptr;
  args.amax_stride = 0;
  {
    float* entries = nullptr;
    int stride = 0;
    chain_take_gradient_maxima(&entries, &stride);
    if (RLG_LEAN_F16 && entries != nullptr) {
      if (stride < (rows + 15) / 16) return static_cast<int>(hipErrorInvalidValue);
      args.amax = entries;
      args.amax_stride = stride;
      args.bx_scales_off = (lds_bytes + 15) & ~15;
      lds_bytes = args.bx_scales_off + kChainMaxLayers * kLeanW * 4;
    }
  }
  if (lds_bytes > 64 * 1024) return static_cast<int>(hipErrorNotSupported);
  LeanArgs la = pk.la;
  la.wf = static_cast<const float*>(frags);
  const int grid = static_cast<int>((rows + 15) / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  chain_take_events(&ev0, &ev1);
  const auto kern = mlp_chain_bwd_lean_kernel;
  if (ev0 != nullptr)
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, args, la, loss);
  else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, args, la, loss);
  RLG_RETURN_LAUNCH_STATUS();
}

// forward and backward fragments of the lean kernels in ONE launch (behind every optimiser step of an agent whose
// launches run them; frags_bwd_or_null: forward only)
int rlg_mlp_chain_pack_frags_both(int num_layers, const float* const* weights, const float* const* biases,
                                  const int* in_features, const int* out_features, void* frags_fwd, void* frags_bwd_or_null,
                                  void* stream) {
  using namespace rlg;
  if (frags_bwd_or_null == nullptr || num_layers < 2)
    return rlg_mlp_chain_pack_frags(num_layers, weights, biases, in_features, out_features, 0, frags_fwd, stream);
  LeanPackArgs p0 = {}, p1 = {};
  if (chain_lean_plan(num_layers, in_features, out_features, 0, &p0) < 0 || chain_lean_plan(num_layers, in_features, out_features, 1, &p1) < 0 ||
      frags_fwd == nullptr)
    return static_cast<int>(hipErrorInvalidValue);
  for (int L = 0; L < num_layers; ++L) {
    p0.w[L] = p1.w[L] = weights[L];
    p0.bias[L] = p1.bias[L] = nullptr;
    (void)biases;
  }
  p0.dst = static_cast<float*>(frags_fwd);
  p1.dst = static_cast<float*>(frags_bwd_or_null);
  const unsigned threads = (p0.total_frags > p1.total_frags ? p0.total_frags : p1.total_frags) * 64u / (RLG_LEAN_F16 ? 2u : 1u);
  hipLaunchKernelGGL(chain_pack_frags2_kernel, dim3((threads + 255u) / 256u, 2), dim3(256), 0, static_cast<hipStream_t>(stream), p0, p1);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_mlp_chain_step_lean(int num_layers, const float* const* biases, const int* in_features, const int* out_features,
                            const int* acts, float* const* act_out, const long long* act_ld, const float* x, long long ldx,
                            const double* rms_mean, const double* rms_var, float rms_eps, float* xn_out,
                            const double* rms_batch, const long long* rms_count, double* rms_mean_out,
                            double* rms_var_out, long long* rms_count_out, float* d_out, long long ld_dout,
                            float* const* dz_out, const long long* dz_ld, double* const* bias_partials,
                            const rlg_ppo_loss_desc* ppo_loss, long long rows, const void* frags_fwd, const void* frags_bwd,
                            void* stream) {
  using namespace rlg;
  if (rows <= 0) return 0;
  if (num_layers < 2 || ppo_loss == nullptr || frags_fwd == nullptr || frags_bwd == nullptr || chain_debug_stamps() != nullptr)
    return static_cast<int>(hipErrorNotSupported);
  // one 8-wave workgroup per CU holds the loss inputs through the forward (> 128 registers): beyond one round of workgroups
  // the two separate launches, which run two workgroups per CU, are faster
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return static_cast<int>(hipErrorNotSupported);
    if ((rows + 15) / 16 > cus) return static_cast<int>(hipErrorNotSupported);
  }
  const float* none[kChainMaxLayers] = {};
  for (int L = 0; L < num_layers && L < kChainMaxLayers; ++L) none[L] = static_cast<const float*>(frags_fwd);
  ChainArgs fa;
  if (chain_fill(fa, num_layers, none, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  LeanPackArgs fpk = {}, bpk = {};
  const long long fplan = chain_lean_plan(num_layers, in_features, out_features, 0, &fpk);
  const long long bplan = chain_lean_plan(num_layers, in_features, out_features, 1, &bpk);
  if (fplan < 0 || bplan < 0) return static_cast<int>(hipErrorNotSupported);
  for (int L = 0; L < num_layers; ++L) {
    fa.layer[L].bias = biases ? biases[L] : nullptr;
    fa.layer[L].h = act_out[L];
    fa.layer[L].ldh = act_ld[L];
    if (act_out[L] == nullptr) return static_cast<int>(hipErrorNotSupported);       // training form only
    if (reinterpret_cast<uintptr_t>(fa.layer[L].bias) % 4 != 0) return static_cast<int>(hipErrorInvalidValue);
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
  fa.lds_b_floats = fpk.la.tile_b_floats;
  fa.lds_split_floats = 0;
  fa.no_ksplit = 1;
  ChainArgs ba;
  if (chain_fill(ba, num_layers, none, in_features, out_features, acts)) return static_cast<int>(hipErrorInvalidValue);
  for (int L = 0; L + 1 < num_layers; ++L) {
    ChainLayer& ly = ba.layer[L];
    ly.h = act_out[L];
    ly.ldh = act_ld[L];
    ly.dz = dz_out[L];
    ly.lddz = dz_ld[L];
    ly.bias_partials = bias_partials ? bias_partials[L] : nullptr;
    if (dz_out[L] == nullptr) return static_cast<int>(hipErrorInvalidValue);
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
  ba.lds_b_floats = bpk.la.tile_b_floats;
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
  int lds_bytes = chain_lean_tile_floats(fplan) * 4;
  if (chain_lean_tile_floats(bplan) * 4 > lds_bytes) lds_bytes = chain_lean_tile_floats(bplan) * 4;
  const int need = static_cast<int>(ppo_loss_lds_bytes(16, d.actions_num, 512));
  if (need > lds_bytes) lds_bytes = need;
  // gradient maxima for the weight-gradient launch (rlg_mlp_chain_gradient_maxima; one entry per 16-row workgroup here): the
  // waves' maxima sit behind everything else in LDS
  ba.amax = nullptr;
  ba.amax_stride = 0;
  {
    float* entries = nullptr;
    int stride = 0;
    chain_take_gradient_maxima(&entries, &stride);
    if (RLG_LEAN_F16 && entries != nullptr) {
      if (stride < (rows + 15) / 16) return static_cast<int>(hipErrorInvalidValue);
      ba.amax = entries;
      ba.amax_stride = stride;
      ba.bx_scales_off = (lds_bytes + 15) & ~15;
      lds_bytes = ba.bx_scales_off + kChainMaxLayers * kLeanW * 4;
    }
  }
  if (lds_bytes > 64 * 1024) return static_cast<int>(hipErrorNotSupported);
  LeanArgs fla = fpk.la, bla = bpk.la;
  fla.wf = static_cast<const float*>(frags_fwd);
  bla.wf = static_cast<const float*>(frags_bwd);
  bool elu_only = true;
  for (int L = 0; L < num_layers; ++L) elu_only = elu_only && (acts[L] == kChElu || acts[L] == kChIdentity);
  const int grid = static_cast<int>((rows + 15) / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  chain_take_events(&ev0, &ev1);
  const auto kern = elu_only ? mlp_chain_step_lean_kernel<kChElu> : mlp_chain_step_lean_kernel<kChAny>;
  if (ev0 != nullptr)
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, ev0, ev1, 0, fa, fla, ba, bla, loss);
  else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kLeanW), static_cast<size_t>(lds_bytes), st, fa, fla, ba, bla, loss);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
