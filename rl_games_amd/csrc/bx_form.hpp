// The product form of the chain's split-product kernels: plane counts, scales and the split helpers shared by the kernels
// (mlp_chain_bx*.hip), the pack launches (mlp_chain_common.hpp, adam_pack.hip) and the host-side layout code.
#pragma once

#include "rlg_device.hpp"
#include "split_bf16.hpp"
#include "split_f16.hpp"

namespace rlg {

// Product form of the chain's split-product kernels (round 6).  1: two fp16 planes per operand, three plane products per
// fp32 product on v_mfma_f32_16x16x32_f16, operands scaled by powers of two (split_f16.hpp).  0: three bf16 planes, six
// products on v_mfma_f32_16x16x32_bf16 (rounds 3 - 5; no scales, the full fp32 range).
#ifndef RLG_BX_F16
#define RLG_BX_F16 1
#endif

constexpr int kBxFrag = 1024;            // bytes of one plane fragment: 64 lanes x 8 half-width values
constexpr int kBxPlanes = RLG_BX_F16 ? 2 : 3;
constexpr int kBxProducts = RLG_BX_F16 ? 3 : 6;
constexpr int kBxChunk = kBxPlanes * kBxFrag;    // the planes of one (block, chunk) / (chunk, row group)
// plane pairs of the products, small terms first
#if RLG_BX_F16
constexpr int kBxPa[kBxProducts] = {1, 0, 0};
constexpr int kBxPb[kBxProducts] = {0, 1, 0};
#else
constexpr int kBxPa[kBxProducts] = {2, 0, 1, 1, 0, 0};
constexpr int kBxPb[kBxProducts] = {0, 2, 1, 0, 1, 0};
#endif
// Scales of the fp16 form (1 in the bf16 form): weights, hidden activations and normalised observations (clamped to
// [-5, 5], models.py:54-56) take fixed powers of two; raw observations and the gradient tiles of the backward are scaled ROW
// BY ROW from the row's largest magnitude (bx_row_scale: a batch row is column j of the MFMA, its scale factors out of
// every sum of that column - rows never influence one another).  An element beyond 65504 / scale becomes Inf in its plane
// and NaN / Inf in everything computed from it.
constexpr float kBxScaleW = RLG_BX_F16 ? 64.0f : 1.0f;         // |w| < 1023
constexpr float kBxScaleH = RLG_BX_F16 ? 16.0f : 1.0f;         // |h| < 4094
constexpr float kBxScaleObsNorm = RLG_BX_F16 ? 4096.0f : 1.0f; // |x| <= 5
constexpr float kBxScaleStepBwd = RLG_BX_F16 ? 0.125f : 1.0f;  // dZ of a layer may exceed the dZ above it 32-fold

__device__ __forceinline__ f32x4 bx_mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
#if RLG_BX_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// 8 values of a lane (times `scale` in the fp16 form) -> its 16 bytes of every plane
__device__ __forceinline__ void bx_split8(const float (&x)[8], float scale, u32x4 (&plane)[kBxPlanes]) {
#if RLG_BX_F16
  f16_split8(x, scale, plane);
#else
  dw_split8(x, plane);
#endif
}

// 4 values -> 8 bytes of every plane
__device__ __forceinline__ void bx_split4(const f32x4& v, float scale, unsigned (&plane)[kBxPlanes][2]) {
#if RLG_BX_F16
#pragma unroll
  for (int q = 0; q < 2; ++q) split_pair_f16(v[2 * q], v[2 * q + 1], scale, plane[0][q], plane[1][q]);
#else
  split4_planes(v, plane);
#endif
}

// Scale of one row from its largest magnitude.  `mine`: the largest |value| among this lane's elements of row (lane & 15)
// of its row group; the row's elements are spread over the four lanes that share lane & 15.  Non-finite elements do not
// take part (they become Inf / NaN in the planes whatever the scale).
__device__ __forceinline__ float bx_row_scale(float mine) {
#if RLG_BX_F16
  mine = __builtin_fmaxf(mine, __shfl_xor(mine, 16));
  mine = __builtin_fmaxf(mine, __shfl_xor(mine, 32));
  return f16_scale_for(mine);
#else
  return 1.0f;
#endif
}
// Gradient maxima for the weight-gradient launch.  Its sums run over the batch rows, so an operand's scale must hold for
// every row of a K-slice: the split-fp16 BACKWARD leaves, per 64-row workgroup and dZ tensor, the largest magnitude it
// produced - entries[(kBxAmaxDz + l) * stride + workgroup] for dZ of layer l (the last layer: the d heads tile it read) -
// with plain stores (every launch overwrites its own entries: nothing to reset), and a wave of the weight-gradient launch
// takes the largest entry over ITS rows.  (Measured on the way: one device-scope atomicMax per wave and tensor instead of
// the per-workgroup stores cost the chain launches 40 us; a maximum that maps NaN to Inf - three VALU instructions per
// element instead of one - 11 us; no tracking at all, the backward's own bound of 8 x the dZ above per layer instead, is
// free but costs the weight gradients of the deep layers 3 - 9 bits: the benchmarked epoch's losses then leave the oracle's
// band.)  The other operand needs none: hidden activations and normalised observations are split under the forward's
// fixed scales.
constexpr int kBxAmaxDz = 0;
// LDS bytes of the backward behind its tiles: 64 row scales of the d heads tile + 8 layers x (up to) 8 waves of gradient maxima
// (+ alignment)
constexpr int kBxBwdScratch = RLG_BX_F16 ? 16 + 64 * 4 + 8 * 8 * 4 : 0;

// largest value of a wave (uniform result): row rotations inside the 16-lane rows, then the four rows through SGPRs
__device__ __forceinline__ float bx_wave_max(float t) {
#define RLG_DPP_MAX(ctrl) t = __builtin_fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), ctrl, 0xf, 0xf, false)))
  RLG_DPP_MAX(0x128);
  RLG_DPP_MAX(0x124);
  RLG_DPP_MAX(0x122);
  RLG_DPP_MAX(0x121);
#undef RLG_DPP_MAX
  const int b = __builtin_bit_cast(int, t);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}
// |x| for the row maximum: finite values only
__device__ __forceinline__ float bx_finite_abs(float x) {
  const float a = __builtin_fabsf(x);
  return a <= 3.4028234664e38f ? a : 0.0f;
}

}  // namespace rlg
