// Two-way split of fp32 values into fp16 planes for the split-product kernels (round 6): with a power-of-two scale S
// that puts the operand's largest magnitude below 2^15,  x S = h0 + h1 + e,  h0 = RN16(x S), h1 = RN16(x S - h0),
// |e| <= max(2^-22 |x S|, 2^-25) in the worst case, ~2^-24 |x S| rms  (fp16: 11 significant bits per plane, gradual
// underflow below 2^-14 - the matrix core keeps subnormal inputs, tools/exp/f16_mfma_probe.hip).  A product is the THREE
// plane products h0 g0 + h0 g1 + h1 g0 on v_mfma_f32_16x16x32_f16 (each exact in fp32, the rate of the bf16 MFMA),
// un-scaled by 1 / (S_x S_y) behind the accumulation.  The dropped h1 g1 and the representation errors are <= 3 * 2^-22
// |x||y| per product in the worst case (the six-product bf16 form of split_bf16.hpp: 3 * 2^-24) and ~2^-24 rms - below the
// roundings of the fp32 accumulation either way: against fp64 the sums of both forms are as accurate as with exact fp32
// products (tools/exp/split_f16_numerics.py, tests/test_split_products_cpu.py, tools/exp/dw_bf16_check.py on the device) -
// at half the MFMAs and two thirds of the planes of the bf16 form.
// What the bf16 form does not need: the scale.  An operand element beyond 65504 / S becomes Inf (non-finite out, never a
// finite wrong value); the launches take their scales from the measured maxima of the tensors they multiply.
#pragma once

#include "rlg_device.hpp"
#include "split_bf16.hpp"

namespace rlg {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 split_f16x2 __attribute__((ext_vector_type(2)));

// RNE conversion of a pair: ONE v_cvt_pk_f16_f32 (a compiler builtin, not asm: the hazard recogniser must see the producer
// of an MFMA operand, see split_bf16.hpp)
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((split_f32x2{lo, hi}), split_f16x2));
}

#ifndef RLG_F16_FMA_MIX
#define RLG_F16_FMA_MIX 1          // 0: residual by v_cvt_f32_f16 + subtraction (the same bits, two more conversion-rate instructions per pair)
#endif

// x * scale - (float) half `hi` of the packed fp16 pair h, ONE rounding: v_fma_mix_f32 reads the fp16 operand in place, so
// the plane is not converted back (conversions issue at a quarter of the plain VALU rate: the back-conversions were a
// third of the split's cycles).  x * scale is exact (a power of two) and so is the difference: the same bits as
// (x * scale) - (float) h.  An asm statement is fine HERE: its result feeds a VALU conversion, not an MFMA (the hazard
// recogniser does not see inside asm - what an MFMA reads must come from a builtin, see split_bf16.hpp).
template <bool kHi>
__device__ __forceinline__ float f16_residual(float x, float scale, unsigned h) {
  float r;
  if constexpr (kHi)
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(scale), "v"(h));
  else
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(scale), "v"(h));
  return r;
}

// One pair of fp32 values, times `scale` -> its dword of each of the two planes: v_pk_mul_f32, v_cvt_pk_f16_f32, two
// v_fma_mix_f32, v_cvt_pk_f16_f32 = 5 VALU instructions, two of them at the conversion rate (the bf16 form: 9, three).
__device__ __forceinline__ void split_pair_f16(float x0, float x1, float scale, unsigned& p0, unsigned& p1) {
  const split_f16x2 h = __builtin_convertvector((split_f32x2{x0 * scale, x1 * scale}), split_f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
#if RLG_F16_FMA_MIX
  const float r0 = f16_residual<false>(x0, scale, p0);
  const float r1 = f16_residual<true>(x1, scale, p0);
#else
  const float r0 = x0 * scale - static_cast<float>(h[0]);
  const float r1 = x1 * scale - static_cast<float>(h[1]);
#endif
  p1 = cvt_pk_f16(r0, r1);
}

// 8 floats (the lane's 8 k values of one 16-wide block), times `scale` -> 2 planes of 8 packed fp16; x[2q], x[2q + 1]
// share a dword.
__device__ __forceinline__ void f16_split8(const float (&x)[8], float scale, u32x4 (&plane)[2]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned p0, p1;
    split_pair_f16(x[2 * q], x[2 * q + 1], scale, p0, p1);
    plane[0][q] = p0;
    plane[1][q] = p1;
  }
}

// power-of-two scale that puts `amax` (the largest magnitude of an operand, >= 0, finite) into [2^13, 2^14): two bits below
// the fp16 overflow threshold.  The exponent is held within +- 100 (operands of 2^-87 .. 2^113 are scaled exactly into that
// range; zero / denormal maxima take the largest scale, and 1 / (S_x S_y) stays a normal fp32 for every pair of scales).
__device__ __forceinline__ float f16_scale_for(float amax) {
  const int e = (__float_as_int(amax) >> 23) & 0xff;            // biased exponent: amax in [2^(e-127), 2^(e-126))
  int s = 127 + 13 - (e - 127);                                 // biased exponent of 2^(13 - (e - 127))
  s = min(max(s, 127 - 100), 127 + 100);
  return __int_as_float(s << 23);
}

}  // namespace rlg
