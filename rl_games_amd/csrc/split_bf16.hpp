// Exact three-way split of fp32 values into bf16 planes for the split-product kernels (mlp_dw.hip, mlp_chain_bx.hip):
// x = x0 + x1 + x2, each plane a bf16 (3 x 8 significant bits cover the 24 of an fp32).
#pragma once

#include "rlg_device.hpp"

namespace rlg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef RLG_SPLIT_PK
#define RLG_SPLIT_PK 1           // 0: scalar residuals (v_sub_f32), the round-2 form - same bits, 11 instead of 9 VALU per pair
#endif

typedef float split_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));

// RNE conversion of a pair: ONE v_cvt_pk_bf16_f32.  NOT an asm statement (rounds 2 - 3 had one here): an instruction
// inside inline asm is invisible to hipcc's hazard recogniser, and a VALU result needs 2 wait states before an MFMA may
// read it as SrcA / SrcB - the weight-gradient kernel's (2, 2) tile ran some of its MFMAs one wait state behind the
// conversion that produced their operand and picked up the previous batch's plane now and then (last-bit noise that
// changed from run to run; tools/exp/determinism_probe.py).  hipcc places `s_nop 1` itself for the builtin form.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((split_f32x2{lo, hi}), split_bf16x2));
}

// One pair of fp32 values (an even-aligned register pair) -> its dword of each of the three planes.  Per pair: 3
// v_cvt_pk_bf16_f32 (RNE) and, for each of the two residuals, v_lshlrev + v_and + ONE v_pk_add_f32 with negated second
// operand (exact: the residual has <= 16 (8) significant bits) = 9 VALU instructions; the splitting is the largest
// VALU item of every split-product kernel.  (Round 6, profiles/r6_coexec_bf16.txt: beside v_mfma_f32_16x16x32_bf16 plain
// VALU instructions overlap, v_pk_add_f32 does NOT - one per MFMA costs + 16.5 cycles.  The loops that call this split do not
// interleave it with MFMAs - they split, then multiply - and with RLG_SPLIT_PK=0 -fno-slp-vectorize (scalar residuals) they
// measured 3 us faster in the weight-gradient launch and 4 us slower in the forward: the packed form stays the default; a loop
// that DOES deal the split out between MFMAs must use the scalar form.)
__device__ __forceinline__ void split_pair(split_f32x2 r, unsigned& p0, unsigned& p1, unsigned& p2) {
  unsigned w = cvt_pk_bf16(r[0], r[1]);
  p0 = w;
#if RLG_SPLIT_PK
  r = r - split_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
#else
  r[0] -= __uint_as_float(w << 16);
  r[1] -= __uint_as_float(w & 0xffff0000u);
#endif
  w = cvt_pk_bf16(r[0], r[1]);
  p1 = w;
#if RLG_SPLIT_PK
  r = r - split_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
#else
  r[0] -= __uint_as_float(w << 16);
  r[1] -= __uint_as_float(w & 0xffff0000u);
#endif
  w = cvt_pk_bf16(r[0], r[1]);
  p2 = w;
}

// 8 floats (the lane's 8 k values of one 16-wide block) -> 3 planes of 8 packed bf16; x[2q], x[2q + 1] share a dword.
__device__ __forceinline__ void dw_split8(const float (&x)[8], u32x4 (&plane)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned p0, p1, p2;
    split_pair(split_f32x2{x[2 * q], x[2 * q + 1]}, p0, p1, p2);
    plane[0][q] = p0;
    plane[1][q] = p1;
    plane[2][q] = p2;
  }
}

// Two blocks at once whose values sit side by side in the registers - xa[u], xb[u] = elements (a, a + 1) of the row
// vector a lane loaded for k value u (the weight-gradient kernel): the packed residual pairs the two COLUMNS of a row
// (adjacent registers as loaded - pairing the rows would cost two moves per pair), the conversion pairs the rows of a
// column as the MFMA operand wants them.  Same values as two dw_split8 calls.
__device__ __forceinline__ void dw_split8x2(const split_f32x2 (&x)[8], u32x4 (&plane_a)[3], u32x4 (&plane_b)[3]) {
#if !RLG_SPLIT_PK
  float xa[8], xb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    xa[u] = x[u][0];
    xb[u] = x[u][1];
  }
  dw_split8(xa, plane_a);
  dw_split8(xb, plane_b);
  return;
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    split_f32x2 r0 = x[2 * q], r1 = x[2 * q + 1];           // rows 2q, 2q + 1; [0] = column a, [1] = column a + 1
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned wa = cvt_pk_bf16(r0[0], r1[0]), wb = cvt_pk_bf16(r0[1], r1[1]);
      plane_a[p][q] = wa;
      plane_b[p][q] = wb;
      if (p < 2) {
        r0 = r0 - split_f32x2{__uint_as_float(wa << 16), __uint_as_float(wb << 16)};
        r1 = r1 - split_f32x2{__uint_as_float(wa & 0xffff0000u), __uint_as_float(wb & 0xffff0000u)};
      }
    }
  }
}

// 4 floats -> 3 planes of 2 dwords (4 packed bf16), same pairing as dw_split8
__device__ __forceinline__ void split4_planes(const f32x4& x, unsigned (&plane)[3][2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
    split_pair(split_f32x2{x[2 * q], x[2 * q + 1]}, plane[0][q], plane[1][q], plane[2][q]);
}

}  // namespace rlg
