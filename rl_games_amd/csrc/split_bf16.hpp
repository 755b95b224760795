// Exact three-way split of fp32 values into bf16 planes for the split-product kernels (mlp_dw.hip, mlp_chain_bx.hip):
// x = x0 + x1 + x2, each plane a bf16 (3 x 8 significant bits cover the 24 of an fp32).
#pragma once

#include "rlg_device.hpp"

namespace rlg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 floats (the lane's 8 k values of one 16-wide block) -> 3 planes of 8 packed bf16.  The conversion is
// an asm statement so that hipcc keeps ONE v_cvt_pk_bf16_f32 (RNE) per pair (it otherwise converts the low
// half a second time for the shift) and leaves the residuals as plain v_sub_f32 (no v_pk_add_f32 + moves).
__device__ __forceinline__ void dw_split8(const float (&x)[8], u32x4 (&plane)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lo = x[2 * q], hi = x[2 * q + 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      unsigned w;
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
      plane[p][q] = w;
      if (p < 2) {
        lo -= __uint_as_float(w << 16);                  // exact: the residual has <= 16 (8) significant bits
        hi -= __uint_as_float(w & 0xffff0000u);
      }
    }
  }
}

// 4 floats -> 3 planes of 2 dwords (4 packed bf16), same pairing as dw_split8
__device__ __forceinline__ void split4_planes(const f32x4& x, unsigned (&plane)[3][2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float lo = x[2 * q], hi = x[2 * q + 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      unsigned w;
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
      plane[p][q] = w;
      if (p < 2) {
        lo -= __uint_as_float(w << 16);
        hi -= __uint_as_float(w & 0xffff0000u);
      }
    }
  }
}

}  // namespace rlg
