// The unit engine of the split-bf16 chain kernels (mlp_chain_bx.hip: backward, mlp_chain_bx_fwd.hip: forward); see
// mlp_chain_bx.hip for the scheme.
#pragma once

#include "mlp_chain_common.hpp"

namespace rlg {

// register class the accumulators are pinned to: AGPRs ("+a") in the 512-register kernels; a build that must fit two waves
// per SIMD defines it as "+v" - without an AGPR constraint anywhere the compiler gives the kernel ONE file of 256
// registers and issues the MFMAs on VGPR accumulators (with one it splits the file 128 + 128)
#ifndef RLG_ACC_CLASS
#define RLG_ACC_CLASS "+a"
#endif
#define RLG_ACC_REG(x) RLG_ACC_CLASS(x)
// (waves per workgroup: RLG_BX_FWD_W / RLG_BX_BWD_W in the two kernels' files.  The tiles fill the LDS - one workgroup per
//  CU; rounds 3 - 5 ran one wave per SIMD because eight waves on the same tile had measured 1.7x SLOWER - with AGPR-pinned
//  accumulators, i.e. 128 + 128 registers and scratch; see RLG_ACC_CLASS above.)

static inline int bx_kc(int K) { return (K + 31) >> 5; }
static inline int bx_nb(int I) { return (I + 15) >> 4; }

// ------------------------------------------------------------------------------------------------
// The unit engine.  A wave's share of one layer: `nunits` output units, unit j = the NF consecutive 16-feature blocks
// ob_of(j) .. +NF-1 for NG row groups (all G, or the one group g_of(j) of a remainder unit):
//   pre(j);  acc[f][g] = sum over the KC chunks of  A(ob + f, chunk) x B(chunk, group);  [request unit j+1's first
//   chunk];  epi(j, acc)
// Two register banks per operand: while the MFMAs of chunk c issue from one bank, chunk c+1 is loaded into the
// other; the last chunk of a unit is followed by the loads of the NEXT unit's first chunk into bank 0, which the
// epilogue's VALU work covers.  sched_group_barrier deals the loads out between the MFMAs.
// ------------------------------------------------------------------------------------------------
template <int G, int NG, int NF, class ObOf, class GOf, class Pre, class Rot, class Epi>
__device__ __forceinline__ void bx_units(rsrc_t pr, unsigned layer_off, int KC, const char* tile_lane, int nunits,
                                         ObOf ob_of, GOf g_of, Pre pre, Rot rot, Epi epi, bool primed,
                                         long long* dbg = nullptr, int dbg_wave = 0, int* dbg_slot = nullptr) {
  if (nunits <= 0) return;
  const unsigned lane16 = static_cast<unsigned>(lane_id()) * 16u;
  const int block_stride = KC * kBxChunk;            // bytes between the fragments of consecutive blocks
  u32x4 a0[NF][kBxPlanes], a1[NF][kBxPlanes], b0[NG][kBxPlanes], b1[NG][kBxPlanes];
  auto unit_off = [&](int j) -> int {
    const int jj = j < nunits ? j : nunits - 1;
    if (kAbl & 64) return static_cast<int>(layer_off);      // timing only: every A load from the same fragments
    return __builtin_amdgcn_readfirstlane(static_cast<int>(layer_off) + ob_of(jj) * block_stride);
  };
  auto group_of = [&](int j) -> int {
    if constexpr (NG == G) return 0;
    const int jj = j < nunits ? j : nunits - 1;
    return g_of(jj);
  };
  auto load_a = [&](u32x4 (&av)[NF][kBxPlanes], int soff) {
    if ((kAbl & 256) && soff != static_cast<int>(layer_off)) return;       // timing only: no weight loads after the first
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int p = 0; p < kBxPlanes; ++p)
        av[f][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, lane16 + static_cast<unsigned>(p * kBxFrag),
                                                                                  (kAbl & 64) ? soff : soff + f * block_stride, 0));
    }
  };
  auto load_b = [&](u32x4 (&bv)[NG][kBxPlanes], int c, int g0) {
    if (kAbl & 128) return;                                  // timing only: no LDS reads
    const char* p = tile_lane + (c * G + g0) * kBxChunk;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int pl = 0; pl < kBxPlanes; ++pl) bv[g][pl] = *reinterpret_cast<const u32x4*>(p + (g * kBxPlanes + pl) * kBxFrag);
    }
  };

  if (kAbl & 128) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int pl = 0; pl < kBxPlanes; ++pl) {
        b0[g][pl] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        b1[g][pl] = b0[g][pl];
        asm volatile("" : "+v"(b0[g][pl]), "+v"(b1[g][pl]));
      }
    }
  }
  int uoff = unit_off(0);
  int g0 = group_of(0);
  load_a(a0, uoff);
  load_b(b0, 0, g0);
  // pre(j) requests what unit j's epilogue will read from global memory (into a "next" register set), rot() makes
  // that set current.  vmcnt retires in issue order on gfx9, so a load that goes to HBM stalls every LATER load's
  // first use: the request for unit j+1 is issued in front of unit j's LAST chunk - behind all of unit j's weight
  // loads, with unit j's epilogue to arrive in.
  // pre(nunits) is the caller's hook for the unit that FOLLOWS this call (the next call's, the next layer's first
  // unit); `primed`: the previous call has requested this call's first unit that way.
  if (!primed) pre(0);
  rot();

  f32x4 acc[NF][NG];
  // The six plane products of one accumulator are issued back to back, small ones first: a dependent chain runs at
  // the full rate (16.3 cycles per MFMA: the accumulator is forwarded inside the matrix core), a rotation through
  // many accumulators does not (21.7 cycles with 16 of them, profiles/r3_mfma_peak_probe.txt).
  // (a unit's first chunk starts from the constant 0 as SrcC: no accumulator is zeroed by hand)
  auto mfmas = [&](auto first_tag, const u32x4 (&av)[NF][kBxPlanes], const u32x4 (&bv)[NG][kBxPlanes]) {
    constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int t = 0; t < kBxProducts; ++t)
          acc[f][g] = bx_mfma(av[f][kBxPa[t]], bv[g][kBxPb[t]], (kFirst && t == 0) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[f][g]);
        if constexpr (kFirst) asm volatile("" : RLG_ACC_REG(acc[f][g]));      // accumulators live in AGPRs
      }
    }
  };
  // chunk c from bank kCur1; kPf: chunk c + 1 is requested into the other bank, its loads dealt out between the MFMAs
  auto step = [&](auto cur_tag, auto pf_tag, int c, auto first_tag) {
    constexpr bool kCur1 = decltype(cur_tag)::value;
    constexpr bool kPf = decltype(pf_tag)::value;
    if constexpr (kPf) {
      load_a(kCur1 ? a0 : a1, (kAbl & 64) ? uoff : uoff + (c + 1) * kBxChunk);
      load_b(kCur1 ? b0 : b1, c + 1, g0);
    }
    mfmas(first_tag, kCur1 ? a1 : a0, kCur1 ? b1 : b0);
    if constexpr (kPf) {
#pragma unroll
      for (int i = 0; i < kBxPlanes * NF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < kBxPlanes * NG; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (kBxProducts * NF * NG - kBxPlanes * (NF + NG) > 0)
        __builtin_amdgcn_sched_group_barrier(0x008, kBxProducts * NF * NG - kBxPlanes * (NF + NG), 0);
    }
    RLG_PIN();
  };
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  // behind a unit's last chunk: the next unit's first chunk is requested into bank 0 (this unit's again behind the
  // last one: harmless), then the epilogue, whose VALU work covers those loads
  auto finish_unit = [&](int j, int jn) {
    uoff = unit_off(jn);
    g0 = group_of(jn);
    load_a(a0, uoff);
    load_b(b0, 0, g0);
    RLG_PIN();
    // wait states between the last MFMA and the first VALU read of an accumulator (tools/audit_mfma.py)
    asm volatile("s_nop 7" : RLG_ACC_REG(acc[0][0]));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (f + g > 0) asm volatile("" : RLG_ACC_REG(acc[f][g]));
      }
    }
    epi(j, acc);
    rot();
    RLG_PIN();
  };

  if (KC == 1) {
    for (int j = 0; j < nunits; ++j) {
      const int jn = j + 1 < nunits ? j + 1 : j;
      pre(j + 1);
      RLG_PIN();
      step(F, F, 0, T);
      finish_unit(j, jn);
    }
    return;
  }
  // KC >= 2.  The unit loop is rotated - a pass = [chunks 1 .. KC-1 of unit j] [request unit j+1's first chunk]
  // [epilogue j] [chunk 0 of unit j+1] - so that the first use of those requested fragments sits in the same
  // straight-line code as the epilogue's stores: hipcc then waits with the exact vmcnt (the stores stay in flight);
  // across a loop edge it falls back to vmcnt(0), i.e. it would wait for the stores to be acknowledged as well.
  if (dbg) chain_stamp(dbg, dbg_wave, *dbg_slot);          // call start (first H claimed)
  step(F, T, 0, T);
  if (dbg) chain_stamp(dbg, dbg_wave, *dbg_slot);          // first chunk
  for (int j = 0; j < nunits; ++j) {
    const int jn = j + 1 < nunits ? j + 1 : j;
    int c = 1;                                     // chunk c sits in bank 1
    for (; c + 2 < KC; c += 2) {
      step(T, T, c, F);
      step(F, T, c + 1, F);
    }
    if (KC - c == 2) {
      step(T, T, c, F);
      pre(j + 1);
      RLG_PIN();
      step(F, F, c + 1, F);
    } else {
      pre(j + 1);
      RLG_PIN();
      step(T, F, c, F);
    }
    if (dbg) chain_stamp(dbg, dbg_wave, *dbg_slot);        // chunks done
    finish_unit(j, jn);
    if (dbg) chain_stamp(dbg, dbg_wave, *dbg_slot);        // epilogue + claim of the next H done
    if (j + 1 < nunits) step(F, T, 0, T);
    if (dbg) chain_stamp(dbg, dbg_wave, *dbg_slot);        // first chunk of the next unit
  }
}

}  // namespace rlg
