// What mlp_chain.hip (unit-structured and pipelined exact-product kernels, the C entry points of the chain) and
// mlp_chain_lean.hip (the lean 16-row kernels and theirs) share: the forward prologue (device) and the host-side helpers
// behind the entry points.  Round 5: the two kernel generations that survive live in files of their own.
#pragma once

#include "mlp_chain_common.hpp"

namespace rlg {

// Forward prologue of a workgroup of W waves that owns 16 * G rows from row0 on: observation tile -> LDS (fragment
// layout) in tile_a, normalised on the way (RunningMeanStd state folded first when a.rms_batch is given; the
// mean / denominator scratch lives in tile_b), normalised observations written to a.xn.  Ends with a barrier.
template <int G, int W>
__device__ __forceinline__ void chain_fwd_prologue(const ChainArgs& a, float* tile_a, float* tile_b, long long row0, int lane,
                                                   int wave, int& stamp) {
    const int in0 = a.layer[0].in;
    const int in0p = (in0 + 3) & ~3;
    const bool norm = a.rms_mean != nullptr;
    const int KC0 = (in0 + 15) >> 4;
    const int nfrag = KC0 * G;
    const bool xv = vec4_ok(a.x, a.ldx);
    const bool xnv = a.xn != nullptr && vec4_ok(a.xn, in0);
    // kProBatch fragments per wave at a time: every load is issued before the first one is used (a
    // rolled loop would pay one HBM round trip per fragment), and the first batch is requested
    // BEFORE the normaliser statistics are prepared, so both latencies overlap.
    constexpr int kProBatch = 8;
    f32x4 xin[kProBatch];
    auto load_frags = [&](int u0) {
#pragma unroll
      for (int k = 0; k < kProBatch; ++k) {
        const int u = u0 + k * W;
        const int c = u / G;
        const int g = u - c * G;
        const long long row = row0 + g * 16 + (lane & 15);
        xin[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (!(kAbl & 32) && u < nfrag && row < a.rows) xin[k] = load_row4(a.x, a.ldx, row, c * 16 + 4 * (lane >> 4), in0, xv);
      }
    };
    auto put_frags = [&](int u0) {
#pragma unroll
      for (int k = 0; k < kProBatch; ++k) {
        const int u = u0 + k * W;
        if (u < nfrag) {
          const int c = u / G;
          const int g = u - c * G;
          const long long row = row0 + g * 16 + (lane & 15);
          const int f = c * 16 + 4 * (lane >> 4);
          f32x4 v = xin[k];
          if (row < a.rows) {
            if (norm) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (f + e < in0) v[e] = clamp_nan((v[e] - tile_b[f + e]) / tile_b[in0p + f + e], -5.0f, 5.0f);
              }
            }
            if (a.xn) store_row4(a.xn, in0, row, f, in0, v, xnv);
          }
          *reinterpret_cast<f32x4*>(tile_a + (u * 64 + lane) * 4) = v;
        }
      }
    };
    load_frags(wave);
    if (norm) {
      chain_norm_stats<W>(a, tile_b, in0, in0p);
      __syncthreads();
    }
    put_frags(wave);
    for (int u0 = wave + W * kProBatch; u0 < nfrag; u0 += W * kProBatch) {
      load_frags(u0);
      put_frags(u0);
    }
    chain_stamp(a.dbg, wave, stamp);
    __syncthreads();
}

// ---- host side (defined in mlp_chain.hip) ------------------------------------------------------------------------------
// ChainArgs from the C arrays: layer table, activation kinds; non-zero: a shape / pointer the kernels do not take
int chain_fill(ChainArgs& args, int num_layers, const float* const* weights, const int* in_features, const int* out_features,
               const int* acts);
// tools only (rlg_mlp_chain_debug_stamps): phase stamps of the next launches, or nullptr
long long* chain_debug_stamps();
// rlg_mlp_chain_time_next: the HIP events the NEXT chain launch carries on its dispatch (taken = cleared)
void chain_take_events(hipEvent_t* ev_start, hipEvent_t* ev_stop);
// rlg_mlp_chain_gradient_maxima: where the NEXT backward launch leaves its gradient maxima (taken = cleared)
void chain_take_gradient_maxima(float** entries, int* stride);
static inline bool vec4_ok_host(const void* p, long long ld) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (ld & 3) == 0; }

}  // namespace rlg
