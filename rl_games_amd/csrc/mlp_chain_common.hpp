// Shared pieces of the fused MLP chain kernels (mlp_chain.hip: exact f32 products; mlp_chain_bx.hip: split-bf16
// products on pre-split planes): launch arguments, buffer-resource helpers, activation maths.
#pragma once

#include "rlg_device.hpp"
#include "ppo_loss_tile.hpp"
#include "bx_form.hpp"
#include "rlg_hip.h"

#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include <utility>

// Timing-only ablations for tools/ablate_chain.sh (-DRLG_ABL=mask builds a library that computes WRONG results and
// shows what a phase costs): 1 no bias/activation maths, 2 no global stores of the epilogues, 4 no epilogue at all,
// 8 no remainder units, 16 no barriers between layers, 32 no prologue loads, 64 no weight traffic (A loads out of range),
// 128 (pipelined kernels) no LDS reads of the B fragments
#ifndef RLG_ABL
#define RLG_ABL 0
#endif

namespace rlg {

constexpr int kAbl = RLG_ABL;
constexpr int kChainMaxLayers = 8;
// K-split scratch of the forward (partial fragments of 256 floats): G = 1: up to 2 units x 3 parts (8 waves) or
// 4 x 1; G = 2: up to 2 units x 1 part; G = 4: no split (a remainder block already has one unit per wave)
static inline int chain_split_floats(int G) { return (G == 1 ? 6 : (G == 2 ? 2 : 0)) * 256; }
constexpr int kChainWideBlocks = 384;   // up to this many 16-row workgroups run with 8 waves instead of 4
constexpr unsigned kOob = 0x40000000u;     // byte offset far outside every weight buffer

enum : int { kChIdentity = 0, kChElu = 1, kChRelu = 2, kChTanh = 3 };

struct ChainLayer {
  const float* w;        // [out, in] row-major
  const float* bias;     // [out] (forward)
  float* h;              // forward: activation output [rows, ldh] or nullptr;  backward: H of this layer (input)
  float* dz;             // backward: dZ of this layer [rows, lddz] (output; nullptr for the last layer)
  double* bias_partials; // backward: [gridDim.x, out] column sums of dZ, or nullptr
  long long ldh, lddz;
  int in, out;
  int act;
};

// weights -> bf16 plane fragments for the split-bf16 kernels (mlp_chain_bx.hip); a forward launch can carry the job as
// extra workgroups at the end of its grid (the planes of the SAME weights, for the backward launch that follows)
struct PackJob {
  const float* w;        // [out, in] row-major
  int in, I, K, KC;
  int transposed;        // A[i][k] = w[k][i] (backward) instead of w[i][k]
  int pair_begin;        // first (block, chunk) pair of this job in the launch
  unsigned dst_off;
};
struct PackArgs {
  PackJob job[kChainMaxLayers];
  int njobs, total_pairs;
  unsigned char* dst;
};

struct ChainArgs {
  ChainLayer layer[kChainMaxLayers];
  int num_layers;
  const float* x;              // forward: raw observations [rows, ldx]; backward: d(last layer output) [rows, ldx]
  long long ldx;
  const double* rms_mean;      // forward: RunningMeanStd state (fp64) or nullptr
  const double* rms_var;
  float rms_eps;
  // forward, training: fold this minibatch's column moments {sum[in], sumsq[in], rows} into the state
  // first (RunningMeanStd.forward in training mode updates, then normalises) - every block folds for
  // itself, block 0 publishes the new state to the OTHER buffer set (the next launch reads that one)
  const double* rms_batch;     // or nullptr: normalise with the state as it is
  const long long* rms_count;
  double* rms_mean_out;
  double* rms_var_out;
  long long* rms_count_out;
  float* xn;                   // forward: normalised observations out [rows, in0] (dW of layer 0 reads them) or nullptr
  long long rows;
  int lds_b_floats;
  int lds_split_floats;        // forward: offset of the K-split scratch (partial fragments of remainder units)
  int no_ksplit;               // tools (RLG_CHAIN_KSPLIT=0): remainder units without the K-split            // start of the second LDS region, in floats
  // pipelined kernels: ONE buffer resource over every weight matrix and bias vector (the flat parameter arena)
  const float* w_base;
  unsigned w_bytes;
  unsigned w_off[kChainMaxLayers];     // byte offset of layer[L].w from w_base
  unsigned b_off[kChainMaxLayers];     // byte offset of layer[L].bias from w_base (forward)
  // split-bf16 kernels (mlp_chain_bx.hip): the weights as pre-split bf16 plane fragments (rlg_mlp_chain_pack_planes)
  const void* planes;
  unsigned planes_bytes;
  unsigned p_off[kChainMaxLayers];     // byte offset of layer L's fragments
  int bx_handoff_off;                  // split-bf16 backward: byte offset in LDS of the loss tile's d heads, or -1
  int bx_handoff_ld;                   //   (the loss writes exactly the array the chain reads: no fence + re-load)
  int bx_scales_off;                   // split-fp16 backward: byte offset in LDS of the 64 row scales of the d heads tile
  float* amax;                         // split-fp16 backward: per-workgroup gradient maxima (csrc/bx_form.hpp kBxAmaxDz) or nullptr
  int amax_stride;                     //   entries per tensor
  // split-bf16 forward: byte offset of tile L (the input of layer L) in LDS, of the normaliser scratch; the tile of
  // layer bx_pass_layer (-1: none) does not fit and is produced / consumed in windows of bx_pass_chunks chunks
  int bx_tile_off[kChainMaxLayers + 1];
  int bx_stats_off;
  int bx_pass_layer, bx_pass_chunks;
  // forward: workgroups fwd_blocks .. gridDim.x-1 pack weight planes (pack.total_pairs > 0) instead of a row tile
  int fwd_blocks;
  PackArgs pack;
  long long* dbg;
  int with_loss;               // backward: evaluate the PPO loss of the tile first (LossArgs)              // tools only: [blocks][4 waves][32] shader-clock stamps per phase, or nullptr
};

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define RLG_PIN() __builtin_amdgcn_sched_barrier(0)

// phase stamps for tools/bench_mlp_chain.py --phases (one lane per wave; no effect when dbg is null)
__device__ __forceinline__ void chain_stamp(long long* dbg, int wave, int& slot) {
  if (dbg != nullptr) {
    const long long t = __builtin_amdgcn_s_memtime();
    if (lane_id() == 0 && slot < 32 && wave < 4) dbg[(static_cast<long long>(blockIdx.x) * 4 + wave) * 32 + slot] = t;
    ++slot;
  }
}

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

__device__ __forceinline__ float chain_act(float v, int act) {
  if (act == kChElu) return v > 0.0f ? v : __expf(v) - 1.0f;
  if (act == kChRelu) return v > 0.0f ? v : (v != v ? v : 0.0f);      // torch.relu(NaN) = NaN
  if (act == kChTanh) return tanhf(v);
  return v;
}
// act' from the layer OUTPUT h (aten's *_backward with is_result = true)
__device__ __forceinline__ float chain_act_grad(float h, int act) {
  if (act == kChElu) return h > 0.0f ? 1.0f : h + 1.0f;
  if (act == kChRelu) return h > 0.0f ? 1.0f : 0.0f;
  if (act == kChTanh) return 1.0f - h * h;
  return 1.0f;
}

// Kernel arguments that the epilogues use are copied into scalar registers ONCE per layer and made
// opaque, otherwise hipcc re-materialises them as s_load from the kernarg segment at every use (a
// scalar-cache round trip plus an lgkmcnt(0) per use - thousands of cycles per output block).
__device__ __forceinline__ int pin_s(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long pin_s(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned long long>(v) & 0xffffffffu));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned long long>(v) >> 32));
  return static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
}
template <class T>
__device__ __forceinline__ T* pin_s(T* ptr) {
  return reinterpret_cast<T*>(pin_s(reinterpret_cast<long long>(ptr)));
}

// activation of a fragment: one wave-uniform switch per fragment, not per element.  HACT >= 0: the
// launch knows that every layer is either HACT or identity (the usual network: one hidden activation,
// linear heads) - the kernel then carries one activation body instead of all of them (48 inlined
// tanhf bodies pushed the G = 4 forward past the 64 KiB instruction cache); HACT = kChAny: per layer.
constexpr int kChAny = -1;
template <int HACT>
__device__ __forceinline__ f32x4 chain_act4(f32x4 v, int act) {
  if constexpr (HACT != kChAny) {
    if (act == kChIdentity) return v;
    act = HACT;
  }
  if (act == kChElu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : __expf(v[e]) - 1.0f;
  } else if (act == kChRelu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : (v[e] != v[e] ? v[e] : 0.0f);
  } else if (act == kChTanh) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
  }
  return v;
}
// acc * act'(h), act' from the layer OUTPUT h (aten's *_backward with is_result = true)
__device__ __forceinline__ f32x4 chain_act_grad4(f32x4 d, f32x4 h, int act) {
  if (act == kChElu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = d[e] * (h[e] > 0.0f ? 1.0f : h[e] + 1.0f);
  } else if (act == kChRelu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = d[e] * (h[e] > 0.0f ? 1.0f : 0.0f);
  } else if (act == kChTanh) {
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = d[e] * (1.0f - h[e] * h[e]);
  }
  return d;
}

// Pointers that went through pin_s (an integer round trip) have lost their address space: hipcc then emits FLAT
// loads / stores, which count on lgkmcnt as well and turn every later LDS wait into lgkmcnt(0).  All arrays
// of these kernels are global memory: say so at the access.
template <class T>
using glob_t = T __attribute__((address_space(1)));
template <class T>
__device__ __forceinline__ glob_t<T>* as_global(T* p) { return (glob_t<T>*)p; }
template <class T>
__device__ __forceinline__ const glob_t<T>* as_global(const T* p) { return (const glob_t<T>*)p; }

// 4 consecutive features [f, f+4) of row `row` of a row-major array, masked to `width`
__device__ __forceinline__ void store_row4(float* base, long long ld, long long row, int f, int width,
                                           const f32x4& v, bool vec_ok) {
  glob_t<float>* p = as_global(base + row * ld + f);
  if (vec_ok && f + 4 <= width) {
    *(glob_t<f32x4>*)p = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (f + e < width) p[e] = v[e];
    }
  }
}
__device__ __forceinline__ f32x4 load_row4(const float* base, long long ld, long long row, int f, int width,
                                           bool vec_ok) {
  const glob_t<float>* p = as_global(base + row * ld + f);
  if (vec_ok && f + 4 <= width) return *(const glob_t<f32x4>*)p;
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (f + e < width) v[e] = p[e];
  }
  return v;
}
__device__ __forceinline__ bool vec4_ok(const void* p, long long ld) {
  return aligned16(p) && (ld & 3) == 0;
}

using b128_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(std::declval<rsrc_t>(), 0u, 0, 0));
__device__ __forceinline__ void buf_store4(rsrc_t r, unsigned off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(b128_t, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store1(rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
// bound of a resource over the (at most) tile_rows rows of a [rows, ld] fp32 array that are left from its base on
__device__ __forceinline__ unsigned tile_bytes(long long rows_left, long long tile_rows, long long ld) {
  const long long r = rows_left < tile_rows ? rows_left : tile_rows;
  return static_cast<unsigned>(r * ld * 4);
}

// sum over the 16 lanes of a row (lanes that share l >> 4), result in all of them: rotations within the row on the
// DPP path (no LDS round trips), fixed order
__device__ __forceinline__ float row16_sum(float t) {
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128, 0xf, 0xf, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x124, 0xf, 0xf, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x122, 0xf, 0xf, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x121, 0xf, 0xf, false));
  return t;
}

// mean32 / denominator of the observation normaliser into scratch[0 .. in0) / scratch[in0p .. in0p + in0), exactly
// like rms_apply_kernel mode 0 (running_mean_std.py:112-113); with a.rms_batch the minibatch's moments are folded into
// the state first (training-mode RunningMeanStd.forward) and workgroup 0 publishes the new state.  No barrier inside.
template <int W>
__device__ __forceinline__ void chain_norm_stats(const ChainArgs& a, float* scratch, int in0, int in0p) {
  for (int f = threadIdx.x; f < in0; f += (64 * W)) {
    double mean = a.rms_mean[f], var = a.rms_var[f];
    if (a.rms_batch) {
      // rms_update_kernel mode 0: population moments of the minibatch, rounded to fp32 like the
      // reference's input.mean / input.var, Chan merge in fp64 (running_mean_std.py:55-67,:74-83)
      const double n = fmax(a.rms_batch[2 * in0], 1.0);
      double bm = a.rms_batch[f] / n;
      double bv = fmax(a.rms_batch[in0 + f] / n - bm * bm, 0.0);
      bm = static_cast<double>(static_cast<float>(bm));
      bv = static_cast<double>(static_cast<float>(bv));
      const long long old_count = *a.rms_count;
      chan_merge(mean, var, static_cast<double>(old_count), bm, bv, static_cast<double>(a.rows));
      if (blockIdx.x == 0) {
        a.rms_mean_out[f] = mean;
        a.rms_var_out[f] = var;
        if (f == 0) *a.rms_count_out = old_count + a.rows;
      }
    }
    scratch[f] = static_cast<float>(mean);
    scratch[in0p + f] = sqrt_rn(static_cast<float>(var) + a.rms_eps);
  }
}

// workgroup `block` (256 threads) of a pack job: one thread per lane of a (block, chunk) fragment pair.  Fragment
// (block ib, chunk c, plane p) = 64 lanes x 8 bf16 = 1 KiB at ((ib * KC + c) * 3 + p) KiB; lane l, element e holds
// A[16 ib + (l & 15)][32 c + (e < 4 ? 4 (l >> 4) + e : 16 + 4 (l >> 4) + e - 4)], zero outside the matrix.
__device__ __forceinline__ void chain_pack_planes_block(const PackArgs& a, int block, int tid) {
  const int t = block * 256 + tid;
  const int pair = t >> 6, lane = t & 63;
  if (pair >= a.total_pairs) return;
  int sel = 0;
  for (int j = 1; j < a.njobs; ++j) sel = (pair >= a.job[j].pair_begin) ? j : sel;
  const PackJob& J = a.job[sel];
  const int local = pair - J.pair_begin;
  const int ib = local / J.KC, c = local - ib * J.KC;
  const int i = ib * 16 + (lane & 15), q4 = 4 * (lane >> 4);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * c + (e < 4 ? q4 + e : 16 + q4 + e - 4);
    float v = 0.0f;
    if (i < J.I && k < J.K) v = J.transposed ? J.w[static_cast<long long>(k) * J.in + i] : J.w[static_cast<long long>(i) * J.in + k];
    x[e] = v;
  }
  u32x4 plane[kBxPlanes];
  bx_split8(x, kBxScaleW, plane);
  unsigned char* dst = a.dst + J.dst_off + static_cast<long long>(local) * kBxChunk + lane * 16;
#pragma unroll
  for (int p = 0; p < kBxPlanes; ++p) *reinterpret_cast<u32x4*>(dst + p * kBxFrag) = plane[p];
}
int chain_bx_pack_blocks(const PackArgs& a);
int chain_bx_pack_launch(const PackArgs& a, hipStream_t st);
// fills the pack job of one direction (2: both, into one buffer); false: nothing to pack / bad arguments / too many jobs
bool chain_bx_fill_pack(PackArgs& args, int num_layers, const float* const* weights, const int* in_features,
                        const int* out_features, int direction, void* planes);

// ---- mlp_chain_bx.hip ----------------------------------------------------------------------------------------------
// LDS bytes of the split-bf16 backward at 16 G rows per workgroup (fills lds_b_floats), -1: does not fit
int chain_bx_bwd_lds(ChainArgs& args, int G);
// byte offsets of every layer's plane fragments (direction 0: forward products, 1: backward), returns the total
long long chain_bx_plane_offsets(int num_layers, const int* in_features, const int* out_features, int direction,
                                 unsigned* offsets);
bool chain_bx_bwd_eligible(const ChainArgs& args);
int chain_bx_prepare();
// forward: LDS plan (fills bx_tile_off / bx_stats_off / bx_pass_*), bytes or -1; eligibility; launch
int chain_bx_fwd_plan(ChainArgs& args);
bool chain_bx_fwd_eligible(const ChainArgs& args);
int chain_bx_launch_fwd(const ChainArgs& args, int lds_bytes, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int chain_bx_fwd_prepare();
int chain_bx_launch_bwd(const ChainArgs& args, int G, int lds_bytes, hipStream_t st, const LossArgs* loss, hipEvent_t ev0,
                        hipEvent_t ev1);

}  // namespace rlg
