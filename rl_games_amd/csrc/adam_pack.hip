// Adam + the split-bf16 chain's weight planes in one launch (round 4; a file of its own since round 6).
//
// Why its own translation unit: this file is compiled with -fno-slp-vectorize (csrc/Makefile).  Under plain -O3 hipcc's SLP
// vectoriser packs adjacent scalar fp32 adds / multiplies of the Adam arithmetic into v_pk_add_f32 / v_pk_mul_f32, and
// several of those results are converted to fp64 by the very next instruction (v_cvt_f64_f32 writing the pk-op's SOURCE
// registers) - the sequence the round-5 diagnosis of the two-rank exp_avg_sq desynchronisation ended at without naming a
// cause (profiles/r5_two_rank_sync.txt; this kernel itself was clean in 89 two-rank runs).  Round 6 measured that
// v_pk_add_f32 runs on the pipe it shares with the matrix core (profiles/r6_coexec_bf16.txt) - nothing the optimiser launch
// gains anything from.  Scalar code here, tools/audit_mfma.py lists any build that lands on the pattern again, and the
// agent checks the ranks' parameter bits once per epoch (multi_gpu_param_check).
#include "mlp_chain_bx.hpp"
#include "optim_common.hpp"

namespace rlg {

long long chain_bx_plane_offsets(int num_layers, const int* in_features, const int* out_features, int direction,
                                 unsigned* offsets);                   // csrc/mlp_chain_bx.hip
long long chain_bx_both_offset(int num_layers, const int* in_features, const int* out_features);

// ---- optimiser step that leaves the weight planes behind (round 4) -------------------------------------------------
// adam_step_kernel (optim.hip) + chain_pack_planes_kernel as ONE launch: the thread that updates a 4 x 4 block of a
// weight matrix W [O][I] holds, afterwards, 4 consecutive k of 4 rows (forward operand A = W: one 8-byte store per
// row and plane) and 4 consecutive k of 4 columns (backward operand A = W^T: one 8-byte store per column and plane) of
// the NEW weights.  Same Adam arithmetic per element as adam_update (optim_common.hpp); the planes are the same bytes
// chain_pack_planes_kernel writes (zero padding outside the matrices is never touched: the buffer is packed once in
// full before the first step).  Everything of the arena that is not one of the chain's matrices (biases, sigma) is
// updated by the flat ranges at the end of the grid.
// (Round 4 also had a form with one thread per (row, 4 columns) - 9.0 us instead of 12.8 us - and round 4's lean kernels an
//  Adam + fp32-fragments launch of the same build, adam_frags_kernel.  Both are gone (round 5): two ranks sharing one GPU ended
//  epochs with a 16-lane group's exp_avg_sq one update apart in 20 % / 80 - 100 % of the runs, while this form (0 of 49) and
//  the plain rlg_adam_step (0 of 40) never did; the defect follows the compiled code of that kernel family, not any of its
//  parts - profiles/r5_two_rank_sync.txt.)
constexpr int kApMaxRanges = 2 * kChainMaxLayers + 2;
struct AdamPackArgs {
  AdamArgs adam;
  // per weight matrix: element offset in the arena, shape, byte offsets of its forward / backward fragments (-1: none)
  long long w_off[kChainMaxLayers];
  int O[kChainMaxLayers], I[kChainMaxLayers];
  long long fwd_off[kChainMaxLayers], bwd_off[kChainMaxLayers];
  int item_begin[kChainMaxLayers + 1];     // first 4 x 4 block of matrix L; [num] = total
  int num;
  unsigned char* planes;
  // the rest of the arena as flat ranges [begin, end)
  long long r_begin[kApMaxRanges], r_end[kApMaxRanges];
  int nranges;
  int matrix_blocks;                       // workgroups that walk 4 x 4 blocks; the flat ranges take the others
};

__device__ __forceinline__ void ap_store8(unsigned char* p, const unsigned (&w)[2]) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2*>(p) = u32x2{w[0], w[1]};
}

__global__ __launch_bounds__(256) void adam_pack_kernel(AdamPackArgs ap) {
  const AdamArgs& a = ap.adam;
  __shared__ float sh_clip;
  __shared__ float sh_norm;
  __shared__ double scratch[256 / kWave];
  // ---- everything this thread will need is requested FIRST (the gradient-norm reduction below is a chain of two memory
  //      round trips and two barriers: the loads of the block overlap with it instead of following it)
  const bool matrix_block = static_cast<int>(blockIdx.x) < ap.matrix_blocks;
  const int item = static_cast<int>(blockIdx.x) * 256 + threadIdx.x;
  const bool has_item = matrix_block && item < ap.item_begin[ap.num];
  int L = 0, O = 0, I = 0, o0 = 0, i0 = 0;
  f32x4 pn[4], g4[4], p4[4], m4[4], v4[4];
  long long idx[4] = {0, 0, 0, 0};
  if (has_item) {
    for (int j = 1; j < ap.num; ++j) L = (item >= ap.item_begin[j]) ? j : L;
    O = ap.O[L];
    I = ap.I[L];
    const int niq = I >> 2;
    const int local = item - ap.item_begin[L];
    const int oq = local / niq, iq = local - oq * niq;
    o0 = 4 * oq;
    i0 = 4 * iq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool in = o0 + r < O;
      idx[r] = ap.w_off[L] + static_cast<long long>(in ? o0 + r : o0) * I + i0;
      g4[r] = *reinterpret_cast<const f32x4*>(a.grads + idx[r]);
      p4[r] = *reinterpret_cast<const f32x4*>(a.params + idx[r]);
      m4[r] = *reinterpret_cast<const f32x4*>(a.exp_avg + idx[r]);
      v4[r] = *reinterpret_cast<const f32x4*>(a.exp_avg_sq + idx[r]);
    }
  }
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0u;
  const long long step = *a.step_counter;
  const int cur = static_cast<int>((step - 1) & 1);
  const double lr = a.lr_slots[cur];

  double sq[1] = {0.0};
  if (a.norm_partials) {           // (as adam_step_kernel: every workgroup computes the same sum in the same order)
    int b = threadIdx.x;
    for (; b + 3 * 256 < a.norm_blocks; b += 4 * 256) {
      const double v0 = a.norm_partials[b], v1 = a.norm_partials[b + 256];
      const double v2 = a.norm_partials[b + 2 * 256], v3 = a.norm_partials[b + 3 * 256];
      sq[0] += v0;
      sq[0] += v1;
      sq[0] += v2;
      sq[0] += v3;
    }
    for (; b < a.norm_blocks; b += 256) sq[0] += a.norm_partials[b];
    block_sum<1, 256>(sq, scratch);
  }
  if (threadIdx.x == 0) {
    float coef = 1.0f, total_norm = 0.0f;
    if (a.norm_partials) {
      total_norm = static_cast<float>(sqrt(sq[0]));
      coef = adam_clip_coef(a.max_norm, total_norm);
    }
    sh_clip = coef;
    sh_norm = total_norm;
  }
  __syncthreads();
  const float clip = sh_clip;
  const AdamScalars k = adam_scalars(a, step, lr);
#ifdef RLG_ADAM_TRACE
  AdamTraceAcc tr;
  if (matrix_block && has_item && !skip) {
    for (int r = 0; r < 4; ++r)
      if (o0 + r < O)
        for (int e = 0; e < 4; ++e) adam_trace_in(tr, a, step, idx[r] + e, g4[r][e], p4[r][e], m4[r][e], v4[r][e]);
  }
#endif

  if (matrix_block) {
    if (has_item && !skip) {
      f32x4 gc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g = (g4[r][e] * a.grad_scale) * clip;
          gc[r][e] = g;
          float p = p4[r][e];
          if (k.wd != 0.0f) g = g + k.wd * p;
          float m = m4[r][e];
          m = m + k.w1 * (g - m);
          float v = v4[r][e];
          v = v * k.b2 + (k.w2 * g) * g;
          const float denom = sqrt_rn(v) / k.bc2_sqrt + k.eps;
          p = p - k.step_size * (m / denom);
          m4[r][e] = m;
          v4[r][e] = v;
          p4[r][e] = p;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pn[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (o0 + r < O) {
          *reinterpret_cast<f32x4*>(a.grads + idx[r]) = gc[r];
          *reinterpret_cast<f32x4*>(a.exp_avg + idx[r]) = m4[r];
          *reinterpret_cast<f32x4*>(a.exp_avg_sq + idx[r]) = v4[r];
          *reinterpret_cast<f32x4*>(a.params + idx[r]) = p4[r];
          pn[r] = p4[r];
#ifdef RLG_ADAM_TRACE
          for (int e = 0; e < 4; ++e) adam_trace_out(tr, idx[r] + e, gc[r][e], p4[r][e], m4[r][e], v4[r][e]);
#endif
        }
      }
      // element slot of feature k inside its 32-feature chunk: lane group q = (k % 16) / 4, half = (k % 32) / 16
      if (ap.fwd_off[L] >= 0) {
        // A = W: block of 16 rows o, chunks over k = i
        const int KC = ((I + 31) >> 5);
        const int c = i0 >> 5, rr = i0 & 31, q = (rr & 15) >> 2, half = rr >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + r;
          if (o < O) {
            unsigned pl[kBxPlanes][2];
            bx_split4(pn[r], kBxScaleW, pl);
            unsigned char* dst = ap.planes + ap.fwd_off[L] + (static_cast<long long>(o >> 4) * KC + c) * kBxChunk +
                                 ((o & 15) + 16 * q) * 16 + half * 8;
#pragma unroll
            for (int p = 0; p < kBxPlanes; ++p) ap_store8(dst + p * kBxFrag, pl[p]);
          }
        }
      }
      if (ap.bwd_off[L] >= 0) {
        // A = W^T: block of 16 rows i, chunks over k = o; the 4 consecutive k are rows o0 .. o0 + 3 (zero past O)
        const int KC = ((O + 31) >> 5);
        const int c = o0 >> 5, rr = o0 & 31, q = (rr & 15) >> 2, half = rr >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          const f32x4 col = {pn[0][e], pn[1][e], pn[2][e], pn[3][e]};
          unsigned pl[kBxPlanes][2];
          bx_split4(col, kBxScaleW, pl);
          unsigned char* dst = ap.planes + ap.bwd_off[L] + (static_cast<long long>(i >> 4) * KC + c) * kBxChunk +
                               ((i & 15) + 16 * q) * 16 + half * 8;
#pragma unroll
          for (int p = 0; p < kBxPlanes; ++p) ap_store8(dst + p * kBxFrag, pl[p]);
        }
      }
    }
  } else if (!skip) {
    // the flat ranges: one thread per element, ranges concatenated
    long long t = (static_cast<long long>(blockIdx.x) - ap.matrix_blocks) * 256 + threadIdx.x;
    for (int r = 0; r < ap.nranges; ++r) {
      const long long len = ap.r_end[r] - ap.r_begin[r];
      if (t < len) {
#ifdef RLG_ADAM_TRACE
        adam_update_traced(tr, a, k, step, ap.r_begin[r] + t, clip);
#else
        adam_update(a, k, ap.r_begin[r] + t, clip);
#endif
        break;
      }
      t -= len;
    }
  }
#ifdef RLG_ADAM_TRACE
  adam_trace_flush(a, step, tr);
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_trace_scalars(a, step, clip, sh_norm, lr);
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) adam_finish(a, cur, lr, skip, sh_norm, clip);
}


}  // namespace rlg

extern "C" {

// rlg_adam_step that also leaves the chain's weight planes (both directions, the layout of rlg_mlp_chain_pack_planes
// direction 2) for the NEW weights - see adam_pack_kernel.  Requirements (else hipErrorInvalidValue, and the caller
// uses rlg_adam_step + rlg_mlp_chain_pack_planes): every weight matrix lies inside [params, params + n), 16-byte
// aligned relative to it, with in_features % 4 == 0; `planes` has been packed in full once (the zero padding).
int rlg_adam_step_pack(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                       const double* norm_partials_or_null, int norm_blocks, float grad_scale, float max_norm,
                       double* lr_slots, const long long* step_counter, double beta1, double beta2, double eps,
                       double weight_decay, int schedule_kind, const float* kl_or_null, float kl_scale,
                       double kl_threshold, double min_lr, double max_lr, double lr_multiplier, float* stats_out_or_null,
                       const unsigned* skip_flag_or_null, int num_layers, const float* const* weights,
                       const int* in_features, const int* out_features, void* planes, void* stream) {
  using namespace rlg;
  if (n <= 0 || !step_counter || num_layers < 1 || num_layers > kChainMaxLayers || planes == nullptr)
    return static_cast<int>(hipErrorInvalidValue);
  if (schedule_kind == 1 && !kl_or_null) return static_cast<int>(hipErrorInvalidValue);
  AdamPackArgs ap;
  AdamArgs& a = ap.adam;
  a.params = params;
  a.grads = grads;
  a.exp_avg = exp_avg;
  a.exp_avg_sq = exp_avg_sq;
  a.n = n;
  a.norm_partials = norm_partials_or_null;
  a.norm_blocks = norm_blocks;
  a.grad_scale = grad_scale;
  a.max_norm = max_norm;
  a.lr_slots = lr_slots;
  a.step_counter = step_counter;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.schedule_kind = schedule_kind;
  a.kl = kl_or_null;
  a.kl_scale = kl_scale;
  a.kl_threshold = kl_threshold;
  a.min_lr = min_lr;
  a.max_lr = max_lr;
  a.lr_multiplier = lr_multiplier;
  a.stats_out = stats_out_or_null;
  a.skip_flag = skip_flag_or_null;
  RLG_ADAM_TRACE_FILL(a);
  unsigned foff[kChainMaxLayers], boff[kChainMaxLayers];
  const long long ftotal = chain_bx_plane_offsets(num_layers, in_features, out_features, 0, foff);
  const long long btotal = chain_bx_plane_offsets(num_layers, in_features, out_features, 1, boff);
  const long long bbase = chain_bx_both_offset(num_layers, in_features, out_features);
  if (bbase + btotal >= static_cast<long long>(kOob) || ftotal >= static_cast<long long>(kOob)) return static_cast<int>(hipErrorInvalidValue);
  ap.num = num_layers;
  ap.planes = static_cast<unsigned char*>(planes);
  int items = 0;
  // matrices sorted by arena offset -> the gaps between them are the flat ranges
  long long begin[kChainMaxLayers], end[kChainMaxLayers];
  for (int L = 0; L < num_layers; ++L) {
    const long long off = weights[L] - params;
    const long long cnt = static_cast<long long>(in_features[L]) * out_features[L];
    if (off < 0 || off + cnt > n || (off & 3) != 0 || (in_features[L] & 3) != 0) return static_cast<int>(hipErrorInvalidValue);
    ap.w_off[L] = off;
    ap.O[L] = out_features[L];
    ap.I[L] = in_features[L];
    ap.fwd_off[L] = foff[L];
    ap.bwd_off[L] = (L >= 1) ? bbase + boff[L] : -1;
    ap.item_begin[L] = items;
    items += ((out_features[L] + 3) >> 2) * (in_features[L] >> 2);
    begin[L] = off;
    end[L] = off + cnt;
  }
  ap.item_begin[num_layers] = items;
  for (int x = 0; x < num_layers; ++x)            // (insertion sort of <= 8 intervals)
    for (int y = x + 1; y < num_layers; ++y)
      if (begin[y] < begin[x]) { std::swap(begin[x], begin[y]); std::swap(end[x], end[y]); }
  ap.nranges = 0;
  long long pos = 0, flat = 0;
  for (int x = 0; x <= num_layers; ++x) {
    const long long stop = (x < num_layers) ? begin[x] : n;
    if (stop < pos) return static_cast<int>(hipErrorInvalidValue);      // overlapping matrices
    if (stop > pos) {
      if (ap.nranges >= kApMaxRanges) return static_cast<int>(hipErrorInvalidValue);
      ap.r_begin[ap.nranges] = pos;
      ap.r_end[ap.nranges] = stop;
      ++ap.nranges;
      flat += stop - pos;
    }
    if (x < num_layers) pos = end[x];
  }
  ap.matrix_blocks = (items + 255) / 256;
  const int grid = ap.matrix_blocks + static_cast<int>((flat + 255) / 256);
  hipLaunchKernelGGL(adam_pack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), ap);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
