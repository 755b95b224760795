// Running statistics and normalisation for gfx950 (MI355X).
//
// Replaces
//   * RunningMeanStd.forward / _update_mean_var_count_from_moments
//     (rl_games/algos_torch/running_mean_std.py:69-114, :55-67) for observations (every
//     minibatch forward, models.py:54-56) and values/returns (a2c_common.py:1600-1620);
//   * the batch advantage normalisation (a2c_common.py:1622-1634, torch.std is UNBIASED),
//     its masked variant (torch_ext.py:172-191) and the EMA variant
//     GeneralizedMovingStats 'mean_std' (rl_games/algos_torch/moving_mean_std.py:52-61,
//     :83-134, :136-150).
//
// Structure: (1) a bandwidth-bound moments pass producing per-block fp64 partial sums
// (sum x, sum x^2 per column, no atomics -> bit-reproducible), (2) a tiny finalise kernel
// that folds the partials into the fp64/int64 running state with the reference's Chan
// merge, (3) a bandwidth-bound element-wise normalise pass.  State dtypes match the
// reference's registered buffers (running_mean/var float64, count int64) so checkpoints
// are interchangeable.

#include "rlg_device.hpp"

namespace rlg {

// ---------------------------------------------------------------------------------
// (1) column moments of a row-major [rows, C] fp32 matrix
// ---------------------------------------------------------------------------------
// Lane mapping: a row is `upr` units of UNIT floats; a wave covers rpp = 64/upr consecutive
// rows per pass (upr <= 64) so that its 64 lanes read one contiguous span.  Each lane keeps
// fp64 accumulators for its UNIT columns; lanes that share a column are combined through LDS.

constexpr int kMomBlock = 256;
constexpr int kMomWaves = kMomBlock / kWave;

template <int UNIT>
struct UnitLoad;
template <>
struct UnitLoad<4> {
  static __device__ __forceinline__ void load(const float* p, float (&x)[4]) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
    x[0] = q[0];
    x[1] = q[1];
    x[2] = q[2];
    x[3] = q[3];
  }
};
template <>
struct UnitLoad<1> {
  static __device__ __forceinline__ void load(const float* p, float (&x)[1]) { x[0] = *p; }
};

// partials layout: [gridDim.x][2*C + 1]  = {sum[C], sumsq[C], count}
template <int UNIT>
__global__ __launch_bounds__(kMomBlock) void column_moments_kernel(
    const float* __restrict__ x, const float* __restrict__ row_mask, long long rows, int C,
    double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [kMomWaves][64][2*UNIT] + C*2+1
  const int upr = C / UNIT;
  const int lane = lane_id();
  const int wave = wave_id();
  const long long gwave = static_cast<long long>(blockIdx.x) * kMomWaves + wave;
  const long long nwaves = static_cast<long long>(gridDim.x) * kMomWaves;
  // blockIdx.y = segment: consecutive bands of `rows` rows (the minibatches of an epoch), own partial rows
  x += static_cast<long long>(blockIdx.y) * rows * C;
  if (row_mask) row_mask += static_cast<long long>(blockIdx.y) * rows;
  double* out = partials + (static_cast<long long>(blockIdx.y) * gridDim.x + blockIdx.x) * (2 * C + 1);
  double* block_acc = smem + kMomWaves * kWave * 2 * UNIT;       // [2*C + 1]
  for (int i = threadIdx.x; i < 2 * C + 1; i += kMomBlock) block_acc[i] = 0.0;
  __syncthreads();

  double cnt = 0.0;
  for (int cb = 0; cb < upr; cb += kWave) {        // column blocks of 64 units
    const int w_upr = min(kWave, upr - cb);        // units of this column block
    const int rpp = kWave / w_upr;                 // rows per wave pass
    const int sub = lane / w_upr;
    const int cu = lane - sub * w_upr;
    const bool active = sub < rpp;
    double s[UNIT], ss[UNIT];
#pragma unroll
    for (int u = 0; u < UNIT; ++u) s[u] = ss[u] = 0.0;
    const long long groups = (rows + rpp - 1) / rpp;
    constexpr int kU = 4;   // independent row groups in flight per lane
    for (long long g0 = gwave; g0 < groups; g0 += nwaves * kU) {
      float v[kU][UNIT];
      float m[kU];
      bool ok[kU];
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const long long g = g0 + q * nwaves;
        const long long row = g * rpp + sub;
        ok[q] = active && g < groups && row < rows;
        const long long r_safe = ok[q] ? row : 0;
        UnitLoad<UNIT>::load(x + r_safe * C + static_cast<long long>(cb + cu) * UNIT, v[q]);
        m[q] = 1.0f;
        if (row_mask) m[q] = row_mask[r_safe];
      }
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        if (!ok[q]) continue;
        if (cb == 0 && cu == 0) cnt += static_cast<double>(m[q]);
#pragma unroll
        for (int u = 0; u < UNIT; ++u) {
          const double d = static_cast<double>(v[q][u]) * static_cast<double>(m[q]);
          s[u] += d;
          ss[u] = fma(d, static_cast<double>(v[q][u]), ss[u]);
        }
      }
    }
    // combine lanes/waves that own the same columns (fixed order -> deterministic)
    double* mine = smem + (wave * kWave + lane) * 2 * UNIT;
#pragma unroll
    for (int u = 0; u < UNIT; ++u) {
      mine[u] = s[u];
      mine[UNIT + u] = ss[u];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < w_upr * UNIT; j += kMomBlock) {
      const int u_idx = j / UNIT, u = j - u_idx * UNIT;
      double a = 0.0, b = 0.0;
      for (int w = 0; w < kMomWaves; ++w) {
        for (int sb = 0; sb < rpp; ++sb) {
          const double* p = smem + (w * kWave + sb * w_upr + u_idx) * 2 * UNIT;
          a += p[u];
          b += p[UNIT + u];
        }
      }
      const int col = (cb + u_idx) * UNIT + u;
      block_acc[col] = a;
      block_acc[C + col] = b;
    }
    __syncthreads();
  }
  // row count: every lane that owns column-unit 0 of column-block 0 counted its rows
  cnt = wave_sum(cnt);
  if (lane == 0) smem[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int w = 0; w < kMomWaves; ++w) c += smem[w];
    block_acc[2 * C] = c;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C + 1; i += kMomBlock) out[i] = block_acc[i];
}

// table[seg][j] = sum over the per-block partial rows of segment seg, in block order (deterministic):
// one {sum[C], sumsq[C], count} row per minibatch, folded later by the fused forward's prologue.
__global__ __launch_bounds__(256) void moments_rows_kernel(const double* __restrict__ partials, int nblocks,
                                                           int W, double* __restrict__ table) {
  const double* src = partials + static_cast<long long>(blockIdx.x) * nblocks * W;
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    double s = 0.0;
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[static_cast<long long>(b + u) * W + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; ++b) s += src[static_cast<long long>(b) * W + j];
    table[static_cast<long long>(blockIdx.x) * W + j] = s;
  }
}

// ---------------------------------------------------------------------------------
// (2) finalise: fold partial sums into the running state (Chan merge, fp64)
// ---------------------------------------------------------------------------------

// chan_merge (running_mean_std.py:55-67) lives in rlg_device.hpp: the fused forward folds with it too.

// mode 0: unmasked  - population variance, batch_count = rows            (:74-75, :83)
// mode 1: masked    - mean/var of get_mean_var_with_masks (unbiased, denominators clamped,
//                     torch_ext.py:182-191), batch_count = total rows     (:72, :83)
// mode 2: selected  - statistics of the rows with mask==1 only, batch_count = #selected
//                     (the value normaliser's `values[valid]` path, a2c_common.py:1609-1611)
//
// One 64-lane block per column: lanes stride over the per-block partials, wave-reduce, lane 0
// merges.  The shared `count` is bumped by whichever block finishes last (ticket), i.e. after
// every block has read the old value.
__global__ __launch_bounds__(64) void rms_update_kernel(const double* __restrict__ partials,
                                                        int nblocks, int C, long long total_rows,
                                                        int mode, double* __restrict__ running_mean,
                                                        double* __restrict__ running_var,
                                                        long long* __restrict__ count,
                                                        unsigned int* __restrict__ ticket) {
  const int W = 2 * C + 1;
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double s = 0.0, ss = 0.0, n_sum = 0.0;
  for (int b = lane; b < nblocks; b += kWave) {
    const double* p = partials + static_cast<long long>(b) * W;
    s += p[c];
    ss += p[C + c];
    n_sum += p[2 * C];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  n_sum = wave_sum(n_sum);
  if (lane != 0) return;
  const long long old_count = *count;
  const double batch_count = (mode == 2) ? n_sum : static_cast<double>(total_rows);
  double bm, bv;
  if (mode == 1) {
    const double sm = fmax(n_sum, 1.0);
    bm = s / sm;
    const double min_sqr = ss / sm - (s / sm) * (s / sm);
    bv = min_sqr * sm / fmax(sm - 1.0, 1.0);
  } else {
    const double n = fmax(n_sum, 1.0);
    bm = s / n;
    bv = fmax(ss / n - bm * bm, 0.0);
  }
  // the reference's batch moments are fp32 tensors: round once before the fp64 merge
  bm = static_cast<double>(static_cast<float>(bm));
  bv = static_cast<double>(static_cast<float>(bv));
  double mean = running_mean[c], var = running_var[c];
  if (batch_count > 0.0) chan_merge(mean, var, static_cast<double>(old_count), bm, bv, batch_count);
  running_mean[c] = mean;
  running_var[c] = var;
  // every block read `count` above; the last one to arrive publishes the new count.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const unsigned int t = atomicAdd(ticket, 1u);
  if (t == static_cast<unsigned int>(C) - 1u) {
    *ticket = 0u;
    *count = old_count + static_cast<long long>(batch_count);
  }
}

// ---------------------------------------------------------------------------------
// (3) element-wise normalise / de-normalise with the running state
// ---------------------------------------------------------------------------------
// mode 0: y = clamp((x - mean32) / sqrt(var32 + eps), -5, 5)            (:112-113)
// mode 1: y = sqrt(var32 + eps) * clamp(x, -5, 5) + mean32   (denorm)    (:106-107)
// mode 2: y = x / sqrt(var32 + eps)                          (norm_only) (:110)
template <int UNIT>
__global__ __launch_bounds__(256) void rms_apply_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, long long rows, int C,
                                                        const double* __restrict__ running_mean,
                                                        const double* __restrict__ running_var,
                                                        float eps, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];  // mean32[C], denom[C]
  float* mean32 = smem_f;
  float* denom = smem_f + C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    mean32[c] = static_cast<float>(running_mean[c]);
    denom[c] = sqrt_rn(static_cast<float>(running_var[c]) + eps);
  }
  __syncthreads();
  const long long total_units = rows * C / UNIT;
  const int upr = C / UNIT;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total_units;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int c0 = static_cast<int>(i % upr) * UNIT;
    float v[UNIT], o[UNIT];
    UnitLoad<UNIT>::load(x + i * UNIT, v);
#pragma unroll
    for (int u = 0; u < UNIT; ++u) {
      const float m = mean32[c0 + u], d = denom[c0 + u];
      if (mode == 0) {
        o[u] = clamp_nan((v[u] - m) / d, -5.0f, 5.0f);
      } else if (mode == 1) {
        o[u] = d * clamp_nan(v[u], -5.0f, 5.0f) + m;
      } else {
        o[u] = v[u] / d;
      }
    }
    if constexpr (UNIT == 4) {
      f32x4 q = {o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(y + i * 4) = q;
    } else {
      y[i] = o[0];
    }
  }
}

// ---------------------------------------------------------------------------------
// prepare_dataset epilogue (value_size == 1): driven by the GAE kernel's partial moments
// ---------------------------------------------------------------------------------

// Generic producer of the {sum adv, sum adv^2, sum v, sum v^2, sum ret, sum ret^2[, sum mask]}
// partials that prepare_finalize_kernel consumes, for batches that did not come out of the
// fused GAE kernel (user-built batch_dict, masked rows).  With a mask every term is weighted
// by it (binary masks: the moments of the valid rows).
__global__ __launch_bounds__(256) void triple_moments_kernel(
    const float* __restrict__ adv, const float* __restrict__ values,
    const float* __restrict__ returns, const float* __restrict__ mask, long long B,
    double* __restrict__ partials, int stride) {
  __shared__ double scratch[7 * 4];
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < B;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const double w = mask ? static_cast<double>(mask[i]) : 1.0;
    const double a = adv[i], v = values[i], r = returns[i];
    m[0] += a * w;
    m[1] += a * a * w;
    m[2] += v * w;
    m[3] += v * v * w;
    m[4] += r * w;
    m[5] += r * r * w;
    m[6] += w;
  }
  block_sum<7, 256>(m, scratch);
  if (threadIdx.x == 0) {
    for (int k = 0; k < stride; ++k) partials[static_cast<long long>(blockIdx.x) * stride + k] = m[k];
  }
}

struct PrepareStats {
  // written by prepare_finalize_kernel, read by prepare_apply_kernel
  float v_mean, v_denom;       // value normaliser after the `values` update
  float r_mean, r_denom;       // ... after the `returns` update
  float a_mean, a_denom;       // advantage mean, std + 1e-8 (or EMA mean / std)
  float pad0, pad1;
};

// flags
constexpr int kPrepNormValue = 1;      // normalize_value
constexpr int kPrepNormAdv = 2;        // normalize_advantage (batch statistics)
constexpr int kPrepFreezeCritic = 4;   // freeze_critic: normalise with frozen stats
constexpr int kPrepEmaAdv = 8;         // normalize_rms_advantage (GeneralizedMovingStats mean_std)

__global__ __launch_bounds__(256) void prepare_finalize_kernel(const double* __restrict__ gae_partials, int ntiles,
                                        int stride, long long B, int flags,
                                        double* __restrict__ running_mean,
                                        double* __restrict__ running_var,
                                        long long* __restrict__ count, float eps,
                                        float* __restrict__ ema_mean, float* __restrict__ ema_sqrs,
                                        int* __restrict__ ema_step, float ema_decay, float ema_factor,
                                        float ema_max, float ema_eps,
                                        PrepareStats* __restrict__ out) {
  __shared__ double scratch[7 * 4];
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = threadIdx.x; t < ntiles; t += 256) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (k < stride) m[k] += gae_partials[static_cast<long long>(t) * stride + k];
    }
  }
  block_sum<7, 256>(m, scratch);
  if (threadIdx.x != 0) return;
  // stride 7 = masked batch: statistics of the valid rows only (a2c_common.py:1605-1615,
  // torch_ext.py:172-191); n_valid rows feed the value normaliser's count.
  const bool masked = stride == 7;
  const double n_rows = masked ? m[6] : static_cast<double>(B);
  const double n = masked ? fmax(m[6], 1.0) : static_cast<double>(B);
  const long long n_count = masked ? static_cast<long long>(m[6]) : B;
  PrepareStats st = {0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 0.f};
  if (flags & kPrepNormValue) {
    double mean = running_mean[0], var = running_var[0];
    long long cnt = *count;
    if (!(flags & kPrepFreezeCritic)) {
      // values first (a2c_common.py:1617-1618) ...
      double bm = static_cast<double>(static_cast<float>(m[2] / n));
      double bv = static_cast<double>(static_cast<float>(fmax(m[3] / n - (m[2] / n) * (m[2] / n), 0.0)));
      if (n_rows > 0.0) chan_merge(mean, var, static_cast<double>(cnt), bm, bv, n_rows);
      cnt += n_count;
    }
    st.v_mean = static_cast<float>(mean);
    st.v_denom = sqrt_rn(static_cast<float>(var) + eps);
    if (!(flags & kPrepFreezeCritic)) {
      // ... then returns (:1619), a second, separate merge
      double bm = static_cast<double>(static_cast<float>(m[4] / n));
      double bv = static_cast<double>(static_cast<float>(fmax(m[5] / n - (m[4] / n) * (m[4] / n), 0.0)));
      if (n_rows > 0.0) chan_merge(mean, var, static_cast<double>(cnt), bm, bv, n_rows);
      cnt += n_count;
      running_mean[0] = mean;
      running_var[0] = var;
      *count = cnt;
    }
    st.r_mean = static_cast<float>(mean);
    st.r_denom = sqrt_rn(static_cast<float>(var) + eps);
    if (masked) {
      // masked branch (a2c_common.py:1605-1615): both updates happen first, then values AND
      // returns are normalised in eval mode with the final statistics
      st.v_mean = st.r_mean;
      st.v_denom = st.r_denom;
    }
  }
  if (flags & kPrepEmaAdv) {
    // moving_mean_std.py:119-122 (_update_stats 'mean_std'), :57-61 (_get_stats)
    float mean = ema_mean[0], sqrs = ema_sqrs[0];
    if (n_rows > 0.0) {   // an all-masked batch leaves the statistics untouched (:108-112)
      const float x_mean = static_cast<float>(m[0] / n);
      const float x_sqr = static_cast<float>(m[1] / n);
      *ema_step += 1;
      const float factor = ema_factor;  // fp32(1 - decay), rounded from the Python double
      mean = mean * ema_decay;
      mean = mean + factor * x_mean;
      sqrs = sqrs * ema_decay;
      sqrs = sqrs + factor * x_sqr;
      ema_mean[0] = mean;
      ema_sqrs[0] = sqrs;
    }
    const float var = sqrs - mean * mean;
    st.a_mean = mean;
    st.a_denom = sqrt_rn(fmaxf(var, 1.0f / (ema_max * ema_max)) + ema_eps);
  } else if (flags & kPrepNormAdv) {
    // advantages.mean(), advantages.std() (unbiased) + 1e-8                  a2c_common.py:1634
    const double mean = m[0] / n;
    double var;
    if (masked) {
      // get_mean_var_with_masks: (sum (a m)^2/sm - mean^2) * sm / max(sm - 1, 1)
      var = fmax(m[1] / n - mean * mean, 0.0) * n / fmax(n - 1.0, 1.0);
    } else {
      var = (n > 1.0) ? fmax(m[1] - n * mean * mean, 0.0) / (n - 1.0) : 0.0;
    }
    st.a_mean = static_cast<float>(mean);
    st.a_denom = static_cast<float>(sqrt(var)) + 1e-8f;
  }
  *out = st;
}

// values_out = norm(values), returns_out = norm(returns), adv_out = (adv - mean)/denom.
// Outputs may alias the inputs (in place).
__global__ __launch_bounds__(256) void prepare_apply_kernel(
    const float* __restrict__ values, const float* __restrict__ returns,
    const float* __restrict__ advantages, float* __restrict__ values_out,
    float* __restrict__ returns_out, float* __restrict__ adv_out, long long B4, long long B,
    int flags, const PrepareStats* __restrict__ stp) {
  const PrepareStats st = *stp;
  const bool nv = flags & kPrepNormValue;
  const bool na = (flags & (kPrepNormAdv | kPrepEmaAdv)) != 0;
  const bool ema = flags & kPrepEmaAdv;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < B4;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(values)[i];
    f32x4 r = reinterpret_cast<const f32x4*>(returns)[i];
    f32x4 a = reinterpret_cast<const f32x4*>(advantages)[i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (nv) {
        v[u] = clamp_nan((v[u] - st.v_mean) / st.v_denom, -5.0f, 5.0f);
        r[u] = clamp_nan((r[u] - st.r_mean) / st.r_denom, -5.0f, 5.0f);
      }
      if (na) {
        a[u] = (a[u] - st.a_mean) / st.a_denom;
        if (ema) a[u] = clamp_nan(a[u], -5.0f, 5.0f);
      }
    }
    reinterpret_cast<f32x4*>(values_out)[i] = v;
    reinterpret_cast<f32x4*>(returns_out)[i] = r;
    reinterpret_cast<f32x4*>(adv_out)[i] = a;
  }
  // tail (B not a multiple of 4, or unaligned bases)
  const long long i = B4 * 4 + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < B) {
    float v = values[i], r = returns[i], a = advantages[i];
    if (nv) {
      v = clamp_nan((v - st.v_mean) / st.v_denom, -5.0f, 5.0f);
      r = clamp_nan((r - st.r_mean) / st.r_denom, -5.0f, 5.0f);
    }
    if (na) {
      a = (a - st.a_mean) / st.a_denom;
      if (ema) a = clamp_nan(a, -5.0f, 5.0f);
    }
    values_out[i] = v;
    returns_out[i] = r;
    adv_out[i] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-rank synchronisation of the running normalisers (multi-GPU, once per epoch).
// replaces rl_games/common/a2c_common.py: _running_stats_totals :43-47, seed_stats_sync_snapshot :50-58,
// merge_rank_stats :61-93, broadcast_rank_stats :124-141 - there a dozen tiny fp64 torch ops and THREE
// collectives per normaliser; here every normaliser of the agent is a segment [count | sum or mean[D] |
// sum of squares or var[D]] of ONE flat fp64 buffer: pack launch -> one collective -> apply launch.
// Arithmetic: the reference's fp64 ops in the reference's order, one rounding each (-ffp-contract=off);
// the count travels as a double (exact below 2^53).
// ------------------------------------------------------------------------------------------------
constexpr int kStatsMaxSegments = 8;

struct StatsSegments {
  double* mean[kStatsMaxSegments];
  double* var[kStatsMaxSegments];
  long long* count[kStatsMaxSegments];
  long long offset[kStatsMaxSegments];     // first double of the segment in the flat buffers
  int dim[kStatsMaxSegments];
  int has_snapshot[kStatsMaxSegments];
  int num;
};

// totals of a state (a2c_common.py:43-47): count, mean * count, (var + mean^2) * count
__device__ __forceinline__ void stats_totals(const StatsSegments& sg, int s, int i, double& value) {
  const double n = static_cast<double>(*sg.count[s]);
  if (i == 0) {
    value = n;
  } else if (i <= sg.dim[s]) {
    value = sg.mean[s][i - 1] * n;
  } else {
    const double m = sg.mean[s][i - 1 - sg.dim[s]];
    value = (sg.var[s][i - 1 - sg.dim[s]] + m * m) * n;
  }
}

// mode 0: out = totals - snapshot   (the epoch's deltas, :72-77; a segment without snapshot sends its totals)
// mode 1: snapshot = totals         (seed after a checkpoint load, :50-58; `out` unused)
// mode 2: out = raw state [count, mean, var]  (broadcast mode, :124-141)
__global__ __launch_bounds__(256) void stats_sync_pack_kernel(StatsSegments sg, double* snapshot, double* out, int mode) {
  for (int s = 0; s < sg.num; ++s) {
    const int len = 1 + 2 * sg.dim[s];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
      const long long at = sg.offset[s] + i;
      if (mode == 2) {
        const int D = sg.dim[s];
        out[at] = i == 0 ? static_cast<double>(*sg.count[s]) : (i <= D ? sg.mean[s][i - 1] : sg.var[s][i - 1 - D]);
        continue;
      }
      double cur;
      stats_totals(sg, s, i, cur);
      if (mode == 1) snapshot[at] = cur;
      else out[at] = sg.has_snapshot[s] ? cur - snapshot[at] : cur;
    }
  }
}

// mode 0: totals = snapshot + reduced deltas -> state, snapshot = totals   (:78-93)
// mode 2: state = reduced (rank 0's raw state)
// Two passes per segment inside one launch are avoided by recomputing n per thread: every element only needs
// the segment's count slot and, for the variance, the matching first-moment slot.
__global__ __launch_bounds__(256) void stats_sync_apply_kernel(StatsSegments sg, double* snapshot, const double* reduced,
                                                               int mode) {
  for (int s = 0; s < sg.num; ++s) {
    const int D = sg.dim[s];
    const long long o = sg.offset[s];
    const bool based = sg.has_snapshot[s] != 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= D; i += gridDim.x * blockDim.x) {
      if (mode == 2) {
        if (i == 0) *sg.count[s] = static_cast<long long>(reduced[o]);
        else {
          sg.mean[s][i - 1] = reduced[o + i];
          sg.var[s][i - 1] = reduced[o + D + i];
        }
        continue;
      }
      // base + delta with the reference's zero base for a segment without snapshot (0.0 + x is exact)
      const double n = (based ? snapshot[o] : 0.0) + reduced[o];
      if (i == 0) continue;                         // the count is written last (below), by its own thread
      const double s1 = (based ? snapshot[o + i] : 0.0) + reduced[o + i];
      const double s2 = (based ? snapshot[o + D + i] : 0.0) + reduced[o + D + i];
      const double mean = s1 / n;
      double var = s2 / n - mean * mean;
      var = var < 1e-8 ? 1e-8 : var;                // clamp_(min=1e-8); a NaN passes through like torch's clamp
      sg.mean[s][i - 1] = mean;
      sg.var[s][i - 1] = var;
      snapshot[o + i] = s1;
      snapshot[o + D + i] = s2;
    }
  }
  // counts: after every thread of the grid has read snapshot[o] (one block only - see the launcher)
  __syncthreads();
  if (mode == 0 && threadIdx.x < sg.num && blockIdx.x == 0) {
    const int s = threadIdx.x;
    const long long o = sg.offset[s];
    const double n = (sg.has_snapshot[s] ? snapshot[o] : 0.0) + reduced[o];
    *sg.count[s] = static_cast<long long>(n);
    snapshot[o] = n;
  }
}

}  // namespace rlg

extern "C" {

int rlg_column_moments_num_blocks(long long rows, int cols) {
  // enough waves to fill the chip, never more than one 4-wave block per 256 rows
  long long need = (rows * cols + 256LL * 64 - 1) / (256LL * 64);
  if (need < 1) need = 1;
  if (need > 1024) need = 1024;
  return static_cast<int>(need);
}

int rlg_column_moments(const float* x, const float* row_mask_or_null, long long rows, int cols,
                       double* partials, int num_blocks, void* stream) {
  using namespace rlg;
  if (rows <= 0 || cols <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (cols % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  const int unit = vec ? 4 : 1;
  const size_t shm = (static_cast<size_t>(kMomWaves) * kWave * 2 * unit + 2 * cols + 1) * sizeof(double);
  if (shm > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
  if (vec) {
    hipLaunchKernelGGL((column_moments_kernel<4>), dim3(num_blocks), dim3(kMomBlock), shm, st, x,
                       row_mask_or_null, rows, cols, partials);
  } else {
    hipLaunchKernelGGL((column_moments_kernel<1>), dim3(num_blocks), dim3(kMomBlock), shm, st, x,
                       row_mask_or_null, rows, cols, partials);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

// Column moments of `num_segments` consecutive bands of rows_per_segment rows (the minibatches of an
// epoch) in two launches: table[seg] = {sum[cols], sumsq[cols], rows}.  `partials` is scratch of
// num_segments * num_blocks * (2*cols+1) doubles.
int rlg_column_moments_segments(const float* x, long long rows_per_segment, int cols, int num_segments,
                                double* partials, int num_blocks, double* table, void* stream) {
  using namespace rlg;
  if (rows_per_segment <= 0 || cols <= 0 || num_segments <= 0 || num_segments > 65535 || num_blocks <= 0)
    return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (cols % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  const int unit = vec ? 4 : 1;
  const size_t shm = (static_cast<size_t>(kMomWaves) * kWave * 2 * unit + 2 * cols + 1) * sizeof(double);
  if (shm > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
  const dim3 grid(num_blocks, num_segments);
  if (vec) {
    hipLaunchKernelGGL((column_moments_kernel<4>), grid, dim3(kMomBlock), shm, st, x,
                       static_cast<const float*>(nullptr), rows_per_segment, cols, partials);
  } else {
    hipLaunchKernelGGL((column_moments_kernel<1>), grid, dim3(kMomBlock), shm, st, x,
                       static_cast<const float*>(nullptr), rows_per_segment, cols, partials);
  }
  hipLaunchKernelGGL(moments_rows_kernel, dim3(num_segments), dim3(256), 0, st, partials, num_blocks,
                     2 * cols + 1, table);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_rms_update(const double* partials, int num_blocks, int cols, long long total_rows,
                   int mode, double* running_mean, double* running_var, long long* count,
                   unsigned int* ticket, void* stream) {
  if (cols <= 0 || mode < 0 || mode > 2 || !ticket) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::rms_update_kernel, dim3(cols), dim3(rlg::kWave), 0,
                     static_cast<hipStream_t>(stream), partials, num_blocks, cols, total_rows, mode,
                     running_mean, running_var, count, ticket);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_rms_apply(const float* x, float* y, long long rows, int cols, const double* running_mean,
                  const double* running_var, float eps, int mode, void* stream) {
  using namespace rlg;
  if (rows <= 0 || cols <= 0) return 0;
  if (mode < 0 || mode > 2) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = (cols % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  const long long units = rows * cols / (vec ? 4 : 1);
  long long grid = (units + 256 * 4 - 1) / (256 * 4);
  if (grid < 1) grid = 1;
  if (grid > 4096) grid = 4096;
  const size_t shm = 2 * static_cast<size_t>(cols) * sizeof(float);
  if (vec) {
    hipLaunchKernelGGL((rms_apply_kernel<4>), dim3(static_cast<int>(grid)), dim3(256), shm, st, x,
                       y, rows, cols, running_mean, running_var, eps, mode);
  } else {
    hipLaunchKernelGGL((rms_apply_kernel<1>), dim3(static_cast<int>(grid)), dim3(256), shm, st, x,
                       y, rows, cols, running_mean, running_var, eps, mode);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_prepare_stats_bytes(void) { return static_cast<int>(sizeof(rlg::PrepareStats)); }

int rlg_triple_moments_num_blocks(long long batch) {
  long long b = (batch + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > 512) b = 512;
  return static_cast<int>(b);
}

int rlg_triple_moments(const float* advantages, const float* values, const float* returns,
                       const float* mask_or_null, long long batch, double* partials, int num_blocks,
                       void* stream) {
  if (batch <= 0) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::triple_moments_kernel, dim3(num_blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), advantages, values, returns, mask_or_null,
                     batch, partials, mask_or_null ? 7 : 6);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_prepare_finalize(const double* gae_partials, int num_tiles, int stride, long long batch,
                         int flags,
                         double* value_running_mean, double* value_running_var,
                         long long* value_count, float eps, float* ema_mean, float* ema_sqrs,
                         int* ema_step, float ema_decay, float ema_factor, float ema_max,
                         float ema_eps, void* stats_out, void* stream) {
  if (stride != 6 && stride != 7) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::prepare_finalize_kernel, dim3(1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gae_partials, num_tiles, stride, batch, flags,
                     value_running_mean, value_running_var, value_count, eps, ema_mean, ema_sqrs,
                     ema_step, ema_decay, ema_factor, ema_max, ema_eps,
                     static_cast<rlg::PrepareStats*>(stats_out));
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_prepare_apply(const float* values, const float* returns, const float* advantages,
                      float* values_out, float* returns_out, float* advantages_out, long long batch,
                      int flags, const void* stats, void* stream) {
  if (batch <= 0) return 0;
  const uintptr_t all = reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(returns) |
                        reinterpret_cast<uintptr_t>(advantages) |
                        reinterpret_cast<uintptr_t>(values_out) |
                        reinterpret_cast<uintptr_t>(returns_out) |
                        reinterpret_cast<uintptr_t>(advantages_out);
  const long long b4 = (all % 16 == 0) ? batch / 4 : 0;
  const long long tail = batch - b4 * 4;
  long long grid;
  if (tail == 0) {
    grid = (b4 + 255) / 256;
    if (grid > 2048) grid = 2048;          // grid-stride over the vector part
  } else {
    const long long work = b4 > tail ? b4 : tail;
    grid = (work + 255) / 256;             // the scalar tail needs one thread per element
  }
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(rlg::prepare_apply_kernel, dim3(static_cast<int>(grid)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), values, returns, advantages, values_out,
                     returns_out, advantages_out, b4, batch, flags,
                     static_cast<const rlg::PrepareStats*>(stats));
  RLG_RETURN_LAUNCH_STATUS();
}

static int stats_segments_fill(rlg::StatsSegments& sg, int num_segments, double* const* means, double* const* vars,
                               long long* const* counts, const int* dims, const int* has_snapshot) {
  if (num_segments < 1 || num_segments > rlg::kStatsMaxSegments) return 1;
  long long off = 0;
  for (int s = 0; s < num_segments; ++s) {
    if (!means[s] || !vars[s] || !counts[s] || dims[s] < 1) return 1;
    sg.mean[s] = means[s];
    sg.var[s] = vars[s];
    sg.count[s] = counts[s];
    sg.dim[s] = dims[s];
    sg.has_snapshot[s] = has_snapshot ? has_snapshot[s] : 1;
    sg.offset[s] = off;
    off += 1 + 2LL * dims[s];
  }
  sg.num = num_segments;
  return 0;
}

long long rlg_stats_sync_flat_size(int num_segments, const int* dims) {
  long long n = 0;
  for (int s = 0; s < num_segments; ++s) n += 1 + 2LL * dims[s];
  return n;
}

int rlg_stats_sync_pack(int num_segments, double* const* means, double* const* vars, long long* const* counts,
                        const int* dims, const int* has_snapshot, double* snapshot, double* out, int mode,
                        void* stream) {
  rlg::StatsSegments sg;
  if (stats_segments_fill(sg, num_segments, means, vars, counts, dims, has_snapshot) || mode < 0 || mode > 2 ||
      (mode != 2 && !snapshot) || (mode != 1 && !out))
    return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(rlg::stats_sync_pack_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sg, snapshot,
                     out, mode);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_stats_sync_apply(int num_segments, double* const* means, double* const* vars, long long* const* counts,
                         const int* dims, const int* has_snapshot, double* snapshot, const double* reduced, int mode,
                         void* stream) {
  rlg::StatsSegments sg;
  if (stats_segments_fill(sg, num_segments, means, vars, counts, dims, has_snapshot) || (mode != 0 && mode != 2) ||
      !reduced || (mode == 0 && !snapshot))
    return static_cast<int>(hipErrorInvalidValue);
  // ONE workgroup: the count slot of the snapshot is read by every thread and rewritten behind a barrier
  hipLaunchKernelGGL(rlg::stats_sync_apply_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sg, snapshot,
                     reduced, mode);
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
