// GAE backward scan for gfx950 (MI355X).
//
// Replaces rl_games/triton_kernels/gae_kernel.py (_gae_kernel :17-60, _pytorch_gae
// :63-80, compute_gae :125-147) and the glue around it in
// rl_games/common/a2c_common.py (discount_values :729-734, mb_returns :1060,
// advantages = returns - values :1598).
//
// Arithmetic contract (bit-exact with the eager fp32 op chain of _pytorch_gae):
//   nnt   = 1 - done_next
//   delta = (r_t + (gamma * v_next) * nnt) - v_t
//   A_t   = delta + ((gamma*tau) * nnt) * A_{t+1}          A_H = 0
//   ret_t = A_t + v_t ;  adv_t = ret_t - v_t                (two roundings, NOT A_t)
// gamma and gamma*tau arrive already rounded to fp32 by the host (gamma*tau is
// multiplied in double first, as Python does).  The file is compiled with
// -ffp-contract=off, so none of the products above fuse into an FMA.
//
// Two kernels:
//  * gae_strided_kernel   - any layout / any H / V>1 / float or u8 dones.  One
//                           thread per (env, value) pair, loads prefetched eight
//                           timesteps at a time.  This is the function seam
//                           (compute_gae) for arbitrary views.
//  * gae_envmajor_kernel  - the rollout buffer's native layout: rewards/values
//                           [N, H] fp32 and dones [N, H] u8, H a compile-time
//                           multiple of 4 (<= 64).  One wave = 64 consecutive
//                           envs.  The wave's 64*H-float tile is fetched with
//                           fully coalesced 16-byte loads (1 KiB per wave
//                           instruction), transposed through a padded LDS tile
//                           so every lane owns one env's row in registers, scanned,
//                           transposed back and stored coalesced.  The same pass
//                           emits returns, returns-values and fp64 partial moments
//                           of (advantages, values, returns) for the normalisers.
//                           Algorithmic traffic: 4+4+1 B read, 4+4 B written per
//                           env-step = 17 B.

#include "rlg_device.hpp"
#include <hip/hip_ext.h>

namespace rlg {

// ---------------------------------------------------------------------------------
// General strided kernel
// ---------------------------------------------------------------------------------

struct GaeStrides {
  long long r_t, r_e, r_v;     // rewards      [H, N, V]
  long long v_t, v_e, v_v;     // values       [H, N, V]
  long long d_t, d_e;          // dones        [H, N]
  long long lv_e, lv_v;        // last_values  [N, V]
  long long ld_e;              // last_dones   [N]
  long long a_t, a_e, a_v;     // advs out     [H, N, V]
  long long q_t, q_e, q_v;     // returns out  [H, N, V] (optional)
};

template <typename DoneT>
__device__ __forceinline__ float done_to_float(DoneT d) {
  return static_cast<float>(d);
}

template <typename DoneT, int U>
__global__ __launch_bounds__(256) void gae_strided_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const DoneT* __restrict__ dones, const float* __restrict__ last_values,
    const DoneT* __restrict__ last_dones, float* __restrict__ advs,
    float* __restrict__ returns, GaeStrides s, int H, int N, int V, float gamma,
    float gamma_tau) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= static_cast<long long>(N) * V) return;
  const long long env = p / V;
  const long long k = p - env * V;

  const float* rp = rewards + env * s.r_e + k * s.r_v;
  const float* vp = values + env * s.v_e + k * s.v_v;
  const DoneT* dp = dones + env * s.d_e;
  float* ap = advs + env * s.a_e + k * s.a_v;
  float* qp = returns ? returns + env * s.q_e + k * s.q_v : nullptr;

  float nv = last_values[env * s.lv_e + k * s.lv_v];
  float nnt = 1.0f - done_to_float(last_dones[env * s.ld_e]);
  float A = 0.0f;

  int t = H - 1;
  for (; t >= U - 1; t -= U) {
    float r_[U], v_[U], d_[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const long long tt = t - j;
      r_[j] = rp[tt * s.r_t];
      v_[j] = vp[tt * s.v_t];
      d_[j] = done_to_float(dp[tt * s.d_t]);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const long long tt = t - j;
      const float vt = v_[j];
      const float delta = (r_[j] + (gamma * nv) * nnt) - vt;
      A = delta + (gamma_tau * nnt) * A;
      ap[tt * s.a_t] = A;
      if (qp) qp[tt * s.q_t] = A + vt;
      nv = vt;
      nnt = 1.0f - d_[j];
    }
  }
  for (; t >= 0; --t) {
    const float vt = vp[t * s.v_t];
    const float delta = (rp[t * s.r_t] + (gamma * nv) * nnt) - vt;
    A = delta + (gamma_tau * nnt) * A;
    ap[t * s.a_t] = A;
    if (qp) qp[t * s.q_t] = A + vt;
    nv = vt;
    nnt = 1.0f - done_to_float(dp[t * s.d_t]);
  }
}

// ---------------------------------------------------------------------------------
// Env-major fused kernel
// ---------------------------------------------------------------------------------

// Row stride (in floats) of the LDS tile.  +4 keeps every row 16-byte aligned for
// ds_read_b128 / ds_write_b128 and spreads 16 consecutive rows over all 64 banks
// (row stride 36 dwords for H=32: start banks 0,36,8,44,... - conflict free).
template <int H>
struct TileGeom {
  static constexpr int kRow = H + 4;
  static constexpr int kChunksPerRow = H / 4;           // 16-byte chunks per env row
  static constexpr int kChunksPerLane = H / 4;          // 64 rows * H/4 chunks / 64 lanes
  static constexpr int kTileFloats = kWave * kRow;
};

// Coalesced global -> registers: the wave's tile is `rows` env rows of H floats, contiguous
// in global memory starting at `g`.  Chunk c = k*64 + lane  ->  row c/(H/4), col 4*(c%(H/4)).
// Chunks past the last live row re-read chunk 0 (always valid) instead of branching, so
// all H/4 loads of the wave are issued back to back with no exec-mask juggling.
template <int H>
__device__ __forceinline__ void tile_load_issue(const float* __restrict__ g, int rows,
                                                f32x4 (&buf)[H / 4]) {
  constexpr int CPR = TileGeom<H>::kChunksPerRow;
  const int lane = lane_id();
  const int live_chunks = rows * CPR;
#pragma unroll
  for (int k = 0; k < H / 4; ++k) {
    const int c = k * kWave + lane;
    const int cc = (c < live_chunks) ? c : 0;
    buf[k] = *reinterpret_cast<const f32x4*>(g + cc * 4);
  }
}

template <int H>
__device__ __forceinline__ void tile_regs_to_lds(const f32x4 (&buf)[H / 4], float* tile) {
  constexpr int CPR = TileGeom<H>::kChunksPerRow;
  constexpr int ROW = TileGeom<H>::kRow;
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < H / 4; ++k) {
    const int c = k * kWave + lane;
    const int row = c / CPR;
    const int col = (c - row * CPR) * 4;
    *reinterpret_cast<f32x4*>(tile + row * ROW + col) = buf[k];
  }
}

template <int H>
__device__ __forceinline__ void tile_lds_to_global(const float* tile, float* __restrict__ g,
                                                   int rows) {
  constexpr int CPR = TileGeom<H>::kChunksPerRow;
  constexpr int ROW = TileGeom<H>::kRow;
  const int lane = lane_id();
  if (rows == kWave) {  // wave-uniform: full tile, unpredicated stores
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const int c = k * kWave + lane;
      const int row = c / CPR;
      const int col = (c - row * CPR) * 4;
      *reinterpret_cast<f32x4*>(g + c * 4) = *reinterpret_cast<const f32x4*>(tile + row * ROW + col);
    }
  } else {
    const int live_chunks = rows * CPR;
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const int c = k * kWave + lane;
      const int row = c / CPR;
      const int col = (c - row * CPR) * 4;
      if (c < live_chunks) {
        *reinterpret_cast<f32x4*>(g + c * 4) =
            *reinterpret_cast<const f32x4*>(tile + row * ROW + col);
      }
    }
  }
}

template <int H>
__device__ __forceinline__ void row_from_lds(const float* tile, float (&x)[H]) {
  constexpr int ROW = TileGeom<H>::kRow;
  const float* rowp = tile + lane_id() * ROW;
#pragma unroll
  for (int j = 0; j < H / 4; ++j) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(rowp + 4 * j);
    x[4 * j + 0] = q[0];
    x[4 * j + 1] = q[1];
    x[4 * j + 2] = q[2];
    x[4 * j + 3] = q[3];
  }
}

template <int H>
__device__ __forceinline__ void row_to_lds(float* tile, const float (&x)[H]) {
  constexpr int ROW = TileGeom<H>::kRow;
  float* rowp = tile + lane_id() * ROW;
#pragma unroll
  for (int j = 0; j < H / 4; ++j) {
    f32x4 q = {x[4 * j + 0], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
    *reinterpret_cast<f32x4*>(rowp + 4 * j) = q;
  }
}

// Waves per workgroup.  Every wave owns a private LDS tile and only ever synchronises with
// itself (wave barriers), so the block size is purely a dispatch-granularity choice; 4 waves
// (one per SIMD) measured ~3 % faster than single-wave workgroups on MI355X.
constexpr int kGaeWavesPerBlock = 4;

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving
  // LDS accesses across the hand-off between lanes of the same wave.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// kRaw = false: write returns + (returns - values) + fp64 partial moments.
// kRaw = true : write the raw GAE values A_t only (compute_gae seam on env-major views).
template <int H, bool kRaw>
__global__ __launch_bounds__(64 * kGaeWavesPerBlock) void gae_envmajor_kernel(
    const float* __restrict__ rewards,      // [N, H]
    const float* __restrict__ values,       // [N, H]
    const uint8_t* __restrict__ dones,      // [N, H]
    const float* __restrict__ last_values,  // [N]
    const uint8_t* __restrict__ last_dones, // [N]
    float* __restrict__ out0,               // kRaw ? gae [N,H] : returns [N,H]
    float* __restrict__ out1,               // kRaw ? unused    : advantages [N,H]
    double* __restrict__ partials,          // !kRaw: [num_tiles, 6] or nullptr
    int N, float gamma, float gamma_tau) {
  static_assert(H % 4 == 0 && H >= 4 && H <= 64, "unsupported horizon for the tile kernel");
  __shared__ __attribute__((aligned(16))) float lds[kGaeWavesPerBlock * 2 * TileGeom<H>::kTileFloats];
  const int tile = blockIdx.x * kGaeWavesPerBlock + wave_id();
  const int env0 = tile * kWave;
  if (env0 >= N) return;  // whole wave exits together; no block-wide barrier is used below
  float* tile_r = lds + wave_id() * 2 * TileGeom<H>::kTileFloats;
  float* tile_v = tile_r + TileGeom<H>::kTileFloats;

  const int rows = min(kWave, N - env0);
  const int lane = lane_id();
  const int env = env0 + lane;
  const bool live = lane < rows;
  const long long base = static_cast<long long>(env0) * H;

  // ---- issue every load of the wave up front (memory-level parallelism) ----
  f32x4 rbuf[H / 4], vbuf[H / 4];
  tile_load_issue<H>(rewards + base, rows, rbuf);
  tile_load_issue<H>(values + base, rows, vbuf);

  uint32_t dw[H / 4];  // this lane's env row of done flags, 4 per dword (little endian)
  {
    const uint8_t* drow = dones + static_cast<long long>(live ? env : env0) * H;
    if constexpr (H % 16 == 0) {
#pragma unroll
      for (int j = 0; j < H / 16; ++j) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(drow + 16 * j);
        dw[4 * j + 0] = q[0];
        dw[4 * j + 1] = q[1];
        dw[4 * j + 2] = q[2];
        dw[4 * j + 3] = q[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < H / 4; ++j) dw[j] = *reinterpret_cast<const uint32_t*>(drow + 4 * j);
    }
  }
  float nv = last_values[live ? env : env0];
  float nnt = 1.0f - static_cast<float>(last_dones[live ? env : env0]);

  // ---- transpose through LDS: coalesced chunks -> one env row per lane ----
  tile_regs_to_lds<H>(rbuf, tile_r);
  tile_regs_to_lds<H>(vbuf, tile_v);
  wave_lds_fence();
  float r[H], v[H];
  row_from_lds<H>(tile_r, r);
  row_from_lds<H>(tile_v, v);

  // ---- the recurrence, entirely in registers ----
  float A = 0.0f;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int t = H - 1 - i;
    const float vt = v[t];
    const float delta = (r[t] + (gamma * nv) * nnt) - vt;
    A = delta + (gamma_tau * nnt) * A;
    r[t] = kRaw ? A : A + vt;          // raw GAE, or returns = A + v (a2c_common.py:1060)
    nv = vt;
    const uint32_t dbyte = (dw[t >> 2] >> (8 * (t & 3))) & 0xffu;
    nnt = 1.0f - static_cast<float>(dbyte);
  }

  // ---- transpose back and store coalesced ----
  wave_lds_fence();
  row_to_lds<H>(tile_r, r);
  wave_lds_fence();
  if constexpr (kRaw) {
    tile_lds_to_global<H>(tile_r, out0 + base, rows);
  } else {
    // tile_v still holds the values in the coalesced chunk layout they were loaded in, so
    // advantages = returns - values (a2c_common.py:1598, the same single fp32 subtraction) and
    // the fp64 moments are formed here, chunk by chunk, on the way out: one LDS tile less to write
    // and no per-env advantage registers.
    constexpr int CPR = TileGeom<H>::kChunksPerRow;
    constexpr int ROW = TileGeom<H>::kRow;
    const int live_chunks = rows * CPR;
    double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < H / 4; ++k) {
      const int c = k * kWave + lane;
      const int row = c / CPR;
      const int col = (c - row * CPR) * 4;
      const f32x4 ret = *reinterpret_cast<const f32x4*>(tile_r + row * ROW + col);
      const f32x4 val = *reinterpret_cast<const f32x4*>(tile_v + row * ROW + col);
      const f32x4 adv = ret - val;
      if (rows == kWave || c < live_chunks) {
        *reinterpret_cast<f32x4*>(out0 + base + c * 4) = ret;
        *reinterpret_cast<f32x4*>(out1 + base + c * 4) = adv;
        if (partials) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double da = adv[e], dv = val[e], dr = ret[e];
            m[0] += da;
            m[1] = fma(da, da, m[1]);
            m[2] += dv;
            m[3] = fma(dv, dv, m[3]);
            m[4] += dr;
            m[5] = fma(dr, dr, m[5]);
          }
        }
      }
    }
    if (partials) {
#pragma unroll
      for (int k = 0; k < 6; ++k) m[k] = wave_sum(m[k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) partials[static_cast<long long>(tile) * 6 + k] = m[k];
      }
    }
  }
}

template <int H, bool kRaw>
static int launch_envmajor(const float* rewards, const float* values, const uint8_t* dones,
                           const float* last_values, const uint8_t* last_dones, float* out0,
                           float* out1, double* partials, int N, float gamma, float gamma_tau,
                           hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  const int tiles = (N + kWave - 1) / kWave;
  const int grid = (tiles + kGaeWavesPerBlock - 1) / kGaeWavesPerBlock;
  if (ev_start && ev_stop) {
    // events bound to THIS dispatch's begin/end timestamps (what rocprofv3 reports), not to
    // separate marker packets before/after it
    hipExtLaunchKernelGGL((gae_envmajor_kernel<H, kRaw>), dim3(grid), dim3(kWave * kGaeWavesPerBlock), 0,
                          stream, ev_start, ev_stop, 0, rewards, values, dones, last_values, last_dones,
                          out0, out1, partials, N, gamma, gamma_tau);
  } else {
    hipLaunchKernelGGL((gae_envmajor_kernel<H, kRaw>), dim3(grid), dim3(kWave * kGaeWavesPerBlock), 0,
                       stream, rewards,
                       values, dones, last_values, last_dones, out0, out1, partials, N, gamma,
                       gamma_tau);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

template <bool kRaw>
static int dispatch_envmajor(int H, const float* rewards, const float* values,
                             const uint8_t* dones, const float* last_values,
                             const uint8_t* last_dones, float* out0, float* out1,
                             double* partials, int N, float gamma, float gamma_tau,
                             hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
#define RLG_CASE(HH)                                                                          \
  case HH:                                                                                    \
    return launch_envmajor<HH, kRaw>(rewards, values, dones, last_values, last_dones, out0,   \
                                     out1, partials, N, gamma, gamma_tau, stream, ev_start, ev_stop)
  switch (H) {
    RLG_CASE(4);
    RLG_CASE(8);
    RLG_CASE(12);
    RLG_CASE(16);
    RLG_CASE(20);
    RLG_CASE(24);
    RLG_CASE(28);
    RLG_CASE(32);
    RLG_CASE(36);
    RLG_CASE(40);
    RLG_CASE(44);
    RLG_CASE(48);
    RLG_CASE(52);
    RLG_CASE(56);
    RLG_CASE(60);
    RLG_CASE(64);
    default:
      return static_cast<int>(hipErrorInvalidValue);
  }
#undef RLG_CASE
}

}  // namespace rlg

// ---------------------------------------------------------------------------------
// C ABI (declared in include/rlg_hip.h)
// ---------------------------------------------------------------------------------

extern "C" {

int rlg_gae_envmajor_supported(int horizon) {
  return (horizon % 4 == 0 && horizon >= 4 && horizon <= 64) ? 1 : 0;
}

int rlg_gae_envmajor_num_partials(int num_envs) { return (num_envs + rlg::kWave - 1) / rlg::kWave; }

int rlg_gae_envmajor_fused(const float* rewards, const float* values, const uint8_t* dones,
                           const float* last_values, const uint8_t* last_dones, float* returns,
                           float* advantages, double* moment_partials, int num_envs, int horizon,
                           float gamma, float gamma_tau, void* stream) {
  if (num_envs <= 0) return 0;
  return rlg::dispatch_envmajor<false>(horizon, rewards, values, dones, last_values, last_dones,
                                       returns, advantages, moment_partials, num_envs, gamma,
                                       gamma_tau, static_cast<hipStream_t>(stream));
}

// Profiling hooks (bench.py): HIP events on the launch stream, attached to the GAE dispatch itself
// (hipExtLaunchKernelGGL start/stop events = the dispatch's begin/end timestamps, the quantity
// rocprofv3 --kernel-trace reports), so neither host latency nor marker packets are included.
int rlg_event_create(void** event_out) {
  hipEvent_t ev;
  const hipError_t e = hipEventCreate(&ev);
  *event_out = static_cast<void*>(ev);
  return static_cast<int>(e);
}

int rlg_event_destroy(void* event) { return static_cast<int>(hipEventDestroy(static_cast<hipEvent_t>(event))); }

int rlg_event_elapsed_us(void* start, void* stop, float* us_out) {
  hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(stop));
  if (e != hipSuccess) return static_cast<int>(e);
  float ms = 0.0f;
  e = hipEventElapsedTime(&ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
  *us_out = ms * 1000.0f;
  return static_cast<int>(e);
}

int rlg_gae_envmajor_fused_timed(const float* rewards, const float* values, const uint8_t* dones,
                                 const float* last_values, const uint8_t* last_dones, float* returns,
                                 float* advantages, double* moment_partials, int num_envs,
                                 int horizon, float gamma, float gamma_tau, void* stream,
                                 void* ev_start, void* ev_stop) {
  if (num_envs <= 0) return 0;
  return rlg::dispatch_envmajor<false>(horizon, rewards, values, dones, last_values, last_dones, returns,
                                       advantages, moment_partials, num_envs, gamma, gamma_tau,
                                       static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start),
                                       static_cast<hipEvent_t>(ev_stop));
}

int rlg_gae_envmajor_raw(const float* rewards, const float* values, const uint8_t* dones,
                         const float* last_values, const uint8_t* last_dones, float* gae_out,
                         int num_envs, int horizon, float gamma, float gamma_tau, void* stream) {
  if (num_envs <= 0) return 0;
  return rlg::dispatch_envmajor<true>(horizon, rewards, values, dones, last_values, last_dones,
                                      gae_out, nullptr, nullptr, num_envs, gamma, gamma_tau,
                                      static_cast<hipStream_t>(stream));
}

int rlg_gae_strided(const float* rewards, const float* values, const void* dones,
                    const float* last_values, const void* last_dones, float* advs,
                    float* returns_or_null, int horizon, int num_envs, int value_size,
                    const long long* strides17, int dones_are_float, float gamma,
                    float gamma_tau, void* stream) {
  const long long total = static_cast<long long>(num_envs) * value_size;
  if (total <= 0 || horizon <= 0) return 0;
  rlg::GaeStrides s;
  s.r_t = strides17[0];
  s.r_e = strides17[1];
  s.r_v = strides17[2];
  s.v_t = strides17[3];
  s.v_e = strides17[4];
  s.v_v = strides17[5];
  s.d_t = strides17[6];
  s.d_e = strides17[7];
  s.lv_e = strides17[8];
  s.lv_v = strides17[9];
  s.ld_e = strides17[10];
  s.a_t = strides17[11];
  s.a_e = strides17[12];
  s.a_v = strides17[13];
  s.q_t = strides17[14];
  s.q_e = strides17[15];
  s.q_v = strides17[16];
  const int block = 256;
  const int grid = static_cast<int>((total + block - 1) / block);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dones_are_float) {
    hipLaunchKernelGGL((rlg::gae_strided_kernel<float, 8>), dim3(grid), dim3(block), 0, st,
                       rewards, values, static_cast<const float*>(dones), last_values,
                       static_cast<const float*>(last_dones), advs, returns_or_null, s, horizon,
                       num_envs, value_size, gamma, gamma_tau);
  } else {
    hipLaunchKernelGGL((rlg::gae_strided_kernel<uint8_t, 8>), dim3(grid), dim3(block), 0, st,
                       rewards, values, static_cast<const uint8_t*>(dones), last_values,
                       static_cast<const uint8_t*>(last_dones), advs, returns_or_null, s, horizon,
                       num_envs, value_size, gamma, gamma_tau);
  }
  RLG_RETURN_LAUNCH_STATUS();
}

}  // extern "C"
