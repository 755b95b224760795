// Development experiment (not part of the product library): A/B variants of the env-major
// GAE kernel + a same-footprint copy kernel, timed with hipEvents over back-to-back launches
// on rotating buffer sets (cold = 16 sets > 256 MB MALL, warm = 1 set).
#include "../../rl_games_amd/csrc/gae.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

using namespace rlg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// EXPERIMENT (measured slower than the wave-private product kernel: 12.4 vs 11.7 us cold, 9.2 vs 8.3 us
// warm, outputs bit-identical - profiles/r1_gae_variants.txt).  Kept here for the record only.
namespace rlg {
// ---------------------------------------------------------------------------------
// Cooperative env-major kernel (horizons that are multiples of 16).
//
// The wave-private kernel above runs exactly one wave per SIMD at 65,536 envs (1,024 tiles on
// 1,024 SIMDs), so every ALU/LDS phase of a wave - transposes, the 32-step chain, the fp64
// moments - is exposed time in which that SIMD issues no memory traffic.  Here one 256-thread
// block owns a 64-env tile and each of its 4 waves owns a QUARTER OF THE HORIZON of all 64 envs:
//   1. all 256 threads load the tile with fully contiguous 16-byte accesses into a shared,
//      padded LDS tile (one barrier);
//   2. wave q reads its H/4 steps of every env (lane = env), pre-computes the terms that do not
//      depend on the carry (delta_t and gamma*tau*nnt_t), then the four waves run their chain
//      segments back to back, handing A_t over through LDS (wave 3 first; 4 short barriers).
//      The chain itself is evaluated in exactly the sequential order of gae_kernel.py:63-80, so
//      the results are bit-identical to the wave-private kernel;
//   3. all threads store returns / (returns - values) coalesced from LDS and accumulate the fp64
//      moments on the way out.
// 4 blocks are resident per CU (16 waves), so one block's ALU phases hide behind the others'
// loads and stores.
// ---------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void gae_envmajor_coop_kernel(
    const float* __restrict__ rewards,      // [N, H]
    const float* __restrict__ values,       // [N, H]
    const uint8_t* __restrict__ dones,      // [N, H]
    const float* __restrict__ last_values,  // [N]
    const uint8_t* __restrict__ last_dones, // [N]
    float* __restrict__ out_ret,            // [N, H]
    float* __restrict__ out_adv,            // [N, H]
    double* __restrict__ partials,          // [num_tiles, 6] or nullptr
    int N, float gamma, float gamma_tau) {
  static_assert(H % 16 == 0 && H >= 16 && H <= 64, "cooperative kernel: horizon multiple of 16");
  constexpr int Q = H / 4;            // steps per wave
  constexpr int RS = H + 4;           // padded row stride (floats): conflict-free b128 row reads
  constexpr int C = H / 4;            // 16-byte chunks per row
  constexpr int DSW = H / 4 + 1;      // dones row stride in dwords (odd: conflict-free b32 reads)
  constexpr int kPer = (kWave * C) / 256;   // chunks per thread per array (= H/16)
  __shared__ __attribute__((aligned(16))) float tr[kWave * RS];
  __shared__ __attribute__((aligned(16))) float tv[kWave * RS];
  __shared__ uint32_t td[kWave * DSW];
  __shared__ float carry[kWave];
  __shared__ double red[6 * 4];

  const int tid = threadIdx.x;
  const int q = tid >> 6;
  const int lane = tid & 63;
  const int env0 = blockIdx.x * kWave;
  const int rows = min(kWave, N - env0);
  const int live_chunks = rows * C;
  const long long base4 = static_cast<long long>(env0) * C;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(rewards) + base4;
  const f32x4* v4 = reinterpret_cast<const f32x4*>(values) + base4;

  // ---- 1. coalesced loads, all issued before the first LDS write ----
  f32x4 rb[kPer], vb[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = min(tid + 256 * k, live_chunks - 1);   // ragged last tile: clamp, never predicate
    rb[k] = r4[c];
    vb[k] = v4[c];
  }
  constexpr int kDoneChunks = (kWave * H) / 16;          // <= 256
  u32x4 db = {0u, 0u, 0u, 0u};
  if (tid < kDoneChunks) {
    const int c = min(tid, (rows * H) / 16 - 1);
    db = reinterpret_cast<const u32x4*>(dones + static_cast<long long>(env0) * H)[c];
  }
  const int env_c = env0 + min(lane, rows - 1);
  float lv = 0.0f;
  uint32_t ld = 0u;
  if (q == 3) {
    lv = last_values[env_c];
    ld = last_dones[env_c];
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = tid + 256 * k;
    const int row = c / C;
    const int col = (c - row * C) * 4;
    *reinterpret_cast<f32x4*>(tr + row * RS + col) = rb[k];
    *reinterpret_cast<f32x4*>(tv + row * RS + col) = vb[k];
  }
  if (tid < kDoneChunks) {
    const int byte = tid * 16;
    const int row = byte / H;
    const int off = (byte - row * H) >> 2;
    uint32_t* dst = td + row * DSW + off;
    dst[0] = db[0];
    dst[1] = db[1];
    dst[2] = db[2];
    dst[3] = db[3];
  }
  __syncthreads();

  // ---- 2. this wave's quarter of every env row ----
  float r[Q], v[Q], delta[Q], coef[Q];
  uint32_t dw[Q / 4 + 1];
  {
    const float* rrow = tr + lane * RS + q * Q;
    const float* vrow = tv + lane * RS + q * Q;
#pragma unroll
    for (int j = 0; j < Q / 4; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(rrow + 4 * j);
      const f32x4 b = *reinterpret_cast<const f32x4*>(vrow + 4 * j);
      r[4 * j + 0] = a[0]; r[4 * j + 1] = a[1]; r[4 * j + 2] = a[2]; r[4 * j + 3] = a[3];
      v[4 * j + 0] = b[0]; v[4 * j + 1] = b[1]; v[4 * j + 2] = b[2]; v[4 * j + 3] = b[3];
    }
#pragma unroll
    for (int j = 0; j < Q / 4 + 1; ++j) dw[j] = td[lane * DSW + q * (Q / 4) + j];   // last: pad (q == 3)
    float nv, nnt;
    if (q == 3) {
      nv = lv;
      nnt = 1.0f - static_cast<float>(ld);
    } else {
      nv = vrow[Q];
      nnt = 1.0f - static_cast<float>(dw[Q / 4] & 0xffu);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int k = Q - 1 - i;
      const float vt = v[k];
      delta[k] = (r[k] + (gamma * nv) * nnt) - vt;     // gae_kernel.py:78, op for op
      coef[k] = gamma_tau * nnt;                        // :79
      nv = vt;
      nnt = 1.0f - static_cast<float>((dw[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q == 3 - s) {            // wave-uniform
      float A = (s == 0) ? 0.0f : carry[lane];
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int k = Q - 1 - i;
        A = delta[k] + coef[k] * A;
        r[k] = A + v[k];                                // returns, a2c_common.py:1060
      }
      carry[lane] = A;
      float* rrow = tr + lane * RS + q * Q;             // only this wave ever touches this quarter
#pragma unroll
      for (int j = 0; j < Q / 4; ++j) {
        const f32x4 o = {r[4 * j + 0], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
        *reinterpret_cast<f32x4*>(rrow + 4 * j) = o;
      }
    }
    __syncthreads();
  }

  // ---- 3. coalesced stores + moments ----
  f32x4* o_ret = reinterpret_cast<f32x4*>(out_ret) + base4;
  f32x4* o_adv = reinterpret_cast<f32x4*>(out_adv) + base4;
  double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = tid + 256 * k;
    const int row = c / C;
    const int col = (c - row * C) * 4;
    const f32x4 ret = *reinterpret_cast<const f32x4*>(tr + row * RS + col);
    const f32x4 val = *reinterpret_cast<const f32x4*>(tv + row * RS + col);
    const f32x4 adv = ret - val;                        // a2c_common.py:1598
    if (c < live_chunks) {
      o_ret[c] = ret;
      o_adv[c] = adv;
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
  if (partials) {
    block_sum<6, 256>(m, red);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) partials[static_cast<long long>(blockIdx.x) * 6 + k] = m[k];
    }
  }
}

}  // namespace rlg

// same bytes as the fused kernel: read r,v (f32) + d (u8), write ret, adv.
__global__ __launch_bounds__(256) void copy_like_kernel(const f32x4* __restrict__ r, const f32x4* __restrict__ v,
                                                        const u32x4* __restrict__ d, f32x4* __restrict__ o0,
                                                        f32x4* __restrict__ o1, int n4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int stride = gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 a = r[i], b = v[i];
    float s = 0.f;
    if ((i & 3) == 0) { u32x4 q = d[i >> 2]; s = (float)(q[0] & 1); }
    o0[i] = a + b + s;
    o1[i] = a - b;
  }
}

// Variant: MOM 0 none / 1 f64 per element / 2 f32 shifted sums per lane -> f64 merge
template <int H, int MOM, int WAVES, int PW = 0>
__global__ __launch_bounds__(64 * WAVES) void gae_var_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values, const uint8_t* __restrict__ dones,
    const float* __restrict__ last_values, const uint8_t* __restrict__ last_dones, float* __restrict__ out0,
    float* __restrict__ out1, double* __restrict__ partials, int N, float gamma, float gamma_tau) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES * 2 * TileGeom<H>::kTileFloats];
  const int w = threadIdx.x >> 6;
  float* tile_r = lds + w * 2 * TileGeom<H>::kTileFloats;
  float* tile_v = tile_r + TileGeom<H>::kTileFloats;
  const int tile = blockIdx.x * WAVES + w;
  const int env0 = tile * kWave;
  if (env0 >= N) return;
  const int rows = min(kWave, N - env0);
  const int lane = lane_id();
  const int env = env0 + lane;
  const bool live = lane < rows;
  const long long base = (long long)env0 * H;
  f32x4 rbuf[H / 4], vbuf[H / 4];
  float r[H], v[H];
  if (PW == 5) {   // EXPERIMENT: rewards stored time-major [H][N]: lane = env, one coalesced dword load per t
#pragma unroll
    for (int t = 0; t < H; ++t) r[t] = rewards[(long long)t * N + (live ? env : env0)];
  } else {
    tile_load_issue<H>(rewards + base, rows, rbuf);
  }
  tile_load_issue<H>(values + base, rows, vbuf);
  uint32_t dw[H / 4];
  {
    const uint8_t* drow = dones + (long long)(live ? env : env0) * H;
#pragma unroll
    for (int j = 0; j < H / 16; ++j) {
      const u32x4 q = *reinterpret_cast<const u32x4*>(drow + 16 * j);
      dw[4 * j + 0] = q[0]; dw[4 * j + 1] = q[1]; dw[4 * j + 2] = q[2]; dw[4 * j + 3] = q[3];
    }
  }
  float nv = last_values[live ? env : env0];
  float nnt = 1.0f - (float)last_dones[live ? env : env0];
  if (PW != 5) tile_regs_to_lds<H>(rbuf, tile_r);
  tile_regs_to_lds<H>(vbuf, tile_v);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (PW != 5) row_from_lds<H>(tile_r, r);
  row_from_lds<H>(tile_v, v);
  double m[6] = {0, 0, 0, 0, 0, 0};
  float f[6] = {0, 0, 0, 0, 0, 0};
  float a_[MOM >= 3 ? H : 1];
  const float kv = v[H - 1];   // per-lane pivots for the shifted sums
  float A = 0.0f;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int t = H - 1 - i;
    const float vt = v[t];
    const float delta = (r[t] + (gamma * nv) * nnt) - vt;
    A = delta + (gamma_tau * nnt) * A;
    const float ret = A + vt;
    const float adv = ret - vt;
    r[t] = ret;
    if (MOM >= 3) { a_[t] = adv; } else { v[t] = adv; }
    if (MOM == 1) {
      const double da = adv, dv = vt, dr = ret;
      m[0] += da; m[1] = fma(da, da, m[1]); m[2] += dv; m[3] = fma(dv, dv, m[3]); m[4] += dr; m[5] = fma(dr, dr, m[5]);
    } else if (MOM == 2) {
      const float sv = vt - kv, sr = ret - kv;
      f[0] += adv; f[1] = fmaf(adv, adv, f[1]); f[2] += sv; f[3] = fmaf(sv, sv, f[3]); f[4] += sr; f[5] = fmaf(sr, sr, f[5]);
    }
    nv = vt;
    const uint32_t dbyte = (dw[t >> 2] >> (8 * (t & 3))) & 0xffu;
    nnt = 1.0f - (float)dbyte;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  row_to_lds<H>(tile_r, r);
  if constexpr (MOM >= 3) row_to_lds<H>(tile_v, a_); else row_to_lds<H>(tile_v, v);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  tile_lds_to_global<H>(tile_r, out0 + base, rows);
  tile_lds_to_global<H>(tile_v, out1 + base, rows);
  if constexpr (MOM == 3) {
#pragma unroll
    for (int t = 0; t < H; ++t) {
      const double da = a_[t], dv = v[t], dr = r[t];
      m[0] += da; m[1] = fma(da, da, m[1]); m[2] += dv; m[3] = fma(dv, dv, m[3]); m[4] += dr; m[5] = fma(dr, dr, m[5]);
    }
  }
  if constexpr (MOM == 4) {
#pragma unroll
    for (int t = 0; t < H; ++t) {
      const float adv = a_[t], sv = v[t] - kv, sr = r[t] - kv;
      f[0] += adv; f[1] = fmaf(adv, adv, f[1]); f[2] += sv; f[3] = fmaf(sv, sv, f[3]); f[4] += sr; f[5] = fmaf(sr, sr, f[5]);
    }
  }
  if (MOM == 2 || MOM == 4) {
    // un-shift per lane in fp64: sum x = s1 + n*k ; sum x^2 = s2 + 2k s1 + n k^2
    const double k = kv, n = H;
    m[0] = f[0]; m[1] = f[1];
    m[2] = (double)f[2] + n * k; m[3] = (double)f[3] + 2.0 * k * (double)f[2] + n * k * k;
    m[4] = (double)f[4] + n * k; m[5] = (double)f[5] + 2.0 * k * (double)f[4] + n * k * k;
  }
  if (MOM != 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = wave_sum(live ? m[k] : 0.0);
    if (PW == 2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) asm volatile("" :: "v"(m[k]));
    } else if (lane == 0) {
      const long long stride = (PW == 1) ? 16 : 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) partials[(long long)tile * stride + k] = m[k];
    }
  }
  if (MOM == 0 && PW == 3 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) partials[(long long)tile * 6 + k] = (double)A;
  }
}

struct Set { float *r, *v, *lv, *o0, *o1; uint8_t *d, *ld; double* part; };

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const bool pmc = argc > 1;   // short mode for rocprofv3 --pmc runs: few launches, cold sets only
  const int N = 65536, H = 32; const int iters = pmc ? 6 : 200;
  const size_t nf = (size_t)N * H;
  for (int nsets : {1, 16}) {
    if (pmc && nsets == 1) continue;
    std::vector<Set> sets(nsets);
    for (auto& s : sets) {
      CK(hipMalloc(&s.r, nf * 4)); CK(hipMalloc(&s.v, nf * 4)); CK(hipMalloc(&s.o0, nf * 4)); CK(hipMalloc(&s.o1, nf * 4));
      CK(hipMalloc(&s.d, nf)); CK(hipMalloc(&s.lv, N * 4)); CK(hipMalloc(&s.ld, N)); CK(hipMalloc(&s.part, 1024 * 16 * 8));
      std::vector<float> h(nf); for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
      CK(hipMemcpy(s.r, h.data(), nf * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s.v, h.data(), nf * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(s.lv, h.data(), N * 4, hipMemcpyHostToDevice));
      std::vector<uint8_t> hd(nf); for (auto& x : hd) x = (rand() % 20) == 0;
      CK(hipMemcpy(s.d, hd.data(), nf, hipMemcpyHostToDevice)); CK(hipMemcpy(s.ld, hd.data(), N, hipMemcpyHostToDevice));
    }
    const double mb = nf * 17.0 / 1e6;
    auto report = [&](const char* name, float us) {
      printf("sets=%2d %-34s %7.2f us  %7.1f GB/s (%.1f%% of 8TB/s)\n", nsets, name, us, mb / us * 1e3, mb / us * 1e3 / 80.0);
    };
    report("copy_like 2048x256", time_it([&](int i) { auto& s = sets[i % nsets];
      hipLaunchKernelGGL(copy_like_kernel, dim3(2048), dim3(256), 0, 0, (f32x4*)s.r, (f32x4*)s.v, (u32x4*)s.d, (f32x4*)s.o0, (f32x4*)s.o1, (int)(nf / 4)); }, iters));
    report("copy_like 8192x256", time_it([&](int i) { auto& s = sets[i % nsets];
      hipLaunchKernelGGL(copy_like_kernel, dim3(8192), dim3(256), 0, 0, (f32x4*)s.r, (f32x4*)s.v, (u32x4*)s.d, (f32x4*)s.o0, (f32x4*)s.o1, (int)(nf / 4)); }, iters));
    report("product fused (f64 mom, 64thr)", time_it([&](int i) { auto& s = sets[i % nsets];
      rlg_gae_envmajor_fused(s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, H, 0.99f, 0.9405f, nullptr); }, iters));
    report("coop 4 waves x H/4 (256thr/tile)", time_it([&](int i) { auto& s = sets[i % nsets];
      hipLaunchKernelGGL((gae_envmajor_coop_kernel<32>), dim3(N / 64), dim3(256), 0, 0, s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, 0.99f, 0.9405f); }, iters));
    if (nsets == 1 && !pmc) {   // bit-equality of the two kernels' outputs
      auto& s = sets[0];
      std::vector<float> a0(nf), a1(nf), b0(nf), b1(nf); std::vector<double> pa(N / 64 * 6), pb(N / 64 * 6);
      rlg_gae_envmajor_fused(s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, H, 0.99f, 0.9405f, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a0.data(), s.o0, nf * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), s.o1, nf * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(pa.data(), s.part, pa.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemset(s.o0, 0, nf * 4)); CK(hipMemset(s.o1, 0, nf * 4));
      hipLaunchKernelGGL((gae_envmajor_coop_kernel<32>), dim3(N / 64), dim3(256), 0, 0, s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, 0.99f, 0.9405f);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(b0.data(), s.o0, nf * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b1.data(), s.o1, nf * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(pb.data(), s.part, pb.size() * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < nf; ++i) bad += (memcmp(&a0[i], &b0[i], 4) != 0) + (memcmp(&a1[i], &b1[i], 4) != 0);
      double pd = 0; for (size_t i = 0; i < pa.size(); ++i) pd = fmax(pd, fabs(pa[i] - pb[i]) / (fabs(pa[i]) + 1e-30));
      printf("coop vs wave-private: %zu differing output words, max rel partial diff %.3e\n", bad, pd);
    }
#define RUNVARP(MOM, WAVES, PW, label) report(label, time_it([&](int i) { auto& s = sets[i % nsets]; \
      hipLaunchKernelGGL((gae_var_kernel<32, MOM, WAVES, PW>), dim3((N / 64 + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, 0, s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, 0.99f, 0.9405f); }, iters));
#define RUNVAR(MOM, WAVES, label) report(label, time_it([&](int i) { auto& s = sets[i % nsets]; \
      hipLaunchKernelGGL((gae_var_kernel<32, MOM, WAVES>), dim3((N / 64 + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, 0, s.r, s.v, s.d, s.lv, s.ld, s.o0, s.o1, s.part, N, 0.99f, 0.9405f); }, iters));
    if (nsets == 16 && !pmc) {
      // mixed residency (round 2): which side costs the cold penalty - the 18.9 MB of inputs or the
      // 16.8 MB of outputs?  inputs rotate over 16 sets (cold) while the outputs stay in one set (warm), and v.v.
      report("product: inputs COLD, outputs WARM", time_it([&](int i) { auto& a = sets[i % nsets]; auto& b = sets[0];
        rlg_gae_envmajor_fused(a.r, a.v, a.d, a.lv, a.ld, b.o0, b.o1, b.part, N, H, 0.99f, 0.9405f, nullptr); }, iters));
      report("product: inputs WARM, outputs COLD", time_it([&](int i) { auto& a = sets[0]; auto& b = sets[i % nsets];
        rlg_gae_envmajor_fused(a.r, a.v, a.d, a.lv, a.ld, b.o0, b.o1, b.part, N, H, 0.99f, 0.9405f, nullptr); }, iters));
      report("nomom 4w: inputs COLD, outputs WARM", time_it([&](int i) { auto& a = sets[i % nsets]; auto& b = sets[0];
        hipLaunchKernelGGL((gae_var_kernel<32, 0, 4>), dim3(N / 64 / 4), dim3(256), 0, 0, a.r, a.v, a.d, a.lv, a.ld, b.o0, b.o1, b.part, N, 0.99f, 0.9405f); }, iters));
      report("nomom 4w: inputs WARM, outputs COLD", time_it([&](int i) { auto& a = sets[0]; auto& b = sets[i % nsets];
        hipLaunchKernelGGL((gae_var_kernel<32, 0, 4>), dim3(N / 64 / 4), dim3(256), 0, 0, a.r, a.v, a.d, a.lv, a.ld, b.o0, b.o1, b.part, N, 0.99f, 0.9405f); }, iters));
    }
    RUNVAR(0, 1, "var nomom 1wave/blk");
    RUNVAR(1, 1, "var f64mom 1wave/blk");
    RUNVAR(2, 1, "var f32mom 1wave/blk");
    RUNVAR(0, 4, "var nomom 4wave/blk");
    RUNVAR(2, 4, "var f32mom 4wave/blk");
    RUNVARP(4, 4, 1, "f32mom-after 4w partial stride128B");
    RUNVARP(4, 4, 2, "f32mom-after 4w NO partial write");
    RUNVARP(0, 4, 3, "nomom 4w + const partial write");
    RUNVARP(0, 4, 5, "nomom 4w, rewards TIME-major (no LDS for r)");
    RUNVARP(3, 4, 5, "f64mom-after 4w, rewards TIME-major");
    RUNVAR(3, 1, "var f64mom-after-store 1w");
    RUNVAR(4, 1, "var f32mom-after-store 1w");
    RUNVAR(3, 4, "var f64mom-after-store 4w");
    RUNVAR(4, 4, "var f32mom-after-store 4w");
    for (auto& s : sets) { hipFree(s.r); hipFree(s.v); hipFree(s.o0); hipFree(s.o1); hipFree(s.d); hipFree(s.lv); hipFree(s.ld); hipFree(s.part); }
  }
  return 0;
}
