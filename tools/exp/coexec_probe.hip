// Probe: do VALU instructions overlap with MFMAs on gfx950 - inside one wave (independent VALU work in the "shadow"
// of a 32-cycle v_mfma_f32_16x16x4_f32), and between two waves of one SIMD (one issuing MFMAs, one VALU)?
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/coexec_probe.hip -o tools/exp/_build/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// same wave: 16 MFMAs (4 accumulators) with K independent v_fma_f32 behind every MFMA
template <int K>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters) {
  f32x4 acc[4];
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.001f + k;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  float t = 0;
  for (int k = 0; k < 8; ++k) t += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
}

// two waves per SIMD (512 threads): MODE bit 0 = waves 0-3 issue MFMAs, bit 1 = waves 4-7 issue VALU
template <int MODE>
__global__ __launch_bounds__(512) void two_waves(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.001f + k;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  if (wave < 4) {
    if (MODE & 1) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s & 3], 0, 0, 0);
      }
    }
  } else if (MODE & 2) {
    for (int it = 0; it < iters; ++it) {
      // 16 x 8 = 128 independent-ish VALU instructions = 512 issue cycles, the time of 16 MFMAs
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      }
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  float t = 0;
  for (int k = 0; k < 8; ++k) t += v[k];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
}

template <class F>
static double timeit(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 5;
}

int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  const int iters = 4000;
  const double mfma_only = timeit([&] { hipLaunchKernelGGL(same_wave<0>, dim3(256), dim3(256), 0, 0, out, iters); });
  printf("one wave per SIMD, 16 MFMAs per iteration: %.1f us = %.1f cycles per MFMA at 2.4 GHz\n", mfma_only, mfma_only * 2400.0 / (16.0 * iters));
#define ROW(K) { const double t = timeit([&] { hipLaunchKernelGGL(same_wave<K>, dim3(256), dim3(256), 0, 0, out, iters); }); \
    printf("  + %d independent v_fma_f32 behind every MFMA: %8.1f us  (+%.1f cycles per MFMA = %.2f cycles per VALU instruction)\n", K, t, \
           (t - mfma_only) * 2400.0 / (16.0 * iters), (t - mfma_only) * 2400.0 / (16.0 * iters) / K); }
  ROW(1) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12)
  const double a = timeit([&] { hipLaunchKernelGGL(two_waves<1>, dim3(256), dim3(512), 0, 0, out, iters); });
  const double b = timeit([&] { hipLaunchKernelGGL(two_waves<2>, dim3(256), dim3(512), 0, 0, out, iters); });
  const double c = timeit([&] { hipLaunchKernelGGL(two_waves<3>, dim3(256), dim3(512), 0, 0, out, iters); });
  printf("two waves per SIMD: MFMA wave alone %.1f us, VALU wave alone %.1f us, both together %.1f us  (sum %.1f, max %.1f)\n", a, b, c, a + b,
         a > b ? a : b);
  return 0;
}
