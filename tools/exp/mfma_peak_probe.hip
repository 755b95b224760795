// Probe: the sustained issue rate of the two MFMA shapes the kernels use, and the shader clock under that load.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_peak_probe.hip -o tools/exp/_build/mfma_peak_probe
// Each wave issues `iters` x 64 MFMAs on ACC independent accumulators, nothing else; grid = blocks x 256 threads
// (256 blocks = one wave per SIMD, 512 = two).  s_memtime at both ends of wave 0 of block 0 gives the ticks of the
// loop; against the HIP-event time of the launch that is the clock the counter (and the matrix core) runs at.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int ACC>
__global__ __launch_bounds__(256) void probe(float* out, long long* ticks, int iters) {
  f32x4 acc[ACC];
  for (int g = 0; g < ACC; ++g) acc[g] = f32x4{0, 0, 0, 0};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  bf16x8 ah, bh;
  for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)(1.0f + e); bh[e] = (__bf16)0.5f; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64 / ACC; ++s) {
#pragma unroll
      for (int g = 0; g < ACC; ++g) {
        if (SHAPE == 0) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
        else acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[g], 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
  for (int g = 1; g < ACC; ++g) s += acc[g];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int SHAPE, int ACC>
void run(const char* name, float* out, long long* ticks, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((probe<SHAPE, ACC>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
  hipEventRecord(e0);
  const int reps = 5;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((probe<SHAPE, ACC>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double us = ms * 1e3 / reps;
  const double flops = 2.0 * 16 * 16 * (SHAPE == 0 ? 4 : 32) * 64.0 * iters * blocks * 4;
  const double per_simd = 64.0 * iters * ((blocks + 255) / 256);
  printf("%-44s blocks %4d  %9.1f us  %8.1f TFLOP/s  loop %lld ticks = %.0f MHz  %.2f ticks/MFMA/SIMD\n", name, blocks, us,
         flops / us / 1e6, t, t / us, (double)t / per_simd * ((blocks + 255) / 256));
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&ticks, 64);
  for (int rep = 0; rep < 2; ++rep)
    for (int blocks : {256, 512}) {
      run<0, 4>("f32 16x16x4, 4 accumulators", out, ticks, blocks, 4000);
      run<0, 16>("f32 16x16x4, 16 accumulators", out, ticks, blocks, 4000);
      run<0, 1>("f32 16x16x4, 1 accumulator (dependent)", out, ticks, blocks, 1000);
      run<1, 4>("bf16 16x16x32, 4 accumulators", out, ticks, blocks, 8000);
      run<1, 16>("bf16 16x16x32, 16 accumulators", out, ticks, blocks, 8000);
      run<1, 1>("bf16 16x16x32, 1 accumulator (dependent)", out, ticks, blocks, 2000);
    }
  return 0;
}
