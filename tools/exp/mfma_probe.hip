// Probe: what limits a 1-wave-per-SIMD f32 MFMA (16x16x4) stream on MI355X?
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_probe.hip -o tools/exp/_build/mfma_probe
// Variants (template MODE): 0 MFMA only; 1 + 4 ds_read_b128 per 16 MFMAs; 2 + one 16-byte
// weight load per 16 MFMAs in the chain kernel's pattern (16 rows x 64 B per wave load) with a
// one-batch (64 MFMA) prefetch; 3 same loads fully coalesced (1 KiB contiguous per wave load);
// 4 = 2 + 1.   Grid: 256 blocks x 256 threads (one wave per SIMD) unless WAVES says otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ w, int w_floats, int ld, float* out, int iters,
                                             unsigned long long* clk) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __shared__ __attribute__((aligned(16))) float lds[16 * 1024];   // 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, w_floats * 4, 0x00020000);
  f32x4 acc[4];
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0, 0, 0, 0};
  f32x4 b[4];
  for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<f32x4*>(lds + (g * 64 + lane) * 4);
  f32x4 a_cur[4], a_nxt[4];
  for (int u = 0; u < 4; ++u) a_cur[u] = f32x4{1.0f, 0.5f, 0.25f, 0.125f};
  // per-lane byte offset of the weight stream
  unsigned off;
  if (MODE == 3) off = (wave * 64 + lane) * 16;                       // 1 KiB contiguous per wave load
  else off = (((wave * 16) + (lane & 15)) * ld + 4 * (lane >> 4)) * 4;   // 16 rows x 64 B
  const unsigned wrap = w_floats * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned o = off + u * (MODE == 3 ? 4096u : 64u);
        a_nxt[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o % wrap, 0, 0));
      }
      off += (MODE == 3 ? 16384u : 256u);
      if (off >= wrap) off -= wrap;
    }
    PIN();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 1 || MODE == 4) {
        f32x4 bn[4];
        for (int g = 0; g < 4; ++g) bn[g] = *reinterpret_cast<f32x4*>(lds + (((it * 4 + u) & 3) * 1024 + g * 64 + lane) * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][0], b[g][0], acc[g], 0, 0, 0);
        PIN();
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][s], b[g][s], acc[g], 0, 0, 0);
        PIN();
        for (int g = 0; g < 4; ++g) b[g] = bn[g];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][s], b[g][s], acc[g], 0, 0, 0);
      }
    }
    PIN();
    if (MODE >= 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a_cur[u] = a_nxt[u];
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int MODE>
void run(const char* name, const float* w, int w_floats, int ld, float* out, int blocks, int iters) {
  static unsigned long long* clk = nullptr;
  if (!clk) hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, w, w_floats, ld, out, iters, clk);
  hipEventRecord(e0);
  const int reps = 5;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, w, w_floats, ld, out, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double flops = 2.0 * 16 * 16 * 4 * 64.0 * iters * blocks * 4;
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-58s blocks %4d  %9.1f us  %7.1f TFLOP/s  shader clock %.0f MHz (s_memtime / s_memrealtime x 100 MHz)\n", name, blocks, us,
         flops / us / 1e6, 100.0 * (double)h[0] / (double)h[1]);
}

int main() {
  const int ld = 400, rows = 432;                 // ~ the 200x400 layer, 691 KB
  const int w_floats = rows * ld;
  float *w, *out;
  hipMalloc(&w, w_floats * 4); hipMalloc(&out, 4096 * 256 * 4);
  std::vector<float> h(w_floats, 0.01f);
  hipMemcpy(w, h.data(), w_floats * 4, hipMemcpyHostToDevice);
  const int iters = 2000;                          // 64 MFMAs each
  for (int blocks : {256, 512}) {
    run<0>("MFMA only (16 acc... 4 accumulators x 4 steps)", w, w_floats, ld, out, blocks, iters);
    run<1>("+ 4 ds_read_b128 per 16 MFMAs", w, w_floats, ld, out, blocks, iters);
    run<2>("+ weight load 16 rows x 64 B per 16 MFMAs (prefetch 64)", w, w_floats, ld, out, blocks, iters);
    run<3>("+ weight load coalesced 1 KiB per 16 MFMAs (prefetch 64)", w, w_floats, ld, out, blocks, iters);
    run<4>("+ both (chain inner loop)", w, w_floats, ld, out, blocks, iters);
  }
  return 0;
}
