// Probe: how many bytes per second can the CUs pull out of the L2 when all of them stream the SAME small array
// (the fused MLP's weights: 580 KB, L2-resident in every XCD)?
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/l2_stream_probe.hip -o tools/exp/_build/l2_stream_probe
// Each wave issues DEPTH 16-byte buffer loads per iteration (1 KiB per wave and load, contiguous), sums them, repeats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ w, unsigned bytes, float* out, int iters) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off = ((blockIdx.x * 4 + wave) * 7919u % (bytes / 1024u)) * 1024u + lane * 16u;   // waves start spread out
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      v[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
      off += 1024u;
      if (off >= bytes) off -= bytes;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int DEPTH>
void run(const float* w, unsigned bytes, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(probe<DEPTH>, dim3(blocks), dim3(256), 0, 0, w, bytes, out, iters);
  hipEventRecord(e0);
  const int reps = 5;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(probe<DEPTH>, dim3(blocks), dim3(256), 0, 0, w, bytes, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double total = 1024.0 * DEPTH * iters * blocks * 4;
  printf("array %7.0f KB  waves/SIMD %d  loads in flight per wave %2d (%3d KB per CU)  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU at 2.4 GHz\n",
         bytes / 1024.0, blocks / 256, DEPTH, DEPTH * 4 * (blocks / 256), us, total / us / 1e6, total / us / 1e6 * 1e12 / 2.4e9 / 256);
}

int main() {
  float *w, *out;
  const unsigned maxb = 64u << 20;
  hipMalloc(&w, maxb); hipMalloc(&out, 4096 * 256 * 4);
  std::vector<float> h(maxb / 4, 0.01f);
  hipMemcpy(w, h.data(), maxb, hipMemcpyHostToDevice);
  for (unsigned kb : {16u, 580u, 2048u, 32768u})
    for (int blocks : {256, 512, 1024, 2048}) {
      const int iters = 20000 / (blocks / 256);
      run<1>(w, kb * 1024u, out, blocks, iters * 4);
      run<4>(w, kb * 1024u, out, blocks, iters);
      run<8>(w, kb * 1024u, out, blocks, iters / 2);
      run<16>(w, kb * 1024u, out, blocks, iters / 4);
    }
  return 0;
}
