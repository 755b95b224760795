// Probe: what does a launch cost before it computes anything?  Empty kernels with the grid / block / LDS shapes of the
// fused MLP launches (and a barrier + a little work per workgroup), timed back to back on one stream.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/launch_probe.hip -o tools/exp/_build/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_empty(float* out, int spin) {
  extern __shared__ float lds[];
  if (spin > 0) {
    // spin: ~`spin` x 64 cycles of dependent VALU work per wave (a workgroup that lives for a while)
    float v = threadIdx.x;
    for (int i = 0; i < spin * 16; ++i) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    __syncthreads();
    if (v == 12345.0f) out[blockIdx.x] = lds[(threadIdx.x + 1) & 255];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0f;
}

static void run(const char* name, int grid, int block, int lds, int spin, float* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, 0, out, spin);
  hipEventRecord(e0);
  const int reps = 200;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, 0, out, spin);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s grid %5d block %4d lds %6d B spin %5d : %7.2f us per launch\n", name, grid, block, lds, spin, ms * 1e3 / reps);
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
  run("tiny", 1, 64, 0, 0, out);
  run("one workgroup per CU, no LDS", 256, 256, 0, 0, out);
  run("forward G=4 shape (512 x 256 thr, 152 KB LDS: 1 per CU, 2 rounds)", 512, 256, 152 * 1024, 0, out);
  run("forward G=2 shape (1024 x 256 thr, 76 KB LDS: 2 per CU, 2 rounds)", 1024, 256, 76 * 1024, 0, out);
  run("backward G=4 shape (512 x 256 thr, 80 KB LDS)", 512, 256, 80 * 1024, 0, out);
  run("per-rank shape (256 x 512 thr, 40 KB LDS)", 256, 512, 40 * 1024, 0, out);
  // with a fixed amount of work per workgroup: time = rounds x work + overheads
  for (int spin : {100, 400}) {
    run("forward G=4 shape + work", 512, 256, 152 * 1024, spin, out);
    run("forward G=2 shape + work", 1024, 256, 76 * 1024, spin, out);
    run("256 workgroups + work (one round)", 256, 256, 152 * 1024, spin, out);
  }
  return 0;
}
