// Probe: does data WRITTEN by a kernel stay in the 256 MB Infinity Cache for a later reader?
//   pattern 0: region written with full coalesced 16-byte stores (time-major rollout fields)
//   pattern 1: region written 4 bytes per 128-byte line, 32 passes (env-major rollout fields)
//   pattern 2: region not written at all before (cold reference: written long ago, then flushed)
// then `noise_mb` of unrelated streaming traffic (read + write), then a reader kernel over the
// 18.9 MB region is timed.   build: hipcc --offload-arch=gfx950 -O3 -w tools/exp/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void write_full(f4* p, long n4, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = f4{v, v, v, v};
}
__global__ void write_partial(float* p, long lines, int slot, float v) {   // one float per 128-B line
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < lines; i += (long)gridDim.x * blockDim.x) p[i * 32 + slot] = v;
}
__global__ void stream(const f4* a, f4* b, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i] + f4{1, 1, 1, 1};
}
__global__ void reader(const f4* p, long n4, float* out) {
  f4 s = {0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) s += p[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.0f;
}
int main() {
  const long region = 18900000 / 16 * 16;           // bytes, the GAE inputs at 65,536 x 32
  const long n4 = region / 16, lines = region / 128;
  float *buf, *na, *nb, *out;
  hipMalloc(&buf, region); hipMalloc(&na, 512L << 20); hipMalloc(&nb, 512L << 20); hipMalloc(&out, 4);
  hipMemset(na, 0, 512L << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int noise_mb : {0, 64, 128, 200, 400}) {
    for (int pattern = 0; pattern < 3; ++pattern) {
      float tot = 0; const int reps = 5;
      for (int r = 0; r < reps; ++r) {
        // flush: stream 1 GB
        stream<<<2048, 256>>>((f4*)na, (f4*)nb, (512L << 20) / 16);
        stream<<<2048, 256>>>((f4*)nb, (f4*)na, (512L << 20) / 16);
        if (pattern == 0) write_full<<<2048, 256>>>((f4*)buf, n4, 1.0f * r);
        if (pattern == 1) for (int s = 0; s < 32; ++s) write_partial<<<1024, 256>>>(buf, lines, s, 1.0f * r);
        if (noise_mb) stream<<<2048, 256>>>((f4*)na, (f4*)nb, ((long)noise_mb << 19) / 16);   // noise_mb/2 read + noise_mb/2 written
        hipEventRecord(e0);
        reader<<<1024, 256>>>((f4*)buf, n4, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
      }
      printf("noise %3d MB  pattern %d (%s)  reader %.2f us  (%.2f TB/s)\n", noise_mb, pattern,
             pattern == 0 ? "full-line stores " : pattern == 1 ? "4 B per line x 32" : "not written      ", tot / reps * 1e3,
             region / (tot / reps * 1e-3) / 1e12);
    }
  }
  return 0;
}
