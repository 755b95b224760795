// Probe (round 6): v_mfma_f32_16x16x32_f16 beside v_mfma_f32_16x16x32_bf16 on gfx950 - rate (dependent chain and a
// rotation over 8 accumulators, one wave per SIMD), subnormal inputs (kept or flushed?), and what the f32 -> f16
// conversion does with values below the fp16 normal range.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/f16_mfma_probe.hip -o tools/exp/_build/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool kF16, bool kChain>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cycles, int iters) {
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  u32x4 b = {0x3c003c00u, 0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u};
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kChain ? 0 : i;
      if constexpr (kF16)
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[k], 0, 0, 0);
      else
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[k], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

// D = A B with A[i][k] = a_val, B[k][j] = b_val for all (i, j, k): D = 32 a b.
__global__ void subnormal_kernel(float a_val, float b_val, float* out) {
  const _Float16 ah = static_cast<_Float16>(a_val), bh = static_cast<_Float16>(b_val);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ah; b[i] = bh; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = acc[0];
    out[1] = static_cast<float>(ah);       // what the conversion kept
    out[2] = static_cast<float>(bh);
  }
}

// the packed conversion the split would use, on a pair
__global__ void cvt_kernel(const float* in, float* out, int n) {
  const int i = threadIdx.x;
  if (2 * i + 1 < n) {
    f16x2 h = __builtin_convertvector((f32x2{in[2 * i], in[2 * i + 1]}), f16x2);
    out[2 * i] = static_cast<float>(h[0]);
    out[2 * i + 1] = static_cast<float>(h[1]);
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool kF16, bool kChain>
static int rate(const char* name, int iters) {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 256 * sizeof(float)));
  CK(hipMalloc(&cyc, sizeof(long long)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<kF16, kChain><<<256, 256>>>(out, cyc, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate_kernel<kF16, kChain><<<256, 256>>>(out, cyc, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  const double n = 8.0 * iters;
  // s_memtime-style counter runs at 100 MHz on this part; the wall clock gives the rate
  printf("%-44s %8.3f ms   %7.2f ns per MFMA and wave   %.1f TFLOP/s (256 CUs x 4 waves)\n", name, ms, ms * 1e6 / n,
         256.0 * 4 * n * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  if (rate<false, true>("bf16 16x16x32, dependent chain", iters)) return 1;
  if (rate<true, true>("f16  16x16x32, dependent chain", iters)) return 1;
  if (rate<false, false>("bf16 16x16x32, rotation over 8 accumulators", iters)) return 1;
  if (rate<true, false>("f16  16x16x32, rotation over 8 accumulators", iters)) return 1;
  float* out; CK(hipMalloc(&out, 64 * sizeof(float)));
  const float tests[][2] = {{1.0f, 1.0f}, {ldexpf(1.0f, -14), 1.0f}, {ldexpf(1.0f, -15), 1.0f}, {ldexpf(1.0f, -20), 1024.0f},
                            {ldexpf(1.0f, -24), 1024.0f}, {ldexpf(3.0f, -25), 4096.0f}, {ldexpf(1.0f, -26), 1024.0f},
                            {ldexpf(1.0f, -20), ldexpf(1.0f, -20)}};
  for (auto& t : tests) {
    subnormal_kernel<<<1, 64>>>(t[0], t[1], out);
    CK(hipDeviceSynchronize());
    float h[3]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("a = %.6e (as f16 %.6e)  b = %.6e (as f16 %.6e):  32 a b = %.6e   MFMA gives %.6e\n", t[0], h[1], t[1], h[2],
           32.0 * h[1] * h[2], h[0]);
  }
  const int n = 16;
  float hin[n] = {6.2e-5f, 6.0e-5f, 3.0e-5f, 1.0e-6f, 5.96e-8f, 3.0e-8f, 2.9e-8f, -1.0e-6f, 65504.0f, 65519.9f, 65520.0f, 1e6f,
                  0.1f, 0.3333333f, -2.71828f, 1000.123f};
  float *din, *dout; CK(hipMalloc(&din, sizeof(hin))); CK(hipMalloc(&dout, sizeof(hin)));
  CK(hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice));
  cvt_kernel<<<1, 64>>>(din, dout, n);
  float hout[n]; CK(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) printf("cvt %.8e -> %.8e\n", hin[i], hout[i]);
  return 0;
}
