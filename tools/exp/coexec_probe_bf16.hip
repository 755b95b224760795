// Probe (round 6): do VALU instructions overlap with v_mfma_f32_16x16x32_bf16 on gfx950?
// The round-3 probe (coexec_probe.hip) only issued v_mfma_f32_16x16x4_f32, which runs at the fp32 VECTOR rate; all three
// dominant kernels of the epoch issue the bf16 MFMA.  Measured here, per filler kind (the instructions of the plane
// split, csrc/split_bf16.hpp):
//   (a) ONE wave per SIMD: K independent fillers behind every MFMA (8 independent accumulators, no dependent chain),
//       cycles per MFMA from s_memtime and from the wall clock;
//   (b) TWO waves per SIMD: an MFMA-only wave next to a VALU-only wave, alone and together (sum or max?);
//   (c) the weight-gradient loop's shape: 96 MFMAs + the 288 VALU instructions of two dw_split8x2 per batch, once
//       split-then-multiply (what mlp_dw.hip did up to round 5) and once interleaved 1 MFMA : 3 VALU.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/coexec_probe_bf16.hip -o tools/exp/_build/coexec_probe_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum Filler { kFma = 0, kCvt = 1, kShiftAnd = 2, kSub = 3, kPkAdd = 4, kPkAddNeg = 5, kMov = 6, kNumFillers = 7 };
static const char* kFillerName[kNumFillers] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32 / v_and_b32 (alternating)", "v_sub_f32",
                                               "v_pk_add_f32", "v_pk_add_f32 neg_lo neg_hi", "v_mov_b32"};

struct FillState {
  float v[8];
  f32x2 p[8];
  unsigned u[8];
};

template <int F>
__device__ __forceinline__ void filler(FillState& s, int k) {
  const int i = k & 7;
  if constexpr (F == kFma) {
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.v[i]) : "v"(s.v[(i + 1) & 7]), "v"(s.v[(i + 2) & 7]));
  } else if constexpr (F == kCvt) {
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[i]) : "v"(s.v[i]), "v"(s.v[(i + 1) & 7]));
  } else if constexpr (F == kShiftAnd) {
    if (k & 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.u[i]) : "v"(s.u[(i + 1) & 7]));
    else asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.u[i]) : "v"(s.u[(i + 1) & 7]));
  } else if constexpr (F == kSub) {
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[i]) : "v"(s.v[(i + 1) & 7]));
  } else if constexpr (F == kPkAdd) {
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s.p[i]) : "v"(s.p[(i + 1) & 7]));
  } else if constexpr (F == kPkAddNeg) {
    asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(s.p[i]) : "v"(s.p[(i + 1) & 7]));
  } else {
    asm volatile("v_mov_b32 %0, %1" : "=v"(s.u[i]) : "v"(s.u[(i + 1) & 7]));
  }
}

__device__ __forceinline__ void fill_init(FillState& s) {
  for (int k = 0; k < 8; ++k) {
    s.v[k] = threadIdx.x * 1e-3f + k;
    s.p[k] = f32x2{threadIdx.x * 1e-3f + k, 1.0f};
    s.u[k] = threadIdx.x * 7 + k;
  }
}
__device__ __forceinline__ float fill_sum(const FillState& s) {
  float t = 0;
  for (int k = 0; k < 8; ++k) t += s.v[k] + s.p[k][0] + s.p[k][1] + __uint_as_float(s.u[k] & 0x3fffffffu);
  return t;
}

__device__ __forceinline__ bf16x8 operand(int seed) {
  u32x4 w;
  for (int q = 0; q < 4; ++q) w[q] = 0x3f803f80u + ((threadIdx.x * 3 + seed + q) & 0x7) * 0x00010001u;
  return __builtin_bit_cast(bf16x8, w);
}

// (a) one wave per SIMD: 8 accumulators, K fillers behind every MFMA
template <int F, int K>
__global__ __launch_bounds__(256) void same_wave(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int g = 0; g < 8; ++g) acc[g] = f32x4{0, 0, 0, 0};
  FillState s;
  fill_init(s);
  const bf16x8 a = operand(1), b = operand(5);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) filler<F>(s, g * K + k);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 r = acc[0];
  for (int g = 1; g < 8; ++g) r += acc[g];
  out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + fill_sum(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// VALU only, one wave per SIMD: the issue cost of the filler by itself (8 * K per iteration)
template <int F, int K>
__global__ __launch_bounds__(256) void valu_only(float* out, long long* cyc, int iters) {
  FillState s;
  fill_init(s);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8 * K; ++k) filler<F>(s, k);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = fill_sum(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// (b) two waves per SIMD (512 threads): MODE bit 0 = waves 0-3 issue 8 MFMAs per iteration, bit 1 = waves 4-7 issue
// 32 fillers per iteration
template <int F, int MODE>
__global__ __launch_bounds__(512) void two_waves(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[8];
  for (int g = 0; g < 8; ++g) acc[g] = f32x4{0, 0, 0, 0};
  FillState s;
  fill_init(s);
  const bf16x8 a = operand(1), b = operand(5);
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (MODE & 1) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g], 0, 0, 0);
      }
    }
  } else if (MODE & 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) filler<F>(s, k);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 r = acc[0];
  for (int g = 1; g < 8; ++g) r += acc[g];
  out[blockIdx.x * 512 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + fill_sum(s);
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// (c) the weight-gradient batch: 16 accumulators, 96 MFMAs and 288 split instructions per batch (two dw_split8x2 per
// operand = per pair of values 3 cvt + 2 x (shift, and, pk_add neg)).  ORDER 0: all VALU, then all MFMAs (the round-5 loop);
// ORDER 1: 3 VALU behind every MFMA (the VALU of the NEXT batch in the shadow of this one's MFMAs);
// ORDER 2: like 1 with scalar v_sub_f32 residuals instead of v_pk_add_f32 (11 instead of 9 per pair -> 352: 3.67 per MFMA)
template <int PK>
__device__ __forceinline__ void split_slice(FillState& s, int q) {
  // one "pair" of dw_split8x2 is 18 instructions for two pairs: 6 cvt, 4 shift, 4 and, 4 pk_add.  Slice q of 6 = 3 of them.
  const int i = q & 7;
  switch (q % 6) {
    case 0:
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[0]) : "v"(s.p[i][0]), "v"(s.p[(i + 1) & 7][0]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[1]) : "v"(s.p[i][1]), "v"(s.p[(i + 1) & 7][1]));
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.u[2]) : "v"(s.u[0]));
      break;
    case 1:
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.u[3]) : "v"(s.u[1]));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.u[4]) : "v"(s.u[0]));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.u[5]) : "v"(s.u[1]));
      break;
    case 2:
    case 4:
      if (PK) {
        asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(s.p[i]) : "v"(s.p[(i + 2) & 7]));
        asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(s.p[(i + 1) & 7]) : "v"(s.p[(i + 3) & 7]));
      } else {
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[i]) : "v"(s.v[(i + 2) & 7]));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[(i + 1) & 7]) : "v"(s.v[(i + 3) & 7]));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[(i + 4) & 7]) : "v"(s.v[(i + 2) & 7]));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.v[(i + 5) & 7]) : "v"(s.v[(i + 3) & 7]));
      }
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[6]) : "v"(s.p[i][0]), "v"(s.p[(i + 1) & 7][0]));
      break;
    case 3:
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[7]) : "v"(s.p[i][1]), "v"(s.p[(i + 1) & 7][1]));
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.u[2]) : "v"(s.u[6]));
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.u[3]) : "v"(s.u[7]));
      break;
    default:
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.u[4]) : "v"(s.u[6]));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.u[5]) : "v"(s.u[7]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u[0]) : "v"(s.p[i][1]), "v"(s.p[(i + 1) & 7][1]));
      break;
  }
}

template <int ORDER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void dw_shape(float* out, long long* cyc, int iters) {
  f32x4 acc[16];
  for (int g = 0; g < 16; ++g) acc[g] = f32x4{0, 0, 0, 0};
  FillState s;
  fill_init(s);
  bf16x8 a[4], b[4];
  for (int g = 0; g < 4; ++g) {
    a[g] = operand(g);
    b[g] = operand(g + 9);
  }
  constexpr int PK = ORDER == 2 ? 0 : 1;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (ORDER == 0) {
#pragma unroll
      for (int q = 0; q < 96; ++q) split_slice<PK>(s, q);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 96; ++m)
        acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(m >> 2) & 3], b[m & 3], acc[m & 15], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int m = 0; m < 96; ++m) {
        acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(m >> 2) & 3], b[m & 3], acc[m & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        split_slice<PK>(s, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 r = acc[0];
  for (int g = 1; g < 16; ++g) r += acc[g];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + fill_sum(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float* g_out;
static long long* g_cyc;

template <class F>
static double timeit(F launch, long long* cycles, int ncyc = 1) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  launch();
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(cycles, g_cyc, sizeof(long long) * ncyc, hipMemcpyDeviceToHost);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms * 1e3 / 5;
}

template <int F>
static void one_filler(int iters) {
  long long c0, c;
  const double base = timeit([&] { hipLaunchKernelGGL((same_wave<F, 0>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters); }, &c0);
  const double n = 8.0 * iters;
  printf("\n== filler %s\n", kFillerName[F]);
  printf("  MFMA only, one wave per SIMD, 8 accumulators:      %8.1f us  %6.2f cycles per MFMA (s_memtime; wall clock at 2.4 GHz %.2f)\n", base,
         c0 / n, base * 2400.0 / n);
  long long cv;
  const double tv = timeit([&] { hipLaunchKernelGGL((valu_only<F, 4>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters); }, &cv);
  printf("  filler only, 32 per iteration:                     %8.1f us  %6.2f cycles per filler instruction\n", tv, cv / (32.0 * iters));
  const double per_valu = cv / (32.0 * iters);
#define ROW(K)                                                                                                                              \
  {                                                                                                                                         \
    const double t = timeit([&] { hipLaunchKernelGGL((same_wave<F, K>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters); }, &c);            \
    printf("  + %2d behind every MFMA: %8.1f us  %6.2f cycles per MFMA  (+%5.2f; serial would be +%5.2f; hidden %4.0f %%)\n", K, t, c / n,   \
           (c - c0) / n, K * per_valu, 100.0 * (1.0 - ((c - c0) / n) / (K * per_valu)));                                                    \
  }
  ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8)
#undef ROW
  long long ca[2], cb[2], cc[2];
  const double a = timeit([&] { hipLaunchKernelGGL((two_waves<F, 1>), dim3(256), dim3(512), 0, 0, g_out, g_cyc, iters); }, ca, 2);
  const double b = timeit([&] { hipLaunchKernelGGL((two_waves<F, 2>), dim3(256), dim3(512), 0, 0, g_out, g_cyc, iters); }, cb, 2);
  const double d = timeit([&] { hipLaunchKernelGGL((two_waves<F, 3>), dim3(256), dim3(512), 0, 0, g_out, g_cyc, iters); }, cc, 2);
  printf("  two waves per SIMD (8 MFMAs | 32 fillers per iteration): MFMA wave alone %.1f us, VALU wave alone %.1f us, together %.1f us  (sum %.1f, max %.1f)\n",
         a, b, d, a + b, a > b ? a : b);
  printf("     cycles per iteration: MFMA wave alone %.1f, VALU wave alone %.1f; together MFMA wave %.1f, VALU wave %.1f\n", (double)ca[0] / iters,
         (double)cb[1] / iters, (double)cc[0] / iters, (double)cc[1] / iters);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
  hipMalloc(&g_out, 512 * 512 * 4);
  hipMalloc(&g_cyc, 64);
  hipMemset(g_cyc, 0, 64);
  printf("# tools/exp/coexec_probe_bf16.hip: v_mfma_f32_16x16x32_bf16 next to the VALU instructions of the plane split (256 workgroups, %d iterations)\n", iters);
  one_filler<kFma>(iters);
  one_filler<kCvt>(iters);
  one_filler<kShiftAnd>(iters);
  one_filler<kSub>(iters);
  one_filler<kPkAdd>(iters);
  one_filler<kPkAddNeg>(iters);
  one_filler<kMov>(iters);

  printf("\n== the weight-gradient batch: 96 MFMAs + the split's VALU instructions per iteration\n");
  const int it2 = iters / 8;
  long long c;
#define SHAPE(ORDER, WAVES, LABEL)                                                                                                              \
  {                                                                                                                                             \
    const double t = timeit([&] { hipLaunchKernelGGL((dw_shape<ORDER, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, g_out, g_cyc, it2); }, &c);   \
    printf("  %-66s %d wave(s) per SIMD: %8.1f us  %7.1f cycles per batch and wave (96 MFMAs = 1536)\n", LABEL, WAVES / 4, t, (double)c / it2); \
  }
  SHAPE(0, 4, "288 VALU (packed residuals), then 96 MFMAs")
  SHAPE(1, 4, "3 VALU (packed residuals) behind every MFMA")
  SHAPE(2, 4, "3.67 VALU (scalar residuals: 352 per batch) behind every MFMA")
  SHAPE(0, 8, "288 VALU (packed residuals), then 96 MFMAs")
  SHAPE(1, 8, "3 VALU (packed residuals) behind every MFMA")
  SHAPE(2, 8, "3.67 VALU (scalar residuals: 352 per batch) behind every MFMA")
#undef SHAPE
  return 0;
}
