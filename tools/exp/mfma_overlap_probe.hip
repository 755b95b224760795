// Ground truth for tools/audit_mfma.py: which overlaps between the destination of a VGPR-form MFMA and its
// sources change the result on gfx950?  Every pattern runs with fixed physical registers inside one asm
// block and is compared with the non-overlapping form on the same random operands.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_overlap_probe.hip -o tools/exp/_build/mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// operands arrive in memory: a[4], b[4], c[4] dwords per lane; result d[4] per lane
#define LOAD3(A0, B0, C0)                                   \
  "global_load_dwordx4 v[" A0 "], %1, off\n"                \
  "global_load_dwordx4 v[" B0 "], %2, off\n"                \
  "global_load_dwordx4 v[" C0 "], %3, off\n"                \
  "s_waitcnt vmcnt(0)\n s_nop 7\n"
#define TAIL(D0) "s_nop 15\n s_nop 15\n s_nop 15\n global_store_dwordx4 %0, v[" D0 "], off\n s_waitcnt vmcnt(0)\n"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "memory"

#define BF16_CASE(ID, D, A, B, C)                                                                   \
  if (which == ID) {                                                                                \
    asm volatile(LOAD3(A, B, C) "v_mfma_f32_16x16x32_bf16 v[" D "], v[" A "], v[" B "], v[" C "]\n" \
                 TAIL(D) : : "v"(pd), "v"(pa), "v"(pb), "v"(pc) : CLOB);                            \
  }
// f32 16x16x4: A and B are ONE register each, moved into place from the first dword of the loaded vectors
#define F32_CASE(ID, D, A1, B1, C)                                                                  \
  if (which == ID) {                                                                                \
    asm volatile(LOAD3("8:11", "12:15", C) "v_mov_b32 " A1 ", v8\n v_mov_b32 " B1 ", v12\n s_nop 7\n" \
                 "v_mfma_f32_16x16x4_f32 v[" D "], " A1 ", " B1 ", v[" C "]\n"                      \
                 TAIL(D) : : "v"(pd), "v"(pa), "v"(pb), "v"(pc) : CLOB);                            \
  }

__global__ void probe(int which, const unsigned* a, const unsigned* b, const unsigned* c, unsigned* d) {
  const int l = threadIdx.x;
  const unsigned* pa = a + 4 * l;
  const unsigned* pb = b + 4 * l;
  const unsigned* pc = c + 4 * l;
  unsigned* pd = d + 4 * l;
  // ---- bf16 16x16x32 (8 bf16 per operand = 4 registers)
  BF16_CASE(0, "28:31", "16:19", "20:23", "24:27")      // reference: nothing overlaps
  BF16_CASE(1, "16:19", "16:19", "20:23", "24:27")      // dst == A
  BF16_CASE(2, "20:23", "16:19", "20:23", "24:27")      // dst == B
  BF16_CASE(3, "24:27", "16:19", "20:23", "24:27")      // dst == C (accumulate in place)
  BF16_CASE(4, "26:29", "16:19", "20:23", "24:27")      // dst partially over C (upper half of C)
  BF16_CASE(5, "22:25", "16:19", "28:31", "24:27")      // dst partially over C (lower half of C)
  BF16_CASE(6, "14:17", "16:19", "20:23", "24:27")      // dst partially over A
  BF16_CASE(7, "18:21", "12:15", "20:23", "24:27")      // dst partially over B
  // ---- f32 16x16x4 (one register per operand)
  F32_CASE(10, "28:31", "v32", "v33", "24:27")     // reference
  F32_CASE(11, "16:19", "v16", "v33", "24:27")     // A = dst reg 0
  F32_CASE(12, "16:19", "v17", "v33", "24:27")     // A = dst reg 1
  F32_CASE(13, "16:19", "v18", "v33", "24:27")     // A = dst reg 2
  F32_CASE(14, "16:19", "v19", "v33", "24:27")     // A = dst reg 3
  F32_CASE(15, "16:19", "v32", "v16", "24:27")     // B = dst reg 0
  F32_CASE(16, "16:19", "v32", "v17", "24:27")     // B = dst reg 1
  F32_CASE(17, "16:19", "v32", "v18", "24:27")     // B = dst reg 2
  F32_CASE(18, "16:19", "v32", "v19", "24:27")     // B = dst reg 3
  F32_CASE(19, "24:27", "v32", "v33", "24:27")     // dst == C
  F32_CASE(20, "26:29", "v32", "v33", "24:27")     // dst partially over C (upper half of C)
  F32_CASE(21, "22:25", "v32", "v33", "24:27")     // dst partially over C (lower half of C)
  F32_CASE(22, "16:19", "v32", "v17", "18:21")     // the instruction hipcc produced: B inside dst, dst partially over C
}

int main() {
  const int n = 64 * 4;
  std::vector<unsigned> a(n), b(n), c(n), ref(n), out(n);
  unsigned *da, *db, *dc, *dd;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
  int bad_total = 0;
  for (int family = 0; family < 2; ++family) {
    srand(17 + family);
    for (int i = 0; i < n; ++i) {
      float fa = (rand() % 2001 - 1000) / 500.0f, fb = (rand() % 2001 - 1000) / 500.0f, fc = (rand() % 2001 - 1000) / 50.0f;
      if (family == 0) {       // two bf16 per dword (truncate: any bit pattern is a fine operand)
        unsigned ua, ub, ua2, ub2;
        float fa2 = (rand() % 2001 - 1000) / 500.0f, fb2 = (rand() % 2001 - 1000) / 500.0f;
        memcpy(&ua, &fa, 4); memcpy(&ub, &fb, 4); memcpy(&ua2, &fa2, 4); memcpy(&ub2, &fb2, 4);
        a[i] = (ua >> 16) | (ua2 & 0xffff0000u);
        b[i] = (ub >> 16) | (ub2 & 0xffff0000u);
      } else {
        memcpy(&a[i], &fa, 4); memcpy(&b[i], &fb, 4);
      }
      memcpy(&c[i], &fc, 4);
    }
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    const int first = family == 0 ? 0 : 10, last = family == 0 ? 7 : 22;
    for (int w = first; w <= last; ++w) {
      hipMemset(dd, 0, n * 4);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, w, da, db, dc, dd);
      hipDeviceSynchronize();
      hipMemcpy(out.data(), dd, n * 4, hipMemcpyDeviceToHost);
      if (w == first) { ref = out; printf("case %2d reference (%s)\n", w, family == 0 ? "bf16 16x16x32" : "f32 16x16x4"); continue; }
      int bad = 0, bad_reg[4] = {0, 0, 0, 0};
      for (int i = 0; i < n; ++i) if (out[i] != ref[i]) { ++bad; ++bad_reg[i & 3]; }
      printf("case %2d: %3d / %d result dwords differ (per result register: %d %d %d %d)\n", w, bad, n, bad_reg[0], bad_reg[1], bad_reg[2], bad_reg[3]);
      bad_total += bad;
    }
  }
  printf("total differing dwords: %d\n", bad_total);
  return 0;
}
