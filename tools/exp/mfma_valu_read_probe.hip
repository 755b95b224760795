// Third ground-truth probe (gfx950): how many wait states does a VALU read of an MFMA result need?
// hipcc (ROCm 7.2) separates v_mfma_f32_16x16x4_f32 from a VALU reader of its VGPR result by 10 wait states;
// the 8-wave backward instance of csrc/mlp_chain.hip produced wrong fragment registers 2 and 3 in that form.
//     MFMA: v[8:11] <- A x B + v[8:11];  k wait states;  v_mov of the four result registers (ascending or
//     descending order);  compared with the same sequence separated by 64 wait states.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_valu_read_probe.hip -o tools/exp/_build/mfma_valu_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CLOB "a0", "a1", "a2", "a3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "memory"
#define LOADS                                   \
  "global_load_dwordx4 v[16:19], %1, off\n"     \
  "global_load_dwordx4 v[20:23], %2, off\n"     \
  "global_load_dwordx4 v[8:11], %3, off\n"      \
  "s_waitcnt vmcnt(0)\n s_nop 7\n"
#define ASC  "v_mov_b32 v12, v8\n v_mov_b32 v13, v9\n v_mov_b32 v14, v10\n v_mov_b32 v15, v11\n"
#define DESC "v_mov_b32 v15, v11\n v_mov_b32 v14, v10\n v_mov_b32 v13, v9\n v_mov_b32 v12, v8\n"
#define TAIL "s_nop 15\n s_nop 15\n global_store_dwordx4 %0, v[12:15], off\n s_waitcnt vmcnt(0)\n"
#define ARGS : : "v"(pd), "v"(pa), "v"(pb), "v"(pc) : CLOB
#define F32(GAP, RD) LOADS "v_mfma_f32_16x16x4_f32 v[8:11], v16, v20, v[8:11]\n" GAP RD TAIL
#define BF(GAP, RD)  LOADS "v_mfma_f32_16x16x32_bf16 v[8:11], v[16:19], v[20:23], v[8:11]\n" GAP RD TAIL
// AGPR form: accumulator a[0:3] (written from v[8:11] first), read back with v_accvgpr_read
#define TOACC "v_accvgpr_write_b32 a0, v8\n v_accvgpr_write_b32 a1, v9\n v_accvgpr_write_b32 a2, v10\n v_accvgpr_write_b32 a3, v11\n s_nop 7\n"
#define AASC  "v_accvgpr_read_b32 v12, a0\n v_accvgpr_read_b32 v13, a1\n v_accvgpr_read_b32 v14, a2\n v_accvgpr_read_b32 v15, a3\n"
#define ADESC "v_accvgpr_read_b32 v15, a3\n v_accvgpr_read_b32 v14, a2\n v_accvgpr_read_b32 v13, a1\n v_accvgpr_read_b32 v12, a0\n"
#define F32A(GAP, RD) LOADS TOACC "v_mfma_f32_16x16x4_f32 a[0:3], v16, v20, a[0:3]\n" GAP RD TAIL
#define BFA(GAP, RD)  LOADS TOACC "v_mfma_f32_16x16x32_bf16 a[0:3], v[16:19], v[20:23], a[0:3]\n" GAP RD TAIL
#define CASE(ID, TEXT) if (which == (ID)) { asm volatile(TEXT ARGS); }
#define GAPS(KIND, RD, BASE)                                                   \
  CASE(BASE + 0, KIND("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n", RD))      \
  CASE(BASE + 1, KIND("", RD))                                                 \
  CASE(BASE + 2, KIND("s_nop 0\n", RD))                                        \
  CASE(BASE + 3, KIND("s_nop 1\n", RD))                                        \
  CASE(BASE + 4, KIND("s_nop 2\n", RD))                                        \
  CASE(BASE + 5, KIND("s_nop 3\n", RD))                                        \
  CASE(BASE + 6, KIND("s_nop 4\n", RD))                                        \
  CASE(BASE + 7, KIND("s_nop 5\n", RD))                                        \
  CASE(BASE + 8, KIND("s_nop 6\n", RD))                                        \
  CASE(BASE + 9, KIND("s_nop 7\n", RD))                                        \
  CASE(BASE + 10, KIND("s_nop 8\n", RD))                                       \
  CASE(BASE + 11, KIND("s_nop 9\n", RD))                                       \
  CASE(BASE + 12, KIND("s_nop 10\n", RD))                                      \
  CASE(BASE + 13, KIND("s_nop 11\n", RD))                                      \
  CASE(BASE + 14, KIND("s_nop 12\n", RD))                                      \
  CASE(BASE + 15, KIND("s_nop 13\n", RD))                                      \
  CASE(BASE + 16, KIND("s_nop 14\n", RD))                                      \
  CASE(BASE + 17, KIND("s_nop 15\n", RD))                                      \
  CASE(BASE + 18, KIND("s_nop 15\n s_nop 1\n", RD))                            \
  CASE(BASE + 19, KIND("s_nop 15\n s_nop 3\n", RD))                            \
  CASE(BASE + 20, KIND("s_nop 15\n s_nop 7\n", RD))

__global__ void probe(int which, const unsigned* a, const unsigned* b, const unsigned* c, unsigned* d) {
  const int l = threadIdx.x;
  const unsigned *pa = a + 4 * l, *pb = b + 4 * l, *pc = c + 4 * l;
  unsigned* pd = d + 4 * l;
  GAPS(F32, ASC, 0)
  GAPS(F32, DESC, 30)
  GAPS(BF, ASC, 100)
  GAPS(BF, DESC, 130)
  GAPS(F32A, AASC, 200)
  GAPS(F32A, ADESC, 230)
  GAPS(BFA, AASC, 300)
  GAPS(BFA, ADESC, 330)
}

static const int kWait[21] = {64, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24};

int main() {
  const int n = 64 * 4;
  std::vector<unsigned> h[3], ref(n), out(n);
  unsigned* dev[4];
  for (int k = 0; k < 4; ++k) hipMalloc(&dev[k], n * 4);
  for (int family = 0; family < 4; ++family) {
    srand(5 + (family & 1));
    for (int k = 0; k < 3; ++k) {
      h[k].resize(n);
      for (int i = 0; i < n; ++i) {
        float f = (rand() % 2001 - 1000) / 500.0f, g = (rand() % 2001 - 1000) / 500.0f;
        unsigned uf, ug;
        memcpy(&uf, &f, 4); memcpy(&ug, &g, 4);
        h[k][i] = ((family & 1) && k != 2) ? ((uf >> 16) | (ug & 0xffff0000u)) : uf;
      }
      hipMemcpy(dev[k], h[k].data(), n * 4, hipMemcpyHostToDevice);
    }
    for (int order = 0; order < 2; ++order) {
      const int base = (family & 1 ? 100 : 0) + (family >= 2 ? 200 : 0) + 30 * order;
      printf("%s%s, result registers read %s after k wait states\n", family & 1 ? "bf16 16x16x32" : "f32 16x16x4",
             family >= 2 ? " (AGPR destination, v_accvgpr_read)" : "",
             order ? "3,2,1,0" : "0,1,2,3");
      for (int g = 0; g < 21; ++g) {
        hipMemset(dev[3], 0, n * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, base + g, dev[0], dev[1], dev[2], dev[3]);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), dev[3], n * 4, hipMemcpyDeviceToHost);
        if (g == 0) { ref = out; continue; }
        int bad[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) bad[i & 3] += out[i] != ref[i];
        printf("   k = %2d: wrong lanes per result register: %2d %2d %2d %2d\n", kWait[g], bad[0], bad[1], bad[2], bad[3]);
      }
    }
  }
  return 0;
}
