// Second ground-truth probe for tools/audit_mfma.py (gfx950): a DEPENDENT pair of MFMAs,
//     MFMA1: v[8:11] <- A1 x B1 + v[8:11]            (accumulates in place)
//     k wait states (s_nop) or k independent MFMAs
//     MFMA2: D2     <- A2 x B2 + v[8:11]             (reads MFMA1's result as SrcC)
// with D2 = v[8:11] (in place), v[12:15] (moves the accumulator) or v[10:13] (moves it onto a partial
// overlap, upper or lower half).  Is the hardware interlocked for every D2, or does the moved form need software wait states?
// The result is compared with the same pair separated by 48 wait states.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_chain_probe.hip -o tools/exp/_build/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "memory"
// a -> v[16:19], b -> v[20:23], c -> v[8:11]; second operand pair a2 -> v[24:27], b2 -> v[28:31]
#define LOADS                                   \
  "global_load_dwordx4 v[16:19], %1, off\n"     \
  "global_load_dwordx4 v[20:23], %2, off\n"     \
  "global_load_dwordx4 v[8:11], %3, off\n"      \
  "global_load_dwordx4 v[24:27], %4, off\n"     \
  "global_load_dwordx4 v[28:31], %5, off\n"     \
  "s_waitcnt vmcnt(0)\n s_nop 7\n"
#define TAIL(D) "s_nop 15\n s_nop 15\n s_nop 15\n global_store_dwordx4 %0, v[" D "], off\n s_waitcnt vmcnt(0)\n"
#define ARGS : : "v"(pd), "v"(pa), "v"(pb), "v"(pc), "v"(pa2), "v"(pb2) : CLOB

#define F32_PAIR(GAP, D2) LOADS "v_mfma_f32_16x16x4_f32 v[8:11], v16, v20, v[8:11]\n" GAP \
                          "v_mfma_f32_16x16x4_f32 v[" D2 "], v24, v28, v[8:11]\n" TAIL(D2)
#define BF_PAIR(GAP, D2)  LOADS "v_mfma_f32_16x16x32_bf16 v[8:11], v[16:19], v[20:23], v[8:11]\n" GAP \
                          "v_mfma_f32_16x16x32_bf16 v[" D2 "], v[24:27], v[28:31], v[8:11]\n" TAIL(D2)
// an independent MFMA as gap filler (its own registers)
#define IND_F32 "v_mfma_f32_16x16x4_f32 v[32:35], v17, v21, v[32:35]\n"
#define IND_BF  "v_mfma_f32_16x16x32_bf16 v[32:35], v[16:19], v[20:23], v[32:35]\n"

#define GAPS(X, PAIR, D2, BASE, IND)                                    \
  X(BASE + 0, PAIR("s_nop 15\n s_nop 15\n s_nop 15\n", D2))            \
  X(BASE + 1, PAIR("", D2))                                            \
  X(BASE + 2, PAIR("s_nop 0\n", D2))                                   \
  X(BASE + 3, PAIR("s_nop 1\n", D2))                                   \
  X(BASE + 4, PAIR("s_nop 3\n", D2))                                   \
  X(BASE + 5, PAIR("s_nop 5\n", D2))                                   \
  X(BASE + 6, PAIR("s_nop 7\n", D2))                                   \
  X(BASE + 7, PAIR("s_nop 9\n", D2))                                   \
  X(BASE + 8, PAIR("s_nop 13\n", D2))                                  \
  X(BASE + 9, PAIR("s_nop 15\n s_nop 3\n", D2))                        \
  X(BASE + 10, PAIR(IND, D2))                                          \
  X(BASE + 11, PAIR(IND IND, D2))                                      \
  X(BASE + 12, PAIR(IND IND IND, D2))
#define CASE(ID, TEXT) if (which == (ID)) { asm volatile(TEXT ARGS); }

__global__ void probe(int which, const unsigned* a, const unsigned* b, const unsigned* c, const unsigned* a2,
                      const unsigned* b2, unsigned* d) {
  const int l = threadIdx.x;
  const unsigned *pa = a + 4 * l, *pb = b + 4 * l, *pc = c + 4 * l, *pa2 = a2 + 4 * l, *pb2 = b2 + 4 * l;
  unsigned* pd = d + 4 * l;
  GAPS(CASE, F32_PAIR, "8:11", 0, IND_F32)
  GAPS(CASE, F32_PAIR, "12:15", 20, IND_F32)
  GAPS(CASE, F32_PAIR, "10:13", 40, IND_F32)
  GAPS(CASE, F32_PAIR, "6:9", 60, IND_F32)
  GAPS(CASE, BF_PAIR, "8:11", 100, IND_BF)
  GAPS(CASE, BF_PAIR, "12:15", 120, IND_BF)
  GAPS(CASE, BF_PAIR, "10:13", 140, IND_BF)
  GAPS(CASE, BF_PAIR, "6:9", 160, IND_BF)
}

static const char* kGap[13] = {"48 wait states (reference)", "back to back", "s_nop 0", "s_nop 1", "s_nop 3", "s_nop 5", "s_nop 7",
                               "s_nop 9", "s_nop 13", "s_nop 15+3", "1 independent MFMA", "2 independent MFMAs", "3 independent MFMAs"};

int main() {
  const int n = 64 * 4;
  std::vector<unsigned> h[5], ref(n), out(n);
  unsigned* dev[6];
  for (int k = 0; k < 6; ++k) hipMalloc(&dev[k], n * 4);
  for (int family = 0; family < 2; ++family) {
    srand(5 + family);
    for (int k = 0; k < 5; ++k) {
      h[k].resize(n);
      for (int i = 0; i < n; ++i) {
        float f = (rand() % 2001 - 1000) / 500.0f, g = (rand() % 2001 - 1000) / 500.0f;
        unsigned uf, ug;
        memcpy(&uf, &f, 4); memcpy(&ug, &g, 4);
        h[k][i] = (family == 1 && k != 2) ? ((uf >> 16) | (ug & 0xffff0000u)) : uf;     // k == 2: the fp32 C operand
      }
      hipMemcpy(dev[k], h[k].data(), n * 4, hipMemcpyHostToDevice);
    }
    for (int form = 0; form < 4; ++form) {
      const int base = (family ? 100 : 0) + 20 * form;
      printf("%s, MFMA2 destination %s\n", family ? "bf16 16x16x32" : "f32 16x16x4",
             form == 0 ? "v[8:11] = SrcC (in place)" : form == 1 ? "v[12:15] (accumulator moves)" :
             form == 2 ? "v[10:13] (moves; dst registers 0,1 = SrcC registers 2,3)" : "v[6:9] (moves; dst registers 2,3 = SrcC registers 0,1)");
      for (int g = 0; g < 13; ++g) {
        hipMemset(dev[5], 0, n * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, base + g, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5]);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), dev[5], n * 4, hipMemcpyDeviceToHost);
        if (g == 0) { ref = out; continue; }
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += out[i] != ref[i];
        printf("   %-24s %3d / %d result dwords differ\n", kGap[g], bad, n);
      }
    }
  }
  return 0;
}
