// One-shot sum all-reduce over peer-mapped device memory (xGMI between the GPUs of a node), launch-only.
//
// Replaces the per-minibatch gradient all-reduce of A2CBase.trancate_gradients_and_step
// (rl_games/common/a2c_common.py:493-509: flatten grads, dist.all_reduce(SUM), divide by world) and the
// KL all-reduce of :1559-1560 (the KL rides in a tail slot of the same flat arena) for the <= 1 MB
// gradient arena of the BASELINE configs.  RCCL is latency-optimised for larger messages and needs a host
// launch per collective; this kernel is a plain launch that can be captured INSIDE the mini-epoch HIP
// graph, so a data-parallel optimiser step needs no host round trip at all.
//
// Protocol (every rank runs the same kernel on its own stream; P = world size):
//   publish : the rank copies its arena into its own staging buffer `stage[e & 1]` (e = the launch
//             ordinal, kept in device memory), releases at system scope, and the last workgroup to
//             finish writes `e` into slot [rank] of EVERY peer's flag array;
//   wait    : one lane per workgroup polls the rank's own flag array until all P slots hold >= e.  The wait is
//             bounded by wall time (s_memrealtime, default 600 s, rlg_ipc_comm_set_timeout / RLG_IPC_TIMEOUT_S:
//             long enough for any legitimate rank skew - a checkpoint written by rank 0, a slow env reset, a first
//             graph capture).  A launch that gives up - or any launch after one that did - is FAIL-SAFE: it sets
//             the sticky error word (state[3]), leaves zeros instead of a sum of stale staging data in `data`,
//             and the Adam launch that follows sees the same word and skips the step, so no parameter is touched
//             by invalid gradients; the host reads the word once per epoch and all ranks raise together;
//   reduce  : out[i] = stage_0[i] + stage_1[i] + ... + stage_{P-1}[i], the SAME order on every rank, so
//             all ranks end with bit-identical sums (the ranks' parameters never drift apart).
// Staging is double buffered by launch parity: a rank overwrites buffer e & 1 again at launch e + 2, and
// it can only get there after every peer has published launch e + 1, i.e. has finished reading launch e.
// One inter-GPU synchronisation per all-reduce.  Traffic per rank: (P - 1) x n floats read over xGMI
// (0.9 MB x 7 = 6.3 MB at P = 8, spread over 7 links), latency bound as intended for this size.
//
// Memory: one allocation per rank (2 staging buffers + flags), fine-grained (system-scope coherent) when the
// runtime provides it, exported with hipIpcGetMemHandle and mapped by the peers with hipIpcOpenMemHandle.

#include "rlg_device.hpp"
#include <cstdlib>
#include <cstring>

namespace rlg {

constexpr int kIpcMaxWorld = 16;
constexpr int kIpcBlocks = 32;          // small grid: co-resident with anything (and with a peer's copy on the same GPU)
constexpr int kIpcThreads = 256;
constexpr unsigned long long kIpcRealtimeHz = 100ull * 1000ull * 1000ull;    // s_memrealtime: constant 100 MHz on gfx9
constexpr double kIpcDefaultTimeoutS = 600.0;

struct IpcPeers {
  float* stage[2][kIpcMaxWorld];        // [parity][rank] staging buffers (own + mapped peers)
  unsigned* flags[kIpcMaxWorld];        // [rank] -> that rank's flag array (unsigned[kIpcMaxWorld])
  float* res[2][kIpcMaxWorld];          // two-phase variant: [parity][rank] reduced-chunk buffers
  unsigned* flags2[kIpcMaxWorld];       // two-phase variant: second flag array (chunk of rank q is reduced)
  unsigned* abort_word[kIpcMaxWorld];   // [rank] -> that rank's abort word (fine-grained, mapped): a rank that gives up
                                        // writes its launch ordinal into EVERY rank's word, so that the peers fail fast too
  unsigned* state;                      // local, device memory: [0] launch ordinal, [1] arrival ticket, [2] done ticket, [3] error (sticky),
                                        // [4] two-phase: chunk ticket
  unsigned long long timeout_ticks;     // of s_memrealtime; 0 = wait for ever
  int rank, world;
};

// Optional by-product (what grad_sumsq_kernel of csrc/optim.hip computes in a launch of its own):
// norm.partials[block] = sum over the block's share of the first norm.n REDUCED elements of
// (x * grad_scale)^2, and the Adam step counter advanced - the reduced gradients pass through registers here.
struct IpcNorm {
  double* partials;            // [kIpcBlocks] or nullptr
  long long n;                 // leading elements that are gradients (the arena's tail slots are not)
  long long* step_counter;     // or nullptr
  float grad_scale;
};

// Bounded wait of one lane for flags[q] >= e, q = 0 .. world-1 (see the protocol notes above); true = gave up.
// A rank that gives up tells every peer (abort words): a peer that is still waiting - or starts a later launch - fails
// at once instead of summing staging buffers the failed rank keeps overwriting.  Between the failure and the moment a
// peer sees the word the ranks can still differ by the steps the peer completed in between; the host's end-of-epoch
// check raises on every rank either way.
__device__ __forceinline__ bool ipc_wait_all(const IpcPeers& p, const unsigned* my, unsigned e) {
  const unsigned* my_abort = p.abort_word[p.rank];
  bool failed = __hip_atomic_load(p.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;   // sticky
  if (!failed && __hip_atomic_load(my_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
    __hip_atomic_store(p.state + 3, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    failed = true;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  for (int q = 0; q < p.world && !failed; ++q) {
    while (static_cast<int>(__hip_atomic_load(my + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 1023u) != 0u) continue;
      const bool peer_gave_up = __hip_atomic_load(my_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
      if (peer_gave_up || (p.timeout_ticks != 0ull && __builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks)) {
        __hip_atomic_store(p.state + 3, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!peer_gave_up) {
          for (int r = 0; r < p.world; ++r)
            __hip_atomic_store(p.abort_word[r], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        failed = true;
        break;
      }
    }
  }
  return failed;
}

__global__ __launch_bounds__(kIpcThreads) void ipc_allreduce_kernel(IpcPeers p, float* __restrict__ data, long long n,
                                                                    IpcNorm norm) {
  __shared__ unsigned s_epoch;
  __shared__ unsigned s_failed;
  if (threadIdx.x == 0) s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  __syncthreads();
  const unsigned e = s_epoch;
  const int par = static_cast<int>(e & 1u);
  const long long n4 = n >> 2;
  const long long tid = static_cast<long long>(blockIdx.x) * kIpcThreads + threadIdx.x;
  const long long nthreads = static_cast<long long>(gridDim.x) * kIpcThreads;

  // ---- publish
  float* mine = p.stage[par][p.rank];
  for (long long i = tid; i < n4; i += nthreads)
    reinterpret_cast<f32x4*>(mine)[i] = reinterpret_cast<const f32x4*>(data)[i];
  for (long long i = (n4 << 2) + tid; i < n; i += nthreads) mine[i] = data[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {                      // last workgroup of this rank: the whole arena is published
      __hip_atomic_store(p.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int q = 0; q < p.world; ++q)
        __hip_atomic_store(p.flags[q] + p.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait for every rank's launch e (epochs compare modulo 2^32: the ordinal wraps after 4e9 optimiser steps)
    const bool failed = ipc_wait_all(p, p.flags[p.rank], e);
    s_failed = failed ? 1u : 0u;
    __threadfence_system();
  }
  __syncthreads();
  const bool failed = s_failed != 0u;

  // ---- reduce, rank order 0 .. P-1 on every rank
  double sq = 0.0;
  for (long long i = tid; i < n4; i += nthreads) {
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!failed) {
      s = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.stage[par][0]) + i);
      for (int q = 1; q < p.world; ++q) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.stage[par][q]) + i);
    }
    reinterpret_cast<f32x4*>(data)[i] = s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (4 * i + k < norm.n) {
        const float g = s[k] * norm.grad_scale;
        sq = fma(static_cast<double>(g), static_cast<double>(g), sq);
      }
    }
  }
  for (long long i = (n4 << 2) + tid; i < n; i += nthreads) {
    float s = 0.0f;
    if (!failed) {
      s = __builtin_nontemporal_load(p.stage[par][0] + i);
      for (int q = 1; q < p.world; ++q) s += __builtin_nontemporal_load(p.stage[par][q] + i);
    }
    data[i] = s;
    if (i < norm.n) {
      const float g = s * norm.grad_scale;
      sq = fma(static_cast<double>(g), static_cast<double>(g), sq);
    }
  }
  if (norm.partials) {
    __shared__ double nscratch[kIpcThreads / kWave];
    double one[1] = {sq};
    block_sum<1, kIpcThreads>(one, nscratch);
    if (threadIdx.x == 0) {
      norm.partials[blockIdx.x] = one[0];
      if (blockIdx.x == 0 && norm.step_counter) *norm.step_counter += 1;
    }
  }

  // ---- the last workgroup to finish advances the launch ordinal
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.state + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(p.state + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.state, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// Two-phase variant (reduce-scatter + all-gather): rank r sums chunk r of every rank's staging buffer, publishes
// the reduced chunk, then collects the other ranks' reduced chunks.  Bytes over each xGMI link: 2 n / P per
// peer instead of n (0.15 MB instead of 0.6 MB at P = 8), at the price of a second inter-GPU synchronisation.
// Chunk q = elements [q * cn, (q + 1) * cn), cn = ceil(n / P) rounded up to 4.  Every rank adds the P partial
// values of an element in rank order 0 .. P-1, as the one-shot kernel does: bit-identical results on all ranks
// and between the two variants.  The gradient-norm partials are accumulated per chunk and combined in chunk
// order, so that they too are the same numbers on every rank.
__global__ __launch_bounds__(kIpcThreads) void ipc_allreduce2_kernel(IpcPeers p, float* __restrict__ data, long long n,
                                                                     IpcNorm norm) {
  __shared__ unsigned s_epoch;
  __shared__ unsigned s_failed;
  if (threadIdx.x == 0) s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  __syncthreads();
  const unsigned e = s_epoch;
  const int par = static_cast<int>(e & 1u);
  const long long tid = static_cast<long long>(blockIdx.x) * kIpcThreads + threadIdx.x;
  const long long nthreads = static_cast<long long>(gridDim.x) * kIpcThreads;
  const long long cn = (((n + p.world - 1) / p.world) + 3) & ~3LL;

  // ---- publish the whole arena
  float* mine = p.stage[par][p.rank];
  const long long n4 = n >> 2;
  for (long long i = tid; i < n4; i += nthreads)
    reinterpret_cast<f32x4*>(mine)[i] = reinterpret_cast<const f32x4*>(data)[i];
  for (long long i = (n4 << 2) + tid; i < n; i += nthreads) mine[i] = data[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(p.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int q = 0; q < p.world; ++q)
        __hip_atomic_store(p.flags[q] + p.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_failed = ipc_wait_all(p, p.flags[p.rank], e) ? 1u : 0u;
    __threadfence_system();
  }
  __syncthreads();
  bool failed = s_failed != 0u;

  // ---- reduce-scatter: this rank's chunk, summed in rank order; into data and into the published chunk buffer
  double sq_chunk[kIpcMaxWorld];
  for (int q = 0; q < kIpcMaxWorld; ++q) sq_chunk[q] = 0.0;
  {
    const long long lo = static_cast<long long>(p.rank) * cn;
    const long long hi = lo + cn < n ? lo + cn : n;
    float* out = p.res[par][p.rank];
    double sq = 0.0;
    for (long long i = lo + tid; i < hi; i += nthreads) {
      float s = 0.0f;
      if (!failed) {
        s = __builtin_nontemporal_load(p.stage[par][0] + i);
        for (int q = 1; q < p.world; ++q) s += __builtin_nontemporal_load(p.stage[par][q] + i);
      }
      out[i] = s;
      data[i] = s;
      if (i < norm.n) {
        const float g = s * norm.grad_scale;
        sq = fma(static_cast<double>(g), static_cast<double>(g), sq);
      }
    }
    sq_chunk[0] = sq;       // (slot 0 = own chunk here; re-ordered by chunk index below)
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.state + 4, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(p.state + 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int q = 0; q < p.world; ++q)
        __hip_atomic_store(p.flags2[q] + p.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_failed = (ipc_wait_all(p, p.flags2[p.rank], e) || failed) ? 1u : 0u;
    __threadfence_system();
  }
  __syncthreads();
  failed = s_failed != 0u;

  // ---- all-gather: the other ranks' reduced chunks
  const double own_sq = sq_chunk[0];
  sq_chunk[0] = 0.0;
  for (int q = 0; q < p.world; ++q) {
    if (q == p.rank) {
      sq_chunk[q] = own_sq;
      if (failed) {          // a failure after the own chunk was summed: leave zeros there as well
        const long long lo = static_cast<long long>(q) * cn, hi = lo + cn < n ? lo + cn : n;
        for (long long i = lo + tid; i < hi; i += nthreads) data[i] = 0.0f;
        sq_chunk[q] = 0.0;
      }
      continue;
    }
    const long long lo = static_cast<long long>(q) * cn;
    const long long hi = lo + cn < n ? lo + cn : n;
    const float* src = p.res[par][q];
    double sq = 0.0;
    for (long long i = lo + tid; i < hi; i += nthreads) {
      const float s = failed ? 0.0f : __builtin_nontemporal_load(src + i);
      data[i] = s;
      if (i < norm.n) {
        const float g = s * norm.grad_scale;
        sq = fma(static_cast<double>(g), static_cast<double>(g), sq);
      }
    }
    sq_chunk[q] = sq;
  }
  if (norm.partials) {
    __shared__ double nscratch[kIpcThreads / kWave];
    double one[1] = {0.0};
    for (int q = 0; q < p.world; ++q) one[0] += sq_chunk[q];       // chunk order: the same sum on every rank
    block_sum<1, kIpcThreads>(one, nscratch);
    if (threadIdx.x == 0) {
      norm.partials[blockIdx.x] = one[0];
      if (blockIdx.x == 0 && norm.step_counter) *norm.step_counter += 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.state + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(p.state + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.state, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct IpcComm {
  IpcPeers peers;
  void* local;                 // base of the local allocation: stage0 | stage1 | flags
  void* opened[kIpcMaxWorld];  // mapped peer bases (nullptr for self)
  long long capacity;          // floats per staging buffer
  size_t stage_bytes;
  int fine_grained;
  int connected;
  int two_phase;               // rlg_ipc_comm_set_variant
  double timeout_s;            // effective bound (<= 0: none)
};

}  // namespace rlg

extern "C" {

int rlg_ipc_handle_bytes(void) { return static_cast<int>(sizeof(hipIpcMemHandle_t)); }

// Allocates the rank's staging memory and exports its IPC handle.  handle_out: rlg_ipc_handle_bytes() bytes.
int rlg_ipc_comm_create(int rank, int world, long long max_floats, void** comm_out, void* handle_out) {
  using namespace rlg;
  if (world < 1 || world > kIpcMaxWorld || rank < 0 || rank >= world || max_floats <= 0) return static_cast<int>(hipErrorInvalidValue);
  IpcComm* c = new IpcComm();
  c->capacity = max_floats;
  c->stage_bytes = (static_cast<size_t>(max_floats) * sizeof(float) + 255) & ~static_cast<size_t>(255);
  const size_t total = 4 * c->stage_bytes + 256;      // stage0 | stage1 | res0 | res1 | flags [0,64) flags2 [128,192) abort word [192,196)
  // Fine-grained (system-scope coherent) memory or nothing: the protocol needs the peers' stores and flags to
  // become visible MID-KERNEL; coarse-grained memory is only coherent at kernel boundaries, and a self-test on
  // it can pass by timing luck.  Without it the caller uses RCCL.
  hipError_t e = hipExtMallocWithFlags(&c->local, total, hipDeviceMallocFinegrained);
  c->fine_grained = (e == hipSuccess) ? 1 : 0;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return static_cast<int>(e);
  }
  e = hipMemset(c->local, 0, total);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->peers.state), 8 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->peers.state, 0, 8 * sizeof(unsigned));
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), c->local);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(c->local); delete c; return static_cast<int>(e); }
  c->peers.rank = rank;
  c->peers.world = world;
  {
    // RLG_IPC_TIMEOUT_S: a positive number of seconds; "0" (or a negative number) = wait for ever; anything that
    // does not parse as a number is ignored (atof's 0 for garbage would have meant "for ever")
    double seconds = kIpcDefaultTimeoutS;
    if (const char* env = std::getenv("RLG_IPC_TIMEOUT_S")) {
      char* end = nullptr;
      const double v = std::strtod(env, &end);
      while (end && (*end == ' ' || *end == '\t' || *end == '\n')) ++end;
      if (end != env && end && *end == '\0') seconds = v;
    }
    c->timeout_s = seconds;
    c->peers.timeout_ticks = seconds > 0.0 ? static_cast<unsigned long long>(seconds * static_cast<double>(kIpcRealtimeHz)) : 0ull;
  }
  for (int q = 0; q < kIpcMaxWorld; ++q) {
    c->opened[q] = nullptr;
    c->peers.stage[0][q] = c->peers.stage[1][q] = nullptr;
    c->peers.flags[q] = nullptr;
    c->peers.res[0][q] = c->peers.res[1][q] = nullptr;
    c->peers.flags2[q] = nullptr;
    c->peers.abort_word[q] = nullptr;
  }
  c->connected = 0;
  c->two_phase = 0;
  if (const char* env = std::getenv("RLG_IPC_TWO_PHASE")) c->two_phase = std::atoi(env) != 0;
  *comm_out = c;
  return 0;
}

// all_handles: world x rlg_ipc_handle_bytes() bytes, rank-major (every rank's handle from rlg_ipc_comm_create).
int rlg_ipc_comm_connect(void* comm, const void* all_handles) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  const char* h = static_cast<const char*>(all_handles);
  for (int q = 0; q < c->peers.world; ++q) {
    char* base;
    if (q == c->peers.rank) {
      base = static_cast<char*>(c->local);
    } else {
      hipIpcMemHandle_t handle;
      std::memcpy(&handle, h + static_cast<size_t>(q) * sizeof(hipIpcMemHandle_t), sizeof(handle));
      void* mapped = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&mapped, handle, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return static_cast<int>(e);
      c->opened[q] = mapped;
      base = static_cast<char*>(mapped);
    }
    c->peers.stage[0][q] = reinterpret_cast<float*>(base);
    c->peers.stage[1][q] = reinterpret_cast<float*>(base + c->stage_bytes);
    c->peers.res[0][q] = reinterpret_cast<float*>(base + 2 * c->stage_bytes);
    c->peers.res[1][q] = reinterpret_cast<float*>(base + 3 * c->stage_bytes);
    c->peers.flags[q] = reinterpret_cast<unsigned*>(base + 4 * c->stage_bytes);
    c->peers.flags2[q] = reinterpret_cast<unsigned*>(base + 4 * c->stage_bytes + 128);
    c->peers.abort_word[q] = reinterpret_cast<unsigned*>(base + 4 * c->stage_bytes + 192);
  }
  c->connected = 1;
  return 0;
}

int rlg_ipc_comm_fine_grained(void* comm) { return static_cast<rlg::IpcComm*>(comm)->fine_grained; }

// Bound of a launch's wait for its peers, in seconds of wall time (<= 0: wait for ever).  Default 600 s or
// RLG_IPC_TIMEOUT_S.  Takes effect for launches (and graph captures) issued afterwards.
int rlg_ipc_comm_set_timeout(void* comm, double seconds) {
  using namespace rlg;
  static_cast<IpcComm*>(comm)->timeout_s = seconds;
  static_cast<IpcComm*>(comm)->peers.timeout_ticks =
      seconds > 0.0 ? static_cast<unsigned long long>(seconds * static_cast<double>(kIpcRealtimeHz)) : 0ull;
  return 0;
}

// The settings in effect (environment defaults included): what the host compares across ranks before the first launch.
int rlg_ipc_comm_get_config(void* comm, int* two_phase_out, double* timeout_s_out) {
  const rlg::IpcComm* c = static_cast<const rlg::IpcComm*>(comm);
  if (two_phase_out) *two_phase_out = c->two_phase;
  if (timeout_s_out) *timeout_s_out = c->timeout_s > 0.0 ? c->timeout_s : 0.0;
  return 0;
}

// 0: one-shot (every rank reads all peers' arenas: one synchronisation), 1: two-phase (reduce-scatter +
// all-gather: 2/P of the bytes per link, two synchronisations).  Default 0 or RLG_IPC_TWO_PHASE.  Both give the
// same bits.  Collective choice: every rank must use the same variant for a given launch.
int rlg_ipc_comm_set_variant(void* comm, int two_phase) {
  static_cast<rlg::IpcComm*>(comm)->two_phase = two_phase != 0;
  return 0;
}

// Device address of the sticky error word (unsigned: 0 = healthy, else the ordinal of the launch that gave up):
// rlg_adam_step takes it as `skip_flag_or_null` so that a step behind a failed all-reduce changes nothing.
int rlg_ipc_comm_error_word(void* comm, unsigned** word_out) {
  *word_out = static_cast<rlg::IpcComm*>(comm)->peers.state + 3;
  return 0;
}

// data[0..n) <- sum over ranks, in place.  Launch-only (capturable); n <= the capacity given at creation.
int rlg_ipc_allreduce_sum(void* comm, float* data, long long n, void* stream) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  if (!c->connected || n <= 0 || n > c->capacity || reinterpret_cast<uintptr_t>(data) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  IpcNorm none = {nullptr, 0, nullptr, 1.0f};
  if (c->two_phase)
    hipLaunchKernelGGL(ipc_allreduce2_kernel, dim3(kIpcBlocks), dim3(kIpcThreads), 0, static_cast<hipStream_t>(stream),
                       c->peers, data, n, none);
  else
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(kIpcBlocks), dim3(kIpcThreads), 0, static_cast<hipStream_t>(stream),
                       c->peers, data, n, none);
  RLG_RETURN_LAUNCH_STATUS();
}

int rlg_ipc_allreduce_norm_blocks(void) { return rlg::kIpcBlocks; }

// The same launch, which also leaves norm_partials[rlg_ipc_allreduce_norm_blocks()] = per-block sums of
// (reduced x * grad_scale)^2 over the first norm_n elements and advances step_counter - the inputs of
// rlg_adam_step's gradient clipping, i.e. rlg_grad_sumsq without its launch.
int rlg_ipc_allreduce_sum_norm(void* comm, float* data, long long n, double* norm_partials, long long norm_n,
                               float grad_scale, long long* step_counter_or_null, void* stream) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  if (!c->connected || n <= 0 || n > c->capacity || reinterpret_cast<uintptr_t>(data) % 16 != 0 || !norm_partials ||
      norm_n < 0 || norm_n > n)
    return static_cast<int>(hipErrorInvalidValue);
  IpcNorm norm = {norm_partials, norm_n, step_counter_or_null, grad_scale};
  if (c->two_phase)
    hipLaunchKernelGGL(ipc_allreduce2_kernel, dim3(kIpcBlocks), dim3(kIpcThreads), 0, static_cast<hipStream_t>(stream),
                       c->peers, data, n, norm);
  else
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(kIpcBlocks), dim3(kIpcThreads), 0, static_cast<hipStream_t>(stream),
                       c->peers, data, n, norm);
  RLG_RETURN_LAUNCH_STATUS();
}

// Synchronises the device and reports: launches completed so far, and the ordinal of a launch that gave up
// waiting for a peer (0 = none).
int rlg_ipc_comm_status(void* comm, unsigned* launches_out, unsigned* timed_out_launch_out) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  unsigned st[8];
  const hipError_t e = hipMemcpy(st, c->peers.state, sizeof(st), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return static_cast<int>(e);
  *launches_out = st[0];
  *timed_out_launch_out = st[3];
  return 0;
}

int rlg_ipc_comm_destroy(void* comm) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < kIpcMaxWorld; ++q) {
    if (c->opened[q]) (void)hipIpcCloseMemHandle(c->opened[q]);
  }
  (void)hipFree(c->peers.state);
  (void)hipFree(c->local);
  delete c;
  return 0;
}

}  // extern "C"
