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
//   wait    : one lane per workgroup polls the rank's own flag array until all P slots hold >= e
//             (bounded: a peer that never arrives sets the error word instead of hanging the GPU);
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
#include <cstring>

namespace rlg {

constexpr int kIpcMaxWorld = 16;
constexpr int kIpcBlocks = 32;          // small grid: co-resident with anything (and with a peer's copy on the same GPU)
constexpr int kIpcThreads = 256;
constexpr unsigned kIpcSpinLimit = 5u * 1000u * 1000u;    // polls of ~2 us (sleep + a system-scope load): ~10 s, then give up

struct IpcPeers {
  float* stage[2][kIpcMaxWorld];        // [parity][rank] staging buffers (own + mapped peers)
  unsigned* flags[kIpcMaxWorld];        // [rank] -> that rank's flag array (unsigned[kIpcMaxWorld])
  unsigned* state;                      // local, device memory: [0] launch ordinal, [1] arrival ticket, [2] done ticket, [3] error
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

__global__ __launch_bounds__(kIpcThreads) void ipc_allreduce_kernel(IpcPeers p, float* __restrict__ data, long long n,
                                                                    IpcNorm norm) {
  __shared__ unsigned s_epoch;
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
    // ---- wait for every rank's launch e
    const unsigned* my = p.flags[p.rank];
    unsigned spins = 0;
    for (int q = 0; q < p.world; ++q) {
      // epochs compare modulo 2^32 (signed difference): the ordinal wraps after 4e9 optimiser steps
      while (static_cast<int>(__hip_atomic_load(my + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kIpcSpinLimit) {
          __hip_atomic_store(p.state + 3, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __threadfence_system();
  }
  __syncthreads();

  // ---- reduce, rank order 0 .. P-1 on every rank
  double sq = 0.0;
  for (long long i = tid; i < n4; i += nthreads) {
    f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.stage[par][0]) + i);
    for (int q = 1; q < p.world; ++q) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.stage[par][q]) + i);
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
    float s = __builtin_nontemporal_load(p.stage[par][0] + i);
    for (int q = 1; q < p.world; ++q) s += __builtin_nontemporal_load(p.stage[par][q] + i);
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

struct IpcComm {
  IpcPeers peers;
  void* local;                 // base of the local allocation: stage0 | stage1 | flags
  void* opened[kIpcMaxWorld];  // mapped peer bases (nullptr for self)
  long long capacity;          // floats per staging buffer
  size_t stage_bytes;
  int fine_grained;
  int connected;
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
  const size_t total = 2 * c->stage_bytes + 256;
  hipError_t e = hipExtMallocWithFlags(&c->local, total, hipDeviceMallocFinegrained);
  c->fine_grained = (e == hipSuccess) ? 1 : 0;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&c->local, total);
  }
  if (e != hipSuccess) { delete c; return static_cast<int>(e); }
  e = hipMemset(c->local, 0, total);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->peers.state), 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->peers.state, 0, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), c->local);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(c->local); delete c; return static_cast<int>(e); }
  c->peers.rank = rank;
  c->peers.world = world;
  for (int q = 0; q < kIpcMaxWorld; ++q) {
    c->opened[q] = nullptr;
    c->peers.stage[0][q] = c->peers.stage[1][q] = nullptr;
    c->peers.flags[q] = nullptr;
  }
  c->connected = 0;
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
    c->peers.flags[q] = reinterpret_cast<unsigned*>(base + 2 * c->stage_bytes);
  }
  c->connected = 1;
  return 0;
}

int rlg_ipc_comm_fine_grained(void* comm) { return static_cast<rlg::IpcComm*>(comm)->fine_grained; }

// data[0..n) <- sum over ranks, in place.  Launch-only (capturable); n <= the capacity given at creation.
int rlg_ipc_allreduce_sum(void* comm, float* data, long long n, void* stream) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  if (!c->connected || n <= 0 || n > c->capacity || reinterpret_cast<uintptr_t>(data) % 16 != 0)
    return static_cast<int>(hipErrorInvalidValue);
  IpcNorm none = {nullptr, 0, nullptr, 1.0f};
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
  hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(kIpcBlocks), dim3(kIpcThreads), 0, static_cast<hipStream_t>(stream),
                     c->peers, data, n, norm);
  RLG_RETURN_LAUNCH_STATUS();
}

// Synchronises the device and reports: launches completed so far, and the ordinal of a launch that gave up
// waiting for a peer (0 = none).
int rlg_ipc_comm_status(void* comm, unsigned* launches_out, unsigned* timed_out_launch_out) {
  using namespace rlg;
  IpcComm* c = static_cast<IpcComm*>(comm);
  unsigned st[4];
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
