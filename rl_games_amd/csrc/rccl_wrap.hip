// RCCL behind the C ABI: the gradient all-reduce of A2CBase.trancate_gradients_and_step
// (rl_games/common/a2c_common.py:493-509: dist.all_reduce(all_grads, op=SUM)) as a plain launch on a HIP stream that
// takes an ncclComm_t - SURVEY.md 8(b): "plus an RCCL wrapper taking ncclComm_t".
//
// Why it exists next to torch.distributed (which IS RCCL on ROCm): a collective issued through torch.distributed runs
// between two graph replays per optimiser step (640 graph launches + 320 collectives per epoch and rank); issued on the
// capturing stream through this wrapper it becomes a node of the mini-epoch HIP graph like the hand-written hipIpc
// all-reduce (csrc/ipc_allreduce.hip), i.e. the fallback for a node on which that kernel fails its self-test keeps the
// one-graph step.  RCCL sums in ring / tree order: every rank receives the SAME bits (the reduced value is broadcast),
// so the ranks' parameters stay identical; the value itself may differ from the rank-ordered sum of the hipIpc kernel
// in the last bit.
//
// librccl is resolved at run time (dlopen, preferring an instance the process has loaded already): librlg_hip.so
// has no link-time dependency on it, and a process without RCCL simply gets rlg_rccl_available() == 0.

#include "rlg_device.hpp"
#include "rlg_hip.h"

#include <dlfcn.h>
#include <cstring>

namespace rlg {

// the part of rccl.h this file needs (ABI of NCCL 2.x / RCCL; checked against /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
constexpr int kNcclUniqueIdBytes = 128;
struct ncclUniqueId { char internal[kNcclUniqueIdBytes]; };
constexpr int kNcclSuccess = 0, kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;

struct RcclApi {
  void* handle = nullptr;
  int (*get_unique_id)(ncclUniqueId*) = nullptr;
  int (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*comm_destroy)(ncclComm_t) = nullptr;
  bool ok = false;
};

static RcclApi& rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {                       // an instance that is loaded already (torch's) first
      a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (a.handle) break;
    }
    for (const char* n : names) {
      if (a.handle) break;
      a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!a.handle) return a;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.ok = a.get_unique_id && a.comm_init_rank && a.all_reduce && a.comm_destroy;
    return a;
  }();
  return api;
}

// ncclResult_t -> the hipError_t convention of this library (0 = success)
static int rccl_status(int r) { return r == kNcclSuccess ? 0 : static_cast<int>(hipErrorUnknown); }

}  // namespace rlg

extern "C" {

int rlg_rccl_available(void) { return rlg::rccl_api().ok ? 1 : 0; }

int rlg_rccl_unique_id_bytes(void) { return rlg::kNcclUniqueIdBytes; }

int rlg_rccl_get_unique_id(void* id_out) {
  using namespace rlg;
  if (!rccl_api().ok || !id_out) return static_cast<int>(hipErrorNotSupported);
  ncclUniqueId id;
  const int r = rccl_api().get_unique_id(&id);
  if (r == kNcclSuccess) std::memcpy(id_out, &id, sizeof(id));
  return rccl_status(r);
}

int rlg_rccl_comm_create(const void* unique_id, int rank, int world, void** comm_out) {
  using namespace rlg;
  if (!rccl_api().ok) return static_cast<int>(hipErrorNotSupported);
  if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) return static_cast<int>(hipErrorInvalidValue);
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const int r = rccl_api().comm_init_rank(&comm, world, id, rank);
  *comm_out = (r == kNcclSuccess) ? comm : nullptr;
  return rccl_status(r);
}

int rlg_rccl_allreduce_sum(void* comm, float* data, long long n, void* stream) {
  using namespace rlg;
  if (!rccl_api().ok) return static_cast<int>(hipErrorNotSupported);
  if (!comm || !data || n <= 0) return static_cast<int>(hipErrorInvalidValue);
  return rccl_status(rccl_api().all_reduce(data, data, static_cast<size_t>(n), kNcclFloat32, kNcclSum,
                                           static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

int rlg_rccl_allreduce_sum_f64(void* comm, double* data, long long n, void* stream) {
  using namespace rlg;
  if (!rccl_api().ok) return static_cast<int>(hipErrorNotSupported);
  if (!comm || !data || n <= 0) return static_cast<int>(hipErrorInvalidValue);
  return rccl_status(rccl_api().all_reduce(data, data, static_cast<size_t>(n), kNcclFloat64, kNcclSum,
                                           static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

int rlg_rccl_comm_destroy(void* comm) {
  using namespace rlg;
  if (!rccl_api().ok) return static_cast<int>(hipErrorNotSupported);
  if (!comm) return 0;
  return rccl_status(rccl_api().comm_destroy(static_cast<ncclComm_t>(comm)));
}

}  // extern "C"
