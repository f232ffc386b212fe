// Library-owned RCCL communicator: one process per GPU, the Monte-Carlo batch shards with no
// data-path collective; the ONE exchange per batch is an all-gather of a fixed 88-byte record per scenario
// {status, iterations, QP solves, rank, p_feas, comp, stat, cost[0..5]} over xGMI, plus a barrier and a max-reduction for the
// benchmark's timing.  RCCL is loaded at run time (dlopen) so that the library also loads on hosts without it; nothing here
// needs PyTorch.  Included by dgsqp_api.hip after struct dgsqp_solver.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !AllReduce) { err = "librccl lacks a required symbol"; lib = nullptr; return false; }
    return true;
  }
} g_rccl;
}  // namespace

struct dgsqp_comm_state {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  dgsqp_stat_record_t *d_rec = nullptr, *d_all = nullptr;
  int64_t rec_cap = 0;          // records d_rec holds; d_all holds world * rec_cap
  double *d_red = nullptr;      // small buffer for reductions
};

#define NCCLCHK(h, call)                                                                                    \
  do {                                                                                                      \
    ncclResult_t r_ = (call);                                                                               \
    if (r_ != ncclSuccess) {                                                                                \
      (h)->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
      return DGSQP_E_DEVICE;                                                                                \
    }                                                                                                       \
  } while (0)

// one record per scenario from the result arrays of the handle's last solve; rows beyond B (padding of the all-gather) get status -1
__global__ void dg_pack_stats_kernel(int64_t B, int64_t Bpad, int M, int rank, const int32_t* __restrict__ status, const int32_t* __restrict__ iters,
                                     const int32_t* __restrict__ qps, const double* __restrict__ cond, const double* __restrict__ cost,
                                     dgsqp_stat_record_t* __restrict__ out) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= Bpad) return;
  dgsqp_stat_record_t r;
  r.rank = rank;
  if (b < B) {
    r.status = status[b]; r.iters = iters[b]; r.qp_solves = qps[b];
    r.p_feas = cond[3 * b]; r.comp = cond[3 * b + 1]; r.stat = cond[3 * b + 2];
    for (int a = 0; a < DGSQP_MAX_AGENTS; a++) r.cost[a] = a < M ? cost[b * M + a] : 0.0;
  } else {
    r.status = -1; r.iters = 0; r.qp_solves = 0; r.p_feas = r.comp = r.stat = 0.0;
    for (int a = 0; a < DGSQP_MAX_AGENTS; a++) r.cost[a] = 0.0;
  }
  out[b] = r;
}
