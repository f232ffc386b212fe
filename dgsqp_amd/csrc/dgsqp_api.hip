// C-ABI of the MI355X batched DG-SQP solver (include/dgsqp.h) and its kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libdgsqp_hip.so dgsqp_api.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <vector>
#include <mutex>
#include <algorithm>

#include "dgsqp_solve.h"
#include "dgsqp_xl.h"
#include "dgsqp_osqp_xl.h"
#include "dgsqp_solve_v2.h"


// one staged batch of a grouped launch: its inputs and its outputs
struct DgBatch { const double* x0; const double* u_ws; SolveOutPtrs O; };
#define DG_GROUP_MAX 64

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// Persistent solve kernel: each workgroup pulls scenarios from a device-wide ticket counter, so that
// scenarios with long SQP runs (iteration counts vary 1..50+) do not serialise a static partition.
__global__ void __launch_bounds__(DG_BLOCK, 2)
dg_solve_kernel(const DgProb* __restrict__ D, int64_t B, const double* __restrict__ x0, const double* __restrict__ u_ws,
                SolveOutPtrs O, double* __restrict__ ws_all, unsigned long long* __restrict__ ticket,
                double* __restrict__ trace, int trace_cap, unsigned int* __restrict__ drained,
                double* __restrict__ itlog, int itlog_cap, const DgBatch* __restrict__ group, int group_n,
                DgCoop* coop, double* coop_payload, int coop_start, int coop_verify, int coop_window, int coop_helpers, DgPark park) {
  Ctx c;
  c.coop = coop;
  c.park = park;
  if (!coop) c.park.entries = nullptr;
  c.ticket = ticket;
  c.coop_start = coop_start; c.coop_verify = coop_verify; c.coop_window = coop_window; c.coop_helpers = coop_helpers;
  c.coop_payload = coop ? coop_payload + (size_t)blockIdx.x * 2 * (2 * dg_prob.n + 2 * dg_prob.nc) : nullptr;
  c.coop_total = (unsigned long long)(B * (group ? group_n : 1));
  c.trace_cap = trace_cap;
  c.itlog_cap = itlog_cap;
  c.ws = (gptr)ws_all + (int64_t)blockIdx.x * dg_prob.ws_doubles;
#ifdef DG_PROF
  const long long wg_t0 = clock64();
  const unsigned long long wall0 = wall_clock64();
#endif
  dev_load_tables();
  if (TID == 0) dg_lds[dg_prob.L.scal + DG_COOP_FLIP] = 0.0;
  if (coop && TID == 0) { unsigned long long zero = 0ull; __hip_atomic_compare_exchange_strong(&coop->t_first, &zero, wall_clock64(), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  while (true) {
    __syncthreads();
    if (TID == 0) dg_lds[dg_prob.L.scal + 63] = (double)atomicAdd(ticket, 1ULL);
    __syncthreads();
    int64_t b = (int64_t)dg_lds[dg_prob.L.scal + 63];
    const DgParkEntry* resume = nullptr;
    if (b >= B * (group ? group_n : 1)) {
      // the queue is empty: from now on this launch only drains.  Tell the host (mapped, fine-grained memory) so that it
      // can start the next independent batch on the compute units that become free.
      if (drained && TID == 0) { __hip_atomic_store(drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      if (!coop) break;
      // cooperative launches: resume a deferred scenario, the one that has cost the most so far first; with none waiting, stay and
      // evaluate line-search trials for the workgroups that are still solving (until a deferred scenario turns up or the launch ends)
      const long long slot = dev_park_pop(c);
      if (slot < 0) {
        if (dev_coop_help(c)) continue;
        break;
      }
      dev_park_load(c, (unsigned int)slot);
      resume = &c.park.entries[slot];
      b = (int64_t)resume->ticket;
      if (TID == 0) c.park.entries[slot].t_resume = wall_clock64() - AT_LOAD(&coop->t_first);
    }
    const int64_t tk = b;
    if (group) {        // grouped launch (dgsqp_launch_staged_group): ticket -> (staged batch, scenario); every batch has its own buffers
      const int64_t gi = b / B;
      b -= gi * B;
      x0 = group[gi].x0; u_ws = group[gi].u_ws; O = group[gi].O;
    }
    c.x0 = (cgptr)x0 + b * dg_prob.nq;
    c.trace = trace ? (gptr)trace + b * (int64_t)(1 + 2 * trace_cap) : nullptr;
    if (c.trace && TID == 0 && !resume) c.trace[0] = 0.0;
    c.itlog = itlog ? (gptr)itlog + b * (1 + (int64_t)itlog_cap * (dg_prob.n + dg_prob.nc)) : nullptr;
#ifdef DG_PROF
    const long long sc_t0 = clock64();
#endif
    bool deferred = false;
    int its = 0;
    const unsigned long long ticks0 = c.park.entries ? dev_bcast_u64(TID == 0 ? wall_clock64() : 0ull) : 0ull;
    if (dg_prob.par.variant == DGSQP_VARIANT_V2) deferred = dev_solve_v2(c, (cgptr)u_ws + b * dg_prob.n, b, O, resume, (long long)tk, ticks0, &its);
    else deferred = dev_solve(c, (cgptr)u_ws + b * dg_prob.n, b, O, resume, (long long)tk, ticks0, &its);
    if (resume && TID == 0) {
      DgParkEntry* e = &c.park.entries[resume - c.park.entries];
      e->t_done = wall_clock64() - AT_LOAD(&coop->t_first); e->final_its = its; e->final_qps = O.qp_solves ? O.qp_solves[b] : 0;
    }
    if (coop && !deferred && TID == 0) {
      __threadfence();
      if (!resume && c.park.entries) {
        __hip_atomic_fetch_add(&coop->done_ticks, wall_clock64() - ticks0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&coop->done_fresh, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_fetch_add(&coop->done_iters, (unsigned long long)its, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&coop->finished, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef DG_PROF
    if (TID == 0 && tk < 16384) dg_prof_scn[tk] = (resume ? dg_prof_scn[tk] : 0ull) + (unsigned long long)(clock64() - sc_t0);
#endif
  }
#ifdef DG_PROF
  if (TID == 0) {
    atomicAdd(&dg_prof[2 * PH_WGTOTAL], (unsigned long long)(clock64() - wg_t0));
    atomicAdd(&dg_prof[2 * PH_WGTOTAL + 1], 1ULL);
    atomicMax(&dg_prof[2 * PH_WGMAX], (unsigned long long)(clock64() - wg_t0));
    atomicMax(&dg_prof[2 * PH_WGMAX + 1], (unsigned long long)(wall_clock64() - wall0));
  }
#endif
}

#include "dgsqp_pid.h"
#include "dgsqp_sampler.h"

// Test hook: one _evaluate(hessian=True) (+ dual init) per scenario, results expanded to dense arrays.
__global__ void __launch_bounds__(DG_BLOCK, 2)
dg_evaluate_kernel(const DgProb* __restrict__ D, int64_t B, const double* __restrict__ x0, const double* __restrict__ u,
                   const double* __restrict__ l, double* q, double* g, double* G, double* Q, double* x, double* l0,
                   double* __restrict__ ws_all) {
  Ctx c;
  c.coop = nullptr; c.coop_payload = nullptr; c.coop_total = 0; c.coop_start = 0; c.coop_verify = 0; c.coop_window = 0; c.coop_helpers = 0;
  c.ws = (gptr)ws_all + (int64_t)blockIdx.x * dg_prob.ws_doubles;
  c.trace = nullptr; c.trace_cap = 0; c.itlog = nullptr; c.itlog_cap = 0;
  const DgLds& L = dg_prob.L;
  const int n = dg_prob.n, nc = dg_prob.nc;
  dev_load_tables();
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    c.x0 = (cgptr)x0 + b * dg_prob.nq;
    __syncthreads();
    if (TID == 0) { dg_lds[L.scal + DG_XVALID] = 0.0; dg_lds[L.scal + DG_REG] = dg_prob.par.reg; dg_lds[L.scal + DG_OSQP_RHO] = 0.1; }
    for (int i = TID; i < n; i += NT) dg_lds[L.u + i] = u[b * n + i];
    for (int r = TID; r < nc; r += NT) dg_lds[L.l + r] = l ? l[b * nc + r] : 0.0;
    __syncthreads();
    dev_evaluate(c, LP(L.u), 0.0, nullptr, true);
    if (q) for (int i = TID; i < n; i += NT) q[b * n + i] = dg_lds[L.q + i];
    if (g) for (int r = TID; r < nc; r += NT) g[b * nc + r] = dg_lds[L.g + r];
    if (x) for (int i = TID; i < (dg_prob.N + 1) * dg_prob.nq; i += NT) x[b * (int64_t)(dg_prob.N + 1) * dg_prob.nq + i] = dg_lds[L.e_x + i];
    if (G)
      for (int64_t t = TID; t < (int64_t)nc * n; t += NT)
        G[b * (int64_t)nc * n + t] = dg_prob.gd_global ? g_row_coef<cgptr>(dg_prob, dev_gd_global(c), (int)(t / n), (int)(t % n))
                                                          : g_row_coef<clptr>(dg_prob, LP(L.gd), (int)(t / n), (int)(t % n));
    if (Q) {
      cgptr Qg = c.ws + dg_prob.ws_q;
      for (int t = TID; t < n * n; t += NT) Q[b * (int64_t)n * n + t] = Qg[t];
    }
    if (l0) {
      dev_dual_init(c);
      for (int r = TID; r < nc; r += NT) l0[b * nc + r] = dg_lds[L.l + r];
    }
    __syncthreads();
  }
}

// Test hook: _solve_qp at the linearisation point (u, l).
__global__ void __launch_bounds__(DG_BLOCK, 2)
dg_qp_kernel(const DgProb* __restrict__ D, int64_t B, const double* __restrict__ x0, const double* __restrict__ u,
             const double* __restrict__ l, double* du, double* lhat, double* Qpd, int32_t* flag, double* info8, double* __restrict__ ws_all) {
  Ctx c;
  c.coop = nullptr; c.coop_payload = nullptr; c.coop_total = 0; c.coop_start = 0; c.coop_verify = 0; c.coop_window = 0; c.coop_helpers = 0;
  c.ws = (gptr)ws_all + (int64_t)blockIdx.x * dg_prob.ws_doubles;
  c.trace = nullptr; c.trace_cap = 0; c.itlog = nullptr; c.itlog_cap = 0;
  const DgLds& L = dg_prob.L;
  const int n = dg_prob.n, nc = dg_prob.nc;
  dev_load_tables();
  // The saved active set is deliberately NOT cleared between the scenarios one workgroup handles: with more scenarios
  // than workgroups every QP after the first is warm-started from an unrelated problem's active set (tests use this).
  if (TID == 0) { dg_lds[L.scal + DG_QP_NPREV] = 0.0; dg_lds[L.scal + DG_PSD_PD] = 0.0; }
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    c.x0 = (cgptr)x0 + b * dg_prob.nq;
    __syncthreads();
    if (TID == 0) { dg_lds[L.scal + DG_XVALID] = 0.0; dg_lds[L.scal + DG_REG] = dg_prob.par.reg; dg_lds[L.scal + DG_OSQP_RHO] = 0.1; }
    for (int i = TID; i < n; i += NT) dg_lds[L.u + i] = u[b * n + i];
    for (int r = TID; r < nc; r += NT) dg_lds[L.l + r] = l[b * nc + r];
    __syncthreads();
    const int f = dev_linearize_and_qp(c, true, nullptr, Qpd ? (gptr)Qpd + b * (int64_t)n * n : nullptr);
    if (du) for (int i = TID; i < n; i += NT) du[b * n + i] = dg_lds[L.o_du + i];
    if (lhat) for (int r = TID; r < nc; r += NT) lhat[b * nc + r] = dg_lds[L.o_lhat + r];
    if (flag && TID == 0) flag[b] = f;
    if (info8 && TID < 8) info8[b * 8 + TID] = dg_prob.osqp ? dg_lds[L.scal + DG_OSQP_INFO + TID] : 0.0;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Overlapping launches on several handles (bench.py --pipeline) need one hardware queue per stream; the HIP runtime creates
// 4 by default and reads this variable when it initialises (first HIP call), so set it when the library is loaded unless
// the user already chose a value.
namespace {
struct DgEnvInit {
  DgEnvInit() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} dg_env_init;
}  // namespace

struct dgsqp_comm_state;
struct dgsqp_solver {
  int device = 0;
  DgProb hp;
  DgProb* dp = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int num_cu = 0, wg_per_cu = 1, max_grid = 0, launched_grid = 0;
  size_t lds_bytes = 0;
  double* ws = nullptr;
  size_t ws_groups = 0;
  unsigned long long* ticket = nullptr;
  unsigned int* drained_host = nullptr;   // mapped host memory: 1 once the last launch has handed out its last scenario
  unsigned int* drained_dev = nullptr;
  // staged batch
  int64_t cap = 0, B = 0;
  double *d_x0 = nullptr, *d_uws = nullptr, *d_u = nullptr, *d_l = nullptr, *d_x = nullptr, *d_cond = nullptr, *d_cost = nullptr;
  int32_t *d_status = nullptr, *d_iters = nullptr, *d_qps = nullptr;
  double* d_trace = nullptr; int trace_cap = 0; int64_t trace_B = 0;    // trace_B: scenarios the buffer holds
  int64_t trace_launch_B = 0;                                             // scenarios of the launch that filled it
  double* d_itlog = nullptr; int itlog_cap = 0; int64_t itlog_B = 0, itlog_launch_B = 0;
  bool in_flight = false;       // a solve launch has been enqueued and not yet waited for
  dgsqp_solver* group_leader = nullptr;   // set while this handle's batch is being solved by another handle's grouped launch
  DgBatch* d_group = nullptr;             // leader: device table of the group's batches (DG_GROUP_MAX entries)
  DgBatch* group_host = nullptr;          // ... and its pinned host image (the source of an asynchronous copy)
  unsigned long long launch_gen = 0;      // leader: counts its launches; members remember the generation they belong to
  unsigned long long group_gen = 0;       // member: launch_gen of the leader's launch that solves this handle's batch
  float last_ms = 0.0f;                   // kernel time of the last completed launch that solved this handle's batch (HIP events)
  DgCoop* d_coop = nullptr;           // cooperative line search: job slots (2 per workgroup) ...
  double* d_coop_payload = nullptr;   // ... and the base points their owners publish
  size_t coop_bytes = 0;
  int coop_mode = 1;                  // 0 off, 1 synchronous calls only (nothing else is waiting for the compute units), 2 every launch
  bool coop_next_sync = false;        // (set by the synchronous entry points around their launch)
  size_t park_last_cap = 0;           // deferral of long scenarios: slots the handle's last launch could use (0: it did not defer)
  int defer_min_it = 8;               // 0: off
  double defer_factor = 2.0;
  bool defer_requested = false;       // dgsqp_set_deferral was called (DG-SQP v2 is only deferred on request)
  dgsqp_comm_state* comm = nullptr;   // RCCL communicator + record buffers (dgsqp_comm.h), owned by the handle
  std::string err;
};
static thread_local std::string g_create_err;

#define HIPCHK(h, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return DGSQP_E_DEVICE;                                                                 \
    }                                                                                        \
  } while (0)

#include "dgsqp_comm.h"

static void free_batch(dgsqp_solver* h) {
  void* ptrs[] = {h->d_x0, h->d_uws, h->d_u, h->d_l, h->d_x, h->d_cond, h->d_cost, h->d_status, h->d_iters, h->d_qps};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  h->d_x0 = h->d_uws = h->d_u = h->d_l = h->d_x = h->d_cond = h->d_cost = nullptr;
  h->d_status = h->d_iters = h->d_qps = nullptr;
  h->cap = 0;
}
static int ensure_batch(dgsqp_solver* h, int64_t B) {
  if (B <= h->cap) return DGSQP_OK;
  free_batch(h);
  const DgProb& D = h->hp;
  HIPCHK(h, hipMalloc(&h->d_x0, sizeof(double) * B * D.nq));
  HIPCHK(h, hipMalloc(&h->d_uws, sizeof(double) * B * D.n));
  HIPCHK(h, hipMalloc(&h->d_u, sizeof(double) * B * D.n));
  HIPCHK(h, hipMalloc(&h->d_l, sizeof(double) * B * D.nc));
  HIPCHK(h, hipMalloc(&h->d_x, sizeof(double) * B * (D.N + 1) * D.nq));
  HIPCHK(h, hipMalloc(&h->d_cond, sizeof(double) * B * 3));
  HIPCHK(h, hipMalloc(&h->d_cost, sizeof(double) * B * D.M));
  HIPCHK(h, hipMalloc(&h->d_status, sizeof(int32_t) * B));
  HIPCHK(h, hipMalloc(&h->d_iters, sizeof(int32_t) * B));
  HIPCHK(h, hipMalloc(&h->d_qps, sizeof(int32_t) * B));
  h->cap = B;
  return DGSQP_OK;
}
static int ensure_ws(dgsqp_solver* h, size_t groups) {
  if (groups <= h->ws_groups) return DGSQP_OK;
  if (h->ws) (void)hipFree(h->ws);
  h->ws = nullptr; h->ws_groups = 0;
  HIPCHK(h, hipMalloc(&h->ws, sizeof(double) * groups * (size_t)h->hp.ws_doubles));
  h->ws_groups = groups;
  return DGSQP_OK;
}
// The kernels read the game from the __constant__ symbol dg_prob, one per device and process.  Handles of DIFFERENT games
// (or parameters) may coexist: the registry below remembers which description the symbol of each device holds and which
// handles have a launch in flight.  A launch whose description differs from the resident one first waits for every launch
// in flight on that device (they would otherwise read the new constants), then uploads its own; launches of the same
// description skip the upload and overlap freely (bench.py --pipeline).
namespace {
std::mutex g_reg_mutex;
struct DgResident { bool valid = false; std::vector<unsigned char> bytes; };
DgResident g_resident[64];
std::vector<dgsqp_solver*> g_handles;
// Deferral of long scenarios: ONE pool of slots per device, taken by the cooperative launch that defers (such launches run when
// nothing else waits for the compute units; a second one that finds the pool busy simply does not defer).  Allocated at the first
// deferring launch on the device -- 4,096 slots or what 16 GB hold -- and kept: no launch pays for an allocation of its own.
struct DgParkPool {
  DgParkEntry* entries = nullptr;
  double* store = nullptr;
  size_t slots = 0, slot_doubles = 0;
  dgsqp_solver* owner = nullptr;            // handle whose launch uses the pool ...
  unsigned long long owner_gen = 0;         // ... and which of its launches
};
DgParkPool g_park[64];
}  // namespace
// the stream the handle's solve in flight runs on: its own, or the leader's for a member of a grouped launch
static hipStream_t active_stream(const dgsqp_solver* h) { return h->group_leader ? h->group_leader->stream : h->stream; }
// kernel time of h's completed launch (events ev[0], ev[1] of the stream it ran on)
static void record_last_ms(dgsqp_solver* h, dgsqp_solver* leader) {
  float ms = 0.0f;
  if (h->launched_grid > 0 && hipEventElapsedTime(&ms, leader->ev[0], leader->ev[1]) == hipSuccess) h->last_ms = ms;
}
// The members of a grouped launch led by L whose kernel has completed (L's stream is synchronised): they leave the group
// with the kernel time of THAT launch.  Called before L's events are re-recorded by its next launch, so that a member's
// later dgsqp_wait / dgsqp_finished never looks at the events of an unrelated kernel.
static void release_members(dgsqp_solver* L) {
  for (dgsqp_solver* o : g_handles)
    if (o->group_leader == L && o->group_gen == L->launch_gen) { record_last_ms(o, L); o->in_flight = false; o->group_leader = nullptr; }
}
static int wait_idle(dgsqp_solver* h) {
  if (h->in_flight) {
    dgsqp_solver* L = h->group_leader ? h->group_leader : h;
    HIPCHK(h, hipStreamSynchronize(L->stream));
    record_last_ms(h, L);
    h->in_flight = false;
    h->group_leader = nullptr;
  }
  return DGSQP_OK;
}
// Call with g_reg_mutex held and keep it until the kernel that needs the constants has been enqueued and the handle is
// marked in flight: from then on a launch of a different game waits for that kernel before it overwrites the symbol.
static int upload_problem(dgsqp_solver* h) {
  DgResident& r = g_resident[h->device & 63];
  if (r.valid && r.bytes.size() == sizeof(DgProb) && memcmp(r.bytes.data(), &h->hp, sizeof(DgProb)) == 0) return DGSQP_OK;
  for (dgsqp_solver* o : g_handles)
    if (o->device == h->device && o->in_flight) {
      if (hipStreamSynchronize(active_stream(o)) != hipSuccess) { h->err = "hipStreamSynchronize of a launch in flight failed"; return DGSQP_E_DEVICE; }
      // (o stays marked in flight: its owner still has to collect it with dgsqp_wait)
    }
  HIPCHK(h, hipMemcpyToSymbolAsync(HIP_SYMBOL(dg_prob), &h->hp, sizeof(DgProb), 0, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  r.bytes.assign((const unsigned char*)&h->hp, (const unsigned char*)&h->hp + sizeof(DgProb));
  r.valid = true;
  return DGSQP_OK;
}
// Cooperative line search for the launch about to be enqueued?  Helpers keep their compute units until the launch's last
// scenario is done: right when nothing else waits for them (synchronous calls), wrong in a pipeline of launches -- the caller
// says so (dgsqp_set_cooperative).  Needs the whole grid resident (it is: at most one workgroup per compute unit).
// (development knobs: DGSQP_COOP_START = rejected trials after which a line search is offered to helpers, default 4;
//  DGSQP_COOP_VERIFY = 1: owners re-evaluate every helper value and count differing bits -- dgsqp_coop_stats)
static int coop_start_trials() { const char* e = getenv("DGSQP_COOP_START"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }
static int coop_window_trials() { const char* e = getenv("DGSQP_COOP_WINDOW"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v; }
static int coop_max_helpers() { const char* e = getenv("DGSQP_COOP_HELPERS"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v; }
static int coop_verify_mode() { const char* e = getenv("DGSQP_COOP_VERIFY"); return e && atoi(e) != 0; }
static bool coop_for_launch(dgsqp_solver* h, int grid) {
  if (!h->d_coop || grid > h->num_cu * 2 + 2 || h->trace_cap > 0) return false;
  // (development knob DGSQP_COOP_MIN_CHAIN: only for games whose rollout is a dependent chain of at least that many evaluations of f_c,
  // N x substeps x stages.  With all idle workgroups helping, the euler games lost -- 409 -> 450 ms --; with the helpers capped at 64
  // they gain as well -- 362 -> 345 ms --, so the default is 0.)
  const dgsqp_problem_t& P = h->hp.P;
  const int stages = P.integrator == DGSQP_INT_RK4 ? 4 : (P.integrator == DGSQP_INT_RK3 ? 3 : (P.integrator == DGSQP_INT_RK2 ? 2 : 1));
  const int chain = P.N * (P.integrator == DGSQP_INT_EULER ? 1 : P.substeps * stages);
  { const char* e = getenv("DGSQP_COOP_MIN_CHAIN"); if (chain < (e ? atoi(e) : 0)) return false; }
  return h->coop_mode == 2 || (h->coop_mode == 1 && h->coop_next_sync);
}
// Deferral of long scenarios for the cooperative launch about to be enqueued on h's stream (DgPark, dgsqp_device.h): a quarter of
// the launch's scenarios may be deferred at a time (bounded by 16 GB of slots).  Off for launches that give every scenario its own
// workgroup and while logs are recorded; DG-SQP v2 only after an explicit dgsqp_set_deferral (round 4).  (development knobs: DGSQP_DEFER = 0 switches it off, DGSQP_DEFER_MIN_IT,
// DGSQP_DEFER_FACTOR override dgsqp_set_deferral.)
static int park_for_launch(dgsqp_solver* h, bool coop, int grid, int64_t total, DgPark* out) {
  memset(out, 0, sizeof(*out));
  h->park_last_cap = 0;
  int min_it = h->defer_min_it;
  double factor = h->defer_factor;
  // qp_method OSQP on the LDS path: what a scenario costs is set by its ADMM iterations, not by its SQP iterations -- "twice the mean
  // iteration count" sets aside scenarios that are not long, and their state copies and late resumes cost more than the tail they
  // save (20 batches of configs[1] in one launch, profiles/r06_osqp_deferral_sweep.txt: 4,410 scen/s at factor 2, 4,914 without deferral,
  // 5,337 at factor 4, 5,108 at 6; the exact QP has its optimum at 2: 11,739 against 11,105 at 3 and 10,129 without)
  if (h->hp.osqp && h->hp.big != 2 && !h->defer_requested) factor = 4.0;
  int time_mode = 0;
  { const char* e = getenv("DGSQP_DEFER_TIME"); if (e) time_mode = atoi(e) != 0; }
  { const char* e = getenv("DGSQP_DEFER"); if (e && atoi(e) == 0) min_it = 0; }
  { const char* e = getenv("DGSQP_DEFER_MIN_IT"); if (e) min_it = atoi(e); }
  { const char* e = getenv("DGSQP_DEFER_FACTOR"); if (e) factor = atof(e); }
  if (!coop || min_it <= 0 || total <= (int64_t)grid || h->trace_cap > 0 || h->itlog_cap > 0) return DGSQP_OK;
  // DG-SQP v2: only when the caller asked for it (dgsqp_set_deferral).  Nearly every v2 scenario runs ~375 iterations and a few run
  // thousands: setting those aside delays exactly the solves that decide the launch's length (48 batches of 512 as 8 x 3 launches:
  // 409 scen/s without, 367 with; as one launch 404 / 411).
  if (h->hp.par.variant == DGSQP_VARIANT_V2 && !h->defer_requested) return DGSQP_OK;
  const size_t slot = (size_t)h->hp.L.total + (size_t)h->hp.ws_doubles;
  double frac = 0.25;
  { const char* e = getenv("DGSQP_DEFER_CAP_FRAC"); if (e) frac = atof(e); }
  size_t cap = (size_t)((double)total * (frac > 0.0 && frac <= 1.0 ? frac : 0.25) + 1.0);
  DgParkPool& pool = g_park[h->device & 63];      // (g_reg_mutex is held by the launch functions)
  if (pool.owner && pool.owner != h && pool.owner->in_flight && pool.owner->launch_gen == pool.owner_gen) return DGSQP_OK;   // busy: no deferral
  // The pool holds what this launch may defer (a quarter of its scenarios) and grows geometrically when a larger launch comes along; its
  // size is bounded by DGSQP_DEFER_POOL_BYTES (default 16 GiB; 0 switches deferral off).  It is only ever re-allocated while no launch uses it.
  size_t limit_bytes = (size_t)16 << 30;
  { const char* e = getenv("DGSQP_DEFER_POOL_BYTES"); if (e) limit_bytes = (size_t)strtoull(e, nullptr, 10); }
  const size_t max_slots = limit_bytes / (slot * sizeof(double));
  if (cap > max_slots) cap = max_slots;
  if (cap < 1) return DGSQP_OK;
  if (pool.slots < cap || pool.slot_doubles < slot) {
    size_t slots = pool.slot_doubles == slot ? 2 * pool.slots : 0;
    if (slots < cap) slots = cap;
    if (slots < 64) slots = 64;
    if (slots > max_slots) slots = max_slots;
    if (pool.entries) (void)hipFree(pool.entries);
    if (pool.store) (void)hipFree(pool.store);
    pool = DgParkPool();
    HIPCHK(h, hipMalloc((void**)&pool.entries, sizeof(DgParkEntry) * slots));
    if (hipMalloc((void**)&pool.store, sizeof(double) * slot * slots) != hipSuccess) {      // no room: solve without deferral
      (void)hipGetLastError();
      (void)hipFree(pool.entries); pool.entries = nullptr;
      return DGSQP_OK;
    }
    pool.slots = slots; pool.slot_doubles = slot;
  }
  if (cap > pool.slots) cap = pool.slots;
  HIPCHK(h, hipMemsetAsync(pool.entries, 0, sizeof(DgParkEntry) * cap, h->stream));
  pool.owner = h; pool.owner_gen = h->launch_gen + 1;      // (the launch about to be enqueued)
  h->park_last_cap = cap;
  out->entries = pool.entries; out->store = pool.store; out->cap = (unsigned int)cap;
  out->min_it = min_it; out->factor_x16 = (int)(factor * 16.0 + 0.5); out->slot_doubles = slot;
  out->time_mode = time_mode;
  return DGSQP_OK;
}
static int grid_for(dgsqp_solver* h, int64_t B) {
  int64_t g = (int64_t)h->max_grid;
  if (B < g) g = B;
  if (g < 1) g = 1;
  return (int)g;
}

// helpers for the two test hooks: temporary device buffers
struct TmpBuf {
  std::vector<void*> ptrs;
  ~TmpBuf() { for (void* p : ptrs) (void)hipFree(p); }
  template <class T> T* alloc(size_t count) { void* p = nullptr; if (hipMalloc(&p, sizeof(T) * (count ? count : 1)) != hipSuccess) return nullptr; ptrs.push_back(p); return (T*)p; }
};

// one mapping from dg_build()'s message to the ABI's error code, shared by dgsqp_create and dgsqp_plan
static int build_error_code(const std::string& msg) {
  const bool too_large = msg.find("LDS") != std::string::npos || msg.find("too many") != std::string::npos || msg.find("not supported yet") != std::string::npos;
  return too_large ? DGSQP_E_TOO_LARGE : DGSQP_E_ARG;
}

extern "C" {

int dgsqp_backend_info(char* buf, int buflen) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    snprintf(buf, buflen, "no HIP device");
    return DGSQP_E_DEVICE;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return DGSQP_E_DEVICE;
  snprintf(buf, buflen, "%s arch=%s CUs=%d LDS/WG=%zu devices=%d", p.name, p.gcnArchName, p.multiProcessorCount, (size_t)p.sharedMemPerBlock, ndev);
  return DGSQP_OK;
}

int dgsqp_create(const dgsqp_problem_t* prob, const dgsqp_params_t* par, int device, dgsqp_handle_t* out) {
  if (!prob || !par || !out) { g_create_err = "null argument"; return DGSQP_E_ARG; }
  dgsqp_solver* h = new dgsqp_solver();
  std::string msg = dg_build(*prob, *par, h->hp);
  if (!msg.empty()) {
    g_create_err = msg;
    delete h;
    return build_error_code(msg);
  }
  if (DG_BLOCK != 512 && (h->hp.big == 2 || h->hp.classic_qp || h->hp.osqp)) {
    // the -DDG_BLOCK=256 build (two workgroups per CU, row N1): explicit-inverse layouts (n <= 128) with the active-set QP only
    g_create_err = "too large: this build (DG_BLOCK = 256, two workgroups per CU) holds the LDS-resident and big explicit-inverse layouts with the active-set QP only";
    delete h;
    return DGSQP_E_TOO_LARGE;
  }
  auto fail = [&](const std::string& m) { g_create_err = m; dgsqp_destroy(h); return DGSQP_E_DEVICE; };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  h->device = device;
  if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return fail("hipGetDeviceProperties failed");
  h->num_cu = p.multiProcessorCount;
  h->lds_bytes = (size_t)h->hp.L.total * sizeof(double);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
  for (auto& e : h->ev) if (hipEventCreate(&e) != hipSuccess) return fail("hipEventCreate failed");
  if (hipMalloc(&h->dp, sizeof(DgProb)) != hipSuccess) return fail("hipMalloc(problem) failed");
  if (hipMemcpy(h->dp, &h->hp, sizeof(DgProb), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy(problem) failed");
  if (hipMalloc(&h->ticket, sizeof(unsigned long long)) != hipSuccess) return fail("hipMalloc(ticket) failed");
  if (hipHostMalloc((void**)&h->drained_host, sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail("hipHostMalloc(flag) failed");
  *h->drained_host = 1u;
  if (hipHostGetDevicePointer((void**)&h->drained_dev, h->drained_host, 0) != hipSuccess) return fail("hipHostGetDevicePointer failed");
  h->coop_bytes = sizeof(DgCoop) + sizeof(DgCoopJob) * 2 * (size_t)(h->num_cu * 2 + 2);
  if (hipMalloc((void**)&h->d_coop, h->coop_bytes) != hipSuccess) return fail("hipMalloc(coop) failed");
  if (hipMalloc((void**)&h->d_coop_payload, sizeof(double) * 2 * (2 * (size_t)h->hp.n + 2 * (size_t)h->hp.nc) * (size_t)(h->num_cu * 2 + 2)) != hipSuccess) return fail("hipMalloc(coop payload) failed");
  const void* kernels[] = {(const void*)dg_solve_kernel, (const void*)dg_evaluate_kernel, (const void*)dg_qp_kernel};
  for (const void* k : kernels) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
    if (e != hipSuccess) return fail(std::string("hipFuncSetAttribute(dynamic LDS): ") + hipGetErrorString(e));
  }
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dg_solve_kernel, DG_BLOCK, h->lds_bytes) != hipSuccess || occ < 1) occ = 1;
  h->wg_per_cu = occ;
  h->max_grid = h->num_cu * occ;
  { std::lock_guard<std::mutex> lk(g_reg_mutex); g_handles.push_back(h); }
  *out = h;
  return DGSQP_OK;
}

void dgsqp_destroy(dgsqp_handle_t h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream && h->in_flight) (void)hipStreamSynchronize(active_stream(h));
  (void)hipStreamSynchronize(h->stream);
  { std::lock_guard<std::mutex> lk(g_reg_mutex); for (dgsqp_solver* o : g_handles) if (o->group_leader == h) { o->group_leader = nullptr; o->in_flight = false; } }
  if (h->d_group) (void)hipFree(h->d_group);
  if (h->group_host) (void)hipHostFree(h->group_host);
  if (h->comm) (void)dgsqp_comm_destroy(h);
  { std::lock_guard<std::mutex> lk(g_reg_mutex); g_handles.erase(std::remove(g_handles.begin(), g_handles.end(), h), g_handles.end()); }
  free_batch(h);
  if (h->ws) (void)hipFree(h->ws);
  if (h->dp) (void)hipFree(h->dp);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->d_coop) (void)hipFree(h->d_coop);
  if (h->d_coop_payload) (void)hipFree(h->d_coop_payload);
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    DgParkPool& pool = g_park[h->device & 63];
    if (pool.owner == h) pool.owner = nullptr;
    if (g_handles.empty()) {          // last handle of the process: the pools go as well
      for (DgParkPool& pl : g_park) {
        if (pl.entries || pl.store) {
          // (the pool's memory belongs to the device it was allocated on)
          if (pl.entries) (void)hipFree(pl.entries);
          if (pl.store) (void)hipFree(pl.store);
          pl = DgParkPool();
        }
      }
    }
  }
  if (h->drained_host) (void)hipHostFree(h->drained_host);
  if (h->d_trace) (void)hipFree(h->d_trace);
  if (h->d_itlog) (void)hipFree(h->d_itlog);
  for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int dgsqp_dims(dgsqp_handle_t h, dgsqp_dims_t* out) {
  if (!h || !out) return DGSQP_E_ARG;
  const DgProb& D = h->hp;
  out->M = D.M; out->N = D.N; out->n_q = D.nq; out->n_u = D.nu; out->n = D.n; out->n_c = D.nc;
  out->n_dense = D.ndense; out->lds_bytes = (int32_t)h->lds_bytes; out->workspace_bytes = D.ws_doubles * (int64_t)sizeof(double);
  out->layout = D.big; out->reserved_ = 0;
  return DGSQP_OK;
}

int dgsqp_plan(const dgsqp_problem_t* prob, const dgsqp_params_t* par, dgsqp_dims_t* out, char* msg, int msglen) {
  if (!prob || !par || !out) return DGSQP_E_ARG;
  static DgProb D;     // ~100 KB: not on the stack (single-threaded helper, like dgsqp_create)
  const std::string err = dg_build(*prob, *par, D);
  if (msg && msglen > 0) snprintf(msg, msglen, "%s", err.c_str());
  memset(out, 0, sizeof(*out));
  out->M = D.M; out->N = D.N; out->n_q = D.nq; out->n_u = D.nu; out->n = D.n; out->n_c = D.nc;
  out->n_dense = D.ndense; out->lds_bytes = D.L.total * 8; out->workspace_bytes = D.ws_doubles * (int64_t)sizeof(double);
  out->layout = D.big;
  if (!err.empty()) return build_error_code(err);
  return DGSQP_OK;
}

const char* dgsqp_last_error(dgsqp_handle_t h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int dgsqp_stage_inputs(dgsqp_handle_t h, int64_t B, const double* x0, const double* u_ws) {
  if (!h || B < 0 || (B > 0 && (!x0 || !u_ws))) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  int rc = wait_idle(h);            // the buffers below belong to the launch in flight, if any
  if (rc) return rc;
  h->B = B;
  if (B == 0) return DGSQP_OK;
  rc = ensure_batch(h, B);
  if (rc) return rc;
  rc = ensure_ws(h, (size_t)grid_for(h, B));
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->d_x0, x0, sizeof(double) * B * h->hp.nq, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_uws, u_ws, sizeof(double) * B * h->hp.n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DGSQP_OK;
}

int dgsqp_launch_staged(dgsqp_handle_t h) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcw = wait_idle(h); if (rcw) return rcw; }     // one launch in flight per handle
  h->launched_grid = 0;
  if (h->B == 0) return DGSQP_OK;
  const int grid = grid_for(h, h->B);
  SolveOutPtrs O{h->d_u, h->d_l, h->d_x, h->d_cond, h->d_cost, h->d_status, h->d_iters, h->d_qps};
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "hipStreamSynchronize failed"; return DGSQP_E_DEVICE; }
  release_members(h);        // (members of an earlier grouped launch led by h: that kernel is done, its events are about to be reused)
  { int rcu = upload_problem(h); if (rcu) return rcu; }
  double* trace = nullptr;
  if (h->trace_cap > 0) {
    if (h->trace_B < h->B) {
      if (h->d_trace) (void)hipFree(h->d_trace);
      h->d_trace = nullptr; h->trace_B = 0;
      HIPCHK(h, hipMalloc(&h->d_trace, sizeof(double) * h->B * (1 + 2 * (size_t)h->trace_cap)));
      h->trace_B = h->B;
    }
    trace = h->d_trace;
    h->trace_launch_B = h->B;
  }
  double* itlog = nullptr;
  if (h->itlog_cap > 0) {
    const size_t per = 1 + (size_t)h->itlog_cap * (h->hp.n + h->hp.nc);
    if (h->itlog_B < h->B) {
      if (h->d_itlog) (void)hipFree(h->d_itlog);
      h->d_itlog = nullptr; h->itlog_B = 0;
      HIPCHK(h, hipMalloc(&h->d_itlog, sizeof(double) * h->B * per));
      h->itlog_B = h->B;
    }
    itlog = h->d_itlog;
    h->itlog_launch_B = h->B;
  }
  HIPCHK(h, hipMemsetAsync(h->ticket, 0, sizeof(unsigned long long), h->stream));
  const bool coop = coop_for_launch(h, grid);
  if (coop) HIPCHK(h, hipMemsetAsync(h->d_coop, 0, h->coop_bytes, h->stream));
  DgPark park;
  { const int rcp = park_for_launch(h, coop, grid, h->B, &park); if (rcp) return rcp; }
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  *h->drained_host = 0u;
  hipLaunchKernelGGL(dg_solve_kernel, dim3(grid), dim3(DG_BLOCK), h->lds_bytes, h->stream, h->dp, h->B, h->d_x0, h->d_uws, O, h->ws, h->ticket, trace, h->trace_cap, h->drained_dev, itlog, h->itlog_cap, (const DgBatch*)nullptr, 0,
                     coop ? h->d_coop : (DgCoop*)nullptr, h->d_coop_payload, coop_start_trials(), coop_verify_mode(), coop_window_trials(), coop_max_helpers(), park);
  HIPCHK(h, hipGetLastError());
  h->launch_gen++;
  h->launched_grid = grid;
  h->in_flight = true;       // (only now: an error return above leaves the handle idle)
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  return DGSQP_OK;
}

int dgsqp_launch_staged_group(const dgsqp_handle_t* hs, int count) {
  if (!hs || count < 1 || count > DG_GROUP_MAX || !hs[0]) return DGSQP_E_ARG;
  dgsqp_solver* L = hs[0];
  if (count == 1) return dgsqp_launch_staged(L);
  HIPCHK(L, hipSetDevice(L->device));
  for (int i = 0; i < count; i++) {
    dgsqp_solver* h = hs[i];
    if (!h) { L->err = "null handle in the group"; return DGSQP_E_ARG; }
    for (int j = 0; j < i; j++) if (hs[j] == h) { L->err = "a handle appears twice in the group"; return DGSQP_E_ARG; }
    if (h->device != L->device || h->B != L->B || L->B <= 0 || memcmp(&h->hp, &L->hp, sizeof(DgProb)) != 0) {
      L->err = "grouped launch: every handle must hold a staged batch of the same size, of the same game, on the same device"; return DGSQP_E_ARG;
    }
    if (h->trace_cap > 0 || h->itlog_cap > 0) { L->err = "grouped launch: event / iterate logs are per single launch"; return DGSQP_E_ARG; }
    const int rcw = wait_idle(h);
    if (rcw) return rcw;
  }
  const int grid = grid_for(L, L->B * count);
  { const int rce = ensure_ws(L, (size_t)grid); if (rce) return rce; }
  if (!L->d_group) HIPCHK(L, hipMalloc(&L->d_group, sizeof(DgBatch) * DG_GROUP_MAX));
  if (!L->group_host) HIPCHK(L, hipHostMalloc((void**)&L->group_host, sizeof(DgBatch) * DG_GROUP_MAX, hipHostMallocDefault));
  for (int i = 0; i < count; i++) {
    dgsqp_solver* h = hs[i];
    L->group_host[i] = DgBatch{h->d_x0, h->d_uws, SolveOutPtrs{h->d_u, h->d_l, h->d_x, h->d_cond, h->d_cost, h->d_status, h->d_iters, h->d_qps}};
  }
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  if (hipStreamSynchronize(L->stream) != hipSuccess) { L->err = "hipStreamSynchronize failed"; return DGSQP_E_DEVICE; }
  release_members(L);
  { int rcu = upload_problem(L); if (rcu) return rcu; }
  HIPCHK(L, hipMemcpyAsync(L->d_group, L->group_host, sizeof(DgBatch) * count, hipMemcpyHostToDevice, L->stream));
  HIPCHK(L, hipMemsetAsync(L->ticket, 0, sizeof(unsigned long long), L->stream));
  const bool coop = coop_for_launch(L, grid);
  if (coop) HIPCHK(L, hipMemsetAsync(L->d_coop, 0, L->coop_bytes, L->stream));
  DgPark park;
  { const int rcp = park_for_launch(L, coop, grid, L->B * count, &park); if (rcp) return rcp; }
  HIPCHK(L, hipEventRecord(L->ev[0], L->stream));
  *L->drained_host = 0u;
  SolveOutPtrs O0 = L->group_host[0].O;
  hipLaunchKernelGGL(dg_solve_kernel, dim3(grid), dim3(DG_BLOCK), L->lds_bytes, L->stream, L->dp, L->B, L->d_x0, L->d_uws, O0, L->ws, L->ticket,
                     (double*)nullptr, 0, L->drained_dev, (double*)nullptr, 0, (const DgBatch*)L->d_group, count,
                     coop ? L->d_coop : (DgCoop*)nullptr, L->d_coop_payload, coop_start_trials(), coop_verify_mode(), coop_window_trials(), coop_max_helpers(), park);
  HIPCHK(L, hipGetLastError());
  L->launch_gen++;
  for (int i = 0; i < count; i++) { hs[i]->launched_grid = grid; hs[i]->in_flight = true; hs[i]->group_leader = i == 0 ? nullptr : L; hs[i]->group_gen = L->launch_gen; }
  HIPCHK(L, hipEventRecord(L->ev[1], L->stream));
  return DGSQP_OK;
}

int dgsqp_draining(dgsqp_handle_t h) {
  if (!h) return 1;
  const dgsqp_solver* L = h->group_leader ? h->group_leader : h;
  return h->launched_grid == 0 || __atomic_load_n(L->drained_host, __ATOMIC_RELAXED) != 0u;
}

int dgsqp_finished(dgsqp_handle_t h) {
  if (!h || !h->in_flight || h->launched_grid == 0) return 1;
  if (hipSetDevice(h->device) != hipSuccess) return 1;
  return hipEventQuery((h->group_leader ? h->group_leader : h)->ev[1]) != hipErrorNotReady;      // (a member whose leader has moved on was released: not in flight)
}

int dgsqp_wait(dgsqp_handle_t h, dgsqp_timing_t* tm) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  if (tm) memset(tm, 0, sizeof(*tm));
  { const int rc = wait_idle(h); if (rc) return rc; }           // (a member of a grouped launch waits for, and reports, the group's kernel)
  if (tm && h->launched_grid > 0) { tm->kernel_ms = h->last_ms; tm->total_ms = h->last_ms; tm->grid = h->launched_grid; tm->block = DG_BLOCK; }
  return DGSQP_OK;
}

int dgsqp_solve_staged(dgsqp_handle_t h, dgsqp_timing_t* tm) {
  if (!h) return DGSQP_E_ARG;
  h->coop_next_sync = true;          // the caller waits for this launch: idle workgroups help with its line searches
  const int rc = dgsqp_launch_staged(h);
  h->coop_next_sync = false;
  if (rc != DGSQP_OK) return rc;
  return dgsqp_wait(h, tm);
}

int dgsqp_set_cooperative(dgsqp_handle_t h, int mode) {
  if (!h || mode < 0 || mode > 2) return DGSQP_E_ARG;
  h->coop_mode = mode;
  return DGSQP_OK;
}

int dgsqp_set_deferral(dgsqp_handle_t h, int min_iters, double factor) {
  if (!h || min_iters < 0 || !(factor >= 0.0) || factor > 1000.0) return DGSQP_E_ARG;
  h->defer_min_it = min_iters;
  h->defer_factor = factor;
  h->defer_requested = true;
  return DGSQP_OK;
}

int dgsqp_reserve_deferral(dgsqp_handle_t h, int64_t scenarios) {
  if (!h || scenarios < 1) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  { const int rcw = wait_idle(h); if (rcw) return rcw; }
  DgParkPool& pool = g_park[h->device & 63];
  dgsqp_solver* const owner = pool.owner;
  const unsigned long long owner_gen = pool.owner_gen;
  DgPark unused;
  const int rc = park_for_launch(h, true, h->max_grid, scenarios, &unused);      // (sizes the device's pool exactly as that launch would)
  h->park_last_cap = 0;
  // no launch follows: the pool must not look taken by this handle's NEXT launch (a plain one would make other handles' cooperative
  // launches find it "busy" and run without deferral).  A re-allocation reset the owner anyway; otherwise put back what was there.
  if (g_park[h->device & 63].owner == h) { g_park[h->device & 63].owner = owner == h ? nullptr : owner; g_park[h->device & 63].owner_gen = owner == h ? 0 : owner_gen; }
  return rc;
}

int dgsqp_deferral_stats(dgsqp_handle_t h, uint64_t* out2) {
  if (!h || !out2 || !h->d_coop) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc = wait_idle(h); if (rc) return rc; }
  DgCoop hdr;
  HIPCHK(h, hipMemcpy(&hdr, h->d_coop, sizeof(DgCoop) - sizeof(DgCoopJob), hipMemcpyDeviceToHost));
  out2[0] = hdr.park_pushed < h->park_last_cap ? hdr.park_pushed : (uint64_t)h->park_last_cap;
  out2[1] = hdr.park_resumed;
  return DGSQP_OK;
}

int dgsqp_deferral_log(dgsqp_handle_t h, uint64_t* out, int64_t cap_rows) {
  if (!h || !out || cap_rows < 0) return -1;
  if (hipSetDevice(h->device) != hipSuccess || wait_idle(h) != DGSQP_OK) return 0;
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  const DgParkPool& pool = g_park[h->device & 63];
  if (!pool.entries || pool.owner != h) return 0;        // (another launch has used the device's pool since)
  uint64_t st[2];
  if (dgsqp_deferral_stats(h, st) != DGSQP_OK) return -1;
  const int64_t n = (int64_t)st[0] < cap_rows ? (int64_t)st[0] : cap_rows;
  std::vector<DgParkEntry> e((size_t)n);
  if (n > 0 && hipMemcpy(e.data(), pool.entries, sizeof(DgParkEntry) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int64_t i = 0; i < n; i++) {
    uint64_t* r = out + 11 * i;
    memcpy(r + 8, e[i].cond, sizeof(double) * 3);
    r[0] = (uint64_t)e[i].ticket; r[1] = (uint64_t)e[i].sqp_it; r[2] = (uint64_t)e[i].total_qp; r[3] = e[i].key;
    r[4] = e[i].t_park; r[5] = e[i].t_resume; r[6] = e[i].t_done; r[7] = ((uint64_t)(uint32_t)e[i].final_qps << 32) | (uint32_t)e[i].final_its;
  }
  return (int)n;
}

int dgsqp_coop_stats(dgsqp_handle_t h, uint64_t* out4 /* six values */) {
  if (!h || !out4 || !h->d_coop) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc = wait_idle(h); if (rc) return rc; }
  DgCoop hdr;
  HIPCHK(h, hipMemcpy(&hdr, h->d_coop, sizeof(DgCoop) - sizeof(DgCoopJob), hipMemcpyDeviceToHost));
  out4[0] = hdr.helped; out4[1] = hdr.helper_regs; out4[2] = hdr.finished; out4[3] = hdr.idle;
  out4[4] = hdr.used; out4[5] = hdr.mismatches;
  return DGSQP_OK;
}

int dgsqp_osqp_counters(dgsqp_handle_t h, uint64_t* out2, int reset) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { const int rc = wait_idle(h); if (rc) return rc; }
  if (out2) {
    unsigned long long v[2] = {0ull, 0ull};
    HIPCHK(h, hipMemcpyFromSymbol(v, HIP_SYMBOL(dg_osqp_count), sizeof v));
    out2[0] = v[0]; out2[1] = v[1];
  }
  if (reset) {
    const unsigned long long z[2] = {0ull, 0ull};
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(dg_osqp_count), z, sizeof z));
  }
  return DGSQP_OK;
}

int dgsqp_fetch_results(dgsqp_handle_t h, double* u_out, double* l_out, double* x_out, int32_t* status, int32_t* iters,
                        int32_t* qp_solves, double* cond, double* cost) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const DgProb& D = h->hp;
  const int64_t B = h->B;
  if (B == 0) return DGSQP_OK;
  { const int rcw = wait_idle(h); if (rcw) return rcw; }      // (a member of a grouped launch was solved on its leader's stream)
  if (u_out) HIPCHK(h, hipMemcpyAsync(u_out, h->d_u, sizeof(double) * B * D.n, hipMemcpyDeviceToHost, h->stream));
  if (l_out) HIPCHK(h, hipMemcpyAsync(l_out, h->d_l, sizeof(double) * B * D.nc, hipMemcpyDeviceToHost, h->stream));
  if (x_out) HIPCHK(h, hipMemcpyAsync(x_out, h->d_x, sizeof(double) * B * (D.N + 1) * D.nq, hipMemcpyDeviceToHost, h->stream));
  if (status) HIPCHK(h, hipMemcpyAsync(status, h->d_status, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
  if (iters) HIPCHK(h, hipMemcpyAsync(iters, h->d_iters, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
  if (qp_solves) HIPCHK(h, hipMemcpyAsync(qp_solves, h->d_qps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
  if (cond) HIPCHK(h, hipMemcpyAsync(cond, h->d_cond, sizeof(double) * B * 3, hipMemcpyDeviceToHost, h->stream));
  if (cost) HIPCHK(h, hipMemcpyAsync(cost, h->d_cost, sizeof(double) * B * D.M, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DGSQP_OK;
}

int dgsqp_solve_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u_ws, double* u_out, double* l_out,
                      double* x_out, int32_t* status, int32_t* iters, int32_t* qp_solves, double* cond, double* cost,
                      dgsqp_timing_t* tm) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  int rc = dgsqp_stage_inputs(h, B, x0, u_ws);
  if (rc) return rc;
  dgsqp_timing_t t2;
  rc = dgsqp_solve_staged(h, &t2);
  if (rc) return rc;
  rc = dgsqp_fetch_results(h, u_out, l_out, x_out, status, iters, qp_solves, cond, cost);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (tm) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
    *tm = t2;
    tm->total_ms = ms;
  }
  return DGSQP_OK;
}

// fp32 boundary: widen / narrow on the device
__global__ void dg_widen_kernel(int64_t n, const float* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}
__global__ void dg_narrow_kernel(int64_t n, const double* __restrict__ src, float* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

int dgsqp_solve_batch_f32(dgsqp_handle_t h, int64_t B, const float* x0, const float* u_ws, float* u_out, float* l_out,
                          float* x_out, int32_t* status, int32_t* iters, int32_t* qp_solves, float* cond, float* cost,
                          dgsqp_timing_t* tm) {
  if (!h || B < 0 || (B > 0 && (!x0 || !u_ws))) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  int rc = wait_idle(h);
  if (rc) return rc;
  h->B = B;
  if (B == 0) return DGSQP_OK;
  rc = ensure_batch(h, B);
  if (rc) return rc;
  rc = ensure_ws(h, (size_t)grid_for(h, B));
  if (rc) return rc;
  const DgProb& D = h->hp;
  const size_t nx = (size_t)(D.N + 1) * D.nq;
  size_t big = (size_t)B * (D.nc > (int)nx ? (size_t)D.nc : nx);
  if ((size_t)B * D.n > big) big = (size_t)B * D.n;
  TmpBuf tb;
  float* f = tb.alloc<float>(big);          // one single-precision staging buffer, reused for every array
  if (!f) { h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  auto in = [&](const float* src, double* dst, size_t cnt) -> int {
    HIPCHK(h, hipMemcpyAsync(f, src, sizeof(float) * cnt, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(dg_widen_kernel, dim3(256), dim3(256), 0, h->stream, (int64_t)cnt, f, dst);
    HIPCHK(h, hipGetLastError());
    return DGSQP_OK;
  };
  if ((rc = in(x0, h->d_x0, (size_t)B * D.nq))) return rc;
  if ((rc = in(u_ws, h->d_uws, (size_t)B * D.n))) return rc;
  dgsqp_timing_t t2;
  rc = dgsqp_solve_staged(h, &t2);
  if (rc) return rc;
  auto out = [&](const double* src, float* dst, size_t cnt) -> int {
    if (!dst) return DGSQP_OK;
    hipLaunchKernelGGL(dg_narrow_kernel, dim3(256), dim3(256), 0, h->stream, (int64_t)cnt, src, f);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(dst, f, sizeof(float) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));       // f is reused by the next array
    return DGSQP_OK;
  };
  if ((rc = out(h->d_u, u_out, (size_t)B * D.n))) return rc;
  if ((rc = out(h->d_l, l_out, (size_t)B * D.nc))) return rc;
  if ((rc = out(h->d_x, x_out, (size_t)B * nx))) return rc;
  if ((rc = out(h->d_cond, cond, (size_t)B * 3))) return rc;
  if ((rc = out(h->d_cost, cost, (size_t)B * D.M))) return rc;
  rc = dgsqp_fetch_results(h, nullptr, nullptr, nullptr, status, iters, qp_solves, nullptr, nullptr);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (tm) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
    *tm = t2;
    tm->total_ms = ms;
  }
  return DGSQP_OK;
}

// Diagnostic build (-DDG_PROF) only: cycles spent on each scenario of the last launch.
int dgsqp_prof_scn(unsigned long long* out, int n) {
#ifdef DG_PROF
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dg_prof_scn), sizeof(unsigned long long) * (n < 16384 ? n : 16384)) != hipSuccess) return DGSQP_E_DEVICE;
  return 0;
#else
  (void)out; (void)n;
  return -1;
#endif
}

// Diagnostic build (-DDG_PROF) only: read and clear the per-phase cycle counters.
int dgsqp_prof_read(unsigned long long* out, int n) {
#ifdef DG_PROF
  unsigned long long tmp[PH_COUNT * 2];
  if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(dg_prof), sizeof(tmp)) != hipSuccess) return DGSQP_E_DEVICE;
  for (int i = 0; i < n && i < PH_COUNT * 2; i++) out[i] = tmp[i];
  memset(tmp, 0, sizeof(tmp));
  if (hipMemcpyToSymbol(HIP_SYMBOL(dg_prof), tmp, sizeof(tmp)) != hipSuccess) return DGSQP_E_DEVICE;
  return PH_COUNT;
#else
  (void)out; (void)n;
  return 0;
#endif
}

int dgsqp_set_trace(dgsqp_handle_t h, int pairs_per_scenario) {
  if (!h || pairs_per_scenario < 0) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  h->trace_cap = pairs_per_scenario;
  h->trace_B = h->trace_launch_B = 0;
  if (h->d_trace) { (void)hipFree(h->d_trace); h->d_trace = nullptr; }
  return DGSQP_OK;
}

int dgsqp_fetch_trace(dgsqp_handle_t h, double* out, int64_t capacity_doubles) {
  if (!h || !out || h->trace_cap <= 0 || !h->d_trace || h->trace_launch_B <= 0) { if (h) h->err = "no trace recorded"; return DGSQP_E_ARG; }
  const int64_t need = h->trace_launch_B * (1 + 2 * (int64_t)h->trace_cap);
  if (capacity_doubles < need) { h->err = "trace buffer too small: need " + std::to_string(need) + " doubles"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  HIPCHK(h, hipMemcpy(out, h->d_trace, sizeof(double) * need, hipMemcpyDeviceToHost));
  return DGSQP_OK;
}

int dgsqp_set_iterate_log(dgsqp_handle_t h, int records_per_scenario) {
  if (!h || records_per_scenario < 0) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  h->itlog_cap = records_per_scenario;
  h->itlog_B = h->itlog_launch_B = 0;
  if (h->d_itlog) { (void)hipFree(h->d_itlog); h->d_itlog = nullptr; }
  return DGSQP_OK;
}

int dgsqp_fetch_iterate_log(dgsqp_handle_t h, double* out, int64_t capacity_doubles) {
  if (!h || !out || h->itlog_cap <= 0 || !h->d_itlog || h->itlog_launch_B <= 0) { if (h) h->err = "no iterate log recorded"; return DGSQP_E_ARG; }
  const int64_t need = h->itlog_launch_B * (1 + (int64_t)h->itlog_cap * (h->hp.n + h->hp.nc));
  if (capacity_doubles < need) { h->err = "iterate-log buffer too small: need " + std::to_string(need) + " doubles"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  HIPCHK(h, hipMemcpy(out, h->d_itlog, sizeof(double) * need, hipMemcpyDeviceToHost));
  return DGSQP_OK;
}

int dgsqp_synchronize(dgsqp_handle_t h) {
  if (!h) return DGSQP_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  // A member of a grouped launch is solved on its LEADER's stream: wait for that kernel first (and leave the group), then for
  // whatever is queued on the handle's own stream (copies, the stats gather).
  { const int rc = wait_idle(h); if (rc) return rc; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DGSQP_OK;
}

int dgsqp_evaluate_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u, const double* l, double* q,
                         double* g, double* G, double* Q, double* x, double* l0) {
  if (!h || B < 0 || !x0 || !u) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  if (B == 0) return DGSQP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const DgProb& D = h->hp;
  const int grid = grid_for(h, B);
  int rc = wait_idle(h);
  if (rc) return rc;
  rc = ensure_ws(h, (size_t)grid);
  if (rc) return rc;
  TmpBuf tb;
  const size_t n = D.n, nc = D.nc, nx = (size_t)(D.N + 1) * D.nq;
  double* dx0 = tb.alloc<double>(B * D.nq); double* du = tb.alloc<double>(B * n);
  double* dl = l ? tb.alloc<double>(B * nc) : nullptr;
  double* dq = q ? tb.alloc<double>(B * n) : nullptr; double* dg = g ? tb.alloc<double>(B * nc) : nullptr;
  double* dG = G ? tb.alloc<double>(B * nc * n) : nullptr; double* dQ = Q ? tb.alloc<double>(B * n * n) : nullptr;
  double* dx = x ? tb.alloc<double>(B * nx) : nullptr; double* dl0 = l0 ? tb.alloc<double>(B * nc) : nullptr;
  if (!dx0 || !du || (l && !dl) || (q && !dq) || (g && !dg) || (G && !dG) || (Q && !dQ) || (x && !dx) || (l0 && !dl0)) { h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  HIPCHK(h, hipMemcpy(dx0, x0, sizeof(double) * B * D.nq, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(du, u, sizeof(double) * B * n, hipMemcpyHostToDevice));
  if (l) HIPCHK(h, hipMemcpy(dl, l, sizeof(double) * B * nc, hipMemcpyHostToDevice));
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  { int rcu = upload_problem(h); if (rcu) return rcu; }
  h->in_flight = true;
  hipLaunchKernelGGL(dg_evaluate_kernel, dim3(grid), dim3(DG_BLOCK), h->lds_bytes, h->stream, h->dp, B, dx0, du, dl, dq, dg, dG, dQ, dx, dl0, h->ws);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->in_flight = false;
  if (q) HIPCHK(h, hipMemcpy(q, dq, sizeof(double) * B * n, hipMemcpyDeviceToHost));
  if (g) HIPCHK(h, hipMemcpy(g, dg, sizeof(double) * B * nc, hipMemcpyDeviceToHost));
  if (G) HIPCHK(h, hipMemcpy(G, dG, sizeof(double) * B * nc * n, hipMemcpyDeviceToHost));
  if (Q) HIPCHK(h, hipMemcpy(Q, dQ, sizeof(double) * B * n * n, hipMemcpyDeviceToHost));
  if (x) HIPCHK(h, hipMemcpy(x, dx, sizeof(double) * B * nx, hipMemcpyDeviceToHost));
  if (l0) HIPCHK(h, hipMemcpy(l0, dl0, sizeof(double) * B * nc, hipMemcpyDeviceToHost));
  return DGSQP_OK;
}

int dgsqp_pid_warm_start_batch(dgsqp_handle_t h, int64_t B, const double* q0, const dgsqp_pid_t* pid, double* u_ws,
                               double* q_ws, int32_t* collide) {
  if (!h || B < 0 || !q0 || !pid || !u_ws || pid->substeps < 1) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  if (B == 0) return DGSQP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const DgProb& D = h->hp;
  for (int a = 0; a < D.M; a++)
    if (D.nqa[a] == 4) { h->err = "the PID lane follower needs a Frenet-frame model (the merge script starts from zero inputs)"; return DGSQP_E_ARG; }
  TmpBuf tb;
  const size_t nx = (size_t)(D.N + 1) * D.nq;
  double* dq0 = tb.alloc<double>(B * D.nq); double* du = tb.alloc<double>(B * D.n);
  double* dq = (q_ws || collide) ? tb.alloc<double>(B * nx) : nullptr;
  int32_t* dc = collide ? tb.alloc<int32_t>(B) : nullptr;
  if (!dq0 || !du || ((q_ws || collide) && !dq) || (collide && !dc)) { h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  HIPCHK(h, hipMemcpy(dq0, q0, sizeof(double) * B * D.nq, hipMemcpyHostToDevice));
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  { int rcu = upload_problem(h); if (rcu) return rcu; }
  h->in_flight = true;
  const int64_t lanes = B * D.M;
  int grid = (int)((lanes + DG_BLOCK - 1) / DG_BLOCK);
  if (grid > 4 * h->num_cu) grid = 4 * h->num_cu;
  hipLaunchKernelGGL(dg_pid_kernel, dim3(grid), dim3(DG_BLOCK), (size_t)D.L.scr * sizeof(double), h->stream, B, dq0, *pid, du, dq);
  HIPCHK(h, hipGetLastError());
  if (collide) {
    hipLaunchKernelGGL(dg_collide_kernel, dim3((int)((B + 255) / 256)), dim3(256), 0, h->stream, B, dq, dc);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->in_flight = false;
  HIPCHK(h, hipMemcpy(u_ws, du, sizeof(double) * B * D.n, hipMemcpyDeviceToHost));
  if (q_ws) HIPCHK(h, hipMemcpy(q_ws, dq, sizeof(double) * B * nx, hipMemcpyDeviceToHost));
  if (collide) HIPCHK(h, hipMemcpy(collide, dc, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  return DGSQP_OK;
}

int dgsqp_sample_batch(dgsqp_handle_t h, int64_t B, const dgsqp_sampler_t* spec, const dgsqp_pid_t* pid, double* x0_out,
                       double* u_ws_out, int64_t* candidates, int stage) {
  if (!h || B < 0 || !spec || spec->kind < 0 || spec->kind > DGSQP_SAMPLER_MERGE) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  const DgProb& D = h->hp;
  const bool merge = spec->kind == DGSQP_SAMPLER_MERGE;
  if (!merge && (!pid || pid->substeps < 1 || spec->n_key < 2 || spec->n_key > DGSQP_MAX_SEGS + 1)) { h->err = "bad sampler description"; return DGSQP_E_ARG; }
  if (spec->kind == DGSQP_SAMPLER_FIRST_SEGMENT && D.M != 2) { h->err = "the first-segment sampler places two cars"; return DGSQP_E_ARG; }
  for (int a = 0; a < D.M; a++)
    if ((D.nqa[a] == 4) != merge) { h->err = "sampler and vehicle model do not fit (merge: unicycles; the others: Frenet-frame models)"; return DGSQP_E_ARG; }
  if (candidates) *candidates = 0;
  if (B == 0) return DGSQP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  { int rc = ensure_batch(h, B); if (rc) return rc; }
  { int rc = ensure_ws(h, (size_t)grid_for(h, B)); if (rc) return rc; }
  const int64_t nround = std::max<int64_t>(256, std::min<int64_t>(2 * B, 1 << 16));        // candidates per round
  const size_t nx = (size_t)(D.N + 1) * D.nq;
  TmpBuf tb;
  double* dq0 = tb.alloc<double>(nround * D.nq); double* du = tb.alloc<double>(nround * D.n); double* dq = tb.alloc<double>(nround * nx);
  int32_t* dok = tb.alloc<int32_t>(nround); int32_t* dcol = tb.alloc<int32_t>(nround); int32_t* dpos = tb.alloc<int32_t>(nround); int32_t* dcnt = tb.alloc<int32_t>(1);
  if (!dq0 || !du || !dq || !dok || !dcol || !dpos || !dcnt) { h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  { int rcu = upload_problem(h); if (rcu) return rcu; }
  int64_t have = 0;
  unsigned long long c0 = 0;
  for (int round = 0; have < B; round++) {
    if (round > 10000) { h->err = "sampler did not produce enough collision-free scenarios"; return DGSQP_E_ARG; }
    hipLaunchKernelGGL(dg_sample_place_kernel, dim3((unsigned)((nround + 255) / 256)), dim3(256), 0, h->stream, nround, c0, *spec, dq0, dok);
    HIPCHK(h, hipGetLastError());
    if (merge) {
      hipLaunchKernelGGL(dg_sample_zero_rollout_kernel, dim3((unsigned)((nround * D.M + 255) / 256)), dim3(256), 0, h->stream, nround, dq0, dq);
    } else {
      int grid = (int)((nround * D.M + DG_BLOCK - 1) / DG_BLOCK);
      if (grid > 4 * h->num_cu) grid = 4 * h->num_cu;
      hipLaunchKernelGGL(dg_pid_kernel, dim3(grid), dim3(DG_BLOCK), (size_t)D.L.scr * sizeof(double), h->stream, nround, dq0, *pid, du, dq);
    }
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(dg_collide_kernel, dim3((unsigned)((nround + 255) / 256)), dim3(256), 0, h->stream, nround, dq, dcol);
    hipLaunchKernelGGL(dg_sample_scan_kernel, dim3(1), dim3(1024), 0, h->stream, nround, dok, dcol, dpos, dcnt);
    hipLaunchKernelGGL(dg_sample_gather_kernel, dim3(1024), dim3(128), 0, h->stream, nround, have, B, dpos, dq0, merge ? (const double*)nullptr : du, h->d_x0, h->d_uws);
    HIPCHK(h, hipGetLastError());
    int32_t cnt = 0;
    HIPCHK(h, hipMemcpyAsync(&cnt, dcnt, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (have + cnt >= B && candidates) {
      // index after the (B - have)-th accepted candidate of this round
      std::vector<int32_t> pos((size_t)nround);
      HIPCHK(h, hipMemcpy(pos.data(), dpos, sizeof(int32_t) * nround, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < nround; i++) if (pos[i] == (int32_t)(B - have - 1)) { *candidates = (int64_t)(c0 + i + 1); break; }
    }
    have += cnt;
    c0 += (unsigned long long)nround;
  }
  if (x0_out) HIPCHK(h, hipMemcpy(x0_out, h->d_x0, sizeof(double) * B * D.nq, hipMemcpyDeviceToHost));
  if (u_ws_out) HIPCHK(h, hipMemcpy(u_ws_out, h->d_uws, sizeof(double) * B * D.n, hipMemcpyDeviceToHost));
  h->B = stage ? B : 0;          // (the staging buffers were used either way: without `stage` nothing is left staged)
  return DGSQP_OK;
}

int dgsqp_qp_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u, const double* l, double* du_out,
                   double* lhat, double* Qpd, int32_t* flag) {
  return dgsqp_qp_batch_info(h, B, x0, u, l, du_out, lhat, Qpd, flag, nullptr);
}

int dgsqp_qp_batch_info(dgsqp_handle_t h, int64_t B, const double* x0, const double* u, const double* l, double* du_out,
                        double* lhat, double* Qpd, int32_t* flag, double* info8) {
  if (!h || B < 0 || !x0 || !u || !l) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  if (B == 0) return DGSQP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const DgProb& D = h->hp;
  const int grid = grid_for(h, B);
  int rc = wait_idle(h);
  if (rc) return rc;
  rc = ensure_ws(h, (size_t)grid);
  if (rc) return rc;
  TmpBuf tb;
  const size_t n = D.n, nc = D.nc;
  double* dx0 = tb.alloc<double>(B * D.nq); double* du = tb.alloc<double>(B * n); double* dl = tb.alloc<double>(B * nc);
  double* ddu = du_out ? tb.alloc<double>(B * n) : nullptr; double* dlh = lhat ? tb.alloc<double>(B * nc) : nullptr;
  double* dQ = Qpd ? tb.alloc<double>(B * n * n) : nullptr; int32_t* df = flag ? tb.alloc<int32_t>(B) : nullptr;
  double* dinfo = info8 ? tb.alloc<double>(B * 8) : nullptr;
  if (!dx0 || !du || !dl || (du_out && !ddu) || (lhat && !dlh) || (Qpd && !dQ) || (flag && !df) || (info8 && !dinfo)) { h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  HIPCHK(h, hipMemcpy(dx0, x0, sizeof(double) * B * D.nq, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(du, u, sizeof(double) * B * n, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dl, l, sizeof(double) * B * nc, hipMemcpyHostToDevice));
  std::unique_lock<std::mutex> game_lock(g_reg_mutex);
  { int rcu = upload_problem(h); if (rcu) return rcu; }
  h->in_flight = true;
  hipLaunchKernelGGL(dg_qp_kernel, dim3(grid), dim3(DG_BLOCK), h->lds_bytes, h->stream, h->dp, B, dx0, du, dl, ddu, dlh, dQ, df, dinfo, h->ws);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->in_flight = false;
  if (du_out) HIPCHK(h, hipMemcpy(du_out, ddu, sizeof(double) * B * n, hipMemcpyDeviceToHost));
  if (lhat) HIPCHK(h, hipMemcpy(lhat, dlh, sizeof(double) * B * nc, hipMemcpyDeviceToHost));
  if (Qpd) HIPCHK(h, hipMemcpy(Qpd, dQ, sizeof(double) * B * n * n, hipMemcpyDeviceToHost));
  if (flag) HIPCHK(h, hipMemcpy(flag, df, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  if (info8) HIPCHK(h, hipMemcpy(info8, dinfo, sizeof(double) * B * 8, hipMemcpyDeviceToHost));
  return DGSQP_OK;
}

// ---- RCCL communicator owned by the handle -----------------------------------------------------------------------------
int dgsqp_comm_unique_id(char* out128) {
  if (!out128) return DGSQP_E_ARG;
  if (!g_rccl.load()) { g_create_err = g_rccl.err; return DGSQP_E_DEVICE; }
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return DGSQP_E_DEVICE; }
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return DGSQP_OK;
}

int dgsqp_comm_init(dgsqp_handle_t h, const char* id128, int rank, int world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) { if (h) h->err = "bad argument"; return DGSQP_E_ARG; }
  if (h->comm) { h->err = "communicator already initialised"; return DGSQP_E_ARG; }
  if (!g_rccl.load()) { h->err = g_rccl.err; return DGSQP_E_DEVICE; }
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  dgsqp_comm_state* c = new dgsqp_comm_state();
  c->rank = rank; c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"); delete c; return DGSQP_E_DEVICE; }
  if (hipMalloc(&c->d_red, sizeof(double) * 64) != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; h->err = "hipMalloc failed"; return DGSQP_E_NOMEM; }
  h->comm = c;
  return DGSQP_OK;
}

int dgsqp_comm_destroy(dgsqp_handle_t h) {
  if (!h || !h->comm) return DGSQP_OK;
  (void)hipSetDevice(h->device);
  dgsqp_comm_state* c = h->comm;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_rec) (void)hipFree(c->d_rec);
  if (c->d_all) (void)hipFree(c->d_all);
  if (c->d_red) (void)hipFree(c->d_red);
  delete c;
  h->comm = nullptr;
  return DGSQP_OK;
}

int dgsqp_gather_stats(dgsqp_handle_t h, int64_t B_pad, dgsqp_stat_record_t* out) {
  if (!h || !out || !h->comm) { if (h) h->err = "dgsqp_comm_init first"; return DGSQP_E_ARG; }
  if (B_pad < h->B || B_pad <= 0) { h->err = "B_pad must be at least this rank's batch size"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = wait_idle(h); if (rc) return rc; }
  dgsqp_comm_state* c = h->comm;
  if (c->rec_cap < B_pad) {
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_all) (void)hipFree(c->d_all);
    c->d_rec = c->d_all = nullptr; c->rec_cap = 0;
    HIPCHK(h, hipMalloc(&c->d_rec, sizeof(dgsqp_stat_record_t) * B_pad));
    HIPCHK(h, hipMalloc(&c->d_all, sizeof(dgsqp_stat_record_t) * B_pad * c->world));
    c->rec_cap = B_pad;
  }
  hipLaunchKernelGGL(dg_pack_stats_kernel, dim3((unsigned)((B_pad + 255) / 256)), dim3(256), 0, h->stream, h->B, B_pad, h->hp.M, c->rank,
                     h->d_status, h->d_iters, h->d_qps, h->d_cond, h->d_cost, c->d_rec);
  HIPCHK(h, hipGetLastError());
  NCCLCHK(h, g_rccl.AllGather(c->d_rec, c->d_all, sizeof(dgsqp_stat_record_t) * B_pad, ncclChar, c->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(out, c->d_all, sizeof(dgsqp_stat_record_t) * B_pad * c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DGSQP_OK;
}

int dgsqp_comm_allreduce_max(dgsqp_handle_t h, double* values, int count) {
  if (!h || !values || count < 1 || count > 64 || !h->comm) { if (h) h->err = "bad argument / dgsqp_comm_init first"; return DGSQP_E_ARG; }
  HIPCHK(h, hipSetDevice(h->device));
  dgsqp_comm_state* c = h->comm;
  HIPCHK(h, hipMemcpyAsync(c->d_red, values, sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
  NCCLCHK(h, g_rccl.AllReduce(c->d_red, c->d_red, count, ncclDouble, ncclMax, c->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(values, c->d_red, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DGSQP_OK;
}

int dgsqp_comm_barrier(dgsqp_handle_t h) {
  double v = 0.0;
  return dgsqp_comm_allreduce_max(h, &v, 1);
}

}  // extern "C"
