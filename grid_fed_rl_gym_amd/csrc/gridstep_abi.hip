// gridstep_abi.hip -- host side of the C ABI declared in include/gridstep.h.
//
// Owns: the compiled topology tables, one slab of per-instance rows in HBM
// (slab[group][row][64 lanes]), staging buffers for the batch-major <-> batch-innermost layout
// change, one HIP stream, optional HIP-event timing of every launch, and (lazily, via dlopen)
// an RCCL communicator for the observation all-gather.  No CPU arithmetic on the data path:
// every entry point either moves bytes or launches kernels.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gridstep.h"
#include "gs_internal.h"
#include "kernels.h"
#include "topology.h"
#include "mesh_schedule.h"

namespace {

thread_local std::string g_last_error;

// ---- RCCL entry points resolved at run time -------------------------------------------------
typedef struct { char internal[128]; } gs_ncclUniqueId;
typedef void* gs_ncclComm_t;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(gs_ncclUniqueId*) = nullptr;
  int (*CommInitRank)(gs_ncclComm_t*, int, gs_ncclUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, gs_ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(gs_ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(gs_ncclComm_t, int*) = nullptr;
  int (*CommUserRank)(gs_ncclComm_t, int*) = nullptr;
  int (*CommCuDevice)(gs_ncclComm_t, int*) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

bool load_rccl(std::string& why) {
  if (g_rccl.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
  if (!lib) { why = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
  RcclApi a; a.lib = lib;
  a.GetUniqueId = (int (*)(gs_ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(gs_ncclComm_t*, int, gs_ncclUniqueId, int))dlsym(lib, "ncclCommInitRank");
  a.AllGather = (int (*)(const void*, void*, size_t, int, gs_ncclComm_t, hipStream_t))dlsym(lib, "ncclAllGather");
  a.CommDestroy = (int (*)(gs_ncclComm_t))dlsym(lib, "ncclCommDestroy");
  a.GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
  a.GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
  a.CommCount = (int (*)(gs_ncclComm_t, int*))dlsym(lib, "ncclCommCount");
  a.CommUserRank = (int (*)(gs_ncclComm_t, int*))dlsym(lib, "ncclCommUserRank");
  a.CommCuDevice = (int (*)(gs_ncclComm_t, int*))dlsym(lib, "ncclCommCuDevice");
  a.GetVersion = (int (*)(int*))dlsym(lib, "ncclGetVersion");
  a.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) { why = "librccl lacks a required symbol"; return false; }
  g_rccl = a;
  return true;
}

struct TimedLaunch { int kid; hipEvent_t a, b; };
constexpr size_t GS_CHECKS_MAX_EVENTS = 4096;

}  // namespace

struct gs_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  int B = 0, Bp = 0, groups = 0, W = 1;
  int n = 0, m = 0, obs_dim = 0, action_dim = 0, state_dim = 0;
  int n_loads = 0, n_gens = 0, n_bats = 0;
  gs_config cfg{};
  HostTopology topo;
  GsTables T{};
  GsRows R{};
  GsSolveCfg SC{};
  GsEnvCfg EC{};
  double total_load = 0.0;
  struct gs_checks* fused = nullptr;      // checks evaluated inside the step kernel's epilogue (gs_checks_set_fused)
  int solve_kernel = 0;     // 0 tree, 1 lu, 2 fbs, 3 dense, 4 tree with LDS messages, 5 fbs with LDS messages, 6 fbs as a dataflow over LDS
  size_t dyn_lds = 0;
  // the env step of a handle whose solver is the dataflow sweep runs the second-generation kernel (kernels_flow2.hip:
  // 32 instances per workgroup, half-waves on different buses) when the feeder fits its tables; gs_solve keeps kernel 6
  bool flow2 = false; GsF2Tables F2{}; std::string flow2_why;
  bool f2_small = false, f2_half = false, f2_wide = false; int f2_iw = 32, f2_nw = 16, f2_npos = 0;
  // A step of the 16-instance sweep kernel goes out as TWO launches, each half of the workgroups, on two streams: consecutive
  // steps of one half need nothing from the other half, so the second stream's kernels slide into the launch gaps and the
  // uneven tails of the first's (two handles of 4096 instances on two streams: 205 M env-steps/s against 186 M for one of
  // 8192).  `forked`: stream2 holds step launches the main stream has not waited for yet; every entry point other than the
  // step itself joins first (GS_ENTER).
  bool split_ok = false, forked = false;
  hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_peer = nullptr, ev_peer2 = nullptr;     // which member of the family (8 instances per workgroup for small feeders)
  // The second-generation step kernels do not write the (|V|, angle) / (flow, |P| / rating) row pairs: those are the first
  // 2 n + 2 m columns of the observation block the step writes anyway.  `rows_stale`: the rows lag behind `last_obs`, the block
  // of the last step; every entry point that reads or partly rewrites them restores them first (ensure_rows).
  bool lean = false, rows_stale = false; const double* last_obs = nullptr;
  // solve_kernel 7: Newton-Raphson with the dense block LU on the matrix cores (kernels_dense.hip), a launch of its own between
  // the two halves of the step / solve
  GsDenseArgs DA{}; int dense_grid = 0; size_t dense_lds = 0; bool dense_blockrow = true;
  // solve_kernel 8: Newton-Raphson with the sparse block LU of an instance in the LDS of a one-wave workgroup (kernels_sparse.hip),
  // launched the same way
  GsSparseArgs SA{}; int sparse_grid = 0; size_t sparse_lds = 0;
  bool nr2 = false;         // ... and likewise the Newton-Raphson step of a radial all-PQ feeder (gs_k_step_nr_flow2) instead of kernel 4
  bool nrm = false;         // ... and of a meshed all-PQ feeder whose block LU stays narrow (gs_k_step_nr_mesh2, mesh_schedule.h) instead of kernel 1
  std::string mesh_why; int mesh_levels = 0, mesh_rows = 0, mesh_units = 0, mesh_messages = 0, mesh_accs = 0;
  unsigned long long* d_stamps = nullptr;
  bool was_reset = false;
  std::vector<void*> allocs;
  char* arena = nullptr; size_t arena_left = 0;      // dev_alloc: the current chunk of small tables
  double* slab = nullptr;
  double* d_in = nullptr; size_t in_doubles = 0;
  double* d_out = nullptr; size_t out_doubles = 0;
  // [B][obs_dim] x 2, owned by the environment path: both written whole at reset, the changing columns of the other one by
  // every step -- so that the all-gather of step k (on its own stream) can run while step k + 1 computes
  double* d_obs2[2] = {nullptr, nullptr}; int obs_cur = 0;
  hipStream_t comm_stream = nullptr; hipEvent_t ev_step = nullptr, ev_gather[2] = {nullptr, nullptr}; bool gather_pending[2] = {false, false};
  int obs_skip0 = 0, obs_skip1 = 0;   // the block of per-instance constants inside an observation
  // host observation arrays whose constant columns are in place (gs_host_obs_bind): gs_step / gs_download_step copy only the
  // changing columns into these -- two strided copies instead of one whole block, 36 % fewer bytes over PCIe on the 123-bus feeder
  std::vector<const double*> bound_obs;
  float* d_obs32 = nullptr;                // float32 copy of the observation block (gs_step_f32 / gs_download_step_f32), on first use
  hipEvent_t ev_scalars = nullptr;
  double* d_actions = nullptr; int n_action_batches = 0;
  // gs_rollout: [T + 1][B][obs_dim] observation sequence, [T][B][A] actions, [T][B] rewards / done flags, and the side
  // list of terminal observations (the rows the in-place resets replaced)
  struct Rollout {
    int T_cap = 0, T = 0, term_cap = 0; uint64_t calls = 0;
    double* obs_seq = nullptr; double* act = nullptr; double* rew = nullptr; uint8_t* done = nullptr;
    int32_t* term_count = nullptr; int32_t* term_idx = nullptr; double* term_obs = nullptr;
    int32_t n_term = 0;
  } ro;
  double* d_cst = nullptr;
  int32_t *map_obs = nullptr, *map_vm = nullptr, *map_va = nullptr, *map_flow = nullptr, *map_load = nullptr,
          *map_p = nullptr, *map_q = nullptr, *map_act = nullptr, *map_state = nullptr;
  int32_t *rows_f = nullptr, *rows_i = nullptr, *rows_u = nullptr;
  double* sc_f = nullptr; int32_t* sc_i = nullptr; uint8_t* sc_u = nullptr;
  uint64_t* d_seeds = nullptr; uint8_t* d_mask = nullptr;
  // gs_fallback_linear: line reactances, dict-order bus lists and staging, created on first use
  std::vector<double> line_x;
  bool fb_ready = false; GsFallbackArgs FB{};
  double *fb_load = nullptr, *fb_gen = nullptr, *fb_tl = nullptr, *fb_tg = nullptr; uint8_t* fb_mask = nullptr; int32_t* fb_applied = nullptr;
  // host copies of the per-instance scalars: ONE page-locked block the device addresses -- gs_k_scalars stores into it itself (three
  // copies through the runtime's staging buffer cost 80 us of a 0.9 ms env.step()); hd_*: the same block as the device sees it
  void* h_pin = nullptr;
  double* h_f = nullptr; int32_t* h_i = nullptr; uint8_t* h_u = nullptr; uint32_t* h_v4 = nullptr;
  double* hd_f = nullptr; int32_t* hd_i = nullptr; uint8_t* hd_u = nullptr; uint32_t* hd_v4 = nullptr;
  // timing
  bool timing = false;
  bool timing_span = false, span_open = false; hipEvent_t span_a = nullptr, span_b = nullptr; int span_kid = 0; int64_t span_launches[8] = {0};
  std::vector<TimedLaunch> timed; size_t timed_used = 0;
  // comm
  gs_ncclComm_t comm = nullptr; int rank = 0, world = 1; double* d_obs_full = nullptr;
  double *d_gather_send = nullptr, *d_gather_recv = nullptr;     // compact observation blocks (changing columns only): [B][nd], [world * B][nd]
  struct GsLoopComm* loop = nullptr;                              // the in-process transport (gs_comm_init_loopback) instead of RCCL
  hipEvent_t ev_full = nullptr;                                   // gs_allgather_obs_view: the gathered block is complete
  mutable std::string err;
};

namespace {

enum { SF_REWARD = 0, SF_VMAX, SF_VMIN, SF_LOSSES, SF_EPREW, SF_MAXMIS, SF_COUNT };
enum { SI_VIOL = 0, SI_STEP, SI_ITERS, SI_STATUS, SI_COUNT };
enum { SU_TERM = 0, SU_TRUNC, SU_CONV, SU_VF0, SU_VF1, SU_VF2, SU_VF3, SU_COUNT };

int fail(gs_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_last_error = buf;
  if (h) h->err = buf;
  return code;
}

// A peer's stream as it crosses the C ABI: NULL = none; hipStreamLegacy (1) = the legacy default stream, i.e. handle 0
static inline hipStream_t peer_stream(void* s) { return s == (void*)hipStreamLegacy ? (hipStream_t)nullptr : (hipStream_t)s; }

#define HIPCHK(h, expr)                                                                           \
  do { hipError_t e_ = (expr);                                                                    \
       if (e_ != hipSuccess) return fail((h), GS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// Work of the second step stream joins the main stream (see gs_handle::forked)
static int join_streams(gs_handle* h) {
  if (!h->forked) return GS_OK;
  HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  h->forked = false;
  return GS_OK;
}
#define GS_ENTER(h)                                                                               \
  do { HIPCHK((h), hipSetDevice((h)->device));                                                    \
       if ((h)->forked) { int rc_ = join_streams(h); if (rc_) return rc_; } } while (0)

constexpr size_t GS_ARENA_SMALL = 64 * 1024, GS_ARENA_CHUNK = 2 * 1024 * 1024;
template <typename X>
int dev_alloc(gs_handle* h, X** p, size_t count) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(X);
  // Tables (a few hundred bytes to a few KB each, forty of them) share 2 MB chunks: as allocations of their own each sat on a
  // page of its own, and a workgroup's first touch of every one of them was an address-translation miss at kernel start.
  if (bytes <= GS_ARENA_SMALL && !GS_EXPERIMENT_ENV("GS_NO_TABLE_ARENA")) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (h->arena_left < need) {
      hipError_t e = hipMalloc(&q, GS_ARENA_CHUNK);
      if (e != hipSuccess) return fail(h, GS_E_NOMEM, "hipMalloc(%zu) failed: %s", (size_t)GS_ARENA_CHUNK, hipGetErrorString(e));
      h->allocs.push_back(q);
      h->arena = (char*)q; h->arena_left = GS_ARENA_CHUNK;
    }
    *p = (X*)h->arena;
    h->arena += need; h->arena_left -= need;
    return GS_OK;
  }
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return fail(h, GS_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  h->allocs.push_back(q);
  *p = (X*)q;
  return GS_OK;
}

template <typename X>
int dev_upload(gs_handle* h, const X** p, const std::vector<X>& v) {
  X* q = nullptr;
  int rc = dev_alloc(h, &q, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(h, hipMemcpy(q, v.data(), v.size() * sizeof(X), hipMemcpyHostToDevice));
  *p = q;
  return GS_OK;
}

int upload_map(gs_handle* h, int32_t** p, const std::vector<int32_t>& v) {
  const int32_t* q = nullptr;
  int rc = dev_upload(h, &q, v);
  *p = const_cast<int32_t*>(q);
  return rc;
}

// ---- timing wrapper -------------------------------------------------------------------------
struct LaunchTimer {
  gs_handle* h; TimedLaunch* t = nullptr;
  LaunchTimer(gs_handle* hh, int kid) : h(hh) {
    if (h->timing_span) {          // one event pair around the whole timed region: no marker packets between the launches
      if (!h->span_open) {
        if (!h->span_a && (hipEventCreate(&h->span_a) != hipSuccess || hipEventCreate(&h->span_b) != hipSuccess)) return;
        (void)hipEventRecord(h->span_a, h->stream);
        h->span_open = true; h->span_kid = kid;
        for (int k = 0; k < GS_K_COUNT; ++k) h->span_launches[k] = 0;
      }
      if (kid >= 0 && kid < GS_K_COUNT) h->span_launches[kid] += 1;
      return;
    }
    if (!h->timing) return;
    if (h->timed_used == h->timed.size()) {
      TimedLaunch n; n.kid = kid;
      if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
      h->timed.push_back(n);
    }
    t = &h->timed[h->timed_used++];
    t->kid = kid;
    (void)hipEventRecord(t->a, h->stream);
  }
  ~LaunchTimer() { if (t) (void)hipEventRecord(t->b, h->stream); }
};

// ---- layout movers --------------------------------------------------------------------------
int launch_pack(gs_handle* h, const int32_t* map, int C, double* dst) {
  if (C <= 0) return GS_OK;
  LaunchTimer lt(h, GS_K_PACK);
  dim3 grid(h->groups, (C + 63) / 64);
  hipLaunchKernelGGL(gs_k_pack, grid, dim3(256), 0, h->stream, map, h->d_cst, C, h->R.total, h->slab, dst, h->B);
  HIPCHK(h, hipGetLastError());
  return GS_OK;
}

int launch_unpack(gs_handle* h, const int32_t* map, int C, const double* src, int stride = 0) {
  if (C <= 0) return GS_OK;
  LaunchTimer lt(h, GS_K_UNPACK);
  dim3 grid(h->groups, (C + 63) / 64);
  hipLaunchKernelGGL(gs_k_unpack, grid, dim3(256), 0, h->stream, map, C, h->R.total, h->slab, src, h->B, stride ? stride : C);
  HIPCHK(h, hipGetLastError());
  return GS_OK;
}

// The result rows a lean step left behind (gs_handle::lean), copied back from the observation block it wrote: exact (the
// block's first 2 n + 2 m columns ARE those rows' values).  Called, after GS_ENTER, by whatever reads or partly rewrites them.
int ensure_rows(gs_handle* h) {
  if (!h->rows_stale) return GS_OK;
  h->rows_stale = false;
  return launch_unpack(h, h->map_obs, 2 * h->n + 2 * h->m, h->last_obs, h->obs_dim);
}

int pack_to_host(gs_handle* h, const int32_t* map, int C, double* host) {
  if (!host || C <= 0) return GS_OK;
  int rc = launch_pack(h, map, C, h->d_out);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(host, h->d_out, (size_t)h->B * C * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int unpack_from_host(gs_handle* h, const int32_t* map, int C, const double* host) {
  if (C <= 0) return GS_OK;
  HIPCHK(h, hipMemcpyAsync(h->d_in, host, (size_t)h->B * C * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return launch_unpack(h, map, C, h->d_in);
}

int fetch_scalars(gs_handle* h, bool sync = true) {
  hipLaunchKernelGGL(gs_k_scalars, dim3(h->groups), dim3(64), 0, h->stream, h->rows_f, (int)SF_COUNT, h->rows_i,
                     (int)SI_COUNT, h->rows_u, (int)SU_COUNT, h->R.total, h->slab, h->hd_f, h->hd_i, h->hd_u, h->Bp, h->hd_v4, (int)SU_VF0);
  HIPCHK(h, hipGetLastError());
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

#if defined(GS_BUILD_EXPERIMENTS)
#define GS_DENSE_PANEL_LAUNCH(h, grid, args, nb) hipLaunchKernelGGL(gs_k_nr_dense_mfma, dim3(grid), dim3(256), (h)->dense_lds, (h)->stream, args, (h)->slab, nb)
#else
#define GS_DENSE_PANEL_LAUNCH(h, grid, args, nb) ((void)0)
#endif
#define GS_DENSE_LAUNCH(h, grid, args, nb)                                                                                          \
  do {                                                                                                                              \
    if ((h)->dense_blockrow) hipLaunchKernelGGL(gs_k_nr_dense_mfma2, dim3(grid), dim3(256), (h)->dense_lds, (h)->stream, args, (h)->slab, nb); \
    else GS_DENSE_PANEL_LAUNCH(h, grid, args, nb);                                                                                 \
  } while (0)

int launch_solve(gs_handle* h) {
  LaunchTimer lt(h, GS_K_SOLVE);
  dim3 grid(h->groups), block(64 * h->W);
#define GS_SOLVE(k) hipLaunchKernelGGL(k, grid, block, h->dyn_lds, h->stream, h->T, h->R, h->SC, h->slab, h->B)
  if (h->solve_kernel == 0) GS_SOLVE(gs_k_nr_tree);
  else if (h->solve_kernel == 4) GS_SOLVE(gs_k_nr_tree_lds);
  else if (h->solve_kernel == 1) GS_SOLVE(gs_k_nr_lu);
  else if (h->solve_kernel == 3) GS_SOLVE(gs_k_nr_dense);
  else if (h->solve_kernel == 7) {
    GS_DENSE_LAUNCH(h, h->dense_grid, h->DA, h->B);
    hipLaunchKernelGGL(gs_k_posts_nr_dmfma, grid, block, h->dyn_lds, h->stream, h->T, h->R, h->SC, h->slab, h->B);
  }
#if defined(GS_BUILD_EXPERIMENTS)
  else if (h->solve_kernel == 8) {
    hipLaunchKernelGGL(gs_k_nr_sparse_lds, dim3(h->sparse_grid), dim3(64 * h->SA.waves), h->sparse_lds, h->stream, h->SA, h->slab, h->B);
    hipLaunchKernelGGL(gs_k_posts_nr_dmfma, grid, block, h->dyn_lds, h->stream, h->T, h->R, h->SC, h->slab, h->B);
  }
#endif
  else if (h->solve_kernel == 5) GS_SOLVE(gs_k_fbs_lds);
  else if (h->solve_kernel == 6) GS_SOLVE(gs_k_fbs_flow);
  else GS_SOLVE(gs_k_fbs);
#undef GS_SOLVE
  HIPCHK(h, hipGetLastError());
  return GS_OK;
}

GsFusedChecks fused_checks_args(gs_handle* h);      // defined with gs_checks below

// obs_out: where the step writes the changing columns of its observation block ([B][obs_dim], constants already in
// place); NULL = the other one of the handle's two observation buffers
int step_kernels(gs_handle* h, const double* d_actions, double* obs_out = nullptr, const GsRolloutStep* rs = nullptr) {
  // one fused launch: actions -> pre-solve dynamics -> load flow -> post-solve dynamics / reward / flags
  { LaunchTimer lt(h, GS_K_SOLVE);
    dim3 grid(h->groups), block(64 * h->W);
    if (!obs_out) {
      const int next = h->obs_cur ^ 1;
      if (h->gather_pending[next]) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather[next], 0));
        if (h->split_ok) HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_gather[next], 0));     // (the second half writes the same buffer)
        h->gather_pending[next] = false;
      }
      h->obs_cur = next;
      obs_out = h->d_obs2[next];
    }
    GsPackArgs pa{h->map_obs, h->d_cst, obs_out, h->obs_dim, (int)std::max<size_t>(1, std::min<size_t>(3, (h->dyn_lds - 49152) / (64 * 65 * sizeof(double)))), 0, 0,
                  h->obs_skip0, h->obs_skip1};
    pa.pair_ok = !(h->obs_dim & 1) && !((h->obs_skip1 - h->obs_skip0) & 1) && pa.tiles_per_pass >= 2 && !GS_EXPERIMENT_ENV("GS_PACK_BY_COLUMN");
    pa.early_pass0 = 2 * h->n + 2 * h->m >= 64 * pa.tiles_per_pass;   // the frequency column (grid_env.py:766) lies beyond the first pass
    pa.lean = h->lean ? 1 : 0;
    if (h->lean) { h->rows_stale = true; h->last_obs = obs_out; }
    const GsFusedChecks fc = fused_checks_args(h);
    const GsRolloutStep rsv = rs ? *rs : GsRolloutStep{};
    if (h->nr2 || h->flow2 || h->nrm) {        // 64 / IW workgroups per 64-instance slab group, each with its own IW instances
      const int per_group = 64 / h->f2_iw, n_wg = h->groups * per_group;
      const dim3 b2(64 * h->f2_nw);
      // (two half-grid launches on two streams, see gs_handle::split_ok; the halves are whole 64-instance slab groups)
      // (per-launch event pairs, gs_timing_enable(1), bracket ONE launch on the main stream: the step stays whole then)
      const bool split = h->split_ok && !h->timing;
      const int n_first = split ? (h->groups / 2) * per_group : n_wg;
      if (split && !h->forked) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        h->forked = true;
      }
      GsF2Tables f2a = h->F2, f2b = h->F2;
      f2a.wg_offset = 0; f2b.wg_offset = n_first;
#define GS_F2(k) do { hipLaunchKernelGGL(k, dim3(n_first), b2, h->F2.lds_bytes, h->stream, h->T, f2a, h->R, h->SC, h->EC, h->slab, h->B, d_actions, h->total_load, pa, fc, rsv); \
                      if (n_first < n_wg) hipLaunchKernelGGL(k, dim3(n_wg - n_first), b2, h->F2.lds_bytes, h->stream2, h->T, f2b, h->R, h->SC, h->EC, h->slab, h->B, d_actions, h->total_load, pa, fc, rsv); } while (0)
      if (h->nrm) { if (fc.enabled) GS_F2(gs_k_stepc_nr_mesh2); else GS_F2(gs_k_step_nr_mesh2); }
      else if (h->nr2) {
        if (h->f2_small) { if (fc.enabled) GS_F2(gs_k_stepc_nr_flow2s); else GS_F2(gs_k_step_nr_flow2s); }
        else { if (fc.enabled) GS_F2(gs_k_stepc_nr_flow2); else GS_F2(gs_k_step_nr_flow2); }
      } else {
        if (h->f2_small) { if (fc.enabled) GS_F2(gs_k_stepc_fbs_flow2s); else GS_F2(gs_k_step_fbs_flow2s); }
        else if (h->f2_wide) { if (fc.enabled) GS_F2(gs_k_stepc_fbs_flow2x); else GS_F2(gs_k_step_fbs_flow2x); }
        else if (h->f2_half) { if (fc.enabled) GS_F2(gs_k_stepc_fbs_flow2h); else GS_F2(gs_k_step_fbs_flow2h); }
#if defined(GS_BUILD_EXPERIMENTS)
        else { if (fc.enabled) GS_F2(gs_k_stepc_fbs_flow2); else GS_F2(gs_k_step_fbs_flow2); }
#endif
      }
#undef GS_F2
      HIPCHK(h, hipGetLastError());
      return GS_OK;
    }
#define GS_STEP(k) hipLaunchKernelGGL(k, grid, block, h->dyn_lds, h->stream, h->T, h->R, h->SC, h->EC, h->slab, h->B, d_actions, h->total_load, pa, fc)
    if (h->solve_kernel == 7) {      // prologue | dense Newton-Raphson, one workgroup per instance | epilogue + observation pack
      GS_STEP(gs_k_pre_nr_dmfma);
      GS_DENSE_LAUNCH(h, h->dense_grid, h->DA, h->B);
      if (fc.enabled) GS_STEP(gs_k_postc_nr_dmfma); else GS_STEP(gs_k_post_nr_dmfma);
    } else
#if defined(GS_BUILD_EXPERIMENTS)
    if (h->solve_kernel == 8) {      // prologue | sparse LU in LDS, one wavefront per instance | epilogue + observation pack
      GS_STEP(gs_k_pre_nr_dmfma);
      hipLaunchKernelGGL(gs_k_nr_sparse_lds, dim3(h->sparse_grid), dim3(64 * h->SA.waves), h->sparse_lds, h->stream, h->SA, h->slab, h->B);
      if (fc.enabled) GS_STEP(gs_k_postc_nr_dmfma); else GS_STEP(gs_k_post_nr_dmfma);
    } else
#endif
    if (fc.enabled) {
      if (h->solve_kernel == 0) GS_STEP(gs_k_stepc_nr_tree);
      else if (h->solve_kernel == 4) GS_STEP(gs_k_stepc_nr_tree_lds);
      else if (h->solve_kernel == 1) GS_STEP(gs_k_stepc_nr_lu);
      else if (h->solve_kernel == 3) GS_STEP(gs_k_stepc_nr_dense);
      else if (h->solve_kernel == 5) GS_STEP(gs_k_stepc_fbs_lds);
      else if (h->solve_kernel == 6) GS_STEP(gs_k_stepc_fbs_flow);
      else GS_STEP(gs_k_stepc_fbs);
    } else {
      if (h->solve_kernel == 0) GS_STEP(gs_k_step_nr_tree);
      else if (h->solve_kernel == 4) GS_STEP(gs_k_step_nr_tree_lds);
      else if (h->solve_kernel == 1) GS_STEP(gs_k_step_nr_lu);
      else if (h->solve_kernel == 3) GS_STEP(gs_k_step_nr_dense);
      else if (h->solve_kernel == 5) GS_STEP(gs_k_step_fbs_lds);
      else if (h->solve_kernel == 6) GS_STEP(gs_k_step_fbs_flow);
      else GS_STEP(gs_k_step_fbs);
    }
#undef GS_STEP
    HIPCHK(h, hipGetLastError()); }
  return GS_OK;     // the observation block was written by the step kernel itself
}

void copy_info(gs_handle* h, double* reward, uint8_t* term, uint8_t* trunc, const gs_info_view* info) {
  const int B = h->B, Bp = h->Bp;
  const double* f = h->h_f; const int32_t* i32 = h->h_i; const uint8_t* u = h->h_u;
  if (reward) memcpy(reward, f + (size_t)SF_REWARD * Bp, B * sizeof(double));
  if (term) memcpy(term, u + (size_t)SU_TERM * Bp, B);
  if (trunc) memcpy(trunc, u + (size_t)SU_TRUNC * Bp, B);
  if (!info) return;
  if (info->power_flow_converged) memcpy(info->power_flow_converged, u + (size_t)SU_CONV * Bp, B);
  if (info->max_voltage) memcpy(info->max_voltage, f + (size_t)SF_VMAX * Bp, B * sizeof(double));
  if (info->min_voltage) memcpy(info->min_voltage, f + (size_t)SF_VMIN * Bp, B * sizeof(double));
  if (info->total_losses) memcpy(info->total_losses, f + (size_t)SF_LOSSES * Bp, B * sizeof(double));
  if (info->violations) memcpy(info->violations, h->h_v4, (size_t)B * 4);      // (the four flags of an instance, interleaved by the kernel)
  if (info->constraint_violations) memcpy(info->constraint_violations, i32 + (size_t)SI_VIOL * Bp, B * sizeof(int32_t));
  if (info->current_step) memcpy(info->current_step, i32 + (size_t)SI_STEP * Bp, B * sizeof(int32_t));
  if (info->episode_reward) memcpy(info->episode_reward, f + (size_t)SF_EPREW * Bp, B * sizeof(double));
  if (info->iterations) memcpy(info->iterations, i32 + (size_t)SI_ITERS * Bp, B * sizeof(int32_t));
  if (info->status) memcpy(info->status, i32 + (size_t)SI_STATUS * Bp, B * sizeof(int32_t));
}


// The first Newton step from the flat start as a constant linear map of the injections (GsF2Tables::mesh_w): for a network whose
// buses other than the slack are all PQ buses,  x = J0^-1 (S_spec - S_calc(flat)) = W [P_spec; 1]  with Q_spec = 0 -- W = the
// angle-equation columns of J0^-1 and the constant term, (2 (n - 1)) x n.  J0: the exact Jacobian (power_flow.py:243-287) at |V| = 1,
// angle 0 (the slack at its set point), inverted by Gauss-Jordan with partial pivoting.  Output in the operand order of
// v_mfma_f64_16x16x4: [tiles row tiles][steps k-steps][64 lanes], A[row = lane & 15][k = lane >> 4], zero-padded.  false: J0 singular
// or the sizes do not fit.
static bool flat_newton_map(const HostTopology& ht, int tiles, int steps, std::vector<double>& wt) {
  const int n_ = ht.n, sl = ht.slack, na = n_ - 1, N2 = 2 * na, K = na + 1;
  if (na < 1 || N2 > 16 * tiles || K > 4 * steps) return false;
  std::vector<double> v0(n_, 1.0);
  if (ht.fixed_v[sl]) v0[sl] = ht.v_set[sl];
  auto act = [&](int i) { return i < sl ? i : i - 1; };
  std::vector<double> Pc(n_, 0.0), Qc(n_, 0.0), J((size_t)N2 * N2, 0.0);
  for (int i = 0; i < n_; ++i)
    for (int q = ht.row_ptr[i]; q < ht.row_ptr[i + 1]; ++q) {
      const int j = ht.col[q];
      const double g = i == j ? ht.Gd[i] : ht.G[q], bq = i == j ? ht.Bd[i] : ht.B[q];
      Pc[i] += v0[i] * v0[j] * g; Qc[i] -= v0[i] * v0[j] * bq;
    }
  for (int i = 0; i < n_; ++i) {
    if (i == sl) continue;
    const int a = act(i);
    const double vi = v0[i];
    J[(size_t)(2 * a) * N2 + 2 * a] = -Qc[i] - vi * vi * ht.Bd[i];
    J[(size_t)(2 * a) * N2 + 2 * a + 1] = Pc[i] / vi + vi * ht.Gd[i];
    J[(size_t)(2 * a + 1) * N2 + 2 * a] = Pc[i] - vi * vi * ht.Gd[i];
    J[(size_t)(2 * a + 1) * N2 + 2 * a + 1] = Qc[i] / vi - vi * ht.Bd[i];
    for (int q = ht.row_ptr[i]; q < ht.row_ptr[i + 1]; ++q) {
      const int j = ht.col[q];
      if (j == i || j == sl) continue;
      const int aj = act(j);
      const double aa = vi * v0[j], gs_bc = -ht.B[q] * aa, gc_bs = ht.G[q] * aa;
      J[(size_t)(2 * a) * N2 + 2 * aj] += gs_bc; J[(size_t)(2 * a) * N2 + 2 * aj + 1] += gc_bs / v0[j];
      J[(size_t)(2 * a + 1) * N2 + 2 * aj] += -gc_bs; J[(size_t)(2 * a + 1) * N2 + 2 * aj + 1] += gs_bc / v0[j];
    }
  }
  std::vector<double> Ji((size_t)N2 * N2, 0.0);
  for (int u = 0; u < N2; ++u) Ji[(size_t)u * N2 + u] = 1.0;
  for (int c = 0; c < N2; ++c) {
    int pr = c;
    for (int r = c + 1; r < N2; ++r) if (std::fabs(J[(size_t)r * N2 + c]) > std::fabs(J[(size_t)pr * N2 + c])) pr = r;
    const double pv = J[(size_t)pr * N2 + c];
    if (!(pv != 0.0) || !std::isfinite(pv)) return false;
    if (pr != c)
      for (int k = 0; k < N2; ++k) { std::swap(J[(size_t)pr * N2 + k], J[(size_t)c * N2 + k]); std::swap(Ji[(size_t)pr * N2 + k], Ji[(size_t)c * N2 + k]); }
    const double ip = 1.0 / pv;
    for (int k = 0; k < N2; ++k) { J[(size_t)c * N2 + k] *= ip; Ji[(size_t)c * N2 + k] *= ip; }
    for (int r = 0; r < N2; ++r) {
      if (r == c) continue;
      const double f = J[(size_t)r * N2 + c];
      if (f == 0.0) continue;
      for (int k = 0; k < N2; ++k) { J[(size_t)r * N2 + k] -= f * J[(size_t)c * N2 + k]; Ji[(size_t)r * N2 + k] -= f * Ji[(size_t)c * N2 + k]; }
    }
  }
  std::vector<double> cst(N2, 0.0);
  for (int u = 0; u < N2; ++u)
    for (int a2 = 0; a2 < na; ++a2) {
      const int bus = a2 < sl ? a2 : a2 + 1;
      cst[u] -= Ji[(size_t)u * N2 + 2 * a2] * Pc[bus] + Ji[(size_t)u * N2 + 2 * a2 + 1] * Qc[bus];
    }
  wt.assign((size_t)tiles * steps * 64, 0.0);
  for (int t = 0; t < tiles; ++t)
    for (int s2 = 0; s2 < steps; ++s2)
      for (int ln = 0; ln < 64; ++ln) {
        const int u = 16 * t + (ln & 15), k = 4 * s2 + (ln >> 4);
        if (u < N2 && k < K) wt[((size_t)t * steps + s2) * 64 + ln] = k < na ? Ji[(size_t)u * N2 + 2 * k] : cst[u];
      }
  return true;
}

}  // namespace

// =============================================================================================
extern "C" {

int gs_version(void) { return GS_ABI_VERSION; }

int gs_build_experiments(void) {
#if defined(GS_BUILD_EXPERIMENTS)
  return 1;
#else
  return 0;
#endif
}

int gs_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

const char* gs_last_error(const gs_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int gs_create(const gs_topology* topo, const gs_config* cfg, int32_t batch, int32_t device,
              int64_t first_instance, gs_handle** out) {
  if (!out) return fail(nullptr, GS_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!topo || !cfg) return fail(nullptr, GS_E_INVALID, "topology / config is NULL");
  if (topo->struct_size != (int32_t)sizeof(gs_topology) || cfg->struct_size != (int32_t)sizeof(gs_config))
    return fail(nullptr, GS_E_INVALID, "struct_size mismatch (ABI %d): topology %d vs %zu, config %d vs %zu",
                GS_ABI_VERSION, topo->struct_size, sizeof(gs_topology), cfg->struct_size, sizeof(gs_config));
  if (batch <= 0) return fail(nullptr, GS_E_INVALID, "batch must be > 0");
  if (cfg->max_iterations < 1) return fail(nullptr, GS_E_INVALID, "max_iterations must be >= 1");
  if (!(cfg->power_base > 0.0)) return fail(nullptr, GS_E_INVALID, "power_base must be > 0");
  if (!(cfg->timestep > 0.0)) return fail(nullptr, GS_E_INVALID, "timestep must be > 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, GS_E_NO_DEVICE, "no HIP device visible: libgridstep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, GS_E_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);

  gs_handle* h = new gs_handle();
  h->device = device;
  h->cfg = *cfg;
  std::string why = gs_compile_topology(*topo, cfg->zero_z_mode, cfg->linear_solver == GS_LINSOLVE_SPARSE_LU,
                                        cfg->solver_kind == GS_SOLVER_NR && cfg->jacobian_mode == GS_JACOBIAN_AS_CODED &&
                                            (cfg->linear_solver == GS_LINSOLVE_AUTO || cfg->linear_solver == GS_LINSOLVE_DENSE_PIVOT),
                                        h->topo);
  if (!why.empty()) { int rc = fail(nullptr, GS_E_INVALID, "topology: %s", why.c_str()); delete h; return rc; }
  const HostTopology& ht = h->topo;
  h->line_x.assign(topo->x, topo->x + topo->m);
  h->B = batch; h->Bp = (batch + 63) / 64 * 64; h->groups = h->Bp / 64;
  int W = cfg->waves_per_group;
  if (const char* e = getenv("GS_WAVES")) W = atoi(e);
  const bool auto_w = W <= 0;
  if (W <= 0) { W = 1; while (W < 16 && h->groups * W * 2 <= 2048) W *= 2; }
  if (W > GS_MAX_WAVES) W = GS_MAX_WAVES;
  h->W = W;
  if (cfg->solver_kind == GS_SOLVER_FBS) {
    if (!ht.fbs_ok) { int rc = fail(nullptr, GS_E_TOPOLOGY, "FBS: %s", ht.fbs_why.c_str()); delete h; return rc; }
    h->solve_kernel = 2;
    const size_t msg_bytes = (size_t)2 * ht.max_level_width * 6 * GS_LANES * sizeof(double);
    if (msg_bytes + 24576 <= 160 * 1024 && !GS_EXPERIMENT_ENV("GS_NO_LDS_TREE")) { h->solve_kernel = 5; h->dyn_lds = msg_bytes; }
    // dataflow sweeps: one 16-byte-per-lane message slot and one flag word per bus in LDS, at most 8 buses per wave
    // (their state lives in registers); flat start only
    const size_t flow_bytes = (size_t)ht.n * 2 * GS_LANES * sizeof(double) + (size_t)ht.n * sizeof(int32_t);
    const int n_items = ht.is_forest ? ht.lvl_ptr[ht.n_levels] : 0;
    // its LDS footprint allows one group per CU whatever W is, so a batch of any size runs it with all 16 waves
    if (auto_w && n_items > 8 * W && n_items <= 8 * GS_MAX_WAVES) { W = GS_MAX_WAVES; h->W = W; }
    if (!cfg->fbs_warm_start && flow_bytes + 24576 <= 160 * 1024 && (n_items + W - 1) / W <= 8 &&
        !getenv("GS_NO_FLOW")) {
      h->solve_kernel = 6; h->dyn_lds = flow_bytes; }
  } else if (cfg->solver_kind == GS_SOLVER_NR) {
    if (cfg->linear_solver == GS_LINSOLVE_TREE && !ht.is_forest) {
      int rc = fail(nullptr, GS_E_TOPOLOGY, "tree elimination requested but the active network has loops"); delete h; return rc; }
    int ls = cfg->linear_solver;
    if (ls == GS_LINSOLVE_AUTO)
      ls = (cfg->jacobian_mode == GS_JACOBIAN_AS_CODED) ? GS_LINSOLVE_DENSE_PIVOT
                                                       : (ht.is_forest ? GS_LINSOLVE_TREE : GS_LINSOLVE_SPARSE_LU);
    // meshed network whose sparse block LU would fill in (more than a quarter of all blocks): dense LU on the matrix cores
    const int na_ = ht.n_active;
    const bool mfma_fits = cfg->jacobian_mode == GS_JACOBIAN_EXACT && na_ >= 1 && 2 * na_ <= 256 && !GS_EXPERIMENT_ENV("GS_NO_DENSE_MFMA");
    if (ls == GS_LINSOLVE_DENSE_MFMA && !mfma_fits) {
      int rc = fail(nullptr, GS_E_TOPOLOGY, "dense_mfma needs the exact Jacobian and at most 128 non-slack buses (have %d)", na_); delete h; return rc; }
    if (cfg->linear_solver == GS_LINSOLVE_AUTO && ls == GS_LINSOLVE_SPARSE_LU && mfma_fits && (long long)ht.lu_n_slots * 4 > (long long)na_ * na_)
      ls = GS_LINSOLVE_DENSE_MFMA;
    // meshed network with few loops: the sparse block LU of an instance in LDS, when its blocks fit beside a second workgroup's
    // (one instance: its blocks + 7 doubles per bus; the shared schedule is about 2.5 x the blocks in bytes: two instances at least)
    const size_t sparse_need = ((size_t)4 * (ht.lu_n_slots + ht.n) + (size_t)7 * ht.n) * sizeof(double);
    const bool sparse_fits = ht.has_lu && !ht.is_forest && ht.lu_n_piv > 0 && ht.n <= 256 && sparse_need <= 32 * 1024;
#if !defined(GS_BUILD_EXPERIMENTS)
    if (ls == GS_LINSOLVE_SPARSE_LDS) {
      int rc = fail(nullptr, GS_E_INVALID, "linear_solver sparse_lds is an experiment (measured, never AUTO's choice): build the library with `make EXPERIMENTS=1`"); delete h; return rc; }
#endif
    if (ls == GS_LINSOLVE_SPARSE_LDS && !sparse_fits) {
      int rc = fail(nullptr, GS_E_TOPOLOGY, "sparse_lds needs a meshed network of at most 256 buses whose block LU fits 32 KB of LDS (%zu bytes here)", sparse_need); delete h; return rc; }
    // (AUTO does not take it: measured on the 123-bus feeder with 26 loops it reaches 12.9 M env-steps/s against the slab-row
    // kernel's 17.7 M -- four instances per CU, each a chain of 7-to-40-lane steps, lose to 64 instances per workgroup on full
    // lanes, bytes or not; DESIGN.md section 7.  GS_SPARSE_LDS_AUTO=1 makes AUTO take it, for measurements.)
    if (cfg->linear_solver == GS_LINSOLVE_AUTO && ls == GS_LINSOLVE_SPARSE_LU && sparse_fits && GS_EXPERIMENT_ENV("GS_SPARSE_LDS_AUTO"))
      ls = GS_LINSOLVE_SPARSE_LDS;
    h->solve_kernel = (ls == GS_LINSOLVE_TREE) ? 0 : (ls == GS_LINSOLVE_SPARSE_LU) ? 1 : (ls == GS_LINSOLVE_DENSE_MFMA) ? 7 : (ls == GS_LINSOLVE_SPARSE_LDS) ? 8 : 3;
    // forest sweeps through LDS messages when two adjacent levels fit next to the 24 KB static block
    const size_t msg_bytes = (size_t)2 * ht.max_level_width * 6 * GS_LANES * sizeof(double);
    if (h->solve_kernel == 0 && msg_bytes + 24576 <= 160 * 1024 && !GS_EXPERIMENT_ENV("GS_NO_LDS_TREE")) h->solve_kernel = 4;
    if (h->solve_kernel == 4) h->dyn_lds = msg_bytes;
  } else { int rc = fail(nullptr, GS_E_INVALID, "unknown solver_kind %d", cfg->solver_kind); delete h; return rc; }
  // the epilogue's cross-wave partials need 48 KB; the observation pack stages two or three 64-column tiles behind them
  h->dyn_lds = std::max<size_t>(49152 + 2 * 64 * 65 * sizeof(double), h->dyn_lds);

  h->n = ht.n; h->m = ht.m; h->n_loads = topo->n_loads; h->n_gens = topo->n_gens; h->n_bats = topo->n_bats;
  h->obs_dim = 2 * h->n + 2 * h->m + 1 + 2 * h->n_loads + h->n_gens + 2 * h->n_bats;     // grid_env.py:307-314
  h->action_dim = h->n_bats + h->n_gens;                                                    // grid_env.py:351
  h->state_dim = 12 + 2 * h->n_bats + h->n_gens + 2 * h->n + 2 * h->m;
  auto bail = [&](int rc) { gs_destroy(h); return rc; };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(nullptr, GS_E_HIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(nullptr, GS_E_HIP, "hipStreamCreate failed"));

  {
    const void* fns[] = {(const void*)gs_k_nr_tree, (const void*)gs_k_step_nr_tree, (const void*)gs_k_nr_tree_lds,
                         (const void*)gs_k_step_nr_tree_lds, (const void*)gs_k_nr_lu, (const void*)gs_k_step_nr_lu,
                         (const void*)gs_k_nr_dense, (const void*)gs_k_step_nr_dense, (const void*)gs_k_fbs,
                         (const void*)gs_k_step_fbs, (const void*)gs_k_fbs_lds, (const void*)gs_k_step_fbs_lds,
                         (const void*)gs_k_stepc_nr_tree, (const void*)gs_k_stepc_nr_tree_lds, (const void*)gs_k_stepc_nr_lu,
                         (const void*)gs_k_stepc_nr_dense, (const void*)gs_k_stepc_fbs, (const void*)gs_k_stepc_fbs_lds,
                         (const void*)gs_k_fbs_flow, (const void*)gs_k_step_fbs_flow, (const void*)gs_k_stepc_fbs_flow,
                         (const void*)gs_k_pre_nr_dmfma, (const void*)gs_k_post_nr_dmfma, (const void*)gs_k_postc_nr_dmfma,
                         (const void*)gs_k_posts_nr_dmfma};
    // the attribute is per function, i.e. shared by every handle of the process: always raise it to
    // the most any handle may ask for (160 KB per workgroup minus the 24 KB static block)
    const int max_dyn = 160 * 1024 - 24576;
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn) != hipSuccess)
        return bail(fail(nullptr, GS_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", max_dyn));
    for (const void* f : {
#if defined(GS_BUILD_EXPERIMENTS)
                          (const void*)gs_k_step_fbs_flow2, (const void*)gs_k_stepc_fbs_flow2,
#endif
                          (const void*)gs_k_step_nr_flow2,
                          (const void*)gs_k_stepc_nr_flow2, (const void*)gs_k_step_fbs_flow2s, (const void*)gs_k_stepc_fbs_flow2s,
                          (const void*)gs_k_step_nr_flow2s, (const void*)gs_k_stepc_nr_flow2s, (const void*)gs_k_step_fbs_flow2h,
                          (const void*)gs_k_stepc_fbs_flow2h, (const void*)gs_k_step_fbs_flow2x, (const void*)gs_k_stepc_fbs_flow2x,
                          (const void*)gs_k_step_nr_mesh2, (const void*)gs_k_stepc_nr_mesh2})      // no static LDS in these
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return bail(fail(nullptr, GS_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", 160 * 1024));
  }

  {
    hipError_t e = hipFuncSetAttribute((const void*)gs_k_nr_dense_mfma2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
#if defined(GS_BUILD_EXPERIMENTS)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gs_k_nr_dense_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
#endif
    if (e != hipSuccess) return bail(fail(nullptr, GS_E_HIP, "hipFuncSetAttribute(gs_k_nr_dense_mfma): %s", hipGetErrorString(e)));
  }

  // ---- rows ----
  GsRows& R = h->R;
  int r = 0;
  auto take = [&](int count) { int at = r; r += count; return at; };
  auto take_even = [&](int count) { r = (r + 1) & ~1; return take(count); };          // blocks whose entries pair up
  auto take_pair = [&](GsFam2& a, GsFam2& b2, int count) { r = (r + 1) & ~1; a.base = r; b2.base = r + 1; r += 2 * count; };
  const int n = h->n, m = h->m;
  take_pair(R.P, R.Q, n); take_pair(R.VM, R.VA, n); take_pair(R.FLOW, R.ENVLOAD, m); R.LOAD = take(m);
  R.LOSSES = take(1); R.MAXMIS = take(1); R.ITERS = take(1); R.CONV = take(1); R.STATUS = take(1);
  // scratch rows are allocated only for the kernel that uses them: the slab is what the step
  // streams through L2 / Infinity Cache, so every unused row costs residency
  const int sk = h->solve_kernel;
  const bool k_tree = sk == 0, k_lu = sk == 1, k_fbs = sk == 2 || sk == 5 || sk == 6, k_dense = sk == 3, k_tree_lds = sk == 4;
  const bool k_rhs = k_tree || k_lu || k_dense;
  take_pair(R.E, R.F, n); take_pair(R.PC, R.QC, n);
  take_pair(R.R0, R.R1, k_rhs ? n : 0); take_pair(R.X0, R.X1, k_rhs ? n : 0);
  R.RVM = take(n);
  R.SV = take_even(k_tree || k_tree_lds ? 2 * n : 0); R.QV = take_even(k_tree ? 2 * n : 0);
  R.TB = take_even(k_tree || k_tree_lds ? 4 * n : 0); R.CB = take_even(k_tree ? 4 * n : 0);
  take_pair(R.JR, R.JI, k_fbs ? n : 0);
  R.LU = take_even(h->solve_kernel == 1 ? 4 * ht.lu_n_slots : 0);
  R.LUD = take_even(h->solve_kernel == 1 ? 4 * n : 0);
  const int dnN = ht.dn_N;
  R.DA = take(h->solve_kernel == 3 ? dnN * dnN : 0);
  R.DB = take(h->solve_kernel == 3 ? dnN : 0); R.DX = take(h->solve_kernel == 3 ? dnN : 0);
  R.DPERM = take(h->solve_kernel == 3 ? dnN : 0);
  R.TIME = take(1); R.STEP = take(1); R.VIOL = take(1); R.TOTLOSS = take(1); R.EPREW = take(1); R.FREQ = take(1);
  R.IRR = take(1); R.WIND = take(1); R.TEMP = take(1); R.CLOUD = take(1); R.SEEDLO = take(1); R.SEEDHI = take(1);
  R.SOC = take(h->n_bats); R.BATP = take(h->n_bats); R.CURT = take(h->n_gens); R.GENP = take(h->n_gens);
  R.REWARD = take(1); R.TERM = take(1); R.TRUNC = take(1); R.VMAX = take(1); R.VMIN = take(1); R.VFLAGS = take(4);
  R.ACT = take(h->action_dim); R.LOADP = take_even(h->n_loads + 1);   // written in pairs by the load-noise draws
  R.total = (r + 1) & ~1;        // rows are stored in pairs (GS_ELEM)

  // ---- per-wave work lists of the forest sweeps (records in the order each wave meets them) ----
  std::vector<GsItemRec> witems;
  std::vector<int32_t> wl_ptr(h->W + 1, 0), ovf_slot;
  if (ht.is_forest) {
    const int maxw = ht.max_level_width;
    const bool flow = h->solve_kernel == 6;      // messages by bus index, items dealt for equal item counts per wave
    std::vector<int> owner(ht.lvl_ptr[ht.n_levels], 0);
    {
      std::vector<int> load(h->W, 0);
      for (int lv = 0; lv < ht.n_levels; ++lv)
        for (int t = ht.lvl_ptr[lv]; t < ht.lvl_ptr[lv + 1]; ++t) {
          int w = (t - ht.lvl_ptr[lv]) % h->W;
          if (flow) { w = 0; for (int v = 1; v < h->W; ++v) if (load[v] < load[w]) w = v; }
          owner[t] = w; ++load[w];
        }
    }
    for (int w = 0; w < h->W; ++w) {
      wl_ptr[w] = (int)witems.size();
      for (int lv = 0; lv < ht.n_levels; ++lv)
        for (int t = ht.lvl_ptr[lv]; t < ht.lvl_ptr[lv + 1]; ++t) {
          if (owner[t] != w) continue;
          GsItemRec r{};
          const int i = ht.lvl_bus[t], p = ht.parent[i];
          r.bus = i; r.parent = p; r.level = lv;
          r.slot = (lv & 1) * maxw + (t - ht.lvl_ptr[lv]);
          r.parent_slot = p >= 0 ? ((lv + 1) & 1) * maxw + ht.lvl_pos[p] : 0;
          r.flags = (ht.th_free[i] ? 1 : 0) | (ht.vm_free[i] ? 2 : 0) |
                    (p >= 0 && ht.th_free[p] ? 4 : 0) | (p >= 0 && ht.vm_free[p] ? 8 : 0);
          r.n_children = ht.child_ptr[i + 1] - ht.child_ptr[i];
          r.ovf0 = (int)ovf_slot.size();
          for (int q = 0; q < r.n_children; ++q) {
            const int ch = ht.child_idx[ht.child_ptr[i] + q];
            const int cs = flow ? ch : ((lv - 1) & 1) * maxw + ht.lvl_pos[ch];
            if (q < GS_ITEM_CHILDREN) r.child_slot[q] = cs; else ovf_slot.push_back(cs);
          }
          if (p >= 0) { r.g = ht.G[ht.parent_pos[i]]; r.b = ht.B[ht.parent_pos[i]]; }
          r.gd = ht.Gd[i]; r.bd = ht.Bd[i];
          if (h->solve_kernel == 5 || flow) {       // FBS flavour: parent includes the slack, (g, b) := z = 1 / y
            const int fp = ht.fbs_parent[i], pos = ht.fbs_parent_pos[i];
            const double yr = -ht.G[pos], yi = -ht.B[pos], yd = yr * yr + yi * yi;
            r.g = yr / yd; r.b = -yi / yd;
            r.gd = yr; r.bd = yi;
            if (p < 0) { r.parent = fp; r.flags |= 16; }
          }
          witems.push_back(r);
        }
    }
    wl_ptr[h->W] = (int)witems.size();
  }

  std::vector<GsInjRec> winj;
  std::vector<int32_t> wi_ptr(h->W + 1, 0);
  for (int w = 0; w < h->W; ++w) {
    wi_ptr[w] = (int)winj.size();
    // dataflow sweep kernel: a wave builds the injections of the buses it solves, in item order, straight into the
    // solver's registers (a bus that is nobody's item -- the slack -- needs no injection there)
    std::vector<int> mine;
    if (h->solve_kernel == 6) for (int k = wl_ptr[w]; k < wl_ptr[w + 1]; ++k) mine.push_back(witems[k].bus);
    else for (int i = w; i < ht.n; i += h->W) mine.push_back(i);
    for (int i : mine) {
      GsInjRec r{};
      r.bus = i;
      r.nl = ht.bl_ptr[i + 1] - ht.bl_ptr[i]; r.ng = ht.bg_ptr[i + 1] - ht.bg_ptr[i]; r.nb = ht.bb_ptr[i + 1] - ht.bb_ptr[i];
      r.generic = (r.nl > 2 || r.ng > 2 || r.nb > 2) ? 1 : 0;
      if (r.nl > 0) r.l0 = ht.bl_idx[ht.bl_ptr[i]];
      if (r.nl > 1) r.l1 = ht.bl_idx[ht.bl_ptr[i] + 1];
      if (r.ng > 0) r.g0 = ht.bg_idx[ht.bg_ptr[i]];
      if (r.ng > 1) r.g1 = ht.bg_idx[ht.bg_ptr[i] + 1];
      if (r.nb > 0) r.b0 = ht.bb_idx[ht.bb_ptr[i]];
      if (r.nb > 1) r.b1 = ht.bb_idx[ht.bb_ptr[i] + 1];
      winj.push_back(r);
    }
  }
  wi_ptr[h->W] = (int)winj.size();

  // mismatch records: each bus' Ybus row in chunks of GS_ELL_K entries (same entry order as the
  // CSR row); buses are dealt to the waves longest row first so that every wave gets about the
  // same number of records
  std::vector<GsBusRec> wbus;
  std::vector<int32_t> wb_ptr(h->W + 1, 0);
  {
    std::vector<int> order(ht.n), nrec(ht.n);
    for (int i = 0; i < ht.n; ++i) { order[i] = i; nrec[i] = std::max(1, (ht.row_ptr[i + 1] - ht.row_ptr[i] + GS_ELL_K - 1) / GS_ELL_K); }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return nrec[a] > nrec[b2]; });
    std::vector<std::vector<int>> mine(h->W);
    std::vector<int> load(h->W, 0);
    for (int i : order) {
      int best = 0;
      for (int w = 1; w < h->W; ++w) if (load[w] < load[best]) best = w;
      mine[best].push_back(i); load[best] += nrec[i];
    }
    for (int w = 0; w < h->W; ++w) {
      wb_ptr[w] = (int)wbus.size();
      std::sort(mine[w].begin(), mine[w].end());
      for (int i : mine[w]) {
        const int p0 = ht.row_ptr[i], p1 = ht.row_ptr[i + 1];
        for (int c0 = 0; c0 < nrec[i]; ++c0) {
          GsBusRec r{};
          r.bus = i;
          r.flags = (ht.th_free[i] ? 1 : 0) | (ht.vm_free[i] ? 2 : 0) | (c0 + 1 < nrec[i] ? 4 : 0) | (c0 > 0 ? 8 : 0);
          for (int k = 0; k < GS_ELL_K; ++k) {
            const int p = p0 + c0 * GS_ELL_K + k;
            if (p < p1) { r.col[k] = ht.col[p]; r.G[k] = ht.G[p]; r.B[k] = ht.B[p]; }
            else { r.col[k] = i; r.G[k] = 0.0; r.B[k] = 0.0; }
          }
          wbus.push_back(r);
        }
      }
    }
    wb_ptr[h->W] = (int)wbus.size();
  }

  // ---- second-generation step kernels (kernels_flow2.hip): IW instances per workgroup, NW waves, NI buses per sub-group ----
  std::vector<GsF2Rec> f2recs; std::vector<int32_t> f2anc; std::vector<double> f2z;
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  // LDS carve-up shared by the members of the family; returns the total
  auto f2_layout = [&](GsF2Tables& F, int NW, int IW, size_t second_region_min, size_t n_table_ints, int zcols, size_t z_bytes = 0) {
    const int nsl = ht.n + 3;
    const size_t SB = (size_t)(IW + 1) * 16;
    size_t off = up16((size_t)nsl * SB);
    F.off_tile = (int32_t)off;
    off += up16(std::max<size_t>({(size_t)nsl * SB, second_region_min, (size_t)ht.m * SB, (size_t)(topo->n_loads + 4) * IW * sizeof(double)}));
    F.off_anc = (int32_t)off; off += up16(n_table_ints * 4);
    F.off_z = (int32_t)off; off += up16(std::max((size_t)nsl * zcols * 8, z_bytes));
    F.off_prof = (int32_t)off; off += up16(24 * sizeof(double));
    F.env_genp = 0; F.env_curt = topo->n_gens; F.env_batp = 2 * topo->n_gens; F.env_soc = 2 * topo->n_gens + topo->n_bats;
    F.off_env = (int32_t)off; off += up16((size_t)(2 * topo->n_gens + 2 * topo->n_bats + 1) * IW * sizeof(double));
    F.off_red = (int32_t)off; off += 2 * (size_t)NW * IW * sizeof(double);
    F.off_atom = (int32_t)off; off += 8 * (size_t)IW * sizeof(unsigned long long) + 16 * (size_t)IW * sizeof(uint32_t);
    F.lds_bytes = (int32_t)off; F.n_slots = nsl; F.slack = ht.slack;
    return off;
  };
  auto f2_devices = [&](GsF2Rec& r, int i) {
    r.nl = ht.bl_ptr[i + 1] - ht.bl_ptr[i]; r.ng = ht.bg_ptr[i + 1] - ht.bg_ptr[i]; r.nb = ht.bb_ptr[i + 1] - ht.bb_ptr[i];
    if (r.nl > 0) r.l0 = ht.bl_idx[ht.bl_ptr[i]];
    if (r.nl > 1) r.l1 = ht.bl_idx[ht.bl_ptr[i] + 1];
    if (r.ng > 0) r.g0 = ht.bg_idx[ht.bg_ptr[i]];
    if (r.ng > 1) r.g1 = ht.bg_idx[ht.bg_ptr[i] + 1];
    if (r.nb > 0) r.b0 = ht.bb_idx[ht.bb_ptr[i]];
    if (r.nb > 1) r.b1 = ht.bb_idx[ht.bb_ptr[i] + 1];
  };
  int max_dev = 0, max_ch = 0;
  for (int i = 0; i < ht.n; ++i) {
    max_dev = std::max({max_dev, ht.bl_ptr[i + 1] - ht.bl_ptr[i], ht.bg_ptr[i + 1] - ht.bg_ptr[i], ht.bb_ptr[i + 1] - ht.bb_ptr[i]});
    if (ht.is_forest) max_ch = std::max(max_ch, ht.child_ptr[i + 1] - ht.child_ptr[i]);
  }
  const int SL_ZERO = ht.n, SL_ONE = ht.n + 1, SL_DUMMY = ht.n + 2, nsl = ht.n + 3;

  // -- sweep solver: one record per position of the preorder of the tree below the slack
  // (eligible wherever the first-generation dataflow kernel is, and -- with the number of waves left to the library -- for
  // feeders beyond its 128 buses)
  if (h->solve_kernel == 6 || (cfg->solver_kind == GS_SOLVER_FBS && auto_w && !cfg->fbs_warm_start && ht.is_forest && ht.fbs_ok &&
                               ht.lvl_ptr[ht.n_levels] > 8 * GS_MAX_WAVES && !getenv("GS_NO_FLOW"))) {
    std::string& why = h->flow2_why;
    std::vector<int> order, size(ht.n, 1), depth(ht.n, 0);
    {
      std::vector<std::vector<int>> kids(ht.n);
      std::vector<int> roots;
      for (int lv = ht.n_levels - 1; lv >= 0; --lv)
        for (int t = ht.lvl_ptr[lv]; t < ht.lvl_ptr[lv + 1]; ++t) {
          const int i = ht.lvl_bus[t], fp = ht.fbs_parent[i];
          if (fp == ht.slack) roots.push_back(i); else kids[fp].push_back(i);
        }
      std::sort(roots.begin(), roots.end());
      for (auto& k : kids) std::sort(k.begin(), k.end());
      std::vector<std::pair<int, int>> stack;
      for (int ri = (int)roots.size() - 1; ri >= 0; --ri) stack.push_back({roots[ri], 1});
      while (!stack.empty()) {
        auto [i, d] = stack.back(); stack.pop_back();
        order.push_back(i); depth[i] = d;
        for (int q = (int)kids[i].size() - 1; q >= 0; --q) stack.push_back({kids[i][q], d + 1});
      }
      for (int p = (int)order.size() - 1; p >= 0; --p) { const int i = order[p], fp = ht.fbs_parent[i]; if (fp != ht.slack) size[fp] += size[i]; }
    }
    const int N = (int)order.size();
    int max_depth = 1;
    for (int i : order) max_depth = std::max(max_depth, depth[i]);
    // forward sweep by pointer jumping, radix 4: round r adds the partial sums of the ancestors 4^r, 2 * 4^r and 3 * 4^r up
    int n_jump = 0;
    while ((1 << (2 * n_jump)) < max_depth) ++n_jump;
    n_jump = std::max(2, (n_jump + 1) & ~1);                         // even: the last round then reads the second buffer
    // small feeders: 8 instances per workgroup, the eight sub-groups of a wavefront on eight buses
    const bool small = N <= GS_F2S_WAVES * (64 / GS_F2S_IW) * GS_F2S_ITEMS && !getenv("GS_NO_FLOW2_SMALL");
    // default: 16 instances per workgroup, two workgroups per CU (GS_FLOW2_IW=32 asks for the 32-instance member, one per CU)
    const bool wide = !small && N > GS_F2_WAVES * 2 * GS_F2_ITEMS;          // 129 ... 256 buses: eight buses per sub-group
    const bool half = !small && !wide && !(GS_EXPERIMENT_ENV("GS_FLOW2_IW") && atoi(GS_EXPERIMENT_ENV("GS_FLOW2_IW")) == 32);
    const int NW = small ? GS_F2S_WAVES : wide ? GS_F2X_WAVES : half ? GS_F2H_WAVES : GS_F2_WAVES,
              NI = small ? GS_F2S_ITEMS : wide ? GS_F2X_ITEMS : half ? GS_F2H_ITEMS : GS_F2_ITEMS,
              IW = small ? GS_F2S_IW : (wide || half) ? GS_F2H_IW : 32;
    const int NPOS = NW * (64 / IW) * NI;
    GsF2Tables& F = h->F2;
    const size_t off = f2_layout(F, NW, IW, 0, (size_t)n_jump * nsl * 4, 2);
    F.n_jump = n_jump;
    if (getenv("GS_NO_FLOW2")) why = "disabled by GS_NO_FLOW2";
    else if (N > NPOS) why = "more than " + std::to_string(NPOS) + " buses below the slack";
    // (the second-generation sweeps hold their stopping criterion, the summed mismatch, in 2^-44 pu fixed point: below ~1e-10 the
    // threshold is a handful of units and every lane's rounding shows; the first-generation kernels compare in double precision)
    else if (!(cfg->tolerance >= 1e-10)) why = "tolerance below 1e-10";
    else if (N != ht.lvl_ptr[ht.n_levels]) why = "part of the forest does not hang off the slack bus";
    else if (max_dev > 2) why = "more than two devices of a kind at one bus";
    else if (off > 160 * 1024) why = "LDS tables do not fit";
    else if (ht.n < 2 || ht.m < 1 || N < 1) why = "trivial network";
    if (why.empty()) {
      h->flow2 = true; h->f2_small = small; h->f2_half = half; h->f2_wide = wide; h->f2_iw = IW; h->f2_nw = NW;
      GsF2Rec idle{}; idle.bus = SL_DUMMY; idle.parent = SL_ONE; idle.last = SL_DUMMY;
      f2recs.assign((size_t)NPOS, idle);
      f2z.assign((size_t)nsl * 2, 0.0);
      f2anc.assign((size_t)n_jump * nsl * 4, SL_ZERO);            // [round][slot][4]: the slot's ancestors 1, 2, 3 steps of 4^round up (no ancestor: ZERO)
      std::vector<int> up1((size_t)nsl, SL_ZERO);                    // parent slot of every slot (the slack's children: ZERO)
      for (int p = 0; p < N; ++p) {
        GsF2Rec& r = f2recs[p];
        const int i = order[p];
        const int fp = ht.fbs_parent[i], pos = ht.fbs_parent_pos[i];
        const double yr = -ht.G[pos], yi = -ht.B[pos], yd = yr * yr + yi * yi;      // branch admittance = -Y_ip; z = 1 / y
        r.bus = i; r.parent = fp; r.flags = 1 | (fp == ht.slack ? 2 : 0); r.last = order[p + size[i] - 1]; r.level = depth[i];
        r.zr = yr / yd; r.zi = -yi / yd; r.yr = yr; r.yi = yi;
        f2z[2 * (size_t)i] = r.zr; f2z[2 * (size_t)i + 1] = r.zi;
        up1[i] = fp == ht.slack ? SL_ZERO : fp;
        f2_devices(r, i);
      }
      {
        std::vector<int> step = up1;                                 // ancestor 4^round steps up
        for (int r = 0; r < n_jump; ++r) {
          for (int sidx = 0; sidx < nsl; ++sidx) {
            int a = sidx;
            for (int k = 0; k < 3; ++k) { a = step[a]; f2anc[((size_t)r * nsl + sidx) * 4 + k] = a; }
          }
          std::vector<int> nxt((size_t)nsl);
          for (int sidx = 0; sidx < nsl; ++sidx) nxt[sidx] = step[step[step[step[sidx]]]];
          step.swap(nxt);
        }
      }
    }
  }

  // -- Newton-Raphson: every (wave, item) holds a group of HV = 64 / IW buses of ONE level of the tree
  if (h->solve_kernel == 4 && ht.fbs_ok) {
    std::string& why = h->flow2_why;
    bool all_pq = true, off_slack = true;
    for (int i = 0; i < ht.n; ++i) {
      if (i != ht.slack && ht.lvl_pos[i] >= 0 && !(ht.th_free[i] && ht.vm_free[i])) all_pq = false;
      if (i != ht.slack && ht.lvl_pos[i] < 0) off_slack = false;                       // a bus outside the forest
      if (ht.lvl_pos[i] >= 0 && ht.parent[i] < 0 && ht.fbs_parent[i] != ht.slack) off_slack = false;
    }
    auto deal = [&](int NW, int HV, std::vector<std::vector<std::vector<int>>>& mine, std::vector<std::vector<int>>& mine_lv) {
      mine.assign(NW, {}); mine_lv.assign(NW, {});
      for (int lv = 0; lv < ht.n_levels; ++lv)
        for (int t = ht.lvl_ptr[lv]; t < ht.lvl_ptr[lv + 1]; t += HV) {
          int w = 0;
          for (int v = 1; v < NW; ++v) if (mine[v].size() < mine[w].size()) w = v;
          std::vector<int> grp;
          for (int q = 0; q < HV; ++q) grp.push_back(t + q < ht.lvl_ptr[lv + 1] ? ht.lvl_bus[t + q] : -1);
          mine[w].push_back(grp); mine_lv[w].push_back(lv);
        }
      int mx = 0;
      for (auto& v : mine) mx = std::max<int>(mx, (int)v.size());
      return mx;
    };
    std::vector<std::vector<std::vector<int>>> mine; std::vector<std::vector<int>> mine_lv;
    bool small = !getenv("GS_NO_FLOW2_SMALL") && deal(GS_F2NS_WAVES, 64 / GS_F2S_IW, mine, mine_lv) <= GS_F2NS_ITEMS;
    const int NW = small ? GS_F2NS_WAVES : GS_F2N_WAVES, NI = small ? GS_F2NS_ITEMS : GS_F2N_ITEMS, IW = small ? GS_F2S_IW : 32, HV = 64 / IW;
    const int max_items = deal(NW, HV, mine, mine_lv);
    const int NPOS = NW * HV * NI, maxw = ht.max_level_width;
    h->f2_npos = NPOS;
    GsF2Tables& F = h->F2;
    // the ring's zero entry: behind the ring's two parities and behind the K slots that share the region
    const size_t ring_entry = (size_t)3 * IW * 16;
    const int ring_zero = (int)std::max<size_t>((size_t)2 * maxw, ((size_t)nsl * (IW + 1) * 16 + ring_entry - 1) / ring_entry);
    const size_t ring_bytes = (size_t)(ring_zero + 1) * ring_entry;
    const int pos_off = (2 * (ht.n + 1) * GS_F2_CHILDREN + nsl + 3) & ~3;
    const int n_ints = pos_off + NPOS * 4;
    const size_t off = f2_layout(F, NW, IW, ring_bytes, (size_t)n_ints, 4);
    F.n_jump = 0; F.n_levels = ht.n_levels; F.pos_off = pos_off; F.n_anc_ints = n_ints; F.ring_zero = ring_zero;
    if (getenv("GS_NO_FLOW2")) why = "disabled by GS_NO_FLOW2";
    else if (!h->SC.jacobian_exact && cfg->jacobian_mode != GS_JACOBIAN_EXACT) why = "as-coded Jacobian";
    else if (!all_pq) why = "a bus below the slack is not a PQ bus";
    else if (!off_slack) why = "part of the network does not hang off the slack bus";
    else if (max_items > NI) why = "more than " + std::to_string(NI) + " bus groups per wave";
    else if (max_ch > GS_F2_CHILDREN) why = "a bus has more than " + std::to_string(GS_F2_CHILDREN) + " children";
    else if (max_dev > 2) why = "more than two devices of a kind at one bus";
    else if (off > 160 * 1024) why = "LDS tables do not fit";
    else if (ht.n < 2 || ht.m < 1) why = "trivial network";
    if (why.empty()) {
      h->nr2 = true; h->f2_small = small; h->f2_iw = IW; h->f2_nw = NW;
      GsF2Rec idle{}; idle.bus = SL_DUMMY; idle.parent = SL_ONE; idle.last = SL_DUMMY; idle.level = -1;
      f2recs.assign((size_t)NPOS, idle);
      f2z.assign((size_t)nsl * 4, 0.0);
      f2anc.assign((size_t)n_ints, 0);
      // rows of n + 1 buses (row n: idle positions); entries beyond a bus's children name the ZERO slot / the ring's zero entry
      int32_t* child_bus = f2anc.data(); int32_t* child_ring = child_bus + (ht.n + 1) * GS_F2_CHILDREN; int32_t* nch = child_ring + (ht.n + 1) * GS_F2_CHILDREN;
      std::fill(child_bus, child_bus + (ht.n + 1) * GS_F2_CHILDREN, SL_ZERO);
      std::fill(child_ring, child_ring + (ht.n + 1) * GS_F2_CHILDREN, ring_zero);
      int32_t* pos_tab = f2anc.data() + pos_off;
      std::vector<int> level_of(ht.n, 0);
      for (int lv = 0; lv < ht.n_levels; ++lv) for (int t = ht.lvl_ptr[lv]; t < ht.lvl_ptr[lv + 1]; ++t) level_of[ht.lvl_bus[t]] = lv;
      auto ring_of = [&](int i) { return (level_of[i] & 1) * maxw + ht.lvl_pos[i]; };
      for (int i = 0; i < ht.n; ++i) {
        nch[i] = ht.child_ptr[i + 1] - ht.child_ptr[i];
        for (int q = ht.child_ptr[i]; q < ht.child_ptr[i + 1]; ++q) {
          const int c = ht.child_idx[q];
          child_bus[i * GS_F2_CHILDREN + (q - ht.child_ptr[i])] = c; child_ring[i * GS_F2_CHILDREN + (q - ht.child_ptr[i])] = ring_of(c);
        }
        if (ht.lvl_pos[i] >= 0) {
          const int pos = ht.fbs_parent_pos[i];
          f2z[4 * (size_t)i] = ht.G[pos]; f2z[4 * (size_t)i + 1] = ht.B[pos]; f2z[4 * (size_t)i + 2] = ht.Gd[i]; f2z[4 * (size_t)i + 3] = ht.Bd[i];
        }
      }
      for (int p = 0; p < NPOS; ++p) { pos_tab[4 * p] = SL_DUMMY; pos_tab[4 * p + 1] = SL_ONE; pos_tab[4 * p + 2] = 0; pos_tab[4 * p + 3] = 0; }
      for (int w = 0; w < NW; ++w)
        for (int j = 0; j < (int)mine[w].size(); ++j) {
          int grp_maxch = 0;
          for (int i : mine[w][j]) if (i >= 0) grp_maxch = std::max(grp_maxch, ht.child_ptr[i + 1] - ht.child_ptr[i]);
          for (int hh = 0; hh < HV; ++hh) {
            const int p = (w * HV + hh) * NI + j;
            GsF2Rec& r = f2recs[p];
            r.level = mine_lv[w][j]; r.pad1 = grp_maxch;        // most children of the group's buses
            const int i = mine[w][j][hh];
            if (i < 0) continue;
            const int fp = ht.fbs_parent[i];
            r.bus = i; r.parent = fp; r.flags = 1 | (fp == ht.slack ? 2 : 0); r.last = i;
            pos_tab[4 * p] = i; pos_tab[4 * p + 1] = fp; pos_tab[4 * p + 2] = ring_of(i); pos_tab[4 * p + 3] = fp == ht.slack ? 0 : ring_of(fp);
            f2_devices(r, i);
          }
        }
    }
  }

  // -- Newton-Raphson on a meshed feeder: the block LU as rows of lane items (mesh_schedule.h), 8 instances per workgroup
  std::vector<int32_t> mesh_items, mesh_rowinfo;
  if (h->solve_kernel == 1 && !ht.is_forest) {
    std::string& why = h->mesh_why;
    const int NW = GS_F2M_WAVES, NI = GS_F2M_ITEMS, IW = GS_F2S_IW, HV = 64 / IW;
    bool all_pq = true;
    for (int i = 0; i < ht.n; ++i) if (i != ht.slack && !(ht.th_free[i] && ht.vm_free[i])) all_pq = false;
    GsF2Tables& F = h->F2;
    MeshSchedule S;
    if (getenv("GS_NO_FLOW2") || getenv("GS_NO_MESH2")) why = "disabled by GS_NO_FLOW2 / GS_NO_MESH2";
    else if (cfg->jacobian_mode != GS_JACOBIAN_EXACT) why = "as-coded Jacobian";
    else if (!all_pq) why = "a bus other than the slack is not a PQ bus";
    else if (ht.fixed_v[ht.slack] == 0) why = "no typed slack bus";
    else if (max_dev > 2) why = "more than two devices of a kind at one bus";
    else if (ht.n < 2 || ht.m < 1) why = "trivial network";
    else {
      const int off_tile = (int)up16((size_t)nsl * (IW + 1) * 16);          // where f2_layout puts the region (below)
      // message units that leave room for a second workgroup on the CU: 80 KB less everything else the workgroup keeps in LDS
      // (an estimate: the Ybus tables' size is known only from the schedule; f2_layout below decides)
      const size_t fixed = up16((size_t)nsl * (IW + 1) * 16) + (size_t)6 * 16 * IW + (size_t)NW * 16 * 16 * IW + (size_t)nsl * IW * 8 +
                           (size_t)(ht.nnz + 8) * 8 + (size_t)(ht.nnz + nsl + 4) * 16 + 4096;
      const int unit_budget = fixed < 80 * 1024 ? (int)((80 * 1024 - fixed) / (16 * IW)) : 1;
      gs_mesh_schedule(ht, NW, NI, IW, off_tile, (IW + 1) * 16, GS_MESH_ACC, unit_budget, S);
      if (!S.ok) why = S.why;
    }
    if (why.empty()) {
      // ints staged at off_anc: every bus's neighbour list; doubles at off_z: the Ybus entries of the pairs, then of the diagonal per slot
      size_t off = f2_layout(F, NW, IW, (size_t)S.region_bytes, S.adj_ent.size(), 0, S.ytab.size() * sizeof(double));
      F.off_scr = (int32_t)off; off += (size_t)NW * 16 * 16 * IW;           // exchange scratch: 16 units per wave
      F.mesh_off_p = (int32_t)off; off += (size_t)nsl * IW * sizeof(double);   // P_spec by voltage slot
      F.lds_bytes = (int32_t)off;
      if (off > 160 * 1024) why = "LDS tables do not fit";
    }
    if (why.empty()) {
      h->nrm = true; h->f2_small = false; h->f2_iw = IW; h->f2_nw = NW; h->f2_npos = NW * HV * NI;
      // ---- iteration 0 as a matrix product (GsF2Tables::mesh_w): the flat-start Jacobian, inverted once on the host
      if (!getenv("GS_NR_NO_FLAT") && ht.n <= 128) {
        std::vector<double> wt;
        if (flat_newton_map(ht, 16, 32, wt)) {
          { const int rcw = dev_upload(h, &F.mesh_w, wt); if (rcw) return bail(rcw); }
          F.mesh_w_steps = 32; F.mesh_slack = ht.slack;
        }
      }
      h->mesh_levels = S.n_levels; h->mesh_rows = S.n_rows; h->mesh_units = S.msg_units; h->mesh_messages = S.n_messages; h->mesh_accs = S.n_accumulators;
      F.n_jump = 0; F.n_levels = S.n_levels; F.pos_off = 0; F.n_anc_ints = (int32_t)S.adj_ent.size(); F.ring_zero = 0;
      F.mesh_nz = (int32_t)S.ytab.size(); F.mesh_pairs = S.n_pairs;
      f2anc = S.adj_ent; f2z = S.ytab;
      GsF2Rec idle{}; idle.bus = SL_DUMMY; idle.parent = SL_ONE; idle.last = SL_DUMMY; idle.level = -1;
      f2recs.assign((size_t)NW * HV * NI, idle);
      for (int w = 0; w < NW; ++w) for (int j = 0; j < NI; ++j) for (int hh = 0; hh < HV; ++hh) {
        const GsMeshItem& it = S.items[((size_t)w * NI + j) * HV + hh];
        if (!(it.flags & GS_MESH_F_PIVOT)) continue;
        GsF2Rec& r = f2recs[((size_t)w * HV + hh) * NI + j];        // position of (wave, sub-group, row) in the frame's numbering
        r.bus = it.bus; r.parent = SL_ONE; r.flags = 1; r.last = it.bus; r.level = S.rowinfo[((size_t)w * NI + j) * 4];
        f2_devices(r, it.bus);
      }
      mesh_items = S.packed; mesh_rowinfo = S.rowinfo_packed;
    }
  }

  // ---- level schedule of the sparse block LU for this handle's W waves (kernels_solve.hip, linsolve_lu) ----
  std::vector<int32_t> lu_a_ptr, lu_a, lu_b_ptr, lu_b, lu_c_ptr, lu_c, lu_r_ptr, lu_r;
  if (ht.has_lu) {
    const int NL = ht.lu_n_levels, Wn = h->W;
    std::vector<std::vector<std::vector<int32_t>>> A(Wn, std::vector<std::vector<int32_t>>(NL)), Bs(Wn, std::vector<std::vector<int32_t>>(NL)),
        Cs(Wn, std::vector<std::vector<int32_t>>(NL)), Rs(Wn, std::vector<std::vector<int32_t>>(NL));
    for (int L = 0; L < NL; ++L) {
      // phase A: one item per (pivot, neighbour), plus one per pivot for the singularity test; dealt round-robin
      int turn = 0;
      std::map<int32_t, std::vector<std::pair<int32_t, int32_t>>> tgt;        // target code -> updates
      for (int t = 0; t < ht.lu_n_piv; ++t) {
        if (ht.lu_piv_level[t] != L) continue;
        const int k = ht.lu_piv_bus[t];
        { auto& a = A[turn++ % Wn][L]; a.push_back(k); a.push_back(-1); }
        for (int q = ht.lu_nb_ptr[t]; q < ht.lu_nb_ptr[t + 1]; ++q) {
          auto& a = A[turn++ % Wn][L]; a.push_back(k); a.push_back(ht.lu_nb_jk[q]);
          tgt[-(1 + ht.n + ht.lu_nb_bus[q])].push_back({ht.lu_nb_jk[q], k});          // r_i -= (A_ik D_k^-1) r_k
        }
        for (int q = ht.lu_pair_ptr[t]; q < ht.lu_pair_ptr[t + 1]; ++q) tgt[ht.lu_pair_ij[q]].push_back({ht.lu_pair_ik[q], ht.lu_pair_kj[q]});
      }
      // phase B: targets dealt to the wave with the fewest updates so far in this level
      std::vector<int> load(Wn, 0);
      std::vector<std::pair<int32_t, std::vector<std::pair<int32_t, int32_t>>>> order(tgt.begin(), tgt.end());
      std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.second.size() > y.second.size(); });
      for (auto& e : order) {
        int w = 0;
        for (int v = 1; v < Wn; ++v) if (load[v] < load[w]) w = v;
        load[w] += (int)e.second.size() + 1;
        auto& b = Bs[w][L];
        b.push_back(e.first); b.push_back((int32_t)e.second.size());
        for (auto& u : e.second) { b.push_back(u.first); b.push_back(u.second); }
        if (e.first < -ht.n) {      // the right-hand-side records alone: all iteration 0 needs (GsTables::lu_flat)
          auto& rr = Rs[w][L];
          rr.push_back(e.first); rr.push_back((int32_t)e.second.size());
          for (auto& u : e.second) { rr.push_back(u.first); rr.push_back(u.second); }
        }
      }
      // phase C: the level's pivots, round-robin
      int tc = 0;
      for (int t = 0; t < ht.lu_n_piv; ++t) if (ht.lu_piv_level[t] == L) Cs[tc++ % Wn][L].push_back(t);
    }
    auto flatten = [&](std::vector<std::vector<std::vector<int32_t>>>& X, std::vector<int32_t>& ptr, std::vector<int32_t>& flat, int unit) {
      for (int w = 0; w < Wn; ++w) {
        for (int L = 0; L < NL; ++L) { ptr.push_back((int32_t)flat.size() / unit); flat.insert(flat.end(), X[w][L].begin(), X[w][L].end()); }
        ptr.push_back((int32_t)flat.size() / unit);
      }
    };
    flatten(A, lu_a_ptr, lu_a, 2); flatten(Bs, lu_b_ptr, lu_b, 1); flatten(Cs, lu_c_ptr, lu_c, 1); flatten(Rs, lu_r_ptr, lu_r, 1);
  }

  // ---- tables ----
  GsTables& T = h->T;
  T.n = n; T.m = m; T.nnz = ht.nnz; T.n_levels = ht.n_levels;
  T.n_loads = h->n_loads; T.n_gens = h->n_gens; T.n_bats = h->n_bats;
  T.lu_n_piv = ht.lu_n_piv; T.lu_n_slots = ht.lu_n_slots; T.lu_n_orig = ht.lu_n_orig; T.lu_n_levels = ht.lu_n_levels;
  T.dn_N = ht.dn_N;
  int rc = 0;
#define UP(field, vec) if ((rc = dev_upload(h, &T.field, ht.vec))) return bail(rc)
  UP(ell_col, ell_col); UP(ell_G, ell_G); UP(ell_B, ell_B); UP(rem_ptr, rem_ptr); UP(rem_col, rem_col);
  UP(rem_G, rem_G); UP(rem_B, rem_B);
  UP(row_ptr, row_ptr); UP(col, col); UP(G, G); UP(Bv, B); UP(Gd, Gd); UP(Bd, Bd);
  UP(th_free, th_free); UP(vm_free, vm_free); UP(v_set, v_set); UP(fixed_v, fixed_v);
  UP(lvl_ptr, lvl_ptr); UP(lvl_bus, lvl_bus); UP(parent, parent); UP(parent_pos, parent_pos);
  UP(child_ptr, child_ptr); UP(child_idx, child_idx); UP(lvl_pos, lvl_pos);
  if ((rc = dev_upload(h, &T.winj, winj)) || (rc = dev_upload(h, &T.wi_ptr, wi_ptr))) return bail(rc);
  if ((rc = dev_upload(h, &T.wbus, wbus)) || (rc = dev_upload(h, &T.wb_ptr, wb_ptr))) return bail(rc);
  if ((rc = dev_upload(h, &T.witems, witems)) || (rc = dev_upload(h, &T.wl_ptr, wl_ptr)) || (rc = dev_upload(h, &T.ovf_slot, ovf_slot))) return bail(rc);
  T.max_level_width = ht.max_level_width;
  UP(fbs_parent, fbs_parent); UP(fbs_parent_pos, fbs_parent_pos);
  UP(lfrom, lfrom); UP(lto, lto); UP(lyr, lyr); UP(lyi, lyi); UP(lrating, lrating); UP(lrating_inv, lrating_inv);
  UP(lu_piv_bus, lu_piv_bus); UP(lu_nb_ptr, lu_nb_ptr); UP(lu_nb_bus, lu_nb_bus); UP(lu_nb_kj, lu_nb_kj);
  UP(lu_nb_jk, lu_nb_jk); UP(lu_pair_ptr, lu_pair_ptr); UP(lu_pair_ik, lu_pair_ik); UP(lu_pair_kj, lu_pair_kj);
  UP(lu_pair_ij, lu_pair_ij); UP(lu_orig_slot, lu_orig_slot); UP(lu_orig_i, lu_orig_i); UP(lu_orig_j, lu_orig_j);
  UP(lu_orig_pos, lu_orig_pos);
  if ((rc = dev_upload(h, &T.lu_a_ptr, lu_a_ptr)) || (rc = dev_upload(h, &T.lu_a, lu_a)) || (rc = dev_upload(h, &T.lu_b_ptr, lu_b_ptr)) ||
      (rc = dev_upload(h, &T.lu_b, lu_b)) || (rc = dev_upload(h, &T.lu_c_ptr, lu_c_ptr)) || (rc = dev_upload(h, &T.lu_c, lu_c)) ||
      (rc = dev_upload(h, &T.lu_r_ptr, lu_r_ptr)) || (rc = dev_upload(h, &T.lu_r, lu_r))) return bail(rc);
  UP(dn_th_idx, dn_th_idx); UP(dn_vm_idx, dn_vm_idx);
  UP(bl_ptr, bl_ptr); UP(bl_idx, bl_idx); UP(bg_ptr, bg_ptr); UP(bg_idx, bg_idx); UP(bb_ptr, bb_ptr); UP(bb_idx, bb_idx);
  UP(load_base, load_base); UP(load_q, load_q); UP(gen_kind, gen_kind); UP(gen_cap, gen_cap); UP(gen_p0, gen_p0);
  UP(gen_p1, gen_p1); UP(gen_p2, gen_p2); UP(bat_cap, bat_cap); UP(bat_rating, bat_rating); UP(bat_eff, bat_eff);
#undef UP
  // ---- dense block LU on the matrix cores (kernels_dense.hip): unknown numbering, Jacobian blocks by column panel, scratch ----
  if (h->solve_kernel == 7) {
    GsDenseArgs& D = h->DA;
    std::vector<int32_t> act_bus, act_of(ht.n, -1);
    for (int i = 0; i < ht.n; ++i) if (ht.th_free[i] || ht.vm_free[i]) { act_of[i] = (int32_t)act_bus.size(); act_bus.push_back(i); }
    const int na = (int)act_bus.size(), NB = (2 * na + 63) / 64;
    // (panel by panel, block row by block row inside a panel: the panel form reads a panel's range, the block-row form a block's)
    std::vector<int32_t> ent_ptr(NB + 1, 0), bent_ptr((size_t)NB * NB + 1, 0), ent;
    std::vector<GsDenseEntry> bent;
    for (int pnl = 0; pnl < NB; ++pnl) {
      ent_ptr[pnl] = (int32_t)ent.size() / 3;
      for (int blk = 0; blk < NB; ++blk) {
        bent_ptr[(size_t)pnl * NB + blk] = (int32_t)ent.size() / 3;
        for (int i = 0; i < ht.n; ++i) {
          if (act_of[i] < 0 || (2 * act_of[i]) / 64 != blk) continue;
          for (int q = ht.row_ptr[i]; q < ht.row_ptr[i + 1]; ++q) {
            const int j = ht.col[q];
            if (act_of[j] < 0 || (2 * act_of[j]) / 64 != pnl) continue;
            ent.push_back(i); ent.push_back(j); ent.push_back(q);
            GsDenseEntry e{};
            e.ib = i; e.jb = j;
            e.dst = ((2 * act_of[i] - 64 * blk) * 66 + (2 * act_of[j] - 64 * pnl)) | (ht.th_free[i] ? 1 << 16 : 0) | (ht.vm_free[i] ? 1 << 17 : 0) |
                    (ht.th_free[j] ? 1 << 18 : 0) | (ht.vm_free[j] ? 1 << 19 : 0);
            e.g = i == j ? ht.Gd[i] : ht.G[q]; e.b = i == j ? ht.Bd[i] : ht.B[q];
            bent.push_back(e);
          }
        }
      }
    }
    if (bent.empty()) bent.push_back(GsDenseEntry{});
    ent_ptr[NB] = bent_ptr[(size_t)NB * NB] = (int32_t)ent.size() / 3;
    D.n = ht.n; D.na = na; D.NB = NB; D.max_it = cfg->max_iterations; D.jacobian_exact = 1; D.rows_total = h->R.total;
    D.tol = cfg->tolerance; D.alpha = cfg->acceleration_factor;
    if ((rc = dev_upload(h, &D.act_bus, act_bus)) || (rc = dev_upload(h, &D.act_of, act_of)) || (rc = dev_upload(h, &D.ent_ptr, ent_ptr)) ||
        (rc = dev_upload(h, &D.ent, ent)) || (rc = dev_upload(h, &D.bent_ptr, bent_ptr)) || (rc = dev_upload(h, &D.bent, bent))) return bail(rc);
    D.row_ptr = T.row_ptr; D.col = T.col; D.G = T.G; D.Bv = T.Bv; D.Gd = T.Gd; D.Bd = T.Bd;
    D.th_free = T.th_free; D.vm_free = T.vm_free; D.fixed_v = T.fixed_v; D.v_set = T.v_set;
    D.R = h->R;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    // persistent grid, instances strided over it.  Block-row form: two block buffers in LDS, two workgroups per CU; panel form (GS_DENSE_PANEL=1
    // in a build with the experiments): the whole 64-column panel in LDS, one workgroup per CU
    const size_t NP = (size_t)64 * NB;
    h->dense_blockrow = !GS_EXPERIMENT_ENV("GS_DENSE_PANEL");
    if (h->dense_blockrow) {
      h->dense_grid = std::max(1, std::min(h->B, 2 * cus));
      h->dense_lds = ((size_t)2 * 64 * 66 + NP + (size_t)8 * ((ht.n + 1) & ~1) + 8) * sizeof(double);
    } else {
      h->dense_grid = std::max(1, std::min(h->B, cus));
      h->dense_lds = (NP * 66 + NP + (size_t)8 * ((ht.n + 1) & ~1) + 2 * 528 + 8) * sizeof(double);
    }
    if (h->dense_lds > 160 * 1024 - 256) return bail(fail(nullptr, GS_E_TOPOLOGY, "dense_mfma: %zu bytes of LDS needed", h->dense_lds));
    double* scratch = nullptr;
    if ((rc = dev_alloc(h, &scratch, (size_t)h->dense_grid * NB * NB * 64 * 64))) return bail(rc);
    D.scratch = scratch;
    // the flat-start Jacobian is the same for every instance: factor it once, here, with the solver kernel itself
    // (bit-identical to what iteration 0 of every solve would compute; GS_DENSE_NO_FLAT=1 keeps it per solve)
    if (!getenv("GS_DENSE_NO_FLAT")) {
      double* flat = nullptr;
      if ((rc = dev_alloc(h, &flat, (size_t)NB * NB * 64 * 64 + 8))) return bail(rc);
      GsDenseArgs once = D;
      once.flat = flat; once.mode = 1; once.max_it = 1;
      GS_DENSE_LAUNCH(h, 1, once, 1);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(fail(nullptr, GS_E_HIP, "dense_mfma: factorisation of the flat-start Jacobian failed"));
      D.flat = flat;
      // ---- and its inverse, for iteration 0 as one product (GsDenseArgs::jinv_t): the same entries as the kernel's assembly
      // (power_flow.py:243-287, exact sign; fixed components and padding unknowns: identity rows and columns), flat start
      if (h->dense_blockrow) {
        const int n_ = ht.n, NPd = 64 * NB;
        std::vector<double> v0(n_, 1.0), Pc(n_, 0.0), Qc(n_, 0.0), Jm((size_t)NPd * NPd, 0.0), Ji((size_t)NPd * NPd, 0.0);
        for (int i = 0; i < n_; ++i) if (ht.fixed_v[i]) v0[i] = ht.v_set[i];
        for (int i = 0; i < n_; ++i)
          for (int q = ht.row_ptr[i]; q < ht.row_ptr[i + 1]; ++q) {
            const int j = ht.col[q];
            Pc[i] += v0[i] * v0[j] * ht.G[q]; Qc[i] -= v0[i] * v0[j] * ht.B[q];
          }
        for (int u = 0; u < NPd; ++u) Jm[(size_t)u * NPd + u] = 1.0;              // padding / fixed components
        for (int i = 0; i < n_; ++i) {
          const int a = act_of[i];
          if (a < 0) continue;
          const bool thi = ht.th_free[i] != 0, vfi = ht.vm_free[i] != 0;
          const double vi = v0[i], vvb = vi * vi * ht.Bd[i];
          Jm[(size_t)(2 * a) * NPd + 2 * a] = thi ? (-Qc[i] - vvb) : 1.0;
          Jm[(size_t)(2 * a) * NPd + 2 * a + 1] = (thi && vfi) ? (Pc[i] / vi + vi * ht.Gd[i]) : 0.0;
          Jm[(size_t)(2 * a + 1) * NPd + 2 * a] = (thi && vfi) ? (Pc[i] - vi * vi * ht.Gd[i]) : 0.0;
          Jm[(size_t)(2 * a + 1) * NPd + 2 * a + 1] = vfi ? (Qc[i] / vi - vi * ht.Bd[i]) : 1.0;
          for (int q = ht.row_ptr[i]; q < ht.row_ptr[i + 1]; ++q) {
            const int j = ht.col[q];
            if (j == i || act_of[j] < 0) continue;
            const int aj = act_of[j];
            const bool thj = ht.th_free[j] != 0, vfj = ht.vm_free[j] != 0;
            const double aa = vi * v0[j], gs_bc = -ht.B[q] * aa, gc_bs = ht.G[q] * aa;
            if (thi && thj) Jm[(size_t)(2 * a) * NPd + 2 * aj] = gs_bc;
            if (thi && vfj) Jm[(size_t)(2 * a) * NPd + 2 * aj + 1] = gc_bs / v0[j];
            if (vfi && thj) Jm[(size_t)(2 * a + 1) * NPd + 2 * aj] = -gc_bs;
            if (vfi && vfj) Jm[(size_t)(2 * a + 1) * NPd + 2 * aj + 1] = gs_bc / v0[j];
          }
        }
        for (int u = 0; u < NPd; ++u) Ji[(size_t)u * NPd + u] = 1.0;
        bool ok = true;
        for (int c = 0; c < NPd && ok; ++c) {
          int pr = c;
          for (int r = c + 1; r < NPd; ++r) if (std::fabs(Jm[(size_t)r * NPd + c]) > std::fabs(Jm[(size_t)pr * NPd + c])) pr = r;
          const double pv = Jm[(size_t)pr * NPd + c];
          if (!(pv != 0.0) || !std::isfinite(pv)) { ok = false; break; }
          if (pr != c)
            for (int k = 0; k < NPd; ++k) { std::swap(Jm[(size_t)pr * NPd + k], Jm[(size_t)c * NPd + k]); std::swap(Ji[(size_t)pr * NPd + k], Ji[(size_t)c * NPd + k]); }
          const double ip = 1.0 / pv;
          for (int k = 0; k < NPd; ++k) { Jm[(size_t)c * NPd + k] *= ip; Ji[(size_t)c * NPd + k] *= ip; }
          for (int r = 0; r < NPd; ++r) {
            if (r == c) continue;
            const double f = Jm[(size_t)r * NPd + c];
            if (f == 0.0) continue;
            for (int k = 0; k < NPd; ++k) { Jm[(size_t)r * NPd + k] -= f * Jm[(size_t)c * NPd + k]; Ji[(size_t)r * NPd + k] -= f * Ji[(size_t)c * NPd + k]; }
          }
        }
        if (ok) {
          std::vector<double> jt((size_t)NPd * NPd);
          for (int u = 0; u < NPd; ++u)
            for (int c = 0; c < NPd; ++c) jt[(size_t)c * NPd + u] = Ji[(size_t)u * NPd + c];
          if ((rc = dev_upload(h, &D.jinv_t, jt))) return bail(rc);
        }
      }
    }
  }
  // a step as two half-grid launches on two streams: only where each half still gives every CU a workgroup
  h->lean = (h->flow2 || h->nr2 || h->nrm) && !getenv("GS_EAGER_ROWS");
  if ((h->flow2 || h->nr2 || h->nrm) && 2 * (size_t)h->F2.lds_bytes <= 160 * 1024 && !getenv("GS_NO_SPLIT") && h->groups * (64 / h->f2_iw) >= 512 &&
      h->groups >= 2) {
    if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)
      return bail(fail(nullptr, GS_E_HIP, "second step stream: hipStreamCreate / hipEventCreate failed"));
    h->split_ok = true;
  }

  if (h->nrm) {
    if ((rc = dev_upload(h, &h->F2.mesh_items, mesh_items)) || (rc = dev_upload(h, &h->F2.mesh_rowinfo, mesh_rowinfo))) return bail(rc);
  }
  if (h->flow2 || h->nr2 || h->nrm) {
    // buses with a voltage set point, for the kernels' flat start (the slack; the first entry travels inside the argument block)
    std::vector<int32_t> fs_slot; std::vector<double> fs_val;
    for (int i = 0; i < ht.n; ++i) if (ht.fixed_v[i]) { fs_slot.push_back(i); fs_val.push_back(ht.v_set[i]); }
    h->F2.n_fixed = (int32_t)fs_slot.size();
    h->F2.fixed_slot0 = fs_slot.empty() ? 0 : fs_slot[0]; h->F2.fixed_val0 = fs_val.empty() ? 1.0 : fs_val[0];
    h->F2.fixed_slot = nullptr; h->F2.fixed_val = nullptr;
    if ((rc = dev_upload(h, &h->F2.recs, f2recs)) || (rc = dev_upload(h, &h->F2.anc, f2anc)) || (rc = dev_upload(h, &h->F2.zbus, f2z)) ||
        (fs_slot.size() > 1 && ((rc = dev_upload(h, &h->F2.fixed_slot, fs_slot)) || (rc = dev_upload(h, &h->F2.fixed_val, fs_val))))) return bail(rc);
  }

  // ---- configs ----
  h->SC.tolerance = cfg->tolerance; h->SC.alpha = cfg->acceleration_factor;
  h->SC.max_iterations = cfg->max_iterations; h->SC.jacobian_exact = (cfg->jacobian_mode == GS_JACOBIAN_EXACT);
  GsEnvCfg& E = h->EC;
  E.timestep = cfg->timestep; E.v_min = cfg->v_min; E.v_max = cfg->v_max; E.f_min = cfg->f_min; E.f_max = cfg->f_max;
  E.safety_penalty = cfg->safety_penalty; E.H = cfg->inertia_H; E.D = cfg->damping_D; E.f0 = cfg->f_nominal;
  E.power_base = cfg->power_base; E.inv_power_base = 1.0 / cfg->power_base; E.episode_length = cfg->episode_length; E.stochastic_loads = cfg->stochastic_loads; E.fbs_warm_start = cfg->fbs_warm_start;
  E.weather_variation = cfg->weather_variation; E.first_instance = first_instance;
  // sum(load.active_power) in list order, starting from 0 like python's sum() (grid_env.py:744)
  h->total_load = 0.0;
  for (int l = 0; l < h->n_loads; ++l) h->total_load += topo->load_base[l];

  // ---- layout maps ----
  std::vector<int32_t> mo, mvm(n), mva(n), mfl(m), mld(m), mp(n), mq(n), mact(h->action_dim), mst;
  std::vector<double> cst;
  for (int i = 0; i < n; ++i) { mo.push_back(R.VM + i); mo.push_back(R.VA + i); }          // grid_env.py:758-759
  for (int k = 0; k < m; ++k) { mo.push_back(R.FLOW + k); mo.push_back(R.ENVLOAD + k); }   // :762-763
  mo.push_back(R.FREQ);                                                                     // :766
  for (int l = 0; l < h->n_loads; ++l) {                                                    // :769-770 (static values)
    cst.push_back(ht.load_base[l]); mo.push_back(-(int)cst.size());
    cst.push_back(ht.load_q[l]); mo.push_back(-(int)cst.size());
  }
  for (int g = 0; g < h->n_gens; ++g) mo.push_back(R.GENP + g);                              // :773-777
  for (int q = 0; q < h->n_bats; ++q) { mo.push_back(R.SOC + q); mo.push_back(R.BATP + q); } // :780-781
  for (int i = 0; i < n; ++i) { mvm[i] = R.VM + i; mva[i] = R.VA + i; mp[i] = R.P + i; mq[i] = R.Q + i; }
  for (int k = 0; k < m; ++k) { mfl[k] = R.FLOW + k; mld[k] = R.LOAD + k; }
  for (int a = 0; a < h->action_dim; ++a) mact[a] = R.ACT + a;
  for (int s : {R.TIME, R.STEP, R.VIOL, R.TOTLOSS, R.EPREW, R.FREQ, R.IRR, R.WIND, R.TEMP, R.CLOUD, R.SEEDLO, R.SEEDHI}) mst.push_back(s);
  for (int q = 0; q < h->n_bats; ++q) mst.push_back(R.SOC + q);
  for (int q = 0; q < h->n_bats; ++q) mst.push_back(R.BATP + q);
  for (int g = 0; g < h->n_gens; ++g) mst.push_back(R.CURT + g);
  for (int i = 0; i < n; ++i) mst.push_back(R.VM + i);
  for (int i = 0; i < n; ++i) mst.push_back(R.VA + i);
  for (int k = 0; k < m; ++k) mst.push_back(R.FLOW + k);
  for (int k = 0; k < m; ++k) mst.push_back(R.ENVLOAD + k);
  {   // the constants of an observation form one block (the static load powers); the step kernel skips it
    int c0 = 0;
    while (c0 < (int)mo.size() && mo[c0] >= 0) ++c0;
    int c1 = c0;
    while (c1 < (int)mo.size() && mo[c1] < 0) ++c1;
    bool one_block = true;
    for (int c = c1; c < (int)mo.size(); ++c) one_block = one_block && mo[c] >= 0;
    if (one_block && !GS_EXPERIMENT_ENV("GS_PACK_ALL_COLUMNS")) { h->obs_skip0 = c0; h->obs_skip1 = c1; }
  }
  if ((int)mo.size() != h->obs_dim || (int)mst.size() != h->state_dim)
    return bail(fail(nullptr, GS_E_INVALID, "internal: layout map size mismatch"));
  { const double* q = nullptr; if ((rc = dev_upload(h, &q, cst))) return bail(rc); h->d_cst = const_cast<double*>(q); }
  if ((rc = upload_map(h, &h->map_obs, mo)) || (rc = upload_map(h, &h->map_vm, mvm)) || (rc = upload_map(h, &h->map_va, mva)) ||
      (rc = upload_map(h, &h->map_flow, mfl)) || (rc = upload_map(h, &h->map_load, mld)) || (rc = upload_map(h, &h->map_p, mp)) ||
      (rc = upload_map(h, &h->map_q, mq)) || (rc = upload_map(h, &h->map_act, mact)) || (rc = upload_map(h, &h->map_state, mst)))
    return bail(rc);
  std::vector<int32_t> rf(SF_COUNT), ri(SI_COUNT), ru(SU_COUNT);
  rf[SF_REWARD] = R.REWARD; rf[SF_VMAX] = R.VMAX; rf[SF_VMIN] = R.VMIN; rf[SF_LOSSES] = R.LOSSES; rf[SF_EPREW] = R.EPREW; rf[SF_MAXMIS] = R.MAXMIS;
  ri[SI_VIOL] = R.VIOL; ri[SI_STEP] = R.STEP; ri[SI_ITERS] = R.ITERS; ri[SI_STATUS] = R.STATUS;
  ru[SU_TERM] = R.TERM; ru[SU_TRUNC] = R.TRUNC; ru[SU_CONV] = R.CONV;
  for (int v = 0; v < 4; ++v) ru[SU_VF0 + v] = R.VFLAGS + v;
  if ((rc = upload_map(h, &h->rows_f, rf)) || (rc = upload_map(h, &h->rows_i, ri)) || (rc = upload_map(h, &h->rows_u, ru))) return bail(rc);

  // ---- big buffers ----
  const size_t slab_doubles = (size_t)h->groups * R.total * GS_LANES;
  if ((rc = dev_alloc(h, &h->slab, slab_doubles))) return bail(rc);
  if (hipMemset(h->slab, 0, slab_doubles * sizeof(double)) != hipSuccess) return bail(fail(nullptr, GS_E_HIP, "hipMemset(slab) failed"));
  const size_t widest = std::max<size_t>({(size_t)h->obs_dim, (size_t)h->state_dim, (size_t)n, (size_t)m, (size_t)h->action_dim, 1});
  h->in_doubles = (size_t)h->B * widest; h->out_doubles = (size_t)h->B * widest;
  if ((rc = dev_alloc(h, &h->d_in, h->in_doubles)) || (rc = dev_alloc(h, &h->d_out, h->out_doubles)) ||
      (rc = dev_alloc(h, &h->d_obs2[0], (size_t)h->Bp * h->obs_dim)) || (rc = dev_alloc(h, &h->d_obs2[1], (size_t)h->Bp * h->obs_dim))) return bail(rc);
  if ((rc = dev_alloc(h, &h->sc_f, (size_t)SF_COUNT * h->Bp)) || (rc = dev_alloc(h, &h->sc_i, (size_t)SI_COUNT * h->Bp)) ||
      (rc = dev_alloc(h, &h->sc_u, (size_t)SU_COUNT * h->Bp)) || (rc = dev_alloc(h, &h->d_seeds, (size_t)h->B)) ||
      (rc = dev_alloc(h, &h->d_mask, (size_t)h->B)))
    return bail(rc);
  {
    const size_t nf = (size_t)SF_COUNT * h->Bp * sizeof(double), ni = (size_t)SI_COUNT * h->Bp * sizeof(int32_t), nv = (size_t)h->Bp * sizeof(uint32_t),
                 nu = (size_t)SU_COUNT * h->Bp;
    void* dp = nullptr;
    if (hipHostMalloc(&h->h_pin, nf + ni + nv + nu, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, h->h_pin, 0) != hipSuccess) {
      (void)hipGetLastError();
      return bail(fail(nullptr, GS_E_NOMEM, "hipHostMalloc(%zu bytes, mapped) for the per-instance scalars failed", nf + ni + nv + nu));
    }
    memset(h->h_pin, 0, nf + ni + nv + nu);
    char* hp = (char*)h->h_pin; char* dv = (char*)dp;
    h->h_f = (double*)hp; h->h_i = (int32_t*)(hp + nf); h->h_v4 = (uint32_t*)(hp + nf + ni); h->h_u = (uint8_t*)(hp + nf + ni + nv);
    h->hd_f = (double*)dv; h->hd_i = (int32_t*)(dv + nf); h->hd_v4 = (uint32_t*)(dv + nf + ni); h->hd_u = (uint8_t*)(dv + nf + ni + nv);
  }
#if defined(GS_BUILD_EXPERIMENTS)
  // ---- sparse block LU in LDS (kernels_sparse.hip): the level schedule without the split over waves, the flat-start factors ----
  if (h->solve_kernel == 8) {
    GsSparseArgs& Sp = h->SA;
    const int NL = ht.lu_n_levels;
    std::vector<int32_t> a_ptr{0}, a, b_ptr{0}, b_rec, b_pair, r_ptr{0}, r_rec, r_pair, c_ptr{0}, cc;
    for (int L = 0; L < NL; ++L) {
      std::map<int32_t, std::vector<std::pair<int32_t, int32_t>>> tgt;        // target code -> updates, as for linsolve_lu
      for (int t = 0; t < ht.lu_n_piv; ++t) {
        if (ht.lu_piv_level[t] != L) continue;
        const int k = ht.lu_piv_bus[t];
        a.push_back(k); a.push_back(-1);
        for (int q = ht.lu_nb_ptr[t]; q < ht.lu_nb_ptr[t + 1]; ++q) {
          a.push_back(k); a.push_back(ht.lu_nb_jk[q]);
          tgt[-(1 + ht.n + ht.lu_nb_bus[q])].push_back({ht.lu_nb_jk[q], k});
        }
        for (int q = ht.lu_pair_ptr[t]; q < ht.lu_pair_ptr[t + 1]; ++q) tgt[ht.lu_pair_ij[q]].push_back({ht.lu_pair_ik[q], ht.lu_pair_kj[q]});
        cc.push_back(t);
      }
      // longest records first: the lanes of one pass then carry records of similar length
      std::vector<std::pair<int32_t, std::vector<std::pair<int32_t, int32_t>>>> order(tgt.begin(), tgt.end());
      std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.second.size() > y.second.size(); });
      for (auto& e : order) {
        b_rec.push_back(e.first); b_rec.push_back((int32_t)e.second.size()); b_rec.push_back((int32_t)b_pair.size() / 2);
        for (auto& u : e.second) { b_pair.push_back(u.first); b_pair.push_back(u.second); }
        if (e.first < -ht.n) {
          r_rec.push_back(e.first); r_rec.push_back((int32_t)e.second.size()); r_rec.push_back((int32_t)r_pair.size() / 2);
          for (auto& u : e.second) { r_pair.push_back(u.first); r_pair.push_back(u.second); }
        }
      }
      a_ptr.push_back((int32_t)a.size() / 2); b_ptr.push_back((int32_t)b_rec.size() / 3); r_ptr.push_back((int32_t)r_rec.size() / 3);
      c_ptr.push_back((int32_t)cc.size());
    }
    Sp.n = ht.n; Sp.n_slots = ht.lu_n_slots; Sp.n_orig = ht.lu_n_orig; Sp.n_piv = ht.lu_n_piv; Sp.n_levels = NL;
    Sp.max_it = cfg->max_iterations; Sp.jacobian_exact = cfg->jacobian_mode == GS_JACOBIAN_EXACT ? 1 : 0; Sp.rows_total = h->R.total;
    Sp.tol = cfg->tolerance; Sp.alpha = cfg->acceleration_factor;
    // one packed copy of everything the elimination chases pointers through: staged into LDS once per workgroup
    std::vector<int32_t> ipack; std::vector<double> dpack;
    auto addi = [&](const std::vector<int32_t>& v) { const int32_t o = (int32_t)ipack.size(); ipack.insert(ipack.end(), v.begin(), v.end()); return o; };
    auto addd = [&](const std::vector<double>& v) { const int32_t o = (int32_t)dpack.size(); dpack.insert(dpack.end(), v.begin(), v.end()); return o; };
    Sp.o_row_ptr = addi(ht.row_ptr); Sp.o_col = addi(ht.col); Sp.o_th_free = addi(ht.th_free); Sp.o_vm_free = addi(ht.vm_free); Sp.o_fixed_v = addi(ht.fixed_v);
    Sp.o_piv_bus = addi(ht.lu_piv_bus); Sp.o_nb_ptr = addi(ht.lu_nb_ptr); Sp.o_nb_bus = addi(ht.lu_nb_bus); Sp.o_nb_kj = addi(ht.lu_nb_kj);
    Sp.o_a_ptr = addi(a_ptr); Sp.o_a = addi(a); Sp.o_b_ptr = addi(b_ptr); Sp.o_b_rec = addi(b_rec); Sp.o_b_pair = addi(b_pair);
    Sp.o_r_ptr = addi(r_ptr); Sp.o_r_rec = addi(r_rec); Sp.o_r_pair = addi(r_pair); Sp.o_c_ptr = addi(c_ptr); Sp.o_c = addi(cc);
    Sp.od_G = addd(ht.G); Sp.od_B = addd(ht.B); Sp.od_Gd = addd(ht.Gd); Sp.od_Bd = addd(ht.Bd); Sp.od_vset = addd(ht.v_set);
    Sp.ipack_n = (int32_t)ipack.size(); Sp.dpack_n = (int32_t)dpack.size();
    if ((rc = dev_upload(h, &Sp.ipack, ipack)) || (rc = dev_upload(h, &Sp.dpack, dpack))) return bail(rc);
    Sp.orig_slot = T.lu_orig_slot; Sp.orig_i = T.lu_orig_i; Sp.orig_j = T.lu_orig_j; Sp.orig_pos = T.lu_orig_pos;
    Sp.R = h->R;
    const size_t shared_bytes = ((((size_t)Sp.dpack_n + 1) & ~(size_t)1) * 8 + (size_t)Sp.ipack_n * 4 + 15) & ~(size_t)15;
    Sp.wave_bytes = (int32_t)((((size_t)4 * (ht.lu_n_slots + ht.n) + (size_t)7 * ht.n) * sizeof(double) + 15) & ~(size_t)15);
    int waves = (int)((160 * 1024 - 512 - (long long)shared_bytes) / Sp.wave_bytes);
    if (const char* e = GS_EXPERIMENT_ENV("GS_SPARSE_LDS_WAVES")) waves = std::min(waves, atoi(e));
    waves = std::max(0, std::min(4, waves));
    if (waves < 1 || ht.n > 256)
      return bail(fail(nullptr, GS_E_TOPOLOGY, "sparse_lds: the schedule (%zu bytes) and one instance (%d bytes) do not fit the LDS, or more than 256 buses", shared_bytes, Sp.wave_bytes));
    Sp.waves = waves;
    h->sparse_lds = shared_bytes + (size_t)waves * Sp.wave_bytes;
    if (hipFuncSetAttribute((const void*)gs_k_nr_sparse_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return bail(fail(nullptr, GS_E_HIP, "hipFuncSetAttribute(gs_k_nr_sparse_lds) failed"));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    h->sparse_grid = std::max(1, std::min((h->B + waves - 1) / waves, cus));      // persistent: one workgroup per CU, instances strided over the wavefronts
    // the flat-start Jacobian is the same for every instance: factor it once, here, with the solver kernel itself (GS_LU_NO_FLAT=1: off)
    if (!getenv("GS_LU_NO_FLAT")) {
      double* flat = nullptr;
      if ((rc = dev_alloc(h, &flat, (size_t)4 * (ht.lu_n_slots + ht.n) + 4))) return bail(rc);
      GsSparseArgs once = Sp;
      once.flat_out = flat; once.mode = 1; once.max_it = 1;
      hipLaunchKernelGGL(gs_k_nr_sparse_lds, dim3(1), dim3(64 * Sp.waves), h->sparse_lds, h->stream, once, h->slab, 1);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(fail(nullptr, GS_E_HIP, "sparse_lds: factorisation of the flat-start Jacobian failed"));
      Sp.flat = flat;
    }
  }
#endif
  // Sparse block LU: iteration 0 of every solve factors the flat-start Jacobian, which is the same for every instance.  One
  // ordinary solve of group 0, capped at one iteration, leaves those factors in the rows of lane 0; they are kept as a table
  // of wave-uniform scalars (GsTables::lu_flat) and iteration 0 then only carries its right-hand side through
  // (kernels_solve.hip, linsolve_lu_flat: bit-identical -- the same blocks, the same operations).  GS_LU_NO_FLAT=1: off.
  if (h->solve_kernel == 1 && ht.lu_n_piv > 0 && !getenv("GS_LU_NO_FLAT")) {
    double* tab = nullptr;
    const int nblk = ht.lu_n_slots + ht.n;
    if ((rc = dev_alloc(h, &tab, (size_t)4 * nblk + 4))) return bail(rc);
    hipLaunchKernelGGL(gs_k_fill_rows, dim3(1), dim3(64), 0, h->stream, R.P.base, 2, ht.n, R.total, h->slab, -0.01);
    hipLaunchKernelGGL(gs_k_fill_rows, dim3(1), dim3(64), 0, h->stream, R.Q.base, 2, ht.n, R.total, h->slab, 0.0);
    GsSolveCfg once = h->SC; once.max_iterations = 1; once.stamps = nullptr;
    hipLaunchKernelGGL(gs_k_nr_lu, dim3(1), dim3(64 * h->W), h->dyn_lds, h->stream, h->T, h->R, once, h->slab, 1);
    if (ht.lu_n_slots > 0)
      hipLaunchKernelGGL(gs_k_gather_lane, dim3((4 * ht.lu_n_slots + 255) / 256), dim3(256), 0, h->stream, R.LU, 4 * ht.lu_n_slots, 0, h->slab, tab);
    hipLaunchKernelGGL(gs_k_gather_lane, dim3((4 * ht.n + 255) / 256), dim3(256), 0, h->stream, R.LUD, 4 * ht.n, 0, h->slab, tab + (size_t)4 * ht.lu_n_slots);
    double status = 0.0;
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(&status, h->slab + GS_ELEM(R.STATUS, 0), sizeof status, hipMemcpyDeviceToHost) != hipSuccess)
      return bail(fail(nullptr, GS_E_HIP, "sparse LU: factorisation of the flat-start Jacobian failed"));
    const double flag = status == (double)GS_STATUS_SINGULAR ? 1.0 : 0.0;
    if (hipMemcpy(tab + (size_t)4 * nblk, &flag, sizeof flag, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->slab, 0, (size_t)R.total * GS_LANES * sizeof(double)) != hipSuccess)      // group 0 as gs_create leaves every group
      return bail(fail(nullptr, GS_E_HIP, "sparse LU: flat-start table"));
    h->T.lu_flat = tab;
  }
  // Newton-Raphson on the second-generation frame: the constants of the flat-start elimination (GsF2Tables::nrflat), written by
  // ONE workgroup of the step kernel itself on the zeroed state of group 0, then group 0 is cleared again.  GS_NR_NO_FLAT=1: off.
  if ((h->nr2 || h->nrm) && h->f2_npos > 0 && !getenv("GS_NR_NO_FLAT")) {
    double* tab = nullptr;
    if ((rc = dev_alloc(h, &tab, (size_t)h->f2_npos * 16))) return bail(rc);
    if (hipMemset(tab, 0, (size_t)h->f2_npos * 16 * sizeof(double)) != hipSuccess ||
        hipMemset(h->d_in, 0, h->in_doubles * sizeof(double)) != hipSuccess) return bail(fail(nullptr, GS_E_HIP, "hipMemset failed"));      // (d_in: zero actions for the capture step)
    GsF2Tables cap = h->F2; cap.nrflat = tab; cap.nrflat_mode = 1; cap.wg_offset = 0;
    GsPackArgs pa{}; GsFusedChecks fc{}; GsRolloutStep rsv{};
    GsSolveCfg sc = h->SC; sc.stamps = nullptr;
    const dim3 b2(64 * h->f2_nw);
    const int Bc = std::min(h->B, h->f2_iw);
    if (h->nrm) hipLaunchKernelGGL(gs_k_step_nr_mesh2, dim3(1), b2, h->F2.lds_bytes, h->stream, h->T, cap, h->R, sc, h->EC, h->slab, Bc, h->d_in, h->total_load, pa, fc, rsv);
    else if (h->f2_small) hipLaunchKernelGGL(gs_k_step_nr_flow2s, dim3(1), b2, h->F2.lds_bytes, h->stream, h->T, cap, h->R, sc, h->EC, h->slab, Bc, h->d_in, h->total_load, pa, fc, rsv);
    else hipLaunchKernelGGL(gs_k_step_nr_flow2, dim3(1), b2, h->F2.lds_bytes, h->stream, h->T, cap, h->R, sc, h->EC, h->slab, Bc, h->d_in, h->total_load, pa, fc, rsv);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemset(h->slab, 0, (size_t)R.total * GS_LANES * sizeof(double)) != hipSuccess)
      return bail(fail(nullptr, GS_E_HIP, "Newton-Raphson: flat-start table"));
    h->F2.nrflat = tab; h->F2.nrflat_mode = 2;
  }
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(nullptr, GS_E_HIP, "hipDeviceSynchronize failed"));
  *out = h;
  return GS_OK;
}

void gs_destroy(gs_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->loop || h->comm) (void)gs_comm_destroy(h);
  if (h->stream2) { (void)hipStreamDestroy(h->stream2); (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); }
  if (h->ev_peer) (void)hipEventDestroy(h->ev_peer);
  if (h->ev_peer2) (void)hipEventDestroy(h->ev_peer2);
  if (h->ev_step) (void)hipEventDestroy(h->ev_step);
  if (h->ev_full) (void)hipEventDestroy(h->ev_full);
  if (h->ev_scalars) (void)hipEventDestroy(h->ev_scalars);
  for (int k = 0; k < 2; ++k) if (h->ev_gather[k]) (void)hipEventDestroy(h->ev_gather[k]);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  for (auto& t : h->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  if (h->span_a) { (void)hipEventDestroy(h->span_a); (void)hipEventDestroy(h->span_b); }
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->d_actions) (void)hipFree(h->d_actions);
  for (void* p : {(void*)h->ro.obs_seq, (void*)h->ro.act, (void*)h->ro.rew, (void*)h->ro.done, (void*)h->ro.term_count,
                  (void*)h->ro.term_idx, (void*)h->ro.term_obs})
    if (p) (void)hipFree(p);
  if (h->h_pin) (void)hipHostFree(h->h_pin);
  if (h->d_obs_full) (void)hipFree(h->d_obs_full);
  if (h->d_gather_send) (void)hipFree(h->d_gather_send);
  if (h->d_gather_recv) (void)hipFree(h->d_gather_recv);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int gs_dims(const gs_handle* h, int32_t* n, int32_t* m, int32_t* obs_dim, int32_t* action_dim,
            int32_t* state_dim, int32_t* batch) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  if (n) *n = h->n; if (m) *m = h->m; if (obs_dim) *obs_dim = h->obs_dim; if (action_dim) *action_dim = h->action_dim;
  if (state_dim) *state_dim = h->state_dim; if (batch) *batch = h->B;
  return GS_OK;
}

int gs_describe(const gs_handle* h, char* buf, int32_t buflen) {
  if (!h || !buf || buflen <= 0) return fail(nullptr, GS_E_INVALID, "bad arguments");
  static const char* kn[] = {"nr_tree", "nr_sparse_lu", "fbs", "nr_dense_pivot", "nr_tree_lds", "fbs_lds", "fbs_flow", "nr_dense_mfma", "nr_sparse_lds"};
  snprintf(buf, buflen,
           "{\"kernel\": \"%s\", \"n\": %d, \"m\": %d, \"nnz\": %d, \"forest\": %s, \"levels\": %d, \"max_level_width\": %d, "
           "\"lu_slots\": %d, \"lu_orig\": %d, \"lu_pairs\": %lld, \"waves_per_group\": %d, \"groups\": %d, "
           "\"rows_per_group\": %d, \"slab_bytes\": %zu, \"obs_dim\": %d, \"action_dim\": %d, "
           "\"instances_per_workgroup\": %d, \"workgroups\": %d, \"step_lds_bytes\": %zu, \"step_launches\": %d, \"solve_kernel\": \"%s\", \"flow2\": \"%s\", "
           "\"mesh2\": \"%s\", \"mesh_levels\": %d, \"mesh_rows\": %d, \"mesh_message_units\": %d, \"mesh_messages\": %d, \"mesh_accumulators\": %d, "
           "\"dense_form\": \"%s\", \"dense_workgroups\": %d, \"dense_lds_bytes\": %zu}",
           h->flow2 ? (h->f2_small ? "fbs_flow2s" : h->f2_wide ? "fbs_flow2x" : h->f2_half ? "fbs_flow2h" : "fbs_flow2") : h->nrm ? "nr_mesh2" : h->nr2 ? (h->f2_small ? "nr_flow2s" : "nr_flow2") : kn[h->solve_kernel], h->n, h->m, h->topo.nnz, h->topo.is_forest ? "true" : "false", h->topo.n_levels,
           h->topo.max_level_width, h->topo.lu_n_slots, h->topo.lu_n_orig, (long long)h->topo.lu_n_pairs, (h->flow2 || h->nr2 || h->nrm) ? h->f2_nw : h->W, h->groups,
           h->R.total, (size_t)h->groups * h->R.total * GS_LANES * sizeof(double), h->obs_dim, h->action_dim,
           (h->flow2 || h->nr2 || h->nrm) ? h->f2_iw : 64, (h->flow2 || h->nr2 || h->nrm) ? (64 / h->f2_iw) * h->groups : h->groups, (h->flow2 || h->nr2 || h->nrm) ? (size_t)h->F2.lds_bytes : h->dyn_lds + 24576, h->split_ok ? 2 : 1,
           kn[h->solve_kernel], (h->flow2 || h->nr2 || h->nrm) ? "on" : (h->flow2_why.empty() ? "n/a" : h->flow2_why.c_str()),
           h->nrm ? "on" : (h->mesh_why.empty() ? "n/a" : h->mesh_why.c_str()), h->mesh_levels, h->mesh_rows, h->mesh_units, h->mesh_messages, h->mesh_accs,
           h->solve_kernel == 7 ? (h->dense_blockrow ? "block_row" : "panel") : "n/a", h->solve_kernel == 7 ? h->dense_grid : 0, h->solve_kernel == 7 ? h->dense_lds : (size_t)0);
  return GS_OK;
}

int gs_synchronize(gs_handle* h) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  return GS_OK;
}

// ---- solver -------------------------------------------------------------------------------
int gs_upload_injections(gs_handle* h, const double* P_spec, const double* Q_spec) {
  if (!h || !P_spec) return fail(h, GS_E_INVALID, "handle / P_spec is NULL");
  GS_ENTER(h);
  int rc = unpack_from_host(h, h->map_p, h->n, P_spec);
  if (rc) return rc;
  if (Q_spec) {
    HIPCHK(h, hipStreamSynchronize(h->stream));      // d_in is reused
    rc = unpack_from_host(h, h->map_q, h->n, Q_spec);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(gs_k_fill_rows, dim3(h->groups), dim3(64), 0, h->stream, h->R.Q.base, 2, h->n, h->R.total, h->slab, 0.0);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs_solve_device(gs_handle* h) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  GS_ENTER(h);
  h->rows_stale = false;            // the solve writes every result row
  return launch_solve(h);
}

int gs_download_solution(gs_handle* h, const gs_solution_view* out) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / view is NULL");
  GS_ENTER(h);
  int rc;
  if ((rc = ensure_rows(h))) return rc;
  if ((rc = pack_to_host(h, h->map_vm, h->n, out->bus_voltages))) return rc;
  if ((rc = pack_to_host(h, h->map_va, h->n, out->bus_angles))) return rc;
  if ((rc = pack_to_host(h, h->map_flow, h->m, out->line_flows))) return rc;
  if ((rc = pack_to_host(h, h->map_load, h->m, out->line_loadings))) return rc;
  if ((rc = fetch_scalars(h))) return rc;
  const int B = h->B, Bp = h->Bp;
  if (out->losses) memcpy(out->losses, h->h_f + (size_t)SF_LOSSES * Bp, B * sizeof(double));
  if (out->max_mismatch) memcpy(out->max_mismatch, h->h_f + (size_t)SF_MAXMIS * Bp, B * sizeof(double));
  if (out->iterations) memcpy(out->iterations, h->h_i + (size_t)SI_ITERS * Bp, B * sizeof(int32_t));
  if (out->status) memcpy(out->status, h->h_i + (size_t)SI_STATUS * Bp, B * sizeof(int32_t));
  if (out->converged) memcpy(out->converged, h->h_u + (size_t)SU_CONV * Bp, B);
  return GS_OK;
}

int gs_solve(gs_handle* h, const double* P_spec, const double* Q_spec, const gs_solution_view* out) {
  int rc = gs_upload_injections(h, P_spec, Q_spec);
  if (rc) return rc;
  if ((rc = gs_solve_device(h))) return rc;
  return out ? gs_download_solution(h, out) : gs_synchronize(h);
}

// ---- environment ----------------------------------------------------------------------------
int gs_reset(gs_handle* h, const uint64_t* seeds, const uint8_t* mask, double* obs_out) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  GS_ENTER(h);
  { int rc0 = ensure_rows(h); if (rc0) return rc0; }      // a masked reset leaves the other instances' rows as they are: they must be current
  if (seeds) HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, (size_t)h->B * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  if (mask) HIPCHK(h, hipMemcpyAsync(h->d_mask, mask, (size_t)h->B, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(gs_k_env_reset, dim3(h->groups), dim3(64), 0, h->stream, h->T, h->R, h->EC, h->slab, h->B,
                     seeds ? h->d_seeds : (const uint64_t*)nullptr, mask ? h->d_mask : (const uint8_t*)nullptr);
  HIPCHK(h, hipGetLastError());
  h->was_reset = true;
  if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));       // no gather may still be reading an observation buffer
  h->gather_pending[0] = h->gather_pending[1] = false; h->obs_cur = 0;
  int rc = launch_pack(h, h->map_obs, h->obs_dim, h->d_obs2[0]);   // every column, the constants included
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->d_obs2[1], h->d_obs2[0], (size_t)h->B * h->obs_dim * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (obs_out) HIPCHK(h, hipMemcpyAsync(obs_out, h->d_obs2[0], (size_t)h->B * h->obs_dim * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs_host_alloc(void** out, size_t bytes) {
  if (!out || bytes == 0) return fail(nullptr, GS_E_INVALID, "gs_host_alloc: out is NULL or bytes == 0");
  *out = nullptr;
  if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, GS_E_NOMEM, "hipHostMalloc(%zu bytes) failed", bytes);
  }
  return GS_OK;
}

int gs_host_free(void* p) {
  if (!p) return GS_OK;
  if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, GS_E_INVALID, "gs_host_free: not a gs_host_alloc pointer"); }
  return GS_OK;
}

// obs32: the observation block rounded to float32 (the dtype the reference DECLARES for its observation space, grid_env.py:346; its
// values are Python floats) -- converted on the device, 22 MB instead of 45 over the link at [8192][684]
static int download_step_impl(gs_handle* h, double* obs, float* obs32, double* reward, uint8_t* terminated, uint8_t* truncated,
                              const gs_info_view* info) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  GS_ENTER(h);
  if (obs32) {
    if (!h->d_obs32) { int rc0 = dev_alloc(h, &h->d_obs32, (size_t)h->Bp * h->obs_dim); if (rc0) return rc0; }
    if (!h->ev_scalars) HIPCHK(h, hipEventCreateWithFlags(&h->ev_scalars, hipEventDisableTiming));
    int rc = fetch_scalars(h, false);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_scalars, h->stream));
    const long long n = (long long)h->B * h->obs_dim;
    hipLaunchKernelGGL(gs_k_obs_to_f32, dim3((unsigned)((n + 2 * 256 - 1) / (2 * 256))), dim3(256), 0, h->stream, h->d_obs2[h->obs_cur], h->d_obs32, n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(obs32, h->d_obs32, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev_scalars));
    copy_info(h, reward, terminated, truncated, info);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GS_OK;
  }
  if (!obs) {
    int rc = fetch_scalars(h);
    if (rc) return rc;
    copy_info(h, reward, terminated, truncated, info);
    return GS_OK;
  }
  // the scalars first (half a megabyte the kernel stores into the handle's page-locked block itself), then the observation block
  // behind them: the caller's reward / flag / info arrays are filled while the block is still crossing the link
  if (!h->ev_scalars) HIPCHK(h, hipEventCreateWithFlags(&h->ev_scalars, hipEventDisableTiming));
  int rc = fetch_scalars(h, false);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev_scalars, h->stream));
  const double* src = h->d_obs2[h->obs_cur];
  const size_t D = (size_t)h->obs_dim, pitch = D * sizeof(double), s0 = (size_t)h->obs_skip0, s1 = (size_t)h->obs_skip1, B = (size_t)h->B;
  const bool bound = s1 > s0 && std::find(h->bound_obs.begin(), h->bound_obs.end(), obs) != h->bound_obs.end();
  if (bound) {
    // The constants are in place (gs_host_obs_bind).  What changes is, in memory order, ONE run per row boundary: the columns of row r
    // behind the constant block and those of row r + 1 in front of it -- a single pitched copy of B - 1 runs of obs_dim - (skip1 -
    // skip0) doubles, plus the head of row 0 and the tail of row B - 1.  Measured at B = 8192 on the 123-bus feeder, per env.step():
    // this 0.78 ms; a kernel storing the same columns into the mapped array 0.82 ms (its stores cross the link 64 bytes at a time);
    // two pitched copies, one either side of the constants, 1.00 ms; the whole block 1.10 ms.
    if (s0 > 0) HIPCHK(h, hipMemcpyAsync(obs, src, s0 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (B > 1)
      HIPCHK(h, hipMemcpy2DAsync(obs + s1, pitch, src + s1, pitch, (D - (s1 - s0)) * sizeof(double), B - 1, hipMemcpyDeviceToHost, h->stream));
    if (D > s1)
      HIPCHK(h, hipMemcpyAsync(obs + (B - 1) * D + s1, src + (B - 1) * D + s1, (D - s1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    HIPCHK(h, hipMemcpyAsync(obs, src, B * pitch, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipEventSynchronize(h->ev_scalars));
  copy_info(h, reward, terminated, truncated, info);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs_download_step(gs_handle* h, double* obs, double* reward, uint8_t* terminated, uint8_t* truncated, const gs_info_view* info) {
  return download_step_impl(h, obs, nullptr, reward, terminated, truncated, info);
}

int gs_download_step_f32(gs_handle* h, float* obs, double* reward, uint8_t* terminated, uint8_t* truncated, const gs_info_view* info) {
  if (!obs) return fail(h, GS_E_INVALID, "gs_download_step_f32: obs is NULL");
  return download_step_impl(h, nullptr, obs, reward, terminated, truncated, info);
}

// A host observation array [B][obs_dim] the caller will hand to gs_step / gs_download_step again and again: its constant columns
// (the static load powers, grid_env.py:769-770 -- they never change) are written here, once, and later downloads into the same
// address move the changing columns only.  The caller must leave those columns alone (or bind again).
int gs_host_obs_bind(gs_handle* h, double* obs) {
  if (!h || !obs) return fail(h, GS_E_INVALID, "handle / obs is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_host_obs_bind before gs_reset");
  GS_ENTER(h);
  if (h->obs_skip1 > h->obs_skip0) {
    const size_t pitch = (size_t)h->obs_dim * sizeof(double);
    HIPCHK(h, hipMemcpy2DAsync(obs + h->obs_skip0, pitch, h->d_obs2[h->obs_cur] + h->obs_skip0, pitch, (size_t)(h->obs_skip1 - h->obs_skip0) * sizeof(double),
                               (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (std::find(h->bound_obs.begin(), h->bound_obs.end(), obs) == h->bound_obs.end()) h->bound_obs.push_back(obs);
  return GS_OK;
}

int gs_host_obs_unbind(gs_handle* h, double* obs) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  auto it = std::find(h->bound_obs.begin(), h->bound_obs.end(), (const double*)obs);
  if (it != h->bound_obs.end()) h->bound_obs.erase(it);
  return GS_OK;
}

int gs_step(gs_handle* h, const double* actions, double* obs, double* reward, uint8_t* terminated,
            uint8_t* truncated, const gs_info_view* info) {
  if (!h || (!actions && h->action_dim > 0)) return fail(h, GS_E_INVALID, "handle / actions is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_step before gs_reset");
  GS_ENTER(h);
  if (h->action_dim > 0)
    HIPCHK(h, hipMemcpyAsync(h->d_in, actions, (size_t)h->B * h->action_dim * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = step_kernels(h, h->d_in);
  if (rc) return rc;
  return gs_download_step(h, obs, reward, terminated, truncated, info);
}

int gs_step_f32(gs_handle* h, const double* actions, float* obs, double* reward, uint8_t* terminated, uint8_t* truncated,
                const gs_info_view* info) {
  if (!h || (!actions && h->action_dim > 0) || !obs) return fail(h, GS_E_INVALID, "handle / actions / obs is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_step_f32 before gs_reset");
  GS_ENTER(h);
  if (h->action_dim > 0)
    HIPCHK(h, hipMemcpyAsync(h->d_in, actions, (size_t)h->B * h->action_dim * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = step_kernels(h, h->d_in);
  if (rc) return rc;
  return download_step_impl(h, nullptr, obs, reward, terminated, truncated, info);
}

int gs_step_device_ptr(gs_handle* h, const double* d_actions, void* producer_stream) {
  if (!h || (!d_actions && h->action_dim > 0)) return fail(h, GS_E_INVALID, "handle / d_actions is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_step_device_ptr before gs_reset");
  HIPCHK(h, hipSetDevice(h->device));
  if (producer_stream) {             // the step waits, on the device, for what the producer has queued so far
    if (!h->ev_peer) HIPCHK(h, hipEventCreateWithFlags(&h->ev_peer, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_peer, peer_stream(producer_stream)));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_peer, 0));
    if (h->forked) HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_peer, 0));
  }
  return step_kernels(h, d_actions);
}

int gs_step_device_view(gs_handle* h, gs_step_device_out* out, void* consumer_stream) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / out is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_step_device_view before gs_reset");
  GS_ENTER(h);
  hipLaunchKernelGGL(gs_k_scalars, dim3(h->groups), dim3(64), 0, h->stream, h->rows_f, (int)SF_COUNT, h->rows_i,
                     (int)SI_COUNT, h->rows_u, (int)SU_COUNT, h->R.total, h->slab, h->sc_f, h->sc_i, h->sc_u, h->Bp, (uint32_t*)nullptr, 0);
  HIPCHK(h, hipGetLastError());
  if (consumer_stream) {
    if (!h->ev_peer2) HIPCHK(h, hipEventCreateWithFlags(&h->ev_peer2, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_peer2, h->stream));
    HIPCHK(h, hipStreamWaitEvent(peer_stream(consumer_stream), h->ev_peer2, 0));
  } else {
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  out->observations = h->d_obs2[h->obs_cur];
  out->reward = h->sc_f + (size_t)SF_REWARD * h->Bp;
  out->terminated = h->sc_u + (size_t)SU_TERM * h->Bp;
  out->truncated = h->sc_u + (size_t)SU_TRUNC * h->Bp;
  out->B = h->B; out->obs_dim = h->obs_dim;
  return GS_OK;
}

int gs_upload_actions(gs_handle* h, const double* actions, int32_t n_batches) {
  if (!h || !actions || n_batches <= 0) return fail(h, GS_E_INVALID, "bad arguments");
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->d_actions) { (void)hipFree(h->d_actions); h->d_actions = nullptr; }
  const size_t bytes = (size_t)n_batches * h->B * std::max(h->action_dim, 1) * sizeof(double);
  HIPCHK(h, hipMalloc((void**)&h->d_actions, bytes));
  HIPCHK(h, hipMemcpy(h->d_actions, actions, (size_t)n_batches * h->B * h->action_dim * sizeof(double), hipMemcpyHostToDevice));
  h->n_action_batches = n_batches;
  return GS_OK;
}

int gs_step_device(gs_handle* h, int32_t k) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_step_device before gs_reset");
  if (k < 0 || k >= h->n_action_batches) return fail(h, GS_E_INVALID, "action batch %d not uploaded (have %d)", k, h->n_action_batches);
  HIPCHK(h, hipSetDevice(h->device));
  return step_kernels(h, h->d_actions + (size_t)k * h->B * h->action_dim);
}


// ---- device-resident rollout collection ---------------------------------------------------------------
// T fused env steps back to back, nothing on the host in between (algorithms/base.py:268-298, batched).
// Device layout (gs_rollout_device_view): obs_seq[T + 1][B][obs_dim] -- slot t is what step t started from, slot
// t + 1 is written by step t's kernel itself (its observation output IS the next slot: no copy) --, act[T][B][A],
// rew[T][B], done[T][B], and the side list of terminal observations (t, b, row) the in-place resets replaced.
static int rollout_ensure(gs_handle* h, int T) {
  gs_handle::Rollout& ro = h->ro;
  if (T <= ro.T_cap) return GS_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (void* p : {(void*)ro.obs_seq, (void*)ro.act, (void*)ro.rew, (void*)ro.done, (void*)ro.term_idx, (void*)ro.term_obs})
    if (p) (void)hipFree(p);
  ro.obs_seq = nullptr; ro.act = nullptr; ro.rew = nullptr; ro.done = nullptr; ro.term_idx = nullptr; ro.term_obs = nullptr; ro.T_cap = 0;
  const size_t B = h->B, D = h->obs_dim, A = std::max(h->action_dim, 1);
  // an instance finishes at most once per min(episode_length, 11) steps (truncation needs more than 10 violating steps
  // since its last reset, grid_env.py:604) plus once for an episode that was already under way
  const int min_ep = std::max(1, std::min(h->cfg.episode_length, 11));
  const size_t cap = B * ((size_t)T / min_ep + 1);
  if (!ro.term_count) HIPCHK(h, hipMalloc((void**)&ro.term_count, sizeof(int32_t)));
  if (hipMalloc((void**)&ro.obs_seq, (size_t)(T + 1) * B * D * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&ro.act, (size_t)T * B * A * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&ro.rew, (size_t)T * B * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&ro.done, (size_t)T * B) != hipSuccess ||
      hipMalloc((void**)&ro.term_idx, cap * 2 * sizeof(int32_t)) != hipSuccess ||
      hipMalloc((void**)&ro.term_obs, cap * D * sizeof(double)) != hipSuccess)
    return fail(h, GS_E_NOMEM, "rollout buffers for T = %d (%.1f MB per step) do not fit", T, (double)B * D * 8e-6);
  ro.T_cap = T; ro.term_cap = (int)std::min<size_t>(cap, 0x7fffffff);
  // the constant columns of every slot, once: the step kernels write only the columns that change
  const long long rows = (long long)(T + 1) * B;
  const int w = h->obs_skip1 - h->obs_skip0;
  if (w > 0) {
    const long long total = rows * w;
    hipLaunchKernelGGL(gs_k_fill_const_columns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, ro.obs_seq, rows,
                       h->obs_dim, h->obs_skip0, h->obs_skip1, h->map_obs, h->d_cst);
    HIPCHK(h, hipGetLastError());
  }
  return GS_OK;
}

int gs_rollout(gs_handle* h, int32_t T, int32_t policy, uint64_t policy_seed, const double* actions) {
  if (!h || T <= 0) return fail(h, GS_E_INVALID, "handle is NULL or T <= 0");
  if (policy != GS_POLICY_UPLOADED && policy != GS_POLICY_RANDOM) return fail(h, GS_E_INVALID, "unknown policy %d", policy);
  if (policy == GS_POLICY_UPLOADED && !actions && h->action_dim > 0) return fail(h, GS_E_INVALID, "GS_POLICY_UPLOADED needs actions[T][B][action_dim]");
  if (!h->was_reset) return fail(h, GS_E_STATE, "gs_rollout before gs_reset");
  GS_ENTER(h);
  int rc = rollout_ensure(h, T);
  if (rc) return rc;
  gs_handle::Rollout& ro = h->ro;
  const size_t B = h->B, D = h->obs_dim, A = h->action_dim;
  ro.T = T; ro.n_term = -1;
  HIPCHK(h, hipMemsetAsync(ro.term_count, 0, sizeof(int32_t), h->stream));
  if (A > 0) {
    if (policy == GS_POLICY_UPLOADED) {
      HIPCHK(h, hipMemcpyAsync(ro.act, actions, (size_t)T * B * A * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
      const long long total = (long long)T * B * ((A + 3) / 4);
      hipLaunchKernelGGL(gs_k_rollout_actions, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, ro.act, (int)T, (int)B, (int)A,
                         policy_seed, h->EC.first_instance, 0u);
      HIPCHK(h, hipGetLastError());
    }
  }
  // slot 0 = the observation the environment stands at
  HIPCHK(h, hipMemcpyAsync(ro.obs_seq, h->d_obs2[h->obs_cur], B * D * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  // second-generation step kernels do the bookkeeping themselves (finished instances are reset at the start of the NEXT
  // step, rewards / flags written at the end of the step): one launch per step, and one small kernel after the last
  // step for the instances it finished; the other kernels are followed by that small kernel after every step
  // (The whole rollout as ONE launch -- each workgroup looping over the T steps by itself, no workgroup needs another --
  // was built in round 2 and is bit-identical, but slower: inlined into a loop the step's ~1 KB argument block stays live
  // across iterations (230 spilled registers); as an out-of-line call reading its arguments from memory the block lands in
  // scratch (59 M env-steps/s against 150 M for a launch per step).)
  const bool fused = h->flow2 || h->nr2 || h->nrm;
  for (int t = 0; t < T; ++t) {
    double* nxt = ro.obs_seq + (size_t)(t + 1) * B * D;
    GsRolloutStep rs{ro.rew, ro.done, ro.obs_seq + (size_t)t * B * D, h->map_obs, h->d_cst, ro.term_count, ro.term_idx, ro.term_obs, ro.term_cap, h->obs_dim, t, 1};
    if ((rc = step_kernels(h, ro.act + (size_t)t * B * A, nxt, fused ? &rs : nullptr))) return rc;
    if (!fused || t == T - 1) {
      if ((rc = join_streams(h))) return rc;
      GsRolloutPostArgs pa{fused ? nullptr : ro.rew, fused ? nullptr : ro.done, nxt, h->map_obs, h->d_cst, ro.term_count, ro.term_idx, ro.term_obs, ro.term_cap, h->obs_dim, t, h->B};
      hipLaunchKernelGGL(gs_k_rollout_post, dim3(h->groups), dim3(256), 0, h->stream, h->T, h->R, h->EC, h->slab, pa);
      HIPCHK(h, hipGetLastError());
    }
  }
  if ((rc = join_streams(h))) return rc;
  // the environment now stands at slot T: that is its current observation for gs_download_step / gs_allgather_obs
  if (h->gather_pending[h->obs_cur]) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gather[h->obs_cur], 0)); h->gather_pending[h->obs_cur] = false; }
  HIPCHK(h, hipMemcpyAsync(h->d_obs2[h->obs_cur], ro.obs_seq + (size_t)T * B * D, B * D * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->rows_stale) h->last_obs = h->d_obs2[h->obs_cur];      // (the slot may be reallocated by a longer rollout; this copy stays)
  ro.calls += 1;
  return GS_OK;      // asynchronous: gs_synchronize / gs_rollout_download / gs_rollout_device_view wait for it
}

static int rollout_finish(gs_handle* h) {
  gs_handle::Rollout& ro = h->ro;
  if (ro.T <= 0) return fail(h, GS_E_STATE, "no rollout has been collected on this handle");
  GS_ENTER(h);
  if (ro.n_term < 0) {
    int32_t n = 0;
    HIPCHK(h, hipMemcpyAsync(&n, ro.term_count, sizeof n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n > ro.term_cap) return fail(h, GS_E_NOMEM, "internal: %d finished episodes exceed the terminal list (%d)", n, ro.term_cap);
    ro.n_term = n;
  }
  return GS_OK;
}

int gs_rollout_device_view(gs_handle* h, gs_rollout_device* out) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / out is NULL");
  int rc = rollout_finish(h);
  if (rc) return rc;
  const gs_handle::Rollout& ro = h->ro;
  out->T = ro.T; out->B = h->B; out->obs_dim = h->obs_dim; out->action_dim = h->action_dim;
  out->obs_seq = ro.obs_seq; out->actions = ro.act; out->rewards = ro.rew; out->terminals = ro.done;
  out->n_terminal = ro.n_term; out->terminal_index = ro.term_idx; out->terminal_obs = ro.term_obs;
  return GS_OK;
}

int gs_rollout_download(gs_handle* h, const gs_rollout_view* out) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / view is NULL");
  int rc = rollout_finish(h);
  if (rc) return rc;
  const gs_handle::Rollout& ro = h->ro;
  const size_t T = ro.T, B = h->B, D = h->obs_dim, A = h->action_dim, blk = B * D;
  if (out->observations) HIPCHK(h, hipMemcpyAsync(out->observations, ro.obs_seq, T * blk * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->next_observations) HIPCHK(h, hipMemcpyAsync(out->next_observations, ro.obs_seq + blk, T * blk * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->final_observation) HIPCHK(h, hipMemcpyAsync(out->final_observation, ro.obs_seq + T * blk, blk * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->actions && A) HIPCHK(h, hipMemcpyAsync(out->actions, ro.act, T * B * A * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->rewards) HIPCHK(h, hipMemcpyAsync(out->rewards, ro.rew, T * B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->terminals) HIPCHK(h, hipMemcpyAsync(out->terminals, ro.done, T * B, hipMemcpyDeviceToHost, h->stream));
  std::vector<int32_t> idx((size_t)ro.n_term * 2);
  std::vector<double> rows(out->next_observations ? (size_t)ro.n_term * D : 0);
  if (ro.n_term && out->next_observations) {
    HIPCHK(h, hipMemcpyAsync(idx.data(), ro.term_idx, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(rows.data(), ro.term_obs, rows.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // next_observations[t][b] of a finished transition is the terminal observation, not the fresh one in slot t + 1
  if (out->next_observations)
    for (int k = 0; k < ro.n_term; ++k)
      memcpy(out->next_observations + ((size_t)idx[2 * k] * B + idx[2 * k + 1]) * D, rows.data() + (size_t)k * D, D * sizeof(double));
  if (out->n_terminal) *out->n_terminal = ro.n_term;
  return GS_OK;
}

// ---- checkpoint ---------------------------------------------------------------------------------
int gs_get_state(gs_handle* h, double* state) {
  if (!h || !state) return fail(h, GS_E_INVALID, "handle / state is NULL");
  GS_ENTER(h);
  int rc = ensure_rows(h);
  if (rc) return rc;
  return pack_to_host(h, h->map_state, h->state_dim, state);
}

int gs_set_state(gs_handle* h, const double* state) {
  if (!h || !state) return fail(h, GS_E_INVALID, "handle / state is NULL");
  GS_ENTER(h);
  h->rows_stale = false;            // the checkpoint carries every result row
  int rc = unpack_from_host(h, h->map_state, h->state_dim, state);
  if (rc) return rc;
  // rows that follow from the checkpoint: the rectangular voltages (what a warm-started sweep solver resumes from) and
  // the uncurtailed renewable powers of the observation
  hipLaunchKernelGGL(gs_k_polar_to_rect, dim3(h->groups), dim3(64), 0, h->stream, h->T, h->R, h->slab, h->B);
  HIPCHK(h, hipGetLastError());
  // both observation buffers whole, as gs_reset leaves them: the step kernel never writes the constant columns, so a
  // handle that is restored without ever having been reset (resume in a new process) must get them here
  if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  h->gather_pending[0] = h->gather_pending[1] = false;
  if ((rc = launch_pack(h, h->map_obs, h->obs_dim, h->d_obs2[h->obs_cur]))) return rc;
  HIPCHK(h, hipMemcpyAsync(h->d_obs2[h->obs_cur ^ 1], h->d_obs2[h->obs_cur], (size_t)h->B * h->obs_dim * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->was_reset = true;
  return GS_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------------------
int gs_comm_unique_id(uint8_t id_out[128]) {
  std::string why;
  if (!load_rccl(why)) return fail(nullptr, GS_E_COMM, "%s", why.c_str());
  gs_ncclUniqueId id;
  int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) return fail(nullptr, GS_E_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
  memcpy(id_out, id.internal, 128);
  return GS_OK;
}

// What both transports need on a member: the gathered block [world * B][obs_dim] with its constant columns in place,
// the compact send / receive blocks, the exchange stream and its events.
static int comm_buffers(gs_handle* h, int rank, int world_size) {
  h->rank = rank; h->world = world_size;
  HIPCHK(h, hipMalloc((void**)&h->d_obs_full, (size_t)world_size * h->B * h->obs_dim * sizeof(double)));
  const int nd = h->obs_dim - (h->obs_skip1 - h->obs_skip0);
  HIPCHK(h, hipMalloc((void**)&h->d_gather_send, (size_t)h->B * nd * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->d_gather_recv, (size_t)world_size * h->B * nd * sizeof(double)));
  if (!h->comm_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_step, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_full, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather[k], hipEventDisableTiming));
  }
  // the constant columns of the gathered block do not depend on the rank (static load powers of the shared feeder):
  // written here once, never sent
  if (h->obs_skip1 > h->obs_skip0) {
    const long long rows = (long long)world_size * h->B, total = rows * (h->obs_skip1 - h->obs_skip0);
    hipLaunchKernelGGL(gs_k_fill_const_columns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->comm_stream, h->d_obs_full, rows,
                       h->obs_dim, h->obs_skip0, h->obs_skip1, h->map_obs, h->d_cst);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  }
  return GS_OK;
}

int gs_comm_init(gs_handle* h, const uint8_t id[128], int32_t rank, int32_t world_size) {
  if (!h || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(h, GS_E_INVALID, "bad arguments");
  if (h->comm || h->loop) return fail(h, GS_E_STATE, "the handle already belongs to a communicator");
  std::string why;
  if (!load_rccl(why)) return fail(h, GS_E_COMM, "%s", why.c_str());
  GS_ENTER(h);
  gs_ncclUniqueId uid; memcpy(uid.internal, id, 128);
  int rc = g_rccl.CommInitRank(&h->comm, world_size, uid, rank);
  if (rc != 0) return fail(h, GS_E_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
  return comm_buffers(h, rank, world_size);
}

// ---- the in-process transport ----------------------------------------------------------------------
// `world` handles of ONE process form the communicator; rank r's compact block reaches rank q by a device-to-device
// copy on q's exchange stream where RCCL would move it over xGMI.  Everything either side of that copy -- compaction,
// slot offsets, expansion into [world * B][obs_dim], constant columns, the double-buffered observation buffers and
// their events -- is the code the RCCL transport runs.  The collective completes when the last member has called
// (the semantics of a grouped RCCL call): that call queues every member's copies.
struct GsLoopComm {
  int world = 0, n_live = 0, n_arrived = 0;
  std::vector<gs_handle*> member;
  std::vector<hipEvent_t> ev_sent;      // rank r's send block is complete (recorded on r's exchange stream)
  std::vector<hipEvent_t> ev_taken;     // rank r has copied every send block of the round (before anyone refills one)
  std::vector<uint8_t> arrived, taken_valid;
  std::vector<double*> host_out;
};

static int loop_complete_body(GsLoopComm* lc) {
  const gs_handle* h0 = lc->member[0];
  const int D = h0->obs_dim, nd = D - (h0->obs_skip1 - h0->obs_skip0);
  const size_t count = (size_t)h0->B * nd;
  for (int r = 0; r < lc->world; ++r) {
    gs_handle* q = lc->member[r];
    HIPCHK(q, hipSetDevice(q->device));
    for (int p = 0; p < lc->world; ++p) {
      if (p != r) HIPCHK(q, hipStreamWaitEvent(q->comm_stream, lc->ev_sent[p], 0));
      HIPCHK(q, hipMemcpyAsync(q->d_gather_recv + (size_t)p * count, lc->member[p]->d_gather_send, count * sizeof(double),
                               hipMemcpyDeviceToDevice, q->comm_stream));
    }
    HIPCHK(q, hipEventRecord(lc->ev_taken[r], q->comm_stream));
    lc->taken_valid[r] = 1;
    const size_t total = count * lc->world;
    hipLaunchKernelGGL(gs_k_obs_compact, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, q->comm_stream, q->d_gather_recv, q->d_obs_full,
                       (long long)lc->world * q->B, D, q->obs_skip0, q->obs_skip1, 1);
    HIPCHK(q, hipGetLastError());
    if (lc->host_out[r])
      HIPCHK(q, hipMemcpyAsync(lc->host_out[r], q->d_obs_full, (size_t)q->B * D * lc->world * sizeof(double), hipMemcpyDeviceToHost, q->comm_stream));
  }
  for (int r = 0; r < lc->world; ++r)
    if (lc->host_out[r]) HIPCHK(lc->member[r], hipStreamSynchronize(lc->member[r]->comm_stream));
  return GS_OK;
}

// The round ends here whether or not it succeeded: a failed round leaves no member "arrived" and keeps no pointer into
// the callers' memory, so that the next call reports its own error (or works) instead of "called twice" / "gather half-way".
static int loop_complete(GsLoopComm* lc) {
  const int rc = loop_complete_body(lc);
  if (rc)       // copies of the failed round may still be queued towards the callers' host arrays: drain before letting go of them
    for (int r = 0; r < lc->world; ++r)
      if (lc->host_out[r] && lc->member[r] && lc->member[r]->comm_stream) { (void)hipSetDevice(lc->member[r]->device); (void)hipStreamSynchronize(lc->member[r]->comm_stream); }
  for (int r = 0; r < lc->world; ++r) { lc->host_out[r] = nullptr; lc->arrived[r] = 0; }
  lc->n_arrived = 0;
  return rc;
}

int gs_comm_init_loopback(gs_handle* const* shards, int32_t nshards) {
  if (!shards || nshards < 1) return fail(nullptr, GS_E_INVALID, "bad arguments");
  for (int r = 0; r < nshards; ++r) {
    gs_handle* h = shards[r];
    if (!h) return fail(nullptr, GS_E_INVALID, "shard %d is NULL", r);
    if (h->comm || h->loop) return fail(h, GS_E_STATE, "shard %d already belongs to a communicator", r);
    if (h->B != shards[0]->B || h->obs_dim != shards[0]->obs_dim || h->obs_skip0 != shards[0]->obs_skip0 || h->obs_skip1 != shards[0]->obs_skip1)
      return fail(h, GS_E_INVALID, "shard %d: batch / observation layout differs from shard 0 (the all-gather needs equal shards)", r);
    for (int q = 0; q < r; ++q) if (shards[q] == h) return fail(h, GS_E_INVALID, "shard %d is shard %d again", r, q);
  }
  GsLoopComm* lc = new GsLoopComm();
  lc->world = lc->n_live = nshards;
  lc->member.assign(shards, shards + nshards);
  lc->ev_sent.assign(nshards, nullptr); lc->ev_taken.assign(nshards, nullptr);
  lc->arrived.assign(nshards, 0); lc->taken_valid.assign(nshards, 0); lc->host_out.assign(nshards, nullptr);
  for (int r = 0; r < nshards; ++r) {
    gs_handle* h = shards[r];
    int rc = GS_OK;
    do {
      hipError_t e = hipSetDevice(h->device);
      if (e == hipSuccess && h->forked) { rc = join_streams(h); if (rc) break; }
      if (e == hipSuccess) e = hipEventCreateWithFlags(&lc->ev_sent[r], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&lc->ev_taken[r], hipEventDisableTiming);
      if (e != hipSuccess) { rc = fail(h, GS_E_HIP, "loopback communicator set-up failed: %s", hipGetErrorString(e)); break; }
      rc = comm_buffers(h, r, nshards);
    } while (0);
    if (rc) {       // undo: members attached so far go back to "no communicator"
      for (int q = 0; q <= r; ++q) { shards[q]->loop = nullptr; (void)gs_comm_destroy(shards[q]); }
      for (int q = 0; q < nshards; ++q) { if (lc->ev_sent[q]) (void)hipEventDestroy(lc->ev_sent[q]); if (lc->ev_taken[q]) (void)hipEventDestroy(lc->ev_taken[q]); }
      delete lc;
      return rc;
    }
    h->loop = lc;
  }
  return GS_OK;
}

// The exchange of one member in three parts, so that a process driving several members through RCCL can put ONLY the
// collectives between ncclGroupStart and ncclGroupEnd: inside a group ncclAllGather merely records the call, the work is
// enqueued on the exchange stream at ncclGroupEnd -- anything launched on that stream in between (the expansion, a
// download) would run BEFORE the collective and see the previous round's block.
//   gather_prepare     behind the step that produced the current observation buffer: compact its changing columns
//   gather_collective  ncclAllGather of the compact blocks (RCCL transport only)
//   gather_finish      expand into [world * B][obs_dim]; optional download
static int gather_prepare(gs_handle* h) {
  GsLoopComm* lc = h->loop;
  const int D = h->obs_dim, nd = D - (h->obs_skip1 - h->obs_skip0);
  const size_t count = (size_t)h->B * nd;
  // On its own stream, behind the step that produced the current observation buffer.  Only the columns that change
  // travel: the block is compacted first (which is also all the gather needs of the observation buffer -- the step
  // after the next one, which reuses that buffer, waits for ev_gather = the end of the compaction, not of the gather),
  // the compact blocks are gathered over xGMI, and expanded into the [world * B][obs_dim] block.
  const int cur = h->obs_cur;
  HIPCHK(h, hipEventRecord(h->ev_step, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_step, 0));
  if (lc)       // loopback only: the peers copy OUT of this send block on their own streams (RCCL reads it on this one)
    for (int r = 0; r < lc->world; ++r)
      if (r != h->rank && lc->taken_valid[r]) HIPCHK(h, hipStreamWaitEvent(h->comm_stream, lc->ev_taken[r], 0));
  hipLaunchKernelGGL(gs_k_obs_compact, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->comm_stream, h->d_obs2[cur], h->d_gather_send,
                     (long long)h->B, D, h->obs_skip0, h->obs_skip1, 0);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev_gather[cur], h->comm_stream));
  h->gather_pending[cur] = true;
  return GS_OK;
}

static int gather_collective(gs_handle* h) {
  const size_t count = (size_t)h->B * (h->obs_dim - (h->obs_skip1 - h->obs_skip0));
  const int rc = g_rccl.AllGather(h->d_gather_send, h->d_gather_recv, count, /*ncclFloat64*/ 8, h->comm, h->comm_stream);
  if (rc != 0) return fail(h, GS_E_COMM, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
  return GS_OK;
}

static int gather_finish(gs_handle* h, double* obs_full_host) {
  const int D = h->obs_dim;
  const size_t total = (size_t)h->B * (D - (h->obs_skip1 - h->obs_skip0)) * h->world;
  hipLaunchKernelGGL(gs_k_obs_compact, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->comm_stream, h->d_gather_recv, h->d_obs_full,
                     (long long)h->world * h->B, D, h->obs_skip0, h->obs_skip1, 1);
  HIPCHK(h, hipGetLastError());
  if (obs_full_host) {
    HIPCHK(h, hipMemcpyAsync(obs_full_host, h->d_obs_full, (size_t)h->B * D * h->world * sizeof(double), hipMemcpyDeviceToHost, h->comm_stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  }
  return GS_OK;
}

int gs_allgather_obs(gs_handle* h, double* obs_full_host) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  if (!h->comm && !h->loop) return fail(h, GS_E_STATE, "gs_allgather_obs before gs_comm_init / gs_comm_init_loopback");
  GsLoopComm* lc = h->loop;
  if (lc && lc->n_live != lc->world) return fail(h, GS_E_STATE, "a member of the loopback communicator has left");
  if (lc && lc->arrived[h->rank]) return fail(h, GS_E_STATE, "rank %d called gs_allgather_obs twice before every member had called once", h->rank);
  GS_ENTER(h);
  int rc = gather_prepare(h);
  if (rc) return rc;
  if (lc) {
    HIPCHK(h, hipEventRecord(lc->ev_sent[h->rank], h->comm_stream));
    lc->arrived[h->rank] = 1; lc->host_out[h->rank] = obs_full_host;
    if (++lc->n_arrived == lc->world) return loop_complete(lc);
    return GS_OK;
  }
  if ((rc = gather_collective(h))) return rc;
  return gather_finish(h, obs_full_host);
}

int gs_allgather_obs_shards(gs_handle* const* shards, int32_t nshards, double* obs_full_host) {
  if (!shards || nshards < 1 || !shards[0]) return fail(nullptr, GS_E_INVALID, "bad arguments");
  GsLoopComm* lc = shards[0]->loop;
  for (int r = 0; r < nshards; ++r) {
    if (!shards[r]) return fail(nullptr, GS_E_INVALID, "shard %d is NULL", r);
    if (shards[r]->loop != lc || (!lc && !shards[r]->comm)) return fail(shards[r], GS_E_STATE, "shard %d is not in the communicator of shard 0", r);
  }
  if (lc && (nshards != lc->world || lc->n_arrived != 0)) return fail(shards[0], GS_E_STATE, "the call must name every member of the loopback communicator once, with no gather half-way");
  if (!lc) {      // one process driving several GPUs through RCCL: the members' collectives form one group (see gather_prepare)
    if (!g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(shards[0], GS_E_COMM, "librccl lacks ncclGroupStart / ncclGroupEnd");
    int rc = GS_OK;
    for (int r = 0; r < nshards; ++r) {
      gs_handle* h = shards[r];
      GS_ENTER(h);
      if ((rc = gather_prepare(h))) return rc;
    }
    g_rccl.GroupStart();
    for (int r = 0; r < nshards && !rc; ++r) {
      if (hipSetDevice(shards[r]->device) != hipSuccess) rc = fail(shards[r], GS_E_HIP, "hipSetDevice failed");
      else rc = gather_collective(shards[r]);
    }
    const int rg = g_rccl.GroupEnd();          // (always closed, also after a failed call inside the group)
    if (rc) return rc;
    if (rg != 0) return fail(shards[0], GS_E_COMM, "ncclGroupEnd: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rg) : "error");
    for (int r = 0; r < nshards; ++r) {
      HIPCHK(shards[r], hipSetDevice(shards[r]->device));
      if ((rc = gather_finish(shards[r], r == 0 ? obs_full_host : nullptr))) return rc;
    }
    return GS_OK;
  }
  for (int r = 0; r < nshards; ++r) {
    int rc = gs_allgather_obs(shards[r], r == 0 ? obs_full_host : nullptr);
    if (rc) return rc;
  }
  return GS_OK;
}

int gs_allgather_obs_view(gs_handle* h, gs_gathered_obs* out, void* consumer_stream) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / out is NULL");
  if (!h->d_obs_full) return fail(h, GS_E_STATE, "no communicator on this handle");
  if (h->loop && h->loop->arrived[h->rank]) return fail(h, GS_E_STATE, "the gather of this round is not complete: not every member has called gs_allgather_obs");
  HIPCHK(h, hipSetDevice(h->device));
  if (consumer_stream) {
    HIPCHK(h, hipEventRecord(h->ev_full, h->comm_stream));
    HIPCHK(h, hipStreamWaitEvent(peer_stream(consumer_stream), h->ev_full, 0));
  } else {
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  }
  out->observations = h->d_obs_full; out->rows = (int64_t)h->world * h->B; out->obs_dim = h->obs_dim;
  out->rank = h->rank; out->world = h->world; out->reserved = 0;
  return GS_OK;
}

int gs_allgather_obs_download(gs_handle* h, double* obs_full_host) {
  if (!h || !obs_full_host) return fail(h, GS_E_INVALID, "handle / obs_full_host is NULL");
  gs_gathered_obs v;
  int rc = gs_allgather_obs_view(h, &v, nullptr);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(obs_full_host, v.observations, (size_t)v.rows * v.obs_dim * sizeof(double), hipMemcpyDeviceToHost));
  return GS_OK;
}

// What the communicator itself says about this member -- asked of RCCL (ncclCommCount / ncclCommUserRank /
// ncclCommCuDevice / ncclGetVersion), not echoed from the arguments of gs_comm_init --, and the device's UUID, so that a
// multi-rank run can show in its own output that N ranks on N different devices took part.
int gs_comm_info(gs_handle* h, gs_comm_info_t* out) {
  if (!h || !out) return fail(h, GS_E_INVALID, "handle / out is NULL");
  if (!h->comm && !h->loop) return fail(h, GS_E_STATE, "no communicator on this handle");
  memset(out, 0, sizeof *out);
  out->transport = h->comm ? 1 : 2;
  out->device = h->device;
  hipUUID uu;
  if (hipDeviceGetUuid(&uu, h->device) == hipSuccess) memcpy(out->device_uuid, uu.bytes, 16);
  if (h->loop) { out->nranks = h->loop->world; out->rank = h->rank; out->comm_device = h->device; return GS_OK; }
  int v = 0;
  out->nranks = -1; out->rank = -1; out->comm_device = -1;
  if (g_rccl.CommCount && g_rccl.CommCount(h->comm, &v) == 0) out->nranks = v;
  if (g_rccl.CommUserRank && g_rccl.CommUserRank(h->comm, &v) == 0) out->rank = v;
  if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(h->comm, &v) == 0) out->comm_device = v;
  if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == 0) out->rccl_version = v;
  return GS_OK;
}

int gs_comm_destroy(gs_handle* h) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  (void)hipSetDevice(h->device);
  if (h->loop) {      // nobody may still be copying out of this member's send block
    GsLoopComm* lc = h->loop;
    for (gs_handle* q : lc->member)
      if (q && q->comm_stream) { (void)hipSetDevice(q->device); (void)hipStreamSynchronize(q->comm_stream); }
    (void)hipSetDevice(h->device);
    lc->member[h->rank] = nullptr; h->loop = nullptr;
    if (--lc->n_live == 0) {
      for (hipEvent_t e : lc->ev_sent) if (e) (void)hipEventDestroy(e);
      for (hipEvent_t e : lc->ev_taken) if (e) (void)hipEventDestroy(e);
      delete lc;
    }
  }
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  h->gather_pending[0] = h->gather_pending[1] = false;
  if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  if (h->d_obs_full) { (void)hipFree(h->d_obs_full); h->d_obs_full = nullptr; }
  if (h->d_gather_send) { (void)hipFree(h->d_gather_send); h->d_gather_send = nullptr; }
  if (h->d_gather_recv) { (void)hipFree(h->d_gather_recv); h->d_gather_recv = nullptr; }
  h->rank = 0; h->world = 1;
  return GS_OK;
}

// ---- the meshed Newton-Raphson member's host schedule, without a device (mesh_schedule.h) -------------------------------
// header[0..15]: ok, n_levels, n_rows, max_rows_per_wave, n_pivots, msg_units, n_messages, n_accumulators, max_degree,
//                unit_bytes, zero_off, dummy_off, body_off, region_bytes, sizeof(GsMeshItem), n_adj;  why: the reason when ok == 0
int gs_mesh_schedule_dump(const gs_topology* topo, int32_t zero_z_mode, int32_t nw, int32_t ni, int32_t acc_cap, int32_t unit_budget, int32_t region_base,
                          int32_t slot_bytes, int32_t* header, char* why, int32_t why_cap, void* items, int32_t* rowinfo,
                          int32_t* adj_off, double* adj_y) {
  if (!topo || !header) return fail(nullptr, GS_E_INVALID, "topology / header is NULL");
  if (topo->struct_size != (int32_t)sizeof(gs_topology)) return fail(nullptr, GS_E_INVALID, "struct_size mismatch");
  if (nw < 1 || ni < 1 || acc_cap < 1) return fail(nullptr, GS_E_INVALID, "bad arguments");
  HostTopology ht;
  const std::string err = gs_compile_topology(*topo, zero_z_mode, false, true, ht);
  if (!err.empty()) return fail(nullptr, GS_E_INVALID, "topology: %s", err.c_str());
  MeshSchedule S;
  gs_mesh_schedule(ht, nw, ni, 8, region_base, slot_bytes, acc_cap, unit_budget, S);
  const int32_t hd[16] = {S.ok ? 1 : 0, S.n_levels, S.n_rows, S.max_rows_per_wave, S.n_pivots, S.msg_units, S.n_messages, S.n_accumulators,
                          S.max_degree, S.unit_bytes, S.zero_off, S.dummy_off, S.body_off, S.region_bytes, (int32_t)sizeof(GsMeshItem), (int32_t)S.adj_off.size()};
  memcpy(header, hd, sizeof hd);
  if (why && why_cap > 0) { strncpy(why, S.why.c_str(), (size_t)why_cap - 1); why[why_cap - 1] = 0; }
  if (!S.ok) return GS_OK;
  if (items) memcpy(items, S.items.data(), S.items.size() * sizeof(MeshItem));
  if (rowinfo) memcpy(rowinfo, S.rowinfo.data(), S.rowinfo.size() * sizeof(int32_t));
  if (adj_off) memcpy(adj_off, S.adj_off.data(), S.adj_off.size() * sizeof(int32_t));
  if (adj_y) memcpy(adj_y, S.adj_y.data(), S.adj_y.size() * sizeof(double));
  return GS_OK;
}

// The same schedule in the form the kernel reads (GS_MESH_W_*): counts[4] = n_pairs, ytab doubles, adj_ent entries, item words
int gs_mesh_schedule_dump_packed(const gs_topology* topo, int32_t zero_z_mode, int32_t nw, int32_t ni, int32_t acc_cap, int32_t unit_budget, int32_t region_base,
                                 int32_t slot_bytes, int32_t* counts, int32_t* packed, int32_t* rowinfo, double* ytab, int32_t* adj_ent) {
  if (!topo || !counts) return fail(nullptr, GS_E_INVALID, "topology / counts is NULL");
  if (topo->struct_size != (int32_t)sizeof(gs_topology)) return fail(nullptr, GS_E_INVALID, "struct_size mismatch");
  HostTopology ht;
  const std::string err = gs_compile_topology(*topo, zero_z_mode, false, true, ht);
  if (!err.empty()) return fail(nullptr, GS_E_INVALID, "topology: %s", err.c_str());
  MeshSchedule S;
  gs_mesh_schedule(ht, nw, ni, 8, region_base, slot_bytes, acc_cap, unit_budget, S);
  if (!S.ok) return fail(nullptr, GS_E_TOPOLOGY, "%s", S.why.c_str());
  counts[0] = S.n_pairs; counts[1] = (int32_t)S.ytab.size(); counts[2] = (int32_t)S.adj_ent.size(); counts[3] = GS_MESH_WORDS;
  if (packed) memcpy(packed, S.packed.data(), S.packed.size() * sizeof(int32_t));
  if (rowinfo) memcpy(rowinfo, S.rowinfo_packed.data(), S.rowinfo_packed.size() * sizeof(int32_t));
  if (ytab) memcpy(ytab, S.ytab.data(), S.ytab.size() * sizeof(double));
  if (adj_ent) memcpy(adj_ent, S.adj_ent.data(), S.adj_ent.size() * sizeof(int32_t));
  return GS_OK;
}

// The flat-start Newton map of the meshed member's iteration 0 (GsF2Tables::mesh_w), host arithmetic only (no GPU): out = W as a plain
// row-major [2 (n - 1)][n] matrix (column n - 1: the constant term; rows = (d theta, d|V|) of the non-slack buses in bus order), so that
// x = W [P_spec of the non-slack buses in bus order; 1].  Test aid.  GS_E_TOPOLOGY: the network has a bus that is neither the slack nor PQ,
// or the flat-start Jacobian is singular.
int gs_flat_newton_map_dump(const gs_topology* topo, int32_t zero_z_mode, double* out) {
  if (!topo || !out) return fail(nullptr, GS_E_INVALID, "topology / out is NULL");
  if (topo->struct_size != (int32_t)sizeof(gs_topology)) return fail(nullptr, GS_E_INVALID, "struct_size mismatch");
  HostTopology ht;
  const std::string err = gs_compile_topology(*topo, zero_z_mode, false, true, ht);
  if (!err.empty()) return fail(nullptr, GS_E_INVALID, "topology: %s", err.c_str());
  for (int i = 0; i < ht.n; ++i)
    if (i != ht.slack && !(ht.th_free[i] && ht.vm_free[i])) return fail(nullptr, GS_E_TOPOLOGY, "a bus other than the slack is not a PQ bus");
  const int na = ht.n - 1, N2 = 2 * na, K = na + 1, tiles = (N2 + 15) / 16, steps = (K + 3) / 4;
  std::vector<double> wt;
  if (!flat_newton_map(ht, tiles, steps, wt)) return fail(nullptr, GS_E_TOPOLOGY, "the flat-start Jacobian is singular");
  for (int u = 0; u < N2; ++u)
    for (int k = 0; k < K; ++k) out[(size_t)u * K + k] = wt[((size_t)(u / 16) * steps + k / 4) * 64 + (u % 16) + 16 * (k % 4)];
  return GS_OK;
}

// ---- measurement ------------------------------------------------------------------------------------
int gs_debug_stamps(gs_handle* h, uint64_t* cycles_out, int32_t n) {
  if (!h || !cycles_out || n < 1 || n > 16) return fail(h, GS_E_INVALID, "bad arguments");
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!h->d_stamps) {
    int rc = dev_alloc(h, &h->d_stamps, 16 + 2 * GS_STAMP_BLOCKS);
    if (rc) return rc;
    HIPCHK(h, hipMemset(h->d_stamps, 0, (16 + 2 * GS_STAMP_BLOCKS) * sizeof(unsigned long long)));
    h->SC.stamps = h->d_stamps;
    h->DA.stamps = h->d_stamps;
    h->SA.stamps = h->d_stamps;
    h->SC.stamp_wave = getenv("GS_STAMP_WAVE") ? atoi(getenv("GS_STAMP_WAVE")) : 0;
    h->SC.block_times = getenv("GS_STAMP_BLOCK_TIMES") ? 1 : 0;
    for (int k = 0; k < n; ++k) cycles_out[k] = 0;
    return GS_OK;
  }
  unsigned long long tmp[16];
  HIPCHK(h, hipMemcpy(tmp, h->d_stamps, sizeof tmp, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->d_stamps, 0, sizeof tmp));
  for (int k = 0; k < n; ++k) cycles_out[k] = tmp[k];
  return GS_OK;
}

// (start, end) of every workgroup of the LAST step launch on the 100 MHz real-time clock (flow2 kernels, armed by
// gs_debug_stamps with GS_STAMP_BLOCK_TIMES set); returns the pairs of the first n_blocks workgroups
int gs_debug_block_times(gs_handle* h, uint64_t* out, int32_t n_blocks) {
  if (!h || !out || n_blocks < 1 || n_blocks > GS_STAMP_BLOCKS) return fail(h, GS_E_INVALID, "bad arguments");
  if (!h->d_stamps) return fail(h, GS_E_STATE, "gs_debug_stamps has not armed the buffer");
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_stamps + 16, (size_t)n_blocks * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->d_stamps + 16, 0, (size_t)2 * GS_STAMP_BLOCKS * sizeof(unsigned long long)));
  return GS_OK;
}

int gs_timing_enable(gs_handle* h, int32_t on) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->timing = on == 1;
  h->timing_span = on == 2;
  h->span_open = false;
  h->timed_used = 0;
  return GS_OK;
}

int gs_timing_read(gs_handle* h, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return fail(h, GS_E_INVALID, "bad arguments");
  GS_ENTER(h);
  if (h->timing_span) {           // call right after the last launch of the region: the closing event goes behind it on the stream
    for (int k = 0; k < GS_K_COUNT; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
    if (h->span_open) {
      HIPCHK(h, hipEventRecord(h->span_b, h->stream));
      HIPCHK(h, hipEventSynchronize(h->span_b));
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->span_a, h->span_b) == hipSuccess) {
        // the whole span is booked on the kernel that was launched most (the step / solve kernel of a measurement loop)
        int best = 0;
        for (int k = 1; k < GS_K_COUNT; ++k) if (h->span_launches[k] > h->span_launches[best]) best = k;
        total_ms[best] = ms;
        for (int k = 0; k < GS_K_COUNT; ++k) launches[k] = h->span_launches[k];
      }
      h->span_open = false;
    }
    return GS_OK;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < GS_K_COUNT; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
  for (size_t i = 0; i < h->timed_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->timed[i].a, h->timed[i].b) == hipSuccess) {
      total_ms[h->timed[i].kid] += ms; launches[h->timed[i].kid] += 1;
    }
  }
  h->timed_used = 0;
  return GS_OK;
}


static int debug_rows_map(gs_handle* h, int32_t which, std::vector<int32_t>& map) {
  const GsRows& R = h->R;
  const int row0[GS_ROWS_COUNT] = {R.VM.base, R.LOAD, R.ENVLOAD.base, R.FLOW.base, R.FREQ, R.CONV, R.ITERS, R.MAXMIS, R.LOADP};
  const int stride[GS_ROWS_COUNT] = {2, 1, 2, 2, 1, 1, 1, 1, 1};
  const int width[GS_ROWS_COUNT] = {h->n, h->m, h->m, h->m, 1, 1, 1, 1, h->n_loads};
  map.resize(width[which]);
  for (int k = 0; k < width[which]; ++k) map[k] = row0[which] + stride[which] * k;
  return width[which];
}

int gs_debug_write_rows(gs_handle* h, int32_t which, const double* values) {
  if (!h || !values || which < 0 || which >= GS_ROWS_COUNT) return fail(h, GS_E_INVALID, "bad arguments");
  GS_ENTER(h);
  { int rc0 = ensure_rows(h); if (rc0) return rc0; }
  std::vector<int32_t> map;
  const int C = debug_rows_map(h, which, map);
  if (C <= 0) return GS_OK;
  if ((size_t)h->B * C > h->in_doubles) return fail(h, GS_E_INVALID, "staging buffer too small");
  int32_t* dmap = nullptr;
  HIPCHK(h, hipMalloc((void**)&dmap, C * sizeof(int32_t)));
  int rc = GS_OK;
  if (hipMemcpyAsync(dmap, map.data(), C * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(h, GS_E_HIP, "map upload failed");
  if (!rc) rc = unpack_from_host(h, dmap, C, values);
  const hipError_t e = hipStreamSynchronize(h->stream);
  (void)hipFree(dmap);
  if (!rc && e != hipSuccess) rc = fail(h, GS_E_HIP, "row write failed");
  return rc;
}

int gs_debug_read_rows(gs_handle* h, int32_t which, double* values) {
  if (!h || !values || which < 0 || which >= GS_ROWS_COUNT) return fail(h, GS_E_INVALID, "bad arguments");
  GS_ENTER(h);
  { int rc0 = ensure_rows(h); if (rc0) return rc0; }
  std::vector<int32_t> map;
  const int C = debug_rows_map(h, which, map);
  if (C <= 0) return GS_OK;
  if ((size_t)h->B * C > h->out_doubles) return fail(h, GS_E_INVALID, "staging buffer too small");
  int32_t* dmap = nullptr;
  HIPCHK(h, hipMalloc((void**)&dmap, C * sizeof(int32_t)));
  int rc = GS_OK;
  if (hipMemcpyAsync(dmap, map.data(), C * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(h, GS_E_HIP, "map upload failed");
  if (!rc) rc = pack_to_host(h, dmap, C, values);
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(dmap);
  return rc;
}

// ---- linear-approximation fallback ------------------------------------------------------------------
int gs_fallback_linear(gs_handle* h, const double* load_w, const double* gen_w, const double* total_load,
                       const double* total_gen, const uint8_t* mask, uint8_t* applied_out, int32_t* n_applied) {
  if (!h) return fail(nullptr, GS_E_INVALID, "handle is NULL");
  if ((load_w == nullptr) != (gen_w == nullptr)) return fail(h, GS_E_INVALID, "load_w and gen_w go together");
  if ((total_load == nullptr) != (total_gen == nullptr)) return fail(h, GS_E_INVALID, "total_load and total_gen go together");
  if (!load_w && total_load) return fail(h, GS_E_INVALID, "totals without per-bus arrays: with the device state the sums are formed on the device");
  if (!load_w && !h->was_reset) return fail(h, GS_E_STATE, "no environment state on the device: call gs_reset first or pass load_w / gen_w");
  GS_ENTER(h);
  { int rc0 = ensure_rows(h); if (rc0) return rc0; }      // only the selected instances are overwritten: the others' rows must be current
  const int B = h->B, n = h->n;
  int rc = GS_OK;
  if (!h->fb_ready) {
    const HostTopology& ht = h->topo;
    // the order in which _calculate_power_injections fills its dicts (grid_env.py:683-720): load buses by first
    // appearance in the load list, then battery buses not seen before (a charging battery adds a load entry);
    // generator buses likewise, then battery buses (a discharging battery adds a generation entry)
    auto order_of = [&](const std::vector<int32_t>& ptr, const std::vector<int32_t>& idx, int count) {
      std::vector<int32_t> bus_of(count, 0), out; std::vector<char> seen(n, 0);
      for (int i = 0; i < n; ++i) for (int p = ptr[i]; p < ptr[i + 1]; ++p) bus_of[idx[p]] = i;
      for (int d = 0; d < count; ++d) if (!seen[bus_of[d]]) { seen[bus_of[d]] = 1; out.push_back(bus_of[d]); }
      std::vector<int32_t> bat_bus(h->n_bats, 0);
      for (int i = 0; i < n; ++i) for (int p = ht.bb_ptr[i]; p < ht.bb_ptr[i + 1]; ++p) bat_bus[ht.bb_idx[p]] = i;
      for (int q = 0; q < h->n_bats; ++q) if (!seen[bat_bus[q]]) { seen[bat_bus[q]] = 1; out.push_back(bat_bus[q]); }
      return out; };
    const std::vector<int32_t> lo = order_of(ht.bl_ptr, ht.bl_idx, h->n_loads), go = order_of(ht.bg_ptr, ht.bg_idx, h->n_gens);
    if ((rc = dev_upload(h, &h->FB.load_order, lo)) || (rc = dev_upload(h, &h->FB.gen_order, go)) ||
        (rc = dev_upload(h, &h->FB.line_x, h->line_x))) return rc;
    h->FB.n_load_order = (int32_t)lo.size(); h->FB.n_gen_order = (int32_t)go.size();
    if ((rc = dev_alloc(h, &h->fb_load, (size_t)B * n)) || (rc = dev_alloc(h, &h->fb_gen, (size_t)B * n)) ||
        (rc = dev_alloc(h, &h->fb_tl, (size_t)B)) || (rc = dev_alloc(h, &h->fb_tg, (size_t)B)) ||
        (rc = dev_alloc(h, &h->fb_mask, (size_t)B)) || (rc = dev_alloc(h, &h->fb_applied, (size_t)B))) return rc;
    h->fb_ready = true;
  }
  GsFallbackArgs A = h->FB;
  A.env_mode = load_w ? 0 : 1;
  if (load_w) {
    HIPCHK(h, hipMemcpyAsync(h->fb_load, load_w, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->fb_gen, gen_w, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    A.load_w = h->fb_load; A.gen_w = h->fb_gen;
    if (total_load) {
      HIPCHK(h, hipMemcpyAsync(h->fb_tl, total_load, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(h->fb_tg, total_gen, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
      A.tot_load = h->fb_tl; A.tot_gen = h->fb_tg;
    }
  }
  if (mask) { HIPCHK(h, hipMemcpyAsync(h->fb_mask, mask, (size_t)B, hipMemcpyHostToDevice, h->stream)); A.mask = h->fb_mask; }
  A.applied = h->fb_applied;
  hipLaunchKernelGGL(gs_k_fallback_linear, dim3(h->groups), dim3(64), 0, h->stream, h->T, h->R, A, h->slab, B);
  HIPCHK(h, hipGetLastError());
  std::vector<int32_t> ap(B);
  HIPCHK(h, hipMemcpyAsync(ap.data(), h->fb_applied, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int32_t cnt = 0;
  for (int b = 0; b < B; ++b) { cnt += ap[b] != 0; if (applied_out) applied_out[b] = ap[b] != 0; }
  if (n_applied) *n_applied = cnt;
  return GS_OK;
}

// ---- post-step checks -------------------------------------------------------------------------------
struct gs_checks {
  gs_handle* h = nullptr;
  GsChecksCfg C{};
  double* prev = nullptr; int32_t* state = nullptr; int32_t* out_i = nullptr; double* out_f = nullptr;
  uint8_t *bus_mask = nullptr, *line_mask = nullptr; double* freq = nullptr; bool use_freq = false, want_masks = true;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t ev_used = 0;
};

}  // extern "C"  (helpers below have C++ linkage)
namespace {
GsFusedChecks fused_checks_args(gs_handle* h) {
  GsFusedChecks f{};
  if (h->fused) {
    gs_checks* c = h->fused;
    f.C = c->C; f.prev = c->prev; f.state = c->state; f.out_i = c->out_i; f.out_f = c->out_f;
    f.bus_mask = c->want_masks ? c->bus_mask : nullptr; f.line_mask = c->want_masks ? c->line_mask : nullptr;
    f.enabled = 1; f.Bp = h->Bp;
  }
  return f;
}
}  // namespace
extern "C" {

int gs_checks_set_fused(gs_checks* c, int32_t on, int32_t want_masks) {
  if (!c) return fail(nullptr, GS_E_INVALID, "checks object is NULL");
  gs_handle* h = c->h;
  if (on && (h->n >= 65536 || h->m >= 65536)) return fail(h, GS_E_INVALID, "fused checks count in 16 bits: fewer than 65536 buses and lines");
  if (on && h->fused && h->fused != c) return fail(h, GS_E_STATE, "another checks object is already fused into this handle's step");
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  c->want_masks = want_masks != 0;
  if (on) h->fused = c; else if (h->fused == c) h->fused = nullptr;
  return GS_OK;
}

int gs_checks_create(gs_handle* h, const gs_checks_config* cfg, gs_checks** out) {
  if (!h || !cfg || !out) return fail(h, GS_E_INVALID, "handle / config / out is NULL");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(gs_checks_config)) return fail(h, GS_E_INVALID, "gs_checks_config.struct_size mismatch");
  if (!(cfg->timestep > 0.0)) return fail(h, GS_E_INVALID, "timestep must be positive");
  if (cfg->loading_source != 0 && cfg->loading_source != 1) return fail(h, GS_E_INVALID, "loading_source must be 0 or 1");
  GS_ENTER(h);
  gs_checks* c = new gs_checks();
  c->h = h;
  GsChecksCfg& C = c->C;
  C.c_vlo = cfg->voltage_limits[0]; C.c_vhi = cfg->voltage_limits[1]; C.c_flo = cfg->frequency_limits[0]; C.c_fhi = cfg->frequency_limits[1];
  C.c_load = cfg->line_loading_limit; C.c_rocv = cfg->rate_voltage; C.c_rocf = cfg->rate_frequency; C.dt = cfg->timestep;
  C.m_vlo = cfg->mon_voltage_limits[0]; C.m_vhi = cfg->mon_voltage_limits[1]; C.m_flo = cfg->mon_frequency_limits[0]; C.m_fhi = cfg->mon_frequency_limits[1];
  C.m_load = cfg->mon_line_loading_limit; C.m_evlo = cfg->mon_emergency_voltage[0]; C.m_evhi = cfg->mon_emergency_voltage[1];
  C.m_eflo = cfg->mon_emergency_frequency[0]; C.m_efhi = cfg->mon_emergency_frequency[1];
  C.q_tol = cfg->quality_tolerance;
  C.n = h->n; C.m = h->m; C.rows_total = h->R.total;
  C.row_vm = h->R.VM.base; C.row_qload = h->R.LOAD; C.row_flow = h->R.FLOW.base;
  C.row_cload = cfg->loading_source ? h->R.ENVLOAD.base : h->R.LOAD; C.stride_cload = cfg->loading_source ? 2 : 1;
  C.row_freq = h->R.FREQ; C.row_conv = h->R.CONV; C.row_iters = h->R.ITERS; C.row_maxmis = h->R.MAXMIS;
  const size_t Bp = h->Bp;
  bool ok = hipMalloc((void**)&c->prev, (size_t)h->groups * (h->n + 1) * GS_LANES * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&c->state, 3 * Bp * sizeof(int32_t)) == hipSuccess &&
            hipMalloc((void**)&c->out_i, (size_t)GS_CI_COUNT * Bp * sizeof(int32_t)) == hipSuccess &&
            hipMalloc((void**)&c->out_f, (size_t)GS_CF_COUNT * Bp * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&c->bus_mask, std::max<size_t>(1, (size_t)h->groups * h->n * GS_LANES)) == hipSuccess &&
            hipMalloc((void**)&c->line_mask, std::max<size_t>(1, (size_t)h->groups * h->m * GS_LANES)) == hipSuccess &&
            hipMalloc((void**)&c->freq, Bp * sizeof(double)) == hipSuccess;
  ok = ok && hipMemset(c->prev, 0, (size_t)h->groups * (h->n + 1) * GS_LANES * sizeof(double)) == hipSuccess &&
       hipMemset(c->state, 0, 3 * Bp * sizeof(int32_t)) == hipSuccess && hipMemset(c->out_i, 0, (size_t)GS_CI_COUNT * Bp * sizeof(int32_t)) == hipSuccess &&
       hipMemset(c->out_f, 0, (size_t)GS_CF_COUNT * Bp * sizeof(double)) == hipSuccess;
  if (!ok) { gs_checks_destroy(c); return fail(h, GS_E_NOMEM, "device allocation for the checks failed"); }
  *out = c;
  return GS_OK;
}

void gs_checks_destroy(gs_checks* c) {
  if (!c) return;
  if (c->h->fused == c) c->h->fused = nullptr;
  (void)hipSetDevice(c->h->device);
  (void)hipStreamSynchronize(c->h->stream);
  for (auto& e : c->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (void* p : {(void*)c->prev, (void*)c->state, (void*)c->out_i, (void*)c->out_f, (void*)c->bus_mask, (void*)c->line_mask, (void*)c->freq})
    if (p) (void)hipFree(p);
  delete c;
}

int gs_checks_set_frequency(gs_checks* c, const double* f) {
  if (!c) return fail(nullptr, GS_E_INVALID, "checks object is NULL");
  gs_handle* h = c->h;
  GS_ENTER(h);
  c->use_freq = f != nullptr;
  if (f) { HIPCHK(h, hipMemcpyAsync(c->freq, f, (size_t)h->B * sizeof(double), hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); }
  return GS_OK;
}

int gs_checks_run(gs_checks* c) {
  if (!c) return fail(nullptr, GS_E_INVALID, "checks object is NULL");
  gs_handle* h = c->h;
  GS_ENTER(h);
  { int rc0 = ensure_rows(h); if (rc0) return rc0; }      // the kernel reads the |V| / loading / flow rows of the last step
  // HIP events only while somebody reads them (gs_checks_timing_enable), and never more than GS_CHECKS_MAX_EVENTS pairs:
  // a per-step safety check over a long run must not grow an event list without bound
  std::pair<hipEvent_t, hipEvent_t>* e = nullptr;
  if (c->timing && c->ev_used < GS_CHECKS_MAX_EVENTS) {
    if (c->ev_used == c->ev.size()) {
      hipEvent_t a, b2;
      HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b2));
      c->ev.emplace_back(a, b2);
    }
    e = &c->ev[c->ev_used++];
    HIPCHK(h, hipEventRecord(e->first, h->stream));
  }
  hipLaunchKernelGGL(gs_k_checks, dim3(h->groups), dim3(1024), 0, h->stream, c->C, h->slab, c->use_freq ? c->freq : (const double*)nullptr,
                     c->prev, c->state, c->out_i, c->out_f, c->bus_mask, c->line_mask, h->B, h->Bp);
  HIPCHK(h, hipGetLastError());
  if (e) HIPCHK(h, hipEventRecord(e->second, h->stream));
  return GS_OK;
}

int gs_checks_download(gs_checks* c, const gs_checks_view* out) {
  if (!c || !out) return fail(c ? c->h : nullptr, GS_E_INVALID, "checks object / view is NULL");
  gs_handle* h = c->h;
  GS_ENTER(h);
  const size_t Bp = h->Bp, B = h->B;
  std::vector<int32_t> ti; std::vector<double> tf; std::vector<uint8_t> tb, tl;
  if (out->ints) { ti.resize((size_t)GS_CI_COUNT * Bp); HIPCHK(h, hipMemcpyAsync(ti.data(), c->out_i, ti.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream)); }
  if (out->reals) { tf.resize((size_t)GS_CF_COUNT * Bp); HIPCHK(h, hipMemcpyAsync(tf.data(), c->out_f, tf.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream)); }
  if (out->bus_mask && h->n) { tb.resize((size_t)h->groups * h->n * GS_LANES); HIPCHK(h, hipMemcpyAsync(tb.data(), c->bus_mask, tb.size(), hipMemcpyDeviceToHost, h->stream)); }
  if (out->line_mask && h->m) { tl.resize((size_t)h->groups * h->m * GS_LANES); HIPCHK(h, hipMemcpyAsync(tl.data(), c->line_mask, tl.size(), hipMemcpyDeviceToHost, h->stream)); }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (out->ints) for (int k = 0; k < GS_CI_COUNT; ++k) memcpy(out->ints + (size_t)k * B, ti.data() + (size_t)k * Bp, B * sizeof(int32_t));
  if (out->reals) for (int k = 0; k < GS_CF_COUNT; ++k) memcpy(out->reals + (size_t)k * B, tf.data() + (size_t)k * Bp, B * sizeof(double));
  auto untile = [&](const std::vector<uint8_t>& t, uint8_t* dst, int width) {     // [group][row][lane] -> [b][row]
    for (size_t b = 0; b < B; ++b) {
      const size_t g = b / GS_LANES, lane = b % GS_LANES;
      for (int r = 0; r < width; ++r) dst[b * width + r] = t[(g * width + r) * GS_LANES + lane];
    }
  };
  if (out->bus_mask && h->n) untile(tb, out->bus_mask, h->n);
  if (out->line_mask && h->m) untile(tl, out->line_mask, h->m);
  return GS_OK;
}

int gs_checks_reset(gs_checks* c, const uint8_t* mask) {
  if (!c) return fail(nullptr, GS_E_INVALID, "checks object is NULL");
  gs_handle* h = c->h;
  GS_ENTER(h);
  uint8_t* dmask = nullptr;
  if (mask) {
    HIPCHK(h, hipMalloc((void**)&dmask, h->B));
    if (hipMemcpyAsync(dmask, mask, h->B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { (void)hipFree(dmask); return fail(h, GS_E_HIP, "mask upload failed"); }
  }
  hipLaunchKernelGGL(gs_k_checks_reset, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, c->state, (const uint8_t*)dmask, h->B, h->Bp);
  const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(h->stream);
  if (dmask) (void)hipFree(dmask);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(h, GS_E_HIP, "checks reset failed");
  return GS_OK;
}

int gs_checks_timing_enable(gs_checks* c, int32_t on) {
  if (!c) return fail(nullptr, GS_E_INVALID, "checks object is NULL");
  GS_ENTER(c->h);
  HIPCHK(c->h, hipStreamSynchronize(c->h->stream));
  c->timing = on != 0; c->ev_used = 0;
  return GS_OK;
}

int gs_checks_timing_read(gs_checks* c, double* total_ms, int64_t* launches) {
  if (!c || !total_ms || !launches) return fail(c ? c->h : nullptr, GS_E_INVALID, "bad arguments");
  gs_handle* h = c->h;
  GS_ENTER(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *total_ms = 0.0; *launches = 0;
  for (size_t k = 0; k < c->ev_used; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[k].first, c->ev[k].second) == hipSuccess) { *total_ms += ms; *launches += 1; }
  }
  c->ev_used = 0;
  return GS_OK;
}

}  // extern "C"
