// gridstep3.hip -- three-phase unbalanced radial load flow (forward/backward sweep) for gfx950:
// BASELINE.json config 5 (8500-node feeder, batch ~1000).  Host side + kernels of the gs3_* ABI.
//
// NEW functionality: the reference advertises UnbalancedPowerFlow (README.md:187-197,
// API_REFERENCE.md:420) but ships no implementation, so nothing here restates reference code;
// the convergence test is the reference's power-mismatch criterion (environments/power_flow.py:
// 150-171) applied per phase, and in the balanced, uncoupled limit the answer reduces to the
// single-phase solution that IS pinned by the reference (tests/test_unbalanced.py).
//
// Mapping (differs from the single-phase kernels, which put one instance on each lane): one
// workgroup per instance, lanes over the nodes of a tree level.  Nodes are renumbered in
// breadth-first order, so a level is a contiguous index range, the children of a node are a
// contiguous range of the next level, and every per-node array is read and written coalesced.
// Per-instance state (18 doubles per node: V, S_spec, D = Z J; 1.2 MB at 8500 nodes) lives in HBM --
// this configuration is HBM-streaming by construction.  What one level hands
// to the next (J going up, V going down) travels through LDS; see gs3_solve_body for the sweep pair.
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gridstep.h"

namespace {

struct Topo3 {
  int32_t n, n_levels, cap;  // cap = widest level (LDS message rows)
  const int32_t* lvl_ptr;    // [n_levels + 1]; level 0 = the source alone
  const int4* idx;           // [n] {parent position, first child position, child count, phase mask} (level order)
  const double* zr;          // [9][n] series impedance of the upstream line (rows/cols of absent phases zeroed)
  const double* zi;
  double vsr[3], vsi[3];     // source voltage
};

// C_DR/C_DI: the voltage drop Z_t J_t of the upstream line.  C_JR/C_JI (line currents) exist only when the
// level messages cannot go through LDS (h->lds_bytes == 0); the state then has 24 rows per node.
enum { C_VR = 0, C_VI = 3, C_P = 6, C_Q = 9, C_DR = 12, C_DI = 15, C_COUNT = 18, C_JR = 18, C_JI = 21, C_COUNT_NOLDS = 24 };

// row base in SGPRs + one 32-bit byte offset per thread
#define ST(comp, t) (*(double*)((char*)(S + (size_t)(comp) * n) + ((unsigned)(t) << 3)))
#define ZT(tab, k, t) (*(const double*)((const char*)((tab) + (size_t)(k) * n) + ((unsigned)(t) << 3)))
#define GS3_CONST __attribute__((address_space(4)))

__device__ __forceinline__ double block_max(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = sh[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = fmax(r, sh[k]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = sh[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r += sh[k];
  __syncthreads();
  return r;
}

extern __shared__ double gs3_msg[];   // [2 (level parity)][6][cap]: J of a level on the way up, V on the way down

// Level barrier.  With LDS messages nothing a level writes to HBM is read by another thread (a node keeps
// its thread in both sweeps), so the barrier only has to order LDS and the loads prefetched for the next
// level stay in flight across it.
template <bool LDSMSG> __device__ __forceinline__ void level_barrier() {
  if (LDSMSG) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

struct UpIn { int4 ix; double vr[3], vi[3], p[3], q[3]; };                // what a node needs on the way up
struct DownIn { int4 ix; double dr[3], di[3], vr[3], vi[3], p[3], q[3]; };     // ... and on the way down

// One sweep pair per iteration, the power mismatch evaluated on the way DOWN.
//   backward (deepest level first):   J_t = -conj(S_spec / V_t) + sum_children J   (children through LDS)
//                                     D_t = Z_t J_t  -> HBM                         (the only use of J_t later)
//   forward  (root's children first): V_t = V_parent - D_t                          (parent through LDS)
// After a forward sweep V_t - V_parent = -Z_t J_t holds exactly, so the current the new voltages imply on
// line t is -J_t and the current drawn at node t is the injection current conj(S_spec / V_old) the backward
// sweep used: the mismatch S_spec - V_new conj(I_old) needs neither Y = Z^-1 nor the children's line
// currents, and it is the number the NEXT backward sweep of the textbook loop would report -- one sweep
// earlier.  The flat start is never written to memory: with V = V_source everywhere the implied currents
// are zero and the first mismatch is |S_spec|.
// Each level step prefetches the own rows of the next level before its barrier, so a step costs LDS
// latency plus arithmetic rather than a round trip to HBM.
template <bool LDSMSG>
__device__ __forceinline__ void gs3_solve_body(const Topo3& T, double* __restrict__ S, double tol, int max_it,
                                               double* sh, double& losses, double& mm, int& it_out, int& conv_out) {
  const int n = T.n, cap = T.cap, L = T.n_levels, tid = threadIdx.x, nth = blockDim.x;
  const GS3_CONST int32_t* lvl = (const GS3_CONST int32_t*)T.lvl_ptr;
  int iters = max_it, conv = 0;
  mm = INFINITY; losses = 0.0;
  if (tid < 3) { ST(C_VR + tid, 0) = T.vsr[tid]; ST(C_VI + tid, 0) = T.vsi[tid]; }     // the source row
  for (int it = 0; it < max_it; ++it) {
    const bool first = it == 0;
    auto load_up = [&](int t) {
      UpIn u;
      u.ix = (*(const int4*)((const char*)T.idx + ((unsigned)t << 4)));
      for (int ph = 0; ph < 3; ++ph) {
        const bool on = (u.ix.w >> ph) & 1;
        u.vr[ph] = first ? (on ? T.vsr[ph] : 0.0) : ST(C_VR + ph, t);
        u.vi[ph] = first ? (on ? T.vsi[ph] : 0.0) : ST(C_VI + ph, t);
        u.p[ph] = ST(C_P + ph, t); u.q[ph] = ST(C_Q + ph, t);
      }
      return u;
    };
    auto load_down = [&](int t) {
      DownIn d;
      d.ix = (*(const int4*)((const char*)T.idx + ((unsigned)t << 4)));
      for (int ph = 0; ph < 3; ++ph) {
        const bool on = (d.ix.w >> ph) & 1;
        d.dr[ph] = ST(C_DR + ph, t); d.di[ph] = ST(C_DI + ph, t);
        d.vr[ph] = first ? (on ? T.vsr[ph] : 0.0) : ST(C_VR + ph, t);
        d.vi[ph] = first ? (on ? T.vsi[ph] : 0.0) : ST(C_VI + ph, t);
        d.p[ph] = ST(C_P + ph, t); d.q[ph] = ST(C_Q + ph, t);
      }
      return d;
    };

    // ---- backward
    double lmax = 0.0, psrc = 0.0;
    UpIn un = {};
    { const int t = lvl[L - 1] + tid; if (L > 1 && t < lvl[L]) un = load_up(t); }
    for (int l = L - 1; l >= 1; --l) {
      const int t0 = lvl[l], t1 = lvl[l + 1];
      double* up = gs3_msg + (size_t)(l & 1) * 6 * cap;
      const double* dn = gs3_msg + (size_t)((l + 1) & 1) * 6 * cap;
      UpIn u = un;
      if (l > 1) { const int t = lvl[l - 1] + tid; if (t < t0) un = load_up(t); }
      for (int t = t0 + tid; t < t1; t += nth) {
        if (t != t0 + tid) u = load_up(t);
        // own injection current first: V, P, Q are dead before the impedance rows arrive
        double jr[3], ji[3];
        for (int ph = 0; ph < 3; ++ph) {
          jr[ph] = 0.0; ji[ph] = 0.0;
          if ((u.ix.w >> ph) & 1) {
            if (first) {
              const double dP = fabs(u.p[ph]), dQ = fabs(u.q[ph]);
              lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
            }
            const double rd = 1.0 / (u.vr[ph] * u.vr[ph] + u.vi[ph] * u.vi[ph]);
            jr[ph] = -(u.p[ph] * u.vr[ph] + u.q[ph] * u.vi[ph]) * rd;
            ji[ph] = -(u.p[ph] * u.vi[ph] - u.q[ph] * u.vr[ph]) * rd;
          }
        }
        asm volatile("" ::: "memory");
        double zr[9], zi[9];
        for (int k = 0; k < 9; ++k) { zr[k] = ZT(T.zr, k, t); zi[k] = ZT(T.zi, k, t); }
        for (int ch = u.ix.y; ch < u.ix.y + u.ix.z; ++ch)
          for (int ph = 0; ph < 3; ++ph) {
            if (LDSMSG) { jr[ph] += dn[ph * cap + (ch - t1)]; ji[ph] += dn[(3 + ph) * cap + (ch - t1)]; }
            else { jr[ph] += ST(C_JR + ph, ch); ji[ph] += ST(C_JI + ph, ch); }
          }
        for (int ph = 0; ph < 3; ++ph) {
          if (l == 1 && ((u.ix.w >> ph) & 1)) psrc += T.vsr[ph] * jr[ph] + T.vsi[ph] * ji[ph];     // the source's share of sum P_calc
          if (LDSMSG) { up[ph * cap + (t - t0)] = jr[ph]; up[(3 + ph) * cap + (t - t0)] = ji[ph]; }
          else { ST(C_JR + ph, t) = jr[ph]; ST(C_JI + ph, t) = ji[ph]; }
        }
        for (int r = 0; r < 3; ++r) {
          double ar = 0.0, ai = 0.0;
          for (int cc = 0; cc < 3; ++cc) {
            ar += zr[3 * r + cc] * jr[cc] - zi[3 * r + cc] * ji[cc];
            ai += zr[3 * r + cc] * ji[cc] + zi[3 * r + cc] * jr[cc];
          }
          ST(C_DR + r, t) = ar; ST(C_DI + r, t) = ai;
        }
      }
      level_barrier<LDSMSG>();
    }
    if (first) {
      mm = block_max(lmax, sh);
      losses = 0.0;
      if (!(mm < INFINITY) || mm < tol) {     // no sweep will follow: the answer is the flat start itself
        for (int t = 1 + tid; t < n; t += nth) {
          const int m = (*(const int4*)((const char*)T.idx + ((unsigned)t << 4))).w;
          for (int ph = 0; ph < 3; ++ph) {
            const bool on = (m >> ph) & 1;
            ST(C_VR + ph, t) = on ? T.vsr[ph] : 0.0; ST(C_VI + ph, t) = on ? T.vsi[ph] : 0.0;
          }
        }
        iters = 1; conv = mm < tol;
        break;
      }
    }

    // ---- forward, with the mismatch / losses at the new voltages
    lmax = 0.0;
    double psum = psrc;
    DownIn dnx = {};
    { const int t = lvl[1] + tid; if (L > 1 && t < lvl[2]) dnx = load_down(t); }
    for (int l = 1; l < L; ++l) {
      const int t0 = lvl[l], t1 = lvl[l + 1], p0 = lvl[l - 1];
      double* dnw = gs3_msg + (size_t)(l & 1) * 6 * cap;
      const double* upr = gs3_msg + (size_t)((l - 1) & 1) * 6 * cap;
      DownIn d = dnx;
      if (l + 1 < L) { const int t = t1 + tid; if (t < lvl[l + 2]) dnx = load_down(t); }
      for (int t = t0 + tid; t < t1; t += nth) {
        if (t != t0 + tid) d = load_down(t);
        const int m = d.ix.w, pt = d.ix.x;
        for (int r = 0; r < 3; ++r) {
          double pr, pi;
          if (l == 1) { pr = T.vsr[r]; pi = T.vsi[r]; }
          else if (LDSMSG) { pr = upr[r * cap + (pt - p0)]; pi = upr[(3 + r) * cap + (pt - p0)]; }
          else { pr = ST(C_VR + r, pt); pi = ST(C_VI + r, pt); }
          const bool on = (m >> r) & 1;
          const double wr = on ? pr - d.dr[r] : 0.0, wi = on ? pi - d.di[r] : 0.0;
          ST(C_VR + r, t) = wr; ST(C_VI + r, t) = wi;
          if (LDSMSG) { dnw[r * cap + (t - t0)] = wr; dnw[(3 + r) * cap + (t - t0)] = wi; }
          if (on) {
            const double rd = 1.0 / (d.vr[r] * d.vr[r] + d.vi[r] * d.vi[r]);
            const double ior = (d.p[r] * d.vr[r] + d.q[r] * d.vi[r]) * rd, ioi = (d.p[r] * d.vi[r] - d.q[r] * d.vr[r]) * rd;
            const double pc = wr * ior + wi * ioi, qc = wi * ior - wr * ioi;
            const double dP = fabs(d.p[r] - pc), dQ = fabs(d.q[r] - qc);
            lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
            psum += pc;
          }
        }
      }
      level_barrier<LDSMSG>();
    }
    if (it + 1 < max_it) {        // what the backward sweep of iteration it + 1 would find
      mm = block_max(lmax, sh);
      losses = block_sum(psum, sh);
      if (!(mm < INFINITY)) { iters = it + 2; break; }
      if (mm < tol) { iters = it + 2; conv = 1; break; }
    }
  }
  it_out = iters;
  conv_out = conv;
}

template <bool LDSMSG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
gs3_k_solve(Topo3 T, double* __restrict__ state, int B, double tol, int max_it, double* __restrict__ out_loss,
            double* __restrict__ out_mm, int32_t* __restrict__ out_it, uint8_t* __restrict__ out_conv) {
  __shared__ double sh[8];
  const int b = blockIdx.x;
  double losses, mm; int it, conv;
  gs3_solve_body<LDSMSG>(T, state + (size_t)b * (LDSMSG ? C_COUNT : C_COUNT_NOLDS) * T.n, tol, max_it, sh, losses, mm, it, conv);
  if (threadIdx.x == 0) {
    out_loss[b] = losses;
    out_mm[b] = mm;
    out_it[b] = it;
    out_conv[b] = (uint8_t)conv;
  }
}

// P/Q [B][n][3] in caller node order -> state rows in level order
extern "C" __global__ void __launch_bounds__(256)
gs3_k_scatter_in(int n, int rows, const int32_t* __restrict__ node_of, const double* __restrict__ P, const double* __restrict__ Q,
                 double* __restrict__ state) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double* S = state + (size_t)b * rows * n;
  const size_t src = ((size_t)b * n + node_of[t]) * 3;
  for (int ph = 0; ph < 3; ++ph) { ST(C_P + ph, t) = P[src + ph]; ST(C_Q + ph, t) = Q ? Q[src + ph] : 0.0; }
}

// V in level order -> [B][n][3] in caller node order
extern "C" __global__ void __launch_bounds__(256)
gs3_k_gather_out(int n, int rows, const int32_t* __restrict__ pos_of, const double* __restrict__ state, double* __restrict__ vre,
                 double* __restrict__ vim) {
  const int b = blockIdx.y;
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n) return;
  const double* S = state + (size_t)b * rows * n;
  const int t = pos_of[node];
  const size_t dst = ((size_t)b * n + node) * 3;
  for (int ph = 0; ph < 3; ++ph) { vre[dst + ph] = ST(C_VR + ph, t); vim[dst + ph] = ST(C_VI + ph, t); }
}

thread_local std::string g3_error;

}  // namespace

struct gs3_handle {
  int device = 0, n = 0, B = 0, n_levels = 0, max_width = 0, max_it = 50, threads = 256, lds_bytes = 0, rows = C_COUNT;
  double tol = 1e-6;
  hipStream_t stream = nullptr;
  Topo3 T{};
  std::vector<void*> allocs;
  int32_t *d_node_of = nullptr, *d_pos_of = nullptr;
  double *d_state = nullptr, *d_p = nullptr, *d_q = nullptr, *d_vre = nullptr, *d_vim = nullptr, *d_loss = nullptr, *d_mm = nullptr;
  int32_t* d_it = nullptr; uint8_t* d_conv = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t ev_used = 0;
  mutable std::string err;
};

namespace {

int fail3(gs3_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g3_error = buf;
  if (h) h->err = buf;
  return code;
}
#define HIP3(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail3((h), GS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

template <typename X> int alloc3(gs3_handle* h, X** p, size_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(X)) != hipSuccess) return fail3(h, GS_E_NOMEM, "hipMalloc of %zu bytes failed", count * sizeof(X));
  h->allocs.push_back(q); *p = (X*)q; return GS_OK;
}
template <typename X> int upload3(gs3_handle* h, const X** p, const std::vector<X>& v) {
  X* q = nullptr; int rc = alloc3(h, &q, v.size()); if (rc) return rc;
  if (!v.empty()) HIP3(h, hipMemcpy(q, v.data(), v.size() * sizeof(X), hipMemcpyHostToDevice));
  *p = q; return GS_OK;
}

// inverse of the sub-matrix of a complex 3x3 on the phases in `mask`; other rows/cols zero
void masked_inverse(const double zr[9], const double zi[9], int mask, double yr[9], double yi[9]) {
  int idx[3], k = 0;
  for (int ph = 0; ph < 3; ++ph) if ((mask >> ph) & 1) idx[k++] = ph;
  for (int q = 0; q < 9; ++q) { yr[q] = 0.0; yi[q] = 0.0; }
  // Gauss-Jordan on the k x k complex block
  double ar[3][6] = {}, ai[3][6] = {};
  for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) { ar[r][c] = zr[3 * idx[r] + idx[c]]; ai[r][c] = zi[3 * idx[r] + idx[c]]; }
  for (int r = 0; r < k; ++r) ar[r][k + r] = 1.0;
  for (int p = 0; p < k; ++p) {
    int best = p; double bm = -1.0;
    for (int r = p; r < k; ++r) { double mg = ar[r][p] * ar[r][p] + ai[r][p] * ai[r][p]; if (mg > bm) { bm = mg; best = r; } }
    for (int c = 0; c < 2 * k; ++c) { std::swap(ar[p][c], ar[best][c]); std::swap(ai[p][c], ai[best][c]); }
    const double dr = ar[p][p], di = ai[p][p], dd = dr * dr + di * di;
    for (int c = 0; c < 2 * k; ++c) { const double xr = ar[p][c], xi = ai[p][c]; ar[p][c] = (xr * dr + xi * di) / dd; ai[p][c] = (xi * dr - xr * di) / dd; }
    for (int r = 0; r < k; ++r) if (r != p) {
      const double fr = ar[r][p], fi = ai[r][p];
      for (int c = 0; c < 2 * k; ++c) { const double xr = ar[p][c], xi = ai[p][c]; ar[r][c] -= fr * xr - fi * xi; ai[r][c] -= fr * xi + fi * xr; }
    }
  }
  for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) { yr[3 * idx[r] + idx[c]] = ar[r][k + c]; yi[3 * idx[r] + idx[c]] = ai[r][k + c]; }
}

}  // namespace

extern "C" {

const char* gs3_last_error(const gs3_handle* h) { return h ? h->err.c_str() : g3_error.c_str(); }

int gs3_create(const gs3_topology* t, double tolerance, int32_t max_iterations, int32_t batch, int32_t device, gs3_handle** out) {
  if (!out) return fail3(nullptr, GS_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!t || t->struct_size != (int32_t)sizeof(gs3_topology)) return fail3(nullptr, GS_E_INVALID, "gs3_topology missing or struct_size mismatch");
  const int n = t->n;
  if (n < 2 || batch < 1 || max_iterations < 1 || !t->parent || !t->phases || !t->z_re || !t->z_im || !t->v_source)
    return fail3(nullptr, GS_E_INVALID, "bad arguments");
  if (t->source < 0 || t->source >= n || t->parent[t->source] != -1) return fail3(nullptr, GS_E_TOPOLOGY, "source must have parent -1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail3(nullptr, GS_E_NO_DEVICE, "no HIP device visible: libgridstep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail3(nullptr, GS_E_NO_DEVICE, "device %d out of range", device);
  // breadth-first level order
  std::vector<std::vector<int>> kids(n);
  for (int i = 0; i < n; ++i) {
    if (i == t->source) continue;
    const int p = t->parent[i];
    if (p < 0 || p >= n) return fail3(nullptr, GS_E_TOPOLOGY, "node %d has no valid parent", i);
    if ((t->phases[i] & ~t->phases[p]) != 0 || t->phases[i] == 0 || t->phases[i] > 7)
      return fail3(nullptr, GS_E_TOPOLOGY, "phases of node %d are not a non-empty subset of its parent's", i);
    kids[p].push_back(i);
  }
  if (t->phases[t->source] != 7) return fail3(nullptr, GS_E_TOPOLOGY, "the source must carry all three phases");
  std::vector<int32_t> node_of, pos_of(n, -1), lvl_ptr{0};
  node_of.push_back(t->source); pos_of[t->source] = 0;
  size_t head = 0;
  while (head < node_of.size()) {
    const size_t end = node_of.size();
    lvl_ptr.push_back((int)end);
    for (; head < end; ++head)
      for (int c : kids[node_of[head]]) { pos_of[c] = (int)node_of.size(); node_of.push_back(c); }
  }
  if ((int)node_of.size() != n) return fail3(nullptr, GS_E_TOPOLOGY, "network is not a tree rooted at the source (%zu of %d nodes reachable)", node_of.size(), n);
  while (lvl_ptr.size() >= 2 && lvl_ptr[lvl_ptr.size() - 1] == lvl_ptr[lvl_ptr.size() - 2]) lvl_ptr.pop_back();
  const int n_levels = (int)lvl_ptr.size() - 1;

  gs3_handle* h = new gs3_handle();
  h->device = device; h->n = n; h->B = batch; h->tol = tolerance; h->max_it = max_iterations; h->n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) h->max_width = std::max(h->max_width, lvl_ptr[l + 1] - lvl_ptr[l]);
  std::vector<int32_t> par(n, 0), cfirst(n, 0), ccount(n, 0), mask(n, 7);
  std::vector<double> zr((size_t)9 * n, 0.0), zi((size_t)9 * n, 0.0);
  for (int tt = 0; tt < n; ++tt) {
    const int node = node_of[tt];
    mask[tt] = t->phases[node];
    par[tt] = node == t->source ? 0 : pos_of[t->parent[node]];
    ccount[tt] = (int)kids[node].size();
    cfirst[tt] = ccount[tt] ? pos_of[kids[node][0]] : 0;
    if (node == t->source) continue;
    double a[9], bb[9], ya[9], yb[9];
    for (int q = 0; q < 9; ++q) {
      const bool on = ((mask[tt] >> (q / 3)) & 1) && ((mask[tt] >> (q % 3)) & 1);
      a[q] = on ? t->z_re[(size_t)node * 9 + q] : 0.0; bb[q] = on ? t->z_im[(size_t)node * 9 + q] : 0.0;
    }
    masked_inverse(a, bb, mask[tt], ya, yb);     // validation only: the sweeps never need Y
    for (int q = 0; q < 9; ++q) {
      zr[(size_t)q * n + tt] = a[q]; zi[(size_t)q * n + tt] = bb[q];
      if (!std::isfinite(ya[q]) || !std::isfinite(yb[q])) { delete h; return fail3(nullptr, GS_E_TOPOLOGY, "line into node %d has a singular impedance block", node); }
    }
  }
  auto bail = [&](int rc) { gs3_destroy(h); return rc; };
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail3(nullptr, GS_E_HIP, "device / stream setup failed"));
  Topo3& T = h->T;
  T.n = n; T.n_levels = n_levels; T.cap = h->max_width;
  const double ang[3] = {0.0, -2.0 * M_PI / 3.0, 2.0 * M_PI / 3.0};
  for (int ph = 0; ph < 3; ++ph) { T.vsr[ph] = t->v_source[ph] * std::cos(ang[ph]); T.vsi[ph] = t->v_source[ph] * std::sin(ang[ph]); }
  int rc;
  std::vector<int4> idx(n);
  for (int tt = 0; tt < n; ++tt) idx[tt] = make_int4(par[tt], cfirst[tt], ccount[tt], mask[tt]);
  if ((rc = upload3(h, &T.lvl_ptr, lvl_ptr)) || (rc = upload3(h, &T.idx, idx)) || (rc = upload3(h, &T.zr, zr)) || (rc = upload3(h, &T.zi, zi)))
    return bail(rc);
  // level messages through LDS when two parities of the widest level fit beside three other resident workgroups
  h->lds_bytes = 2 * 6 * h->max_width * (int)sizeof(double);
  if (h->lds_bytes > 38 * 1024 || getenv("GS3_NO_LDS")) h->lds_bytes = 0;
  h->rows = h->lds_bytes ? C_COUNT : C_COUNT_NOLDS;
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, node_of))) return bail(rc); h->d_node_of = const_cast<int32_t*>(q); }
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, pos_of))) return bail(rc); h->d_pos_of = const_cast<int32_t*>(q); }
  if (const char* e = getenv("GS3_THREADS")) h->threads = std::max(64, std::min(256, atoi(e) / 64 * 64));
  const size_t bn3 = (size_t)batch * n * 3;
  if ((rc = alloc3(h, &h->d_state, (size_t)batch * h->rows * n)) || (rc = alloc3(h, &h->d_p, bn3)) || (rc = alloc3(h, &h->d_q, bn3)) ||
      (rc = alloc3(h, &h->d_vre, bn3)) || (rc = alloc3(h, &h->d_vim, bn3)) || (rc = alloc3(h, &h->d_loss, batch)) ||
      (rc = alloc3(h, &h->d_mm, batch)) || (rc = alloc3(h, &h->d_it, batch)) || (rc = alloc3(h, &h->d_conv, batch)))
    return bail(rc);
  if (hipMemset(h->d_state, 0, (size_t)batch * h->rows * n * sizeof(double)) != hipSuccess) return bail(fail3(nullptr, GS_E_HIP, "hipMemset failed"));
  *out = h;
  return GS_OK;
}

void gs3_destroy(gs3_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int gs3_upload_injections(gs3_handle* h, const double* P, const double* Q) {
  if (!h || !P) return fail3(h, GS_E_INVALID, "handle / P_spec is NULL");
  HIP3(h, hipSetDevice(h->device));
  const size_t bytes = (size_t)h->B * h->n * 3 * sizeof(double);
  HIP3(h, hipMemcpyAsync(h->d_p, P, bytes, hipMemcpyHostToDevice, h->stream));
  if (Q) HIP3(h, hipMemcpyAsync(h->d_q, Q, bytes, hipMemcpyHostToDevice, h->stream));
  dim3 grid((h->n + 255) / 256, h->B);
  hipLaunchKernelGGL(gs3_k_scatter_in, grid, dim3(256), 0, h->stream, h->n, h->rows, h->d_node_of, h->d_p, Q ? h->d_q : (const double*)nullptr, h->d_state);
  HIP3(h, hipGetLastError());
  HIP3(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs3_solve_device(gs3_handle* h) {
  if (!h) return fail3(nullptr, GS_E_INVALID, "handle is NULL");
  HIP3(h, hipSetDevice(h->device));
  if (h->ev_used == h->ev.size()) {
    hipEvent_t a, b2;
    HIP3(h, hipEventCreate(&a)); HIP3(h, hipEventCreate(&b2));
    h->ev.emplace_back(a, b2);
  }
  auto& e = h->ev[h->ev_used++];
  HIP3(h, hipEventRecord(e.first, h->stream));
  if (h->lds_bytes)
    hipLaunchKernelGGL(gs3_k_solve<true>, dim3(h->B), dim3(h->threads), h->lds_bytes, h->stream, h->T, h->d_state, h->B, h->tol, h->max_it, h->d_loss, h->d_mm, h->d_it, h->d_conv);
  else
    hipLaunchKernelGGL(gs3_k_solve<false>, dim3(h->B), dim3(h->threads), 0, h->stream, h->T, h->d_state, h->B, h->tol, h->max_it, h->d_loss, h->d_mm, h->d_it, h->d_conv);
  HIP3(h, hipGetLastError());
  HIP3(h, hipEventRecord(e.second, h->stream));
  return GS_OK;
}

int gs3_download_solution(gs3_handle* h, const gs3_solution_view* out) {
  if (!h || !out) return fail3(h, GS_E_INVALID, "handle / view is NULL");
  HIP3(h, hipSetDevice(h->device));
  const size_t bn3 = (size_t)h->B * h->n * 3;
  if (out->v_re || out->v_im) {
    dim3 grid((h->n + 255) / 256, h->B);
    hipLaunchKernelGGL(gs3_k_gather_out, grid, dim3(256), 0, h->stream, h->n, h->rows, h->d_pos_of, h->d_state, h->d_vre, h->d_vim);
    HIP3(h, hipGetLastError());
    if (out->v_re) HIP3(h, hipMemcpyAsync(out->v_re, h->d_vre, bn3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (out->v_im) HIP3(h, hipMemcpyAsync(out->v_im, h->d_vim, bn3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  if (out->losses) HIP3(h, hipMemcpyAsync(out->losses, h->d_loss, h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->max_mismatch) HIP3(h, hipMemcpyAsync(out->max_mismatch, h->d_mm, h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out->iterations) HIP3(h, hipMemcpyAsync(out->iterations, h->d_it, h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  if (out->converged) HIP3(h, hipMemcpyAsync(out->converged, h->d_conv, h->B, hipMemcpyDeviceToHost, h->stream));
  HIP3(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs3_solve(gs3_handle* h, const double* P, const double* Q, const gs3_solution_view* out) {
  int rc = gs3_upload_injections(h, P, Q);
  if (rc) return rc;
  if ((rc = gs3_solve_device(h))) return rc;
  return out ? gs3_download_solution(h, out) : gs3_synchronize(h);
}

int gs3_synchronize(gs3_handle* h) {
  if (!h) return fail3(nullptr, GS_E_INVALID, "handle is NULL");
  HIP3(h, hipSetDevice(h->device));
  HIP3(h, hipStreamSynchronize(h->stream));
  return GS_OK;
}

int gs3_timing_read(gs3_handle* h, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return fail3(h, GS_E_INVALID, "bad arguments");
  HIP3(h, hipSetDevice(h->device));
  HIP3(h, hipStreamSynchronize(h->stream));
  *total_ms = 0.0; *launches = 0;
  for (size_t k = 0; k < h->ev_used; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev[k].first, h->ev[k].second) == hipSuccess) { *total_ms += ms; *launches += 1; }
  }
  h->ev_used = 0;
  return GS_OK;
}

int gs3_describe(const gs3_handle* h, char* buf, int32_t buflen) {
  if (!h || !buf || buflen <= 0) return fail3(nullptr, GS_E_INVALID, "bad arguments");
  snprintf(buf, buflen, "{\"kernel\": \"fbs3\", \"n\": %d, \"levels\": %d, \"max_level_width\": %d, \"batch\": %d, \"lds_messages\": %d, \"state_bytes\": %zu}",
           h->n, h->n_levels, h->max_width, h->B, h->lds_bytes, (size_t)h->B * h->rows * h->n * sizeof(double));
  return GS_OK;
}

}  // extern "C"
