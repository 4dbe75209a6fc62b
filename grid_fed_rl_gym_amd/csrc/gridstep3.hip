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
// Per-instance state (24 doubles per node: V, S_spec, J, K; 1.6 MB at 8500 nodes) lives in HBM --
// this configuration is HBM-streaming by construction.
//
//   backward (deepest level first):  K_t = Y_t (V_t - V_parent)              current implied by the voltages
//                                    S_calc = V_t conj(K_t - sum_children K)  -> mismatch, losses
//                                    J_t = -conj(S_spec / V_t) + sum_children J
//   forward  (root's children first): V_t = V_parent - Z_t J_t
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gridstep.h"

namespace {

struct Topo3 {
  int32_t n, n_levels;
  const int32_t* lvl_ptr;    // [n_levels + 1]; level 0 = the source alone
  const int32_t* par;        // [n] position of the parent (level order)
  const int32_t* cfirst;     // [n] first child position
  const int32_t* ccount;     // [n]
  const int32_t* mask;       // [n] phase mask of the node
  const double* zr;          // [9][n] series impedance of the upstream line (rows/cols of absent phases zeroed)
  const double* zi;
  const double* yr;          // [9][n] its inverse on the present phases
  const double* yi;
  double vsr[3], vsi[3];     // source voltage
};

enum { C_VR = 0, C_VI = 3, C_P = 6, C_Q = 9, C_JR = 12, C_JI = 15, C_KR = 18, C_KI = 21, C_COUNT = 24 };

#define ST(comp, t) S[(size_t)(comp) * n + (t)]

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

extern "C" __global__ void __launch_bounds__(256)
gs3_k_solve(Topo3 T, double* __restrict__ state, int B, double tol, int max_it, double* __restrict__ out_loss,
            double* __restrict__ out_mm, int32_t* __restrict__ out_it, uint8_t* __restrict__ out_conv) {
  __shared__ double sh[8];
  const int n = T.n;
  const int b = blockIdx.x;
  double* S = state + (size_t)b * C_COUNT * n;
  // flat start: every present phase at the source voltage
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int m = T.mask[t];
    for (int ph = 0; ph < 3; ++ph) {
      const bool on = (m >> ph) & 1;
      ST(C_VR + ph, t) = on ? T.vsr[ph] : 0.0;
      ST(C_VI + ph, t) = on ? T.vsi[ph] : 0.0;
    }
  }
  __syncthreads();
  int it = 0, conv = 0;
  double mm = INFINITY, losses = 0.0;
  for (it = 0; it < max_it; ++it) {
    double lmax = 0.0, psum = 0.0;
    for (int l = T.n_levels - 1; l >= 1; --l) {
      const int t1 = T.lvl_ptr[l + 1];
      for (int t = T.lvl_ptr[l] + threadIdx.x; t < t1; t += blockDim.x) {
        const int m = T.mask[t], pt = T.par[t];
        double vr[3], vi[3], dr[3], di[3], pr[3], pi[3];
        for (int ph = 0; ph < 3; ++ph) {
          vr[ph] = ST(C_VR + ph, t); vi[ph] = ST(C_VI + ph, t);
          pr[ph] = ST(C_VR + ph, pt); pi[ph] = ST(C_VI + ph, pt);
          dr[ph] = vr[ph] - pr[ph]; di[ph] = vi[ph] - pi[ph];
        }
        double kr[3], ki[3];
        for (int r = 0; r < 3; ++r) {
          double ar = 0.0, ai = 0.0;
          for (int cc = 0; cc < 3; ++cc) {
            const double yr = T.yr[(size_t)(3 * r + cc) * n + t], yi = T.yi[(size_t)(3 * r + cc) * n + t];
            ar += yr * dr[cc] - yi * di[cc];
            ai += yr * di[cc] + yi * dr[cc];
          }
          kr[r] = ar; ki[r] = ai;
        }
        double sjr[3] = {0, 0, 0}, sji[3] = {0, 0, 0}, skr[3] = {0, 0, 0}, ski[3] = {0, 0, 0};
        const int c0 = T.cfirst[t], c1 = c0 + T.ccount[t];
        for (int ch = c0; ch < c1; ++ch)
          for (int ph = 0; ph < 3; ++ph) {
            sjr[ph] += ST(C_JR + ph, ch); sji[ph] += ST(C_JI + ph, ch);
            skr[ph] += ST(C_KR + ph, ch); ski[ph] += ST(C_KI + ph, ch);
          }
        for (int ph = 0; ph < 3; ++ph) {
          double jr = sjr[ph], ji = sji[ph];
          if ((m >> ph) & 1) {
            const double p = ST(C_P + ph, t), q = ST(C_Q + ph, t);
            const double icr = kr[ph] - skr[ph], ici = ki[ph] - ski[ph];
            const double pc = vr[ph] * icr + vi[ph] * ici, qc = vi[ph] * icr - vr[ph] * ici;
            const double dP = fabs(p - pc), dQ = fabs(q - qc);
            lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
            psum += pc;
            if (pt == 0) psum -= pr[ph] * kr[ph] + pi[ph] * ki[ph];     // the source's share
            const double rd = 1.0 / (vr[ph] * vr[ph] + vi[ph] * vi[ph]);
            jr -= (p * vr[ph] + q * vi[ph]) * rd;
            ji += (q * vr[ph] - p * vi[ph]) * rd;
          }
          ST(C_JR + ph, t) = jr; ST(C_JI + ph, t) = ji;
          ST(C_KR + ph, t) = kr[ph]; ST(C_KI + ph, t) = ki[ph];
        }
      }
      __syncthreads();
    }
    mm = block_max(lmax, sh);
    losses = block_sum(psum, sh);
    if (!(mm < INFINITY)) break;
    if (mm < tol) { conv = 1; break; }
    for (int l = 1; l < T.n_levels; ++l) {
      const int t1 = T.lvl_ptr[l + 1];
      for (int t = T.lvl_ptr[l] + threadIdx.x; t < t1; t += blockDim.x) {
        const int m = T.mask[t], pt = T.par[t];
        double jr[3], ji[3];
        for (int ph = 0; ph < 3; ++ph) { jr[ph] = ST(C_JR + ph, t); ji[ph] = ST(C_JI + ph, t); }
        for (int r = 0; r < 3; ++r) {
          double ar = 0.0, ai = 0.0;
          for (int cc = 0; cc < 3; ++cc) {
            const double zr = T.zr[(size_t)(3 * r + cc) * n + t], zi = T.zi[(size_t)(3 * r + cc) * n + t];
            ar += zr * jr[cc] - zi * ji[cc];
            ai += zr * ji[cc] + zi * jr[cc];
          }
          const bool on = (m >> r) & 1;
          ST(C_VR + r, t) = on ? ST(C_VR + r, pt) - ar : 0.0;
          ST(C_VI + r, t) = on ? ST(C_VI + r, pt) - ai : 0.0;
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    out_loss[b] = losses;
    out_mm[b] = mm;
    out_it[b] = (it < max_it) ? it + 1 : max_it;
    out_conv[b] = (uint8_t)conv;
  }
}

// P/Q [B][n][3] in caller node order -> state rows in level order
extern "C" __global__ void __launch_bounds__(256)
gs3_k_scatter_in(int n, const int32_t* __restrict__ node_of, const double* __restrict__ P, const double* __restrict__ Q,
                 double* __restrict__ state) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double* S = state + (size_t)b * C_COUNT * n;
  const size_t src = ((size_t)b * n + node_of[t]) * 3;
  for (int ph = 0; ph < 3; ++ph) { ST(C_P + ph, t) = P[src + ph]; ST(C_Q + ph, t) = Q ? Q[src + ph] : 0.0; }
}

// V in level order -> [B][n][3] in caller node order
extern "C" __global__ void __launch_bounds__(256)
gs3_k_gather_out(int n, const int32_t* __restrict__ pos_of, const double* __restrict__ state, double* __restrict__ vre,
                 double* __restrict__ vim) {
  const int b = blockIdx.y;
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n) return;
  const double* S = state + (size_t)b * C_COUNT * n;
  const int t = pos_of[node];
  const size_t dst = ((size_t)b * n + node) * 3;
  for (int ph = 0; ph < 3; ++ph) { vre[dst + ph] = ST(C_VR + ph, t); vim[dst + ph] = ST(C_VI + ph, t); }
}

thread_local std::string g3_error;

}  // namespace

struct gs3_handle {
  int device = 0, n = 0, B = 0, n_levels = 0, max_width = 0, max_it = 50;
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
  std::vector<double> zr((size_t)9 * n, 0.0), zi((size_t)9 * n, 0.0), yr((size_t)9 * n, 0.0), yi((size_t)9 * n, 0.0);
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
    masked_inverse(a, bb, mask[tt], ya, yb);
    for (int q = 0; q < 9; ++q) {
      zr[(size_t)q * n + tt] = a[q]; zi[(size_t)q * n + tt] = bb[q];
      yr[(size_t)q * n + tt] = ya[q]; yi[(size_t)q * n + tt] = yb[q];
      if (!std::isfinite(ya[q]) || !std::isfinite(yb[q])) { delete h; return fail3(nullptr, GS_E_TOPOLOGY, "line into node %d has a singular impedance block", node); }
    }
  }
  auto bail = [&](int rc) { gs3_destroy(h); return rc; };
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail3(nullptr, GS_E_HIP, "device / stream setup failed"));
  Topo3& T = h->T;
  T.n = n; T.n_levels = n_levels;
  const double ang[3] = {0.0, -2.0 * M_PI / 3.0, 2.0 * M_PI / 3.0};
  for (int ph = 0; ph < 3; ++ph) { T.vsr[ph] = t->v_source[ph] * std::cos(ang[ph]); T.vsi[ph] = t->v_source[ph] * std::sin(ang[ph]); }
  int rc;
  if ((rc = upload3(h, &T.lvl_ptr, lvl_ptr)) || (rc = upload3(h, &T.par, par)) || (rc = upload3(h, &T.cfirst, cfirst)) ||
      (rc = upload3(h, &T.ccount, ccount)) || (rc = upload3(h, &T.mask, mask)) || (rc = upload3(h, &T.zr, zr)) ||
      (rc = upload3(h, &T.zi, zi)) || (rc = upload3(h, &T.yr, yr)) || (rc = upload3(h, &T.yi, yi)))
    return bail(rc);
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, node_of))) return bail(rc); h->d_node_of = const_cast<int32_t*>(q); }
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, pos_of))) return bail(rc); h->d_pos_of = const_cast<int32_t*>(q); }
  const size_t bn3 = (size_t)batch * n * 3;
  if ((rc = alloc3(h, &h->d_state, (size_t)batch * C_COUNT * n)) || (rc = alloc3(h, &h->d_p, bn3)) || (rc = alloc3(h, &h->d_q, bn3)) ||
      (rc = alloc3(h, &h->d_vre, bn3)) || (rc = alloc3(h, &h->d_vim, bn3)) || (rc = alloc3(h, &h->d_loss, batch)) ||
      (rc = alloc3(h, &h->d_mm, batch)) || (rc = alloc3(h, &h->d_it, batch)) || (rc = alloc3(h, &h->d_conv, batch)))
    return bail(rc);
  if (hipMemset(h->d_state, 0, (size_t)batch * C_COUNT * n * sizeof(double)) != hipSuccess) return bail(fail3(nullptr, GS_E_HIP, "hipMemset failed"));
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
  hipLaunchKernelGGL(gs3_k_scatter_in, grid, dim3(256), 0, h->stream, h->n, h->d_node_of, h->d_p, Q ? h->d_q : (const double*)nullptr, h->d_state);
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
  hipLaunchKernelGGL(gs3_k_solve, dim3(h->B), dim3(256), 0, h->stream, h->T, h->d_state, h->B, h->tol, h->max_it, h->d_loss, h->d_mm, h->d_it, h->d_conv);
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
    hipLaunchKernelGGL(gs3_k_gather_out, grid, dim3(256), 0, h->stream, h->n, h->d_pos_of, h->d_state, h->d_vre, h->d_vim);
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
  snprintf(buf, buflen, "{\"kernel\": \"fbs3\", \"n\": %d, \"levels\": %d, \"max_level_width\": %d, \"batch\": %d, \"state_bytes\": %zu}",
           h->n, h->n_levels, h->max_width, h->B, (size_t)h->B * C_COUNT * h->n * sizeof(double));
  return GS_OK;
}

}  // extern "C"
