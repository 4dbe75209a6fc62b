// gridstep3.hip -- three-phase unbalanced radial load flow (forward/backward sweep) for gfx950:
// BASELINE.json config 5 (8500-node feeder, batch ~1000).  Host side + kernels of the gs3_* ABI.
//
// NEW functionality: the reference advertises UnbalancedPowerFlow (README.md:187-197,
// API_REFERENCE.md:420) but ships no implementation, so nothing here restates reference code;
// the convergence test is the reference's power-mismatch criterion (environments/power_flow.py:
// 150-171) applied per phase, and in the balanced, uncoupled limit the answer reduces to the
// single-phase solution that IS pinned by the reference (tests/test_unbalanced.py).
//
// Two kernels.  gs3_k_resident (gridstep3_resident.h, the one a feeder of up to ~9 700 conductors gets): one workgroup
// keeps the whole instance in its CU's registers and LDS, the sweeps are prefix sums over depth-first orders of the tree.
// gs3_k_solve (below; larger feeders, GS3_NO_RESIDENT=1): level by level, the state streamed through HBM.
//
// Mapping of gs3_k_solve (differs from the single-phase kernels, which put one instance on each lane): one
// workgroup per instance, lanes over the PHASE CONDUCTORS ("slots") of a tree level.  A distribution
// feeder is mostly single-phase laterals (1.1 conductors per node on the 8500-node case), so storing
// three phases per node would move 2.7x the bytes that carry information.  Nodes are numbered
// breadth-first; inside a level the slots are ordered phase-major (all phase-a conductors in node
// order, then b, then c).  Consequences:
//   * a level is a contiguous slot range and every per-slot row is read and written coalesced,
//   * the same-phase children of a slot are a contiguous range of the next level,
//   * the other phases of the same node ("siblings", needed for the mutual impedances) are two
//     slot indices in the same level.
// Per-instance state is 6 doubles per slot (V, S_spec, D = voltage drop of the upstream line) and lives
// in HBM -- this configuration is HBM-streaming by construction.  What one
// level hands to the next (J going up, V going down) travels through LDS.
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
#if defined(GS_BUILD_EXPERIMENTS)        /* gs_internal.h: switches that exist to measure alternatives */
#define GS_EXPERIMENT_ENV(name) getenv(name)
#else
#define GS_EXPERIMENT_ENV(name) ((const char*)nullptr)
#endif

namespace {

struct Topo3 {
  int32_t n, ns, n_levels, cap;   // nodes, slots, levels, widest level in slots
  const int32_t* lvl_mutual; // same padding: 1 where a level has a node with more than one phase (its mutual impedances are read)
  const int32_t* lvl_ptr;    // [PAD + n_levels + 1 + PAD] slot ranges, padded with empty levels; level 0 = the three source conductors
  const int4* idx;           // [ns] {first child slot, child count | phase << 28, sibling slot A, sibling slot B}: the backward sweep's
                             //      (A / B = the lower / higher of the two other phases; -1 = absent)
  const int32_t* par;        // [ns] parent slot | phase << 30: all the forward sweep needs (one 4-byte load)
  const double2* z;          // [3][ns] row of the upstream line's Z for this conductor: (re, im) x {own, A, B}
  double vsr[3], vsi[3];     // source voltage
};

// rows of the per-instance state, [row][ns] of (re, im) / (P, Q) pairs -- one 16-byte load per lane and row;
// J only when the level messages cannot go through LDS
enum { C_V = 0, C_S, C_D, C_COUNT, C_J = C_COUNT, C_COUNT_NOLDS };
enum { Z_D = 0, Z_A, Z_B };

// row base in SGPRs + one 32-bit byte offset per thread
#define ST(row, s) (*(double2*)((char*)(S + (size_t)(row) * ns) + ((unsigned)(s) << 4)))
#define ZT(row, s) (*(const double2*)((const char*)(T.z + (size_t)(row) * ns) + ((unsigned)(s) << 4)))
#define IDX(s) (*(const int4*)((const char*)T.idx + ((unsigned)(s) << 4)))
#define PAR(s) (*(const int32_t*)((const char*)T.par + ((unsigned)(s) << 2)))
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

#define GS3_LVL_PAD 8             /* empty levels on either side of the level table */
#ifndef GS3_PREFETCH_LEVELS
#define GS3_PREFETCH_LEVELS 1     /* levels whose rows are in flight ahead of the one being processed */
#endif

extern __shared__ double gs3_msg[];   // [2 (level parity)][2 (re, im)][cap]: J of a level on the way up, V on the way down

// Level barrier.  With LDS messages nothing a level writes to HBM is read by another thread (a slot keeps
// its thread in both sweeps), so the barrier only has to order LDS and the loads prefetched for the next
// level stay in flight across it.
template <bool LDSMSG> __device__ __forceinline__ void level_barrier() {
  if (LDSMSG) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else {      // J rows in HBM: written by one wave, read by another -- device-scope release / acquire around the rendezvous
              // (a workgroup-scope __syncthreads() neither drains the stores nor refreshes the L1; see kernels_solve.hip)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

struct ZRow { double dr, di, ar, ai, br, bi; };               // a conductor's row of Z
struct UpIn { int4 ix; int sb; double vr, vi, p, q; ZRow z; };   // what a slot needs on the way up
struct DownIn { int4 ix; double dr, di, vr, vi, p, q; };      // ... and on the way down

// One sweep pair per iteration, the power mismatch evaluated on the way DOWN.
//   backward (deepest level first):   J_s = -conj(S_spec / V_s) + sum over same-phase children   (through LDS)
//                                     D_s = sum_phases Z[s][.] J[.]  -> HBM        (siblings' J through LDS)
//   forward  (root's children first): V_s = V_parent - D_s                          (parent through LDS)
// After a forward sweep V - V_parent = -Z J holds exactly, so the current the new voltages imply on a line is
// -J and the current drawn at a node is the injection current conj(S_spec / V_old) the backward sweep used:
// the mismatch S_spec - V_new conj(I_old) needs neither Y = Z^-1 nor the children's line currents, and it is
// the number the NEXT backward sweep of the textbook loop would report -- one sweep earlier.  The flat start
// is never written to memory: with V = V_source everywhere the implied currents are zero and the first
// mismatch is |S_spec|.
// Every level step prefetches the rows of the next level before its barrier, and D of a level is finished
// one step late (its Z row is requested when the level is processed and consumed after the barrier that
// publishes the siblings' J), so a step costs LDS latency plus arithmetic rather than round trips to HBM.
// The steady-state level step issues the same memory instructions whatever the level looks like: rows are
// requested unconditionally at a clamped slot (a lane beyond the end of its level re-reads the level's last
// slot), the level table is padded with empty levels at both ends, passes beyond the first (levels wider
// than the workgroup) run in their own loop, and the first iteration (flat start, nothing to read for V) is
// its own instantiation.  Only then can the compiler wait for exactly the rows it needs (s_waitcnt vmcnt(N)
// with the younger prefetches still in flight) instead of draining the queue.
struct Sweep3 {
  const Topo3& T;
  double2* __restrict__ S;
  int ns, cap, L, nth, nw, wave, lane;   // nw wavefronts; wave is wave-uniform (an SGPR)
  const GS3_CONST int32_t* lvl;      // lvl[-PAD .. L + PAD]
  const GS3_CONST int32_t* lmu;      // lvl_mutual, same indexing

  __device__ __forceinline__ double src_r(int ph) const { return ph == 0 ? T.vsr[0] : (ph == 1 ? T.vsr[1] : T.vsr[2]); }
  __device__ __forceinline__ double src_i(int ph) const { return ph == 0 ? T.vsi[0] : (ph == 1 ? T.vsi[1] : T.vsi[2]); }
  static __device__ __forceinline__ int clamp_to(int s, int end) { return min(s, max(end - 1, 0)); }
  // Which slots of level l a wavefront takes rotates with l: the widths are rarely a multiple of the workgroup, the
  // waves that find no slot in a level skip its arithmetic, and the rotation spreads that saving over all four SIMDs.
  // A slot still keeps its thread in both sweeps (the mapping depends on l only).
  __device__ __forceinline__ int wlo(int l) const { return ((wave + l + 64) % nw) << 6; }     // l >= -GS3_LVL_PAD
  __device__ __forceinline__ int off(int l) const { return wlo(l) + lane; }

  template <bool FIRST> __device__ __forceinline__ UpIn load_up(int s, bool mutual = true) const {
    UpIn u;
    u.ix = IDX(s); u.sb = u.ix.w;
    const int ph = (unsigned)u.ix.y >> 28;
    if (FIRST) { u.vr = src_r(ph); u.vi = src_i(ph); }
    else { const double2 v = ST(C_V, s); u.vr = v.x; u.vi = v.y; }
    const double2 pq = ST(C_S, s); u.p = pq.x; u.q = pq.y;
    if (mutual) u.z = load_z(s);                       // level-uniform: single-phase levels read the self impedance only
    else { const double2 d = ZT(Z_D, s); u.z = ZRow{d.x, d.y, 0.0, 0.0, 0.0, 0.0}; }
    return u;
  }
  __device__ __forceinline__ ZRow load_z(int s) const {
    ZRow z;
    const double2 d = ZT(Z_D, s), a = ZT(Z_A, s), b = ZT(Z_B, s);
    z.dr = d.x; z.di = d.y; z.ar = a.x; z.ai = a.y; z.br = b.x; z.bi = b.y;
    return z;
  }
  template <bool FIRST> __device__ __forceinline__ DownIn load_down(int s) const {
    DownIn d;
    d.ix = make_int4(PAR(s), 0, 0, 0);
    const int ph = (unsigned)d.ix.x >> 30;
    const double2 dd = ST(C_D, s); d.dr = dd.x; d.di = dd.y;
    if (FIRST) { d.vr = src_r(ph); d.vi = src_i(ph); }
    else { const double2 v = ST(C_V, s); d.vr = v.x; d.vi = v.y; }
    const double2 pq = ST(C_S, s); d.p = pq.x; d.q = pq.y;
    return d;
  }
  // J of slot `s` of the level whose messages sit in `buf` and whose first slot is `base`
  template <bool LDSMSG> __device__ __forceinline__ void msg_j(const double* buf, int base, int s, double& jr, double& ji) const {
    if (LDSMSG) { jr = buf[s - base]; ji = buf[cap + s - base]; }
    else { const double2 j = ST(C_J, s); jr = j.x; ji = j.y; }
  }
  // D = Z-row times the currents of the node's conductors, for slot s of a level whose J are published.
  // An absent sibling reads the slot's own J against a zero impedance.
  template <bool LDSMSG> __device__ __forceinline__ void drop_of(const double* buf, int base, int s, const ZRow& z, int sa, int sb,
                                                                 double& ar, double& ai) const {
    double jr, ji, xr, xi, yr, yi;
    msg_j<LDSMSG>(buf, base, s, jr, ji);
    msg_j<LDSMSG>(buf, base, sa >= 0 ? sa : s, xr, xi);
    msg_j<LDSMSG>(buf, base, sb >= 0 ? sb : s, yr, yi);
    ar = z.dr * jr - z.di * ji; ai = z.dr * ji + z.di * jr;
    ar += z.ar * xr - z.ai * xi; ai += z.ar * xi + z.ai * xr;
    ar += z.br * yr - z.bi * yi; ai += z.br * yi + z.bi * yr;
  }
  // own injection current + same-phase children -> J of a slot (children in `dn`, whose level starts at s1)
  template <bool LDSMSG, bool FIRST>
  __device__ __forceinline__ void line_current(const UpIn& u, const double* dn, int s1, bool at_source, double& lmax, double& psrc,
                                               double& jr, double& ji) const {
    const int ph = (unsigned)u.ix.y >> 28, cc = u.ix.y & 0xffff;
    if (FIRST) {
      const double dP = fabs(u.p), dQ = fabs(u.q);
      lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
    }
    const double rd = 1.0 / (u.vr * u.vr + u.vi * u.vi);
    jr = -(u.p * u.vr + u.q * u.vi) * rd; ji = -(u.p * u.vi - u.q * u.vr) * rd;
    for (int ch = u.ix.x; ch < u.ix.x + cc; ++ch) { double cr, ci; msg_j<LDSMSG>(dn, s1, ch, cr, ci); jr += cr; ji += ci; }
    if (at_source) psrc += src_r(ph) * jr + src_i(ph) * ji;     // the source's share of sum P_calc
  }

  // ---- backward sweep: J up through the levels, D = Z J to HBM; returns max |S_spec| (FIRST) and the source's P
  template <bool LDSMSG, int PF, bool FIRST> __device__ __forceinline__ void backward(double& lmax, double& psrc) const {
    UpIn uq[PF];             // rows of the next PF levels, requested ahead (uq[0] = the level processed next)
    ZRow zc = {};            // Z row of this thread's first slot in the level processed last (l + 1)
    int zsa = -1, zsb = -1;
#pragma unroll
    for (int k = 0; k < PF; ++k) uq[k] = load_up<FIRST>(clamp_to(lvl[L - 1 - k] + off(L - 1 - k), lvl[L - k]), lmu[L - 1 - k] != 0);
    int mu = lmu[L - 1 - PF];          // whether the level prefetched next carries mutual terms (read one step ahead, like w)
    int w[PF + 3];           // w[j] = lvl[l - PF + j]: the prefetched level .. the level being finished
#pragma unroll
    for (int j = 0; j < PF + 3; ++j) w[j] = lvl[L - 1 - PF + j];
    for (int l = L - 1; l >= 1; --l) {
      const int s0 = w[PF], s1 = w[PF + 1], e1 = w[PF + 2];
      const int wnext = lvl[l - PF - 1], munext = lmu[l - PF - 1];
      double* up = gs3_msg + (size_t)(l & 1) * 2 * cap;
      const double* dn = gs3_msg + (size_t)((l + 1) & 1) * 2 * cap;
      const UpIn u = uq[0];
#pragma unroll
      for (int k = 0; k + 1 < PF; ++k) uq[k] = uq[k + 1];
      uq[PF - 1] = load_up<FIRST>(clamp_to(w[0] + off(l - PF), w[1]), mu != 0);
      {                      // finish level l + 1: its J are all in `dn` now
        const int s = s1 + off(l + 1);
        double ar, ai;
        if (s1 + wlo(l + 1) < e1) {
          drop_of<LDSMSG>(dn, s1, min(s, e1 - 1), zc, zsa, zsb, ar, ai);
          if (s < e1) ST(C_D, s) = make_double2(ar, ai);
        }
        if (e1 - s1 > nth)
          for (int s2 = s + nth; s2 < e1; s2 += nth) {
            const int4 ix = IDX(s2); const int sb = ix.w;
            drop_of<LDSMSG>(dn, s1, s2, load_z(s2), ix.z, sb, ar, ai);
            ST(C_D, s2) = make_double2(ar, ai);
          }
      }
      {
        const int s = s0 + off(l);
        double jr, ji;
        zsa = -1; zsb = -1;
        if (s0 + wlo(l) < s1) {
          line_current<LDSMSG, FIRST>(u, dn, s1, l == 1 && s < s1, lmax, psrc, jr, ji);
          if (s < s1) {
            if (LDSMSG) { up[s - s0] = jr; up[cap + s - s0] = ji; }
            else ST(C_J, s) = make_double2(jr, ji);
            zsa = u.ix.z; zsb = u.sb;
          }
        }
        if (s1 - s0 > nth)
          for (int s2 = s + nth; s2 < s1; s2 += nth) {
            line_current<LDSMSG, FIRST>(load_up<FIRST>(s2), dn, s1, l == 1, lmax, psrc, jr, ji);
            if (LDSMSG) { up[s2 - s0] = jr; up[cap + s2 - s0] = ji; }
            else ST(C_J, s2) = make_double2(jr, ji);
          }
      }
      zc = u.z;
#pragma unroll
      for (int j = PF + 2; j > 0; --j) w[j] = w[j - 1];
      w[0] = wnext; mu = munext;
      level_barrier<LDSMSG>();
    }
    {                        // finish level 1
      const int s1 = lvl[1], e1 = lvl[2];
      const double* dn = gs3_msg + (size_t)2 * cap;
      double ar, ai;
      for (int s = s1 + off(1); s < e1; s += nth) {
        if (s == s1 + off(1)) drop_of<LDSMSG>(dn, s1, s, zc, zsa, zsb, ar, ai);
        else { const int4 ix = IDX(s); drop_of<LDSMSG>(dn, s1, s, load_z(s), ix.z, ix.w, ar, ai); }
        ST(C_D, s) = make_double2(ar, ai);
      }
      level_barrier<LDSMSG>();     // the forward sweep reuses level 1's buffer
    }
  }

  template <bool LDSMSG>
  __device__ __forceinline__ void new_voltage(const DownIn& d, const double* upr, int p0, int l, bool count, double& lmax, double& psum,
                                              double& wr, double& wi) const {
    const int ph = (unsigned)d.ix.x >> 30, ps = d.ix.x & 0x3fffffff;
    double pr, pi;
    if (l == 1) { pr = src_r(ph); pi = src_i(ph); }
    else if (LDSMSG) { pr = upr[ps - p0]; pi = upr[cap + ps - p0]; }
    else { const double2 pv = ST(C_V, ps); pr = pv.x; pi = pv.y; }
    wr = pr - d.dr; wi = pi - d.di;
    const double rd = 1.0 / (d.vr * d.vr + d.vi * d.vi);
    const double ior = (d.p * d.vr + d.q * d.vi) * rd, ioi = (d.p * d.vi - d.q * d.vr) * rd;
    const double pc = wr * ior + wi * ioi, qc = wi * ior - wr * ioi;
    const double dP = fabs(d.p - pc), dQ = fabs(d.q - qc);
    if (count) {
      lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
      psum += pc;
    }
  }

  // ---- forward sweep: V down through the levels, with the mismatch / sum of P_calc at the new voltages
  template <bool LDSMSG, int PF, bool FIRST> __device__ __forceinline__ void forward(double& lmax, double& psum) const {
    DownIn dq[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) dq[k] = load_down<FIRST>(clamp_to(lvl[1 + k] + off(1 + k), lvl[2 + k]));
    int v[PF + 3];           // v[j] = lvl[l - 1 + j]: the parent level .. the end of the prefetched level
#pragma unroll
    for (int j = 0; j < PF + 3; ++j) v[j] = lvl[j];
    for (int l = 1; l < L; ++l) {
      const int p0 = v[0], s0 = v[1], s1 = v[2];
      const int vnext = lvl[l + PF + 2];
      double* dnw = gs3_msg + (size_t)(l & 1) * 2 * cap;
      const double* upr = gs3_msg + (size_t)((l - 1) & 1) * 2 * cap;
      const DownIn d = dq[0];
#pragma unroll
      for (int k = 0; k + 1 < PF; ++k) dq[k] = dq[k + 1];
      dq[PF - 1] = load_down<FIRST>(clamp_to(v[PF + 1] + off(l + PF), v[PF + 2]));
      const int s = s0 + off(l);
      double wr, wi;
      // a lane beyond the level holds a clamped copy of the last slot: valid arithmetic, nothing counted or stored
      if (s0 + wlo(l) < s1) {
        new_voltage<LDSMSG>(d, upr, p0, l, s < s1, lmax, psum, wr, wi);
        if (s < s1) {
          ST(C_V, s) = make_double2(wr, wi);
          if (LDSMSG) { dnw[s - s0] = wr; dnw[cap + s - s0] = wi; }
        }
      }
      if (s1 - s0 > nth)
        for (int s2 = s + nth; s2 < s1; s2 += nth) {
          new_voltage<LDSMSG>(load_down<FIRST>(s2), upr, p0, l, true, lmax, psum, wr, wi);
          ST(C_V, s2) = make_double2(wr, wi);
          if (LDSMSG) { dnw[s2 - s0] = wr; dnw[cap + s2 - s0] = wi; }
        }
#pragma unroll
      for (int j = 0; j < PF + 2; ++j) v[j] = v[j + 1];
      v[PF + 2] = vnext;
      level_barrier<LDSMSG>();
    }
  }
};

template <bool LDSMSG, int PF>
__device__ __forceinline__ void gs3_solve_body(const Topo3& T, double2* __restrict__ S, double tol, int max_it,
                                               double* sh, double& losses, double& mm, int& it_out, int& conv_out) {
  const int ns = T.ns, tid = threadIdx.x, nth = blockDim.x;
  const Sweep3 sw = {T, S, ns, T.cap, T.n_levels, nth, nth >> 6, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63,
                     (const GS3_CONST int32_t*)T.lvl_ptr + GS3_LVL_PAD, (const GS3_CONST int32_t*)T.lvl_mutual + GS3_LVL_PAD};
  int iters = max_it, conv = 0;
  mm = INFINITY; losses = 0.0;
  if (tid < 3) ST(C_V, tid) = make_double2(T.vsr[tid], T.vsi[tid]);     // the source's three conductors
  for (int it = 0; it < max_it; ++it) {
    double lmax = 0.0, psrc = 0.0;
    if (it == 0) {
      sw.backward<LDSMSG, PF, true>(lmax, psrc);
      mm = block_max(lmax, sh);
      losses = 0.0;
      if (!(mm < INFINITY) || mm < tol) {     // no sweep will follow: the answer is the flat start itself
        for (int s = 3 + tid; s < ns; s += nth) {
          const int ph = (unsigned)IDX(s).y >> 28;
          ST(C_V, s) = make_double2(sw.src_r(ph), sw.src_i(ph));
        }
        iters = 1; conv = mm < tol;
        break;
      }
    } else {
      sw.backward<LDSMSG, PF, false>(lmax, psrc);
    }
    lmax = 0.0;
    double psum = psrc;
    if (it == 0) sw.forward<LDSMSG, PF, true>(lmax, psum);
    else sw.forward<LDSMSG, PF, false>(lmax, psum);
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

template <bool LDSMSG, int PF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
gs3_k_solve(Topo3 T, double2* __restrict__ state, int B, double tol, int max_it, double* __restrict__ out_loss,
            double* __restrict__ out_mm, int32_t* __restrict__ out_it, uint8_t* __restrict__ out_conv) {
  __shared__ double sh[8];
  const int b = blockIdx.x;
  double losses, mm; int it, conv;
  gs3_solve_body<LDSMSG, PF>(T, state + (size_t)b * (LDSMSG ? C_COUNT : C_COUNT_NOLDS) * T.ns, tol, max_it, sh, losses, mm, it, conv);
  if (threadIdx.x == 0) {
    out_loss[b] = losses;
    out_mm[b] = mm;
    out_it[b] = it;
    out_conv[b] = (uint8_t)conv;
  }
}

#include "gridstep3_resident.h"

typedef void (*res_fn)(Res3, double2*, int, double, int, double*, double*, int32_t*, uint8_t*);
res_fn resident_kernel(int K, int MK) {
  if (K == 3) return MK ? gs3_k_resident<3, 3> : gs3_k_resident<3, 0>;
  if (K == 9) return MK ? gs3_k_resident<9, 4> : gs3_k_resident<9, 0>;
  return MK ? gs3_k_resident<19, 4> : gs3_k_resident<19, 0>;
}

// P/Q [B][n][3] in caller node order -> the P, Q rows of the slots (src_of < 0: a padding entry of the resident layout)
extern "C" __global__ void __launch_bounds__(256)
gs3_k_scatter_in(int n, int ns, size_t stride, const int32_t* __restrict__ src_of, const double* __restrict__ P,
                 const double* __restrict__ Q, double2* __restrict__ state) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns || src_of[s] < 0) return;
  double2* S = state + (size_t)b * stride;
  const size_t src = (size_t)b * n * 3 + src_of[s];
  ST(C_S, s) = make_double2(P[src], Q ? Q[src] : 0.0);
}

// V rows of the slots -> [B][n][3] in caller node order (absent phases 0)
extern "C" __global__ void __launch_bounds__(256)
gs3_k_gather_out(int n, int ns, size_t stride, const int32_t* __restrict__ slot_of, const double2* __restrict__ state,
                 double* __restrict__ vre, double* __restrict__ vim) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 3 * n) return;
  double2* S = const_cast<double2*>(state) + (size_t)b * stride;
  const int s = slot_of[e];
  const size_t dst = (size_t)b * n * 3 + e;
  const double2 v = s >= 0 ? ST(C_V, s) : make_double2(0.0, 0.0);
  vre[dst] = v.x; vim[dst] = v.y;
}

thread_local std::string g3_error;

}  // namespace

struct gs3_handle {
  int device = 0, n = 0, ns = 0, B = 0, n_levels = 0, max_width = 0, max_it = 50, threads = 256, lds_bytes = 0, rows = C_COUNT, prefetch = GS3_PREFETCH_LEVELS;
  int ns_store = 0;          // entries per state row: ns, or K x threads in the resident layout
  size_t inst_stride = 0;    // double2 entries from one instance's rows to the next's
  int resident_k = 0;        // > 0: gs3_k_resident<resident_k, resident_mk> with `threads` threads solves the instance inside one CU
  int resident_mk = 0;
  double tol = 1e-6;
  hipStream_t stream = nullptr;
  Topo3 T{};
  Res3 R{};
  std::vector<void*> allocs;
  int32_t *d_src_of = nullptr, *d_slot_of = nullptr;
  double2* d_state = nullptr;
  double  *d_p = nullptr, *d_q = nullptr, *d_vre = nullptr, *d_vim = nullptr, *d_loss = nullptr, *d_mm = nullptr;
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

  // slots: the conductors of each level, phase-major
  std::vector<int32_t> slot_of((size_t)3 * n, -1), src_of, slvl{0};
  for (int l = 0; l < n_levels; ++l) {
    for (int ph = 0; ph < 3; ++ph)
      for (int tt = lvl_ptr[l]; tt < lvl_ptr[l + 1]; ++tt) {
        const int node = node_of[tt];
        if ((t->phases[node] >> ph) & 1) { slot_of[(size_t)node * 3 + ph] = (int)src_of.size(); src_of.push_back(node * 3 + ph); }
      }
    slvl.push_back((int)src_of.size());
  }
  const int ns = (int)src_of.size();

  gs3_handle* h = new gs3_handle();
  h->device = device; h->n = n; h->ns = ns; h->B = batch; h->tol = tolerance; h->max_it = max_iterations; h->n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) h->max_width = std::max(h->max_width, slvl[l + 1] - slvl[l]);
  std::vector<int4> idx(ns, make_int4(0, 0, 0, -1));
  std::vector<int32_t> par(ns, 0);
  std::vector<double2> z((size_t)3 * ns, make_double2(0.0, 0.0));
  for (int sl = 0; sl < ns; ++sl) {
    const int node = src_of[sl] / 3, ph = src_of[sl] % 3, mask = t->phases[node];
    const int oa = ph == 0 ? 1 : 0, ob = ph == 2 ? 1 : 2;      // the other two phases, ascending
    int cfirst = 0, ccount = 0;
    for (int c : kids[node]) {       // children in level order; those carrying `ph` are consecutive slots
      const int cs = slot_of[(size_t)c * 3 + ph];
      if (cs < 0) continue;
      if (!ccount) cfirst = cs;
      else if (cs != cfirst + ccount) { delete h; return fail3(nullptr, GS_E_TOPOLOGY, "children of node %d are not consecutive slots", node); }
      ++ccount;
    }
    if (ccount > 0xffff) { delete h; return fail3(nullptr, GS_E_TOPOLOGY, "node %d has more than 65535 children", node); }
    const int parent = node == t->source ? 0 : slot_of[(size_t)t->parent[node] * 3 + ph];
    idx[sl] = make_int4(cfirst, ccount | (ph << 28), slot_of[(size_t)node * 3 + oa], slot_of[(size_t)node * 3 + ob]);
    par[sl] = parent | (ph << 30);
    if (node == t->source) continue;
    double a[9], bb[9], ya[9], yb[9];
    for (int q = 0; q < 9; ++q) {
      const bool on = ((mask >> (q / 3)) & 1) && ((mask >> (q % 3)) & 1);
      a[q] = on ? t->z_re[(size_t)node * 9 + q] : 0.0; bb[q] = on ? t->z_im[(size_t)node * 9 + q] : 0.0;
    }
    if (ph == __builtin_ctz(mask)) {      // once per node: the impedance block must be invertible (a physical line)
      masked_inverse(a, bb, mask, ya, yb);
      for (int q = 0; q < 9; ++q)
        if (!std::isfinite(ya[q]) || !std::isfinite(yb[q])) { delete h; return fail3(nullptr, GS_E_TOPOLOGY, "line into node %d has a singular impedance block", node); }
    }
    z[(size_t)Z_D * ns + sl] = make_double2(a[3 * ph + ph], bb[3 * ph + ph]);
    z[(size_t)Z_A * ns + sl] = make_double2(a[3 * ph + oa], bb[3 * ph + oa]);
    z[(size_t)Z_B * ns + sl] = make_double2(a[3 * ph + ob], bb[3 * ph + ob]);
  }
  auto bail = [&](int rc) { gs3_destroy(h); return rc; };
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail3(nullptr, GS_E_HIP, "device / stream setup failed"));
  Topo3& T = h->T;
  T.n = n; T.ns = ns; T.n_levels = n_levels; T.cap = h->max_width;
  const double ang[3] = {0.0, -2.0 * M_PI / 3.0, 2.0 * M_PI / 3.0};
  for (int ph = 0; ph < 3; ++ph) { T.vsr[ph] = t->v_source[ph] * std::cos(ang[ph]); T.vsi[ph] = t->v_source[ph] * std::sin(ang[ph]); }
  int rc;
  std::vector<int32_t> slvl_pad(GS3_LVL_PAD, 0);
  slvl_pad.insert(slvl_pad.end(), slvl.begin(), slvl.end());
  slvl_pad.insert(slvl_pad.end(), GS3_LVL_PAD, ns);
  std::vector<int32_t> mutual_pad(slvl_pad.size(), 0);
  for (int l = 0; l < n_levels; ++l)
    for (int sl = slvl[l]; sl < slvl[l + 1]; ++sl)
      if (idx[sl].z >= 0 || idx[sl].w >= 0) { mutual_pad[GS3_LVL_PAD + l] = 1; break; }
  if ((rc = upload3(h, &T.lvl_mutual, mutual_pad)) || (rc = upload3(h, &T.lvl_ptr, slvl_pad)) || (rc = upload3(h, &T.idx, idx)) || (rc = upload3(h, &T.par, par)) || (rc = upload3(h, &T.z, z)))
    return bail(rc);
  // level messages through LDS when two parities of the widest level fit beside three other resident workgroups
  h->lds_bytes = 2 * 2 * h->max_width * (int)sizeof(double);
  if (h->lds_bytes > 38 * 1024 || GS_EXPERIMENT_ENV("GS3_NO_LDS")) h->lds_bytes = 0;
  h->rows = h->lds_bytes ? C_COUNT : C_COUNT_NOLDS;
  h->ns_store = ns;
  // Resident layout (gridstep3_resident.h) when the conductors fit in one CU: ns + 1 LDS entries beside 1 KB of scratch,
  // at most 19 positions per thread of at most 512.
  {
    int lds_max = 0;
    (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
    lds_max = std::max(lds_max, 64 * 1024);
    if (const char* e = GS_EXPERIMENT_ENV("GS3_RESIDENT_LDS")) lds_max = atoi(e);
    int K = 0;
    for (int k : {3, 9, 19}) if (!K && (ns + k - 1) / k <= 512) K = k;
    const int nthr = K ? std::max(64, ((ns + K - 1) / K + 63) / 64 * 64) : 0;
    const int need = (((K * nthr + 4) & ~3) + 72) * (int)sizeof(double2);
    if (K && ns <= GS3_RESIDENT_MAX_CONDUCTORS && need <= lds_max && !getenv("GS3_NO_RESIDENT")) {
      h->resident_k = K;
      h->threads = nthr;
      h->lds_bytes = need;
      h->rows = 2;
      h->ns_store = K * nthr;
    }
  }
  if (h->resident_k) {
    const int K = h->resident_k, nt = h->threads, npad = h->ns_store;
    // depth-first pre- and postorder of each phase's tree, the three blocks one after the other
    std::vector<int32_t> pre((size_t)3 * n, -1), post((size_t)3 * n, -1), size((size_t)3 * n, 0);
    std::vector<int32_t> pos_node, pos_ph;
    int off = 0;
    for (int ph = 0; ph < 3; ++ph) {
      int cpre = off, cpost = off;
      std::vector<std::pair<int, size_t>> stack{{t->source, 0}};
      pre[(size_t)t->source * 3 + ph] = cpre++; pos_node.push_back(t->source); pos_ph.push_back(ph);
      while (!stack.empty()) {
        auto& top = stack.back();
        const int node = top.first;
        bool went = false;
        while (top.second < kids[node].size()) {
          const int c = kids[node][top.second++];
          if (!((t->phases[c] >> ph) & 1)) continue;
          pre[(size_t)c * 3 + ph] = cpre++; pos_node.push_back(c); pos_ph.push_back(ph);
          stack.push_back({c, 0});
          went = true;
          break;
        }
        if (went) continue;
        post[(size_t)node * 3 + ph] = cpost++;
        size[(size_t)node * 3 + ph] = cpre - pre[(size_t)node * 3 + ph];
        stack.pop_back();
      }
      off = cpre;
    }
    if (off != ns) return bail(fail3(nullptr, GS_E_TOPOLOGY, "internal: %d conductors in depth-first order, %d expected", off, ns));
    auto mem = [&](int p) { return (p % K) * nt + p / K; };
    std::vector<int32_t> rpk(npad, 0), rsrc(npad, -1), rslot((size_t)3 * n, -1);
    for (int p = ns; p < npad; ++p) rpk[mem(p)] = (int32_t)((unsigned)p << 14);        // padding: size 0, postorder index = position
    std::vector<double2> rzd(npad, make_double2(0.0, 0.0)), rmz_a, rmz_b, rmzp((size_t)2 * npad, make_double2(0.0, 0.0));
    std::vector<int4> rmut;
    std::vector<int2> rmutp(npad, make_int2(0, 0));
    for (int p = 0; p < ns; ++p) {
      const int node = pos_node[p], ph = pos_ph[p], e = (int)((size_t)node * 3 + ph);
      const int oa = ph == 0 ? 1 : 0, ob = ph == 2 ? 1 : 2;
      rpk[mem(p)] = (int32_t)((unsigned)size[e] | (unsigned)post[e] << 14 | (node == t->source ? 1u << 28 : 0u) | (unsigned)ph << 29);
      rslot[e] = mem(p);
      if (node == t->source) continue;                           // nothing is drawn at the source: its S entries stay zero
      rsrc[mem(p)] = e;
      const int sl = slot_of[e];                                 // the level layout's row of Z for the same conductor
      rzd[mem(p)] = z[(size_t)Z_D * ns + sl];
      const int pa = pre[(size_t)node * 3 + oa], pb = pre[(size_t)node * 3 + ob];
      if (pa < 0 && pb < 0) continue;
      const int ea = pa >= 0 ? pa + size[(size_t)node * 3 + oa] : 0, eb = pb >= 0 ? pb + size[(size_t)node * 3 + ob] : 0;
      rmut.push_back(make_int4(std::max(pa, 0) | ea << 14, std::max(pb, 0) | eb << 14, post[e], 0));
      rmz_a.push_back(pa >= 0 ? z[(size_t)Z_A * ns + sl] : make_double2(0.0, 0.0));
      rmz_b.push_back(pb >= 0 ? z[(size_t)Z_B * ns + sl] : make_double2(0.0, 0.0));
      rmutp[mem(p)] = make_int2(rmut.back().x, rmut.back().y);
      rmzp[mem(p)] = rmz_a.back(); rmzp[(size_t)npad + mem(p)] = rmz_b.back();
    }
    Res3& R = h->R;
    R.ns = ns; R.npad = npad; R.K = K; R.M = (int)rmut.size();
    // the list dealt over the threads (3 or 4 entries each) where it is short; else every position computes its own mutual term
    h->resident_mk = R.M <= (K == 3 ? 3 : 4) * nt && !getenv("GS3_DENSE_MUTUAL") ? (K == 3 ? 3 : 4) : 0;
    if (!h->resident_mk && ((rc = upload3(h, &R.mutp, rmutp)) || (rc = upload3(h, &R.mzp, rmzp)))) return bail(rc);
    rmz_a.insert(rmz_a.end(), rmz_b.begin(), rmz_b.end());
    for (int ph = 0; ph < 3; ++ph) { R.vsr[ph] = T.vsr[ph]; R.vsi[ph] = T.vsi[ph]; R.off[ph] = pre[(size_t)t->source * 3 + ph]; }
    R.off[3] = ns;
    if (GS_EXPERIMENT_ENV("GS3_STAMPS")) { if ((rc = alloc3(h, &R.stamps, 16))) return bail(rc); (void)hipMemset(R.stamps, 0, 16 * sizeof(long long)); }
    if ((rc = upload3(h, &R.pk, rpk)) || (rc = upload3(h, &R.zd, rzd)) || (rc = upload3(h, &R.mut, rmut)) || (rc = upload3(h, &R.mz, rmz_a))) return bail(rc);
    src_of = rsrc; slot_of = rslot;
    if (hipFuncSetAttribute((const void*)resident_kernel(K, h->resident_mk), hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes) != hipSuccess)
      return bail(fail3(nullptr, GS_E_HIP, "cannot reserve %d bytes of LDS for the resident kernel", h->lds_bytes));
  }
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, src_of))) return bail(rc); h->d_src_of = const_cast<int32_t*>(q); }
  { const int32_t* q = nullptr; if ((rc = upload3(h, &q, slot_of))) return bail(rc); h->d_slot_of = const_cast<int32_t*>(q); }
  if (const char* e = GS_EXPERIMENT_ENV("GS3_PREFETCH")) h->prefetch = std::max(1, std::min(3, atoi(e)));
  if (const char* e = GS_EXPERIMENT_ENV("GS3_THREADS")) if (!h->resident_k) h->threads = std::max(64, std::min(256, atoi(e) / 64 * 64));
  const size_t bn3 = (size_t)batch * n * 3;
  // Instances of the resident layout start 256 x (37 mod 128) bytes apart: with a stride that is a multiple of 16 KB all
  // workgroups, which run in step, would ask the same few memory channels for the same row at the same time.
  h->inst_stride = (size_t)h->rows * h->ns_store;
  if (h->resident_k && !GS_EXPERIMENT_ENV("GS3_NO_STRIDE_PAD")) {
    size_t units = (h->inst_stride * sizeof(double2) + 255) / 256;
    while (units % 128 != 37) ++units;
    h->inst_stride = units * 256 / sizeof(double2);
  }
  h->R.stride = h->inst_stride;
  if ((rc = alloc3(h, &h->d_state, (size_t)batch * h->inst_stride)) || (rc = alloc3(h, &h->d_p, bn3)) || (rc = alloc3(h, &h->d_q, bn3)) ||
      (rc = alloc3(h, &h->d_vre, bn3)) || (rc = alloc3(h, &h->d_vim, bn3)) || (rc = alloc3(h, &h->d_loss, batch)) ||
      (rc = alloc3(h, &h->d_mm, batch)) || (rc = alloc3(h, &h->d_it, batch)) || (rc = alloc3(h, &h->d_conv, batch)))
    return bail(rc);
  if (hipMemset(h->d_state, 0, (size_t)batch * h->inst_stride * sizeof(double2)) != hipSuccess) return bail(fail3(nullptr, GS_E_HIP, "hipMemset failed"));
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
  dim3 grid((h->ns_store + 255) / 256, h->B);
  hipLaunchKernelGGL(gs3_k_scatter_in, grid, dim3(256), 0, h->stream, h->n, h->ns_store, h->inst_stride, h->d_src_of, h->d_p, Q ? h->d_q : (const double*)nullptr, h->d_state);
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
  typedef void (*solve_fn)(Topo3, double2*, int, double, int, double*, double*, int32_t*, uint8_t*);
  static const solve_fn with_lds[3] = {gs3_k_solve<true, 1>, gs3_k_solve<true, 2>, gs3_k_solve<true, 3>};
  if (h->resident_k) {
    const res_fn rf = resident_kernel(h->resident_k, h->resident_mk);
    hipLaunchKernelGGL(rf, dim3(h->B), dim3(h->threads), h->lds_bytes, h->stream, h->R, h->d_state, h->B, h->tol, h->max_it, h->d_loss, h->d_mm, h->d_it, h->d_conv);
    if (h->R.stamps) {       // development aid: where the cycles of an iteration go (100 MHz clock of workgroup 0)
      long long st[16];
      HIP3(h, hipStreamSynchronize(h->stream));
      HIP3(h, hipMemcpy(st, h->R.stamps, sizeof st, hipMemcpyDeviceToHost));
      fprintf(stderr, "gs3 resident stamps (cycles since the top of iteration 1):");
      for (int i = 1; i <= 9; ++i) fprintf(stderr, " %lld", st[i] - st[0]);
      fprintf(stderr, "; from the kernel's first instruction: loads landed %lld, first mismatch %lld, iteration 1 starts %lld, loop left %lld, end %lld\n",
              st[11] - st[10], st[12] - st[10], st[0] - st[10], st[13] - st[10], st[14] - st[10]);
    }
  } else {
    const solve_fn fn = h->lds_bytes ? with_lds[h->prefetch - 1] : gs3_k_solve<false, 1>;
    hipLaunchKernelGGL(fn, dim3(h->B), dim3(h->threads), h->lds_bytes, h->stream, h->T, h->d_state, h->B, h->tol, h->max_it, h->d_loss, h->d_mm, h->d_it, h->d_conv);
  }
  HIP3(h, hipGetLastError());
  HIP3(h, hipEventRecord(e.second, h->stream));
  return GS_OK;
}

int gs3_download_solution(gs3_handle* h, const gs3_solution_view* out) {
  if (!h || !out) return fail3(h, GS_E_INVALID, "handle / view is NULL");
  HIP3(h, hipSetDevice(h->device));
  const size_t bn3 = (size_t)h->B * h->n * 3;
  if (out->v_re || out->v_im) {
    dim3 grid((3 * h->n + 255) / 256, h->B);
    hipLaunchKernelGGL(gs3_k_gather_out, grid, dim3(256), 0, h->stream, h->n, h->ns_store, h->inst_stride, h->d_slot_of, h->d_state, h->d_vre, h->d_vim);
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
  snprintf(buf, buflen, "{\"kernel\": \"%s\", \"n\": %d, \"conductors\": %d, \"levels\": %d, \"max_level_width\": %d, \"batch\": %d, \"lds_messages\": %d, \"state_bytes\": %zu, "
           "\"threads\": %d, \"positions_per_thread\": %d, \"mutual_entries\": %d, \"mutual_per_thread\": %d, \"lds_bytes\": %d}",
           h->resident_k ? "fbs3_resident" : "fbs3", h->n, h->ns, h->n_levels, h->max_width, h->B, h->resident_k ? 0 : h->lds_bytes,
           (size_t)h->B * h->inst_stride * sizeof(double2), h->threads, h->resident_k, h->R.M, h->resident_mk, h->lds_bytes);
  return GS_OK;
}

}  // extern "C"
