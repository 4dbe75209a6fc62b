// kernels_solve.hip -- batched AC load-flow and fused env.step() kernels for gfx950 (MI355X, wave64).
//
// Mapping: lane = feeder instance, workgroup = W wavefronts sharing one
// 64-instance group, waves split buses / lines / forest levels.  All per-instance data lives
// in the group's slab, row-major [row][64 lanes], so every load/store below is one fully
// coalesced 512-byte wave access; topology tables are wave-uniform and come through the
// scalar cache (constant address space -> s_load).  No lane ever diverges from its wave:
// per-instance convergence is handled with a `done` predicate, and a group leaves the Newton
// loop when all of its 64 instances are done.
//
// Reference arithmetic restated (paths relative to /root/reference/grid_fed_rl/environments/):
//   mismatch            power_flow.py:150-171
//   Jacobian entries    power_flow.py:243-287   (J11 diagonal sign: as coded :248, or exact)
//   corrections         power_flow.py:297-327
//   line flows, losses  power_flow.py:329-358, 198-200
//   env step            grid_env.py:410-619 (see env_device.h for the pieces before the solve)
// The linear solve (power_flow.py:187, LAPACK dgesv on the dense Jacobian) is replaced by a
// 2x2-block elimination on the Jacobian's own sparsity: a level-scheduled forest sweep for
// radial feeders, a statically scheduled block LU (host-side minimum-degree symbolic
// factorisation) for meshed ones; a dense partially pivoted LU is kept for as-coded parity.
//
// Every solver exists twice: gs_k_<solver> (solve only, behind gs_solve) and gs_k_step_<solver>
// (the whole transition in one launch, behind gs_step): actions -> batteries / curtailment ->
// clock -> weather -> renewables -> loads -> injections -> load flow -> line flows -> losses ->
// grid state -> frequency -> reward -> flags.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]
#define ROW2(r) S.pair((size_t)(r) * GS_LANES)        /* (row r, row r + 1) of the lane as one double2; r even */
#include "env_device.h"

__device__ __forceinline__ double finite_or_inf(double v) { return (fabs(v) < INFINITY) ? v : INFINITY; }

// LDS scratch for cross-wave reductions.  red/flag are double-buffered by an iteration parity so
// that a fast wave can enter the next reduction before a slow one has finished reading the
// previous one; `post` carries the partial results of the epilogue.
struct GsShared {
  double red[2][GS_MAX_WAVES][GS_LANES];
  int flag[2][GS_MAX_WAVES][GS_LANES];
};

// Dynamic LDS (size chosen by the host, gs_dyn_lds_bytes()): during the forest sweeps it holds the
// child -> parent / parent -> child messages of two adjacent levels, [slot][6][64 lanes]; in the
// epilogue it is reused for the cross-wave partial results.
extern __shared__ __attribute__((aligned(16))) double gs_dyn[];
#define GS_MSG_DOUBLES 6
#define GS_EPI_DOUBLES (4 * GS_MAX_WAVES * GS_LANES)   /* post[4][W][64] doubles, then posti[4][W][64] ints */
#define GS_PACK_LDS_DOUBLES (GS_EPI_DOUBLES + 2 * GS_MAX_WAVES * GS_LANES)   /* the pack tiles start after post[4] and posti[4] (48 KB) */

// Barrier for waves that exchanged data through LDS only: wait for this wave's LDS traffic, then
// rendezvous.  Unlike __syncthreads() it does not drain outstanding global loads/stores, so
// prefetches issued before it stay in flight across it.
// Barrier that publishes ROWS (global memory) between the waves of a group.  __syncthreads() is a workgroup-scope
// release / acquire, for which the compiler emits no s_waitcnt vmcnt(0) (the vector memory operations of one compute
// unit reach its L1 in issue order); this one drains the wave's stores first all the same -- it costs nothing
// measurable, and it took one suspect off the list while the cause of the corrupted row sectors was being found
// (the store-data hazard of GsPairRef::put, gs_internal.h).
__device__ __forceinline__ void gs_rows_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifndef GS_EXP_PLAIN_SYNC
#define __syncthreads() gs_rows_barrier()
#endif
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Ctx {
  const GsTables& T;
  const GsRows& R;
  GsLaneRows S;
  GsShared& sh;
  int lane, wave, W;
  unsigned long long* stamps;
  unsigned long long tlast;
  int stamp_wave;
  double fsum;              // this lane's sum of |dP| + |dQ| over the wave's buses since the last convergence check (the sweeps' criterion)
};

// Diagnostic phase stamps (gs_debug_stamps): block 0 / wave 0 / lane 0 adds the cycles since its
// previous stamp to slot `k`.  Off (one scalar branch) unless the host armed a buffer; the values
// go to that buffer only and feed nothing.
enum { ST_PROLOGUE = 0, ST_INIT, ST_MISMATCH, ST_BOTTOM_UP, ST_FLAG, ST_TOP_DOWN, ST_FINAL_MISMATCH, ST_EPILOGUE,
       ST_PRO_SCALAR, ST_PRO_SPARE, ST_EPI_BUSES, ST_EPI_LINES, ST_EPI_REDUCE, ST_EPI_SCALARS, ST_COUNT };   // ST_PROLOGUE = injections, ST_EPILOGUE = observation pack
__device__ __forceinline__ void stamp(Ctx& c, int k) {
  if (c.stamps == nullptr) return;
  const unsigned long long now = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && c.wave == c.stamp_wave && c.lane == 0) c.stamps[k] += now - c.tlast;
  c.tlast = now;
}

__device__ __forceinline__ double wg_max(Ctx& c, int par, double v) {
  c.sh.red[par][c.wave][c.lane] = v;
  __syncthreads();
  double r = c.sh.red[par][0][c.lane];
  for (int w = 1; w < c.W; ++w) r = fmax(r, c.sh.red[par][w][c.lane]);
  return r;
}

// Maximum and, in wave order, sum over the waves of the group (one value each per lane = instance).  Uses both halves of
// `red`, so it ends with a second barrier: nobody starts the next reduction before everybody has read this one.
__device__ __forceinline__ double wg_max_sum(Ctx& c, double vmax, double vsum, double* sum_out) {
  c.sh.red[0][c.wave][c.lane] = vmax;
  c.sh.red[1][c.wave][c.lane] = vsum;
  __syncthreads();
  double r = c.sh.red[0][0][c.lane], s = c.sh.red[1][0][c.lane];
  for (int w = 1; w < c.W; ++w) { r = fmax(r, c.sh.red[0][w][c.lane]); s += c.sh.red[1][w][c.lane]; }
  __syncthreads();
  *sum_out = s;
  return r;
}

__device__ __forceinline__ int wg_or(Ctx& c, int par, int v) {
  c.sh.flag[par][c.wave][c.lane] = v;
  __syncthreads();
  int r = c.sh.flag[par][0][c.lane];
  for (int w = 1; w < c.W; ++w) r |= c.sh.flag[par][w][c.lane];
  return r;
}

// ---- flat start (power_flow.py:103, 128-136) ---------------------------------------------
__device__ __forceinline__ void flat_start(Ctx& c) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  for (int i = c.wave; i < T.n; i += c.W) {
    const double vm = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
    ROW2(R.VM + i) = make_double2(vm, 0.0);
    ROW2(R.E + i) = make_double2(vm, 0.0);          // vm * cos(0), vm * sin(0): exact
    ROW(R.RVM + i) = 1.0 / vm;
  }
}

// ---- polar -> rectangular ------------------------------------------------------------------
__device__ __forceinline__ void to_rect(Ctx& c) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  for (int i = c.wave; i < T.n; i += c.W) {
    const double2 v = ROW2(R.VM + i);
    double s, cs;
    sincos(v.y, &s, &cs);
    ROW2(R.E + i) = make_double2(v.x * cs, v.x * s);
  }
}

// ---- S = V conj(Y V) by CSR rows; dP, dQ; returns this wave's max |mismatch| (inf if non-finite)
// With a_ij = e_i e_j + f_i f_j = Vi Vj cos(th_i - th_j), b_ij = f_i e_j - e_i f_j = Vi Vj sin(..):
//   P_i = sum_j G_ij a_ij + B_ij b_ij,   Q_i = sum_j G_ij b_ij - B_ij a_ij.
// STORE: 0 = P_calc only (FBS: all it needs is the losses), 1 = P_calc, Q_calc (LDS forest solve forms
// the right-hand side itself), 2 = also the mismatch rows R0/R1 (the other linear solves)
template <int STORE>
__device__ __forceinline__ double mismatch_rows(Ctx& c) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double lmax = 0.0;
  const int k0 = cld(T.wb_ptr, c.wave), k1 = cld(T.wb_ptr, c.wave + 1);
  const GS_CONST GsBusRec* recs = (const GS_CONST GsBusRec*)T.wbus;
  // header + neighbour indices of the wave's next bus are fetched one bus ahead (scalar loads in
  // flight while the current bus computes); all vector loads of a bus are issued back to back
  int hb = 0, hf = 0, hc[GS_ELL_K] = {};
  if (k0 < k1) {
    hb = recs[k0].bus; hf = recs[k0].flags;
#pragma unroll
    for (int k = 0; k < GS_ELL_K; ++k) hc[k] = recs[k0].col[k];
  }
  double P = 0.0, Q = 0.0;
  for (int q = k0; q < k1; ++q) {
    const int i = hb, fl = hf;
    const double2 vi = ROW2(R.E + i);
    const double ei = vi.x, fi = vi.y;
    double ej[GS_ELL_K], fj[GS_ELL_K];
#pragma unroll
    for (int k = 0; k < GS_ELL_K; ++k) { const double2 vj = ROW2(R.E + hc[k]); ej[k] = vj.x; fj[k] = vj.y; }
    const double2 sp = ROW2(R.P + i);
    const double ps = sp.x, qs = sp.y;
    if (q + 1 < k1) {
      hb = recs[q + 1].bus; hf = recs[q + 1].flags;
#pragma unroll
      for (int k = 0; k < GS_ELL_K; ++k) hc[k] = recs[q + 1].col[k];
    }
    if (!(fl & 8)) { P = 0.0; Q = 0.0; }               // a continuation record keeps accumulating
#pragma unroll
    for (int k = 0; k < GS_ELL_K; ++k) {
      const double g = recs[q].G[k], b = recs[q].B[k];
      const double a = ei * ej[k] + fi * fj[k];
      const double bb = fi * ej[k] - ei * fj[k];
      P += g * a + b * bb;
      Q += g * bb - b * a;
    }
    if (fl & 4) continue;                              // the row continues in the next record
    if (STORE >= 1) ROW2(R.PC + i) = make_double2(P, Q);
    else ROW(R.PC + i) = P;
    const double dP = (fl & 1) ? (ps - P) : 0.0;
    const double dQ = (fl & 2) ? (qs - Q) : 0.0;
    if (STORE >= 2) ROW2(R.R0 + i) = make_double2(dP, dQ);
    lmax = fmax(lmax, fmax(finite_or_inf(fabs(dP)), finite_or_inf(fabs(dQ))));
    c.fsum += fabs(dP) + fabs(dQ);
  }
  return lmax;
}

// ---- Jacobian blocks --------------------------------------------------------------------------
struct Blk { double a00, a01, a10, a11; };

// diagonal block of bus i from values (power_flow.py:247-248, 259-260, 270-271, 283-284)
__device__ __forceinline__ Blk diag_from(const GsTables& T, int i, int exact, double vm, double P, double Q) {
  const double gd = cld(T.Gd, i), bd = cld(T.Bd, i);
  const int th = cld(T.th_free, i), vf = cld(T.vm_free, i);
  Blk d;
  const double vvb = vm * vm * bd;
  d.a00 = th ? (exact ? (-Q - vvb) : (-Q + vvb)) : 1.0;
  d.a01 = (th && vf) ? (P / vm + vm * gd) : 0.0;
  d.a10 = (th && vf) ? (P - vm * vm * gd) : 0.0;
  d.a11 = vf ? (Q / vm - vm * bd) : 1.0;
  return d;
}

__device__ __forceinline__ Blk diag_block(Ctx& c, int i, int exact) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  return diag_from(c.T, i, exact, ROW(R.VM + i), ROW(R.PC + i), ROW(R.QC + i));
}

// off-diagonal block, row bus i, column bus j, from values (power_flow.py:251, 263, 274, 287)
__device__ __forceinline__ Blk offdiag_from(const GsTables& T, int i, int j, double g, double b, double ei, double fi,
                                            double ej, double fj, double vmj) {
  const double a = ei * ej + fi * fj;
  const double bb = fi * ej - ei * fj;
  const double gs_bc = g * bb - b * a;     // Vi Vj (G sin - B cos)
  const double gc_bs = g * a + b * bb;     // Vi Vj (G cos + B sin)
  const int thi = cld(T.th_free, i), vfi = cld(T.vm_free, i);
  const int thj = cld(T.th_free, j), vfj = cld(T.vm_free, j);
  Blk u;
  u.a00 = (thi && thj) ? gs_bc : 0.0;
  u.a01 = (thi && vfj) ? gc_bs / vmj : 0.0;
  u.a10 = (vfi && thj) ? -gc_bs : 0.0;
  u.a11 = (vfi && vfj) ? gs_bc / vmj : 0.0;
  return u;
}

__device__ __forceinline__ Blk offdiag_block(Ctx& c, int i, int j, double g, double b) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  return offdiag_from(c.T, i, j, g, b, ROW(R.E + i), ROW(R.F + i), ROW(R.E + j), ROW(R.F + j), ROW(R.VM + j));
}

__device__ __forceinline__ Blk mul(const Blk& x, const Blk& y) {
  Blk z;
  z.a00 = x.a00 * y.a00 + x.a01 * y.a10;
  z.a01 = x.a00 * y.a01 + x.a01 * y.a11;
  z.a10 = x.a10 * y.a00 + x.a11 * y.a10;
  z.a11 = x.a10 * y.a01 + x.a11 * y.a11;
  return z;
}

// inverse of a 2x2 block; *sing is set when the determinant is exactly zero or non-finite
// (the reference's LinAlgError case, power_flow.py:188-190: only an exactly singular matrix raises)
__device__ __forceinline__ Blk inv2(const Blk& d, int* sing) {
  const double det = d.a00 * d.a11 - d.a01 * d.a10;
  if (!(det != 0.0) || !(fabs(det) < INFINITY)) *sing = 1;
  const double r = 1.0 / det;
  Blk z;
  z.a00 = d.a11 * r; z.a01 = -d.a01 * r; z.a10 = -d.a10 * r; z.a11 = d.a00 * r;
  return z;
}

__device__ __forceinline__ Blk load_blk(GsLaneRows S, int row) {
  const double2 lo = ROW2(row), hi = ROW2(row + 2);     // blocks start on even rows
  Blk b; b.a00 = lo.x; b.a01 = lo.y; b.a10 = hi.x; b.a11 = hi.y; return b;
}
__device__ __forceinline__ void store_blk(GsLaneRows S, int row, const Blk& b) {
  ROW2(row) = make_double2(b.a00, b.a01); ROW2(row + 2) = make_double2(b.a10, b.a11);
}

// ---- apply the Newton step to bus i (power_flow.py:315-327); keeps Vm >= 0 like the
// reference's abs/angle round trip does
__device__ __forceinline__ void apply_step(Ctx& c, int i, double alpha, bool upd) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  if (!upd) return;
  double vm = ROW(R.VM + i), va = ROW(R.VA + i);
  if (cld(T.th_free, i)) va += alpha * ROW(R.X0 + i);
  if (cld(T.vm_free, i)) vm += alpha * ROW(R.X1 + i);
  if (vm < 0.0) { vm = -vm; va += M_PI; }
  ROW(R.VM + i) = vm;
  ROW(R.VA + i) = va;
}

// per-lane Newton bookkeeping (power_flow.py:148, 168-171, 204)
struct NrState {
  double mm; int iters, conv, status; bool done;
};

__device__ __forceinline__ void nr_check(NrState& st, double mm, int it, double tol) {
  if (!st.done) {
    st.mm = mm;
    st.iters = it + 1;
    if (!(mm < INFINITY)) { st.status = GS_STATUS_NAN; st.done = true; }
    else if (mm < tol) { st.conv = 1; st.status = GS_STATUS_OK; st.done = true; }
  }
}

// The sweeps stop on the mismatch SUMMED over the buses (oracle_np.fbs_solve says why: on a radial feeder the sum bounds
// the error of every line flow); `mm`, the maximum, is what max_mismatch reports.
__device__ __forceinline__ void fbs_check(NrState& st, double mm, double sum, int it, double tol) {
  if (!st.done) {
    st.mm = mm;
    st.iters = it + 1;
    if (!(mm < INFINITY)) { st.status = GS_STATUS_NAN; st.done = true; }
    else if (2.0 * sum < tol) { st.conv = 1; st.status = GS_STATUS_OK; st.done = true; }   // (the factor 2: second-order part of a flow's error)
  }
}

enum { KIND_TREE = 0, KIND_LU = 1, KIND_FBS = 2, KIND_DENSE = 3, KIND_TREE_LDS = 4, KIND_FBS_LDS = 5, KIND_FBS_FLOW = 6 };

// =============================================================================================
// Linear solves.  Each takes the mismatch in R0/R1 and the current E/F/VM/PC/QC rows, leaves the
// Newton step applied to VM/VA for the lanes still iterating, and reports exact singularity.
// =============================================================================================

// ---- radial (forest) Jacobian: level-scheduled 2x2-block elimination, zero fill ---------------
//   bottom-up:  D_i = J_ii - sum_children C_c ;  r_i = rhs_i - sum_children q_c
//               T_i = D_i^-1 J_ip ; s_i = D_i^-1 r_i ; C_i = J_pi T_i ; q_i = J_pi s_i
//   top-down:   x_i = s_i - T_i x_p
__device__ __forceinline__ void linsolve_tree(Ctx& c, const GsSolveCfg& C, NrState& st, int par) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  int sing = 0;
  for (int lv = 0; lv < T.n_levels; ++lv) {
    const int t1 = cld(T.lvl_ptr, lv + 1);
    for (int t = cld(T.lvl_ptr, lv) + c.wave; t < t1; t += c.W) {
      const int i = cld(T.lvl_bus, t);
      Blk d = diag_block(c, i, C.jacobian_exact);
      double r0 = ROW(R.R0 + i), r1 = ROW(R.R1 + i);
      const int c1 = cld(T.child_ptr, i + 1);
      for (int cp = cld(T.child_ptr, i); cp < c1; ++cp) {
        const int ch = cld(T.child_idx, cp);
        const Blk cb = load_blk(S, R.CB + 4 * ch);
        d.a00 -= cb.a00; d.a01 -= cb.a01; d.a10 -= cb.a10; d.a11 -= cb.a11;
        r0 -= ROW(R.QV + 2 * ch); r1 -= ROW(R.QV + 2 * ch + 1);
      }
      const Blk inv = inv2(d, &sing);
      const double s0 = inv.a00 * r0 + inv.a01 * r1, s1 = inv.a10 * r0 + inv.a11 * r1;
      ROW(R.SV + 2 * i) = s0; ROW(R.SV + 2 * i + 1) = s1;
      const int p = cld(T.parent, i);
      if (p >= 0) {
        const int pp = cld(T.parent_pos, i);
        const double g = cld(T.G, pp), b = cld(T.Bv, pp);
        const Blk u = offdiag_block(c, i, p, g, b);    // J(i, p)
        const Blk l = offdiag_block(c, p, i, g, b);    // J(p, i); Ybus is symmetric
        const Blk tb = mul(inv, u);
        store_blk(S, R.TB + 4 * i, tb);
        store_blk(S, R.CB + 4 * i, mul(l, tb));
        ROW(R.QV + 2 * i) = l.a00 * s0 + l.a01 * s1;
        ROW(R.QV + 2 * i + 1) = l.a10 * s0 + l.a11 * s1;
      }
    }
    __syncthreads();
  }
  const int sing_all = wg_or(c, par, sing);
  if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
  const bool upd = !st.done;
  for (int lv = T.n_levels - 1; lv >= 0; --lv) {
    const int t1 = cld(T.lvl_ptr, lv + 1);
    for (int t = cld(T.lvl_ptr, lv) + c.wave; t < t1; t += c.W) {
      const int i = cld(T.lvl_bus, t);
      double x0 = ROW(R.SV + 2 * i), x1 = ROW(R.SV + 2 * i + 1);
      const int p = cld(T.parent, i);
      if (p >= 0) {
        const Blk tb = load_blk(S, R.TB + 4 * i);
        const double xp0 = ROW(R.X0 + p), xp1 = ROW(R.X1 + p);
        x0 -= tb.a00 * xp0 + tb.a01 * xp1;
        x1 -= tb.a10 * xp0 + tb.a11 * xp1;
      }
      ROW(R.X0 + i) = x0; ROW(R.X1 + i) = x1;
      apply_step(c, i, C.alpha, upd);
    }
    __syncthreads();
  }
}

// ---- the same forest elimination with its critical path taken out of HBM --------------------------
// The sweeps are a chain of n_levels dependent phases; what makes a phase long is not arithmetic
// but the round trip of its operands through L2/HBM and the store drain at the barrier.  Here
//  * the child -> parent messages (C, q) and the parent -> child messages (x) of two adjacent
//    levels live in LDS slots ([parity][position in level][6][64 lanes]);
//  * the operands that do NOT depend on the sweep (V, calculated P/Q, mismatch of the bus and of
//    its parent) are fetched one work item ahead, and the phases are separated by lds_barrier(),
//    which leaves those global loads (and the T/s stores) in flight;
//  * the rectangular voltage is recomputed in the top-down pass, right where V is updated.
// T_i and s_i go to the slab (HBM) in the bottom-up pass and come back, prefetched, in the
// top-down pass on the same wave.  Same arithmetic, same order of operations as linsolve_tree.
struct BuOperands { double vm, rvm, pc, qc, r0, r1, ei, fi, ep, fp, rvmp; };
struct TdOperands { double t00, t01, t10, t11, s0, s1, vm, va, rvm, e, f; };

__device__ __forceinline__ GsItemRec load_item(const GsTables& T, int k) {
  const GS_CONST GsItemRec* p = (const GS_CONST GsItemRec*)T.witems + k;
  GsItemRec r;
  r.bus = p->bus; r.parent = p->parent; r.slot = p->slot; r.parent_slot = p->parent_slot;
  r.n_children = p->n_children; r.flags = p->flags; r.level = p->level; r.ovf0 = p->ovf0;
#pragma unroll
  for (int q = 0; q < GS_ITEM_CHILDREN; ++q) r.child_slot[q] = p->child_slot[q];
  r.g = p->g; r.b = p->b; r.gd = p->gd; r.bd = p->bd;
  return r;
}

__device__ __forceinline__ BuOperands fetch_bu(Ctx& c, const GsItemRec& r) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  BuOperands o;
  const int i = r.bus, pj = r.parent >= 0 ? r.parent : r.bus;
  const double2 sc = ROW2(R.PC + i), sp = ROW2(R.P + i), vi = ROW2(R.E + i), vp = ROW2(R.E + pj);
  o.vm = ROW(R.VM + i); o.rvm = ROW(R.RVM + i); o.pc = sc.x; o.qc = sc.y;
  o.r0 = sp.x; o.r1 = sp.y;                          // specified injections; the mismatch is formed in the sweep
  o.ei = vi.x; o.fi = vi.y;
  o.ep = vp.x; o.fp = vp.y; o.rvmp = ROW(R.RVM + pj);
  return o;
}

__device__ __forceinline__ TdOperands fetch_td(Ctx& c, const GsItemRec& r) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  TdOperands o;
  const int i = r.bus;
  const double2 t0 = ROW2(R.TB + 4 * i), t1 = ROW2(R.TB + 4 * i + 2), sv = ROW2(R.SV + 2 * i), vp = ROW2(R.VM + i), vr = ROW2(R.E + i);
  o.t00 = t0.x; o.t01 = t0.y; o.t10 = t1.x; o.t11 = t1.y;
  o.s0 = sv.x; o.s1 = sv.y;
  o.vm = vp.x; o.va = vp.y; o.rvm = ROW(R.RVM + i);
  o.e = vr.x; o.f = vr.y;
  return o;
}

// J(i, j) / J(j, i) of the edge bus-parent from values and record flags (no table look-ups)
__device__ __forceinline__ Blk edge_block(double g, double b, double ei, double fi, double ej, double fj, double rvmj,
                                          int thi, int vfi, int thj, int vfj) {
  const double a = ei * ej + fi * fj;
  const double bb = fi * ej - ei * fj;
  const double gs_bc = g * bb - b * a;
  const double gc_bs = g * a + b * bb;
  Blk u;
  u.a00 = (thi && thj) ? gs_bc : 0.0;
  u.a01 = (thi && vfj) ? gc_bs * rvmj : 0.0;
  u.a10 = (vfi && thj) ? -gc_bs : 0.0;
  u.a11 = (vfi && vfj) ? gs_bc * rvmj : 0.0;
  return u;
}

__device__ __forceinline__ void linsolve_tree_lds(Ctx& c, const GsSolveCfg& C, NrState& st, int par) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double* msg = gs_dyn + c.lane;                       // slot s, component k at msg[(s * 6 + k) * 64]
#define MSG(slot, k) msg[((size_t)(slot) * GS_MSG_DOUBLES + (k)) * GS_LANES]
  const int k0 = cld(T.wl_ptr, c.wave), k1 = cld(T.wl_ptr, c.wave + 1);
  int sing = 0;
  // ---------------- bottom-up ----------------
  {
    int lv = 0;
    GsItemRec rn{};
    BuOperands on{};
    if (k0 < k1) { rn = load_item(T, k0); on = fetch_bu(c, rn); }
    for (int k = k0; k < k1; ++k) {
      const GsItemRec r = rn;
      const BuOperands o = on;
      if (k + 1 < k1) { rn = load_item(T, k + 1); on = fetch_bu(c, rn); }   // next item: in flight from here
      while (lv < r.level) { lds_barrier(); ++lv; }
      const int i = r.bus;
      const int thi = r.flags & 1, vfi = (r.flags >> 1) & 1, thp = (r.flags >> 2) & 1, vfp = (r.flags >> 3) & 1;
      // diagonal block (power_flow.py:247-248, 259-260, 270-271, 283-284)
      const double rvm = o.rvm;
      Blk d;
      {
        const double vvb = o.vm * o.vm * r.bd;
        d.a00 = thi ? (C.jacobian_exact ? (-o.qc - vvb) : (-o.qc + vvb)) : 1.0;
        d.a01 = (thi && vfi) ? (o.pc * rvm + o.vm * r.gd) : 0.0;
        d.a10 = (thi && vfi) ? (o.pc - o.vm * o.vm * r.gd) : 0.0;
        d.a11 = vfi ? (o.qc * rvm - o.vm * r.bd) : 1.0;
      }
      double r0 = thi ? (o.r0 - o.pc) : 0.0, r1 = vfi ? (o.r1 - o.qc) : 0.0;   // power_flow.py:159-165
      const int nch = r.n_children;
#pragma unroll
      for (int q = 0; q < GS_ITEM_CHILDREN; ++q) {       // static indices: the record stays in SGPRs
        if (q < nch) {
          const int slot = r.child_slot[q];
          d.a00 -= MSG(slot, 0); d.a01 -= MSG(slot, 1); d.a10 -= MSG(slot, 2); d.a11 -= MSG(slot, 3);
          r0 -= MSG(slot, 4); r1 -= MSG(slot, 5);
        }
      }
      for (int q = GS_ITEM_CHILDREN; q < nch; ++q) {
        const int slot = cld(T.ovf_slot, r.ovf0 + q - GS_ITEM_CHILDREN);
        d.a00 -= MSG(slot, 0); d.a01 -= MSG(slot, 1); d.a10 -= MSG(slot, 2); d.a11 -= MSG(slot, 3);
        r0 -= MSG(slot, 4); r1 -= MSG(slot, 5);
      }
      const Blk inv = inv2(d, &sing);
      const double s0 = inv.a00 * r0 + inv.a01 * r1, s1 = inv.a10 * r0 + inv.a11 * r1;
      ROW2(R.SV + 2 * i) = make_double2(s0, s1);
      if (r.parent >= 0) {
        const double rvmp = o.rvmp;
        const Blk u = edge_block(r.g, r.b, o.ei, o.fi, o.ep, o.fp, rvmp, thi, vfi, thp, vfp);    // J(i, p)
        const Blk l = edge_block(r.g, r.b, o.ep, o.fp, o.ei, o.fi, rvm, thp, vfp, thi, vfi);     // J(p, i)
        const Blk tb = mul(inv, u);
        store_blk(S, R.TB + 4 * i, tb);
        const Blk cb = mul(l, tb);
        MSG(r.slot, 0) = cb.a00; MSG(r.slot, 1) = cb.a01; MSG(r.slot, 2) = cb.a10; MSG(r.slot, 3) = cb.a11;
        MSG(r.slot, 4) = l.a00 * s0 + l.a01 * s1;
        MSG(r.slot, 5) = l.a10 * s0 + l.a11 * s1;
      }
    }
    while (lv < T.n_levels) { lds_barrier(); ++lv; }
  }
  stamp(c, ST_BOTTOM_UP);
  const int sing_all = wg_or(c, par, sing);            // full barrier: also drains the T/s stores
  stamp(c, ST_FLAG);
  if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
  const bool upd = !st.done;
  // ---------------- top-down: substitution, voltage update, new rectangular voltage ----------------
  {
    int lv = T.n_levels - 1;
    GsItemRec rn{};
    TdOperands on{};
    if (k0 < k1) { rn = load_item(T, k1 - 1); on = fetch_td(c, rn); }
    for (int k = k1 - 1; k >= k0; --k) {
      const GsItemRec r = rn;
      const TdOperands o = on;
      if (k - 1 >= k0) { rn = load_item(T, k - 1); on = fetch_td(c, rn); }
      while (lv > r.level) { lds_barrier(); --lv; }
      const int i = r.bus;
      double x0 = o.s0, x1 = o.s1;
      if (r.parent >= 0) {
        const double xp0 = MSG(r.parent_slot, 0), xp1 = MSG(r.parent_slot, 1);
        x0 -= o.t00 * xp0 + o.t01 * xp1;
        x1 -= o.t10 * xp0 + o.t11 * xp1;
      }
      MSG(r.slot, 0) = x0; MSG(r.slot, 1) = x1;
      if (upd) {                                       // power_flow.py:315-327
        const double dth = (r.flags & 1) ? C.alpha * x0 : 0.0;
        const double vmn = (r.flags & 2) ? o.vm + C.alpha * x1 : o.vm;
        double vm = vmn, va = o.va + dth;
        double en, fn;
        if (__any(fabs(dth) > 0.5 || !(vmn > 0.0))) {    // large step or sign flip somewhere in the wave: full evaluation
          if (vm < 0.0) { vm = -vm; va += M_PI; }
          double sn, cs;
          sincos(va, &sn, &cs);
          en = vm * cs; fn = vm * sn;
        } else {
          // V' = (Vm'/Vm) V e^{j dth}: rotate the rectangular voltage by the increment; sin/cos of a
          // small angle from their Taylor series (|dth| <= 0.5: truncation < 1e-21)
          const double z = dth * dth;
          double sp = -1.0 / 355687428096000.0;                    // -1/17!
          sp = sp * z + 1.0 / 1307674368000.0;                     // 1/15!
          sp = sp * z - 1.0 / 6227020800.0;                        // -1/13!
          sp = sp * z + 1.0 / 39916800.0;                          // 1/11!
          sp = sp * z - 1.0 / 362880.0;                            // -1/9!
          sp = sp * z + 1.0 / 5040.0;                              // 1/7!
          sp = sp * z - 1.0 / 120.0;                               // -1/5!
          sp = sp * z + 1.0 / 6.0;                                 // 1/3!
          const double sn = dth - dth * z * sp;
          double cp = 1.0 / 20922789888000.0;                      // 1/16!
          cp = cp * z - 1.0 / 87178291200.0;                       // -1/14!
          cp = cp * z + 1.0 / 479001600.0;                         // 1/12!
          cp = cp * z - 1.0 / 3628800.0;                           // -1/10!
          cp = cp * z + 1.0 / 40320.0;                             // 1/8!
          cp = cp * z - 1.0 / 720.0;                               // -1/6!
          cp = cp * z + 1.0 / 24.0;                                // 1/4!
          cp = cp * z - 0.5;                                       // -1/2!
          const double cs = 1.0 + z * cp;
          const double ratio = vmn * o.rvm;
          en = ratio * (o.e * cs - o.f * sn);
          fn = ratio * (o.e * sn + o.f * cs);
        }
        ROW2(R.VM + i) = make_double2(vm, va); ROW(R.RVM + i) = 1.0 / vm;
        ROW2(R.E + i) = make_double2(en, fn);
      }
    }
    while (lv >= 0) { lds_barrier(); --lv; }
  }
#undef MSG
  __syncthreads();                                     // V, E, F of every bus visible to every wave
  stamp(c, ST_TOP_DOWN);
}

// ---- meshed Jacobian: statically scheduled 2x2-block sparse LU ------------------------------------
// The host ordered the active buses by minimum degree, listed for every pivot its remaining neighbours and every (i, j)
// block its elimination touches (fill blocks own slots), and grouped the pivots into the LEVELS of the elimination DAG:
// pivots of one level are not adjacent in the filled graph, so they are eliminated together (round 3; one pivot per
// barrier before: 122 dependent steps for a 123-bus feeder with 26 loops that has 18 levels).  Per level:
//   phase A  every block of a pivot's column is scaled in place, A_ik <- A_ik D_k^-1 (one item per block, dealt over the waves);
//   phase B  every block the level touches is owned by ONE wave, which subtracts all of the level's updates
//            (A_ik D_k^-1) A_kj from it in registers and stores it once; the right-hand side is one more column.
// Back substitution walks the levels the other way, the pivots of a level dealt over the waves.
__device__ __forceinline__ void linsolve_lu(Ctx& c, const GsSolveCfg& C, NrState& st, int par) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  for (int t = c.wave; t < T.lu_n_piv; t += c.W) {
    const int i = cld(T.lu_piv_bus, t);
    store_blk(S, R.LUD + 4 * i, diag_block(c, i, C.jacobian_exact));
  }
  for (int q = c.wave; q < T.lu_n_orig; q += c.W) {
    const int pos = cld(T.lu_orig_pos, q);
    store_blk(S, R.LU + 4 * cld(T.lu_orig_slot, q),
              offdiag_block(c, cld(T.lu_orig_i, q), cld(T.lu_orig_j, q), cld(T.G, pos), cld(T.Bv, pos)));
  }
  for (int s = T.lu_n_orig + c.wave; s < T.lu_n_slots; s += c.W) {
    Blk z; z.a00 = z.a01 = z.a10 = z.a11 = 0.0;
    store_blk(S, R.LU + 4 * s, z);
  }
  __syncthreads();
  int sing = 0;
  const int NL = T.lu_n_levels, pw = c.wave * (NL + 1);
  for (int L = 0; L < NL; ++L) {
    {  // phase A, two items at a time: the four block loads of a pair are in flight together (an item is one memory latency)
      const int q1 = cld(T.lu_a_ptr, pw + L + 1);
      for (int q = cld(T.lu_a_ptr, pw + L); q < q1; q += 2) {
        const bool two = q + 1 < q1;
        const int qb = two ? q + 1 : q;
        const int k0 = cld(T.lu_a, 2 * q), s0 = cld(T.lu_a, 2 * q + 1), k1 = cld(T.lu_a, 2 * qb), s1 = cld(T.lu_a, 2 * qb + 1);
        const Blk d0 = load_blk(S, R.LUD + 4 * k0), d1 = load_blk(S, R.LUD + 4 * k1);
        const Blk a0 = load_blk(S, R.LU + 4 * (s0 >= 0 ? s0 : 0)), a1 = load_blk(S, R.LU + 4 * (s1 >= 0 ? s1 : 0));
        const Blk i0 = inv2(d0, &sing);
        if (s0 >= 0) store_blk(S, R.LU + 4 * s0, mul(a0, i0));
        if (two) { const Blk i1 = inv2(d1, &sing); if (s1 >= 0) store_blk(S, R.LU + 4 * s1, mul(a1, i1)); }
      }
    }
    __syncthreads();
    {  // phase B
      int p = cld(T.lu_b_ptr, pw + L);
      const int p1 = cld(T.lu_b_ptr, pw + L + 1);
      while (p < p1) {
        const int tgt = cld(T.lu_b, p), cnt = cld(T.lu_b, p + 1);
        p += 2;
        if (tgt < -T.n) {                                 // right-hand side of bus i: r_i -= (A_ik D_k^-1) r_k
          const int i = -tgt - 1 - T.n;
          double r0 = ROW(R.R0 + i), r1 = ROW(R.R1 + i);
          for (int u = 0; u < cnt; ++u, p += 2) {
            const Blk l = load_blk(S, R.LU + 4 * cld(T.lu_b, p));
            const int k = cld(T.lu_b, p + 1);
            const double rk0 = ROW(R.R0 + k), rk1 = ROW(R.R1 + k);
            r0 -= l.a00 * rk0 + l.a01 * rk1;
            r1 -= l.a10 * rk0 + l.a11 * rk1;
          }
          ROW(R.R0 + i) = r0; ROW(R.R1 + i) = r1;
        } else {
          const int row = (tgt >= 0) ? (R.LU + 4 * tgt) : (R.LUD + 4 * (-tgt - 1));
          Blk a = load_blk(S, row);
          for (int u = 0; u < cnt; ++u, p += 2) {
            const Blk upd = mul(load_blk(S, R.LU + 4 * cld(T.lu_b, p)), load_blk(S, R.LU + 4 * cld(T.lu_b, p + 1)));
            a.a00 -= upd.a00; a.a01 -= upd.a01; a.a10 -= upd.a10; a.a11 -= upd.a11;
          }
          store_blk(S, row, a);
        }
      }
    }
    __syncthreads();
  }
  stamp(c, ST_BOTTOM_UP);
  const int sing_all = wg_or(c, par, sing);
  if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
  const bool upd = !st.done;
  for (int L = NL - 1; L >= 0; --L) {      // back substitution: x_k = D_k^-1 (r_k - sum_j A_kj x_j), every j in a higher level
    const int q1 = cld(T.lu_c_ptr, pw + L + 1);
    for (int q = cld(T.lu_c_ptr, pw + L); q < q1; ++q) {
      const int t = cld(T.lu_c, q), k = cld(T.lu_piv_bus, t);
      int dummy = 0;
      const Blk inv = inv2(load_blk(S, R.LUD + 4 * k), &dummy);
      double r0 = ROW(R.R0 + k), r1 = ROW(R.R1 + k);
      const int n1 = cld(T.lu_nb_ptr, t + 1);
      for (int u = cld(T.lu_nb_ptr, t); u < n1; ++u) {
        const int j = cld(T.lu_nb_bus, u);
        const Blk akj = load_blk(S, R.LU + 4 * cld(T.lu_nb_kj, u));
        const double xj0 = ROW(R.X0 + j), xj1 = ROW(R.X1 + j);
        r0 -= akj.a00 * xj0 + akj.a01 * xj1;
        r1 -= akj.a10 * xj0 + akj.a11 * xj1;
      }
      ROW(R.X0 + k) = inv.a00 * r0 + inv.a01 * r1;
      ROW(R.X1 + k) = inv.a10 * r0 + inv.a11 * r1;
      apply_step(c, k, C.alpha, upd);
    }
    __syncthreads();
  }
  stamp(c, ST_TOP_DOWN);
}

// Iteration 0 with the handle's flat-start factors (GsTables::lu_flat): the blocks are wave-uniform scalars, only the
// right-hand side and the solution go through the rows.
__device__ __forceinline__ Blk flat_blk(const GsTables& T, int idx) {
  Blk b; b.a00 = cld(T.lu_flat, 4 * idx); b.a01 = cld(T.lu_flat, 4 * idx + 1); b.a10 = cld(T.lu_flat, 4 * idx + 2); b.a11 = cld(T.lu_flat, 4 * idx + 3);
  return b;
}
__device__ __forceinline__ void linsolve_lu_flat(Ctx& c, const GsSolveCfg& C, NrState& st, int par) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  const int NL = T.lu_n_levels, pw = c.wave * (NL + 1);
  for (int L = 0; L < NL; ++L) {          // forward: r_i -= (A_ik D_k^-1) r_k, level by level
    int p = cld(T.lu_r_ptr, pw + L);
    const int p1 = cld(T.lu_r_ptr, pw + L + 1);
    while (p < p1) {
      const int i = -cld(T.lu_r, p) - 1 - T.n, cnt = cld(T.lu_r, p + 1);
      p += 2;
      double r0 = ROW(R.R0 + i), r1 = ROW(R.R1 + i);
      for (int u = 0; u < cnt; ++u, p += 2) {
        const Blk l = flat_blk(T, cld(T.lu_r, p));
        const int k = cld(T.lu_r, p + 1);
        const double rk0 = ROW(R.R0 + k), rk1 = ROW(R.R1 + k);
        r0 -= l.a00 * rk0 + l.a01 * rk1;
        r1 -= l.a10 * rk0 + l.a11 * rk1;
      }
      ROW(R.R0 + i) = r0; ROW(R.R1 + i) = r1;
    }
    __syncthreads();
  }
  stamp(c, ST_BOTTOM_UP);
  const bool sing_all = cld(T.lu_flat, 4 * (T.lu_n_slots + T.n)) != 0.0;
  if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
  const bool upd = !st.done;
  for (int L = NL - 1; L >= 0; --L) {
    const int q1 = cld(T.lu_c_ptr, pw + L + 1);
    for (int q = cld(T.lu_c_ptr, pw + L); q < q1; ++q) {
      const int t = cld(T.lu_c, q), k = cld(T.lu_piv_bus, t);
      int dummy = 0;
      const Blk inv = inv2(flat_blk(T, T.lu_n_slots + k), &dummy);
      double r0 = ROW(R.R0 + k), r1 = ROW(R.R1 + k);
      const int n1 = cld(T.lu_nb_ptr, t + 1);
      for (int u = cld(T.lu_nb_ptr, t); u < n1; ++u) {
        const int j = cld(T.lu_nb_bus, u);
        const Blk akj = flat_blk(T, cld(T.lu_nb_kj, u));
        const double xj0 = ROW(R.X0 + j), xj1 = ROW(R.X1 + j);
        r0 -= akj.a00 * xj0 + akj.a01 * xj1;
        r1 -= akj.a10 * xj0 + akj.a11 * xj1;
      }
      ROW(R.X0 + k) = inv.a00 * r0 + inv.a01 * r1;
      ROW(R.X1 + k) = inv.a10 * r0 + inv.a11 * r1;
      apply_step(c, k, C.alpha, upd);
    }
    __syncthreads();
  }
  stamp(c, ST_TOP_DOWN);
}

// ---- dense, partially pivoted LU per instance: the reference-faithful linear solve -------------
// (np.linalg.solve = LAPACK dgesv, power_flow.py:187): same unknown order, same pivot rule
// (largest |a_ik| in the column), exact-zero pivot = singular.  Row exchanges are per instance,
// so matrix rows are reached through a per-lane permutation (a gather: each lane reads its own
// row at its own lane slot).  This path exists for parity with the as-coded Jacobian, whose 2x2
// diagonal blocks can be exactly singular; it is not the fast path.
#define DA_AT(prow, cc) S.lane_row(((size_t)R.DA + (size_t)(prow) * N + (size_t)(cc)) * GS_LANES)     /* prow differs between lanes */

__device__ __forceinline__ void linsolve_dense(Ctx& c, const GsSolveCfg& C, NrState& st, int par) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  const int N = T.dn_N;
  for (int q = c.wave; q < N * N; q += c.W) ROW(R.DA + q) = 0.0;
  for (int q = c.wave; q < N; q += c.W) ROW(R.DPERM + q) = (double)q;
  __syncthreads();
  for (int i = c.wave; i < T.n; i += c.W) {
    const int ri0 = cld(T.dn_th_idx, i), ri1 = cld(T.dn_vm_idx, i);
    if (ri0 >= 0) ROW(R.DB + ri0) = ROW(R.R0 + i);
    if (ri1 >= 0) ROW(R.DB + ri1) = ROW(R.R1 + i);
    const int p1 = cld(T.row_ptr, i + 1);
    for (int p = cld(T.row_ptr, i); p < p1; ++p) {
      const int j = cld(T.col, p);
      const int cj0 = cld(T.dn_th_idx, j), cj1 = cld(T.dn_vm_idx, j);
      const Blk blk = (j == i) ? diag_block(c, i, C.jacobian_exact) : offdiag_block(c, i, j, cld(T.G, p), cld(T.Bv, p));
      if (ri0 >= 0 && cj0 >= 0) ROW(R.DA + ri0 * N + cj0) = blk.a00;
      if (ri0 >= 0 && cj1 >= 0) ROW(R.DA + ri0 * N + cj1) = blk.a01;
      if (ri1 >= 0 && cj0 >= 0) ROW(R.DA + ri1 * N + cj0) = blk.a10;
      if (ri1 >= 0 && cj1 >= 0) ROW(R.DA + ri1 * N + cj1) = blk.a11;
    }
  }
  __syncthreads();
  int sing = 0;
  for (int k = 0; k < N; ++k) {
    if (c.wave == 0) {
      double best = -1.0; int bi = k;
      for (int i = k; i < N; ++i) {
        const int pi = (int)ROW(R.DPERM + i);
        const double v = fabs(DA_AT(pi, k));
        if (v > best) { best = v; bi = i; }
      }
      if (!(best > 0.0)) sing = 1;
      const double pk = ROW(R.DPERM + k);
      const double pb = S.lane_row((size_t)(R.DPERM + bi) * GS_LANES);
      S.lane_row((size_t)(R.DPERM + bi) * GS_LANES) = pk;
      ROW(R.DPERM + k) = pb;
    }
    __syncthreads();
    const int pk = (int)ROW(R.DPERM + k);
    const double akk = DA_AT(pk, k);
    const double bk = S.lane_row((size_t)(R.DB + pk) * GS_LANES);
    for (int i = k + 1 + c.wave; i < N; i += c.W) {
      const int pi = (int)ROW(R.DPERM + i);
      const double l = DA_AT(pi, k) / akk;
      if (__any(l != 0.0)) {
        for (int cc = k + 1; cc < N; ++cc) DA_AT(pi, cc) -= l * DA_AT(pk, cc);
        S.lane_row((size_t)(R.DB + pi) * GS_LANES) -= l * bk;
      }
    }
    __syncthreads();
  }
  const int sing_all = wg_or(c, par, sing);
  if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
  const bool upd = !st.done;
  if (c.wave == 0) {
    for (int k = N - 1; k >= 0; --k) {
      const int pk = (int)ROW(R.DPERM + k);
      double s = S.lane_row((size_t)(R.DB + pk) * GS_LANES);
      for (int cc = k + 1; cc < N; ++cc) s -= DA_AT(pk, cc) * ROW(R.DX + cc);
      ROW(R.DX + k) = s / DA_AT(pk, k);
    }
    for (int i = 0; i < T.n; ++i) {
      const int c0 = cld(T.dn_th_idx, i), c1 = cld(T.dn_vm_idx, i);
      ROW(R.X0 + i) = (c0 >= 0) ? ROW(R.DX + c0) : 0.0;
      ROW(R.X1 + i) = (c1 >= 0) ? ROW(R.DX + c1) : 0.0;
      apply_step(c, i, C.alpha, upd);
    }
  }
  __syncthreads();
}

// =============================================================================================
// Newton-Raphson driver (power_flow.py:143-193).  Returns with E/F/PC/QC describing the final V
// (recomputed when the loop ended on the iteration cap, i.e. after an update).
// =============================================================================================
// FLAT_DONE: the environment prologue's injection pass (which ends in a row barrier) wrote the flat start already
template <int KIND, bool FLAT_DONE>
__device__ __forceinline__ void newton_loop(Ctx& c, const GsSolveCfg& C, NrState& st) {
  if (!FLAT_DONE) {
    flat_start(c);
    __syncthreads();
  }
  bool stale = true;
  for (int it = 0; it < C.max_iterations; ++it) {
    if (KIND != KIND_TREE_LDS && it > 0) {              // flat_start wrote E/F; the LDS forest solve refreshes them itself
      to_rect(c);
      __syncthreads();
    }
    if (it == 0) stamp(c, ST_INIT);
    const double lm = mismatch_rows<KIND == KIND_TREE_LDS ? 1 : 2>(c);
    stamp(c, ST_MISMATCH);
    const double mm = wg_max(c, it & 1, lm);
    stamp(c, ST_FINAL_MISMATCH);
    nr_check(st, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;
    if (KIND == KIND_TREE) linsolve_tree(c, C, st, it & 1);
    else if (KIND == KIND_TREE_LDS) linsolve_tree_lds(c, C, st, it & 1);
    else if (KIND == KIND_LU) { if (it == 0 && c.T.lu_flat != nullptr) linsolve_lu_flat(c, C, st, it & 1); else linsolve_lu(c, C, st, it & 1); }
    else linsolve_dense(c, C, st, it & 1);
    stale = true;
  }
  if (stale) {
    if (KIND != KIND_TREE_LDS) {
      to_rect(c);
      __syncthreads();
    }
    (void)mismatch_rows<0>(c);
  }
  __syncthreads();
  stamp(c, ST_FINAL_MISMATCH);
}

// =============================================================================================
// Forward/backward sweep on a radial feeder (constant-power buses).  New functionality: the
// reference names DistributionPowerFlow in README.md:187-197 but ships no implementation.
// Same convergence test as Newton (power mismatch < tolerance) so a converged answer satisfies
// the reference's own acceptance criterion.  State is rectangular; no trigonometry in the loop.
//   backward:  J_i = -conj(S_i / V_i) + sum_children J_c        (branch current parent -> i)
//   forward:   V_i = V_parent - J_i / y_i
// =============================================================================================
__device__ __forceinline__ void fbs_loop(Ctx& c, const GsSolveCfg& C, NrState& st) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  for (int i = c.wave; i < T.n; i += c.W) {
    ROW(R.E + i) = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
    ROW(R.F + i) = 0.0;
  }
  __syncthreads();
  bool stale = true;
  for (int it = 0; it < C.max_iterations; ++it) {
    c.fsum = 0.0;
    const double lm = mismatch_rows<0>(c);
    double sum;
    const double mm = wg_max_sum(c, lm, c.fsum, &sum);
    fbs_check(st, mm, sum, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;
    const bool upd = !st.done;
    for (int lv = 0; lv < T.n_levels; ++lv) {          // backward sweep, deepest level first
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + c.wave; t < t1; t += c.W) {
        const int i = cld(T.lvl_bus, t);
        const double e = ROW(R.E + i), f = ROW(R.F + i), p = ROW(R.P + i), q = ROW(R.Q + i);
        const double d = e * e + f * f;
        // conj((p + jq) / (e + jf)) = ((p e + q f) - j (q e - p f)) / d
        double jr = -(p * e + q * f) / d, ji = (q * e - p * f) / d;
        const int c1 = cld(T.child_ptr, i + 1);
        for (int cp = cld(T.child_ptr, i); cp < c1; ++cp) {
          const int ch = cld(T.child_idx, cp);
          jr += ROW(R.JR + ch); ji += ROW(R.JI + ch);
        }
        ROW(R.JR + i) = jr; ROW(R.JI + i) = ji;
      }
      __syncthreads();
    }
    for (int lv = T.n_levels - 1; lv >= 0; --lv) {     // forward sweep, roots first
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + c.wave; t < t1; t += c.W) {
        const int i = cld(T.lvl_bus, t);
        const int p = cld(T.fbs_parent, i), pp = cld(T.fbs_parent_pos, i);
        const double yr = -cld(T.G, pp), yi = -cld(T.Bv, pp);      // branch admittance = -Y_ip
        const double yd = yr * yr + yi * yi;
        const double jr = ROW(R.JR + i), ji = ROW(R.JI + i);
        const double dr = (jr * yr + ji * yi) / yd, di = (ji * yr - jr * yi) / yd;   // J / y
        if (upd) {
          ROW(R.E + i) = ROW(R.E + p) - dr;
          ROW(R.F + i) = ROW(R.F + p) - di;
        }
      }
      __syncthreads();
    }
    stale = true;
  }
  if (stale) (void)mismatch_rows<0>(c);
  __syncthreads();
  for (int i = c.wave; i < T.n; i += c.W) ROW(R.VM + i) = hypot(ROW(R.E + i), ROW(R.F + i));
  __syncthreads();
}

// ---- the same sweep with its level messages in LDS, its operands prefetched, and the mismatch
// evaluated INSIDE the backward sweep ------------------------------------------------------------
// S_calc = V conj(Y V) (power_flow.py:150) is restated branch by branch: with K_i = y_i (V_i - V_parent)
// the current bus i sends towards its parent, (Y V)_i = K_i - sum_children K_c, so a bus can form
// its own mismatch from its parent's voltage and its children's K -- which travel up in the same
// LDS message as the branch currents J.  One sweep therefore yields both the convergence test at the
// current V and the currents for the next forward sweep; there is no separate mismatch pass.
// Records are the per-wave forest items with parent = the FBS parent (the slack for a root),
// (g, b) = z = 1/y of the branch to the parent and (gd, bd) = y.
struct FbsOperands { double e, f, p, q, ep, fp; };

__device__ __forceinline__ FbsOperands fetch_fbs(Ctx& c, const GsItemRec& r) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  FbsOperands o;
  const double2 v = ROW2(R.E + r.bus), sp = ROW2(R.P + r.bus), vp = ROW2(R.E + r.parent);
  o.e = v.x; o.f = v.y; o.p = sp.x; o.q = sp.y;
  o.ep = vp.x; o.fp = vp.y;
  return o;
}

__device__ __forceinline__ void fbs_backward(Ctx& c, int k0, int k1, double* lmax_out, double* psum_out) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double* msg = gs_dyn + c.lane;
#define MSG(slot, k) msg[((size_t)(slot) * GS_MSG_DOUBLES + (k)) * GS_LANES]
  double lmax = 0.0, psum = 0.0, bad = 0.0;
  int lv = 0;
  GsItemRec rn{};
  FbsOperands on{};
  if (k0 < k1) { rn = load_item(T, k0); on = fetch_fbs(c, rn); }
  for (int k = k0; k < k1; ++k) {
    const GsItemRec r = rn;
    const FbsOperands o = on;
    if (k + 1 < k1) { rn = load_item(T, k + 1); on = fetch_fbs(c, rn); }
    while (lv < r.level) { lds_barrier(); ++lv; }
    // current towards the parent implied by the voltages: K = y (V_i - V_p)
    const double dr = o.e - o.ep, di = o.f - o.fp;
    const double kr = r.gd * dr - r.bd * di, ki = r.gd * di + r.bd * dr;
    double sjr = 0.0, sji = 0.0, skr = 0.0, ski = 0.0;
    const int nch = r.n_children;
#pragma unroll
    for (int u = 0; u < GS_ITEM_CHILDREN; ++u) {
      if (u < nch) {
        const int slot = r.child_slot[u];
        sjr += MSG(slot, 0); sji += MSG(slot, 1); skr += MSG(slot, 2); ski += MSG(slot, 3);
      }
    }
    for (int u = GS_ITEM_CHILDREN; u < nch; ++u) {
      const int slot = cld(T.ovf_slot, r.ovf0 + u - GS_ITEM_CHILDREN);
      sjr += MSG(slot, 0); sji += MSG(slot, 1); skr += MSG(slot, 2); ski += MSG(slot, 3);
    }
    // (Y V)_i = K_i - sum K_c ;  S_calc = V conj(Y V) ; mismatch (power_flow.py:150-168)
    const double icr = kr - skr, ici = ki - ski;
    const double pc = o.e * icr + o.f * ici, qc = o.f * icr - o.e * ici;
    const double dP = o.p - pc, dQ = o.q - qc;
    lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ)));         // fmax drops a NaN; `bad` (x * 0 is NaN for NaN / inf) keeps it
    c.fsum += fabs(dP) + fabs(dQ);
    bad = fma(dP, 0.0, fma(dQ, 0.0, bad));
    psum += pc;
    if (r.flags & 16) psum -= o.ep * kr + o.fp * ki;    // the slack's share: Re(V_s conj(-K_root))
    // branch current for the next forward sweep: J_i = -conj(S_spec / V_i) + sum J_c
    const double rd = 1.0 / (o.e * o.e + o.f * o.f);
    const double jr = sjr - (o.p * o.e + o.q * o.f) * rd, ji = sji + (o.q * o.e - o.p * o.f) * rd;
    MSG(r.slot, 0) = jr; MSG(r.slot, 1) = ji; MSG(r.slot, 2) = kr; MSG(r.slot, 3) = ki;
    ROW2(R.JR + r.bus) = make_double2(jr, ji);
  }
  while (lv < T.n_levels) { lds_barrier(); ++lv; }
#undef MSG
  if (bad != bad) lmax = INFINITY;                         // a non-finite mismatch anywhere in this wave's items
  *lmax_out = lmax; *psum_out = psum;
}

// Backward sweep without the mismatch: J_i = -conj(S_spec / V_i) + sum J_c, up through LDS and into the J rows.
struct FbsLightOperands { double e, f, p, q; };
__device__ __forceinline__ FbsLightOperands fetch_fbs_light(Ctx& c, int bus) {
  const GsRows& R = c.R; GsLaneRows S = c.S;
  const double2 v = ROW2(R.E + bus), sp = ROW2(R.P + bus);
  return FbsLightOperands{v.x, v.y, sp.x, sp.y};
}

__device__ __forceinline__ void fbs_backward_light(Ctx& c, int k0, int k1) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double* msg = gs_dyn + c.lane;
#define MSG(slot, k) msg[((size_t)(slot) * GS_MSG_DOUBLES + (k)) * GS_LANES]
  int lv = 0;
  GsItemRec rn{};
  FbsLightOperands on{};
  if (k0 < k1) { rn = load_item(T, k0); on = fetch_fbs_light(c, rn.bus); }
  for (int k = k0; k < k1; ++k) {
    const GsItemRec r = rn;
    const FbsLightOperands o = on;
    if (k + 1 < k1) { rn = load_item(T, k + 1); on = fetch_fbs_light(c, rn.bus); }
    // own injection current first: it does not depend on the children, so the division overlaps the wait for them
    const double rd = 1.0 / (o.e * o.e + o.f * o.f);
    double jr = -(o.p * o.e + o.q * o.f) * rd, ji = (o.q * o.e - o.p * o.f) * rd;
    while (lv < r.level) { lds_barrier(); ++lv; }
    const int nch = r.n_children;
#pragma unroll
    for (int u = 0; u < GS_ITEM_CHILDREN; ++u) {
      if (u < nch) { const int slot = r.child_slot[u]; jr += MSG(slot, 0); ji += MSG(slot, 1); }
    }
    for (int u = GS_ITEM_CHILDREN; u < nch; ++u) {
      const int slot = cld(T.ovf_slot, r.ovf0 + u - GS_ITEM_CHILDREN);
      jr += MSG(slot, 0); ji += MSG(slot, 1);
    }
    MSG(r.slot, 0) = jr; MSG(r.slot, 1) = ji;
    ROW2(R.JR + r.bus) = make_double2(jr, ji);
  }
  while (lv < T.n_levels) { lds_barrier(); ++lv; }
#undef MSG
}

// The sweep solver.  The first backward sweep is the full one above (mismatch at the flat start from K = y dV, which
// is not zero when the slack's set point is not 1).  After that the mismatch is evaluated on the way DOWN: a forward
// sweep leaves V_i - V_parent = -z_i J_i exactly, so the current the new voltages draw at bus i is the injection
// current conj(S_spec / V_old) the backward sweep used, and S_spec - V_new conj(I_old) is what the next backward
// sweep of the textbook loop would report -- one sweep earlier, and the later backward sweeps carry J only.
// Iteration counts, mismatch and losses are those of the textbook loop (oracle: fbs_solve); an instance that has
// converged keeps its rows, its mismatch and its loss sum.
// FLAT_DONE: the flat start was already written (by the environment prologue's injection pass, which ends in a barrier)
template <bool FLAT_DONE>
__device__ __forceinline__ double fbs_loop_lds(Ctx& c, const GsSolveCfg& C, NrState& st) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double* msg = gs_dyn + c.lane;
#define MSG(slot, k) msg[((size_t)(slot) * GS_MSG_DOUBLES + (k)) * GS_LANES]
  if (!FLAT_DONE) {
    for (int i = c.wave; i < T.n; i += c.W) ROW2(R.E + i) = make_double2(cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0, 0.0);
    __syncthreads();
  }
  const int k0 = cld(T.wl_ptr, c.wave), k1 = cld(T.wl_ptr, c.wave + 1);
  double psum = 0.0;
  {
    double lmax, sum;
    stamp(c, ST_INIT);
    c.fsum = 0.0;
    fbs_backward(c, k0, k1, &lmax, &psum);
    stamp(c, ST_BOTTOM_UP);
    const double mm = wg_max_sum(c, lmax, c.fsum, &sum);   // full barrier: also drains the J rows
    stamp(c, ST_FLAG);
    fbs_check(st, mm, sum, 0, C.tolerance);
  }
  for (int it = 0; it < C.max_iterations && !__all(st.done); ++it) {
    const bool upd = !st.done;
    double lmax = 0.0, pnew = 0.0, bad = 0.0;
    c.fsum = 0.0;
    {  // forward sweep: V_i = V_parent - z_i J_i, and the mismatch / sum of P_calc at the new voltages
      int lv = T.n_levels - 1;
      GsItemRec rn{};
      double jr = 0, ji = 0;
      FbsLightOperands on{};
      if (k0 < k1) { rn = load_item(T, k1 - 1); const double2 j = ROW2(R.JR + rn.bus); jr = j.x; ji = j.y; on = fetch_fbs_light(c, rn.bus); }
      for (int k = k1 - 1; k >= k0; --k) {
        const GsItemRec r = rn;
        const double cjr = jr, cji = ji;
        const FbsLightOperands o = on;
        if (k - 1 >= k0) { rn = load_item(T, k - 1); const double2 j = ROW2(R.JR + rn.bus); jr = j.x; ji = j.y; on = fetch_fbs_light(c, rn.bus); }
        // the current this bus drew in the backward sweep, I_old = conj(S_spec / V_old): independent of the parent
        const double rd = 1.0 / (o.e * o.e + o.f * o.f);
        const double icr = (o.p * o.e + o.q * o.f) * rd, ici = (o.p * o.f - o.q * o.e) * rd;
        while (lv > r.level) { lds_barrier(); --lv; }
        const bool root = r.flags & 16;                 // parent is the slack bus: its voltage is the set point, never updated
        double ep, fp;
        if (root) { ep = cld(T.v_set, r.parent); fp = 0.0; }
        else { ep = MSG(r.parent_slot, 0); fp = MSG(r.parent_slot, 1); }
        const double en = ep - (cjr * r.g - cji * r.b), fn = fp - (cjr * r.b + cji * r.g);
        MSG(r.slot, 0) = en; MSG(r.slot, 1) = fn;
        if (upd) ROW2(R.E + r.bus) = make_double2(en, fn);
        // S_calc = V_new conj(I_old); mismatch (power_flow.py:150-168)
        const double pc = en * icr + fn * ici, qc = fn * icr - en * ici;
        const double dP = o.p - pc, dQ = o.q - qc;
        lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ)));
        c.fsum += fabs(dP) + fabs(dQ);
        bad = fma(dP, 0.0, fma(dQ, 0.0, bad));
        pnew += pc;
        if (root) pnew += ep * cjr;                      // the slack's share: Re(V_s conj(J_root)), V_s real
      }
      while (lv >= 0) { lds_barrier(); --lv; }
    }
    if (bad != bad) lmax = INFINITY;
    stamp(c, ST_TOP_DOWN);
    if (upd) psum = pnew;                                // losses at the voltages just stored
    if (it + 1 >= C.max_iterations) { __syncthreads(); break; }   // iteration cap: mismatch / count stay the last backward sweep's
    double sum;
    const double mm = wg_max_sum(c, lmax, c.fsum, &sum);   // full barrier
    stamp(c, ST_FLAG);
    fbs_check(st, mm, sum, it + 1, C.tolerance);
    if (__all(st.done)) break;
    fbs_backward_light(c, k0, k1);
    __syncthreads();                                     // the J rows are read back by the forward sweep
    stamp(c, ST_BOTTOM_UP);
  }
#undef MSG
  return psum;
}

// ---- the sweep solver as a DATAFLOW over LDS: no level barriers -----------------------------------
// A level barrier makes every wave wait for the slowest item of the level, and an item is one wave's serial
// instruction stream (division, mismatch, record decode: ~1 k cycles) -- 11 levels x 6 sweeps of that are the
// critical path of the barrier version above.  What a bus really waits for is ONE message: its parent's voltage on
// the way down, its children's currents on the way up.  Here every bus owns a 16-byte-per-lane message slot in LDS
// (so nothing is reused inside a sweep and no barrier is needed to protect a slot) plus a flag word; a producer
// writes the message, then the sweep's epoch number into the flag; a consumer polls the flag, then reads the
// message.  Everything else of an item (division, mismatch, row stores, the next record) runs after the flag is
// posted, i.e. off the critical path, and the sweeps of one iteration flow into each other without a barrier.
// Waves take their items in level order, so the item with the lowest level among those not yet done can always run:
// no deadlock (all W waves of a workgroup are resident); the poll is bounded all the same.
// Slot reuse is safe without barriers: V_i replaces J_i only after the parent consumed J_i (the parent's forward item
// follows its backward item, and i's forward item waits for the parent's), J_i replaces V_i only after the children
// consumed V_i (their backward items follow their forward items, and i's backward item waits for theirs).
// Flat start only (every non-slack voltage is exactly 1, so no child sends a K): warm-started handles use fbs_lds.
// The LDS executes the instructions of a compute unit in the order they were issued to it, so "message, then flag"
// by the producer and "flag, then message" by the consumer need no s_waitcnt in between: the consumer asks for the
// flag and -- speculatively -- the message in one round trip, and asks again when the flag was not there yet.
// volatile keeps the compiler from reordering or merging the accesses.
#define GS_FLOW_SPIN_CAP (1 << 18)
typedef volatile __attribute__((address_space(3))) double* GsLdsD;
typedef volatile __attribute__((address_space(3))) int* GsLdsI;
__device__ __forceinline__ double2 flow_take(const double* m, const int* flag, int epoch) {
  GsLdsD lm = (GsLdsD)m; GsLdsI lf = (GsLdsI)flag;
  double2 v;
  bool seen = false;
  for (int spin = 0; spin < GS_FLOW_SPIN_CAP; ++spin) {
    const int f = *lf;
    v.x = lm[0]; v.y = lm[GS_LANES];
    if (__builtin_amdgcn_readfirstlane(f) - epoch >= 0) { seen = true; break; }
  }
  // a hand-off that never arrived must not pass for data: NaN makes the mismatch of this group non-finite, i.e. its
  // instances end with GS_STATUS_NAN instead of "converged" on a stale message
  if (!seen) v = make_double2(NAN, NAN);
  return v;
}
__device__ __forceinline__ void flow_give(Ctx& c, double* m, int* flag, int epoch, double x, double y) {
  GsLdsD lm = (GsLdsD)m; GsLdsI lf = (GsLdsI)flag;
  lm[0] = x; lm[GS_LANES] = y;
  if (c.lane == 0) *lf = epoch;
}

// One iteration is three passes over a wave's own items, and only the two CHAIN passes wait for anybody:
//   forward chain  (epoch 2t):   wait for the parent's V, V_i = V_parent - z_i J_i with J_i from the own slot, post V_i
//   body           (no waits):   mismatch S_spec - V_new conj(I_old), the losses sum, I_new = conj(S_spec / V_new),
//                                V_new into the E rows -- all the arithmetic
//   [workgroup maximum of the mismatch = the only barrier of the iteration; convergence check]
//   backward chain (epoch 2t+1): J_i = -I_new + sum of the children's J, post J_i
// so the depth of the feeder is paid in LDS round trips (a few hundred cycles per level), not in whole items.
// A wave owns at most GS_FLOW_ITEMS buses (the host selects this kernel only then), the passes are unrolled over
// them, and S_spec and the injection current of each stay in REGISTERS for the whole solve: the iteration reads
// no row at all (with one-item-ahead prefetch an item cost the latency of its row loads, ~1 k cycles), I_old of the
// next mismatch is simply I_new of this one (one division per bus and iteration instead of two), and J never
// leaves LDS.
#define GS_FLOW_ITEMS 8
// HAVE_P: the environment prologue left S_spec = (Pin[j], 0) of the wave's buses in registers
template <bool HAVE_P>
__device__ __forceinline__ double fbs_loop_flow(Ctx& c, const GsSolveCfg& C, NrState& st, const double* Pin) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  double* msg = gs_dyn + c.lane;
  int* flags = (int*)(gs_dyn + (size_t)T.n * 2 * GS_LANES);
  const GS_CONST GsItemRec* recs = (const GS_CONST GsItemRec*)T.witems;
#define FMSG(bus, k) msg[((size_t)(bus) * 2 + (k)) * GS_LANES]
  const int k0 = cld(T.wl_ptr, c.wave);
  const int nit = min(cld(T.wl_ptr, c.wave + 1) - k0, GS_FLOW_ITEMS);
  double P[GS_FLOW_ITEMS], Q[GS_FLOW_ITEMS], IR[GS_FLOW_ITEMS], II[GS_FLOW_ITEMS];
#pragma unroll
  for (int j = 0; j < GS_FLOW_ITEMS; ++j) {          // all S_spec rows of the wave in flight together
    P[j] = HAVE_P ? Pin[j] : 0.0; Q[j] = 0.0;
    if (!HAVE_P && j < nit) { const double2 sp = ROW2(R.P + recs[k0 + j].bus); P[j] = sp.x; Q[j] = sp.y; }
  }
  double psum = 0.0;
  int epoch = 1;
  stamp(c, ST_INIT);
  {  // first backward chain, at the flat start: V_i = 1 for every bus but the slack, so K_i = y_i (1 - V_parent) is
     // non-zero for the children of the slack only and no bus receives a K from below; S_calc_i = V_i conj(K_i)
    double lmax = 0.0, bad = 0.0;
    c.fsum = 0.0;
    GsItemRec rn{};
    if (nit > 0) rn = load_item(T, k0);
#pragma unroll
    for (int j = 0; j < GS_FLOW_ITEMS; ++j) {
      if (j < nit) {
        const GsItemRec r = rn;
        const double p = P[j], q = Q[j];
        const bool root = r.flags & 16;
        const double ep = root ? cld(T.v_set, r.parent) : 1.0;
        const double dr = 1.0 - ep;
        const double kr = r.gd * dr, ki = r.bd * dr;
        const double pc = kr, qc = -ki;                      // V = 1: S_calc = conj(K)
        const double dP = p - pc, dQ = q - qc;
        lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ)));
        c.fsum += fabs(dP) + fabs(dQ);
        bad = fma(dP, 0.0, fma(dQ, 0.0, bad));
        psum += pc;
        if (root) psum -= ep * kr;                           // the slack's share: Re(V_s conj(-K_root))
        IR[j] = p; II[j] = -q;                               // I = conj(S_spec / V) at V = 1
        double jr = -p, ji = q;                              // J_i = -I_i + sum J_c
        const int nch = r.n_children;
#pragma unroll
        for (int u = 0; u < GS_ITEM_CHILDREN; ++u) {
          if (u < nch) { const int ch = r.child_slot[u]; const double2 jc = flow_take(&FMSG(ch, 0), flags + ch, epoch); jr += jc.x; ji += jc.y; }
        }
        for (int u = GS_ITEM_CHILDREN; u < nch; ++u) {
          const int ch = cld(T.ovf_slot, r.ovf0 + u - GS_ITEM_CHILDREN);
          const double2 jc = flow_take(&FMSG(ch, 0), flags + ch, epoch); jr += jc.x; ji += jc.y;
        }
        flow_give(c, &FMSG(r.bus, 0), flags + r.bus, epoch, jr, ji);
        if (j + 1 < nit) rn = load_item(T, k0 + j + 1);
      } else { IR[j] = 0.0; II[j] = 0.0; }
    }
    if (bad != bad) lmax = INFINITY;
    stamp(c, ST_BOTTOM_UP);
    double sum;
    const double mm = wg_max_sum(c, lmax, c.fsum, &sum);
    stamp(c, ST_FLAG);
    fbs_check(st, mm, sum, 0, C.tolerance);
  }
  for (int it = 0; it < C.max_iterations && !__all(st.done); ++it) {
    const bool upd = !st.done;
    double lmax = 0.0, pnew = 0.0, bad = 0.0;
    c.fsum = 0.0;
    ++epoch;
    {  // forward chain: V_i = V_parent - z_i J_i
      GsItemRec rn{};
      if (nit > 0) rn = load_item(T, k0 + nit - 1);
#pragma unroll
      for (int j = GS_FLOW_ITEMS - 1; j >= 0; --j) {
        if (j < nit) {
          const GsItemRec r = rn;
          const double cjr = FMSG(r.bus, 0), cji = FMSG(r.bus, 1);                 // this wave's own backward item left it there
          const double zr = cjr * r.g - cji * r.b, zi = cjr * r.b + cji * r.g;     // z J: known before the parent is
          const bool root = r.flags & 16;                 // parent is the slack bus: its voltage is the set point, never updated
          double ep, fp;
          if (root) { ep = cld(T.v_set, r.parent); fp = 0.0; pnew += ep * cjr; }   // the slack's share of the losses sum: Re(V_s conj(J_root)), V_s real
          else { const double2 vp = flow_take(&FMSG(r.parent, 0), flags + r.parent, epoch); ep = vp.x; fp = vp.y; }
          flow_give(c, &FMSG(r.bus, 0), flags + r.bus, epoch, ep - zr, fp - zi);
          if (j > 0) rn = load_item(T, k0 + j - 1);       // after the post: a scalar load in flight would turn every LDS wait of the poll into a wait for it
        }
      }
    }
    stamp(c, ST_TOP_DOWN);
    // body: mismatch and sum of P_calc at the new voltages (power_flow.py:150-168), the next injection currents
#pragma unroll
    for (int j = 0; j < GS_FLOW_ITEMS; ++j) {
      if (j < nit) {
        const int bus = recs[k0 + j].bus;
        const double en = FMSG(bus, 0), fn = FMSG(bus, 1);
        const double p = P[j], q = Q[j];
        const double pc = en * IR[j] + fn * II[j], qc = fn * IR[j] - en * II[j];     // S_calc = V_new conj(I_old)
        const double dP = p - pc, dQ = q - qc;
        lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ)));
        c.fsum += fabs(dP) + fabs(dQ);
        bad = fma(dP, 0.0, fma(dQ, 0.0, bad));
        pnew += pc;
      }
    }
    if (bad != bad) lmax = INFINITY;
    stamp(c, ST_MISMATCH);
    if (upd) psum = pnew;                                // losses at the voltages just stored
    if (it + 1 >= C.max_iterations) break;               // iteration cap: mismatch / count stay the last check's
    double sum;
    const double mm = wg_max_sum(c, lmax, c.fsum, &sum);   // full barrier
    stamp(c, ST_FLAG);
    fbs_check(st, mm, sum, it + 1, C.tolerance);
    if (__all(st.done)) break;
    ++epoch;
    // I_new = conj(S_spec / V_new) -- for the lanes that go on.  A lane that has just converged keeps the current that
    // produced its voltages: its J and therefore its V repeat bit for bit in every later sweep of the group, so the
    // slots hold every lane's final voltages whenever the group stops (the epilogue reads them there)
#pragma unroll
    for (int j = 0; j < GS_FLOW_ITEMS; ++j) {
      if (j < nit) {
        const int bus = recs[k0 + j].bus;
        const double en = FMSG(bus, 0), fn = FMSG(bus, 1);
        const double rd = 1.0 / (en * en + fn * fn);
        if (!st.done) { IR[j] = (P[j] * en + Q[j] * fn) * rd; II[j] = (P[j] * fn - Q[j] * en) * rd; }
      }
    }
    {  // backward chain: J_i = -I_i + sum J_c
      GsItemRec rn{};
      if (nit > 0) rn = load_item(T, k0);
#pragma unroll
      for (int j = 0; j < GS_FLOW_ITEMS; ++j) {
        if (j < nit) {
          const GsItemRec r = rn;
          double jr = -IR[j], ji = -II[j];
          const int nch = r.n_children;
#pragma unroll
          for (int u = 0; u < GS_ITEM_CHILDREN; ++u) {
            if (u < nch) { const int ch = r.child_slot[u]; const double2 jc = flow_take(&FMSG(ch, 0), flags + ch, epoch); jr += jc.x; ji += jc.y; }
          }
          for (int u = GS_ITEM_CHILDREN; u < nch; ++u) {
            const int ch = cld(T.ovf_slot, r.ovf0 + u - GS_ITEM_CHILDREN);
            const double2 jc = flow_take(&FMSG(ch, 0), flags + ch, epoch); jr += jc.x; ji += jc.y;
          }
          flow_give(c, &FMSG(r.bus, 0), flags + r.bus, epoch, jr, ji);
          if (j + 1 < nit) rn = load_item(T, k0 + j + 1);
        }
      }
    }
    stamp(c, ST_BOTTOM_UP);
  }
  __syncthreads();          // the epilogue reads the voltages of every bus from the slots
#undef FMSG
  return psum;
}

// Bus voltage angle from (e, f).  Distribution feeders sit within a few degrees of the slack, and libm's atan2 is
// ~105 vector instructions (the epilogue's bus loop is bound by them): when every lane of the wave has e > 0 and
// |f| <= e / 8 the angle is t - t^3/3 + t^5/5 - ... with t = f / e, whose tenth term is below 2^-63 of the first;
// any other wave takes libm.  Either way the result is the correctly rounded angle to within 2 ulp.
__device__ __forceinline__ double bus_angle(double f, double e) {
  if (!__all(e > 0.0 && fabs(f) <= 0.125 * e)) return atan2(f, e);
  const double t = f / e, z = t * t;
  double p = -1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z - 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z - 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z - 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z - 1.0 / 3.0;
  return t + t * z * p;
}

// =============================================================================================
// Epilogue: line flows (power_flow.py:340-356), losses (:198-200), wrapped angles, scalars; with
// ENV also everything of step() that follows the load flow (grid_env.py:553-617).
// =============================================================================================
// LDSV: the final (e, f) of every bus are in the per-bus LDS slots of the dataflow sweeps, not in the E rows.  A lane
// that stopped at the very first check keeps the flat start (its slots went on iterating with the rest of the group).
template <int LDSV>
__device__ __forceinline__ double2 final_ef(Ctx& c, int i, bool flat_lane, bool any_flat) {
  if (!LDSV) { const GsRows& R = c.R; GsLaneRows S = c.S; return ROW2(R.E + i); }
  const double* m = gs_dyn + ((size_t)i * 2) * GS_LANES + c.lane;
  double2 v = make_double2(m[0], m[GS_LANES]);
  if (any_flat) {
    const double fv = cld(c.T.fixed_v, i) ? cld(c.T.v_set, i) : 1.0;
    if (flat_lane) v = make_double2(fv, 0.0);
  }
  return v;
}

template <int ENV, int WRAP_VA, int CHK, int LDSV>
__device__ __forceinline__ void epilogue_impl(Ctx& c, const GsEnvCfg& E, const NrState& st, double total_load, bool have_psum,
                                         double psum, const GsFusedChecks& FC, int b) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  const bool flat_lane = LDSV && st.done && st.iters <= 1, any_flat = LDSV && __any(flat_lane);
  double lsum = have_psum ? psum : 0.0, dev = 0.0, vmax = -INFINITY, vmin = INFINITY;
  int over = 0, vflags = 0;
  // fused post-step checks (kernels_checks.hip is the stand-alone form; same arithmetic, same outputs)
  const bool chk = ENV && CHK && FC.enabled;            // CHK = 0: the plain step kernels carry none of this
  const GsChecksCfg& K = FC.C;
  double* Pv = chk ? FC.prev + (size_t)blockIdx.x * (T.n + 1) * GS_LANES + c.lane : nullptr;
  int k_nlow = 0, k_nhigh = 0, k_mhigh = 0, k_mlow = 0, k_mem = 0, k_cover = 0, k_mover = 0, k_vbad = 0, k_fbad = 0;
  double k_dv = 0.0, k_ql = -INFINITY;
  // Both loops are a handful of rows per wave, each a round trip to L2 / Infinity Cache: four items per trip, their
  // rows requested before any of the arithmetic (atan2, sqrt, divisions) starts.
  for (int i0 = c.wave; i0 < T.n; i0 += 4 * c.W) {
    double xa[4], xb[4], pcs[4], pvs[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * c.W, T.n - 1);
      const double2 x = WRAP_VA ? (double2)ROW2(R.VM + i) : final_ef<LDSV>(c, i, flat_lane, any_flat);      // (|V|, angle) or (e, f)
      xa[u] = WRAP_VA ? x.y : x.x;
      xb[u] = WRAP_VA ? x.x : x.y;
      pcs[u] = have_psum ? 0.0 : ROW(R.PC + i);
      if (chk) pvs[u] = Pv[(size_t)i * GS_LANES];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * c.W;
      if (i >= T.n) break;
      if (!have_psum) lsum += pcs[u];
      double v;
      if (WRAP_VA) {                                          // np.angle: wrap theta to (-pi, pi]
        const double va = xa[u];
        ROW(R.VA + i) = va - (2.0 * M_PI) * rint(va * (1.0 / (2.0 * M_PI)));
        v = xb[u];
      } else {                                                // sweeps work on (e, f): polar form here, once
        const double e = xa[u], f = xb[u];
        v = sqrt(e * e + f * f);
        ROW2(R.VM + i) = make_double2(v, bus_angle(f, e));
      }
      if (ENV) {                                              // reward / flags, grid_env.py:790-792, base.py:156-159
        dev += fabs(v - 1.0);
        vmax = fmax(vmax, v); vmin = fmin(vmin, v);
        vflags |= (v > E.v_max) ? 1 : 0;
        vflags |= (v < E.v_min) ? 2 : 0;
      }
      if (chk) {
        const bool cl = v < K.c_vlo, ch = !cl && v > K.c_vhi;                // safety.py:129-137 (elif)
        const bool mh = v > K.m_vhi, ml = v < K.m_vlo;                       // :333-337
        const bool em = v > K.m_evhi || v < K.m_evlo;                        // :340-341
        k_nlow += cl; k_nhigh += ch; k_mhigh += mh; k_mlow += ml; k_mem += em;
        const double d = fabs(v - pvs[u]);                                   // :168 (np.max propagates NaN)
        k_dv = (d != d || k_dv != k_dv) ? NAN : fmax(k_dv, d);
        Pv[(size_t)i * GS_LANES] = v;                                        // :181-184
        if (!(fabs(v) < INFINITY)) k_vbad = 1;                               // robust_power_flow.py:643-647
        if (FC.bus_mask) FC.bus_mask[((size_t)blockIdx.x * T.n + i) * GS_LANES + c.lane] = (uint8_t)(cl | (ch << 1) | (ml << 2) | (mh << 3) | (em << 4));
      }
    }
  }
  stamp(c, ST_EPI_BUSES);
  for (int k0 = c.wave; k0 < T.m; k0 += 4 * c.W) {
    double ei_[4], fi_[4], ej_[4], fj_[4];
    // every scalar of the four lines first (one wait for all of them: a scalar load in flight makes each later LDS /
    // scalar wait a wait for it), then the eight voltages, then the arithmetic
    int li_[4], lj_[4];
    double yr_[4], yi_[4], rt_[4], ri_[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u * c.W, T.m - 1);
      li_[u] = cld(T.lfrom, k); lj_[u] = cld(T.lto, k);
      yr_[u] = cld(T.lyr, k); yi_[u] = cld(T.lyi, k); rt_[u] = cld(T.lrating, k); ri_[u] = cld(T.lrating_inv, k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 vi = final_ef<LDSV>(c, li_[u], flat_lane, any_flat), vj = final_ef<LDSV>(c, lj_[u], flat_lane, any_flat);
      ei_[u] = vi.x; fi_[u] = vi.y; ej_[u] = vj.x; fj_[u] = vj.y;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * c.W;
      if (k >= T.m) break;
      const double yr = yr_[u], yi = yi_[u], rating = rt_[u], rinv = ri_[u];
      const double ei = ei_[u], fi = fi_[u];
      const double dr = ei - ej_[u], di = fi - fj_[u];
      const double ir = yr * dr - yi * di, ii = yr * di + yi * dr;      // I = y (Vi - Vj)
      const double sr = ei * ir + fi * ii, si = fi * ir - ei * ii;      // S = Vi conj(I)
      const double ql = (rating > 0.0) ? gs_div_by(sqrt(sr * sr + si * si), rating, rinv) : 0.0;
      ROW(R.LOAD + k) = ql;
      if (ENV) {                                                        // Line.update_state, base.py:261-264
        const double ld = (rating > 0.0) ? gs_div_by(fabs(sr), rating, rinv) : 0.0;
        ROW2(R.FLOW + k) = make_double2(sr, ld);
        over += (ld > 0.8) ? 1 : 0;
        if (chk) {
          const double cld_ = K.stride_cload == 2 ? ld : ql;             // which loading the limits apply to
          const bool co = cld_ > K.c_load, mo = cld_ > K.m_load;         // safety.py:150-154, :364-365
          k_cover += co; k_mover += mo;
          k_ql = (ql != ql || k_ql != k_ql) ? NAN : fmax(k_ql, ql);
          if (!(fabs(sr) < INFINITY)) k_fbad = 1;
          if (FC.line_mask) FC.line_mask[((size_t)blockIdx.x * T.m + k) * GS_LANES + c.lane] = (uint8_t)(co | (mo << 1));
        }
      } else {
        ROW(R.FLOW + k) = sr;
      }
    }
  }
  stamp(c, ST_EPI_LINES);
  if (LDSV) __syncthreads();                                    // the partial results overlay the slots the loops read
  double* post = gs_dyn + c.lane;                               // post[k][wave][lane]
  int* posti = (int*)(gs_dyn + GS_EPI_DOUBLES) + c.lane;        // posti[k][wave][lane]
#define POST(k, w) post[((k) * GS_MAX_WAVES + (w)) * GS_LANES]
#define POSTI(k, w) posti[((k) * GS_MAX_WAVES + (w)) * GS_LANES]
  POST(0, c.wave) = lsum;
  if (ENV) {
    POST(1, c.wave) = dev; POST(2, c.wave) = vmax; POST(3, c.wave) = vmin;
    POSTI(0, c.wave) = over; POSTI(1, c.wave) = vflags;
    if (chk) {          // counts are below 65536 (the host refuses to fuse otherwise): two per word, summed over the waves
      POSTI(2, c.wave) = k_nlow | (k_nhigh << 16); POSTI(3, c.wave) = k_mhigh | (k_mlow << 16);
      c.sh.flag[0][c.wave][c.lane] = k_mem | (k_cover << 16); c.sh.flag[1][c.wave][c.lane] = k_mover | (k_vbad << 16) | (k_fbad << 24);
      c.sh.red[0][c.wave][c.lane] = k_dv; c.sh.red[1][c.wave][c.lane] = k_ql;      // the solver's reduction arrays are free by now
    }
  }
  __syncthreads();
  stamp(c, ST_EPI_REDUCE);
  if (c.wave != 0) return;
  double losses = POST(0, 0);
  for (int w = 1; w < c.W; ++w) losses += POST(0, w);
  ROW(R.LOSSES) = losses;
  ROW(R.MAXMIS) = st.mm;
  ROW(R.ITERS) = (double)st.iters;
  ROW(R.CONV) = (double)st.conv;
  ROW(R.STATUS) = (double)st.status;
  if (!ENV) return;
  dev = POST(1, 0); vmax = POST(2, 0); vmin = POST(3, 0);
  over = POSTI(0, 0); vflags = POSTI(1, 0);
  for (int w = 1; w < c.W; ++w) {
    dev += POST(1, w);
    vmax = fmax(vmax, POST(2, w)); vmin = fmin(vmin, POST(3, w));
    over += POSTI(0, w); vflags |= POSTI(1, w);
  }
  int kA = 0, kB = 0, kC = 0, kD = 0;
  if (chk) {
    k_dv = c.sh.red[0][0][c.lane]; k_ql = c.sh.red[1][0][c.lane];
    for (int w = 0; w < c.W; ++w) {
      kA += POSTI(2, w); kB += POSTI(3, w); kC += c.sh.flag[0][w][c.lane]; kD += c.sh.flag[1][w][c.lane];
      const double dv = c.sh.red[0][w][c.lane], ql = c.sh.red[1][w][c.lane];
      k_dv = (dv != dv || k_dv != k_dv) ? NAN : fmax(k_dv, dv);
      k_ql = (ql != ql || k_ql != k_ql) ? NAN : fmax(k_ql, ql);
    }
  }
#undef POST
#undef POSTI
  const double dt = E.timestep;
  // the rows of the scalar state, requested together (each is a round trip to L2; stores in between would serialise them)
  const double totloss0 = ROW(R.TOTLOSS), f_old = ROW(R.FREQ), viol0 = ROW(R.VIOL), step0 = ROW(R.STEP), eprew0 = ROW(R.EPREW);
  double total_gen = 0.0, total_curt = 0.0;                            // grid_env.py:744-751, 807-816
  for (int g = 0; g < T.n_gens; ++g) {
    const double p = ROW(R.GENP + g);
    total_gen += p;
    total_curt += p * (1.0 - ROW(R.CURT + g));
  }
  const double totloss = totloss0 + losses * dt / 3600.0;              // grid_env.py:739
  ROW(R.TOTLOSS) = totloss;
  const double imbalance = (total_gen - total_load - losses * E.power_base) / 1e6;
  double f = f_old;                                                    // dynamics.py:260-273
  f += ((imbalance - E.D * (f - E.f0)) / (2.0 * E.H * E.f0)) * dt;
  f = fmax(55.0, fmin(65.0, f));
  ROW(R.FREQ) = f;
  double reward = 0.0;                                                 // grid_env.py:785-826
  reward -= dev * 10.0;
  reward -= fabs(f - 60.0) * 20.0;
  reward -= (double)(over * 50);
  reward -= totloss * 0.1;
  reward += (total_gen - total_curt) * 1e-5;
  for (int q = 0; q < T.n_bats; ++q) {
    const double soc = ROW(R.SOC + q);
    reward += (soc >= 0.2 && soc <= 0.8) ? 1.0 : -5.0;
  }
  const int vhigh = vflags & 1, vlow = (vflags >> 1) & 1, fhigh = f > E.f_max, flow_ = f < E.f_min;
  double viol = viol0, trunc = 0.0;
  if (vhigh | vlow | fhigh | flow_) {
    viol += 1.0;
    if (viol > 10.0) { trunc = 1.0; reward -= E.safety_penalty; }     // grid_env.py:604-606
  }
  ROW(R.VIOL) = viol;
  ROW(R.TRUNC) = trunc;
  ROW(R.TERM) = (step0 >= (double)E.episode_length) ? 1.0 : 0.0;   // base.py:140-142
  ROW(R.REWARD) = reward;
  ROW(R.EPREW) = eprew0 + reward;
  ROW(R.VMAX) = vmax; ROW(R.VMIN) = vmin;
  ROW(R.VFLAGS + 0) = (double)vhigh; ROW(R.VFLAGS + 1) = (double)vlow;
  ROW(R.VFLAGS + 2) = (double)fhigh; ROW(R.VFLAGS + 3) = (double)flow_;
  if (chk && b < (int)(FC.Bp) && Pv != nullptr) {
    // same finalisation as gs_k_checks (kernels_checks.hip), on the values of this very step
    const int c_nlow = kA & 0xffff, c_nhigh = kA >> 16, m_nhigh = kB & 0xffff, m_nlow = kB >> 16, m_nem = kC & 0xffff, c_nover = kC >> 16;
    const int m_nover = kD & 0xffff, vbad = (kD >> 16) & 0xff, fbad = kD >> 24;
    int32_t* has_prev = FC.state + b; int32_t* consec = FC.state + FC.Bp + b; int32_t* emode = FC.state + 2 * FC.Bp + b;
#define OI(k) FC.out_i[(size_t)(k) * FC.Bp + b]
#define OF(k) FC.out_f[(size_t)(k) * FC.Bp + b]
    const int c_flow = f < K.c_flo, c_fhigh = !c_flow && f > K.c_fhi;        // safety.py:140-147
    const double vrate = k_dv / K.dt;                                        // NaN stays NaN
    const double frate = fabs(f - Pv[(size_t)T.n * GS_LANES]) / K.dt;        // :174
    const int hp = *has_prev;
    const int c_vr = hp && vrate > K.c_rocv, c_fr = hp && frate > K.c_rocf;
    Pv[(size_t)T.n * GS_LANES] = f; *has_prev = 1;
    const int c_total = c_nlow + c_nhigh + c_flow + c_fhigh + c_nover + c_vr + c_fr;
    OI(GS_CI_C_NLOW) = c_nlow; OI(GS_CI_C_NHIGH) = c_nhigh; OI(GS_CI_C_FLOW) = c_flow; OI(GS_CI_C_FHIGH) = c_fhigh;
    OI(GS_CI_C_NOVER) = c_nover; OI(GS_CI_C_VRATE) = c_vr; OI(GS_CI_C_FRATE) = c_fr; OI(GS_CI_C_TOTAL) = c_total;
    OI(GS_CI_C_SEVERITY) = c_total > 5 ? 3 : (c_total > 2 ? 2 : (c_total > 0 ? 1 : 0));
    OF(GS_CF_VRATE) = vrate; OF(GS_CF_FRATE) = frate;
    const int m_fhigh = f > K.m_fhi, m_flow = !m_fhigh && f < K.m_flo, m_fem = f > K.m_efhi || f < K.m_eflo;
    const int m_total = m_nhigh + m_nlow + m_nem + m_fhigh + m_flow + m_fem + m_nover;
    const int cs = m_total > 0 ? *consec + 1 : 0;
    const int trigger = (m_nem > 0) || m_fem || cs > 5 || m_total > 10;
    const int mode = *emode | trigger;
    *consec = cs; *emode = mode;
    OI(GS_CI_M_NHIGH) = m_nhigh; OI(GS_CI_M_NLOW) = m_nlow; OI(GS_CI_M_NEMERG) = m_nem; OI(GS_CI_M_FHIGH) = m_fhigh; OI(GS_CI_M_FLOW) = m_flow;
    OI(GS_CI_M_FEMERG) = m_fem; OI(GS_CI_M_NOVER) = m_nover; OI(GS_CI_M_TOTAL) = m_total; OI(GS_CI_M_ACTION) = trigger;
    OI(GS_CI_M_CONSEC) = cs; OI(GS_CI_M_EMODE) = mode;
    double q = 1.0;                                                          // robust_power_flow.py:615-657
    if (vmin < 0.8 || vmax > 1.2) q *= 0.3;
    else if (vmin < 0.9 || vmax > 1.1) q *= 0.7;
    if (T.m > 0 && k_ql == k_ql) { if (k_ql > 2.0) q *= 0.2; else if (k_ql > 1.0) q *= 0.5; }
    if (st.mm > K.q_tol * 100.0) q *= 0.6;
    if (st.iters <= 5) q *= 1.1; else if (st.iters > 20) q *= 0.9;
    q = fmin(q, 1.0);
    if (!st.conv || vbad || fbad) q = 0.0;
    OF(GS_CF_QUALITY) = q;
#undef OI
#undef OF
  }
  stamp(c, ST_EPI_SCALARS);
}

template <int ENV, int WRAP_VA, int CHK, int LDSV>
__device__ __forceinline__ void epilogue(Ctx& c, const GsEnvCfg& E, const NrState& st, double total_load, bool have_psum,
                                         double psum, const GsFusedChecks& FC, int b) {
  epilogue_impl<ENV, WRAP_VA, CHK, LDSV>(c, E, st, total_load, have_psum, psum, FC, b);   // waves != 0 leave it after the reduction
}

// Observation block of this group, batch-major, written straight from the step kernel: 64-column
// tiles of slab rows are transposed through LDS so that both the row reads (512 B) and the
// obs[b][c0..c0+63] writes (512 B) are coalesced.  Column order: grid_env.py:753-783.
// The first pass starts while wave 0 is still in the scalar part of the epilogue: its columns (|V|, angle) were
// final at the epilogue's reduction barrier, the tiles sit behind the partials wave 0 is reading, and the other
// waves share out wave 0's rows of that pass.
__device__ __forceinline__ void pack_observations_by_column(Ctx& c, const GsPackArgs& A, int B) {
  double* tiles = gs_dyn + GS_PACK_LDS_DOUBLES;
  const int g = blockIdx.x;
  const int TG = A.tiles_per_pass;                     // 64-column tiles staged per pass (LDS permitting)
  const int span = 64 * TG;
  // the step moves only the columns that can change: position j of that list is column j, or j shifted past the
  // block of constants
  const int gap = A.skip1 - A.skip0, n_dyn = A.obs_dim - gap;
  if (!A.early_pass0) __syncthreads();                 // small networks: the first pass already holds the frequency column
  for (int c0 = 0; c0 < n_dyn; c0 += span) {
    // gather: this wave's rows of the pass, four independent loads in flight at a time
    const bool early = A.early_pass0 && c0 == 0 && c.W > 1;     // pass 0: waves 1 .. W-1 only
    const int gw = early ? c.wave - 1 : c.wave, GW = early ? c.W - 1 : c.W;
    for (int j = gw; j < span && gw >= 0; j += 4 * GW) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = j + u * GW, jj = c0 + cc;
        v[u] = 0.0;
        if (cc < span && jj < n_dyn) {
          const int s = cld(A.map, jj < A.skip0 ? jj : jj + gap);
          v[u] = (s >= 0) ? (double)c.S[(size_t)s * GS_LANES] : cld(A.cst, -s - 1);      // rows of other waves: sc0 loads, like every row load
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = j + u * GW;
        if (cc < span) tiles[(cc >> 6) * (64 * 65) + (cc & 63) * 65 + c.lane] = v[u];
      }
    }
    __syncthreads();
    for (int t = 0; t < TG; ++t) {
      const int jj = c0 + t * 64 + c.lane, col = jj < A.skip0 ? jj : jj + gap;
      for (int r = c.wave; r < GS_LANES; r += c.W) {
        const int b = g * GS_LANES + r;
        if (b < B && jj < n_dyn) A.out[(size_t)b * A.obs_dim + col] = tiles[t * (64 * 65) + c.lane * 65 + r];
      }
    }
    __syncthreads();
  }
}

// The same with two columns per lane on both sides: the observation lists (|V|, angle) per bus and (flow, loading) per
// line, which are row pairs of the slab (one 16-byte load), and an instance's columns 2L, 2L+1 are one 16-byte store.
// Needs an even obs_dim and an even block of constants (every row of `out` and every column pair 16-byte aligned).
__device__ __forceinline__ void pack_observations(Ctx& c, const GsPackArgs& A, int B) {
  if (!A.pair_ok) { pack_observations_by_column(c, A, B); return; }
  double* tiles = gs_dyn + GS_PACK_LDS_DOUBLES;
  const int g = blockIdx.x;
  const int TG = A.tiles_per_pass & ~1;                // whole 128-column spans
  const int span = 64 * TG, hspan = span >> 1;
  const int gap = A.skip1 - A.skip0, n_dyn = A.obs_dim - gap;
  if (!A.early_pass0) __syncthreads();
  for (int c0 = 0; c0 < n_dyn; c0 += span) {
    const bool early = A.early_pass0 && c0 == 0 && c.W > 1;     // pass 0: waves 1 .. W-1 only
    const int gw = early ? c.wave - 1 : c.wave, GW = early ? c.W - 1 : c.W;
    for (int j = gw; j < hspan && gw >= 0; j += 4 * GW) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = j + u * GW, jj = c0 + 2 * pp;            // columns jj, jj + 1 of the changing list
        v[u] = make_double2(0.0, 0.0);
        if (pp < hspan && jj < n_dyn) {
          const int s0 = cld(A.map, jj < A.skip0 ? jj : jj + gap);
          const int s1 = (jj + 1 < n_dyn) ? cld(A.map, jj + 1 < A.skip0 ? jj + 1 : jj + 1 + gap) : s0;
          if (s0 >= 0 && !(s0 & 1) && s1 == s0 + 1) v[u] = (double2)c.S.pair((size_t)s0 * GS_LANES);      // rows of other waves: sc0 loads
          else {
            v[u].x = (s0 >= 0) ? (double)c.S[(size_t)s0 * GS_LANES] : cld(A.cst, -s0 - 1);
            v[u].y = (s1 >= 0) ? (double)c.S[(size_t)s1 * GS_LANES] : cld(A.cst, -s1 - 1);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = 2 * (j + u * GW);
        if (cc < span) {
          tiles[(cc >> 6) * (64 * 65) + (cc & 63) * 65 + c.lane] = v[u].x;
          tiles[(cc >> 6) * (64 * 65) + ((cc & 63) + 1) * 65 + c.lane] = v[u].y;
        }
      }
    }
    __syncthreads();
    for (int t = 0; t < TG; t += 2) {                  // a 128-column span: lane L carries columns 2L and 2L + 1
      const int cc = t * 64 + 2 * c.lane, jj = c0 + cc;
      const int col0 = jj < A.skip0 ? jj : jj + gap, col1 = jj + 1 < A.skip0 ? jj + 1 : jj + 1 + gap;
      const double* tp = tiles + (cc >> 6) * (64 * 65) + (cc & 63) * 65;
      for (int r = c.wave; r < GS_LANES; r += c.W) {
        const int b = g * GS_LANES + r;
        if (b >= B || jj >= n_dyn) continue;
        double* o = A.out + (size_t)b * A.obs_dim;
        if (jj + 1 < n_dyn && col1 == col0 + 1) *(double2*)(o + col0) = make_double2(tp[r], tp[65 + r]);
        else { o[col0] = tp[r]; if (jj + 1 < n_dyn) o[col1] = tp[65 + r]; }
      }
    }
    __syncthreads();
  }
}

// Everything of step() that precedes the load flow, spread over the W waves (grid_env.py:433-477).
// FLAT_FBS: the injection pass also writes the sweep solver's flat start (e, f) of the buses it visits.
// FLOW_REGS: the wave's injection records are its solver items in item order (dataflow kernel), and P goes into the
// solver's registers Pout[0 .. 7] instead of the P rows: no store, no reload, no barrier between the two.
// FLAT_NR: ... or the Newton solvers' flat start (|V|, angle, e, f, 1 / |V|: power_flow.py:103, 128-136)
template <bool FLAT_FBS, bool FLOW_REGS, bool FLAT_NR>
__device__ __forceinline__ void prologue_env(Ctx& c, const GsEnvCfg& E, const double* __restrict__ actions, int b, bool valid, double* Pout) {
  const GsTables& T = c.T; const GsRows& R = c.R; GsLaneRows S = c.S;
  const uint64_t inst = (uint64_t)(E.first_instance + b);
  // every wave derives the new clock from the old rows; then wave 0 applies the actions (batteries, curtailment) and
  // advances the clock rows, wave 1 draws the weather and evaluates the renewables, and the other waves draw the load
  // powers -- three chains that do not depend on each other (the scalar chain of one wave was the longest of them)
  const double told = ROW(R.TIME), tnew = told + E.timestep;
  const uint32_t snew = (uint32_t)(ROW(R.STEP) + 1.0);
  const uint64_t seed = lane_seed(S, R);
  __syncthreads();
  const int wave_b = c.W > 1 ? 1 : 0;               // who takes the weather chain
  const int rng0 = c.W > 2 ? 2 : 0;                 // first wave of the load draws (every wave when there are only one or two)
  if (c.wave == 0 && valid) env_actions_clock(T, R, E, S, actions + (size_t)b * (T.n_bats + T.n_gens));
  if (c.wave == wave_b) {
    const GsWeather wx = weather_step(R, E, S, inst, tnew, snew, valid);
    const double elev = solar_elevation(valid ? tnew : told);
    for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power_w(T, g, elev, wx);
  }
  if (c.wave >= rng0) {
    const int l0 = c.wave - rng0, ls = c.W - rng0;
    const double prof = E.stochastic_loads ? daily_profile(tnew) : 1.0;
    if (E.stochastic_loads) {                               // loads 4p .. 4p + 3 share one Philox call (two Box-Muller pairs)
      for (int p = l0; 4 * p < T.n_loads; p += ls) {
        double z[4];
        rng_normal_quad(seed, inst, snew, DRAW_LOAD0 + p, z);
        const int l = 4 * p;                                // LOADP starts on an even row: (l, l + 1) and (l + 2, l + 3) are row pairs
        const double lp0 = load_power_z(T, l, z[0], prof);
        if (l + 1 < T.n_loads) ROW2(R.LOADP + l) = make_double2(lp0, load_power_z(T, l + 1, z[1], prof));
        else ROW(R.LOADP + l) = lp0;
        if (l + 2 < T.n_loads) {
          const double lp2 = load_power_z(T, l + 2, z[2], prof);
          if (l + 3 < T.n_loads) ROW2(R.LOADP + l + 2) = make_double2(lp2, load_power_z(T, l + 3, z[3], prof));
          else ROW(R.LOADP + l + 2) = lp2;
        }
      }
    } else {
      for (int l = l0; l < T.n_loads; l += ls) ROW(R.LOADP + l) = cld(T.load_base, l);
    }
  }
  stamp(c, ST_PRO_SPARE);          // this wave's own share; ST_PRO_SCALAR is then the wait for the slowest wave
  __syncthreads();
  stamp(c, ST_PRO_SCALAR);
  {
    const GS_CONST GsInjRec* recs = (const GS_CONST GsInjRec*)T.winj;
    const int k1 = cld(T.wi_ptr, c.wave + 1);
    // four buses per trip, their load rows requested before any division or store: a bus is otherwise one
    // round trip to L2 after another (most buses carry one load and nothing else)
    const int kbase = cld(T.wi_ptr, c.wave);
    // NB buses per trip, every row they need requested before any division or store: a bus is otherwise one round
    // trip to L2 after another (most buses carry one load and nothing else; a generator or battery adds rows)
    auto bus_batch = [&](auto nb_tag, const int k0, const int slot0) {
      constexpr int NB = decltype(nb_tag)::value;
      double lp0[NB], lp1[NB], gp0[NB], gc0[NB], bp0[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int k = min(k0 + u, k1 - 1);
        lp0[u] = ROW(R.LOADP + (recs[k].nl > 0 ? recs[k].l0 : 0));
        lp1[u] = recs[k].nl > 1 ? (double)ROW(R.LOADP + recs[k].l1) : 0.0;      // rare: most buses carry one load
        gp0[u] = 0.0; gc0[u] = 0.0; bp0[u] = 0.0;
        if (recs[k].ng > 0) { gp0[u] = ROW(R.GENP + recs[k].g0); gc0[u] = ROW(R.CURT + recs[k].g0); }
        if (recs[k].nb > 0) bp0[u] = ROW(R.BATP + recs[k].b0);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int k = k0 + u;
        if (k >= k1) break;
        const int i = recs[k].bus;
        if (FLAT_FBS && !E.fbs_warm_start) ROW2(R.E + i) = make_double2(cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0, 0.0);
        if (FLAT_NR) {
          const double vm = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
          ROW2(R.VM + i) = make_double2(vm, 0.0);
          ROW2(R.E + i) = make_double2(vm, 0.0);
          ROW(R.RVM + i) = 1.0 / vm;
        }
        if (recs[k].generic) {
          bus_injection(T, R, E, S, i);
          if (FLOW_REGS) Pout[slot0 + u] = ROW(R.P + i);
          continue;
        }
        // same accumulation order as bus_injection: loads, then generators, then batteries
        double ls = 0.0, gs = 0.0;
        if (recs[k].nl > 0) ls += lp0[u];
        if (recs[k].nl > 1) ls += lp1[u];
        if (recs[k].ng > 0) gs += gp0[u] * gc0[u];
        if (recs[k].ng > 1) gs += ROW(R.GENP + recs[k].g1) * ROW(R.CURT + recs[k].g1);
        if (recs[k].nb > 0) { const double bp = bp0[u]; if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp); }
        if (recs[k].nb > 1) { const double bp = ROW(R.BATP + recs[k].b1); if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp); }
        const double pinj = (0.0 - gs_div_by(ls, E.power_base, E.inv_power_base)) + gs_div_by(gs, E.power_base, E.inv_power_base);
        if (FLOW_REGS) Pout[slot0 + u] = pinj;
        else ROW2(R.P + i) = make_double2(pinj, 0.0);
      }
    };
    if (FLOW_REGS) {
      // At most 8 buses per wave, nearly all of them one or two loads and nothing else: a straight-line pass -- every
      // record, then every load row (unconditionally: an absent load re-reads row l0 and is masked out), then the
      // arithmetic; the few buses that carry a generator, a battery or more than two loads go through the general
      // routine afterwards.  Same accumulation order as bus_injection (x + 0.0 = x for the load powers, which are >= 0).
      int nl[8], l0[8], l1[8], special[8], bus[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const GS_CONST GsInjRec* q = recs + min(kbase + u, k1 - 1);
        nl[u] = q->nl; l0[u] = nl[u] > 0 ? q->l0 : 0; l1[u] = nl[u] > 1 ? q->l1 : l0[u];
        special[u] = q->generic | q->ng | q->nb; bus[u] = q->bus;
      }
      double lp0[8], lp1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { lp0[u] = ROW(R.LOADP + l0[u]); lp1[u] = ROW(R.LOADP + l1[u]); }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double ls = (nl[u] > 0 ? lp0[u] : 0.0) + (nl[u] > 1 ? lp1[u] : 0.0);
        Pout[u] = (0.0 - gs_div_by(ls, E.power_base, E.inv_power_base)) + gs_div_by(0.0, E.power_base, E.inv_power_base);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kbase + u < k1 && special[u]) { bus_injection(T, R, E, S, bus[u]); Pout[u] = ROW(R.P + bus[u]); }
      }
    } else {
      for (int k0 = kbase; k0 < k1; k0 += 4) bus_batch(std::integral_constant<int, 4>{}, k0, 0);
    }
  }
  if (!FLOW_REGS) __syncthreads();          // FLOW_REGS: nothing of this pass is read by another wave
}

// PHASE 0: the whole step / solve in one launch.  PHASE 1 / 2: what comes before / after the load flow, for a solver that is
// its own kernel (kernels_dense.hip: one workgroup per instance around MFMA tiles; it leaves |V| / angle, (e, f), P / Q
// calculated and the convergence record in the rows, exactly where newton_loop leaves them).
template <int KIND, int ENV, int CHK, int PHASE = 0>
__device__ __forceinline__ void main_body(const GsTables& T, const GsRows& R, const GsSolveCfg& C, const GsEnvCfg& E,
                                          double* __restrict__ slab, int B, const double* __restrict__ actions,
                                          double total_load, const GsPackArgs& PA, const GsFusedChecks& FC) {
  __shared__ GsShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  const int b = blockIdx.x * GS_LANES + lane;
  const bool valid = b < B;
  Ctx c{T, R, S, sh, lane, wave, W, C.stamps, 0ull, C.stamp_wave};
  if (C.stamps) c.tlast = __builtin_readcyclecounter();
  if (KIND == KIND_FBS_FLOW) {     // dataflow sweeps: flags at epoch 0 = nothing posted; a bus that is nobody's item (the slack, an
    // islanded bus) keeps its flat-start voltage in its slot for good.  The prologue / the solver's first barrier publish both.
    int* flags = (int*)(gs_dyn + (size_t)T.n * 2 * GS_LANES);
    for (int i = threadIdx.x; i < T.n; i += blockDim.x) flags[i] = 0;
    for (int i = wave; i < T.n; i += W)
      if (cld(T.lvl_pos, i) < 0) {
        gs_dyn[((size_t)i * 2) * GS_LANES + lane] = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
        gs_dyn[((size_t)i * 2 + 1) * GS_LANES + lane] = 0.0;
      }
    if (!ENV) __syncthreads();
  }
  double Pinj[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // KIND_FBS_FLOW + ENV: S_spec of the wave's own buses
  constexpr bool kNewton = KIND == KIND_TREE || KIND == KIND_TREE_LDS || KIND == KIND_LU || KIND == KIND_DENSE;
  if (ENV && PHASE != 2) prologue_env<KIND == KIND_FBS_LDS, KIND == KIND_FBS_FLOW, kNewton>(c, E, actions, b, valid, Pinj);
  stamp(c, ST_PROLOGUE);
  if (PHASE == 1) return;
  NrState st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  double psum = 0.0;
  if (PHASE == 2) {          // the convergence record the solver kernel left
    st.mm = ROW(R.MAXMIS); st.iters = (int)ROW(R.ITERS); st.conv = ROW(R.CONV) != 0.0 ? 1 : 0; st.status = (int)ROW(R.STATUS); st.done = true;
  } else
  if (KIND == KIND_FBS) fbs_loop(c, C, st);
  else if (KIND == KIND_FBS_LDS) psum = fbs_loop_lds<ENV != 0>(c, C, st);
  else if (KIND == KIND_FBS_FLOW) psum = fbs_loop_flow<ENV != 0>(c, C, st, Pinj);
  else newton_loop<KIND, ENV != 0>(c, C, st);
  constexpr bool kFbs = KIND == KIND_FBS || KIND == KIND_FBS_LDS || KIND == KIND_FBS_FLOW;
  epilogue<ENV, !kFbs, CHK, KIND == KIND_FBS_FLOW>(c, E, st, total_load, KIND == KIND_FBS_LDS || KIND == KIND_FBS_FLOW, psum, FC, valid ? b : 0x7fffffff);   // FBS keeps no polar angle: atan2 there
  if (ENV && PA.out != nullptr) pack_observations(c, PA, B);     // rows of pass 0 visible since the epilogue's barrier
  stamp(c, ST_EPILOGUE);
}

// The kernels read their arguments where they use them, through a pointer to the argument block the compiler cannot see
// through (kernels_flow2.hip F2_ARGS_IN_PLACE says why: taken from the formal parameters every scalar word is loaded at the
// top and parked in vector lanes -- 250 to 340 spilled scalar registers in the step kernels of this file).
struct GsStepArgBlock { GsTables T; GsRows R; GsSolveCfg C; GsEnvCfg E; double* slab; int B; const double* actions; double total_load;
                        GsPackArgs PA; GsFusedChecks FC; };
struct GsSolveArgBlock { GsTables T; GsRows R; GsSolveCfg C; double* slab; int B; };
#define GS_ARGS_IN_PLACE(Block)                                                                               \
  const __attribute__((address_space(4))) char* ka_ = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(ka_));                                                                               \
  const Block* A = (const Block*)ka_
#define GS_DEFINE_KERNELS(name, KIND)                                                                         \
  extern "C" __global__ void __launch_bounds__(1024)                                                          \
  gs_k_##name(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {                         \
    GsEnvCfg E{};                                                                                             \
    GsPackArgs PA{};                                                                                          \
    GsFusedChecks FC{};                                                                                       \
    GS_ARGS_IN_PLACE(GsSolveArgBlock);                                                                        \
    main_body<KIND, 0, 0>(A->T, A->R, A->C, E, A->slab, A->B, nullptr, 0.0, PA, FC);                          \
  }                                                                                                           \
  extern "C" __global__ void __launch_bounds__(1024)                                                          \
  gs_k_step_##name(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,          \
                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC) {  \
    GS_ARGS_IN_PLACE(GsStepArgBlock);                                                                         \
    main_body<KIND, 1, 0>(A->T, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC);    \
  }                                                                                                           \
  extern "C" __global__ void __launch_bounds__(1024)   /* the step with the post-step checks in its epilogue */ \
  gs_k_stepc_##name(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,         \
                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC) { \
    GS_ARGS_IN_PLACE(GsStepArgBlock);                                                                         \
    main_body<KIND, 1, 1>(A->T, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC);    \
  }

// the two halves of a step / solve around the dense MFMA Newton-Raphson kernel (kernels_dense.hip)
extern "C" __global__ void __launch_bounds__(1024)
gs_k_pre_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                  const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC) {
  GS_ARGS_IN_PLACE(GsStepArgBlock);
  main_body<KIND_LU, 1, 0, 1>(A->T, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC);
}
extern "C" __global__ void __launch_bounds__(1024)
gs_k_post_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC) {
  GS_ARGS_IN_PLACE(GsStepArgBlock);
  main_body<KIND_LU, 1, 0, 2>(A->T, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC);
}
extern "C" __global__ void __launch_bounds__(1024)
gs_k_postc_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC) {
  GS_ARGS_IN_PLACE(GsStepArgBlock);
  main_body<KIND_LU, 1, 1, 2>(A->T, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC);
}
extern "C" __global__ void __launch_bounds__(1024)       // solver-only API (gs_solve): line flows, losses, wrapped angles
gs_k_posts_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {
  GsEnvCfg E{}; GsPackArgs PA{}; GsFusedChecks FC{};
  GS_ARGS_IN_PLACE(GsSolveArgBlock);
  main_body<KIND_LU, 0, 0, 2>(A->T, A->R, A->C, E, A->slab, A->B, nullptr, 0.0, PA, FC);
}

GS_DEFINE_KERNELS(nr_tree, KIND_TREE)
GS_DEFINE_KERNELS(nr_tree_lds, KIND_TREE_LDS)
GS_DEFINE_KERNELS(nr_lu, KIND_LU)
GS_DEFINE_KERNELS(nr_dense, KIND_DENSE)
GS_DEFINE_KERNELS(fbs, KIND_FBS)
GS_DEFINE_KERNELS(fbs_lds, KIND_FBS_LDS)
GS_DEFINE_KERNELS(fbs_flow, KIND_FBS_FLOW)
