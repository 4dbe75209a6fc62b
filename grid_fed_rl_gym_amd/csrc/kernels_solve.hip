// kernels_solve.hip -- batched AC load-flow kernels for gfx950 (MI355X, wave64).
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
// The linear solve (power_flow.py:187, LAPACK dgesv on the dense Jacobian) is replaced by a
// 2x2-block elimination on the Jacobian's own sparsity: a level-scheduled forest sweep for
// radial feeders, a statically scheduled block LU (host-side minimum-degree symbolic
// factorisation) for meshed ones.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]

template <typename X>
__device__ __forceinline__ X cld(const X* p, int i) {
  return ((const GS_CONST X*)p)[i];
}

__device__ __forceinline__ double finite_or_inf(double v) { return (fabs(v) < INFINITY) ? v : INFINITY; }

// LDS scratch for cross-wave reductions, double-buffered by a phase parity so that a fast
// wave can enter the next reduction before a slow one has finished reading the previous one.
struct GsShared {
  double red[2][GS_MAX_WAVES][GS_LANES];
  int flag[2][GS_MAX_WAVES][GS_LANES];
};

__device__ __forceinline__ double wg_max(GsShared& sh, int par, int wave, int W, int lane, double v) {
  sh.red[par][wave][lane] = v;
  __syncthreads();
  double r = sh.red[par][0][lane];
  for (int w = 1; w < W; ++w) r = fmax(r, sh.red[par][w][lane]);
  return r;
}

__device__ __forceinline__ double wg_sum(GsShared& sh, int par, int wave, int W, int lane, double v) {
  sh.red[par][wave][lane] = v;
  __syncthreads();
  double r = sh.red[par][0][lane];
  for (int w = 1; w < W; ++w) r += sh.red[par][w][lane];
  return r;
}

__device__ __forceinline__ int wg_or(GsShared& sh, int par, int wave, int W, int lane, int v) {
  sh.flag[par][wave][lane] = v;
  __syncthreads();
  int r = sh.flag[par][0][lane];
  for (int w = 1; w < W; ++w) r |= sh.flag[par][w][lane];
  return r;
}

// ---- flat start (power_flow.py:103, 128-136) ---------------------------------------------
__device__ __forceinline__ void flat_start(const GsTables& T, const GsRows& R, double* S, int wave, int W) {
  for (int i = wave; i < T.n; i += W) {
    ROW(R.VM + i) = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
    ROW(R.VA + i) = 0.0;
  }
}

// ---- polar -> rectangular ------------------------------------------------------------------
__device__ __forceinline__ void to_rect(const GsTables& T, const GsRows& R, double* S, int wave, int W) {
  for (int i = wave; i < T.n; i += W) {
    const double vm = ROW(R.VM + i), va = ROW(R.VA + i);
    double s, c;
    sincos(va, &s, &c);
    ROW(R.E + i) = vm * c;
    ROW(R.F + i) = vm * s;
  }
}

// ---- S = V conj(Y V) by CSR rows; dP, dQ; returns this wave's max |mismatch| (inf if non-finite)
// With a_ij = e_i e_j + f_i f_j = Vi Vj cos(th_i - th_j), b_ij = f_i e_j - e_i f_j = Vi Vj sin(..):
//   P_i = sum_j G_ij a_ij + B_ij b_ij,   Q_i = sum_j G_ij b_ij - B_ij a_ij.
__device__ __forceinline__ double mismatch_rows(const GsTables& T, const GsRows& R, double* S, int wave, int W) {
  double lmax = 0.0;
  for (int i = wave; i < T.n; i += W) {
    const double ei = ROW(R.E + i), fi = ROW(R.F + i);
    double P = 0.0, Q = 0.0;
    const int p0 = cld(T.row_ptr, i), p1 = cld(T.row_ptr, i + 1);
    for (int p = p0; p < p1; ++p) {
      const int j = cld(T.col, p);
      const double g = cld(T.G, p), b = cld(T.Bv, p);
      const double ej = ROW(R.E + j), fj = ROW(R.F + j);
      const double a = ei * ej + fi * fj;
      const double bb = fi * ej - ei * fj;
      P += g * a + b * bb;
      Q += g * bb - b * a;
    }
    ROW(R.PC + i) = P;
    ROW(R.QC + i) = Q;
    const double dP = cld(T.th_free, i) ? (ROW(R.P + i) - P) : 0.0;
    const double dQ = cld(T.vm_free, i) ? (ROW(R.Q + i) - Q) : 0.0;
    ROW(R.R0 + i) = dP;
    ROW(R.R1 + i) = dQ;
    lmax = fmax(lmax, fmax(finite_or_inf(fabs(dP)), finite_or_inf(fabs(dQ))));
  }
  return lmax;
}

// ---- diagonal Jacobian block of bus i (power_flow.py:247-248, 259-260, 270-271, 283-284) ----
struct Blk { double a00, a01, a10, a11; };

__device__ __forceinline__ Blk diag_block(const GsTables& T, const GsRows& R, double* S, int i, int exact) {
  const double vm = ROW(R.VM + i), P = ROW(R.PC + i), Q = ROW(R.QC + i);
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

// off-diagonal Jacobian block (row bus i, column bus j) (power_flow.py:251, 263, 274, 287)
__device__ __forceinline__ Blk offdiag_block(const GsTables& T, const GsRows& R, double* S, int i, int j, double g, double b) {
  const double ei = ROW(R.E + i), fi = ROW(R.F + i), ej = ROW(R.E + j), fj = ROW(R.F + j);
  const double vmj = ROW(R.VM + j);
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

__device__ __forceinline__ Blk load_blk(double* S, int row) {
  Blk b; b.a00 = ROW(row); b.a01 = ROW(row + 1); b.a10 = ROW(row + 2); b.a11 = ROW(row + 3); return b;
}
__device__ __forceinline__ void store_blk(double* S, int row, const Blk& b) {
  ROW(row) = b.a00; ROW(row + 1) = b.a01; ROW(row + 2) = b.a10; ROW(row + 3) = b.a11;
}

// ---- apply the Newton step to bus i (power_flow.py:315-327); keeps Vm >= 0 like the
// reference's abs/angle round trip does
__device__ __forceinline__ void apply_step(const GsTables& T, const GsRows& R, double* S, int i, double alpha, bool upd) {
  if (!upd) return;
  double vm = ROW(R.VM + i), va = ROW(R.VA + i);
  if (cld(T.th_free, i)) va += alpha * ROW(R.X0 + i);
  if (cld(T.vm_free, i)) vm += alpha * ROW(R.X1 + i);
  if (vm < 0.0) { vm = -vm; va += M_PI; }
  ROW(R.VM + i) = vm;
  ROW(R.VA + i) = va;
}

// ---- final: line flows (power_flow.py:340-356), losses (:198-200), wrapped angles, scalars ---
__device__ __forceinline__ void finish(const GsTables& T, const GsRows& R, double* S, GsShared& sh, int wave, int W,
                                       int lane, bool recompute, double mm, int iters, int conv, int status) {
  if (recompute) {
    __syncthreads();
    to_rect(T, R, S, wave, W);
    __syncthreads();
    (void)mismatch_rows(T, R, S, wave, W);
  }
  __syncthreads();
  double lsum = 0.0;
  for (int i = wave; i < T.n; i += W) lsum += ROW(R.PC + i);
  for (int k = wave; k < T.m; k += W) {
    const int i = cld(T.lfrom, k), j = cld(T.lto, k);
    const double yr = cld(T.lyr, k), yi = cld(T.lyi, k), rating = cld(T.lrating, k);
    const double ei = ROW(R.E + i), fi = ROW(R.F + i);
    const double dr = ei - ROW(R.E + j), di = fi - ROW(R.F + j);
    const double ir = yr * dr - yi * di, ii = yr * di + yi * dr;      // I = y (Vi - Vj)
    const double sr = ei * ir + fi * ii, si = fi * ir - ei * ii;      // S = Vi conj(I)
    ROW(R.FLOW + k) = sr;
    ROW(R.LOAD + k) = (rating > 0.0) ? hypot(sr, si) / rating : 0.0;
  }
  const double losses = wg_sum(sh, 0, wave, W, lane, lsum);
  // np.angle of V = Vm exp(j theta): wrap to (-pi, pi]
  for (int i = wave; i < T.n; i += W) ROW(R.VA + i) = atan2(ROW(R.F + i), ROW(R.E + i));
  if (wave == 0) {
    ROW(R.LOSSES) = losses;
    ROW(R.MAXMIS) = mm;
    ROW(R.ITERS) = (double)iters;
    ROW(R.CONV) = (double)conv;
    ROW(R.STATUS) = (double)status;
  }
}

// per-lane Newton bookkeeping shared by the two NR kernels (power_flow.py:148, 168-171, 204)
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

// =============================================================================================
// Newton-Raphson, radial (forest) Jacobian: level-scheduled 2x2-block elimination, zero fill.
//   bottom-up:  D_i = J_ii - sum_children C_c ;  r_i = rhs_i - sum_children q_c
//               T_i = D_i^-1 J_ip ; s_i = D_i^-1 r_i ; C_i = J_pi T_i ; q_i = J_pi s_i
//   top-down:   x_i = s_i - T_i x_p
// =============================================================================================
extern "C" __global__ void __launch_bounds__(1024)
gs_k_nr_tree(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {
  __shared__ GsShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  const bool valid = (int)(blockIdx.x * GS_LANES + lane) < B;

  NrState st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  flat_start(T, R, S, wave, W);
  __syncthreads();
  bool stale = true;   // E/F/PC describe an older V than VM/VA
  int it = 0;
  for (; it < C.max_iterations; ++it) {
    to_rect(T, R, S, wave, W);
    __syncthreads();
    const double lmax = mismatch_rows(T, R, S, wave, W);
    const double mm = wg_max(sh, it & 1, wave, W, lane, lmax);
    nr_check(st, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;

    // ---- bottom-up elimination ----
    int sing = 0;
    for (int lv = 0; lv < T.n_levels; ++lv) {
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + wave; t < t1; t += W) {
        const int i = cld(T.lvl_bus, t);
        Blk d = diag_block(T, R, S, i, C.jacobian_exact);
        double r0 = ROW(R.R0 + i), r1 = ROW(R.R1 + i);
        const int c1 = cld(T.child_ptr, i + 1);
        for (int cp = cld(T.child_ptr, i); cp < c1; ++cp) {
          const int c = cld(T.child_idx, cp);
          const Blk cb = load_blk(S, R.CB + 4 * c);
          d.a00 -= cb.a00; d.a01 -= cb.a01; d.a10 -= cb.a10; d.a11 -= cb.a11;
          r0 -= ROW(R.QV + 2 * c); r1 -= ROW(R.QV + 2 * c + 1);
        }
        const Blk inv = inv2(d, &sing);
        const double s0 = inv.a00 * r0 + inv.a01 * r1, s1 = inv.a10 * r0 + inv.a11 * r1;
        ROW(R.SV + 2 * i) = s0; ROW(R.SV + 2 * i + 1) = s1;
        const int p = cld(T.parent, i);
        if (p >= 0) {
          const int pp = cld(T.parent_pos, i);
          const double g = cld(T.G, pp), b = cld(T.Bv, pp);
          const Blk u = offdiag_block(T, R, S, i, p, g, b);    // J(i, p)
          const Blk l = offdiag_block(T, R, S, p, i, g, b);    // J(p, i); Ybus is symmetric
          const Blk tb = mul(inv, u);
          store_blk(S, R.TB + 4 * i, tb);
          store_blk(S, R.CB + 4 * i, mul(l, tb));
          ROW(R.QV + 2 * i) = l.a00 * s0 + l.a01 * s1;
          ROW(R.QV + 2 * i + 1) = l.a10 * s0 + l.a11 * s1;
        }
      }
      __syncthreads();
    }
    const int sing_all = wg_or(sh, it & 1, wave, W, lane, sing);
    if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
    const bool upd = !st.done;

    // ---- top-down substitution + voltage update ----
    for (int lv = T.n_levels - 1; lv >= 0; --lv) {
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + wave; t < t1; t += W) {
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
        apply_step(T, R, S, i, C.alpha, upd);
      }
      __syncthreads();
    }
    stale = true;
  }
  finish(T, R, S, sh, wave, W, lane, stale, st.mm, st.iters, st.conv, st.status);
}

// =============================================================================================
// Newton-Raphson, general (meshed) Jacobian: statically scheduled 2x2-block sparse LU.
// The host ordered the active buses by minimum degree and listed, for every pivot, its
// remaining neighbours and every (i, j) block its elimination touches; fill blocks own slots.
// Waves split the pair updates of a pivot; pivots are sequential (one barrier each).
// =============================================================================================
extern "C" __global__ void __launch_bounds__(1024)
gs_k_nr_lu(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {
  __shared__ GsShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  const bool valid = (int)(blockIdx.x * GS_LANES + lane) < B;

  NrState st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  flat_start(T, R, S, wave, W);
  __syncthreads();
  bool stale = true;
  int it = 0;
  for (; it < C.max_iterations; ++it) {
    to_rect(T, R, S, wave, W);
    __syncthreads();
    const double lmax = mismatch_rows(T, R, S, wave, W);
    const double mm = wg_max(sh, it & 1, wave, W, lane, lmax);
    nr_check(st, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;

    // ---- assemble: diagonal blocks, original off-diagonal blocks, zero the fill slots ----
    for (int t = wave; t < T.lu_n_piv; t += W) {
      const int i = cld(T.lu_piv_bus, t);
      store_blk(S, R.LUD + 4 * i, diag_block(T, R, S, i, C.jacobian_exact));
    }
    for (int q = wave; q < T.lu_n_orig; q += W) {
      const int pos = cld(T.lu_orig_pos, q);
      store_blk(S, R.LU + 4 * cld(T.lu_orig_slot, q),
                offdiag_block(T, R, S, cld(T.lu_orig_i, q), cld(T.lu_orig_j, q), cld(T.G, pos), cld(T.Bv, pos)));
    }
    for (int s = T.lu_n_orig + wave; s < T.lu_n_slots; s += W) {
      Blk z; z.a00 = z.a01 = z.a10 = z.a11 = 0.0;
      store_blk(S, R.LU + 4 * s, z);
    }
    __syncthreads();

    // ---- right-looking elimination ----
    int sing = 0;
    for (int t = 0; t < T.lu_n_piv; ++t) {
      const int k = cld(T.lu_piv_bus, t);
      const Blk inv = inv2(load_blk(S, R.LUD + 4 * k), &sing);
      const double rk0 = ROW(R.R0 + k), rk1 = ROW(R.R1 + k);
      const double s0 = inv.a00 * rk0 + inv.a01 * rk1, s1 = inv.a10 * rk0 + inv.a11 * rk1;
      const int q1 = cld(T.lu_pair_ptr, t + 1);
      for (int q = cld(T.lu_pair_ptr, t) + wave; q < q1; q += W) {
        const Blk aik = load_blk(S, R.LU + 4 * cld(T.lu_pair_ik, q));
        const Blk akj = load_blk(S, R.LU + 4 * cld(T.lu_pair_kj, q));
        const Blk upd = mul(mul(aik, inv), akj);
        const int tgt = cld(T.lu_pair_ij, q);
        const int row = (tgt >= 0) ? (R.LU + 4 * tgt) : (R.LUD + 4 * (-tgt - 1));
        Blk a = load_blk(S, row);
        a.a00 -= upd.a00; a.a01 -= upd.a01; a.a10 -= upd.a10; a.a11 -= upd.a11;
        store_blk(S, row, a);
      }
      const int n1 = cld(T.lu_nb_ptr, t + 1);
      for (int q = cld(T.lu_nb_ptr, t) + wave; q < n1; q += W) {
        const int i = cld(T.lu_nb_bus, q);
        const Blk aik = load_blk(S, R.LU + 4 * cld(T.lu_nb_jk, q));
        ROW(R.R0 + i) -= aik.a00 * s0 + aik.a01 * s1;
        ROW(R.R1 + i) -= aik.a10 * s0 + aik.a11 * s1;
      }
      __syncthreads();
    }
    const int sing_all = wg_or(sh, it & 1, wave, W, lane, sing);
    if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
    const bool upd = !st.done;

    // ---- back substitution (sequential in reverse pivot order; wave 0) + update ----
    if (wave == 0) {
      for (int t = T.lu_n_piv - 1; t >= 0; --t) {
        const int k = cld(T.lu_piv_bus, t);
        int dummy = 0;
        const Blk inv = inv2(load_blk(S, R.LUD + 4 * k), &dummy);
        double r0 = ROW(R.R0 + k), r1 = ROW(R.R1 + k);
        const int n1 = cld(T.lu_nb_ptr, t + 1);
        for (int q = cld(T.lu_nb_ptr, t); q < n1; ++q) {
          const int j = cld(T.lu_nb_bus, q);
          const Blk akj = load_blk(S, R.LU + 4 * cld(T.lu_nb_kj, q));
          const double xj0 = ROW(R.X0 + j), xj1 = ROW(R.X1 + j);
          r0 -= akj.a00 * xj0 + akj.a01 * xj1;
          r1 -= akj.a10 * xj0 + akj.a11 * xj1;
        }
        ROW(R.X0 + k) = inv.a00 * r0 + inv.a01 * r1;
        ROW(R.X1 + k) = inv.a10 * r0 + inv.a11 * r1;
        apply_step(T, R, S, k, C.alpha, upd);
      }
    }
    __syncthreads();
    stale = true;
  }
  finish(T, R, S, sh, wave, W, lane, stale, st.mm, st.iters, st.conv, st.status);
}

// =============================================================================================
// Forward/backward sweep on a radial feeder (constant-power buses).  New functionality: the
// reference names DistributionPowerFlow in README.md:187-197 but ships no implementation.
// Same convergence test as Newton (power mismatch < tolerance) so a converged answer satisfies
// the reference's own acceptance criterion.  State is rectangular; no trigonometry in the loop.
//   backward:  J_i = -conj(S_i / V_i) + sum_children J_c        (branch current parent -> i)
//   forward:   V_i = V_parent - J_i / y_i
// =============================================================================================
extern "C" __global__ void __launch_bounds__(1024)
gs_k_fbs(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {
  __shared__ GsShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  const bool valid = (int)(blockIdx.x * GS_LANES + lane) < B;

  NrState st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  for (int i = wave; i < T.n; i += W) {
    ROW(R.E + i) = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0;
    ROW(R.F + i) = 0.0;
  }
  __syncthreads();
  bool stale = true;
  int it = 0;
  for (; it < C.max_iterations; ++it) {
    const double lmax = mismatch_rows(T, R, S, wave, W);
    const double mm = wg_max(sh, it & 1, wave, W, lane, lmax);
    nr_check(st, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;
    const bool upd = !st.done;
    // backward sweep, deepest level first
    for (int lv = 0; lv < T.n_levels; ++lv) {
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + wave; t < t1; t += W) {
        const int i = cld(T.lvl_bus, t);
        const double e = ROW(R.E + i), f = ROW(R.F + i), p = ROW(R.P + i), q = ROW(R.Q + i);
        const double d = e * e + f * f;
        // conj((p + jq) / (e + jf)) = ((p e + q f) - j (q e - p f)) / d
        double jr = -(p * e + q * f) / d, ji = (q * e - p * f) / d;
        const int c1 = cld(T.child_ptr, i + 1);
        for (int cp = cld(T.child_ptr, i); cp < c1; ++cp) {
          const int c = cld(T.child_idx, cp);
          jr += ROW(R.JR + c); ji += ROW(R.JI + c);
        }
        ROW(R.JR + i) = jr; ROW(R.JI + i) = ji;
      }
      __syncthreads();
    }
    // forward sweep, roots first
    for (int lv = T.n_levels - 1; lv >= 0; --lv) {
      const int t1 = cld(T.lvl_ptr, lv + 1);
      for (int t = cld(T.lvl_ptr, lv) + wave; t < t1; t += W) {
        const int i = cld(T.lvl_bus, t);
        const int p = cld(T.fbs_parent, i), pp = cld(T.fbs_parent_pos, i);
        // series admittance of the branch = -Y_ip (parallel lines merged)
        const double yr = -cld(T.G, pp), yi = -cld(T.Bv, pp);
        const double yd = yr * yr + yi * yi;
        const double jr = ROW(R.JR + i), ji = ROW(R.JI + i);
        // J / y = J conj(y) / |y|^2
        const double dr = (jr * yr + ji * yi) / yd, di = (ji * yr - jr * yi) / yd;
        if (upd) {
          ROW(R.E + i) = ROW(R.E + p) - dr;
          ROW(R.F + i) = ROW(R.F + p) - di;
        }
      }
      __syncthreads();
    }
    stale = true;
  }
  if (stale) {
    (void)mismatch_rows(T, R, S, wave, W);
  }
  __syncthreads();
  for (int i = wave; i < T.n; i += W) ROW(R.VM + i) = hypot(ROW(R.E + i), ROW(R.F + i));
  finish(T, R, S, sh, wave, W, lane, false, st.mm, st.iters, st.conv, st.status);
}

// =============================================================================================
// Newton-Raphson with a dense, partially pivoted LU per instance -- the reference-faithful
// linear solve (np.linalg.solve = LAPACK dgesv, power_flow.py:187): same unknown order, same
// pivot rule (largest |a_ik| in the column), exact-zero pivot = singular.  Row exchanges are
// per instance, so matrix rows are reached through a per-lane permutation (a gather: each lane
// reads its own row at its own lane slot).  This path exists for parity with the as-coded
// Jacobian, whose 2x2 diagonal blocks can be exactly singular; it is not the fast path.
// =============================================================================================
#define DA_AT(prow, c) S[((size_t)R.DA + (size_t)(prow) * N + (size_t)(c)) * GS_LANES]

extern "C" __global__ void __launch_bounds__(1024)
gs_k_nr_dense(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B) {
  __shared__ GsShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  const int N = T.dn_N;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  const bool valid = (int)(blockIdx.x * GS_LANES + lane) < B;

  NrState st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  flat_start(T, R, S, wave, W);
  __syncthreads();
  bool stale = true;
  int it = 0;
  for (; it < C.max_iterations; ++it) {
    to_rect(T, R, S, wave, W);
    __syncthreads();
    const double lmax = mismatch_rows(T, R, S, wave, W);
    const double mm = wg_max(sh, it & 1, wave, W, lane, lmax);
    nr_check(st, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done)) break;

    // ---- assemble the dense Jacobian and right-hand side ----
    for (int q = wave; q < N * N; q += W) ROW(R.DA + q) = 0.0;
    for (int q = wave; q < N; q += W) ROW(R.DPERM + q) = (double)q;
    __syncthreads();
    for (int i = wave; i < T.n; i += W) {
      const int ri0 = cld(T.dn_th_idx, i), ri1 = cld(T.dn_vm_idx, i);
      if (ri0 >= 0) ROW(R.DB + ri0) = ROW(R.R0 + i);
      if (ri1 >= 0) ROW(R.DB + ri1) = ROW(R.R1 + i);
      const int p1 = cld(T.row_ptr, i + 1);
      for (int p = cld(T.row_ptr, i); p < p1; ++p) {
        const int j = cld(T.col, p);
        const int cj0 = cld(T.dn_th_idx, j), cj1 = cld(T.dn_vm_idx, j);
        const Blk blk = (j == i) ? diag_block(T, R, S, i, C.jacobian_exact)
                                 : offdiag_block(T, R, S, i, j, cld(T.G, p), cld(T.Bv, p));
        if (ri0 >= 0 && cj0 >= 0) ROW(R.DA + ri0 * N + cj0) = blk.a00;
        if (ri0 >= 0 && cj1 >= 0) ROW(R.DA + ri0 * N + cj1) = blk.a01;
        if (ri1 >= 0 && cj0 >= 0) ROW(R.DA + ri1 * N + cj0) = blk.a10;
        if (ri1 >= 0 && cj1 >= 0) ROW(R.DA + ri1 * N + cj1) = blk.a11;
      }
    }
    __syncthreads();

    // ---- LU with partial pivoting (per lane) ----
    int sing = 0;
    for (int k = 0; k < N; ++k) {
      if (wave == 0) {
        double best = -1.0; int bi = k;
        for (int i = k; i < N; ++i) {
          const int pi = (int)ROW(R.DPERM + i);
          const double v = fabs(DA_AT(pi, k));
          if (v > best) { best = v; bi = i; }
        }
        if (!(best > 0.0)) sing = 1;
        const double pk = ROW(R.DPERM + k);
        const double pb = S[(size_t)(R.DPERM + bi) * GS_LANES];
        S[(size_t)(R.DPERM + bi) * GS_LANES] = pk;
        ROW(R.DPERM + k) = pb;
      }
      __syncthreads();
      const int pk = (int)ROW(R.DPERM + k);
      const double akk = DA_AT(pk, k);
      const double bk = S[(size_t)(R.DB + pk) * GS_LANES];
      for (int i = k + 1 + wave; i < N; i += W) {
        const int pi = (int)ROW(R.DPERM + i);
        const double l = DA_AT(pi, k) / akk;
        if (__any(l != 0.0)) {
          for (int c = k + 1; c < N; ++c) DA_AT(pi, c) -= l * DA_AT(pk, c);
          S[(size_t)(R.DB + pi) * GS_LANES] -= l * bk;
        }
      }
      __syncthreads();
    }
    const int sing_all = wg_or(sh, it & 1, wave, W, lane, sing);
    if (!st.done && sing_all) { st.status = GS_STATUS_SINGULAR; st.done = true; }
    const bool upd = !st.done;
    if (wave == 0) {
      for (int k = N - 1; k >= 0; --k) {
        const int pk = (int)ROW(R.DPERM + k);
        double s = S[(size_t)(R.DB + pk) * GS_LANES];
        for (int c = k + 1; c < N; ++c) s -= DA_AT(pk, c) * ROW(R.DX + c);
        ROW(R.DX + k) = s / DA_AT(pk, k);
      }
      for (int i = 0; i < T.n; ++i) {
        const int c0 = cld(T.dn_th_idx, i), c1 = cld(T.dn_vm_idx, i);
        ROW(R.X0 + i) = (c0 >= 0) ? ROW(R.DX + c0) : 0.0;
        ROW(R.X1 + i) = (c1 >= 0) ? ROW(R.DX + c1) : 0.0;
        apply_step(T, R, S, i, C.alpha, upd);
      }
    }
    __syncthreads();
    stale = true;
  }
  finish(T, R, S, sh, wave, W, lane, stale, st.mm, st.iters, st.conv, st.status);
}
