// kernels_sparse.hip -- Newton-Raphson for meshed networks with FEW loops: the 2x2-block sparse LU with every block of an
// instance in LDS, one wavefront per instance (gs_k_nr_sparse_lds, between gs_k_pre_nr_dmfma / gs_k_post_nr_dmfma).
//
// Why.  gs_k_step_nr_lu (kernels_solve.hip) puts an instance on a lane and the blocks of the factorisation in slab rows: every
// block operation is a 2 KB row access of a 64-instance group, and a batched step of the 123-bus feeder with 26 loops moves
// 1.47 GB through the fabric -- 20 x the algorithmic bytes, which is what bounds it (profiles/r03v7_meshed_loops26_*).  The
// factorisation of ONE instance is small: 480 off-diagonal + 123 diagonal blocks = 19 KB.  So here an instance's blocks, its
// right-hand side and its bus voltages live in LDS for the whole Newton loop (26 KB), ONE wavefront works on an instance -- its
// lanes go over what is independent inside a step: the buses (mismatch, corrections), the blocks (assembly), the items of a
// LEVEL of the elimination DAG (topology.cpp: pivots of one level are not adjacent in the filled graph) -- and needs no
// barrier, only the order of its own LDS accesses.  A workgroup is four such wavefronts (one per SIMD) that share one copy of
// the schedule and of the Ybus rows in LDS (50 KB): read from global memory, every step of the elimination was a chain of
// two or three dependent loads (the first version: 290 k cycles per instance, slower than the slab-row kernel).  HBM sees an
// instance's P / Q going in and its voltages coming out.
//
// Arithmetic: the block formulas, the order of the updates of a target and the back substitution are those of linsolve_lu
// (kernels_solve.hip; reference: mismatch power_flow.py:150-171, Jacobian entries :243-287, corrections :297-327); mismatch and
// corrections follow gs_k_nr_dense_mfma (Ybus rows in row order, sincos of the new angle).  The flat-start Jacobian is the same
// for every instance: its factors are computed once per handle (mode 1) and iteration 0 only substitutes.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"
#include "kernels.h"

namespace {

struct SBlk { double a00, a01, a10, a11; };

__device__ __forceinline__ SBlk sp_mul(const SBlk& x, const SBlk& y) {
  SBlk z;
  z.a00 = x.a00 * y.a00 + x.a01 * y.a10;
  z.a01 = x.a00 * y.a01 + x.a01 * y.a11;
  z.a10 = x.a10 * y.a00 + x.a11 * y.a10;
  z.a11 = x.a10 * y.a01 + x.a11 * y.a11;
  return z;
}
// inverse of a 2x2 block; sing is set when the determinant is exactly zero or non-finite (power_flow.py:188-190)
__device__ __forceinline__ SBlk sp_inv2(const SBlk& d, int& sing) {
  const double det = d.a00 * d.a11 - d.a01 * d.a10;
  if (!(det != 0.0) || !(fabs(det) < INFINITY)) sing = 1;
  const double r = 1.0 / det;
  SBlk z;
  z.a00 = d.a11 * r; z.a01 = -d.a01 * r; z.a10 = -d.a10 * r; z.a11 = d.a00 * r;
  return z;
}
__device__ __forceinline__ SBlk sp_ld(const double2* blk, int b) {
  const double2 lo = blk[2 * b], hi = blk[2 * b + 1];
  return SBlk{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void sp_st(double2* blk, int b, const SBlk& v) {
  blk[2 * b] = make_double2(v.a00, v.a01); blk[2 * b + 1] = make_double2(v.a10, v.a11);
}
__device__ __forceinline__ SBlk sp_flat(const double* flat, int b) {
  return SBlk{flat[4 * b], flat[4 * b + 1], flat[4 * b + 2], flat[4 * b + 3]};
}
// order of one wavefront's LDS accesses (its lanes hand data to each other; no other wavefront touches the instance)
__device__ __forceinline__ void sp_wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double sp_wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

struct SpStamp {
  unsigned long long* p; unsigned long long t;
  __device__ __forceinline__ void hit(int k) {
    if (p == nullptr) return;
    const unsigned long long now = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) p[k] += now - t;
    t = now;
  }
};

}  // namespace

extern __shared__ double gsp_lds[];

#define SP_MAXP 4                   /* buses per lane: lane l owns buses l, l + 64, ... (n <= 256) */

// diagnostic phase stamps (gs_debug_stamps): 0 mismatch, 1 assembly, 2 elimination, 3 back substitution, 4 corrections, 5 row I/O
extern "C" __global__ void __launch_bounds__(256)
gs_k_nr_sparse_lds(GsSparseArgs A, double* __restrict__ slab, int B) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = A.n, NS = A.n_slots, NL = A.n_levels;
  // ---- the workgroup's copy of the schedule and of the Ybus rows
  double* dp = gsp_lds;                                   // [dpack_n]
  int* ip = (int*)(dp + ((A.dpack_n + 1) & ~1));          // [ipack_n]
  for (int k = threadIdx.x; k < A.dpack_n; k += blockDim.x) dp[k] = A.dpack[k];
  for (int k = threadIdx.x; k < A.ipack_n; k += blockDim.x) ip[k] = A.ipack[k];
  __syncthreads();                                        // the only barrier of the kernel
  const int* row_ptr = ip + A.o_row_ptr; const int* col = ip + A.o_col;
  const int* th_free = ip + A.o_th_free; const int* vm_free = ip + A.o_vm_free; const int* fixed_v = ip + A.o_fixed_v;
  const int* piv_bus = ip + A.o_piv_bus; const int* nb_ptr = ip + A.o_nb_ptr; const int* nb_bus = ip + A.o_nb_bus; const int* nb_kj = ip + A.o_nb_kj;
  const int* a_ptr = ip + A.o_a_ptr; const int* a_it = ip + A.o_a;
  const int* b_ptr = ip + A.o_b_ptr; const int* b_rec = ip + A.o_b_rec; const int* b_pair = ip + A.o_b_pair;
  const int* r_ptr = ip + A.o_r_ptr; const int* r_rec = ip + A.o_r_rec; const int* r_pair = ip + A.o_r_pair;
  const int* c_ptr = ip + A.o_c_ptr; const int* c_it = ip + A.o_c;
  const double* Gv = dp + A.od_G; const double* Bvv = dp + A.od_B; const double* Gd = dp + A.od_Gd; const double* Bd = dp + A.od_Bd;
  const double* v_set = dp + A.od_vset;
  // ---- this wavefront's instance
  char* base = (char*)gsp_lds + (((size_t)((A.dpack_n + 1) & ~1) * 8 + (size_t)A.ipack_n * 4 + 15) & ~(size_t)15) + (size_t)wave * A.wave_bytes;
  double2* blk = (double2*)base;                          // [(NS + n) blocks x 2]: off-diagonal slots, then the diagonal block of bus i at NS + i
  double2* rhs = blk + 2 * (size_t)(NS + n);              // [n] (r_theta, r_v) of a bus; forward substitution in place
  double2* xs = rhs + n;                                  // [n] the Newton step
  double2* ef = xs + n;                                   // [n] (e, f)
  double* vm = (double*)(ef + n);                         // [n] |V| (the off-diagonal blocks of a bus's neighbours divide by it)
  SpStamp stp{A.stamps, 0ull};
  if (A.stamps) stp.t = __builtin_readcyclecounter();

  for (int b = blockIdx.x * A.waves + wave; b < B; b += gridDim.x * A.waves) {
    const int g = b >> 6, L6 = b & 63;
    double* Sg = slab + (size_t)g * A.rows_total * GS_LANES;
    auto row = [&](int r) -> double& { return Sg[GS_ELEM(r, L6)]; };
    const GsRows& R = A.R;
    // per-bus state of the lane's buses (angle, P / Q calculated and specified): registers
    double va[SP_MAXP], pc[SP_MAXP], qc[SP_MAXP], ps[SP_MAXP], qs[SP_MAXP];
    // ---- flat start (power_flow.py:125-134) and the specified injections
#pragma unroll
    for (int q = 0; q < SP_MAXP; ++q) {
      const int i = lane + 64 * q;
      va[q] = 0.0; pc[q] = 0.0; qc[q] = 0.0; ps[q] = 0.0; qs[q] = 0.0;
      if (i < n) {
        const double v0 = fixed_v[i] ? v_set[i] : 1.0;
        vm[i] = v0; ef[i] = make_double2(v0, 0.0);
        if (!A.mode) { ps[q] = row(R.P.base + 2 * i); qs[q] = row(R.Q.base + 2 * i); }
      }
    }
    sp_wsync();
    stp.hit(5);
    double mm = INFINITY; int iters = 0, conv = 0, status = GS_STATUS_MAX_ITER;
    bool stale = true;
    auto mismatch = [&](bool with_rhs) -> double {         // S = V conj(Y V) by Ybus rows, entries in row order (power_flow.py:150-171)
      double lmax = 0.0;
#pragma unroll
      for (int q = 0; q < SP_MAXP; ++q) {
        const int i = lane + 64 * q;
        if (i < n) {
          const double2 vi = ef[i];
          double P = 0.0, Q = 0.0;
          for (int p = row_ptr[i]; p < row_ptr[i + 1]; ++p) {
            const double2 vj = ef[col[p]];
            const double gg = Gv[p], bb0 = Bvv[p];
            const double a = vi.x * vj.x + vi.y * vj.y;
            const double bb = vi.y * vj.x - vi.x * vj.y;
            P += gg * a + bb0 * bb;
            Q += gg * bb - bb0 * a;
          }
          pc[q] = P; qc[q] = Q;
          if (with_rhs) {
            const double dP = th_free[i] ? (ps[q] - P) : 0.0, dQ = vm_free[i] ? (qs[q] - Q) : 0.0;
            rhs[i] = make_double2(dP, dQ);
            const double ap = fabs(dP), aq = fabs(dQ);
            lmax = fmax(lmax, fmax(ap < INFINITY ? ap : INFINITY, aq < INFINITY ? aq : INFINITY));
          }
        }
      }
      return lmax;
    };
    for (int it = 0; it < A.max_it; ++it) {
      mm = sp_wave_max(mismatch(true));
      sp_wsync();
      stp.hit(0);
      iters = it + 1;
      stale = false;
      if (!A.mode) {
        if (!(mm < INFINITY)) { status = GS_STATUS_NAN; break; }
        if (mm < A.tol) { conv = 1; status = GS_STATUS_OK; break; }
      }
      int sing = 0;
      const bool use_flat = A.flat != nullptr && it == 0 && !A.mode;
      if (use_flat) {
        // iteration 0: the handle's flat-start factors; forward substitution r_i -= (A_ik D_k^-1) r_k, level by level
        sing = A.flat[(size_t)4 * (NS + n)] != 0.0 ? 1 : 0;
        for (int L = 0; L < NL; ++L) {
          for (int rec = r_ptr[L] + lane; rec < r_ptr[L + 1]; rec += 64) {
            const int i = -r_rec[3 * rec] - 1 - n, cnt = r_rec[3 * rec + 1], off = r_rec[3 * rec + 2];
            double2 r = rhs[i];
            for (int u = 0; u < cnt; ++u) {
              const SBlk l = sp_flat(A.flat, r_pair[2 * (off + u)]);
              const double2 rk = rhs[r_pair[2 * (off + u) + 1]];
              r.x -= l.a00 * rk.x + l.a01 * rk.y;
              r.y -= l.a10 * rk.x + l.a11 * rk.y;
            }
            rhs[i] = r;
          }
          sp_wsync();
        }
        stp.hit(2);
      } else {
        // ---- Jacobian blocks (power_flow.py:243-287): the diagonal blocks by the buses' owners, the network's off-diagonal blocks, zero fill
#pragma unroll
        for (int q = 0; q < SP_MAXP; ++q) {
          const int i = lane + 64 * q;
          if (i < n && (th_free[i] || vm_free[i])) {
            const double gd = Gd[i], bd = Bd[i], v = vm[i], P = pc[q], Q = qc[q];
            const int th = th_free[i], vf = vm_free[i];
            const double vvb = v * v * bd;
            SBlk d;
            d.a00 = th ? (A.jacobian_exact ? (-Q - vvb) : (-Q + vvb)) : 1.0;
            d.a01 = (th && vf) ? (P / v + v * gd) : 0.0;
            d.a10 = (th && vf) ? (P - v * v * gd) : 0.0;
            d.a11 = vf ? (Q / v - v * bd) : 1.0;
            sp_st(blk, NS + i, d);
          }
        }
        for (int q = lane; q < A.n_orig; q += 64) {
          const int i = A.orig_i[q], j = A.orig_j[q], pos = A.orig_pos[q];
          const double gg = Gv[pos], bb0 = Bvv[pos];
          const double2 vi = ef[i], vj = ef[j];
          const double a = vi.x * vj.x + vi.y * vj.y;
          const double bb = vi.y * vj.x - vi.x * vj.y;
          const double gs_bc = gg * bb - bb0 * a, gc_bs = gg * a + bb0 * bb;
          const int thi = th_free[i], vfi = vm_free[i], thj = th_free[j], vfj = vm_free[j];
          SBlk u;
          u.a00 = (thi && thj) ? gs_bc : 0.0;
          u.a01 = (thi && vfj) ? gc_bs / vm[j] : 0.0;
          u.a10 = (vfi && thj) ? -gc_bs : 0.0;
          u.a11 = (vfi && vfj) ? gs_bc / vm[j] : 0.0;
          sp_st(blk, A.orig_slot[q], u);
        }
        for (int s = A.n_orig + lane; s < NS; s += 64) sp_st(blk, s, SBlk{0.0, 0.0, 0.0, 0.0});
        sp_wsync();
        stp.hit(1);
        // ---- elimination by levels: phase A scales the columns of the level's pivots, A_ik <- A_ik D_k^-1; phase B: every
        // block (and right-hand side) the level touches is owned by one lane, which subtracts all of the level's updates from it
        for (int L = 0; L < NL; ++L) {
          for (int q = a_ptr[L] + lane; q < a_ptr[L + 1]; q += 64) {
            const int k = a_it[2 * q], s = a_it[2 * q + 1];
            const SBlk inv = sp_inv2(sp_ld(blk, NS + k), sing);
            if (s >= 0) sp_st(blk, s, sp_mul(sp_ld(blk, s), inv));
          }
          sp_wsync();
          for (int rec = b_ptr[L] + lane; rec < b_ptr[L + 1]; rec += 64) {
            const int tgt = b_rec[3 * rec], cnt = b_rec[3 * rec + 1], off = b_rec[3 * rec + 2];
            if (tgt < -n) {                                   // right-hand side of bus i: r_i -= (A_ik D_k^-1) r_k
              const int i = -tgt - 1 - n;
              double2 r = rhs[i];
              for (int u = 0; u < cnt; ++u) {
                const SBlk l = sp_ld(blk, b_pair[2 * (off + u)]);
                const double2 rk = rhs[b_pair[2 * (off + u) + 1]];
                r.x -= l.a00 * rk.x + l.a01 * rk.y;
                r.y -= l.a10 * rk.x + l.a11 * rk.y;
              }
              rhs[i] = r;
            } else {
              const int slot = tgt >= 0 ? tgt : NS + (-tgt - 1);
              SBlk a = sp_ld(blk, slot);
              for (int u = 0; u < cnt; ++u) {
                const SBlk upd = sp_mul(sp_ld(blk, b_pair[2 * (off + u)]), sp_ld(blk, b_pair[2 * (off + u) + 1]));
                a.a00 -= upd.a00; a.a01 -= upd.a01; a.a10 -= upd.a10; a.a11 -= upd.a11;
              }
              sp_st(blk, slot, a);
            }
          }
          sp_wsync();
        }
        stp.hit(2);
        if (A.mode) {        // the handle's flat-start factors: the blocks as they stand, the flag behind them
          for (int e = lane; e < 2 * (NS + n); e += 64) { const double2 v = blk[e]; A.flat_out[2 * e] = v.x; A.flat_out[2 * e + 1] = v.y; }
          const int any = __any(sing);
          if (lane == 0) A.flat_out[(size_t)4 * (NS + n)] = any ? 1.0 : 0.0;
          return;
        }
      }
      sing = __any(sing) ? 1 : 0;
      if (sing) { status = GS_STATUS_SINGULAR; break; }        // power_flow.py:188-190: keep the current voltages
      // ---- back substitution, levels in reverse: x_k = D_k^-1 (r_k - sum_j A_kj x_j), every j in a higher level
      for (int L = NL - 1; L >= 0; --L) {
        for (int q = c_ptr[L] + lane; q < c_ptr[L + 1]; q += 64) {
          const int t = c_it[q], k = piv_bus[t];
          int dummy = 0;
          const SBlk inv = sp_inv2(use_flat ? sp_flat(A.flat, NS + k) : sp_ld(blk, NS + k), dummy);
          double2 r = rhs[k];
          for (int u = nb_ptr[t]; u < nb_ptr[t + 1]; ++u) {
            const SBlk akj = use_flat ? sp_flat(A.flat, nb_kj[u]) : sp_ld(blk, nb_kj[u]);
            const double2 xj = xs[nb_bus[u]];
            r.x -= akj.a00 * xj.x + akj.a01 * xj.y;
            r.y -= akj.a10 * xj.x + akj.a11 * xj.y;
          }
          xs[k] = make_double2(inv.a00 * r.x + inv.a01 * r.y, inv.a10 * r.x + inv.a11 * r.y);
        }
        sp_wsync();
      }
      stp.hit(3);
      // ---- corrections (power_flow.py:297-327) and the new rectangular voltages
#pragma unroll
      for (int q = 0; q < SP_MAXP; ++q) {
        const int i = lane + 64 * q;
        if (i < n && (th_free[i] || vm_free[i])) {
          double v = vm[i], th = va[q];
          const double2 x = xs[i];
          if (th_free[i]) th += A.alpha * x.x;
          if (vm_free[i]) v += A.alpha * x.y;
          if (v < 0.0) { v = -v; th += M_PI; }
          double sn, cs;
          sincos(th, &sn, &cs);
          vm[i] = v; va[q] = th; ef[i] = make_double2(v * cs, v * sn);
        }
      }
      sp_wsync();
      stp.hit(4);
      stale = true;
    }
    if (stale) { (void)mismatch(false); }     // iteration cap reached after an update: P / Q calculated at the final voltages (the epilogue's losses)
    // ---- what newton_loop leaves in the rows
#pragma unroll
    for (int q = 0; q < SP_MAXP; ++q) {
      const int i = lane + 64 * q;
      if (i < n) {
        const double2 v = ef[i];
        row(R.VM.base + 2 * i) = vm[i]; row(R.VA.base + 2 * i) = va[q];
        row(R.E.base + 2 * i) = v.x; row(R.F.base + 2 * i) = v.y;
        row(R.PC.base + 2 * i) = pc[q]; row(R.QC.base + 2 * i) = qc[q];
      }
    }
    if (lane == 0) { row(R.MAXMIS) = mm; row(R.ITERS) = (double)iters; row(R.CONV) = (double)conv; row(R.STATUS) = (double)status; }
    sp_wsync();
    stp.hit(5);
  }
}
