// kernels_dense.hip -- Newton-Raphson for MESHED feeders whose Jacobian fills in under elimination (the reference's
// ScalableFeeder recipe, feeders/synthetic.py:233: ~1000 lines on 123 buses -- the sparse block LU of kernels_solve.hip
// ends up with 58 % of all blocks): one WORKGROUP per instance, the Jacobian as a dense matrix in 64-wide panels through
// LDS, eliminated by blocks with v_mfma_f64_16x16x4_f64.  north_star: "MFMA only if the per-feeder Jacobian is densified
// as a batched small GEMM" -- this is that case; the sparse and radial paths stay on the vector units.
//
// The reference solves the same system densely too (np.linalg.solve, power_flow.py:186-190).  Restated here, one
// instance per workgroup (paths relative to /root/reference/grid_fed_rl/environments/): flat start power_flow.py:125-134,
// mismatch :150-171, Jacobian entries :243-287 (exact sign of the J11 diagonal), corrections :297-327, iteration count /
// convergence :143-193, 204.  Same arithmetic per entry as the lane-per-instance kernels (mismatch_rows, diag_from,
// offdiag_from, apply_step, to_rect of kernels_solve.hip); what differs is the linear solve:
//
//   unknowns  u = 2 a + {0: angle, 1: magnitude} for the a-th non-slack bus, padded with identity rows to NP = 64 * NB;
//   block LU with 64 x 64 blocks, LEFT-looking: panel j (all NP rows of 64 columns) is assembled in LDS from the Ybus rows,
//     C_i -= L_ik C_k for every earlier panel k (L_ik from this workgroup's scratch in global memory -- 512 KB that stays
//     in the Infinity Cache --, C_k from the panel itself), the diagonal block is inverted in place (Gauss-Jordan, no
//     pivoting: the exact Jacobian's diagonal blocks are dominant, as for the 2x2-block LU), L_ij = C_i D_j^-1, and the
//     right-hand side is carried along; back substitution x_j = D_j^-1 (r_j - sum_{k > j} U_jk x_k) from the scratch.
//   Every 64^3 product is 64 MFMAs per wavefront (4 wavefronts = the 4 tile rows of the block); the A operands of a product
//   that come from global memory are requested up front (16 registers), B operands and accumulators go through LDS.
//
// Per Newton iteration and instance: 2/3 N^3 + ... = 9.7 MFLOP at N = 244; BASELINE-sized batches run as a persistent grid
// (one workgroup per CU, instances b = workgroup, workgroup + grid, ...).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"

#pragma clang fp contract(off)

#define DB 64                 /* block size */
#define DLD 66                /* leading dimension of the panel in LDS (doubles): even (16-byte rows), not a multiple of 32 banks */

typedef double gd_v4 __attribute__((ext_vector_type(4)));
typedef double gd_v2 __attribute__((ext_vector_type(2)));

extern __shared__ __attribute__((aligned(16))) double gd_lds[];

__device__ __forceinline__ void gd_sync() { __syncthreads(); }
// a barrier that waits for this wavefront's LDS traffic only: global loads requested before it stay in flight
__device__ __forceinline__ void gd_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// block maximum / OR over the 256 threads (sh: 8 doubles)
__device__ __forceinline__ double gd_block_max(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return r;
}

// C (64 x 64 in LDS, leading dimension ldc) <- C - A B   (SUB)   or   A B   (!SUB; C may alias A when A is in LDS: a wavefront
// reads only the rows of A it writes, and it has all of them in registers before the first store)
// A: 64 x 64, row-major, leading dimension lda, in global memory or LDS; B: 64 x 64 in LDS, leading dimension ldb.
// Wavefront w owns rows 16 w .. 16 w + 15 of C: four 16 x 16 tiles; v_mfma_f64_16x16x4_f64 operand maps (MI355X guide):
// A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], C/D[row = (lane >> 4) + 4 reg][col = lane & 15].
// the sixteen A operands of a product for this lane (rows 16 w .. 16 w + 15 of A, k = 4 s + (lane >> 4))
__device__ __forceinline__ void gd_load_a(double (&a)[16], const double* A, int lda) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, q = l >> 4;
#pragma unroll
  for (int s = 0; s < 16; ++s) a[s] = A[(size_t)(16 * w + r) * lda + 4 * s + q];
}
template <bool NEG>
__device__ __forceinline__ void gd_mfma_acc(gd_v4 (&acc)[4], const double (&a)[16], const double* B, int ldb);
template <bool SUB>
__device__ __forceinline__ void gd_gemm64_a(double* C, int ldc, const double (&a)[16], const double* B, int ldb) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  gd_v4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[t][g] = SUB ? C[(16 * w + q + 4 * g) * ldc + 16 * t + r] : 0.0;
  gd_mfma_acc<SUB>(acc, a, B, ldb);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) C[(16 * w + q + 4 * g) * ldc + 16 * t + r] = acc[t][g];
}
template <bool SUB>
__device__ __forceinline__ void gd_gemm64(double* C, int ldc, const double* A, int lda, const double* B, int ldb) {
  double a[16];
  gd_load_a(a, A, lda);
  gd_gemm64_a<SUB>(C, ldc, a, B, ldb);
}

// In-place inverse of the 64 x 64 block D (LDS, leading dimension ld) by BLOCK Gauss-Jordan elimination without pivoting:
// sixteen steps with 4 x 4 pivot blocks instead of sixty-four scalar ones (a step is a chain of LDS write -> barrier -> LDS
// read -> arithmetic, ~1000 cycles whatever it computes; the scalar form spent 94 k cycles per inversion, half the kernel).
// Thread (tr, tc) = (tid >> 4, tid & 15) keeps the 4 x 4 tile rows 4 tr.., columns 4 tc.. in registers throughout.  Step kb:
// the owner of the pivot tile P inverts it in registers and publishes P^-1; the owners of P's block row publish their tiles
// R_j, the owners of its block column their tiles C_i.  With T_j = P^-1 R_j, in-place Gauss-Jordan is
//     M_ij -= C_i T_j  (i, j != kb),   row block kb <- T_j,   column block kb <- -C_i P^-1,   pivot tile <- P^-1,
// the row block's owners form T_j and publish it (second barrier), their tiles and the pivot tile are then final; every other
// tile takes M -= C_i T'_j with T'_kb = I + P^-1 for the pivot's tile column, which yields -C_i P^-1 there (the block form of
// the scalar identity 1 + 1/m_kk; it loses log10 |P| digits in that column block, three of sixteen for these matrices).
// buf: [2 parities][rowb 4 x 64 | colb 64 x 4 | pinv 16] doubles.  Returns 1 if a pivot was zero or not finite.
#define GD_GJ_DOUBLES (2 * 528)
__device__ __forceinline__ double gd_rcp(double x) {
  double p = __builtin_amdgcn_rcp(x);
  p = __builtin_fma(p, __builtin_fma(-x, p, 1.0), p);
  return __builtin_fma(p, __builtin_fma(-x, p, 1.0), p);
}
__device__ __forceinline__ int gd_invert64(double* D, int ld, double* buf) {
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  double m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 lo = *(const double2*)(D + (4 * tr + i) * ld + 4 * tc), hi = *(const double2*)(D + (4 * tr + i) * ld + 4 * tc + 2);
    m[i][0] = lo.x; m[i][1] = lo.y; m[i][2] = hi.x; m[i][3] = hi.y;
  }
  int bad = 0;
  for (int kb = 0; kb < 16; ++kb) {
    double* rowb = buf + (kb & 1) * 528, *colb = rowb + 256, *pinv = colb + 256;
    const bool rown = tr == kb, cown = tc == kb;
    // (1) the pivot tile's owner inverts it in registers and publishes P^-1; the column block's owners publish their tiles C_i
    if (cown) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(double2*)(colb + 16 * tr + 4 * i) = make_double2(m[i][0], m[i][1]);
        *(double2*)(colb + 16 * tr + 4 * i + 2) = make_double2(m[i][2], m[i][3]);
      }
    }
    if (rown && cown) {      // in-place Gauss-Jordan on 4 x 4, static indices
      double a[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = m[i][j];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double piv = a[k][k];
        if (!(piv != 0.0) || !(fabs(piv) < INFINITY)) bad = 1;
        const double p = gd_rcp(piv);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j != k) a[k][j] *= p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i == k) continue;
          const double f = a[i][k];
#pragma unroll
          for (int j = 0; j < 4; ++j) if (j != k) a[i][j] = __builtin_fma(-f, a[k][j], a[i][j]);
          a[i][k] = -(f * p);
        }
        a[k][k] = p;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(double2*)(pinv + 4 * i) = make_double2(a[i][0], a[i][1]);
        *(double2*)(pinv + 4 * i + 2) = make_double2(a[i][2], a[i][3]);
      }
    }
    __syncthreads();
    // (2) the row block's owners form T_j = P^-1 R_j once (every thread of a tile column formed it for itself before: half
    // of a step's arithmetic and LDS reads) and publish it; their own tiles take it as they stand
    if (rown) {
      double pi[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 p0 = *(const double2*)(pinv + 4 * i), p1 = *(const double2*)(pinv + 4 * i + 2);
        pi[i][0] = p0.x; pi[i][1] = p0.y; pi[i][2] = p1.x; pi[i][3] = p1.y;
      }
      double t[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double v = pi[i][0] * m[0][j];
          v = __builtin_fma(pi[i][1], m[1][j], v); v = __builtin_fma(pi[i][2], m[2][j], v); v = __builtin_fma(pi[i][3], m[3][j], v);
          t[i][j] = v;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // the tile column of the pivot publishes T' = I + P^-1 (what makes the uniform update below right for the column block)
        const double d0 = (i == 0 ? 1.0 : 0.0), d1 = (i == 1 ? 1.0 : 0.0), d2 = (i == 2 ? 1.0 : 0.0), d3 = (i == 3 ? 1.0 : 0.0);
        const double2 lo = cown ? make_double2(pi[i][0] + d0, pi[i][1] + d1) : make_double2(t[i][0], t[i][1]);
        const double2 hi = cown ? make_double2(pi[i][2] + d2, pi[i][3] + d3) : make_double2(t[i][2], t[i][3]);
        *(double2*)(rowb + 64 * i + 4 * tc) = lo;
        *(double2*)(rowb + 64 * i + 4 * tc + 2) = hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = cown ? pi[i][j] : t[i][j];
      }
    }
    __syncthreads();
    // (3) everybody else: M -= C_i T_j
    if (!rown) {
      double t[4][4], c[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 r0 = *(const double2*)(rowb + 64 * i + 4 * tc), r1 = *(const double2*)(rowb + 64 * i + 4 * tc + 2);
        const double2 c0 = *(const double2*)(colb + 16 * tr + 4 * i), c1 = *(const double2*)(colb + 16 * tr + 4 * i + 2);
        t[i][0] = r0.x; t[i][1] = r0.y; t[i][2] = r1.x; t[i][3] = r1.y;
        c[i][0] = c0.x; c[i][1] = c0.y; c[i][2] = c1.x; c[i][3] = c1.y;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double v = m[i][j];
          v = __builtin_fma(-c[i][0], t[0][j], v); v = __builtin_fma(-c[i][1], t[1][j], v);
          v = __builtin_fma(-c[i][2], t[2][j], v); v = __builtin_fma(-c[i][3], t[3][j], v);
          m[i][j] = v;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *(double2*)(D + (4 * tr + i) * ld + 4 * tc) = make_double2(m[i][0], m[i][1]);
    *(double2*)(D + (4 * tr + i) * ld + 4 * tc + 2) = make_double2(m[i][2], m[i][3]);
  }
  __syncthreads();
  return bad;
}

// y (64, LDS) <- y - M x  (M: 64 x 64 row-major with leading dimension ld, global or LDS; x: 64 in LDS), all 256 threads:
// four threads per row, sixteen columns each, partial sums added in a fixed order.  Ends with a barrier.
template <bool ASSIGN_NEG>      // false: y -= M x; true: y = M x (no subtraction)
__device__ __forceinline__ void gd_matvec64(double* y, const double* M, int ld, const double* x) {
  const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += M[(size_t)row * ld + 16 * part + c] * x[16 * part + c];
  const double s1 = __shfl_xor(s, 1);
  const double s01 = (part & 1) ? s1 + s : s + s1;           // (part 0 + part 1) or (part 2 + part 3), lower part first
  const double s2 = __shfl_xor(s01, 2);
  const double tot = (part & 2) ? s2 + s01 : s01 + s2;
  __syncthreads();                                           // every reader of the old y / x is done
  if (part == 0) y[row] = ASSIGN_NEG ? tot : y[row] - tot;
  __syncthreads();
}

// One block step of the back substitution, x_j = D_j^-1 (r_j - sum_{q < cnt} U_q x_q), in one pass: the row segment of D_j^-1 is requested
// together with those of the U blocks (two dependent round trips to the scratch before), the difference goes through `tmp` (64 doubles
// of LDS) and x_j overwrites r_j.  Same sums in the same order as gd_matvec64_sum followed by gd_matvec64<true>; two barriers, not five.
__device__ __forceinline__ void gd_backstep(double* y, const double* U0, const double* x0, int cnt, const double* Dinv, double* tmp) {
  const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
  double dv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) dv[c] = Dinv[(size_t)row * DB + 16 * part + c];
  double s = 0.0;
  for (int q = 0; q < cnt; ++q) {
    const double* M = U0 + (size_t)q * DB * DB + (size_t)row * DB + 16 * part;
    const double* x = x0 + q * DB + 16 * part;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += M[c] * x[c];
  }
  {
    const double s1 = __shfl_xor(s, 1);
    const double s01 = (part & 1) ? s1 + s : s + s1;
    const double s2 = __shfl_xor(s01, 2);
    const double tot = (part & 2) ? s2 + s01 : s01 + s2;
    if (part == 0) tmp[row] = cnt > 0 ? y[row] - tot : y[row];
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) t += dv[c] * tmp[16 * part + c];
  const double t1 = __shfl_xor(t, 1);
  const double t01 = (part & 1) ? t1 + t : t + t1;
  const double t2 = __shfl_xor(t01, 2);
  const double tot2 = (part & 2) ? t2 + t01 : t01 + t2;
  if (part == 0) y[row] = tot2;
  __syncthreads();
}

// scratch block addresses (64 x 64 row-major each): L(i, j) i > j, U(j, k) k > j, Dinv(j)
__device__ __forceinline__ double* gd_blk(double* base, int NB, int kind, int a, int b) {
  // kind 0: L(a, b), a > b -> index a (a - 1) / 2 + b (a block row of L is contiguous);  kind 1: U(a, b), b > a -> NBT +
  // a (NB - 1) - a (a - 1) / 2 + (b - a - 1) (a block row of U is contiguous);  kind 2: Dinv(a)
  const int nbt = NB * (NB - 1) / 2;
  const int idx = kind == 0 ? a * (a - 1) / 2 + b : kind == 1 ? nbt + a * (NB - 1) - a * (a - 1) / 2 + (b - a - 1) : 2 * nbt + a;
  return base + (size_t)idx * DB * DB;
}

// y (64, LDS) <- y - sum_{q < cnt} M_q x_q: the blocks M_q (global, 64 x 64 row-major) are `mstride` doubles apart, the vectors x_q
// `xstride` doubles apart in LDS.  One pass: every load of the sum is in flight before the first is used (a block at a time
// each cost a round trip to the Infinity Cache and two barriers).
__device__ __forceinline__ void gd_matvec64_sum(double* y, const double* M0, long mstride, const double* x0, int xstride, int cnt) {
  const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
  double s = 0.0;
  for (int q = 0; q < cnt; ++q) {
    const double* M = M0 + (size_t)q * mstride + (size_t)row * DB + 16 * part;
    const double* x = x0 + q * xstride + 16 * part;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += M[c] * x[c];
  }
  const double s1 = __shfl_xor(s, 1);
  const double s01 = (part & 1) ? s1 + s : s + s1;
  const double s2 = __shfl_xor(s01, 2);
  const double tot = (part & 2) ? s2 + s01 : s01 + s2;
  __syncthreads();
  if (part == 0) y[row] -= tot;
  __syncthreads();
}

// ---- the block-row form (gs_k_nr_dense_mfma2): accumulators in registers, BOTH operands of an update from the scratch ------------
// acc (rows 16 w .. 16 w + 15 of a 64 x 64 block, four 16 x 16 tiles in the C/D layout) -= A B: the A operands in registers (gd_load_a),
// B a 64 x 64 block in LDS.
// (the B operands of two k-steps -- eight LDS reads -- are requested before the eight MFMAs of the two steps before them are
// issued: read where they are used, every pair of MFMAs waited for an LDS round trip)
template <bool NEG>
__device__ __forceinline__ void gd_mfma_acc(gd_v4 (&acc)[4], const double (&a)[16], const double* B, int ldb) {
  const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  const double* Bq = B + q * ldb + r;
  double b[2][2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) b[0][s][t] = Bq[4 * s * ldb + 16 * t];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < 7) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) b[(c + 1) & 1][s][t] = Bq[4 * (2 * (c + 1) + s) * ldb + 16 * t];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const double as = NEG ? -a[2 * c + s] : a[2 * c + s];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, b[c & 1][s][t], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void gd_acc_load(gd_v4 (&acc)[4], const double* C, int ldc) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, q = l >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[t][g] = C[(16 * w + q + 4 * g) * ldc + 16 * t + r];
}
__device__ __forceinline__ void gd_acc_store(const gd_v4 (&acc)[4], double* C, int ldc) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, q = l >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) C[(size_t)(16 * w + q + 4 * g) * ldc + 16 * t + r] = acc[t][g];
}

__device__ __forceinline__ int gd_invert4(double (&a)[4][4]) {      // in-place Gauss-Jordan on 4 x 4, static indices; 1: a pivot was zero / not finite
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double piv = a[k][k];
    if (!(piv != 0.0) || !(fabs(piv) < INFINITY)) bad = 1;
    const double p = gd_rcp(piv);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j != k) a[k][j] *= p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == k) continue;
      const double f = a[i][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) if (j != k) a[i][j] = __builtin_fma(-f, a[k][j], a[i][j]);
      a[i][k] = -(f * p);
    }
    a[k][k] = p;
  }
  return bad;
}
// The inversion for the block-row kernel: the same block Gauss-Jordan with 4 x 4 pivot blocks, but the 64 x 64 block lives in MFMA
// accumulators (wavefront w: rows 16 w .. 16 w + 15) and a step's rank-4 update  M -= C T  is ONE v_mfma_f64_16x16x4 per 16 x 16 tile.
// The FP64 matrix rate equals the vector rate on this part, so the arithmetic costs the same pipe time -- but it is 4 instructions
// instead of 64 multiply-adds per lane, on a pipe the neighbouring workgroup's vector instructions do not queue for, and the step
// needs ONE barrier: the pivot row block R (4 x 64, raw) and column block C (64 x 4) are published, and every lane inverts the pivot
// tile for itself (same instructions on the same operands), forms the entries of T = P^-1 R its B operands need (the pivot's tile
// column: T' = I + P^-1, which leaves -C P^-1 there), and issues the four MFMAs; the lanes holding the pivot's rows then overwrite
// them with T (P^-1 in the pivot tile).  buf: [2 parities][R as [64 columns][4] | C as [64 rows][4]] doubles.
__device__ __forceinline__ int gd_invert64_mfma(double* D, int ld, double* buf) {
  // (the lane's coordinates through an opaque copy: what the steps derive from them is formed where it is used, not in front of the
  // kernel's outer loops)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int w = __builtin_amdgcn_readfirstlane(tid_ >> 6), l = tid_ & 63, r = l & 15, q = l >> 4;
  gd_v4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[t][g] = D[(16 * w + q + 4 * g) * ld + 16 * t + r];
  int bad = 0;
  // (the tile of the pivot's columns is named at compile time -- four copies of the step --, the register of its rows is selected
  // at run time: unrolled over all sixteen steps, every step's lane masks were formed up front and kept: 196 spilled scalar registers)
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma nounroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kb = 4 * tq + gq;
      double* rowb = buf + (kb & 1) * 512, *colb = rowb + 256;
      const int wq = tq;                                    // the wavefront holding the pivot's rows (accumulator register gq);
      const int rq = 4 * gq;                                // tile tq, lane columns rq .. rq + 3 hold the pivot's columns
      if (w == wq) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double v = gq == 0 ? acc[t][0] : gq == 1 ? acc[t][1] : gq == 2 ? acc[t][2] : acc[t][3];
          rowb[(16 * t + r) * 4 + q] = v;
        }
      }
      if ((r >> 2) == gq) {
        const gd_v4 c = acc[tq];
#pragma unroll
        for (int g = 0; g < 4; ++g) colb[(16 * w + q + 4 * g) * 4 + (r & 3)] = c[g];
      }
      gd_sync_lds();                                        // (one barrier per step; it waits for LDS only)
      double pi[4][4];
      {
        const double* pp = rowb + (16 * tq + rq) * 4;           // P[k][c] at pp[4 c + k]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const gd_v2 lo = *(const gd_v2*)(pp + 4 * c), hi = *(const gd_v2*)(pp + 4 * c + 2);
          pi[0][c] = lo.x; pi[1][c] = lo.y; pi[2][c] = hi.x; pi[3][c] = hi.y;
        }
      }
      bad |= gd_invert4(pi);
      const double pq0 = q == 0 ? pi[0][0] : q == 1 ? pi[1][0] : q == 2 ? pi[2][0] : pi[3][0];
      const double pq1 = q == 0 ? pi[0][1] : q == 1 ? pi[1][1] : q == 2 ? pi[2][1] : pi[3][1];
      const double pq2 = q == 0 ? pi[0][2] : q == 1 ? pi[1][2] : q == 2 ? pi[2][2] : pi[3][2];
      const double pq3 = q == 0 ? pi[0][3] : q == 1 ? pi[1][3] : q == 2 ? pi[2][3] : pi[3][3];      // row q of P^-1
      const double a = -colb[(16 * w + r) * 4 + q];
      double tt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const gd_v2 lo = *(const gd_v2*)(rowb + (16 * t + r) * 4), hi = *(const gd_v2*)(rowb + (16 * t + r) * 4 + 2);
        double v = pq0 * lo.x;
        v = __builtin_fma(pq1, lo.y, v); v = __builtin_fma(pq2, hi.x, v); v = __builtin_fma(pq3, hi.y, v);
        tt[t] = v;
      }
      const bool pcol = (r >> 2) == gq;                     // this lane's column of tile tq is one of the pivot's
      const int rc = r & 3;
      const double pqc = rc == 0 ? pq0 : rc == 1 ? pq1 : rc == 2 ? pq2 : pq3;      // P^-1[q][column]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool pt = pcol && t == tq;
        const double b = pt ? pqc + (q == rc ? 1.0 : 0.0) : tt[t];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
      if (w == wq) {                                        // the pivot's rows: T, and P^-1 in the pivot tile
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double v = (pcol && t == tq) ? pqc : tt[t];
          if (gq == 0) acc[t][0] = v; else if (gq == 1) acc[t][1] = v; else if (gq == 2) acc[t][2] = v; else acc[t][3] = v;
        }
      }
    }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) D[(size_t)(16 * w + q + 4 * g) * ld + 16 * t + r] = acc[t][g];
  __syncthreads();
  return bad;
}

// diagnostic phase stamps: slots 0 mismatch, 1 panel assembly, 2 updates from earlier panels (MFMA), 3 U / D^-1 copies,
// 4 Gauss-Jordan, 5 L = C D^-1 (MFMA) + copies + forward substitution, 6 back substitution, 7 corrections, 8 row I/O
struct GdStamp {
  unsigned long long* p; unsigned long long t;
  __device__ __forceinline__ void hit(int k) {
    if (p == nullptr) return;
    const unsigned long long now = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) p[k] += now - t;
    t = now;
  }
};

// BLOCKROW false: the panel form described at the top (one workgroup per CU: the panel takes 135 KB of LDS).  BLOCKROW true: the
// same left-looking block LU one 64 x 64 block at a time -- block (i, j) is assembled in LDS, taken into MFMA accumulators,
// updated with  -= sum_{k < min(i, j)} L_ik U_kj  from the scratch (both operands: nothing of the panel stays in LDS), and leaves as
// U_ij (i < j, straight from the registers), as D_j^-1 (Gauss-Jordan in LDS) or as L_ij = C D_j^-1.  Two block buffers instead of a
// panel: 78 KB of LDS, TWO workgroups per CU -- the solve is a chain of short dependent phases (sixteen-step Gauss-Jordan, block
// substitutions, assembly) at one wavefront per SIMD, and a second instance beside it is what fills the gaps.
template <bool BLOCKROW>
__device__ __forceinline__ void gd_newton(GsDenseArgs A_, double* __restrict__ slab_, int B_) {
  // (arguments read in place through an opaque pointer, kernels_flow2.hip F2_ARGS_IN_PLACE: no scalar words parked in vector lanes)
  struct ArgBlock { GsDenseArgs A; double* slab; int B; };
  const __attribute__((address_space(4))) char* ka_ = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka_));
  const GsDenseArgs& A = ((const ArgBlock*)ka_)->A;
  double* __restrict__ const slab = ((const ArgBlock*)ka_)->slab;
  const int B = ((const ArgBlock*)ka_)->B;
  const int tid = threadIdx.x, n = A.n, na = A.na, NB = A.NB, NP = DB * NB;
  // LDS carve-up
  double* panel = gd_lds;                                  // [NP][DLD]   (BLOCKROW: bufC [64][DLD], bufD [64][DLD])
  double* bufC = gd_lds; double* bufD = gd_lds + (size_t)DB * DLD;
  double* rhs = panel + (size_t)(BLOCKROW ? 2 * DB : NP) * DLD;                  // [NP] right-hand side, then the solution
  double* ve = rhs + NP;                                   // [n] e, f, |V|, angle, P calc, Q calc, P spec, Q spec
  const int n2 = (n + 1) & ~1;                             // (keeps what follows 16-byte aligned)
  double* vf = ve + n2; double* vm = vf + n2; double* va = vm + n2; double* pc = va + n2; double* qc = pc + n2; double* ps = qc + n2; double* qs = ps + n2;
  // [1056] pivot row / column blocks of the Gauss-Jordan steps, two parities (BLOCKROW: in bufC, which is idle while a diagonal block
  // is inverted in bufD; the back substitution's 64 doubles likewise)
  double* gjbuf = BLOCKROW ? bufC : qs + n2;
  double* red = BLOCKROW ? qs + n2 : gjbuf + GD_GJ_DOUBLES;                     // [8]
  double* scr = A.scratch + (size_t)blockIdx.x * (size_t)(NB * NB) * DB * DB;
  GdStamp stp{A.stamps, 0ull};
  if (A.stamps) stp.t = __builtin_readcyclecounter();

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int g = b >> 6, L = b & 63;
    // this instance's rows: pair (2 k, 2 k + 1) of lane L at group base + k * 1024 + L * 16 bytes (GS_ELEM)
    double* Sg = slab + (size_t)g * A.rows_total * GS_LANES;
    auto row = [&](int r) -> double& { return Sg[GS_ELEM(r, L)]; };
    const GsRows& R = A.R;
    // ---- flat start (power_flow.py:125-134) and the specified injections
    for (int i = tid; i < n; i += blockDim.x) {
      const double v0 = A.fixed_v[i] ? A.v_set[i] : 1.0;
      vm[i] = v0; va[i] = 0.0; ve[i] = v0; vf[i] = 0.0;
      ps[i] = A.mode ? 0.0 : row(R.P.base + 2 * i); qs[i] = A.mode ? 0.0 : row(R.Q.base + 2 * i);
    }
    __syncthreads();
    stp.hit(8);
    double mm = INFINITY; int iters = 0, conv = 0, status = GS_STATUS_MAX_ITER;
    bool stale = true;
    for (int it = 0; it < A.max_it; ++it) {
      // ---- mismatch (power_flow.py:150-171): S = V conj(Y V) by Ybus rows, entries in row order
      // (two threads per bus, the even and the odd entries of its row, two entries in flight each: a row of the ScalableFeeder-like
      // network has ~17 entries and every one is a round trip to the Ybus arrays; sum = even part + odd part)
      double lmax = 0.0;
      for (int i = tid >> 1; i < n; i += blockDim.x >> 1) {
        const double ei = ve[i], fi = vf[i];
        double P = 0.0, Q = 0.0, P2 = 0.0, Q2 = 0.0;
        const int r1 = A.row_ptr[i + 1];
        for (int p = A.row_ptr[i] + (tid & 1); p < r1; p += 4) {
          const int p2 = p + 2 < r1 ? p + 2 : p;
          const int j = A.col[p], j2 = A.col[p2];
          const double gg = A.G[p], bb0 = A.Bv[p], gg2 = p + 2 < r1 ? A.G[p2] : 0.0, bb2 = p + 2 < r1 ? A.Bv[p2] : 0.0;
          const double a = ei * ve[j] + fi * vf[j], a2 = ei * ve[j2] + fi * vf[j2];
          const double bb = fi * ve[j] - ei * vf[j], bq = fi * ve[j2] - ei * vf[j2];
          P += gg * a + bb0 * bb;
          Q += gg * bb - bb0 * a;
          P2 += gg2 * a2 + bb2 * bq;
          Q2 += gg2 * bq - bb2 * a2;
        }
        P += P2; Q += Q2;
        {
          const double Po = __shfl_xor(P, 1), Qo = __shfl_xor(Q, 1);
          P = (tid & 1) ? Po + P : P + Po; Q = (tid & 1) ? Qo + Q : Q + Qo;
        }
        if (tid & 1) continue;
        pc[i] = P; qc[i] = Q;
        const double dP = A.th_free[i] ? (ps[i] - P) : 0.0, dQ = A.vm_free[i] ? (qs[i] - Q) : 0.0;
        const int a_ = A.act_of[i];
        if (a_ >= 0) { rhs[2 * a_] = dP; rhs[2 * a_ + 1] = dQ; }
        const double ap = fabs(dP), aq = fabs(dQ);
        lmax = fmax(lmax, fmax(ap < INFINITY ? ap : INFINITY, aq < INFINITY ? aq : INFINITY));
      }
      for (int u = 2 * na + tid; u < NP; u += blockDim.x) rhs[u] = 0.0;
      mm = gd_block_max(lmax, red);                      // (its barriers also publish pc / qc / rhs)
      stp.hit(0);
      iters = it + 1;
      stale = false;
      if (!A.mode) {
        if (!(mm < INFINITY)) { status = GS_STATUS_NAN; break; }
        if (mm < A.tol) { conv = 1; status = GS_STATUS_OK; break; }
      }
      int sing = 0;
      bool skip_subst = false;
      double* fs = scr;                                    // where this iteration's block factors live
      if (A.flat != nullptr && it == 0 && !A.mode) {
        // iteration 0: the flat-start factors of the handle; forward substitution r_i -= L_ij r_j from the stored blocks
        fs = A.flat;
        sing = A.flat[(size_t)NB * NB * DB * DB] != 0.0 ? 1 : 0;
        if constexpr (BLOCKROW) {
          if (A.jinv_t != nullptr) {
            // x = J0^-1 rhs as one product: thread u forms sum_c J0^-1[u][c] rhs[c] over the transposed inverse (a coalesced 8-byte
            // column read per c, the same 512 KB for every workgroup -> L2; rhs[c] is an LDS broadcast), sixty-four columns in flight
            double acc_x = 0.0;
            if (tid < NP) {
              const double* jt = A.jinv_t + tid;
              // (NP is a multiple of 64; a round trip to the table under load is ~2 k cycles: sixteen columns in flight made the
              // product as long as the substitutions it replaces)
              for (int c0 = 0; c0 < NP; c0 += 64) {
                double jv[64];
#pragma unroll
                for (int e = 0; e < 64; ++e) jv[e] = jt[(size_t)(c0 + e) * NP];
#pragma unroll
                for (int e = 0; e < 64; ++e) acc_x = __builtin_fma(jv[e], rhs[c0 + e], acc_x);
              }
            }
            __syncthreads();                                 // every thread has read rhs
            if (tid < NP) rhs[tid] = acc_x;
            __syncthreads();
            stp.hit(6);
            skip_subst = true;
          }
        }
        if (!skip_subst) {
          for (int i = 1; i < NB; ++i) gd_matvec64_sum(rhs + DB * i, gd_blk(fs, NB, 0, i, 0), (long)DB * DB, rhs, DB, i);
          stp.hit(5);
        }
      } else {
      if (A.mode) scr = A.flat;
      // ---- block LU, left-looking over the panels
      if constexpr (BLOCKROW) {
      // (the first entry of the next block for this thread -- bus pair, place in the block, Ybus entry: 32 bytes -- is requested a block ahead)
      GsDenseEntry pre = A.bent[min(A.bent_ptr[0] + tid, A.bent_ptr[NB * NB] - 1)];
      for (int j = 0; j < NB; ++j)
        for (int i = 0; i < NB; ++i) {
          double* buf = i == j ? bufD : bufC;
          const int q0 = A.bent_ptr[j * NB + i], q1 = A.bent_ptr[j * NB + i + 1];
          GsDenseEntry en = pre;
          {
            const int nb = j * NB + i + 1;
            if (nb < NB * NB) pre = A.bent[min(A.bent_ptr[nb] + tid, A.bent_ptr[NB * NB] - 1)];
          }
          __syncthreads();                                  // (whoever read this buffer last is through)
          for (int k = tid; k < DB * (DLD / 2); k += blockDim.x) ((double2*)buf)[k] = make_double2(0.0, 0.0);
          gd_sync_lds();
          // Jacobian blocks of block (i, j) (power_flow.py:243-287); padding unknowns get a unit diagonal
          for (int q = q0 + tid; q < q1; q += blockDim.x) {
            if (q != q0 + tid) en = A.bent[q];
            const int ib = en.ib, jb = en.jb;
            const int thi = (en.dst >> 16) & 1, vfi = (en.dst >> 17) & 1, thj = (en.dst >> 18) & 1, vfj = (en.dst >> 19) & 1;
            double b00, b01, b10, b11;
            if (ib == jb) {
              const double gd = en.g, bd = en.b, v = vm[ib], P = pc[ib], Q = qc[ib];
              const double vvb = v * v * bd;
              b00 = thi ? (A.jacobian_exact ? (-Q - vvb) : (-Q + vvb)) : 1.0;
              b01 = (thi && vfi) ? (P / v + v * gd) : 0.0;
              b10 = (thi && vfi) ? (P - v * v * gd) : 0.0;
              b11 = vfi ? (Q / v - v * bd) : 1.0;
            } else {
              const double gg = en.g, bb0 = en.b;
              const double a = ve[ib] * ve[jb] + vf[ib] * vf[jb];
              const double bb = vf[ib] * ve[jb] - ve[ib] * vf[jb];
              const double gs_bc = gg * bb - bb0 * a, gc_bs = gg * a + bb0 * bb;
              b00 = (thi && thj) ? gs_bc : 0.0;
              b01 = (thi && vfj) ? gc_bs / vm[jb] : 0.0;
              b10 = (vfi && thj) ? -gc_bs : 0.0;
              b11 = (vfi && vfj) ? gs_bc / vm[jb] : 0.0;
            }
            double* d0 = buf + (en.dst & 0xffff);
            d0[0] = b00; d0[1] = b01; d0[DLD] = b10; d0[DLD + 1] = b11;
          }
          if (i == j)
            for (int u = max(2 * na, DB * j) + tid; u < DB * (j + 1); u += blockDim.x) buf[(size_t)(u - DB * j) * DLD + (u - DB * j)] = 1.0;
          gd_sync_lds();
          stp.hit(1);
          gd_v4 acc[4];
          gd_acc_load(acc, buf, DLD);
          const int kmax = i < j ? i : j;
          // -= sum_k L_ik U_kj.  U_kj goes through bufC (free once the accumulators are loaded; every thread brings 16 of its doubles),
          // L_ik straight into the A operands; U of product k + 1 is requested before product k's MFMAs are issued.
          if (kmax > 0) {
            const int urow = tid >> 2, ucol = 16 * (tid & 3);
            gd_v2 un[8];
            {
              const double* U = gd_blk(scr, NB, 1, 0, j) + (size_t)urow * DB + ucol;
#pragma unroll
              for (int c = 0; c < 8; ++c) un[c] = *(const gd_v2*)(U + 2 * c);
            }
            for (int k = 0; k < kmax; ++k) {
              double ac[16];
              gd_load_a(ac, gd_blk(scr, NB, 0, i, k), DB);     // (in flight across the two barriers: they wait for LDS only)
              gd_sync_lds();                                // bufC: the accumulators are loaded / the previous product has read its B operands
#pragma unroll
              for (int c = 0; c < 8; ++c) *(gd_v2*)(bufC + (size_t)urow * DLD + ucol + 2 * c) = un[c];
              gd_sync_lds();
              if (k + 1 < kmax) {
                const double* U = gd_blk(scr, NB, 1, k + 1, j) + (size_t)urow * DB + ucol;
#pragma unroll
                for (int c = 0; c < 8; ++c) un[c] = *(const gd_v2*)(U + 2 * c);
              }
              gd_mfma_acc<true>(acc, ac, bufC, DLD);
            }
          }
          stp.hit(2);
          if (i < j) {                 // U_ij: straight to the scratch (the updates of the block rows below read it from there)
            gd_acc_store(acc, gd_blk(scr, NB, 1, i, j), DB);
            __threadfence_block();
            stp.hit(3);
          } else if (i == j) {         // diagonal block: inverse in place, kept in bufD for this panel's L blocks and in the scratch
            if (kmax > 0) gd_acc_store(acc, buf, DLD);          // (a wavefront's own rows)
            __syncthreads();
            stp.hit(2);
            sing |= gd_invert64_mfma(bufD, DLD, gjbuf);
            stp.hit(4);
            double* Dv = gd_blk(scr, NB, 2, j, 0);
            for (int e = tid; e < DB * DB; e += blockDim.x) Dv[e] = bufD[(size_t)(e >> 6) * DLD + (e & 63)];
            __threadfence_block();
          } else {                     // L_ij = C D_j^-1, in place and to the scratch; the right-hand side follows: r_i -= L_ij r_j
            if (kmax > 0) { __syncthreads(); gd_acc_store(acc, buf, DLD); }
            gd_gemm64<false>(bufC, DLD, bufC, DLD, bufD, DLD);
            __syncthreads();
            double* Lij = gd_blk(scr, NB, 0, i, j);
            for (int e = tid; e < DB * DB; e += blockDim.x) Lij[e] = bufC[(size_t)(e >> 6) * DLD + (e & 63)];
            gd_matvec64<false>(rhs + DB * i, bufC, DLD, rhs + DB * j);
            __threadfence_block();
            stp.hit(5);
          }
        }
      __threadfence_block();
      __syncthreads();
      } else {
      for (int j = 0; j < NB; ++j) {
        for (int k = tid; k < NP * (DLD / 2); k += blockDim.x) ((double2*)panel)[k] = make_double2(0.0, 0.0);
        __syncthreads();
        // Jacobian blocks of the panel's columns (power_flow.py:243-287); padding unknowns get a unit diagonal
        for (int q = A.ent_ptr[j] + tid; q < A.ent_ptr[j + 1]; q += blockDim.x) {
          const int i = A.ent[3 * q], jb = A.ent[3 * q + 1], pos = A.ent[3 * q + 2];
          const int ai = A.act_of[i], aj = A.act_of[jb];
          const int thi = A.th_free[i], vfi = A.vm_free[i];
          double b00, b01, b10, b11;
          if (i == jb) {
            const double gd = A.Gd[i], bd = A.Bd[i], v = vm[i], P = pc[i], Q = qc[i];
            const double vvb = v * v * bd;
            b00 = thi ? (A.jacobian_exact ? (-Q - vvb) : (-Q + vvb)) : 1.0;
            b01 = (thi && vfi) ? (P / v + v * gd) : 0.0;
            b10 = (thi && vfi) ? (P - v * v * gd) : 0.0;
            b11 = vfi ? (Q / v - v * bd) : 1.0;
          } else {
            const double gg = A.G[pos], bb0 = A.Bv[pos];
            const double a = ve[i] * ve[jb] + vf[i] * vf[jb];
            const double bb = vf[i] * ve[jb] - ve[i] * vf[jb];
            const double gs_bc = gg * bb - bb0 * a, gc_bs = gg * a + bb0 * bb;
            const int thj = A.th_free[jb], vfj = A.vm_free[jb];
            b00 = (thi && thj) ? gs_bc : 0.0;
            b01 = (thi && vfj) ? gc_bs / vm[jb] : 0.0;
            b10 = (vfi && thj) ? -gc_bs : 0.0;
            b11 = (vfi && vfj) ? gs_bc / vm[jb] : 0.0;
          }
          const int c0 = 2 * aj - DB * j;
          double* d0 = panel + (size_t)(2 * ai) * DLD + c0;
          d0[0] = b00; d0[1] = b01; d0[DLD] = b10; d0[DLD + 1] = b11;
        }
        for (int u = max(2 * na, DB * j) + tid; u < DB * (j + 1); u += blockDim.x) panel[(size_t)u * DLD + (u - DB * j)] = 1.0;
        __syncthreads();
        stp.hit(1);
        // updates from the earlier panels: C_i -= L_ik C_k, i > k (C_k itself is final once the panels before k are through)
        // (the A operands of the NEXT product -- a block of L in global memory -- are requested before the current one's MFMAs run)
        if (j > 0) {
          double an[16];
          gd_load_a(an, gd_blk(scr, NB, 0, 1, 0), DB);
          for (int k = 0; k < j; ++k)
            for (int i = k + 1; i < NB; ++i) {
              double ac[16];
#pragma unroll
              for (int s = 0; s < 16; ++s) ac[s] = an[s];
              int in = i + 1, kn = k;
              if (in >= NB) { kn = k + 1; in = kn + 1; }
              if (kn < j && in < NB) gd_load_a(an, gd_blk(scr, NB, 0, in, kn), DB);
              gd_gemm64_a<true>(panel + (size_t)(DB * i) * DLD, DLD, ac, panel + (size_t)(DB * k) * DLD, DLD);
              __syncthreads();
            }
        }
        stp.hit(2);
        // U blocks of this panel column to the scratch (rows of the panels before j)
        for (int k = 0; k < j; ++k) {
          double* U = gd_blk(scr, NB, 1, k, j);
          for (int e = tid; e < DB * DB; e += blockDim.x) U[e] = panel[(size_t)(DB * k + (e >> 6)) * DLD + (e & 63)];
        }
        stp.hit(3);
        // diagonal block: inverse in place, kept in the scratch for the back substitution
        sing |= gd_invert64(panel + (size_t)(DB * j) * DLD, DLD, gjbuf);
        stp.hit(4);
        {
          double* Dv = gd_blk(scr, NB, 2, j, 0);
          for (int e = tid; e < DB * DB; e += blockDim.x) Dv[e] = panel[(size_t)(DB * j + (e >> 6)) * DLD + (e & 63)];
        }
        // L_ij = C_i D_j^-1 for the block rows below, in place and to the scratch; the right-hand side follows: r_i -= L_ij r_j
        for (int i = j + 1; i < NB; ++i) {
          double* Ci = panel + (size_t)(DB * i) * DLD;
          gd_gemm64<false>(Ci, DLD, Ci, DLD, panel + (size_t)(DB * j) * DLD, DLD);
          __syncthreads();
          double* Lij = gd_blk(scr, NB, 0, i, j);
          for (int e = tid; e < DB * DB; e += blockDim.x) Lij[e] = Ci[(size_t)(e >> 6) * DLD + (e & 63)];
          gd_matvec64<false>(rhs + DB * i, Ci, DLD, rhs + DB * j);
        }
        __threadfence_block();
        __syncthreads();
        stp.hit(5);
      }
      }
      if (A.mode) {        // the handle's flat-start factors are in place; the flag behind them says whether a pivot was singular
        sing = gd_block_max(sing ? 1.0 : 0.0, red) != 0.0 ? 1 : 0;
        if (tid == 0) A.flat[(size_t)NB * NB * DB * DB] = sing ? 1.0 : 0.0;
        return;
      }
      }
      // ---- back substitution: x_j = D_j^-1 (r_j - sum_{k > j} U_jk x_k); x overwrites r block by block
      if (!skip_subst)
        for (int j = NB - 1; j >= 0; --j)
          gd_backstep(rhs + DB * j, gd_blk(fs, NB, 1, j, j + 1), rhs + DB * (j + 1), NB - 1 - j, gd_blk(fs, NB, 2, j, 0), gjbuf);
      stp.hit(6);
      sing = gd_block_max(sing ? 1.0 : 0.0, red) != 0.0 ? 1 : 0;
      if (sing) { status = GS_STATUS_SINGULAR; break; }      // power_flow.py:188-190: keep the current voltages
      // ---- corrections (power_flow.py:297-327) and the new rectangular voltages
      for (int i = tid; i < n; i += blockDim.x) {
        const int a_ = A.act_of[i];
        if (a_ < 0) continue;
        double v = vm[i], th = va[i];
        if (A.th_free[i]) th += A.alpha * rhs[2 * a_];
        if (A.vm_free[i]) v += A.alpha * rhs[2 * a_ + 1];
        if (v < 0.0) { v = -v; th += M_PI; }
        double sn, cs;
        sincos(th, &sn, &cs);
        vm[i] = v; va[i] = th; ve[i] = v * cs; vf[i] = v * sn;
      }
      __syncthreads();
      stp.hit(7);
      stale = true;
    }
    if (stale) {      // iteration cap reached after an update: P / Q calculated at the final voltages (the epilogue's losses)
      for (int i = tid; i < n; i += blockDim.x) {
        const double ei = ve[i], fi = vf[i];
        double P = 0.0, Q = 0.0;
        for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p) {
          const int j = A.col[p];
          const double a = ei * ve[j] + fi * vf[j], bb = fi * ve[j] - ei * vf[j];
          P += A.G[p] * a + A.Bv[p] * bb;
          Q += A.G[p] * bb - A.Bv[p] * a;
        }
        pc[i] = P; qc[i] = Q;
      }
      __syncthreads();
    }
    // ---- what newton_loop leaves in the rows
    for (int i = tid; i < n; i += blockDim.x) {
      row(R.VM.base + 2 * i) = vm[i]; row(R.VA.base + 2 * i) = va[i];
      row(R.E.base + 2 * i) = ve[i]; row(R.F.base + 2 * i) = vf[i];
      row(R.PC.base + 2 * i) = pc[i]; row(R.QC.base + 2 * i) = qc[i];
    }
    if (tid == 0) { row(R.MAXMIS) = mm; row(R.ITERS) = (double)iters; row(R.CONV) = (double)conv; row(R.STATUS) = (double)status; }
    __syncthreads();
    stp.hit(8);
  }
}

#if defined(GS_BUILD_EXPERIMENTS)      // the panel form, kept for A/B measurements (GS_DENSE_PANEL=1 in a library built with `make EXPERIMENTS=1`)
extern "C" __global__ void __launch_bounds__(256)
gs_k_nr_dense_mfma(GsDenseArgs A_, double* __restrict__ slab_, int B_) { gd_newton<false>(A_, slab_, B_); }
#endif

extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gs_k_nr_dense_mfma2(GsDenseArgs A_, double* __restrict__ slab_, int B_) { gd_newton<true>(A_, slab_, B_); }
