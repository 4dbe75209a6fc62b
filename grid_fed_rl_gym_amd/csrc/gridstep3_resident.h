// gridstep3_resident.h -- three-phase forward/backward sweep with the whole instance resident in one CU (included by
// gridstep3.hip; same gs3_* ABI, same answer as gs3_k_solve to rounding).
//
// gs3_k_solve walks the 60 levels of the 8500-node feeder one barrier at a time and streams three state rows per sweep: a
// solve is a chain of ~400 dependent level steps, each paying an HBM round trip that four resident workgroups hide only
// partly (0.6 ms per 1024 solves, 43 % of HBM peak on the bytes of SURVEY.md 8(d)).  This kernel removes both the chain
// and the stream.  The sweeps are tree reductions, and a tree reduction over a depth-first numbering is a prefix sum:
//
//   positions   the conductors of phase a in depth-first PREORDER of the phase-a tree, then phase b, then phase c.  The
//               subtree of a conductor is a contiguous range [p, e(p)) of its phase block, and so is its set of descendants
//               in POSTORDER, [g(p), post(p)).
//   backward    line current J_p = sum of the injection currents I over the subtree = X[e(p)] - X[p], X = exclusive prefix
//               sum of I over positions.  (The prefix runs across the phase blocks; the blocks before p cancel in the
//               difference.)  D_p = Z_p . (J_p, J_sibling A, J_sibling B).
//   forward     V_p = V_source - sum of D over the ancestors-or-self of p.  a is an ancestor-or-self of p exactly when it
//               comes no later than p in preorder and no earlier in postorder; the conductors that come before p in BOTH
//               orders are those before p's first descendant in postorder, so
//               sum over ancestors-or-self = Xi[p] - Y[g(p)],   Xi = inclusive preorder prefix of D, Y = exclusive
//               postorder prefix of D: one sample at the thread's own position and one gather at a static index.
//
// So an iteration is three workgroup-wide prefix sums and three gathers, with no dependence on the depth of the feeder.  One
// workgroup of up to 512 threads solves one instance; a thread owns K consecutive positions (K up to 19: the arrays of a
// thread take 9 K registers, what is left of the 256 a wave may have at two waves per SIMD are the temporaries -- with 1024
// threads and 128 registers the compiler spilled the currents to scratch and the kernel waited for them one by one).  What persists between
// iterations is the injection current of each conductor, in registers (2 K doubles per thread; the mismatch of the new
// voltages needs exactly that current, see gs3_k_solve); the array being gathered from (X, then J, then D / Y) is the one
// thing that has to be visible to other threads and lives in LDS: (conductors + 1) x 16 bytes, 150 KB of the CU's 160 KB for
// the 8500-node case.  D is stored at the conductor's POSTORDER index, so the postorder prefix runs over a thread's own
// contiguous entries, in place.  All a position needs to know of the tree fits in one 32-bit word -- subtree size, postorder
// index, phase, "is the source" -- kept in registers: e = p + size, g = post - size + 1.  The mutual terms of D exist on the
// ~17 % of conductors that share a node with another phase; they are a separate compact list (conductor, its two
// siblings, the two impedances) dealt over the threads, added to D through LDS -- or, where most conductors have
// siblings, computed by every position for itself (template parameter MK = 0).  Per iteration HBM sees S (150 KB per
// workgroup; the working set of the 256 resident workgroups fits the MALL) and, per solve, V once; the self impedances
// (16 bytes per conductor) come from L2.  Feeders whose conductors do not fit (more than ~9 700: 19 x 512) take
// gs3_k_solve.
//
// Rounding: prefix differences carry an absolute error of a few ulp of the LARGEST prefix (the feeder's total current, the
// summed drops of a phase) instead of the subtree's own sum: ~1e-15 in V on the 8500-node case, far inside the 1e-10 the
// tests hold the two kernels and the oracles to.  The convergence test, the iteration count and the loss accounting follow
// gs3_k_solve line by line.
#pragma once

struct Res3 {
  int32_t ns, npad, K, M;      // conductors; K x threads; positions per thread; entries of the mutual list
  const int32_t* pk;           // [npad], storage order: subtree size | postorder index << 14 | is-source << 28 | phase << 29
  const double2* zd;           // [npad], storage order: self impedance of the upstream line (zero for the source's conductors)
  const int4* mut;             // [M]: {a | e(a) << 14, b | e(b) << 14, postorder index of the conductor, 0}, a / b = positions of the
                               //      node's other two conductors; an absent one is (0, 0) with a zero impedance
  const double2* mz;           // [2][M]: mutual impedances to A and to B
  // the same per POSITION, for feeders where most conductors have siblings (the list would be one entry per position):
  const int2* mutp;            // [npad], storage order: {a | e(a) << 14, b | e(b) << 14}
  const double2* mzp;          // [2][npad], storage order
  // Positions ns .. npad - 1 are padding and behave like conductors that draw nothing and hang nowhere: S = 0, Z = 0,
  // subtree size 0, postorder index = position.  Then no access needs a guard: X of a padding position is the total, its
  // J and D are 0, and its V comes out as the source voltage.
  double vsr[3], vsi[3];
  int32_t off[4];              // first position of each phase block, ns: the source's conductor of phase ph is position off[ph]
  size_t stride;               // double2 entries between the rows (V, S) of consecutive instances
  long long* stamps;           // development: clock of workgroup 0 at the phase boundaries of its second iteration (GS3_STAMPS=1), or NULL
};
// storage order: element k of thread t at k * threads + t (one coalesced access per k)
#define R3_SIZE(w) ((w) & 0x3fff)
#define R3_POST(w) (((w) >> 14) & 0x3fff)
#define R3_ROOT(w) (((w) >> 28) & 1)
#define R3_PH(w) ((unsigned)(w) >> 29)
#define GS3_RESIDENT_MAX_CONDUCTORS 16383

template <int CTRL, int ROWS> __device__ __forceinline__ double r3_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 16 lanes of a row
__device__ __forceinline__ double r3_row_scan(double v) {
  v += r3_dpp<0x111, 0xf>(v);      // row_shr:1
  v += r3_dpp<0x112, 0xf>(v);      // row_shr:2
  v += r3_dpp<0x114, 0xf>(v);      // row_shr:4
  v += r3_dpp<0x118, 0xf>(v);      // row_shr:8
  return v;
}
// ... over the 64 lanes of a wavefront
__device__ __forceinline__ double r3_wave_scan(double v) {
  v = r3_row_scan(v);
  v += r3_dpp<0x142, 0xa>(v);      // row_bcast:15 -> rows 1 and 3 take lane 15 of the row before
  v += r3_dpp<0x143, 0xc>(v);      // row_bcast:31 -> rows 2 and 3 take lane 31
  return v;
}
__device__ __forceinline__ double r3_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// 1 / d for a normal d (|V|^2 here): the hardware estimate and two Newton steps, ~1 ulp.  (A true division is twelve
// instructions and a dozen temporaries per position, times the positions the scheduler interleaves.)
__device__ __forceinline__ double r3_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}
__device__ __forceinline__ void r3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct R3Wg {
  double2* scr;        // [2][16] wave totals of the prefix sums, [2][16] of the reductions; zeroed at start
  int par, rpar, wave, lane;

  // in: the thread's total; out: the sum over the threads before it (exclusive), and the workgroup's total
  __device__ __forceinline__ void scan(double& re, double& im, double& tr, double& ti) {
    const double sr = r3_wave_scan(re), si = r3_wave_scan(im);
    if (lane == 63) scr[par * 16 + wave] = make_double2(sr, si);
    r3_barrier();
    const double2 t = scr[par * 16 + (lane & 15)];
    const double wr = r3_row_scan(t.x), wi = r3_row_scan(t.y);
    tr = r3_readlane(wr, 15); ti = r3_readlane(wi, 15);
    double or_ = 0.0, oi = 0.0;
    if (wave > 0) { or_ = r3_readlane(wr, wave - 1); oi = r3_readlane(wi, wave - 1); }
    re = or_ + sr - re; im = oi + si - im;
    par ^= 1;
  }
  // max of m and sum of s over the workgroup
  __device__ __forceinline__ void max_sum(double& m, double& s) {
    for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o)); s += __shfl_xor(s, o); }
    if (lane == 0) scr[32 + rpar * 16 + wave] = make_double2(m, s);
    r3_barrier();
    const double2 t = scr[32 + rpar * 16 + (lane & 15)];     // entries of absent waves: (0, 0); m >= 0 always
    m = t.x; s = t.y;
    for (int o = 8; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o)); s += __shfl_xor(s, o); }
    rpar ^= 1;
  }
};

extern __shared__ double2 r3_lds[];

// K positions per thread (odd: a thread's 16-byte LDS accesses, K x 16 bytes apart across lanes, then fall on distinct banks);
// MK entries of the mutual list per thread (4: the usual feeder, mostly single-phase laterals); MK = 0: no list, every
// position computes its own mutual term from the per-position tables (feeders where most nodes are multi-phase)
template <int K, int MK>
__global__ void __launch_bounds__(512)
gs3_k_resident(Res3 T, double2* __restrict__ state, int B, double tol, int max_it, double* __restrict__ out_loss,
               double* __restrict__ out_mm, int32_t* __restrict__ out_it, uint8_t* __restrict__ out_conv) {
  // what is kept in flight ahead of its use: self impedances, injections (16-byte global loads); gathers from LDS per
  // scheduling fence; positions of the mismatch loop interleaved.  (With 256 threads, K = 37 and one wave per SIMD the
  // arrays spill over into AGPRs and there is room for rings of 8 -- measured 3.4 M solves/s against 4.3 M for this form.)
  constexpr int GZ = 2, GS = 2, FA = 4, FB = 2;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int npad = T.npad, M = T.M, nt = blockDim.x;
  int tid = threadIdx.x;
  double2* __restrict__ A = r3_lds;                                  // [npad + 1]: X, then J, then D / Y
  R3Wg wg = {r3_lds + ((npad + 4) & ~3), 0, 0, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
  if (tid < 64) wg.scr[tid] = make_double2(0.0, 0.0);
  double2* __restrict__ srcv = wg.scr + 64;                          // [3] source voltage by phase: one LDS read where a position needs it
  if (tid < 3) srcv[tid] = make_double2(T.vsr[tid], T.vsi[tid]);
  double2* __restrict__ Vrow = state + (size_t)blockIdx.x * T.stride;
  const double vsr0 = T.vsr[0], vsr1 = T.vsr[1], vsr2 = T.vsr[2], vsi0 = T.vsi[0], vsi1 = T.vsi[1], vsi2 = T.vsi[2];
#define R3_SRC_R(ph) ((ph) == 0 ? vsr0 : ((ph) == 1 ? vsr1 : vsr2))
#define R3_SRC_I(ph) ((ph) == 0 ? vsi0 : ((ph) == 1 ? vsi1 : vsi2))
  // Global rows through buffer descriptors: the descriptor and the row offset k * threads sit in SGPRs, the thread's
  // offset is one VGPR for all rows (a 64-bit address per thread and row would cost 2 K registers per table).  An entry
  // beyond the end of a table reads as zero.
  const __amdgpu_buffer_rsrc_t rs_pk = __builtin_amdgcn_make_buffer_rsrc((void*)T.pk, 0, npad * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_zd = __builtin_amdgcn_make_buffer_rsrc((void*)T.zd, 0, npad * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(Vrow + npad), 0, npad * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vrow, 0, npad * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mut = __builtin_amdgcn_make_buffer_rsrc((void*)T.mut, 0, M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mz = __builtin_amdgcn_make_buffer_rsrc((void*)T.mz, 0, 2 * M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mutp = __builtin_amdgcn_make_buffer_rsrc((void*)T.mutp, 0, MK == 0 ? npad * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mzp = __builtin_amdgcn_make_buffer_rsrc((void*)T.mzp, 0, MK == 0 ? 2 * npad * 16 : 0, 0x00020000);
#define R3_LD16(rs, k) __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)tid << 4, (k) * nt * 16, 0))
#define R3_LD4(rs, k) ((int)__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)tid << 2, (k) * nt * 4, 0))
  // What follows from the thread index and from pk -- addresses, indices, the source voltage of a position's phase -- is
  // recomputed where it is used: the compiler would otherwise hoist all of it out of the iteration loop, a dozen registers
  // per position, and spill.
#define R3_OPAQUE() do { asm volatile("" : "+v"(tid)); _Pragma("unroll") for (int k = 0; k < K; ++k) asm volatile("" : "+v"(pk[k])); } while (0)
// The scheduler would issue all K gathers of a loop at once (four result registers each) and the allocator then spills the
// arrays: a fence after every n positions bounds what is in flight.
#define R3_EVERY(k, n) do { if ((k) % (n) == (n) - 1) __builtin_amdgcn_sched_barrier(0); } while (0)
// ... and the arithmetic on what was gathered has to stay in front of the fence too (pure register work otherwise sinks
// below the next barrier, and the K gathered operands wait for it in registers)
#define R3_PIN(x, y) asm volatile("" : "+v"(x), "+v"(y))
#define R3_OWN (A + tid * K)                  /* the thread's K entries: one address register, the rest immediate offsets */

#define R3_STAMP0(i) do { if (T.stamps && blockIdx.x == 0 && tid == 0) T.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
  R3_STAMP0(10);
  int pk[K];                // the tree as this thread's positions see it
  double ar[K], ai[K];      // injection currents of the thread's conductors: what persists between iterations
  double br[K], bi[K];      // X -> J -> D, then Xi -> V
  double lmax = 0.0;
  r3_barrier();             // scratch zeroed
  // ---- flat start: V = V_source everywhere, first mismatch = |S_spec|
  const double rd0 = r3_rcp(vsr0 * vsr0 + vsi0 * vsi0), rd1 = r3_rcp(vsr1 * vsr1 + vsi1 * vsi1), rd2 = r3_rcp(vsr2 * vsr2 + vsi2 * vsi2);
  // (all 2 K loads of the thread in flight at once -- S lands in the registers that take the voltages afterwards; four at a
  // time behind fences, as everywhere else, this start-up was five HBM round trips = 10 of a workgroup's 54 us)
#pragma unroll
  for (int k = 0; k < K; ++k) {
    pk[k] = R3_LD4(rs_pk, k);
    const double2 s = R3_LD16(rs_s, k);                           // (zero at the source's own conductors: gs3_k_scatter_in skips them)
    br[k] = s.x; bi[k] = s.y;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int ph = R3_PH(pk[k]);
    const double sx = br[k], sy = bi[k];
    const double dP = fabs(sx), dQ = fabs(sy);
    lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
    const double vr = R3_SRC_R(ph), vi = R3_SRC_I(ph), rd = ph == 0 ? rd0 : (ph == 1 ? rd1 : rd2);
    ar[k] = -(sx * vr + sy * vi) * rd; ai[k] = -(sx * vi - sy * vr) * rd;
    br[k] = vr; bi[k] = vi;
    R3_PIN(ar[k], ai[k]);
    R3_PIN(br[k], bi[k]);
    R3_EVERY(k, 4);
  }
  double mm, losses = 0.0, zero = 0.0;
  R3_STAMP0(11);
  mm = lmax; wg.max_sum(mm, zero);
  R3_STAMP0(12);
  int iters = max_it, conv = 0;
  if (!(mm < INFINITY) || mm < tol) {       // no sweep will follow: the answer is the flat start itself
    iters = 1; conv = mm < tol;
  } else {
    for (int it = 0; it < max_it; ++it) {
      R3_OPAQUE();
#define R3_STAMP(i) do { if (T.stamps && it == 1 && blockIdx.x == 0 && tid == 0) T.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
      R3_STAMP(0);
      // ---- backward: X = exclusive prefix of I over positions
      double rr = 0.0, ri = 0.0, tr, ti;
#pragma unroll
      for (int k = 0; k < K; ++k) { br[k] = rr; bi[k] = ri; rr += ar[k]; ri += ai[k]; }
      wg.scan(rr, ri, tr, ti);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        br[k] += rr; bi[k] += ri;
        R3_OWN[k] = make_double2(br[k], bi[k]);
      }
      if (tid == 0) A[npad] = make_double2(tr, ti);
      // (requested before the barrier, used after it) the mutual list's entries of this thread, the first self impedances
      constexpr int MKA = MK > 0 ? MK : 1;
      int4 mx[MKA]; double2 mza[MKA], mzb[MKA];
      int2 pq[GZ]; double2 paq[GZ], pbq[GZ];                    // (MK = 0) ring of per-position sibling words and mutual impedances
      if (MK == 0) {
#pragma unroll
        for (int k = 0; k < GZ; ++k) {
          pq[k] = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(rs_mutp, (unsigned)tid << 3, k * nt * 8, 0));
          paq[k] = R3_LD16(rs_mzp, k);
          pbq[k] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs_mzp, (unsigned)tid << 4, (npad + k * nt) * 16, 0));
        }
      }
#pragma unroll
      for (int j = 0; j < MK; ++j) {
        mx[j] = __builtin_bit_cast(int4, __builtin_amdgcn_raw_buffer_load_b128(rs_mut, (unsigned)tid << 4, j * nt * 16, 0));
        mza[j] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs_mz, (unsigned)tid << 4, j * nt * 16, 0));
        mzb[j] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs_mz, (unsigned)tid << 4, (M + j * nt) * 16, 0));
      }
      double2 zq[GZ];
#pragma unroll
      for (int k = 0; k < GZ; ++k) zq[k] = R3_LD16(rs_zd, k);
      r3_barrier();
      R3_STAMP(1);
      // the source's share of sum P_calc: V_source . (J of the phase's root = the phase's total current), by one thread
      double psrc = 0.0;
      if (tid == 0) {
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          const double2 x1 = A[T.off[ph + 1]], x0 = A[T.off[ph]];
          psrc += R3_SRC_R(ph) * (x1.x - x0.x) + R3_SRC_I(ph) * (x1.y - x0.y);
        }
      }
      // J = X[e] - X[own], and at once D = Z-row . (J, J of the node's other conductors): the own part here, ...
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double2 xe = R3_OWN[k + R3_SIZE(pk[k])];
        const double2 zd = zq[k % GZ];
        if (k + GZ < K) zq[k % GZ] = R3_LD16(rs_zd, k + GZ);
        const double jr = xe.x - br[k], ji = xe.y - bi[k];
        br[k] = zd.x * jr - zd.y * ji; bi[k] = zd.x * ji + zd.y * jr;
        if (MK == 0) {                       // the siblings' J as differences of X, like the own
          const int2 w = pq[k % GZ]; const double2 za = paq[k % GZ], zb = pbq[k % GZ];
          if (k + GZ < K) {
            pq[k % GZ] = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(rs_mutp, (unsigned)tid << 3, (k + GZ) * nt * 8, 0));
            paq[k % GZ] = R3_LD16(rs_mzp, k + GZ);
            pbq[k % GZ] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs_mzp, (unsigned)tid << 4, (npad + (k + GZ) * nt) * 16, 0));
          }
          const double2 a1 = A[(unsigned)w.x >> 14], a0 = A[w.x & 0x3fff], b1 = A[(unsigned)w.y >> 14], b0 = A[w.y & 0x3fff];
          const double jar = a1.x - a0.x, jai = a1.y - a0.y, jbr = b1.x - b0.x, jbi = b1.y - b0.y;
          br[k] += za.x * jar - za.y * jai + zb.x * jbr - zb.y * jbi;
          bi[k] += za.x * jai + za.y * jar + zb.x * jbi + zb.y * jbr;
        }
        R3_PIN(br[k], bi[k]);
        R3_EVERY(k, MK == 0 ? 2 : FA);
      }
      // ... the mutual part from the compact list, the siblings' J again as differences of X
      double mr[MKA], mi[MKA];
#pragma unroll
      for (int j = 0; j < MK; ++j) {
        const double2 a1 = A[(unsigned)mx[j].x >> 14], a0 = A[mx[j].x & 0x3fff], b1 = A[(unsigned)mx[j].y >> 14], b0 = A[mx[j].y & 0x3fff];
        const double jar = a1.x - a0.x, jai = a1.y - a0.y, jbr = b1.x - b0.x, jbi = b1.y - b0.y;
        mr[j] = mza[j].x * jar - mza[j].y * jai + mzb[j].x * jbr - mzb[j].y * jbi;
        mi[j] = mza[j].x * jai + mza[j].y * jar + mzb[j].x * jbi + mzb[j].y * jbr;
        R3_PIN(mr[j], mi[j]);
        R3_EVERY(j, 2);
      }
      r3_barrier();                          // every X has been read
      R3_STAMP(2);
      // ---- forward: D at the conductor's postorder index
      R3_OPAQUE();
#pragma unroll
      for (int k = 0; k < K; ++k) A[R3_POST(pk[k])] = make_double2(br[k], bi[k]);
      if (MK > 0 && M > 0) {
        r3_barrier();
#pragma unroll
        for (int j = 0; j < MK; ++j)
          if (tid + j * nt < M) { double2 d = A[mx[j].z]; d.x += mr[j]; d.y += mi[j]; A[mx[j].z] = d; }
        r3_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) { const double2 d = A[R3_POST(pk[k])]; br[k] = d.x; bi[k] = d.y; }
      }
      R3_STAMP(3);
      // Xi = inclusive preorder prefix of D (registers)
      rr = 0.0; ri = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) { rr += br[k]; ri += bi[k]; br[k] = rr; bi[k] = ri; }
      wg.scan(rr, ri, tr, ti);               // its barrier also publishes D
      R3_STAMP(4);
#pragma unroll
      for (int k = 0; k < K; ++k) { br[k] += rr; bi[k] += ri; }
      // Y = exclusive postorder prefix of D, in place: postorder index q is entry q
      double yr = 0.0, yi = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) { const double2 d = R3_OWN[k]; yr += d.x; yi += d.y; if (k % FA == FA - 1) R3_PIN(yr, yi); R3_EVERY(k, FA); }
      wg.scan(yr, yi, tr, ti);
      R3_STAMP(5);
      double2 sq[GS];
#pragma unroll
      for (int k = 0; k < GS; ++k) sq[k] = R3_LD16(rs_s, k);    // S of the first positions, needed after the gathers
#pragma unroll
      for (int k = 0; k < K; ++k) { const double2 d = R3_OWN[k]; R3_OWN[k] = make_double2(yr, yi); yr += d.x; yi += d.y; if (k % FA == FA - 1) R3_PIN(yr, yi); R3_EVERY(k, FA); }
      if (tid == 0) A[npad] = make_double2(tr, ti);
      r3_barrier();
      R3_STAMP(6);
      // V = V_source - (Xi[own] - Y[g])
      R3_OPAQUE();
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double2 y = A[R3_POST(pk[k]) - R3_SIZE(pk[k]) + 1], sv = srcv[R3_PH(pk[k])];
        br[k] = sv.x - (br[k] - y.x); bi[k] = sv.y - (bi[k] - y.y);
        R3_PIN(br[k], bi[k]);
        R3_EVERY(k, FA);
      }
      R3_STAMP(7);
      if (it + 1 >= max_it) break;           // the last sweep allowed: its voltages are the answer, nobody asks how good they are
      // the mismatch at the new voltages = what the backward sweep of iteration it + 1 would find, and the next injection currents
      lmax = 0.0;
      double psum = psrc;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double2 s = sq[k % GS];
        if (k + GS < K) sq[k % GS] = R3_LD16(rs_s, k + GS);
        const double wr = br[k], wi = bi[k];
        const double pc = -(wr * ar[k] + wi * ai[k]), qc = -(wi * ar[k] - wr * ai[k]);
        const double dP = fabs(s.x - pc), dQ = fabs(s.y - qc);
        lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
        psum += pc;
        const double rd = r3_rcp(wr * wr + wi * wi);
        ar[k] = -(s.x * wr + s.y * wi) * rd; ai[k] = -(s.x * wi - s.y * wr) * rd;
        R3_PIN(ar[k], ai[k]);
        R3_PIN(lmax, psum);
        R3_EVERY(k, FB);
      }
      R3_STAMP(8);
      mm = lmax; losses = psum;
      wg.max_sum(mm, losses);
      R3_STAMP(9);
      if (!(mm < INFINITY)) { iters = it + 2; break; }
      if (mm < tol) { iters = it + 2; conv = 1; break; }
    }
  }
  R3_STAMP0(13);
#pragma unroll
  for (int k = 0; k < K; ++k)                // 16-byte store: row offset in the vector offset (see GsPairRef::put, gs_internal.h)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_double2(br[k], bi[k])), rs_v, ((unsigned)tid << 4) + (unsigned)(k * nt * 16), 0, 0);
  if (tid == 0) {
    out_loss[blockIdx.x] = losses;
    out_mm[blockIdx.x] = mm;
    out_it[blockIdx.x] = iters;
    out_conv[blockIdx.x] = (uint8_t)conv;
  }
  R3_STAMP0(14);
#undef R3_STAMP0
#undef R3_SRC_R
#undef R3_SRC_I
#undef R3_LD16
#undef R3_LD4
#undef R3_OPAQUE
#undef R3_OWN
#undef R3_EVERY
#undef R3_PIN
#undef R3_STAMP
}
