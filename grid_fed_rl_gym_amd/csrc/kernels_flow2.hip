// kernels_flow2.hip -- the fused env.step() for radial feeders, second generation: IW instances per workgroup (16 by
// default; 32, or 8 for small feeders), the 64 / IW sub-groups of every wavefront on DIFFERENT buses; the sweep load flow
// as a prefix sum + pointer jumping, Newton-Raphson as a level-by-level 2x2-block elimination with its state in registers
// and LDS.  One template (f2_step), the members at the end of the file; the host (gridstep_abi.hip) picks one by feeder
// size, and launches a step of the two-workgroups-per-CU members as two half grids on two streams (gs_handle::split_ok).
// The text below describes the 32-instance sweep member the family started from.
//
// Why: the dataflow sweep kernel of kernels_solve.hip (gs_k_step_fbs_flow) gives a 64-instance group a whole compute
// unit (its per-bus LDS slots fill it), so BASELINE.json's config 3 -- 8192 instances -- ran on 128 of the chip's 256
// CUs, and per-group time is set by the number of instructions a wave issues, not by how many lanes are active.
// Here a workgroup owns half a slab group (32 instances), lanes 0..31 and 32..63 of a wavefront carry the same 32
// instances on two different buses / lines / load quads, so every bus-parallel phase issues half the instructions per
// workgroup and twice as many workgroups (256 at B = 8192) fill the chip.  Per-bus constants that used to be
// wave-uniform scalars are uniform per HALF and live in vector registers (GsF2Rec records, one vector load each).
//
// Also new against the first generation:
//  * LDS slots are [bus][33 x 16 B] (lane-interleaved (re, im) pairs, one ds_read_b128 / ds_write_b128 per message);
//    with the 33rd entry as padding the epilogue converts every slot IN PLACE to (|V|, angle) and the slots ARE the
//    transposed observation tile: the observation block is written from LDS, the |V| / angle / flow rows the epilogue
//    stores are never read back (that re-read was a third of the step's fabric traffic);
//  * load powers, renewable powers, curtailment, battery state go to the injection pass through LDS, not rows;
//  * cross-wave maxima / counts are LDS integer atomics on the bit patterns (exact, order-independent); only the two
//    floating-point SUMS (losses, voltage deviation) keep per-wave partials added in wave order.
//
// Reference arithmetic restated (paths relative to /root/reference/grid_fed_rl/environments/): as kernels_solve.hip:
// mismatch power_flow.py:150-171, line flows / losses :329-358 / :198-200, env step grid_env.py:410-619.
// The sweep itself is the one of fbs_loop_flow (kernels_solve.hip): same per-bus operations in the same order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <limits.h>
#include <type_traits>

#include "../../include/gridstep.h"
#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]
#include "env_device.h"

// No implicit contraction anywhere in this file: the plain and the checks-carrying instantiation of the kernel are
// compared bit for bit, and which multiply the compiler fuses into which add depends on the code around it.  Fused
// multiply-adds are written out where they are wanted.
#pragma clang fp contract(off)

extern __shared__ __attribute__((aligned(16))) char f2_lds[];

__device__ __forceinline__ void f2_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void f2_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Every LDS access of the kernel goes through an explicit address-space-3 pointer: the compiler does not rewrite
// VOLATILE generic accesses to LDS ones by itself (they would become flat instructions with 64-bit addresses).
#define F2_AS3 __attribute__((address_space(3)))
#define F2_P(type, off) ((type F2_AS3*)((F2_AS3 char*)f2_lds + (off)))
#define F2_VP(type, off) ((volatile type F2_AS3*)((F2_AS3 char*)f2_lds + (off)))
typedef double f2_v2 __attribute__((ext_vector_type(2)));        // 16 bytes: one ds_read_b128 / ds_write_b128
typedef int f2_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 f2_ld2(unsigned off) { const f2_v2 v = *F2_P(const f2_v2, off); return make_double2(v.x, v.y); }
__device__ __forceinline__ void f2_st2(unsigned off, double2 v) { f2_v2 w; w.x = v.x; w.y = v.y; *F2_P(f2_v2, off) = w; }
__device__ __forceinline__ double f2_ld(unsigned off) { return *F2_P(const double, off); }
// 16-byte store of data nobody on the GPU reads again (the observation block): non-temporal, so that 58 MB per step do not
// push the topology tables every workgroup reads at its start out of the L2
__device__ __forceinline__ void f2_stream2(double* p, double2 v) {
  f2_v2 w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, (f2_v2*)p);
}
__device__ __forceinline__ void f2_st(unsigned off, double v) { *F2_P(double, off) = v; }

// Keeps the compiler from hoisting everything derived from a per-item index (slot, table and ring addresses of 8 items x
// three phases: ~100 registers) out of the Newton loop: the addresses are a shift and an add away wherever they are used.
#define F2_OPAQUE(x) asm volatile("" : "+v"(x))
// a wave-uniform value the compiler must hold on to rather than read again from the kernel arguments
#define F2_KEEP(x) asm volatile("" : "+s"(x))
// LDS integer atomics (exact and order-independent; workgroup scope is all LDS needs)
#define atomicMax(p, v) __hip_atomic_fetch_max((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define atomicMin(p, v) __hip_atomic_fetch_min((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define atomicAdd(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define atomicOr(p, v) __hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ __forceinline__ double f2_xhalf(double v) { return __shfl_xor(v, 32); }      // the same instance's value in the other half
// Cross-sub-group exchanges without the LDS pipe (a __shfl of a double is two ds_bpermute_b32; the sweeps are bound by LDS
// instruction issue).  gfx950's swaps, with both operands the same register (tools/permlane_probe.hip checks this on the GPU):
//   f2_halves32(x) = (x of lane i mod 32, x of lane 32 + i mod 32)                        v_permlane32_swap
//   f2_rows16(x)   = (x of the EVEN 16-lane row of this lane's row pair, x of the ODD row) v_permlane16_swap
struct F2Two { double a, b; };
__device__ __forceinline__ F2Two f2_halves32(double x) {
  const uint2 u = __builtin_bit_cast(uint2, x);
  const auto lo = __builtin_amdgcn_permlane32_swap(u.x, u.x, false, false), hi = __builtin_amdgcn_permlane32_swap(u.y, u.y, false, false);
  return F2Two{__builtin_bit_cast(double, make_uint2(lo[0], hi[0])), __builtin_bit_cast(double, make_uint2(lo[1], hi[1]))};
}
__device__ __forceinline__ F2Two f2_rows16(double x) {
  const uint2 u = __builtin_bit_cast(uint2, x);
  const auto lo = __builtin_amdgcn_permlane16_swap(u.x, u.x, false, false), hi = __builtin_amdgcn_permlane16_swap(u.y, u.y, false, false);
  return F2Two{__builtin_bit_cast(double, make_uint2(lo[0], hi[0])), __builtin_bit_cast(double, make_uint2(lo[1], hi[1]))};
}
// the same instance's values in all HV sub-groups of the wave: maximum, and sum in a fixed order (the same on every lane:
// sub-group order for IW = 32, pairs first for IW = 16 and 8: (0 + 1) + (2 + 3) ...)
// the value of the lane 8 further on in this lane's row of 16 (rotation by 8: DPP row_ror, no LDS)
__device__ __forceinline__ double f2_ror8(double x) {
  const uint2 u = __builtin_bit_cast(uint2, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.x, 0x128, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.y, 0x128, 0xf, 0xf, false);
  return __builtin_bit_cast(double, make_uint2(lo, hi));
}
template <int IW> __device__ __forceinline__ double f2_xmax(double v) {
  if constexpr (IW == 32) { const F2Two t = f2_halves32(v); return fmax(t.a, t.b); }
  else if constexpr (IW == 16) { const F2Two r = f2_rows16(v); const F2Two t = f2_halves32(fmax(r.a, r.b)); return fmax(t.a, t.b); }
  else if constexpr (IW == 8) { return f2_xmax<16>(fmax(v, f2_ror8(v))); }
  else {
#pragma unroll
    for (int o = IW; o < 64; o <<= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
  }
}
// (IW = 8: ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)), pairs first -- eight LDS shuffles until round 3)
template <int IW> __device__ __forceinline__ double f2_xsum(double v, int l) {
  if constexpr (IW == 32) { const F2Two t = f2_halves32(v); return t.a + t.b; }
  else if constexpr (IW == 16) { const F2Two r = f2_rows16(v); const F2Two t = f2_halves32(r.a + r.b); return t.a + t.b; }
  else if constexpr (IW == 8) { return f2_xsum<16>(v + f2_ror8(v), l); }
  else {
    double s = __shfl(v, l);
#pragma unroll
    for (int k = 1; k < 64 / IW; ++k) s += __shfl(v, k * IW + l);
    return s;
  }
}
// exclusive prefix over the sub-groups before this one (in sub-group order) and the total of all of them
template <int IW> __device__ __forceinline__ void f2_xscan(double v, int l, int hv, double& pre, double& tot) {
  if constexpr (IW == 32) { const F2Two t = f2_halves32(v); pre = hv ? t.a : 0.0; tot = t.a + t.b; }
  else if constexpr (IW == 16) {
    const F2Two r = f2_rows16(v);                      // (sub-group 0 | 2, sub-group 1 | 3) of this lane's pair
    const F2Two t = f2_halves32(r.a + r.b);            // (0 + 1, 2 + 3)
    pre = ((hv & 2) ? t.a : 0.0) + ((hv & 1) ? r.a : 0.0);
    tot = t.a + t.b;
  } else {
    pre = 0.0; tot = 0.0;
#pragma unroll
    for (int k = 0; k < 64 / IW; ++k) { const double x = __shfl(v, k * IW + l); if (k < hv) pre += x; tot += x; }
  }
}
__device__ __forceinline__ unsigned long long f2_bits(double v) { return __builtin_bit_cast(unsigned long long, v); }
__device__ __forceinline__ double f2_dbl(unsigned long long b) { return __builtin_bit_cast(double, b); }

// per-lane rows (row index differs between the halves of a wave)
__device__ __forceinline__ GsRowRef f2_row(const GsLaneRows& S, int row) { return S.lane_row((size_t)row * GS_LANES); }
__device__ __forceinline__ GsPairRef f2_pair(const GsLaneRows& S, int even_row) {
  return GsPairRef{S.r, S.lane16 + (((unsigned)even_row >> 1) << 10), 0};
}

// 1 / x and 1 / sqrt(x) for normal, finite x of ordinary magnitude (voltages, determinants): the hardware estimate and two
// Newton steps, accurate to about an ulp.  The compiler's IEEE division is ~40 instructions, its square root ~30, and the
// Newton-Raphson item (three reciprocals, two square roots) was bound by exactly that instruction count.
__device__ __forceinline__ double f2_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double f2_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(0.5 * y, __builtin_fma(-(x * y), y, 1.0), y);
  return __builtin_fma(0.5 * y, __builtin_fma(-(x * y), y, 1.0), y);
}

// The kernels take their tables by value: 1.6 KB of kernel arguments, which the compiler reads in a dozen batches at the top
// of the kernel, waiting for each.  A fresh launch finds none of the block's 64-byte lines in the scalar cache, so each batch
// that reaches a new line is a miss of its own, one after the other (measured: 7 k cycles from the wave's first instruction
// to the end of the LDS initialisation).  One scalar load per line, all in flight together, first: the batches then hit.
// (The results are not used.  F2ArgBlock mirrors the parameter list of the step kernels; every touched line lies inside it.)
struct F2ArgBlock { GsTables T; GsF2Tables F; GsRows R; GsSolveCfg C; GsEnvCfg E; double* slab; int B; const double* actions; double total_load;
                    GsPackArgs PA; GsFusedChecks FC; GsRolloutStep RS; };
__device__ __forceinline__ void f2_touch_arguments() {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned sink;
  static_assert(sizeof(F2ArgBlock) >= 0x684 && sizeof(F2ArgBlock) <= 0x6c0, "argument block size changed: adjust the touched lines");
  asm volatile(
      "s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\t"
      "s_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x140\n\ts_load_dword %0, %1, 0x180\n\ts_load_dword %0, %1, 0x1c0\n\t"
      "s_load_dword %0, %1, 0x200\n\ts_load_dword %0, %1, 0x240\n\ts_load_dword %0, %1, 0x280\n\ts_load_dword %0, %1, 0x2c0\n\t"
      "s_load_dword %0, %1, 0x300\n\ts_load_dword %0, %1, 0x340\n\ts_load_dword %0, %1, 0x380\n\ts_load_dword %0, %1, 0x3c0\n\t"
      "s_load_dword %0, %1, 0x400\n\ts_load_dword %0, %1, 0x440\n\ts_load_dword %0, %1, 0x480\n\ts_load_dword %0, %1, 0x4c0\n\t"
      "s_load_dword %0, %1, 0x500\n\ts_load_dword %0, %1, 0x540\n\ts_load_dword %0, %1, 0x580\n\ts_load_dword %0, %1, 0x5c0\n\t"
      "s_load_dword %0, %1, 0x600\n\ts_load_dword %0, %1, 0x640\n\ts_load_dword %0, %1, 0x680\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(sink) : "s"(ka) : "memory");
}

struct F2State { double mm; int iters, conv, status; bool done; };
// mm: the maximum mismatch (what PowerFlowSolution.max_mismatch reports); crit: what is held against the tolerance -- the
// same maximum for Newton-Raphson (power_flow.py:168-171), the mismatch summed over the buses for the sweeps (below)
__device__ __forceinline__ void f2_check(F2State& st, double mm, double crit, int it, double tol) {      // power_flow.py:148, 168-171, 204
  if (!st.done) {
    st.mm = mm; st.iters = it + 1;
    if (!(mm < INFINITY)) { st.status = GS_STATUS_NAN; st.done = true; }
    else if (crit < tol) { st.conv = 1; st.status = GS_STATUS_OK; st.done = true; }
  }
}
// The sweeps stop on sum_i |dP_i| + |dQ_i| < tolerance / 2 (oracle_np.fbs_solve says why: on a radial feeder the sum bounds the
// error of every line flow; the maximum alone left the head-of-feeder flows n_loads x tolerance off).  Across the sub-groups and
// waves of a workgroup the sum is an LDS integer add of fixed-point values (2^-44 pu: exact and order-independent like the
// maxima); inside a lane the additions are in item order.  The conversion is the add-2^52 trick (the integer is the low 52 bits
// of the sum's mantissa: two vector instructions; the compiler's double -> u64 conversion is ten, in a kernel bound by vector
// instruction issue), a lane's share beyond 255 pu saturates there (a NaN is caught through the maximum), and the criterion is
// compared as an integer: sum 2^44 < tolerance 2^43.
#define F2_SUM_SCALE 0x1p44
__device__ __forceinline__ unsigned long long f2_fix(double s) {
  const double d = __builtin_fma(fmin(s, 255.0), F2_SUM_SCALE, 0x1p52);
  return __builtin_bit_cast(unsigned long long, d) & 0x000fffffffffffffull;
}
__device__ __forceinline__ unsigned long long f2_fix_tolerance(double tol) {      // the integer the summed mismatch must stay below
  const double t = fmin(fmax(tol, 0.0), 255.0) * (0.5 * F2_SUM_SCALE);
  return (unsigned long long)t;
}
__device__ __forceinline__ void f2_check_sum(F2State& st, double mm, unsigned long long sum_fix, unsigned long long tol_fix, int it) {
  if (!st.done) {
    st.mm = mm; st.iters = it + 1;
    if (!(mm < INFINITY)) { st.status = GS_STATUS_NAN; st.done = true; }
    else if (sum_fix < tol_fix) { st.conv = 1; st.status = GS_STATUS_OK; st.done = true; }
  }
}

// Bus voltage angle from (e, f) (see bus_angle in kernels_solve.hip): series for small angles, libm otherwise.
__device__ __forceinline__ double f2_angle(double f, double e) {
  if (!__all(e > 0.0 && fabs(f) <= 0.125 * e)) return atan2(f, e);
  const double t = f / e, z = t * t;
  double p = -1.0 / 19.0;
  p = __builtin_fma(p, z, 1.0 / 17.0); p = __builtin_fma(p, z, -1.0 / 15.0); p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, -1.0 / 11.0); p = __builtin_fma(p, z, 1.0 / 9.0); p = __builtin_fma(p, z, -1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0); p = __builtin_fma(p, z, -1.0 / 3.0);
  return __builtin_fma(t * z, p, t);
}

// Taylor coefficients of (sin x - x) / x^3 and (cos x - 1) / x^2 in x^2, highest power first (|x| <= 0.5: truncation < 1e-21)
__device__ const double kF2Series[16] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0,
                                         1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0,
                                         1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0,
                                         -1.0 / 720.0, 1.0 / 24.0, -0.5};
enum { F2_ST_PROLOGUE = 0, F2_ST_INIT, F2_ST_MISMATCH, F2_ST_BOTTOM_UP, F2_ST_FLAG, F2_ST_TOP_DOWN, F2_ST_FINAL_MISMATCH, F2_ST_EPILOGUE,
       F2_ST_PRO_SCALAR, F2_ST_PRO_SPARE, F2_ST_EPI_BUSES, F2_ST_EPI_LINES, F2_ST_EPI_REDUCE, F2_ST_EPI_SCALARS };
struct F2Stamp {
  unsigned long long* p; unsigned long long t; bool mine;
  __device__ __forceinline__ void hit(int k) {
    if (p == nullptr) return;
    const unsigned long long now = __builtin_readcyclecounter();
    if (mine) p[k] += now - t;
    t = now;
  }
};

// SOLVER: 0 forward/backward sweep, 1 Newton-Raphson.  NW wavefronts per workgroup, NI buses per half wave (NW * 2 * NI = 128
// positions): the sweep kernel runs 16 x 4, Newton-Raphson -- whose bus state (voltage, the T and s of the elimination)
// lives in registers across its two sweeps -- 8 x 8, i.e. twice the registers per wave.
enum { F2_FBS = 0, F2_NR = 1, F2_NRM = 2 };      // NRM: Newton-Raphson on a meshed feeder (block LU with fill-in, mesh_schedule.h)
template <int SOLVER, int CHK, int NW, int NI, int IW>
__device__ __forceinline__ void f2_step(const GsTables& T, const GsF2Tables& F, const GsRows& R, const GsSolveCfg& C, const GsEnvCfg& E,
                                        double* __restrict__ slab, int B, const double* __restrict__ actions, double total_load,
                                        const GsPackArgs& PA, const GsFusedChecks& FC, const GsRolloutStep& RS) {
  f2_touch_arguments();
  // IW instances per workgroup (32; 16 or 8 for small feeders, where more of a wavefront's lanes go to different buses):
  // lane = hv * IW + l, sub-group hv of the wave works on its own bus for instance l
  constexpr int HV = 64 / IW;                          // sub-groups (buses) per wavefront
  constexpr unsigned SB = (unsigned)(IW + 1) * 16u;    // bytes of an LDS slot: IW lanes x 16 B + one entry of padding
  const int lane = threadIdx.x & 63, l = lane & (IW - 1), hv = lane / IW;
  // (a 24-bit multiply-add: full rate; the compiler's choice for a 32-bit product it cannot bound is the quarter-rate v_mad_u64_u32)
  auto f2_slot = [](int slot, int ll) -> unsigned { return __umul24((unsigned)slot, SB) + ((unsigned)ll << 4); };
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (a step may be launched as two halves of the grid on two streams, gs_internal.h GsF2Tables::wg_offset: bid is the
  // workgroup's index in the whole grid)
  const int bid = (int)blockIdx.x + F.wg_offset;
  const int g = bid / HV, hs = bid % HV, L = hs * IW + l;      // HV workgroups share a 64-instance slab group
  const int b = g * GS_LANES + L;
  const bool valid = b < B;
  const GsLaneRows S = gs_lane_rows(slab, g, R.total, L);
  // Scalars of the start-up, read from the kernel arguments ONCE: the compiler treats an argument as free to read again
  // wherever it is short of scalar registers, and the start-up was 47 scalar loads, most of them waited for one by one
  // (~150 cycles each, 4 k cycles before the first barrier); a value that went through F2_KEEP is kept (or parked in a
  // vector lane) instead.
  int n = T.n, m = T.m, nsl = F.n_slots;
  int o_env = F.off_env, o_tile = F.off_tile, o_red = F.off_red, o_atom = F.off_atom, o_anc = F.off_anc, o_z = F.off_z, o_prof = F.off_prof;
  int n_tab = SOLVER == F2_FBS ? F.n_jump * nsl * 4 : F.n_anc_ints;  // sweeps: ancestor table; Newton-Raphson: child tables + per-position indices
  const int32_t* anc_g = F.anc; const double* zbus_g = F.zbus;
  F2_KEEP(n); F2_KEEP(m); F2_KEEP(nsl); F2_KEEP(o_env); F2_KEEP(o_tile); F2_KEEP(o_red); F2_KEEP(o_atom); F2_KEEP(o_anc); F2_KEEP(o_z);
  F2_KEEP(o_prof); F2_KEEP(n_tab); F2_KEEP(anc_g); F2_KEEP(zbus_g);
  const int SL_ZERO = n;                 // slots n, n + 1, n + 2: ZERO (0, 0), ONE (1, 0), DUMMY (idle positions)
  double F2_AS3* const env_lds = F2_P(double, o_env);                 // [row][32 lanes]
  double F2_AS3* const loadp_lds = F2_P(double, o_tile);               // [load][32 lanes], dead before the line tile is written
  double F2_AS3* const red_lsum = F2_P(double, o_red);                 // [16 waves][32 lanes]
  double F2_AS3* const red_dev = red_lsum + NW * IW;
  unsigned long long F2_AS3* const cell = F2_P(unsigned long long, o_atom);   // [3][32] convergence maxima, then [3] vmax, [4] vmin bits, [5..7] convergence sums
  unsigned F2_AS3* const icell = (unsigned F2_AS3*)(cell + 8 * IW);                // [16][32] integer counts
  // (Two workgroups share a CU in the 16-instance members and the instruction arbiter serves the oldest wave first: the
  // workgroup that arrived first finishes in 32 us, the other in 42 (tools/block_times.py).  Flipping s_setprio at every phase
  // boundary, the two in opposite states, evens them out -- 38 to 42 us each -- and leaves the launch at 42: the CU's
  // throughput, not the sharing, sets the time.  Not kept.)
  const unsigned long long t_entry = __builtin_readcyclecounter();
  F2Stamp stp{C.stamps, 0ull, bid == 0 && wave == C.stamp_wave && lane == 0};
  if (C.stamps) stp.t = t_entry;
  if (C.stamps && C.block_times && threadIdx.x == 0 && bid < GS_STAMP_BLOCKS) C.stamps[16 + 2 * bid] = __builtin_amdgcn_s_memrealtime();

  // ---- start-up: every global load the prologue's first barrier waits for is asked for FIRST -- the clock and seed rows, the
  // first blockDim entries of the two tables that go to LDS -- and the LDS-only initialisation runs while they fly (the
  // first version loaded a bus-type flag per slot inside the flat-start loop and the tables after it: four to six
  // round trips in a row, 9 k cycles = 13 % of the launch before the first useful instruction)
  double told = ROW(R.TIME), kold = ROW(R.STEP);
  uint64_t seed = lane_seed(S, R);
  const int n_z = SOLVER == F2_NRM ? F.mesh_nz : (SOLVER == F2_FBS ? 2 : 4) * nsl;                        // z per bus; Newton-Raphson (G_ip, B_ip, G_ii, B_ii) per bus
  const int tab0 = (int)threadIdx.x < n_tab ? anc_g[threadIdx.x] : 0;
  const double z0 = (int)threadIdx.x < n_z ? zbus_g[threadIdx.x] : 0.0;
  // (the load profile's 24 factors too: read from the module's table where it is used, every draw wave waited a round trip
  // of its own for two of them -- 3 to 5 k cycles behind the clock)
  const double prof0 = threadIdx.x < 24 ? kDailyProfile[threadIdx.x] : 0.0;
  // flat start in every slot: 1 + 0j, the ZERO slot 0, buses with a voltage set point (the slack; F.fixed_*) their set point
  constexpr int NT = 64 * NW;         // the workgroup's size (NT is a scalar load of its own wherever it is used)
  for (int k = threadIdx.x; k < nsl * IW; k += NT) {
    const int s = k / IW, ll = k & (IW - 1);
    f2_st2(f2_slot(s, ll), make_double2(s == SL_ZERO ? 0.0 : 1.0, 0.0));
  }
  for (int k = threadIdx.x; k < 8 * IW; k += NT) cell[k] = (k >= 4 * IW && k < 5 * IW) ? 0x7ff0000000000000ull : 0ull;   // [4] = vmin starts at +inf
  for (int k = threadIdx.x; k < 16 * IW; k += NT) icell[k] = 0u;
  if ((int)threadIdx.x < n_tab) F2_P(int, o_anc)[threadIdx.x] = tab0;
  for (int k = threadIdx.x + NT; k < n_tab; k += NT) F2_P(int, o_anc)[k] = anc_g[k];
  if ((int)threadIdx.x < n_z) F2_P(double, o_z)[threadIdx.x] = z0;
  if (threadIdx.x < 24) F2_P(double, o_prof)[threadIdx.x] = prof0;
  for (int k = threadIdx.x + NT; k < n_z; k += NT) F2_P(double, o_z)[k] = zbus_g[k];
  f2_lds_sync();                      // (the set points below overwrite slots other threads have just initialised)
  for (int q = 0; q < F.n_fixed; ++q) {
    const int s = q == 0 ? F.fixed_slot0 : F.fixed_slot[q];
    const double e = q == 0 ? F.fixed_val0 : F.fixed_val[q];
    if ((int)threadIdx.x < IW) f2_st2(f2_slot(s, threadIdx.x), make_double2(e, 0.0));
  }

  // ---- inside a rollout: the instances the previous step finished are reset here, where the reference calls env.reset()
  // (algorithms/base.py:289-290) -- terminal observation to the side list, next seed of the instance's chain, fresh
  // observation into the slot this step starts from.  Not rare under random actions (an episode is truncated after 10-20
  // steps), so the rows are moved by the whole workgroup.
  const uint64_t inst = (uint64_t)(E.first_instance + b);
  // (the clock and seed rows were asked for at the top, in the same round trip as the rollout's flags; re-read after a reset)
  if (RS.active && RS.t > 0) {
    int F2_AS3* const fin = F2_P(int, o_red);        // [IW] entry of the side list (>= 0), -1 list full, -2 not finished
    if (wave == 0 && hv == 0) {
      int kx = -2;
      const double te = ROW(R.TERM), tr = ROW(R.TRUNC);
      if (valid && (te != 0.0 || tr != 0.0)) {
        kx = __hip_atomic_fetch_add(RS.term_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kx < RS.term_cap) { RS.term_idx[2 * kx] = RS.t - 1; RS.term_idx[2 * kx + 1] = b; }
        else kx = -1;
      }
      fin[l] = kx;
    }
    f2_lds_sync();
    // the whole workgroup moves the observation rows (obs_dim columns each, consecutive threads on consecutive columns)
    int any = 0;
    for (int k = 0; k < IW; ++k) {
      const int kx = fin[k];
      any |= (kx != -2) ? 1 : 0;
      if (kx >= 0) {
        const double* row = RS.obs_prev + (size_t)(b - l + k) * RS.obs_dim;
        double* dst = RS.term_obs + (size_t)kx * RS.obs_dim;
        for (int c = threadIdx.x; c < RS.obs_dim; c += NT) dst[c] = row[c];
      }
    }
    any = __builtin_amdgcn_readfirstlane(any);
    if (any) {
      if (wave == 0 && hv == 0 && fin[l] != -2) env_reset_lane_scalars(T, R, E, S, inst, next_episode_seed(seed, inst));
      for (int k = 0; k < IW; ++k) {           // the per-bus / per-line rows of the finished instances: every thread takes a share
        if (fin[k] == -2) continue;
        const GsLaneRows Sk = gs_lane_rows(slab, g, R.total, hs * IW + k);
        for (int j = threadIdx.x; j < n + m; j += NT) env_reset_element(T, R, Sk, j);
      }
      f2_sync();                     // the reset rows are in memory (vmcnt covers stores) before the other waves gather them
      for (int k = 0; k < IW; ++k) {
        if (fin[k] == -2) continue;
        const GsLaneRows Sk = gs_lane_rows(slab, g, R.total, hs * IW + k);
        double* row = RS.obs_prev + (size_t)(b - l + k) * RS.obs_dim;
        for (int c = threadIdx.x; c < RS.obs_dim; c += NT) {
          const int sidx = RS.map[c];
          row[c] = (sidx >= 0) ? Sk.lane_row((size_t)sidx * GS_LANES).get() : RS.cst[-sidx - 1];
        }
      }
      told = ROW(R.TIME); kold = ROW(R.STEP); seed = lane_seed(S, R);      // (behind the barrier that published the reset rows)
    }
  }
  // ---- clock; then three independent chains on different waves (grid_env.py:433-477) --------------------------------
  const double tnew = told + E.timestep;
  const uint32_t snew = (uint32_t)(kold + 1.0);
  f2_sync();                         // every wave has read the clock and seed rows before wave 0 moves them on
  stp.hit(15);                       // (kernel start, LDS tables, clock round trip)
  // ---- the records of this lane's items: position p = ((wave * HV + sub-group) * NI + j) of the forest's preorder.  Asked for
  // HERE, before the chains: vector loads and stores return in order, and behind the chains' row stores the injection pass
  // waited ~3 k cycles for its records
  const GsF2Rec* const rec0 = F.recs + ((size_t)(wave * HV + hv) * NI);
  int ibus[NI], ilast[NI];
  unsigned roots = 0u;                   // bit j: item j hangs off the slack bus
  f2_i4 rdev[NI][2]; int rdev_b1[NI];    // devices at the bus (GsF2Rec: nl, l0, l1, ng | g0, g1, nb, b0 | b1)
  if constexpr (SOLVER != F2_NRM) {      // (the meshed member, with more rows than it has registers for their records, reads them in the injection pass)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const f2_i4 a = *(const f2_i4*)&rec0[j].bus;
    ibus[j] = a.x; ilast[j] = a.w; roots |= (a.z & 2) ? (1u << j) : 0u;
    rdev[j][0] = *(const f2_i4*)&rec0[j].nl; rdev[j][1] = *(const f2_i4*)&rec0[j].g0; rdev_b1[j] = rec0[j].b1;
  }
  }
  const int nb = T.n_bats, ng = T.n_gens, nl_ = T.n_loads;
  if (wave == 0) {
    // _apply_actions (grid_env.py:621-651, dynamics.py:189-220): the halves take alternate batteries / generators
    const double* act = actions + (size_t)(valid ? b : 0) * (nb + ng);
    const double dt = E.timestep;
    for (int q0 = 0; q0 < nb; q0 += HV) {
      const int q = q0 + hv; const bool on = q < nb; const int qq = on ? q : 0;
      const double a = act[qq], rating = T.bat_rating[qq], cap = T.bat_cap[qq], eff = T.bat_eff[qq];
      double soc = f2_row(S, R.SOC + qq), bp = f2_row(S, R.BATP + qq);
      const double cmd = a * rating;
      if (valid && cmd > 0.0) {                               // discharge, dynamics.py:206-220
        const double p = fmin(cmd, rating);
        const double e = fmin(gs_div_by(p * dt, 3600.0, 1.0 / 3600.0), soc * cap * eff);      // (x / 3600 correctly rounded, without the division sequence)
        soc -= e / (cap * eff);
        bp = e * 3600.0 / dt;
      } else if (valid && cmd < 0.0) {                        // charge, dynamics.py:189-204
        const double p = fmin(-cmd, rating);
        const double max_e = (1.0 - soc) * cap;
        const double e = fmin(gs_div_by(p * dt, 3600.0, 1.0 / 3600.0), max_e / eff);
        soc += e * eff / cap;
        bp = -(e * 3600.0 / dt);
      }
      if (on) {
        if (valid && cmd != 0.0) { f2_row(S, R.SOC + q) = soc; f2_row(S, R.BATP + q) = bp; }
        env_lds[(F.env_soc + q) * IW + l] = soc; env_lds[(F.env_batp + q) * IW + l] = bp;
      }
    }
    for (int g0 = 0; g0 < ng; g0 += HV) {
      const int gi = g0 + hv; const bool on = gi < ng; const int gg = on ? gi : 0;
      const double c = valid ? (act[nb + gg] + 1.0) / 2.0 : (double)f2_row(S, R.CURT + gg);
      if (on) { if (valid) f2_row(S, R.CURT + gi) = c; env_lds[(F.env_curt + gi) * IW + l] = c; }
    }
    if (valid && hv == 0) { ROW(R.TIME) = told + dt; ROW(R.STEP) = kold + 1.0; }      // grid_env.py:470-471
  }
  if (wave == (NW >= 2 ? 1 : 0)) {
    // _update_weather + renewable models (grid_env.py:653-681, dynamics.py:120-142, 158-170); the halves take alternate generators
    const GsWeather wx = weather_step(R, E, S, inst, tnew, snew, valid);
    const double elev = solar_elevation(valid ? tnew : told);
    for (int g0 = 0; g0 < ng; g0 += HV) {
      const int gi = g0 + hv; const bool on = gi < ng; const int gg = on ? gi : 0;
      const double cap = T.gen_cap[gg], p0 = T.gen_p0[gg], p1 = T.gen_p1[gg], p2 = T.gen_p2[gg];
      double pw;
      if (T.gen_kind[gg] == 0) {
        const double irr = 1000.0 * elev * (1.0 - 0.8 * wx.cloud);
        const double tf = 1.0 - 0.004 * fmax(0.0, wx.temp - 25.0);
        pw = fmin(irr * p1 * p0 * tf, cap);
      } else if (wx.wind < p0 || wx.wind > p2) pw = 0.0;
      else if (wx.wind <= p1) { const double q = (wx.wind - p0) / (p1 - p0); pw = cap * (q * q * q); }
      else pw = cap;
      if (on) { f2_row(S, R.GENP + gi) = pw; env_lds[(F.env_genp + gi) * IW + l] = pw; }
    }
  }
  if (NW < 3 || wave >= 2) {
    // realised load powers (dynamics.py:54-75): loads 4 p .. 4 p + 3 share one Philox call; one quad per sub-group of a wave
    // (workgroups of one or two waves: every wave draws, the first after its two scalar chains)
    const double prof = E.stochastic_loads ? daily_profile_from(F2_P(const double, o_prof), tnew) : 1.0;
    for (int p = (NW < 3 ? wave : wave - 2) * HV + hv; 4 * p < nl_; p += (NW < 3 ? NW : NW - 2) * HV) {
      const int l0 = 4 * p;
      double base4[4];                 // asked for before the draw, whose ~350 instructions cover the round trip
#pragma unroll
      for (int k = 0; k < 4; ++k) base4[k] = T.load_base[min(l0 + k, nl_ - 1)];
      double z[4] = {0.0, 0.0, 0.0, 0.0};
      if (E.stochastic_loads) rng_normal_quad(seed, inst, snew, DRAW_LOAD0 + p, z);
      double lp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double base = base4[k];
        lp[k] = E.stochastic_loads ? fmax(0.0, base * (prof * (1.0 + 0.1 * z[k])) * 1.0) : base;
        if (l0 + k < nl_) loadp_lds[(l0 + k) * IW + l] = lp[k];
      }
      // LOADP starts on an even row: (l0, l0 + 1) and (l0 + 2, l0 + 3) are row pairs
      if (l0 + 1 < nl_) f2_pair(S, R.LOADP + l0) = make_double2(lp[0], lp[1]); else f2_row(S, R.LOADP + l0) = lp[0];
      if (l0 + 3 < nl_) f2_pair(S, R.LOADP + l0 + 2) = make_double2(lp[2], lp[3]); else if (l0 + 2 < nl_) f2_row(S, R.LOADP + l0 + 2) = lp[2];
    }
  }
  stp.hit(F2_ST_PRO_SPARE);
  f2_lds_sync();                                  // what the injection pass reads is in LDS
  stp.hit(F2_ST_PRO_SCALAR);

  // ---- injections of this lane's buses, reference accumulation order (grid_env.py:683-720, power_flow.py:112-121) ----
  double Pj[NI], IR[NI], II[NI], JR[NI], JI[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    if constexpr (SOLVER == F2_NRM) { rdev[j][0] = *(const f2_i4*)&rec0[j].nl; rdev[j][1] = *(const f2_i4*)&rec0[j].g0; rdev_b1[j] = rec0[j].b1; }
    const int nl = rdev[j][0].x, l0 = rdev[j][0].y, l1 = rdev[j][0].z, ngj = rdev[j][0].w, g0 = rdev[j][1].x, g1 = rdev[j][1].y,
              nbj = rdev[j][1].z, b0 = rdev[j][1].w, b1 = rdev_b1[j];
    double ls = 0.0, gs = 0.0;
    if (nl > 0) ls += loadp_lds[l0 * IW + l];
    if (nl > 1) ls += loadp_lds[l1 * IW + l];
    if (ngj > 0) gs += env_lds[(F.env_genp + g0) * IW + l] * env_lds[(F.env_curt + g0) * IW + l];
    if (ngj > 1) gs += env_lds[(F.env_genp + g1) * IW + l] * env_lds[(F.env_curt + g1) * IW + l];
    if (nbj > 0) { const double bp = env_lds[(F.env_batp + b0) * IW + l]; if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp); }
    if (nbj > 1) { const double bp = env_lds[(F.env_batp + b1) * IW + l]; if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp); }
    Pj[j] = (0.0 - gs_div_by(ls, E.power_base, E.inv_power_base)) + gs_div_by(gs, E.power_base, E.inv_power_base);
    if constexpr (SOLVER == F2_NRM) {      // (the meshed member keeps P_spec by bus in LDS: its rows' registers are taken by the elimination)
      f2_st((unsigned)F.mesh_off_p + 64u * (unsigned)rec0[j].bus + ((unsigned)l << 3), Pj[j]);
    }
    IR[j] = 0.0; II[j] = 0.0; JR[j] = 0.0; JI[j] = 0.0;
  }
  stp.hit(F2_ST_PROLOGUE);
  f2_lds_sync();                     // the load powers (tile region) have been read: the region becomes the solver's second buffer
  if constexpr (SOLVER != F2_NRM) {
    if (wave == 0) f2_st2((unsigned)o_tile + f2_slot(SL_ZERO, l), make_double2(0.0, 0.0));      // "no ancestor" reads as 0 in both buffers
  }

  F2State st; st.mm = INFINITY; st.iters = 0; st.conv = 0; st.status = GS_STATUS_MAX_ITER; st.done = !valid;
  double psum = 0.0;
  int check = 0;
  // cross-wave maximum of the mismatch: LDS integer maximum on the bit pattern (mismatch >= 0; inf and NaN patterns are
  // the largest), three cells in rotation so that nobody clears one that is still being read
  auto wg_max = [&](double lmax) -> double {
    const int c0 = check % 3, c1 = (check + 1) % 3;
    ++check;
    // (every sub-group posts its own maximum: the LDS atomic reduces across sub-groups; an exchange first is three dependent
    // LDS shuffles in the 8-instance member)
    if (wave == 0 && hv == 0) cell[c1 * IW + l] = 0ull;
    atomicMax(cell + c0 * IW + l, f2_bits(lmax));
    f2_lds_sync();
    return f2_dbl(cell[c0 * IW + l]);
  };
  // the same with the mismatch summed over the buses beside its maximum (the sweeps' stopping criterion)
  // (in two parts, so that the barrier between them can be one the caller needs anyway)
  auto post_max_sum = [&](double lmax, double lsum) -> int {
    const int c0 = check % 3, c1 = (check + 1) % 3;
    ++check;
    // every sub-group posts its own share (the LDS atomics do the cross-sub-group reduction: the kernel is bound by vector
    // instruction issue, and the two exchanges through v_permlane swaps were 25 vector instructions per check)
    if (wave == 0 && hv == 0) { cell[c1 * IW + l] = 0ull; cell[(5 + c1) * IW + l] = 0ull; }
    atomicMax(cell + c0 * IW + l, f2_bits(lmax)); atomicAdd(cell + (5 + c0) * IW + l, f2_fix(lsum));
    return c0;
  };
  auto read_max_sum = [&](int c0, unsigned long long& sum_out) -> double {
    sum_out = cell[(5 + c0) * IW + l];                                     // the summed mismatch in units of 2^-44 pu
    return f2_dbl(cell[c0 * IW + l]);
  };
  auto wg_max_sum = [&](double lmax, double lsum, unsigned long long& sum_out) -> double {
    const int c0 = post_max_sum(lmax, lsum);
    f2_lds_sync();
    return read_max_sum(c0, sum_out);
  };

  if constexpr (SOLVER == F2_NR) {
  // ================= Newton-Raphson (power_flow.py:143-193), exact Jacobian, 2x2-block elimination along the tree =================
  // Same arithmetic as newton_loop / linsolve_tree_lds of kernels_solve.hip (mismatch :150-171, Jacobian entries :243-287,
  // corrections :297-327); what differs is where the data lives.  There a bus's state went through slab rows every
  // iteration (T_i, s_i, V, 1/V, e, f, P/Q calculated: 9x the algorithmic traffic); here vm, va, P_spec and -- between
  // the bottom-up and the top-down sweep -- T_i and s_i stay in the REGISTERS of the half wave that owns the bus,
  // (e, f) of every bus sit in the LDS slots (neighbours read them), branch currents K = y (V_i - V_parent) and the
  // child -> parent / parent -> child messages share the second LDS region.  The two sweeps are level-synchronous
  // (2x2 inverses do not compose into prefix sums): a wave's items are pairs of buses of one level, walked in level order
  // with an LDS barrier per level.
  const unsigned bufA = 0u, bufB = (unsigned)o_tile;
  const int F2_AS3* const tab = F2_P(const int, o_anc);
  // tab: child buses [n + 1][8], ring slots of the children's messages [n + 1][8] (row n: an idle position), child counts
  // [n_slots] (the kernel does not read them: an entry beyond a bus's children names the ZERO slot / the ring's zero entry,
  // so that a lane subtracts every entry up to its group's largest child count -- no per-lane predicate, no selects)
  const int F2_AS3* const pos_tab = tab + F.pos_off;                // [positions][4]: bus, parent, ring, parent's ring
  int ilev[NI];                                                     // level of the wave's j-th pair (-1: none), wave-uniform
#pragma unroll
  for (int j = 0; j < NI; ++j) ilev[j] = __builtin_amdgcn_readfirstlane(rec0[j].level);
  int imax[NI];                                                     // most children of the pair's two buses, wave-uniform
#pragma unroll
  for (int j = 0; j < NI; ++j) imax[j] = __builtin_amdgcn_readfirstlane(rec0[j].pad1);
  const int pos0 = (wave * HV + hv) * NI;
  const int NL = F.n_levels;
  auto ring3 = [&](int slot, int part) -> unsigned { return bufB + ((unsigned)(slot * 3 + part) * (unsigned)IW + (unsigned)l) * 16u; };
  // the ring's zero entry (what the ring table names beyond a bus's children) lies behind everything the solver writes in
  // the region: written once; the first reader is a barrier or more away
  if (wave == 0 && hv == 0) {
#pragma unroll
    for (int part = 0; part < 3; ++part) f2_st2(ring3(F.ring_zero, part), make_double2(0.0, 0.0));
  }
  // flat start: the slots hold it already (every bus below the slack is a PQ bus, the host checks).  |V| and the angle are
  // not kept: |V| = sqrt(e^2 + f^2) where it is needed, corrections rotate (e, f) by the angle increment
  double pcj[NI], qcj[NI];
  // S = V conj(Y V) through the branch currents: (Y V)_i = K_i - sum over children K_c; returns this lane's max |mismatch|
  const bool flat_use = F.nrflat != nullptr && F.nrflat_mode == 2, flat_cap = F.nrflat != nullptr && F.nrflat_mode == 1;
  double* const flat_tab = F.nrflat + (size_t)pos0 * 16;      // this lane's items: [j][16]
  // iteration 0 with the handle's flat-start table: P / Q calculated and the losses share are constants of the position
  auto mismatch_flat = [&]() -> double {
    double lmax = 0.0, bad = 0.0, ps = 0.0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const double2 c0 = *(const double2*)(flat_tab + 16 * j), c1 = *(const double2*)(flat_tab + 16 * j + 2);
      const double pc = c0.x, qc = c0.y;
      pcj[j] = pc; qcj[j] = qc;
      const double dP = Pj[j] - pc, dQ = 0.0 - qc;
      if (ibus[j] < n) { lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ))); bad = __builtin_fma(dP, 0.0, __builtin_fma(dQ, 0.0, bad)); }
      ps += c1.x;
    }
    if (bad != bad) lmax = INFINITY;
    psum = ps;
    return lmax;
  };
  auto mismatch = [&](bool want_max) -> double {
    double kr[NI], ki[NI], ee[NI], ff[NI], sl[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int bus = ibus[j]; F2_OPAQUE(bus);
      const int par = pos_tab[(pos0 + j) * 4 + 1];
      const double2 v = f2_ld2(bufA + f2_slot(bus, l)), vp = f2_ld2(bufA + f2_slot(par, l)), y = f2_ld2(o_z + 32u * bus);
      const double dr = v.x - vp.x, di = v.y - vp.y;
      // branch admittance = -Y_ip
      kr[j] = __builtin_fma(-y.x, dr, y.y * di); ki[j] = -__builtin_fma(y.x, di, y.y * dr);
      ee[j] = v.x; ff[j] = v.y;
      sl[j] = ((roots >> j) & 1u) ? __builtin_fma(vp.x, kr[j], vp.y * ki[j]) : 0.0;      // Re(V_s conj(K_root)): the slack's share of the losses sum, negated below
      f2_st2(bufB + f2_slot(bus, l), make_double2(kr[j], ki[j]));
    }
    if (wave == 0) f2_st2(bufB + f2_slot(SL_ZERO, l), make_double2(0.0, 0.0));      // "no child" reads as 0 (the ring shares the region and may have written here)
    f2_lds_sync();
    double lmax = 0.0, bad = 0.0, ps = 0.0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int bus = ibus[j]; F2_OPAQUE(bus);
      double icr = kr[j], ici = ki[j];
      {   // the children's currents: the row of child indices first, then every current in one batch (no wait in between)
        const int cb = bus < n ? bus : n;
        const f2_i4 c_lo = *F2_P(const f2_i4, o_anc + 32u * cb), c_hi = *F2_P(const f2_i4, o_anc + 32u * cb + 16u);
        int im = imax[j]; F2_KEEP(im);      // (compared here: hoisted out of the Newton loop the 64 wave masks u < imax[j] live in vector lanes)
#pragma unroll
        for (int u = 0; u < GS_F2_CHILDREN; ++u) {
          if (u < im) {
            const int c = u < 4 ? c_lo[u & 3] : c_hi[u & 3];
            const double2 kc = f2_ld2(bufB + f2_slot(c, l));
            icr -= kc.x; ici -= kc.y;
          }
        }
      }
      const double pc = __builtin_fma(ee[j], icr, ff[j] * ici), qc = __builtin_fma(ff[j], icr, -(ee[j] * ici));
      pcj[j] = pc; qcj[j] = qc;
      const double dP = Pj[j] - pc, dQ = 0.0 - qc;
      if (bus < n) { lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ))); bad = __builtin_fma(dP, 0.0, __builtin_fma(dQ, 0.0, bad)); }
      ps += pc; ps -= sl[j];
      if (flat_cap && want_max && l == 0) {      // (capture: the first mismatch of the capture launch is the flat-start one)
        flat_tab[16 * j] = pc; flat_tab[16 * j + 1] = qc; flat_tab[16 * j + 2] = pc - sl[j]; flat_tab[16 * j + 3] = 0.0;
      }
    }
    if (bad != bad) lmax = INFINITY;
    psum = ps;
    return lmax;
  };
  bool stale = true;
  for (int it = 0; it < C.max_iterations; ++it) {
    const bool flat_it = flat_use && it == 0;
    const double lm = flat_it ? mismatch_flat() : mismatch(it == 0);
    stp.hit(F2_ST_MISMATCH);
    const double mm = wg_max(lm);                     // (its barrier also protects the K slots before the ring reuses the region)
    stp.hit(F2_ST_FLAG);
    f2_check(st, mm, mm, it, C.tolerance);
    stale = false;
    if (__all(st.done) && !(flat_cap && it == 0)) break;      // (the capture launch always runs its first elimination)
    // ---------------- bottom-up: D_i = J_ii - sum C_c, r_i = rhs_i - sum q_c, T_i = D_i^-1 J_ip, s_i = D_i^-1 r_i ----------------
    double T00[NI], T01[NI], T10[NI], T11[NI], s0[NI], s1[NI];
    int sing = 0;
    {
      int lv = 0;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const int lev = ilev[j];
        const bool live = lev >= 0;                   // an item of this wave (both halves share the level)
        T00[j] = 0.0; T01[j] = 0.0; T10[j] = 0.0; T11[j] = 0.0; s0[j] = 0.0; s1[j] = 0.0;
        if (live && flat_it) {
          // iteration 0 with the handle's flat-start table: D^-1, T and L of the position are constants; only the right-hand
          // side travels: s = D^-1 (r - sum of the children's q), q to the parent = L s
          int bus = ibus[j]; F2_OPAQUE(bus);
          int pj = pos0 + j; F2_OPAQUE(pj);
          const f2_i4 px = *F2_P(const f2_i4, o_anc + 4u * F.pos_off + 16u * pj);
          const double* tb = flat_tab + 16 * j;
          const double2 iv0 = *(const double2*)(tb + 4), iv1 = *(const double2*)(tb + 6), tt0 = *(const double2*)(tb + 8), tt1 = *(const double2*)(tb + 10),
                        ll0 = *(const double2*)(tb + 12), ll1 = *(const double2*)(tb + 14);
          double r0 = Pj[j] - pcj[j], r1 = 0.0 - qcj[j];
          const int cbr = bus < n ? bus : n;
          const f2_i4 r_lo = *F2_P(const f2_i4, o_anc + 4u * ((n + 1) * 8) + 32u * cbr), r_hi = *F2_P(const f2_i4, o_anc + 4u * ((n + 1) * 8) + 32u * cbr + 16u);
          while (lv < lev) { f2_lds_sync(); ++lv; }
          int im = imax[j]; F2_KEEP(im);
#pragma unroll
          for (int u0 = 0; u0 < GS_F2_CHILDREN; u0 += 2) {
            if (u0 < im) {
              const int ca = u0 < 4 ? r_lo[u0 & 3] : r_hi[u0 & 3], cb2 = u0 + 1 < 4 ? r_lo[(u0 + 1) & 3] : r_hi[(u0 + 1) & 3];
              const double2 aq = f2_ld2(ring3(ca, 2)), bq = f2_ld2(ring3(cb2, 2));
              r0 -= aq.x; r1 -= aq.y;
              r0 -= bq.x; r1 -= bq.y;
            }
          }
          s0[j] = iv0.x * r0 + iv0.y * r1; s1[j] = iv1.x * r0 + iv1.y * r1;
          if (bus < n && !((roots >> j) & 1u)) {
            T00[j] = tt0.x; T01[j] = tt0.y; T10[j] = tt1.x; T11[j] = tt1.y;
            f2_st2(ring3(px[2], 2), make_double2(ll0.x * s0[j] + ll0.y * s1[j], ll1.x * s0[j] + ll1.y * s1[j]));
          }
        } else
        if (live) {
        // everything that does not depend on the children's messages comes BEFORE the wait for the item's level (it
        // overlaps the levels below): own and parent voltage, 1 / |V| of both, the branch's two off-diagonal blocks, the
        // diagonal block and right-hand side without the children's contributions
        int bus = ibus[j]; F2_OPAQUE(bus);
        int pj = pos0 + j; F2_OPAQUE(pj);
        const f2_i4 px = *F2_P(const f2_i4, o_anc + 4u * F.pos_off + 16u * pj);      // bus, parent, ring, parent's ring
        const double2 yo = f2_ld2(o_z + 32u * bus), yd = f2_ld2(o_z + 32u * bus + 16u);       // (G_ip, B_ip), (G_ii, B_ii)
        const double2 v = f2_ld2(bufA + f2_slot(bus, l)), vp = f2_ld2(bufA + f2_slot(px[1], l));
        const double v2 = __builtin_fma(v.x, v.x, v.y * v.y), rvm = f2_rsq(v2), vmi = v2 * rvm;
        const double rvmp = f2_rsq(__builtin_fma(vp.x, vp.x, vp.y * vp.y));
        const double pc = pcj[j], qc = qcj[j];
        // diagonal block (power_flow.py:247-248 exact sign, 259-260, 270-271, 283-284)
        const double vvb = vmi * vmi * yd.y;
        double d00 = -qc - vvb, d01 = pc * rvm + vmi * yd.x, d10 = pc - vmi * vmi * yd.x, d11 = qc * rvm - vmi * yd.y;
        double r0 = Pj[j] - pc, r1 = 0.0 - qc;                                   // power_flow.py:159-165
        // J(i, p) and J(p, i) of the branch (power_flow.py:251, 263, 274, 287)
        const double a = v.x * vp.x + v.y * vp.y, bbi = v.y * vp.x - v.x * vp.y, bbp = -bbi;
        const double gsi = yo.x * bbi - yo.y * a, gci = yo.x * a + yo.y * bbi;             // row i, column p
        const double gsp = yo.x * bbp - yo.y * a, gcp = yo.x * a + yo.y * bbp;             // row p, column i
        const double u00 = gsi, u01 = gci * rvmp, u10 = -gci, u11 = gsi * rvmp;
        const double l00 = gsp, l01 = gcp * rvm, l10 = -gcp, l11 = gsp * rvm;
        const int cbr = bus < n ? bus : n;
        const f2_i4 r_lo = *F2_P(const f2_i4, o_anc + 4u * ((n + 1) * 8) + 32u * cbr), r_hi = *F2_P(const f2_i4, o_anc + 4u * ((n + 1) * 8) + 32u * cbr + 16u);
        while (lv < lev) { f2_lds_sync(); ++lv; }
        // the children's messages, two children (six 16-byte reads) per round trip
        int im = imax[j]; F2_KEEP(im);
#pragma unroll
        for (int u0 = 0; u0 < GS_F2_CHILDREN; u0 += 2) {
          if (u0 < im) {
            const int ca = u0 < 4 ? r_lo[u0 & 3] : r_hi[u0 & 3], cb2 = u0 + 1 < 4 ? r_lo[(u0 + 1) & 3] : r_hi[(u0 + 1) & 3];
            const double2 a0 = f2_ld2(ring3(ca, 0)), a1 = f2_ld2(ring3(ca, 1)), aq = f2_ld2(ring3(ca, 2));
            const double2 b0 = f2_ld2(ring3(cb2, 0)), b1 = f2_ld2(ring3(cb2, 1)), bq = f2_ld2(ring3(cb2, 2));
            d00 -= a0.x; d01 -= a0.y; d10 -= a1.x; d11 -= a1.y; r0 -= aq.x; r1 -= aq.y;
            d00 -= b0.x; d01 -= b0.y; d10 -= b1.x; d11 -= b1.y; r0 -= bq.x; r1 -= bq.y;
          }
        }
        const double det = d00 * d11 - d01 * d10;
        if (live && bus < n && (!(det != 0.0) || !(fabs(det) < INFINITY))) sing = 1;       // power_flow.py:188-190: only an exactly singular matrix raises
        const double rdet = f2_rcp(det);
        const double i00 = d11 * rdet, i01 = -d01 * rdet, i10 = -d10 * rdet, i11 = d00 * rdet;
        s0[j] = i00 * r0 + i01 * r1; s1[j] = i10 * r0 + i11 * r1;
        if (live && bus < n && !((roots >> j) & 1u)) {
          const double t00 = i00 * u00 + i01 * u10, t01 = i00 * u01 + i01 * u11, t10 = i10 * u00 + i11 * u10, t11 = i10 * u01 + i11 * u11;
          T00[j] = t00; T01[j] = t01; T10[j] = t10; T11[j] = t11;
          f2_st2(ring3(px[2], 0), make_double2(l00 * t00 + l01 * t10, l00 * t01 + l01 * t11));
          f2_st2(ring3(px[2], 1), make_double2(l10 * t00 + l11 * t10, l10 * t01 + l11 * t11));
          f2_st2(ring3(px[2], 2), make_double2(l00 * s0[j] + l01 * s1[j], l10 * s0[j] + l11 * s1[j]));
        }
        if (flat_cap && it == 0 && l == 0) {       // capture launch: this position's constants of the flat-start elimination
          double* tb = flat_tab + 16 * j;
          tb[4] = i00; tb[5] = i01; tb[6] = i10; tb[7] = i11;
          tb[8] = T00[j]; tb[9] = T01[j]; tb[10] = T10[j]; tb[11] = T11[j];
          tb[12] = l00; tb[13] = l01; tb[14] = l10; tb[15] = l11;
        }
              }
      }
      while (lv < NL) { f2_lds_sync(); ++lv; }
    }
    stp.hit(F2_ST_BOTTOM_UP);
    {  // exact singularity anywhere in the instance: stop it where it is (status 2), as the reference's LinAlgError break does.
       // The flag is raised here and read behind the top-down pass, whose level barriers publish it (the pass only forms the
       // step; whether it is applied is decided afterwards): a barrier of its own cost ~800 cycles per iteration.
      if (sing) atomicOr(icell + 15 * IW + l, 1u);             // (any sub-group of the instance; no exchange first)
    }
    stp.hit(F2_ST_INIT);
    // ---------------- top-down: x_i = s_i - T_i x_p; corrections (power_flow.py:315-327); new (e, f) into the slots ----------------
    {
      int lv = NL - 1;
#pragma unroll
      for (int j = NI - 1; j >= 0; --j) {
        __builtin_amdgcn_sched_barrier(0);
        const int lev = ilev[j];
        const bool live = lev >= 0;
        if (live) { while (lv > lev) { f2_lds_sync(); --lv; } }
        int bus = ibus[j]; F2_OPAQUE(bus);
        int pj = pos0 + j; F2_OPAQUE(pj);
        const f2_i4 px = *F2_P(const f2_i4, o_anc + 4u * F.pos_off + 16u * pj);
        double x0 = s0[j], x1 = s1[j];
        if (live && bus < n && !((roots >> j) & 1u)) {
          const double2 xp = f2_ld2(ring3(px[3], 0));
          x0 -= T00[j] * xp.x + T01[j] * xp.y;
          x1 -= T10[j] * xp.x + T11[j] * xp.y;
        }
        if (live && bus < n) f2_st2(ring3(px[2], 0), make_double2(x0, x1));
        s0[j] = x0; s1[j] = x1;                       // the Newton step of the bus; applied below, when T is no longer live
      }
      while (lv >= 0) { f2_lds_sync(); --lv; }
    }
    {
      const unsigned sa = icell[15 * IW + l];
      if (!st.done && sa) { st.status = GS_STATUS_SINGULAR; st.done = true; }
    }
    const bool upd = !st.done;
    stp.hit(F2_ST_TOP_DOWN);
    // corrections (power_flow.py:315-327): theta += alpha dtheta, |V| += alpha d|V|, as a rotation and scaling of (e, f)
    if (__any(upd)) {
      // the sixteen series coefficients once per pass, in scalar registers (a load per use and item before round 3:
      // 128 scalar loads per wave and iteration on the pass's critical path)
      double kcs[16];
      {
        const GS_CONST double* kc0 = (const GS_CONST double*)kF2Series;
#pragma unroll
        for (int q = 0; q < 16; ++q) kcs[q] = kc0[q];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        int bus = ibus[j]; F2_OPAQUE(bus);
        if (upd && bus < n) {
          const double x0 = s0[j], x1 = s1[j];
          const double2 v = f2_ld2(bufA + f2_slot(bus, l));
          const double v2 = __builtin_fma(v.x, v.x, v.y * v.y), rvm0 = f2_rsq(v2), vm0 = v2 * rvm0;
          const double dth = C.alpha * x0, vmn = vm0 + C.alpha * x1;
          // V' = (Vm' / Vm) V e^{j dth}: the rectangular voltage scaled and rotated by the increment -- a negative Vm' gives
          // (|Vm'|, angle + pi), what the reference's abs / angle round trip yields.  sin and cos of the increment from
          // their Taylor series (|h| <= 0.5: truncation < 1e-21; coefficients through the scalar path: as 64-bit literals
          // they would occupy sixteen vector register pairs for the whole Newton loop).  A wave with a step beyond half a
          // radian (diverging iterates) reduces the angle to [-pi, pi], takes an eighth of it and doubles three times.
          const bool big = __any(fabs(dth) > 0.5);
          double h = dth;
          if (big) {
            const double k = rint(dth * 0.15915494309189535);
            h = __builtin_fma(-k, 6.283185307179586, dth);
            h = __builtin_fma(-k, 2.4492935982947064e-16, h);
            h *= 0.125;
          }
          const double* kc = kcs;
          const double z = h * h;
          double sp = kc[0];
          sp = __builtin_fma(sp, z, kc[1]); sp = __builtin_fma(sp, z, kc[2]); sp = __builtin_fma(sp, z, kc[3]);
          sp = __builtin_fma(sp, z, kc[4]); sp = __builtin_fma(sp, z, kc[5]); sp = __builtin_fma(sp, z, kc[6]); sp = __builtin_fma(sp, z, kc[7]);
          double sn = h - h * z * sp;
          double cp = kc[8];
          cp = __builtin_fma(cp, z, kc[9]); cp = __builtin_fma(cp, z, kc[10]); cp = __builtin_fma(cp, z, kc[11]);
          cp = __builtin_fma(cp, z, kc[12]); cp = __builtin_fma(cp, z, kc[13]); cp = __builtin_fma(cp, z, kc[14]); cp = __builtin_fma(cp, z, kc[15]);
          double cs = __builtin_fma(z, cp, 1.0);
          if (big) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { const double c2 = __builtin_fma(cs, cs, -(sn * sn)), s2 = 2.0 * cs * sn; cs = c2; sn = s2; }
          }
          const double ratio = vmn * rvm0;
          const double en = ratio * (v.x * cs - v.y * sn), fn = ratio * (v.x * sn + v.y * cs);
          f2_st2(bufA + f2_slot(bus, l), make_double2(en, fn));
        }
      }
    }
    f2_lds_sync();                   // the new voltages are read by the neighbours' lanes in the next mismatch
    stp.hit(14);
    stale = true;
  }
  if (stale) { (void)mismatch(false); }       // iteration cap reached after an update: the losses sum at the final voltages
  } else if constexpr (SOLVER == F2_NRM) {
  // ================= Newton-Raphson on a MESHED feeder (power_flow.py:143-193; the linear solve of :186-190 as a block LU) =================
  // mesh_schedule.h describes the elimination: pull model, accumulating messages, a pivot of degree d on a group of d
  // sub-groups of one wavefront row.  Here: 8 instances per workgroup, NW wavefronts, each walking its rows (at most NI) in
  // level order; a row is 8 sub-groups = 8 lane items of 16 words (GS_MESH_W_*), read from global memory ONE ROW AHEAD (the
  // table is the same for every workgroup: it stays in the L1 / L2; read where it is used, a row cost two dependent round
  // trips of ~1 k cycles -- 400 k cycles per step).  LDS: the voltage slots (as everywhere in this file); the Ybus entries
  // of the connected pairs and of the diagonal, and every bus's neighbour list (staged by the frame at off_z / off_anc);
  // the region behind the slots = the ZERO message, a DUMMY to write to, and the body -- accumulators during the
  // elimination, the x slots during the back substitution --; a scratch of 3 units per sub-group and wave for the exchanges
  // inside a group (D^-1 and s from lane 0, every lane's T, the partial sums of the back substitution: LDS executes a wave's
  // instructions in order, so a write followed by a read of the same wave needs no barrier).  Registers: T of every row's
  // item and s / x of its pivot (NI x 6 doubles), P_spec.
  static_assert(IW == 8, "the meshed member is laid out for 8 instances per workgroup");
  constexpr unsigned UB = 16u * IW;                              // a unit: 16 bytes per instance
  const unsigned l16 = (unsigned)l << 4;
  const unsigned R0l = (unsigned)o_tile + l16;                   // unit u of the region, this lane's share: R0l + u * UB
  // this wave's exchange scratch, 16 units, used two ways one after the other: every lane's T at 2 units per sub-group, the back
  // substitution's partial sums at 1 unit per sub-group
  const unsigned scr = (unsigned)F.off_scr + (unsigned)wave * (16u * UB) + l16;
  const unsigned o_ptab = (unsigned)F.mesh_off_p + ((unsigned)l << 3);
  const unsigned o_pair = (unsigned)o_z, o_diag = (unsigned)o_z + 16u * (unsigned)(F.mesh_pairs + 1);
  const GS_CONST int32_t* const rinfo = (const GS_CONST int32_t*)F.mesh_rowinfo + (size_t)wave * NI * 4;      // wave-uniform: scalar loads
  const f2_i4* const items = (const f2_i4*)F.mesh_items + ((size_t)wave * NI * HV + hv) * 4;                   // row j: items[j * HV * 4 + 0..3]
  const int NL = F.n_levels;
  const unsigned null_ent = (unsigned)F.n_anc_ints - 1u;
  if (threadIdx.x < 3 * IW) f2_st2((unsigned)o_tile + (threadIdx.x >> 3) * UB + ((threadIdx.x & 7u) << 4), make_double2(0.0, 0.0));   // the ZERO message (first read: behind the first check's barrier)
  // row j of the register arrays, j wave-uniform: a scalar compare-and-branch chain around the moves (the asm keeps the compiler
  // from turning it into 2 (NI - 1) selects per value)
#define F2_ROW_CASE(Q, STMT) case Q: if constexpr (Q < NI) { constexpr int RQ = Q < NI ? Q : 0; STMT; asm volatile(""); } break;
#define F2_ROW(j, STMT) switch (j) { F2_ROW_CASE(0, STMT) F2_ROW_CASE(1, STMT) F2_ROW_CASE(2, STMT) F2_ROW_CASE(3, STMT) F2_ROW_CASE(4, STMT) F2_ROW_CASE(5, STMT) \
                                     F2_ROW_CASE(6, STMT) F2_ROW_CASE(7, STMT) F2_ROW_CASE(8, STMT) F2_ROW_CASE(9, STMT) F2_ROW_CASE(10, STMT) F2_ROW_CASE(11, STMT) default: break; }
  static_assert(NI <= 12, "F2_ROW covers 12 rows");
  // an item's words 0-3 (what a row needs before its level's messages) are read a row ahead, words 4-15 at the row's start
  auto load_item = [&](int j) -> f2_i4 { return items[(size_t)j * (HV * 4)]; };
  auto lo16 = [](int w) -> unsigned { return (unsigned)w & 0xffffu; };
  auto hi16 = [](int w) -> unsigned { return (unsigned)w >> 16; };
  auto unit_at = [&](unsigned u) -> unsigned { return R0l + u * UB; };
  auto vslot = [&](unsigned slot) -> unsigned { return __umul24(slot, SB) + l16; };
  // (Y V)_k over the neighbours of the item's bus (the LDS neighbour list: pair | other bus << 16), then the diagonal; S = V conj(I)
  auto calc_pq = [&](const f2_i4& it, int nadj, double2 vk, double& pc, double& qc) {
    const unsigned ap = hi16(it.w), cnt = (unsigned)it.y >> GS_MESH_F_NADJ_SHIFT;
    const double2 yd = f2_ld2(o_diag + 16u * lo16(it.w));
    // (no predicate on the lane's own count: beyond it a lane reads the list's last entry -- no branch, the ZERO slot -- so that the
    // reads of four entries go out together; with `if (u < cnt)` every entry was two LDS round trips of its own)
    double ir = 0.0, ii = 0.0;
    for (int u0 = 0; u0 < nadj; u0 += 4) {
      int e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = *F2_P(const int, (unsigned)o_anc + 4u * ((unsigned)(u0 + q) < cnt ? ap + (unsigned)(u0 + q) : null_ent));
      double2 y[4], v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { y[q] = f2_ld2(o_pair + 16u * lo16(e[q])); v[q] = f2_ld2(vslot(hi16(e[q]))); }
#pragma unroll
      for (int q = 0; q < 4; ++q) { ir += __builtin_fma(y[q].x, v[q].x, -(y[q].y * v[q].y)); ii += __builtin_fma(y[q].x, v[q].y, y[q].y * v[q].x); }
    }
    if (cnt) { ir += __builtin_fma(yd.x, vk.x, -(yd.y * vk.y)); ii += __builtin_fma(yd.x, vk.y, yd.y * vk.x); }
    pc = __builtin_fma(vk.x, ir, vk.y * ii); qc = __builtin_fma(vk.y, ir, -(vk.x * ii));        // power_flow.py:150-157
  };
  // Iteration 0 of every solve starts from the flat start, where the Jacobian -- and with it every pivot's D^-1, T(k, j) and the
  // column block A(j, k) its messages are formed with, P and Q calculated -- does not depend on the instance.  The handle keeps them
  // in a table, 16 doubles per item (D^-1, T, A(j, k), P calc, Q calc), written ONCE by one workgroup of this kernel (nrflat_mode 1,
  // gs_create); iteration 0 of every later step (mode 2) only carries its right-hand side through: the q parts of the CQ messages
  // up, x down.  Mode 0: every iteration eliminates for itself (GS_NR_NO_FLAT=1).
  const bool flat_use = F.nrflat != nullptr && F.nrflat_mode == 2, flat_cap = F.nrflat != nullptr && F.nrflat_mode == 1;
  double* const flat_tab = F.nrflat + ((size_t)wave * NI * HV + hv) * 16;      // row j: flat_tab + j * HV * 16
  auto mismatch = [&](bool flat_it, bool capture) -> double {
    double lmax = 0.0, bad = 0.0, ps = 0.0;
    f2_i4 nxt = load_item(0);
#pragma unroll 2
    for (int j = 0; j < NI; ++j) {
      const int lev = rinfo[4 * j];
      if (lev < 0) break;                                        // a wave's rows are the first of its NI
      const int nadj = rinfo[4 * j + 2];
      const f2_i4 it = nxt;
      if (j + 1 < NI) nxt = load_item(j + 1);
      const int fl = it.y;
      const double2 vk = f2_ld2(vslot(lo16(it.x)));
      double pc, qc;
      if (flat_it) { const double2 c = *(const double2*)(flat_tab + (size_t)j * (HV * 16) + 12); pc = c.x; qc = c.y; }
      else calc_pq(it, nadj, vk, pc, qc);
      if (capture && l == 0) { double* tb = flat_tab + (size_t)j * (HV * 16); tb[12] = pc; tb[13] = qc; tb[14] = 0.0; tb[15] = 0.0; }
      const double Pk = f2_ld(o_ptab + 64u * lo16(it.x));
      const double dP = Pk - pc, dQ = 0.0 - qc;                  // power_flow.py:159-165
      if (fl & GS_MESH_F_PIVOT) { lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ))); bad = __builtin_fma(dP, 0.0, __builtin_fma(dQ, 0.0, bad)); }
      if (fl & (GS_MESH_F_PIVOT | GS_MESH_F_SLACKPOS)) ps += pc;                                    // total losses = sum of P_calc over ALL buses (:198-200)
    }
    if (bad != bad) lmax = INFINITY;
    psum = ps;
    return lmax;
  };
  bool stale = true;
  int it_first = 0;
  bool all_done = false;
  // Iteration 0 with the matrix product (below) stands in FRONT of the loop and of the arrays the elimination keeps T / s in: inside
  // the loop their 120 registers count as live across the product (the compiler cannot see through the row switch that the next
  // elimination writes every entry before the back substitution reads it), and the product had nothing to prefetch into.
  if (flat_use && F.mesh_w != nullptr && C.max_iterations > 0) {
    // (the product's first two batches of A operands are requested here, in front of the flat-start check they do not depend on)
    constexpr int WS = 32, WCH = 16;                                  // k-steps of 4 (host: zero-padded to 32); MFMAs a batch
    const double* const wp = F.mesh_w + (size_t)wave * WS * 64 + (threadIdx.x & 63);      // tile wave + 4 i: + i * 4 WS * 64
    double ab[3][WCH];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < WCH; ++e) ab[c][e] = wp[(size_t)(c / 2) * (4 * WS * 64) + (size_t)(WCH * (c % 2) + e) * 64];
    const double lm = mismatch(true, false);
    stp.hit(F2_ST_MISMATCH);
    const double mm = wg_max(lm);
    stp.hit(F2_ST_FLAG);
    f2_check(st, mm, mm, 0, C.tolerance);
    stale = false;
    it_first = 1;
    if (__all(st.done)) all_done = true;
    else {
      // ---------------- iteration 0 as a matrix product: x = W [P_spec; 1] on the matrix cores (GsF2Tables::mesh_w) ----------------
      // At the flat start the first Newton step is a constant linear map of the instance's injections; carried through the levels of
      // the elimination (q parts up, x down, the handle's table of D^-1 / T) it was 61 k of a step's 213 k cycles.  As a product:
      // 16 row tiles x mesh_w_steps k-steps of v_mfma_f64_16x16x4, four tiles a wavefront; operand A = W (global memory, in operand
      // order: one coalesced 512-byte read per MFMA, shared by every workgroup -> L2), operand B = P_spec of four buses x the
      // workgroup's 8 instances from the LDS table (columns 8-15 of the tile: zero), result rows = (d theta, d|V|) of a bus.
      {
        typedef double f2_v4 __attribute__((ext_vector_type(4)));
        const int ln = threadIdx.x & 63, q4 = ln >> 4, r16 = ln & 15;
        constexpr int S = WS;                                           // k-steps of 4: n - 1 buses + the constant column <= 128
        const int na_ = n - 1, sl = F.mesh_slack;
        f2_v4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f2_v4{0.0, 0.0, 0.0, 0.0};
        constexpr size_t tstride = (size_t)4 * S * 64;
        // The B operands of all 32 k-steps first (LDS), then sixteen MFMAs a batch (half a tile), their A operands requested two
        // batches ahead: W is shared by every workgroup but comes from the far side of the L2 often enough (~2 k cycles) that a k-step at
        // a time, or four with two batches ahead, left the product waiting (56 k cycles for 128 MFMAs of 64 cycles each).
        double bv[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
          const int a = 4 * s2 + q4;
          const unsigned bus = (unsigned)(a < sl ? a : a + 1);
          double b = 0.0;
          if (r16 < IW) { if (a < na_) b = f2_ld((unsigned)F.mesh_off_p + 64u * bus + 8u * (unsigned)r16); else if (a == na_) b = 1.0; }
          bv[s2] = b;
        }
        constexpr int CH = WCH, NCH = 4 * S / CH;                        // 8 batches: batch c = tile c / 2, k-steps 16 (c % 2) ..
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (c + 2 < NCH) {
#pragma unroll
            for (int e = 0; e < CH; ++e) ab[(c + 2) % 3][e] = wp[(size_t)((c + 2) / 2) * tstride + (size_t)(CH * ((c + 2) % 2) + e) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < CH; ++e) acc[c / 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[c % 3][e], bv[CH * (c % 2) + e], acc[c / 2], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // C/D[row = (lane >> 4) + 4 reg][col = lane & 15]: unknown u = 16 (wave + 4 i) + 4 g + q4 = 2 a + {0: angle, 1: magnitude}
        if (r16 < IW) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int u = 16 * (wave + 4 * i) + 4 * g + q4, a = u >> 1;
              if (a < na_) {
                const unsigned bus = (unsigned)(a < sl ? a : a + 1);
                *F2_P(double, (unsigned)o_tile + (6u + bus) * UB + ((unsigned)r16 << 4) + ((unsigned)(u & 1) << 3)) = acc[i][g];
              }
            }
        }
      }
      stp.hit(14);
      f2_lds_sync();
      // corrections (power_flow.py:315-327) as a rotation and scaling of (e, f): the update of the back substitution below, row by row
      {
        const bool upd = !st.done;
        const GS_CONST double* kc = (const GS_CONST double*)kF2Series;
        // (two rows in flight: the rows are independent here, and a row alone is a chain of two LDS reads, a reciprocal square root and two series)
#pragma unroll 2
        for (int j = 0; j < NI; ++j) {
          if (rinfo[4 * j] < 0) break;
          const f2_i4 ia = items[(size_t)j * (HV * 4)];
          if (upd && (ia.y & GS_MESH_F_PIVOT)) {
            const double2 xk = f2_ld2(unit_at(6u + lo16(ia.x)));
            const unsigned vo = vslot(lo16(ia.x));
            const double2 v = f2_ld2(vo);
            const double v2 = __builtin_fma(v.x, v.x, v.y * v.y), rvm0 = f2_rsq(v2), vm0 = v2 * rvm0;
            const double dth = C.alpha * xk.x, vmn = vm0 + C.alpha * xk.y;
            const bool big = __any(fabs(dth) > 0.5);
            double h = dth;
            if (big) {
              const double k = rint(dth * 0.15915494309189535);
              h = __builtin_fma(-k, 6.283185307179586, dth);
              h = __builtin_fma(-k, 2.4492935982947064e-16, h);
              h *= 0.125;
            }
            const double z = h * h;
            double sp = kc[0];
            sp = __builtin_fma(sp, z, kc[1]); sp = __builtin_fma(sp, z, kc[2]); sp = __builtin_fma(sp, z, kc[3]);
            sp = __builtin_fma(sp, z, kc[4]); sp = __builtin_fma(sp, z, kc[5]); sp = __builtin_fma(sp, z, kc[6]); sp = __builtin_fma(sp, z, kc[7]);
            double sn = h - h * z * sp;
            double cp = kc[8];
            cp = __builtin_fma(cp, z, kc[9]); cp = __builtin_fma(cp, z, kc[10]); cp = __builtin_fma(cp, z, kc[11]);
            cp = __builtin_fma(cp, z, kc[12]); cp = __builtin_fma(cp, z, kc[13]); cp = __builtin_fma(cp, z, kc[14]); cp = __builtin_fma(cp, z, kc[15]);
            double cs = __builtin_fma(z, cp, 1.0);
            if (big) {
#pragma unroll
              for (int q = 0; q < 3; ++q) { const double c2 = __builtin_fma(cs, cs, -(sn * sn)), s2 = 2.0 * cs * sn; cs = c2; sn = s2; }
            }
            const double ratio = vmn * rvm0;
            f2_st2(vo, make_double2(ratio * (v.x * cs - v.y * sn), ratio * (v.x * sn + v.y * cs)));
          }
        }
      }
      stp.hit(F2_ST_TOP_DOWN);
      f2_lds_sync();                 // the new voltages are read by the neighbours' lanes in the next mismatch
      stale = true;
    }
  }
  if (!all_done)
  for (int it_ = it_first; it_ < C.max_iterations; ++it_) {
    const bool flat_it = flat_use && it_ == 0;
    const double lm = mismatch(flat_it, flat_cap && it_ == 0);      // (capture: the first mismatch of the capture launch is the flat-start one)
    stp.hit(F2_ST_MISMATCH);
    const double mm = wg_max(lm);
    stp.hit(F2_ST_FLAG);
    f2_check(st, mm, mm, it_, C.tolerance);
    stale = false;
    if (__all(st.done) && !(flat_cap && it_ == 0)) break;        // (the capture launch always runs its first elimination)
    // T / s of the wave's rows: written by this iteration's elimination, read by its back substitution -- declared HERE, so that their
    // 120 registers are not live across the mismatch pass above (declared in front of the loop they were: the compiler cannot see
    // through the row switch that every entry is written before it is read)
    double T00[NI], T01[NI], T10[NI], T11[NI], sx0[NI], sx1[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { T00[j] = 0.0; T01[j] = 0.0; T10[j] = 0.0; T11[j] = 0.0; sx0[j] = 0.0; sx1[j] = 0.0; }
    // ---------------- elimination, level by level ----------------
    {
      int lv = 0;
      f2_i4 nxt = load_item(0);
#pragma nounroll
      for (int j = 0; j < NI; ++j) {
        const int lev = rinfo[4 * j];
        if (lev < 0) break;
        const int pk = rinfo[4 * j + 1], nadj = rinfo[4 * j + 2];
        const int lev_next = j + 1 < NI ? rinfo[4 * (j + 1)] : -1;
        const int g_row = pk & 255, ncq = (pk >> 8) & 255, nrw = (pk >> 16) & 255, ncl = (pk >> 24) & 255;
        const f2_i4 it = nxt;
        const f2_i4* const late = items + (size_t)j * (HV * 4) + 1;
        const f2_i4 pl0 = late[0], pl1 = late[1], pl2 = late[2];      // words 4-15: cq_in (2), rw_in (2) | cl_in (2), mout 0-3 (2) | mout 4-7 (2), -
        if (j + 1 < NI) nxt = load_item(j + 1);
        // (The barrier that publishes a row comes right behind the row's last message, below: a wave that went from its row into
        // the register-file store and the next row's preparations first kept the wave of the next level waiting for exactly that
        // long.)  Everything that does not depend on the messages comes before the wait for the row's level: it runs while
        // another wave is on the levels in between
        const int fl = it.y;
        double t00, t01, t10, t11, s0, s1;                        // what the row leaves in its registers: T(k, j_t) and the pivot's s
        if (flat_it) {
          // iteration 0 with the handle's flat-start table: s = D^-1 (r + the q parts addressed to this pivot); q to row j_t = -A(j_t, k) s
          const double* tb = flat_tab + (size_t)j * (HV * 16);
          const double2 iv0 = *(const double2*)tb, iv1 = *(const double2*)(tb + 2), tt0 = *(const double2*)(tb + 4), tt1 = *(const double2*)(tb + 6),
                        cc0 = *(const double2*)(tb + 8), cc1 = *(const double2*)(tb + 10), pq = *(const double2*)(tb + 12);
          const double Pk = f2_ld(o_ptab + 64u * lo16(it.x));
          double r0 = Pk - pq.x, r1 = 0.0 - pq.y;
          const unsigned tl = ((unsigned)fl >> GS_MESH_F_T_SHIFT) & 15u;
          while (lv < lev) { f2_lds_sync(); ++lv; }
          {
            const double2 m0 = f2_ld2(unit_at(lo16(pl0.x)) + 2u * UB), m1 = f2_ld2(unit_at(hi16(pl0.x)) + 2u * UB);
            r0 += m0.x; r1 += m0.y; r0 += m1.x; r1 += m1.y;
          }
          if (ncq > 2) {
            const double2 m0 = f2_ld2(unit_at(lo16(pl0.y)) + 2u * UB), m1 = f2_ld2(unit_at(hi16(pl0.y)) + 2u * UB);
            r0 += m0.x; r1 += m0.y; r0 += m1.x; r1 += m1.y;
          }
          s0 = __builtin_fma(iv0.x, r0, iv0.y * r1); s1 = __builtin_fma(iv1.x, r0, iv1.y * r1);      // (every lane of the group: the same s)
          t00 = tt0.x; t01 = tt0.y; t10 = tt1.x; t11 = tt1.y;
          const bool rmw = ((fl >> (GS_MESH_F_RMW_SHIFT + tl)) & 1) != 0;
          const unsigned o = unit_at(hi16(it.z)) + 2u * UB;
          const double2 old = f2_ld2(rmw ? o : R0l);
          f2_st2(o, make_double2(__builtin_fma(-cc0.x, s0, __builtin_fma(-cc0.y, s1, old.x)), __builtin_fma(-cc1.x, s0, __builtin_fma(-cc1.y, s1, old.y))));
        } else {
        const bool pivot = (fl & GS_MESH_F_PIVOT) != 0;
        const unsigned hv0 = ((unsigned)fl >> GS_MESH_F_HV0_SHIFT) & 15u, tl = ((unsigned)fl >> GS_MESH_F_T_SHIFT) & 15u;
        const double2 vk = f2_ld2(vslot(lo16(it.x))), vj = f2_ld2(vslot(hi16(it.x)));
        const double2 ykj = f2_ld2(o_pair + 16u * lo16(it.z)), ykk = f2_ld2(o_diag + 16u * lo16(it.w));
        double pc, qc;
        calc_pq(it, nadj, vk, pc, qc);                            // (every lane of a group: the pivot bus's)
        const double Pk = f2_ld(o_ptab + 64u * lo16(it.x));
        const double v2 = __builtin_fma(vk.x, vk.x, vk.y * vk.y), rvk = f2_rsq(v2), vmk = v2 * rvk;
        const double rvj = f2_rsq(__builtin_fma(vj.x, vj.x, vj.y * vj.y));
        // diagonal block (power_flow.py:247-248 exact sign, 259-260, 270-271, 283-284) and right-hand side (:159-165)
        double d00 = -qc - v2 * ykk.y, d01 = pc * rvk + vmk * ykk.x, d10 = pc - v2 * ykk.x, d11 = qc * rvk - vmk * ykk.y;
        double r0 = Pk - pc, r1 = 0.0 - qc;
        // the branch's two off-diagonal blocks (power_flow.py:251, 263, 274, 287): row k column j, row j column k
        const double a = vk.x * vj.x + vk.y * vj.y, bk = vk.y * vj.x - vk.x * vj.y, bj = -bk;
        const double gsk = ykj.x * bk - ykj.y * a, gck = ykj.x * a + ykj.y * bk;
        const double gsj = ykj.x * bj - ykj.y * a, gcj = ykj.x * a + ykj.y * bj;
        double a00 = gsk, a01 = gck * rvj, a10 = -gck, a11 = gsk * rvj;             // A(k, j)
        double c00 = gsj, c01 = gcj * rvk, c10 = -gcj, c11 = gsj * rvk;             // A(j, k)
        while (lv < lev) { f2_lds_sync(); ++lv; }
        // ---- pull: what earlier pivots addressed to this one (lists of units, two per word; entries beyond a list name the ZERO
        // message).  The first two entries of every list unconditionally, their reads together (a wave-uniform `if` per entry was a
        // round trip per entry, on the row's critical path); the other two only in rows that have such lists
        auto pull_cq = [&](int w) {
          const unsigned q0 = unit_at(lo16(w)), q1 = unit_at(hi16(w));
          const double2 m00 = f2_ld2(q0), m01 = f2_ld2(q0 + UB), m0q = f2_ld2(q0 + 2u * UB), m10 = f2_ld2(q1), m11 = f2_ld2(q1 + UB), m1q = f2_ld2(q1 + 2u * UB);
          d00 += m00.x; d01 += m00.y; d10 += m01.x; d11 += m01.y; r0 += m0q.x; r1 += m0q.y;
          d00 += m10.x; d01 += m10.y; d10 += m11.x; d11 += m11.y; r0 += m1q.x; r1 += m1q.y;
        };
        auto pull_rc = [&](int wr, int wc) {
          const unsigned w0 = unit_at(lo16(wr)), w1 = unit_at(hi16(wr)), e0 = unit_at(lo16(wc)), e1 = unit_at(hi16(wc));
          const double2 r00 = f2_ld2(w0), r01 = f2_ld2(w0 + UB), r10 = f2_ld2(w1), r11 = f2_ld2(w1 + UB);
          const double2 l00 = f2_ld2(e0), l01 = f2_ld2(e0 + UB), l10 = f2_ld2(e1), l11 = f2_ld2(e1 + UB);
          a00 += r00.x; a01 += r00.y; a10 += r01.x; a11 += r01.y; a00 += r10.x; a01 += r10.y; a10 += r11.x; a11 += r11.y;
          c00 += l00.x; c01 += l00.y; c10 += l01.x; c11 += l01.y; c00 += l10.x; c01 += l10.y; c10 += l11.x; c11 += l11.y;
        };
        // (the six reads of (D, r) first: the inverse starts on them while the eight of the off-diagonal blocks are still under way;
        // all fourteen at once were 40 more spilled registers and no faster)
        pull_cq(pl0.x);
        if (ncq > 2) pull_cq(pl0.y);
        pull_rc(pl0.z, pl1.x);
        if ((nrw | ncl) > 2) pull_rc(pl0.w, pl1.y);
        // ---- the pivot: D^-1, s (every lane of the group for itself)
        const double det = __builtin_fma(d00, d11, -(d01 * d10));
        // power_flow.py:188-190: only an exactly singular matrix raises.  Posted at once: the barriers of the levels that follow
        // publish it, and the back substitution knows before its first row whether the step will be applied
        if (pivot && (!(det != 0.0) || !(fabs(det) < INFINITY))) atomicOr(icell + 15 * IW + l, 1u);
        const double rdet = f2_rcp(det);
        double i00 = d11 * rdet, i01 = -d01 * rdet, i10 = -d10 * rdet, i11 = d00 * rdet;
        s0 = __builtin_fma(i00, r0, i01 * r1); s1 = __builtin_fma(i10, r0, i11 * r1);
        t00 = __builtin_fma(i00, a00, i01 * a10); t01 = __builtin_fma(i00, a01, i01 * a11); t10 = __builtin_fma(i10, a00, i11 * a10);
        t11 = __builtin_fma(i10, a01, i11 * a11);                                                                  // T(k, j) = D^-1 A(k, j)
        if (flat_cap && it_ == 0 && l == 0) {      // capture launch: this item's constants of the flat-start elimination
          double* tb = flat_tab + (size_t)j * (HV * 16);
          tb[0] = i00; tb[1] = i01; tb[2] = i10; tb[3] = i11; tb[4] = t00; tb[5] = t01; tb[6] = t10; tb[7] = t11;
          tb[8] = c00; tb[9] = c01; tb[10] = c10; tb[11] = c11;
        }
        if (g_row > 1) { unsigned hvx = (unsigned)hv; F2_OPAQUE(hvx); const unsigned o = scr + hvx * (2u * UB); f2_st2(o, make_double2(t00, t01)); f2_st2(o + UB, make_double2(t10, t11)); asm volatile("" ::: "memory"); }
        // ---- what this pivot sends on: row j_t of its messages.  q first (it needs nothing from the group)
        {
          const bool rmw = ((fl >> (GS_MESH_F_RMW_SHIFT + tl)) & 1) != 0;
          const unsigned o = unit_at(hi16(it.z)) + 2u * UB;
          const double2 old = f2_ld2(rmw ? o : R0l);
          f2_st2(o, make_double2(__builtin_fma(-c00, s0, __builtin_fma(-c01, s1, old.x)), __builtin_fma(-c10, s0, __builtin_fma(-c11, s1, old.y))));
        }
        // the M blocks, G outputs at a time: every read of the batch (the group's T from the scratch, what the accumulators hold) goes
        // out before the first product (one output per wave-uniform `if` was a round trip per output)
        auto send = [&](auto gc, int t_first) {
          constexpr int G = decltype(gc)::value;
          double2 ta[G], tb[G], oa[G], ob[G]; unsigned oo[G];
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int t2 = t_first + q;
            const int w = t2 < 2 ? pl1.z : t2 < 4 ? pl1.w : t2 < 6 ? pl2.x : pl2.y;      // words 10 .. 13
            oo[q] = unit_at((t2 & 1) ? hi16(w) : lo16(w));
            const unsigned ro = ((fl >> (GS_MESH_F_RMW_SHIFT + t2)) & 1) ? oo[q] : R0l;
            oa[q] = f2_ld2(ro); ob[q] = f2_ld2(ro + UB);
            if (g_row > 1) { const unsigned o = scr + ((hv0 + (unsigned)t2) & 7u) * (2u * UB); ta[q] = f2_ld2(o); tb[q] = f2_ld2(o + UB); }
            else { ta[q] = make_double2(t00, t01); tb[q] = make_double2(t10, t11); }
          }
#pragma unroll
          for (int q = 0; q < G; ++q) {
            f2_st2(oo[q], make_double2(__builtin_fma(-c00, ta[q].x, __builtin_fma(-c01, tb[q].x, oa[q].x)), __builtin_fma(-c00, ta[q].y, __builtin_fma(-c01, tb[q].y, oa[q].y))));
            f2_st2(oo[q] + UB, make_double2(__builtin_fma(-c10, ta[q].x, __builtin_fma(-c11, tb[q].x, ob[q].x)), __builtin_fma(-c10, ta[q].y, __builtin_fma(-c11, tb[q].y, ob[q].y))));
          }
        };
        // (two at a time: four of them are 64 registers of reads in flight, in a kernel that has none to spare)
        if (g_row == 1) send(std::integral_constant<int, 1>{}, 0);
        else { for (int t0 = 0; t0 < g_row; t0 += 2) send(std::integral_constant<int, 2>{}, t0); }
        }
        if (lev_next != lev) { f2_lds_sync(); ++lv; }             // the row's messages are out; the wave's next row (if any) is of a later level
        F2_ROW(j, (T00[RQ] = t00, T01[RQ] = t01, T10[RQ] = t10, T11[RQ] = t11, sx0[RQ] = s0, sx1[RQ] = s1));
      }
      while (lv < NL) { f2_lds_sync(); ++lv; }
    }
    stp.hit(F2_ST_BOTTOM_UP);
    {  // exact singularity anywhere in the instance: stop it where it is (status 2), as the reference's LinAlgError break does
      const unsigned sa = icell[15 * IW + l];
      if (!st.done && sa) { st.status = GS_STATUS_SINGULAR; st.done = true; }
    }
    const bool upd = !st.done;
    double kcs[16];                  // the series of the voltage rotation, in scalar registers for the whole pass
    {
      const GS_CONST double* kc0 = (const GS_CONST double*)kF2Series;
#pragma unroll
      for (int q = 0; q < 16; ++q) kcs[q] = kc0[q];
    }
    stp.hit(F2_ST_INIT);
    // ---------------- back substitution: x_k = s_k - sum_j T(k, j) x_j, levels downwards; the x slots share the messages' body ----------------
    {
      int lv = NL - 1;
      f2_i4 ia_next = items[(size_t)(NI - 1) * (HV * 4)];         // bus | nbr, flags: the item's first 16 bytes, a row ahead
#pragma nounroll
      for (int j = NI - 1; j >= 0; --j) {
        const int lev = rinfo[4 * j];
        const f2_i4 ia = ia_next;
        if (j > 0) ia_next = items[(size_t)(j - 1) * (HV * 4)];
        if (lev < 0) continue;
        const int lev_next = j > 0 ? rinfo[4 * (j - 1)] : -1;     // (the rows of a wave are its first ones: j - 1 exists whenever j > 0)
        const int g_row = rinfo[4 * j + 1] & 255;
        const int fl = ia.y;
        double t00 = 0.0, t01 = 0.0, t10 = 0.0, t11 = 0.0, s0 = 0.0, s1 = 0.0;
        F2_ROW(j, (t00 = T00[RQ], t01 = T01[RQ], t10 = T10[RQ], t11 = T11[RQ], s0 = sx0[RQ], s1 = sx1[RQ]));
        while (lv > lev) { f2_lds_sync(); --lv; }
        const double2 xj = f2_ld2((fl & GS_MESH_F_NBR) ? unit_at(6u + hi16(ia.x)) : R0l);
        double p0 = __builtin_fma(t00, xj.x, t01 * xj.y), p1 = __builtin_fma(t10, xj.x, t11 * xj.y);
        if (g_row > 1) {                                          // a group's partial sums, added by its lane 0 in lane order
          // (the sub-group index through an opaque copy: the eight read addresses below do not depend on the row, the compiler
          // computed them once in front of the loop, had no registers for them, and every one came back from scratch memory behind a
          // wait for ALL outstanding loads -- the next row's item among them: most of the back substitution's time)
          unsigned hvx = (unsigned)hv; F2_OPAQUE(hvx);
          f2_st2(scr + hvx * UB, make_double2(p0, p1));
          asm volatile("" ::: "memory");
          const unsigned gl = ((unsigned)fl >> GS_MESH_F_G_SHIFT) & 15u;
          double a0 = 0.0, a1 = 0.0;
          auto gsum = [&](auto gc, int t_first) {                 // all reads of a batch first, then the sum in lane order
            constexpr int G = decltype(gc)::value;
            double2 q[G];
#pragma unroll
            for (int t2 = 0; t2 < G; ++t2) q[t2] = f2_ld2(scr + ((hvx + (unsigned)(t_first + t2)) & 7u) * UB);
#pragma unroll
            for (int t2 = 0; t2 < G; ++t2) { if ((unsigned)(t_first + t2) < gl) { a0 += q[t2].x; a1 += q[t2].y; } }
          };
          if (g_row == 2) gsum(std::integral_constant<int, 2>{}, 0);
          else { gsum(std::integral_constant<int, 4>{}, 0); if (g_row > 4) gsum(std::integral_constant<int, 4>{}, 4); }
          p0 = a0; p1 = a1;
        }
        const double x0 = s0 - p0, x1 = s1 - p1;
        if (fl & GS_MESH_F_PIVOT) f2_st2(unit_at(6u + lo16(ia.x)), make_double2(x0, x1));
        if (lev_next != lev && lv > 0) { f2_lds_sync(); --lv; }   // x is out: the barrier before everything else the row still does
        // corrections (power_flow.py:315-327) as a rotation and scaling of (e, f), see the radial member above -- here, behind the
        // barrier, while the level below is at work (nobody reads a voltage before the next mismatch; as a pass of its own the
        // update was 10 k cycles per iteration)
        if (upd && (fl & GS_MESH_F_PIVOT)) {
          const unsigned vo = vslot(lo16(ia.x));
          const double2 v = f2_ld2(vo);
          const double v2 = __builtin_fma(v.x, v.x, v.y * v.y), rvm0 = f2_rsq(v2), vm0 = v2 * rvm0;
          const double dth = C.alpha * x0, vmn = vm0 + C.alpha * x1;
          const bool big = __any(fabs(dth) > 0.5);
          double h = dth;
          if (big) {
            const double k = rint(dth * 0.15915494309189535);
            h = __builtin_fma(-k, 6.283185307179586, dth);
            h = __builtin_fma(-k, 2.4492935982947064e-16, h);
            h *= 0.125;
          }
          const double* kc = kcs;
          const double z = h * h;
          double sp = kc[0];
          sp = __builtin_fma(sp, z, kc[1]); sp = __builtin_fma(sp, z, kc[2]); sp = __builtin_fma(sp, z, kc[3]);
          sp = __builtin_fma(sp, z, kc[4]); sp = __builtin_fma(sp, z, kc[5]); sp = __builtin_fma(sp, z, kc[6]); sp = __builtin_fma(sp, z, kc[7]);
          double sn = h - h * z * sp;
          double cp = kc[8];
          cp = __builtin_fma(cp, z, kc[9]); cp = __builtin_fma(cp, z, kc[10]); cp = __builtin_fma(cp, z, kc[11]);
          cp = __builtin_fma(cp, z, kc[12]); cp = __builtin_fma(cp, z, kc[13]); cp = __builtin_fma(cp, z, kc[14]); cp = __builtin_fma(cp, z, kc[15]);
          double cs = __builtin_fma(z, cp, 1.0);
          if (big) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { const double c2 = __builtin_fma(cs, cs, -(sn * sn)), s2 = 2.0 * cs * sn; cs = c2; sn = s2; }
          }
          const double ratio = vmn * rvm0;
          f2_st2(vo, make_double2(ratio * (v.x * cs - v.y * sn), ratio * (v.x * sn + v.y * cs)));
        }
      }
      while (lv > 0) { f2_lds_sync(); --lv; }
    }
    stp.hit(F2_ST_TOP_DOWN);
    f2_lds_sync();                   // the new voltages are read by the neighbours' lanes in the next mismatch
    stp.hit(14);
    stale = true;
  }
  if (stale) { (void)mismatch(false, false); }          // iteration cap reached after an update: the losses sum at the final voltages
#undef F2_ROW
#undef F2_ROW_CASE
  } else {
  // ================= the sweeps =================
  // The algorithm is the one of fbs_loop_flow (kernels_solve.hip): flat start, mismatch S_spec - V conj(I) evaluated on
  // the way down, one division per bus and iteration, converged lanes keep their currents.  What differs is how the two
  // tree recurrences are evaluated.  With per-bus messages a sweep costs one LDS hand-off per tree LEVEL (11 on the
  // IEEE-123 feeder, ~900 cycles each with sixteen waves polling), and the six to eight sweeps of a solve were 60 % of
  // the step.  Both recurrences are sums, so they have log-depth forms with plain barriers in between:
  //   backward  J_i = -I_i + sum_children J_c = sum of -I over the SUBTREE of i.  The buses sit in preorder of the forest,
  //             a subtree is the contiguous range [p_i, last_i], so J_i = Q[p_i - 1] - Q[last_i] with Q the inclusive
  //             prefix sums of I: a scan of <= 128 values per instance = local scan of a lane's items, the 16 wave
  //             totals through LDS, one barrier, offsets, Q to LDS, one barrier, one read.
  //   forward   V_i = V_slack - sum of z_k J_k over the PATH root .. i.  Pointer jumping, radix 4: S_i starts as D_i = z_i J_i
  //             and in round r adds the S of the ancestors 4^r, 2 * 4^r, 3 * 4^r steps up (ancestor tables from the host);
  //             ceil(log4 depth) rounds, two LDS buffers in turn, a barrier per round.  (Round 2 jumped by powers of two:
  //             four rounds for the IEEE-123 tree = three more LDS stores and two more barriers per bus and sweep.)
  // The sums associate differently from the sequential recurrences: results agree with them to a few ulp of |V|
  // (absolute ~1e-16 in J, ~1e-15 in V; the tests compare at 1e-12).
  const unsigned bufA = 0u, bufB = (unsigned)o_tile;
  f2_v2 F2_AS3* const tot_lds = F2_P(f2_v2, o_red);            // [16 waves][32 lanes] wave totals of the scan
  // backward sweep: J of this lane's buses from the injection currents of all buses.  at_barrier() runs behind the sweep's
  // first barrier and ends the sweep there if it returns true (the flat-start convergence check rides on that barrier)
  auto backward = [&](auto&& at_barrier) {
    double qr[NI], qi[NI];
    // local inclusive scan of +I (J = -(Q[last] - Q[p - 1]) is formed as Q[p - 1] - Q[last] below: no negations, no 0 + x)
    double ar = IR[0], ai = II[0];
    qr[0] = ar; qi[0] = ai;
#pragma unroll
    for (int j = 1; j < NI; ++j) { ar += IR[j]; ai += II[j]; qr[j] = ar; qi[j] = ai; }
    // totals of the sub-groups before this one inside the wave (in sub-group order), and of the whole wave
    double pre_r, pre_i, wt_r, wt_i;
    f2_xscan<IW>(ar, l, hv, pre_r, wt_r);
    f2_xscan<IW>(ai, l, hv, pre_i, wt_i);
    if (hv == 0) { f2_v2 t; t.x = wt_r; t.y = wt_i; tot_lds[wave * IW + l] = t; }
    f2_lds_sync();
    if (at_barrier()) return;
    double br = 0.0, bi = 0.0;                                    // sum of the totals of the waves before this one
    if constexpr (NW % HV == 0 && HV > 1) {
      // every sub-group reads the totals of NW / HV waves (its share, in wave order) and the shares are added across the
      // sub-groups without the LDS pipe: two 16-byte reads per lane instead of eight for the 8-wave member
      constexpr int TPS = NW / HV;
      f2_v2 tw[TPS];
#pragma unroll
      for (int w = 0; w < TPS; ++w) tw[w] = tot_lds[(hv * TPS + w) * IW + l];
#pragma unroll
      for (int w = 0; w < TPS; ++w) { const bool before = hv * TPS + w < wave; br += before ? tw[w].x : 0.0; bi += before ? tw[w].y : 0.0; }
      br = f2_xsum<IW>(br, l); bi = f2_xsum<IW>(bi, l);
    } else {
      constexpr int TB = NW < 8 ? NW : 8;
#pragma unroll
      for (int w0 = 0; w0 < NW; w0 += TB) {                       // eight totals per LDS round trip
        f2_v2 tw[TB];
#pragma unroll
        for (int w = 0; w < TB; ++w) tw[w] = tot_lds[(w0 + w) * IW + l];
#pragma unroll
        for (int w = 0; w < TB; ++w) { if (w0 + w < wave) { br += tw[w].x; bi += tw[w].y; } }
      }
    }
    br += pre_r; bi += pre_i;
    // Q of position p is filed under the BUS at that position (buffer B shares the slot numbering of buffer A, whose
    // "no ancestor" slot must stay zero); idle positions all file under the DUMMY slot, which nobody reads
#pragma unroll
    for (int j = 0; j < NI; ++j) f2_st2(bufB + f2_slot(ibus[j], l), make_double2(br + qr[j], bi + qi[j]));      // Q[p]
    f2_lds_sync();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const double2 ql = f2_ld2(bufB + f2_slot(ilast[j], l));       // Q at the last position of the bus's subtree
      const double er = j ? br + qr[j ? j - 1 : 0] : br, ei = j ? bi + qi[j ? j - 1 : 0] : bi;             // Q[p - 1]
      // (an idle position gets some finite difference of prefix sums: its impedance is 0, so is its drop, and nothing else reads its J)
      JR[j] = er - ql.x; JI[j] = ei - ql.y;
    }
  };
  stp.hit(F2_ST_INIT);
  const unsigned long long tol_fix = f2_fix_tolerance(C.tolerance);
  const bool any_root = __any(roots != 0u);                   // (the slack's one to three children sit in one or two waves)
  {  // at the flat start: every voltage but the slack's is 1, S_calc = conj(K) with K = y (1 - V_slack) at the roots
    double lmax = 0.0, lsum = 0.0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const GsF2Rec* q = rec0 + j;
      const double p = Pj[j];
      const bool root = (roots >> j) & 1u;
      const double ep = root ? f2_ld(f2_slot(F.slack, l)) : 1.0;
      const double dr = 1.0 - ep;
      const double kr = q->yr * dr, ki = q->yi * dr;
      const double pc = kr, qc = -ki;
      const double dP = p - pc, dQ = 0.0 - qc;
      lmax = fmax(lmax, fmax(fabs(dP), fabs(dQ)));
      lsum += fabs(dP) + fabs(dQ);
      psum += pc;
      if (root) psum -= ep * kr;                              // the slack's share: Re(V_s conj(-K_root))
      IR[j] = p; II[j] = -0.0;                                // I = conj(S_spec / V) at V = 1
    }
    if (!(lsum < INFINITY)) lmax = INFINITY;     // a NaN or infinite mismatch anywhere shows in the sum (fmax would drop a NaN)
    // the flat-start check shares the first backward sweep's barrier: the currents of that sweep are the same whether an
    // instance stops here or not (a lane that stops here reports the flat start, epilogue)
    const int c0 = post_max_sum(lmax, lsum);
    backward([&]() -> bool {
      unsigned long long sum;
      const double mm = read_max_sum(c0, sum);
      f2_check_sum(st, mm, sum, tol_fix, 0);
      return __all(st.done);
    });
    stp.hit(F2_ST_BOTTOM_UP);
  }
  const double vs_r = f2_ld(f2_slot(F.slack, l));               // the slack's set point (real)
  const int R2 = F.n_jump;                                      // even
  for (int it = 0; it < C.max_iterations && !__all(st.done); ++it) {
    const bool upd = !st.done;
    double lmax = 0.0, pnew = 0.0, lsum = 0.0;
    // forward sweep by pointer jumping; buffers alternate so that the last round reads B (then A may take the voltages)
    double sr[NI], si[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const double2 z = f2_ld2(o_z + 16u * ibus[j]);
      sr[j] = __builtin_fma(JR[j], z.x, -(JI[j] * z.y)); si[j] = __builtin_fma(JR[j], z.y, JI[j] * z.x);      // D = z J
      f2_st2(bufA + f2_slot(ibus[j], l), make_double2(sr[j], si[j]));
    }
    if (any_root) {                                             // the slack's share of the losses sum: Re(V_s conj(J_root)), V_s real
#pragma unroll
      for (int j = 0; j < NI; ++j) { if ((roots >> j) & 1u) pnew += vs_r * JR[j]; }
    }
    f2_lds_sync();
    for (int r = 0; r < R2; ++r) {
      const unsigned rd = (r & 1) ? bufB : bufA, wr = (r & 1) ? bufA : bufB;
      // radix 4: the partial sums of the ancestors 4^r, 2 * 4^r and 3 * 4^r steps up (one 16-byte table entry per bus and
      // round), added in that order -- depth 16 in two rounds, one LDS store per bus between them.  (Three 10-bit slot
      // numbers in one word were tried: a quarter of the table's LDS bytes, but a bit-field extract more per gather, and the
      // kernel is bound by vector instruction issue.)
      f2_i4 aq[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) aq[j] = *F2_P(const f2_i4, (unsigned)o_anc + 16u * (unsigned)(r * nsl + ibus[j]));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double2 sa[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) sa[j] = f2_ld2(rd + f2_slot(aq[j][k], l));
#pragma unroll
        for (int j = 0; j < NI; ++j) { sr[j] += sa[j].x; si[j] += sa[j].y; }
      }
      if (r + 1 < R2) {
#pragma unroll
        for (int j = 0; j < NI; ++j) f2_st2(wr + f2_slot(ibus[j], l), make_double2(sr[j], si[j]));
        f2_lds_sync();
      }
    }
    stp.hit(F2_ST_TOP_DOWN);
    // V = V_slack - S; mismatch and sum of P_calc at the new voltages (power_flow.py:150-168).  The voltages go to buffer A
    // (its last reader was round R2 - 2, a barrier ago) when the loop ends -- `publish` below, ONE exit: with a `break` at the
    // cap and another behind the check the compiler duplicated the body and spilled 50 registers.
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const double en = vs_r - sr[j], fn = -si[j];            // (a negation rides on its consumers' operands; 0.0 - x is an instruction)
      sr[j] = en; si[j] = fn;
      const double pc = __builtin_fma(en, IR[j], fn * II[j]), qc = __builtin_fma(fn, IR[j], -(en * II[j]));     // S_calc = V_new conj(I_old)
      const double dP = Pj[j] - pc;                           // dQ = 0 - Q_calc enters through its magnitude only
      lmax = fmax(lmax, fmax(fabs(dP), fabs(qc)));
      lsum += fabs(dP) + fabs(qc);
      pnew += pc;
    }
    if (!(lsum < INFINITY)) lmax = INFINITY;
    stp.hit(F2_ST_MISMATCH);
    if (upd) psum = pnew;                                     // losses at the voltages of this sweep
    auto publish = [&]() {
#pragma unroll
      for (int j = 0; j < NI; ++j) f2_st2(bufA + f2_slot(ibus[j], l), make_double2(sr[j], si[j]));
    };
    const bool cap = it + 1 >= C.max_iterations;              // iteration cap: mismatch / count stay the last check's
    if (!cap) {
      unsigned long long sum;
      const double mm = wg_max_sum(lmax, lsum, sum);
      stp.hit(F2_ST_FLAG);
      f2_check_sum(st, mm, sum, tol_fix, it + 1);
    }
    if (cap || __all(st.done)) { publish(); break; }
    // I_new = conj(S_spec / V_new) for the lanes that go on; a lane that has converged keeps the current that produced
    // its voltages: its J and V repeat bit for bit while the rest of the group iterates
    // (one branch around the block instead of a select per component; an idle position has P = 0 and so a zero current)
    if (!st.done) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const double rd = f2_rcp(__builtin_fma(sr[j], sr[j], si[j] * si[j]));
        IR[j] = (Pj[j] * sr[j]) * rd; II[j] = (Pj[j] * si[j]) * rd;
      }
    }
    backward([]() -> bool { return false; });
    stp.hit(F2_ST_BOTTOM_UP);
  }
  }
  f2_sync();                         // every slot final
  stp.hit(F2_ST_FINAL_MISMATCH);

  // ================= epilogue =================
  // a lane that stopped at the very first check keeps the flat start (its slots went on iterating with its group)
  const bool flat_lane = st.done && st.iters <= 1, any_flat = __any(flat_lane);
  auto final_ef = [&](int bus) -> double2 {
    double2 v = f2_ld2(f2_slot(bus, l));
    if (any_flat) { const double fv = T.fixed_v[bus] ? T.v_set[bus] : 1.0; if (flat_lane) v = make_double2(fv, 0.0); }
    return v;
  };
  // scalar state of wave 0's part, requested now, used after the second barrier
  double totloss0 = 0.0, f_old = 0.0, viol0 = 0.0, step0 = 0.0, eprew0 = 0.0;
  if (wave == 0) { totloss0 = ROW(R.TOTLOSS); f_old = ROW(R.FREQ); viol0 = ROW(R.VIOL); eprew0 = ROW(R.EPREW); step0 = valid ? kold + 1.0 : kold; }
  const bool chk = CHK && FC.enabled;
  if (chk && wave == 0 && hv == 0) { cell[l] = 0ull; cell[IW + l] = 0ull; }      // the first two convergence cells take two maxima of the checks
  const GsChecksCfg& K = FC.C;
  double* Pv = chk ? FC.prev + (size_t)g * (n + 1) * GS_LANES + L : nullptr;
  // The previous step's |V| of this lane's buses (and the previous frequency, and the three words of the monitor's state
  // wave 0 updates at the very end), asked for NOW: read where they are used, each waited behind the stores issued before it
  // (vector loads and stores return in order) -- 1.5 us of the fused step.
  constexpr int BUS_PASSES = CHK ? NI + 1 : 1;               // passes of the bus loop: ceil(n / (HV * NW)) <= NI + 1
  double pvp[BUS_PASSES];
  double pv_f = 0.0;
  int ck_has_prev = 0, ck_consec = 0, ck_emode = 0;
  if constexpr (CHK) {
    if (chk) {
#pragma unroll
      for (int q = 0; q < BUS_PASSES; ++q) { const int i0 = wave * HV + hv + q * HV * NW; pvp[q] = i0 < n ? Pv[(size_t)i0 * GS_LANES] : 0.0; }
      if (wave == 0) pv_f = Pv[(size_t)n * GS_LANES];
      if (wave == 0 && hv == 0 && b < (int)FC.Bp && valid) { ck_has_prev = FC.state[b]; ck_consec = FC.state[FC.Bp + b]; ck_emode = FC.state[2 * FC.Bp + b]; }
    }
  }
  int k_nlow = 0, k_nhigh = 0, k_mhigh = 0, k_mlow = 0, k_mem = 0, k_cover = 0, k_mover = 0, k_vbad = 0, k_fbad = 0;
  double k_dv = 0.0, k_ql = 0.0; int k_dvnan = 0, k_qlnan = 0;

  // lines (power_flow.py:340-356; Line.update_state, base.py:261-264): two lines per wave instruction
  int over = 0;
  for (int k0 = wave * HV + hv; k0 < ((m + HV * NW - 1) / (HV * NW)) * (HV * NW); k0 += HV * NW) {
    const bool on = k0 < m; const int k = on ? k0 : m - 1;
    const int li = T.lfrom[k], lj = T.lto[k];
    const double yr = T.lyr[k], yi = T.lyi[k], rating = T.lrating[k], rinv = T.lrating_inv[k];
    const double2 vi = final_ef(li), vj = final_ef(lj);
    const double dr = vi.x - vj.x, di = vi.y - vj.y;
    const double ir = __builtin_fma(yr, dr, -(yi * di)), ii = __builtin_fma(yr, di, yi * dr);        // I = y (Vi - Vj)
    const double sr = __builtin_fma(vi.x, ir, vi.y * ii), si = __builtin_fma(vi.y, ir, -(vi.x * ii));  // S = Vi conj(I)
    const double ql = (rating > 0.0) ? gs_div_by(sqrt(__builtin_fma(sr, sr, si * si)), rating, rinv) : 0.0;
    const double ld = (rating > 0.0) ? gs_div_by(fabs(sr), rating, rinv) : 0.0;
    if (on) {
      f2_row(S, R.LOAD + k) = ql;
      if (!PA.lean) f2_pair(S, R.FLOW + k) = make_double2(sr, ld);
      f2_st2(o_tile + (unsigned)k * SB + ((unsigned)l << 4), make_double2(sr, ld));
      over += (ld > 0.8) ? 1 : 0;
      if (chk) {
        const double cld_ = K.stride_cload == 2 ? ld : ql;             // which loading the limits apply to
        const bool co = cld_ > K.c_load, mo = cld_ > K.m_load;         // safety.py:150-154, :364-365
        k_cover += co; k_mover += mo;
        if (ql != ql) k_qlnan = 1; else k_ql = fmax(k_ql, ql);
        if (!(fabs(sr) < INFINITY)) k_fbad = 1;
        if (FC.line_mask) FC.line_mask[((size_t)g * m + k) * GS_LANES + L] = (uint8_t)(co | (mo << 1));
      }
    }
  }
  stp.hit(F2_ST_EPI_LINES);
  f2_lds_sync();                     // every (e, f) has been read: the slots may now be converted in place
  // buses: (e, f) -> (|V|, angle) in place (the slots become the observation tile), state rows, reward / flag partials
  double dev = 0.0, vmax = -INFINITY, vmin = INFINITY;
  int vflags = 0;
  auto bus_pass = [&](int i0, double pv) {
    const bool on = i0 < n; const int i = on ? i0 : n - 1;
    double2 ef = final_ef(i);
    if (!on) ef = make_double2(1.0, 0.0);                     // keep the wave on the series branch of the angle
    const double v = sqrt(__builtin_fma(ef.x, ef.x, ef.y * ef.y));
    const double ang = f2_angle(ef.y, ef.x);
    if (on) {
      f2_st2(f2_slot(i, l), make_double2(v, ang));
      if (!PA.lean) f2_pair(S, R.VM + i) = make_double2(v, ang);
      dev += fabs(v - 1.0);                                   // reward / flags, grid_env.py:790-792, base.py:156-159
      vmax = fmax(vmax, v); vmin = fmin(vmin, v);
      vflags |= (v > E.v_max) ? 1 : 0;
      vflags |= (v < E.v_min) ? 2 : 0;
      if (chk) {
        const bool cl = v < K.c_vlo, ch = !cl && v > K.c_vhi;                // safety.py:129-137 (elif)
        const bool mh = v > K.m_vhi, ml = v < K.m_vlo;                       // :333-337
        const bool em = v > K.m_evhi || v < K.m_evlo;                        // :340-341
        k_nlow += cl; k_nhigh += ch; k_mhigh += mh; k_mlow += ml; k_mem += em;
        const double d = fabs(v - pv);                                       // :168 (np.max propagates NaN)
        if (d != d) k_dvnan = 1; else k_dv = fmax(k_dv, d);
        Pv[(size_t)i * GS_LANES] = v;                                        // :181-184
        if (!(fabs(v) < INFINITY)) k_vbad = 1;                               // robust_power_flow.py:643-647
        if (FC.bus_mask) FC.bus_mask[((size_t)g * n + i) * GS_LANES + L] = (uint8_t)(cl | (ch << 1) | (ml << 2) | (mh << 3) | (em << 4));
      }
    }
  };
  const int bus_passes = (n + HV * NW - 1) / (HV * NW);
  if constexpr (CHK) {
    if (chk) {
#pragma unroll
      for (int q = 0; q < BUS_PASSES; ++q) { if (q < bus_passes) bus_pass(wave * HV + hv + q * HV * NW, pvp[q]); }
    } else {
      for (int i0 = wave * HV + hv; i0 < bus_passes * (HV * NW); i0 += HV * NW) bus_pass(i0, 0.0);
    }
  } else {
    // (the kernels without the checks keep the loop as it was written for them: the same statements, no lambda in between --
    // the Newton-Raphson member came out 0.7 % slower through it)
    for (int i0 = wave * HV + hv; i0 < bus_passes * (HV * NW); i0 += HV * NW) {
      const bool on = i0 < n; const int i = on ? i0 : n - 1;
      double2 ef = final_ef(i);
      if (!on) ef = make_double2(1.0, 0.0);                     // keep the wave on the series branch of the angle
      const double v = sqrt(__builtin_fma(ef.x, ef.x, ef.y * ef.y));
      const double ang = f2_angle(ef.y, ef.x);
      if (on) {
        f2_st2(f2_slot(i, l), make_double2(v, ang));
        if (!PA.lean) f2_pair(S, R.VM + i) = make_double2(v, ang);
        dev += fabs(v - 1.0);                                   // reward / flags, grid_env.py:790-792, base.py:156-159
        vmax = fmax(vmax, v); vmin = fmin(vmin, v);
        vflags |= (v > E.v_max) ? 1 : 0;
        vflags |= (v < E.v_min) ? 2 : 0;
      }
    }
  }
  stp.hit(F2_ST_EPI_BUSES);
  {  // partial results: sums keep per-wave partials (added in wave order below), everything else is an integer atomic
    const double ls2 = f2_xsum<IW>(psum, l), dv2 = f2_xsum<IW>(dev, l);            // over the wave's sub-groups, in sub-group order
    if (hv == 0) { red_lsum[wave * IW + l] = ls2; red_dev[wave * IW + l] = dv2; }
    if (vmax == vmax && vmax > -INFINITY) atomicMax(cell + 3 * IW + l, f2_bits(vmax));       // |V| >= 0: bit patterns order like the values
    if (vmin == vmin && vmin < INFINITY) atomicMin(cell + 4 * IW + l, f2_bits(vmin));
    if (over) atomicAdd(icell + 0 * IW + l, (unsigned)over);
    if (vflags) atomicOr(icell + 1 * IW + l, (unsigned)vflags);
    if (chk) {
      if (k_nlow) atomicAdd(icell + 2 * IW + l, (unsigned)k_nlow);
      if (k_nhigh) atomicAdd(icell + 3 * IW + l, (unsigned)k_nhigh);
      if (k_mhigh) atomicAdd(icell + 4 * IW + l, (unsigned)k_mhigh);
      if (k_mlow) atomicAdd(icell + 5 * IW + l, (unsigned)k_mlow);
      if (k_mem) atomicAdd(icell + 6 * IW + l, (unsigned)k_mem);
      if (k_cover) atomicAdd(icell + 7 * IW + l, (unsigned)k_cover);
      if (k_mover) atomicAdd(icell + 8 * IW + l, (unsigned)k_mover);
      const unsigned fl = (unsigned)(k_vbad | (k_fbad << 1) | (k_dvnan << 2) | (k_qlnan << 3));
      if (fl) atomicOr(icell + 9 * IW + l, fl);
      // the convergence cells are free by now: [0] max |dV|, [1] max loading (both >= 0)
      atomicMax(cell + 0 * IW + l, f2_bits(k_dv));
      atomicMax(cell + 1 * IW + l, f2_bits(k_ql));
    }
  }
  // (the convergence cells [0], [1] were last read before the solver's final barrier; cell [2] is not reused)
  f2_lds_sync();
  stp.hit(F2_ST_EPI_REDUCE);

  constexpr int WO0 = NW > 1 ? 1 : 0;             // the waves that write the observation block out: all but wave 0 (which has the scalar part)
  if (wave >= WO0) {
    // ---- observation block of this workgroup's IW instances, straight from the two LDS tiles: lane = column pair ----
    // (|V|, angle) per bus and (flow, loading) per line are adjacent observation columns (grid_env.py:758-763)
    if (PA.out != nullptr) {
      for (int r = wave - WO0; r < IW; r += NW - WO0) {
        const int br = g * GS_LANES + hs * IW + r;
        if (br >= B) continue;
        double* o = PA.out + (size_t)br * PA.obs_dim;
        if (!(PA.obs_dim & 1)) {            // every row of the block starts on a 16-byte boundary: one store per column pair
          for (int cp = lane; cp < n; cp += 64) f2_stream2(o + 2 * cp, f2_ld2(f2_slot(cp, r)));
          for (int cp = lane; cp < m; cp += 64) f2_stream2(o + 2 * n + 2 * cp, f2_ld2(o_tile + (unsigned)cp * SB + ((unsigned)r << 4)));
        } else {
          for (int cp = lane; cp < n; cp += 64) { const double2 v = f2_ld2(f2_slot(cp, r)); o[2 * cp] = v.x; o[2 * cp + 1] = v.y; }
          for (int cp = lane; cp < m; cp += 64) {
            const double2 v = f2_ld2(o_tile + (unsigned)cp * SB + ((unsigned)r << 4)); o[2 * n + 2 * cp] = v.x; o[2 * n + 2 * cp + 1] = v.y; }
        }
      }
    }
    stp.hit(F2_ST_EPILOGUE);
    if (C.stamps && C.block_times && lane == 0 && bid < GS_STAMP_BLOCKS)
      atomicMax(&C.stamps[17 + 2 * bid], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    if (wave != 0) return;
  }
  // ---- wave 0: everything of step() that follows the load flow, per instance (grid_env.py:553-617) ----
  double losses = 0.0;
  dev = 0.0;
  for (int w = 0; w < NW; ++w) { losses += red_lsum[w * IW + l]; dev += red_dev[w * IW + l]; }
  vmax = f2_dbl(cell[3 * IW + l]); vmin = f2_dbl(cell[4 * IW + l]);
  over = (int)icell[0 * IW + l]; vflags = (int)icell[1 * IW + l];
  const bool st_lane = hv == 0;                       // both halves hold the same values; half 0 stores
  if (st_lane) {
    ROW(R.LOSSES) = losses; ROW(R.MAXMIS) = st.mm; ROW(R.ITERS) = (double)st.iters; ROW(R.CONV) = (double)st.conv; ROW(R.STATUS) = (double)st.status;
  }
  const double dt = E.timestep;
  double total_gen = 0.0, total_curt = 0.0;                            // grid_env.py:744-751, 807-816
  for (int gi = 0; gi < ng; ++gi) {
    const double p = env_lds[(F.env_genp + gi) * IW + l];
    total_gen += p;
    total_curt += p * (1.0 - env_lds[(F.env_curt + gi) * IW + l]);
  }
  const double totloss = totloss0 + gs_div_by(losses * dt, 3600.0, 1.0 / 3600.0);      // grid_env.py:739
  const double imbalance = gs_div_by(total_gen - total_load - losses * E.power_base, 1e6, 1.0 / 1e6);
  double f = f_old;                                                    // dynamics.py:260-273
  f += ((imbalance - E.D * (f - E.f0)) / (2.0 * E.H * E.f0)) * dt;
  f = fmax(55.0, fmin(65.0, f));
  double reward = 0.0;                                                 // grid_env.py:785-826
  reward -= dev * 10.0;
  reward -= fabs(f - 60.0) * 20.0;
  reward -= (double)(over * 50);
  reward -= totloss * 0.1;
  reward += (total_gen - total_curt) * 1e-5;
  for (int q = 0; q < nb; ++q) {
    const double soc = env_lds[(F.env_soc + q) * IW + l];
    reward += (soc >= 0.2 && soc <= 0.8) ? 1.0 : -5.0;
  }
  const int vhigh = vflags & 1, vlow = (vflags >> 1) & 1, fhigh = f > E.f_max, flow_ = f < E.f_min;
  double viol = viol0, trunc = 0.0;
  if (vhigh | vlow | fhigh | flow_) {
    viol += 1.0;
    if (viol > 10.0) { trunc = 1.0; reward -= E.safety_penalty; }     // grid_env.py:604-606
  }
  if (st_lane) {
    ROW(R.TOTLOSS) = totloss; ROW(R.FREQ) = f; ROW(R.VIOL) = viol; ROW(R.TRUNC) = trunc;
    ROW(R.TERM) = (step0 >= (double)E.episode_length) ? 1.0 : 0.0;     // base.py:140-142
    ROW(R.REWARD) = reward; ROW(R.EPREW) = eprew0 + reward;
    ROW(R.VMAX) = vmax; ROW(R.VMIN) = vmin;
    ROW(R.VFLAGS + 0) = (double)vhigh; ROW(R.VFLAGS + 1) = (double)vlow;
    ROW(R.VFLAGS + 2) = (double)fhigh; ROW(R.VFLAGS + 3) = (double)flow_;
    if (RS.active && valid) {        // the rollout's [T][B] arrays: reward, and bit 0 terminated / bit 1 truncated
      RS.rew[(size_t)RS.t * B + b] = reward;
      RS.done[(size_t)RS.t * B + b] = (uint8_t)((step0 >= (double)E.episode_length ? 1 : 0) | (trunc != 0.0 ? 2 : 0));
    }
    // the observation columns that are neither bus nor line pairs: frequency, renewable powers, battery state
    // (grid_env.py:766, 773-781); the static load columns in between are written at reset and never change
    if (PA.out != nullptr && valid) {
      double* o = PA.out + (size_t)b * PA.obs_dim;
      const int c_f = 2 * n + 2 * m, c_g = c_f + 1 + 2 * nl_, c_b = c_g + ng;
      o[c_f] = f;
      for (int gi = 0; gi < ng; ++gi) o[c_g + gi] = env_lds[(F.env_genp + gi) * IW + l];
      for (int q = 0; q < nb; ++q) { o[c_b + 2 * q] = env_lds[(F.env_soc + q) * IW + l]; o[c_b + 2 * q + 1] = env_lds[(F.env_batp + q) * IW + l]; }
    }
  }
  if (chk && st_lane && b < (int)FC.Bp && valid) {
    // same finalisation as gs_k_checks (kernels_checks.hip) / the first-generation fused epilogue, on the values of this very step
    const int c_nlow = (int)icell[2 * IW + l], c_nhigh = (int)icell[3 * IW + l], m_nhigh = (int)icell[4 * IW + l], m_nlow = (int)icell[5 * IW + l];
    const int m_nem = (int)icell[6 * IW + l], c_nover = (int)icell[7 * IW + l], m_nover = (int)icell[8 * IW + l];
    const unsigned fl = icell[9 * IW + l];
    const int vbad = fl & 1, fbad = (fl >> 1) & 1;
    k_dv = (fl & 4) ? NAN : f2_dbl(cell[0 * IW + l]);
    k_ql = (fl & 8) ? NAN : f2_dbl(cell[1 * IW + l]);
    if (m == 0) k_ql = -INFINITY;
    int32_t* has_prev = FC.state + b; int32_t* consec = FC.state + FC.Bp + b; int32_t* emode = FC.state + 2 * FC.Bp + b;
#define OI(k) FC.out_i[(size_t)(k) * FC.Bp + b]
#define OF(k) FC.out_f[(size_t)(k) * FC.Bp + b]
    const int c_flow = f < K.c_flo, c_fhigh = !c_flow && f > K.c_fhi;        // safety.py:140-147
    const double vrate = k_dv / K.dt;                                        // NaN stays NaN
    const double frate = fabs(f - pv_f) / K.dt;                              // :174
    const int hp = ck_has_prev;
    const int c_vr = hp && vrate > K.c_rocv, c_fr = hp && frate > K.c_rocf;
    Pv[(size_t)n * GS_LANES] = f; *has_prev = 1;
    const int c_total = c_nlow + c_nhigh + c_flow + c_fhigh + c_nover + c_vr + c_fr;
    OI(GS_CI_C_NLOW) = c_nlow; OI(GS_CI_C_NHIGH) = c_nhigh; OI(GS_CI_C_FLOW) = c_flow; OI(GS_CI_C_FHIGH) = c_fhigh;
    OI(GS_CI_C_NOVER) = c_nover; OI(GS_CI_C_VRATE) = c_vr; OI(GS_CI_C_FRATE) = c_fr; OI(GS_CI_C_TOTAL) = c_total;
    OI(GS_CI_C_SEVERITY) = c_total > 5 ? 3 : (c_total > 2 ? 2 : (c_total > 0 ? 1 : 0));
    OF(GS_CF_VRATE) = vrate; OF(GS_CF_FRATE) = frate;
    const int m_fhigh = f > K.m_fhi, m_flow = !m_fhigh && f < K.m_flo, m_fem = f > K.m_efhi || f < K.m_eflo;
    const int m_total = m_nhigh + m_nlow + m_nem + m_fhigh + m_flow + m_fem + m_nover;
    const int cs = m_total > 0 ? ck_consec + 1 : 0;
    const int trigger = (m_nem > 0) || m_fem || cs > 5 || m_total > 10;
    const int mode = ck_emode | trigger;
    *consec = cs; *emode = mode;
    OI(GS_CI_M_NHIGH) = m_nhigh; OI(GS_CI_M_NLOW) = m_nlow; OI(GS_CI_M_NEMERG) = m_nem; OI(GS_CI_M_FHIGH) = m_fhigh; OI(GS_CI_M_FLOW) = m_flow;
    OI(GS_CI_M_FEMERG) = m_fem; OI(GS_CI_M_NOVER) = m_nover; OI(GS_CI_M_TOTAL) = m_total; OI(GS_CI_M_ACTION) = trigger;
    OI(GS_CI_M_CONSEC) = cs; OI(GS_CI_M_EMODE) = mode;
    double q = 1.0;                                                          // robust_power_flow.py:615-657
    if (vmin < 0.8 || vmax > 1.2) q *= 0.3;
    else if (vmin < 0.9 || vmax > 1.1) q *= 0.7;
    if (m > 0 && k_ql == k_ql) { if (k_ql > 2.0) q *= 0.2; else if (k_ql > 1.0) q *= 0.5; }
    if (st.mm > K.q_tol * 100.0) q *= 0.6;
    if (st.iters <= 5) q *= 1.1; else if (st.iters > 20) q *= 0.9;
    q = fmin(q, 1.0);
    if (!st.conv || vbad || fbad) q = 0.0;
    OF(GS_CF_QUALITY) = q;
#undef OI
#undef OF
  }
  stp.hit(F2_ST_EPI_SCALARS);
  if (C.stamps && C.block_times && lane == 0 && bid < GS_STAMP_BLOCKS)
    atomicMax(&C.stamps[17 + 2 * bid], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// The step kernels read their arguments where they use them, through a pointer to the argument block the compiler cannot see
// through.  Taken from the formal parameters, all ~400 scalar words are loaded at the top of the kernel and, for lack of
// scalar registers, parked in vector lanes: a quarter of the vector instructions of a step were v_writelane / v_readlane,
// in kernels bound by vector instruction issue (rocprofv3 SQ_ACTIVE_INST_VALU: 75 % of the launch).
#define F2_ARGS_IN_PLACE                                                                                                   \
  const __attribute__((address_space(4))) char* ka_ = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(ka_));                                                                                            \
  const F2ArgBlock* A = (const F2ArgBlock*)ka_
#define F2_KERNELS_OCC(name, SOLVER, NW, NI, IW, OCC)                                                                      \
  extern "C" __global__ void __launch_bounds__(64 * NW) OCC                                                                \
  gs_k_step_##name(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,         \
                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS) { \
    F2_ARGS_IN_PLACE;                                                                                                      \
    f2_step<SOLVER, 0, NW, NI, IW>(A->T, A->F, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC, A->RS); \
  }                                                                                                                        \
  extern "C" __global__ void __launch_bounds__(64 * NW) OCC /* the step with the post-step checks in its epilogue */       \
  gs_k_stepc_##name(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,        \
                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS) { \
    F2_ARGS_IN_PLACE;                                                                                                      \
    f2_step<SOLVER, 1, NW, NI, IW>(A->T, A->F, A->R, A->C, A->E, A->slab, A->B, A->actions, A->total_load, A->PA, A->FC, A->RS); \
  }
#define F2_KERNELS(name, SOLVER, NW, NI, IW) F2_KERNELS_OCC(name, SOLVER, NW, NI, IW, )
#if defined(GS_BUILD_EXPERIMENTS)
F2_KERNELS(fbs_flow2, F2_FBS, 16, 4, 32)       // up to 128 buses below the slack, 32 instances per workgroup (GS_FLOW2_IW=32; the 16-instance member is the default)
#endif
F2_KERNELS(nr_flow2, F2_NR, 8, 8, 32)
F2_KERNELS(fbs_flow2s, F2_FBS, GS_F2S_WAVES, GS_F2S_ITEMS, GS_F2S_IW)        // up to 16 buses: 8 instances per workgroup, eight buses per wavefront
F2_KERNELS(nr_flow2s, F2_NR, GS_F2NS_WAVES, GS_F2NS_ITEMS, GS_F2S_IW)          // up to 4 groups of 8 same-level buses
// 16 instances per workgroup, four buses per wavefront, 8 waves: two workgroups share a CU (four waves per SIMD as above),
// so that one's LDS-bound solver phase runs beside the other's VALU-bound prologue / epilogue: +6 % at B = 8192, +11 % at
// 16384, +30 % at 4096 over the 32-instance member (the host's default for the sweep solver; 8 instances per workgroup
// with four workgroups per CU was tried too: -20 %, the per-instance scalar chains then fill an eighth of a wavefront)
F2_KERNELS_OCC(fbs_flow2h, F2_FBS, GS_F2H_WAVES, GS_F2H_ITEMS, GS_F2H_IW, __attribute__((amdgpu_waves_per_eu(4, 4))))
// up to 256 buses below the slack: eight buses per sub-group (twice the registers: two waves per SIMD), one workgroup per CU
F2_KERNELS(fbs_flow2x, F2_FBS, GS_F2X_WAVES, GS_F2X_ITEMS, GS_F2H_IW)
// Newton-Raphson on a meshed feeder (a few loops on a tree): 8 instances per workgroup, 4 wavefronts, up to GS_F2M_ITEMS rows each,
// two workgroups per CU (two waves per SIMD: 256 registers)
F2_KERNELS_OCC(nr_mesh2, F2_NRM, GS_F2M_WAVES, GS_F2M_ITEMS, GS_F2S_IW, __attribute__((amdgpu_waves_per_eu(2, 2))))
