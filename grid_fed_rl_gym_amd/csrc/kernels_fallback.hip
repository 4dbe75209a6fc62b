// kernels_fallback.hip -- the reference's linear-approximation fallback for rejected load flows, per instance.
//
// Reference arithmetic restated: LinearApproximationSolver.solve, environments/robust_power_flow.py:336-398 (relative to
// /root/reference/grid_fed_rl/): per-bus voltage rules, the angle recurrence over the lines in list order, one flow value
// for every line, 3 % losses.  The fallback chain (AdvancedRobustPowerFlowSolver.solve, :523-613) runs it on a network
// whose Newton-Raphson answer failed the quality gate; here it runs for the instances of the batch that were rejected
// (a host mask) or did not converge, and overwrites their solution rows -- the other instances keep theirs.
// Lane = instance, one wavefront per 64 instances; the work is a few hundred row accesses, no attempt at speed.
// No implicit contraction: the reference's expressions round after every operation.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]
#define ROW2(r) S.pair((size_t)(r) * GS_LANES)
#include "env_device.h"
#include "kernels.h"

#pragma clang fp contract(off)

// totals of the reference's two dicts at bus i (grid_env.py:683-720; the accumulation order of bus_injection)
__device__ __forceinline__ void bus_load_gen(const GsTables& T, const GsRows& R, GsLaneRows S, int i, double* ls_out, double* gs_out) {
  double ls = 0.0, gs = 0.0;
  for (int p = cld(T.bl_ptr, i); p < cld(T.bl_ptr, i + 1); ++p) ls += ROW(R.LOADP + cld(T.bl_idx, p));
  for (int p = cld(T.bg_ptr, i); p < cld(T.bg_ptr, i + 1); ++p) {
    const int g = cld(T.bg_idx, p);
    gs += ROW(R.GENP + g) * ROW(R.CURT + g);
  }
  for (int p = cld(T.bb_ptr, i); p < cld(T.bb_ptr, i + 1); ++p) {
    const double bp = ROW(R.BATP + cld(T.bb_idx, p));
    if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp);
  }
  *ls_out = ls; *gs_out = gs;
}

extern "C" __global__ void __launch_bounds__(64)
gs_k_fallback_linear(GsTables T, GsRows R, GsFallbackArgs A, double* __restrict__ slab, int B) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  const bool valid = b < B;
  const bool apply = valid && (A.mask ? A.mask[b] != 0 : (double)ROW(R.CONV) == 0.0);
  if (valid && A.applied) A.applied[b] = apply ? 1 : 0;
  if (!__any(apply)) return;
  const size_t row = (size_t)(valid ? b : 0) * T.n;
  // sums of the dict values, in the order the reference's dicts were filled
  double tl = 0.0, tg = 0.0;
  if (A.env_mode) {
    for (int k = 0; k < A.n_load_order; ++k) { double ls, gs; bus_load_gen(T, R, S, cld(A.load_order, k), &ls, &gs); tl += ls; }
    for (int k = 0; k < A.n_gen_order; ++k) { double ls, gs; bus_load_gen(T, R, S, cld(A.gen_order, k), &ls, &gs); tg += gs; }
  } else if (A.tot_load) {
    tl = valid ? A.tot_load[b] : 0.0; tg = valid ? A.tot_gen[b] : 0.0;
  } else {
    for (int i = 0; i < T.n; ++i) { tl += A.load_w[row + i]; tg += A.gen_w[row + i]; }
  }
  for (int i = 0; i < T.n; ++i) {                                         // robust_power_flow.py:356-370
    double ld, gn;
    if (A.env_mode) bus_load_gen(T, R, S, i, &ld, &gn);
    else { ld = A.load_w[row + i]; gn = A.gen_w[row + i]; }
    const bool slack = cld(T.th_free, i) == 0;
    double v = 1.0;
    if (!slack && ld != 0.0) v = fmin(fmax(1.0 - (ld / 10e6) * 0.05, 0.85), 1.15);
    if (!slack && gn != 0.0) v = fmin(v + (gn / 20e6) * 0.02, 1.10);
    if (apply) ROW2(R.VM + i) = make_double2(v, 0.0);
  }
  double pfl = 0.0;
  if (T.n > 1) {                                                          // :373-382
    pfl = (tg - tl) / (double)(T.m > 1 ? T.m : 1);
    for (int k = 0; k < T.m; ++k) {
      const double x = cld(A.line_x, k);
      if (x > 0.0) {
        const double th = ROW(R.VA + cld(T.lfrom, k));
        if (apply) ROW(R.VA + cld(T.lto, k)) = th - pfl * x / 100.0;
      }
    }
  }
  const double f = fabs(pfl);                                             // :385-386
  for (int k = 0; k < T.m; ++k) {
    const double rating = cld(T.lrating, k);
    const double ld = rating > 0.0 ? f / rating : 0.0;
    if (apply) { ROW2(R.FLOW + k) = make_double2(f, ld); ROW(R.LOAD + k) = ld; }
  }
  if (apply) {
    ROW(R.LOSSES) = tl > 0.0 ? tl * 0.03 : 0.0;                           // :388
    ROW(R.CONV) = 1.0; ROW(R.ITERS) = 1.0; ROW(R.MAXMIS) = 0.0; ROW(R.STATUS) = (double)GS_STATUS_FALLBACK_LINEAR;
  }
}
