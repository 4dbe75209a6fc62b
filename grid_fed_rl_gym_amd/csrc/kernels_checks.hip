// kernels_checks.hip -- post-step checks on the state a step / solve left in the slab:
//   SafetyChecker.check_constraints + get_violation_severity   utils/safety.py:114-203
//   SafetyMonitor.check_constraints                            utils/safety.py:313-394
//   _assess_solution_quality                                   robust_power_flow.py:615-657
// One workgroup per 64-instance group (lane = instance), four wavefronts split the buses and the lines; the
// rows are read once for all three checks.  Stateful like the reference classes: previous voltages /
// frequency (rate of change), consecutive-violation counter and the sticky emergency mode live in buffers
// owned by the gs_checks object.  thermal_data is not modelled (no temperatures in the environment), so the
// 'critical' severity cannot occur.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/gridstep.h"
#include "gs_internal.h"
#include "kernels.h"

#define CK_WAVES 16

extern "C" __global__ void __launch_bounds__(64 * CK_WAVES)
gs_k_checks(GsChecksCfg C, const double* __restrict__ slab, const double* __restrict__ freq_override, double* __restrict__ prev,
            int32_t* __restrict__ state, int32_t* __restrict__ out_i, double* __restrict__ out_f, uint8_t* __restrict__ bus_mask,
            uint8_t* __restrict__ line_mask, int B, int Bp) {
  __shared__ int pi[8][CK_WAVES][GS_LANES];
  __shared__ double pd[4][CK_WAVES][GS_LANES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.x;
  const int b = g * GS_LANES + lane;
  const double* S = slab + (size_t)g * C.rows_total * GS_LANES;
  double* P = prev + (size_t)g * (C.n + 1) * GS_LANES + lane;
#define ROWS(r) S[GS_ELEM((r), lane)]
  int c_nlow = 0, c_nhigh = 0, m_nhigh = 0, m_nlow = 0, m_nem = 0, c_nover = 0, m_nover = 0, flags = 0;   // flags: 1 rate NaN, 2 v non-finite, 4 flow non-finite, 8 quality loading NaN
  double dvmax = 0.0, vmin = INFINITY, vmax = -INFINITY, qlmax = -INFINITY;
  // four rows per trip, all loads first: a group's rows come from HBM / Infinity Cache and a wave has only
  // n / 16 of them, so the loads of a trip must overlap
  for (int i0 = wave; i0 < C.n; i0 += 4 * CK_WAVES) {
    double vv[4], pp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * CK_WAVES, C.n - 1);
      vv[u] = ROWS(C.row_vm + 2 * i); pp[u] = P[(size_t)i * GS_LANES];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * CK_WAVES;
      if (i >= C.n) break;
      const double v = vv[u];
      const bool cl = v < C.c_vlo, ch = !cl && v > C.c_vhi;                  // safety.py:129-137 (elif)
      const bool mh = v > C.m_vhi, ml = v < C.m_vlo;                         // :333-337
      const bool em = v > C.m_evhi || v < C.m_evlo;                          // :340-341
      c_nlow += cl; c_nhigh += ch; m_nhigh += mh; m_nlow += ml; m_nem += em;
      const double d = fabs(v - pp[u]);                                      // :168
      if (d != d) flags |= 1;
      dvmax = fmax(dvmax, d);
      P[(size_t)i * GS_LANES] = v;                                           // :181-184
      if (!(fabs(v) < INFINITY)) flags |= 2;                                 // robust_power_flow.py:643-647
      vmin = fmin(vmin, v); vmax = fmax(vmax, v);
      if (bus_mask) bus_mask[((size_t)g * C.n + i) * GS_LANES + lane] = (uint8_t)(cl | (ch << 1) | (ml << 2) | (mh << 3) | (em << 4));
    }
  }
  for (int k0 = wave; k0 < C.m; k0 += 4 * CK_WAVES) {
    double l1[4], l2[4], l3[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u * CK_WAVES, C.m - 1);
      l1[u] = ROWS(C.row_cload + C.stride_cload * k); l2[u] = ROWS(C.row_qload + k); l3[u] = ROWS(C.row_flow + 2 * k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * CK_WAVES;
      if (k >= C.m) break;
      const double ld = l1[u], ql = l2[u], fl = l3[u];
      const bool co = ld > C.c_load, mo = ld > C.m_load;                     // safety.py:150-154, :364-365
      c_nover += co; m_nover += mo;
      if (ql != ql) flags |= 8;
      qlmax = fmax(qlmax, ql);
      if (!(fabs(fl) < INFINITY)) flags |= 4;
      if (line_mask) line_mask[((size_t)g * C.m + k) * GS_LANES + lane] = (uint8_t)(co | (mo << 1));
    }
  }
  pi[0][wave][lane] = c_nlow; pi[1][wave][lane] = c_nhigh; pi[2][wave][lane] = m_nhigh; pi[3][wave][lane] = m_nlow;
  pi[4][wave][lane] = m_nem; pi[5][wave][lane] = c_nover; pi[6][wave][lane] = m_nover; pi[7][wave][lane] = flags;
  pd[0][wave][lane] = dvmax; pd[1][wave][lane] = vmin; pd[2][wave][lane] = vmax; pd[3][wave][lane] = qlmax;
  __syncthreads();
  if (wave != 0 || b >= B) return;
  for (int w = 1; w < CK_WAVES; ++w) {
    c_nlow += pi[0][w][lane]; c_nhigh += pi[1][w][lane]; m_nhigh += pi[2][w][lane]; m_nlow += pi[3][w][lane];
    m_nem += pi[4][w][lane]; c_nover += pi[5][w][lane]; m_nover += pi[6][w][lane]; flags |= pi[7][w][lane];
    dvmax = fmax(dvmax, pd[0][w][lane]); vmin = fmin(vmin, pd[1][w][lane]); vmax = fmax(vmax, pd[2][w][lane]);
    qlmax = fmax(qlmax, pd[3][w][lane]);
  }
  const double f = freq_override ? freq_override[b] : ROWS(C.row_freq);
  int32_t* has_prev = state + b; int32_t* consec = state + Bp + b; int32_t* emode = state + 2 * Bp + b;
#define OI(k) out_i[(size_t)(k) * Bp + b]
#define OF(k) out_f[(size_t)(k) * Bp + b]
  // ---- SafetyChecker
  const int c_flow = f < C.c_flo, c_fhigh = !c_flow && f > C.c_fhi;        // :140-147
  const double vrate = (flags & 1) ? NAN : dvmax / C.dt;                   // :168 (np.max propagates NaN)
  const double frate = fabs(f - P[(size_t)C.n * GS_LANES]) / C.dt;         // :174
  const int hp = *has_prev;
  const int c_vr = hp && vrate > C.c_rocv, c_fr = hp && frate > C.c_rocf;  // :166-178
  P[(size_t)C.n * GS_LANES] = f; *has_prev = 1;
  const int c_total = c_nlow + c_nhigh + c_flow + c_fhigh + c_nover + c_vr + c_fr;
  OI(GS_CI_C_NLOW) = c_nlow; OI(GS_CI_C_NHIGH) = c_nhigh; OI(GS_CI_C_FLOW) = c_flow; OI(GS_CI_C_FHIGH) = c_fhigh;
  OI(GS_CI_C_NOVER) = c_nover; OI(GS_CI_C_VRATE) = c_vr; OI(GS_CI_C_FRATE) = c_fr; OI(GS_CI_C_TOTAL) = c_total;
  OI(GS_CI_C_SEVERITY) = c_total > 5 ? 3 : (c_total > 2 ? 2 : (c_total > 0 ? 1 : 0));     // :188-203 without thermal data
  OF(GS_CF_VRATE) = vrate; OF(GS_CF_FRATE) = frate;
  // ---- SafetyMonitor
  const int m_fhigh = f > C.m_fhi, m_flow = !m_fhigh && f < C.m_flo;       // :352-355
  const int m_fem = f > C.m_efhi || f < C.m_eflo;                          // :358-361
  const int m_total = m_nhigh + m_nlow + m_nem + m_fhigh + m_flow + m_fem + m_nover;     // :368-376
  const int cs = m_total > 0 ? *consec + 1 : 0;                            // :379-383
  const int trigger = (m_nem > 0) || m_fem || cs > 5 || m_total > 10;      // :386-390
  const int mode = *emode | trigger;
  *consec = cs; *emode = mode;
  OI(GS_CI_M_NHIGH) = m_nhigh; OI(GS_CI_M_NLOW) = m_nlow; OI(GS_CI_M_NEMERG) = m_nem; OI(GS_CI_M_FHIGH) = m_fhigh; OI(GS_CI_M_FLOW) = m_flow;
  OI(GS_CI_M_FEMERG) = m_fem; OI(GS_CI_M_NOVER) = m_nover; OI(GS_CI_M_TOTAL) = m_total; OI(GS_CI_M_ACTION) = trigger;
  OI(GS_CI_M_CONSEC) = cs; OI(GS_CI_M_EMODE) = mode;
  // ---- solution quality (robust_power_flow.py:615-657)
  double q = 1.0;
  if (C.n > 0) {
    if (vmin < 0.8 || vmax > 1.2) q *= 0.3;
    else if (vmin < 0.9 || vmax > 1.1) q *= 0.7;
  }
  if (C.m > 0 && !(flags & 8)) {
    if (qlmax > 2.0) q *= 0.2;
    else if (qlmax > 1.0) q *= 0.5;
  }
  if (ROWS(C.row_maxmis) > C.q_tol * 100.0) q *= 0.6;
  const double its = ROWS(C.row_iters);
  if (its <= 5.0) q *= 1.1;
  else if (its > 20.0) q *= 0.9;
  q = fmin(q, 1.0);
  if (ROWS(C.row_conv) == 0.0 || (flags & 6)) q = 0.0;
  OF(GS_CF_QUALITY) = q;
#undef OI
#undef OF
#undef ROWS
}

// forget the stateful parts of the instances whose mask byte is non-zero (all when mask == nullptr)
extern "C" __global__ void gs_k_checks_reset(int32_t* __restrict__ state, const uint8_t* __restrict__ mask, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || (mask && !mask[b])) return;
  state[b] = 0; state[Bp + b] = 0; state[2 * Bp + b] = 0;
}
