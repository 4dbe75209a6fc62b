// kernels_env.hip -- GridEnvironment.reset() for 64 instances per wavefront (lane = instance).
// The step itself is fused into the solver kernels (kernels_solve.hip: gs_k_step_*).
//
// Reference arithmetic restated: reset, environments/grid_env.py:360-408 (relative to
// /root/reference/grid_fed_rl/).
#include <hip/hip_runtime.h>
#include <math.h>

#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]
#include "env_device.h"

extern "C" __global__ void __launch_bounds__(64)
gs_k_env_reset(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B,
               const uint64_t* __restrict__ seeds, const uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  if (b >= B) return;
  if (mask && !mask[b]) return;
  const uint64_t seed = seeds ? seeds[b] : 0ull;
  ROW(R.SEEDLO) = (double)(uint32_t)seed;
  ROW(R.SEEDHI) = (double)(uint32_t)(seed >> 32);
  ROW(R.TIME) = 0.0; ROW(R.STEP) = 0.0; ROW(R.VIOL) = 0.0; ROW(R.TOTLOSS) = 0.0; ROW(R.EPREW) = 0.0;
  ROW(R.FREQ) = 60.0;                                           // grid_env.py:394
  ROW(R.IRR) = 0.0; ROW(R.WIND) = 5.0; ROW(R.TEMP) = 25.0; ROW(R.CLOUD) = 0.3;   // grid_env.py:213-218
  for (int i = 0; i < T.n; ++i) {
    ROW(R.VM + i) = 1.0; ROW(R.VA + i) = 0.0;
    ROW(R.E + i) = cld(T.fixed_v, i) ? cld(T.v_set, i) : 1.0; ROW(R.F + i) = 0.0;     // the flat start, for a warm-started sweep solver
  }
  for (int k = 0; k < T.m; ++k) { ROW(R.FLOW + k) = 0.0; ROW(R.ENVLOAD + k) = 0.0; ROW(R.LOAD + k) = 0.0; }
  for (int q = 0; q < T.n_bats; ++q) { ROW(R.SOC + q) = 0.5; ROW(R.BATP + q) = 0.0; }      // grid_env.py:397-399
  for (int g = 0; g < T.n_gens; ++g) ROW(R.CURT + g) = 1.0;
  weather_update(R, E, S, (uint64_t)(E.first_instance + b));      // grid_env.py:402
  for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power(T, R, S, g);
  ROW(R.REWARD) = 0.0; ROW(R.TERM) = 0.0; ROW(R.TRUNC) = 0.0; ROW(R.VMAX) = 1.0; ROW(R.VMIN) = 1.0;
  for (int v = 0; v < 4; ++v) ROW(R.VFLAGS + v) = 0.0;
  ROW(R.LOSSES) = 0.0; ROW(R.MAXMIS) = 0.0; ROW(R.ITERS) = 0.0; ROW(R.CONV) = 0.0; ROW(R.STATUS) = 0.0;
}

// (e, f) from (|V|, angle) for every bus of the masked-in instances: after gs_set_state, so that a warm-started
// sweep solver resumes from the checkpointed voltages.
extern "C" __global__ void __launch_bounds__(64)
gs_k_polar_to_rect(GsTables T, GsRows R, double* __restrict__ slab, int B) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  if (b >= B) return;
  for (int i = 0; i < T.n; ++i) {
    const double vm = ROW(R.VM + i), va = ROW(R.VA + i);
    double s, c;
    sincos(va, &s, &c);
    ROW(R.E + i) = vm * c; ROW(R.F + i) = vm * s;
  }
}
