// kernels_env.hip -- GridEnvironment.reset() for 64 instances per wavefront (lane = instance), and the small
// kernels of the device-resident rollout collector (gs_rollout): action sampling, per-step bookkeeping with the
// in-place reset of finished instances.  The step itself is fused into the solver kernels
// (kernels_solve.hip: gs_k_step_*).
//
// Reference arithmetic restated (relative to /root/reference/grid_fed_rl/): reset, environments/grid_env.py:360-408;
// rollout loop, algorithms/base.py:268-298.
#include <hip/hip_runtime.h>
#include <math.h>

#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]
#include "env_device.h"

// seeds == NULL: the instance's stream runs on (next_episode_seed of the seed it holds; a handle that was never
// seeded holds 0), as the reference's reset(seed=None) leaves its global streams running
extern "C" __global__ void __launch_bounds__(64)
gs_k_env_reset(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B,
               const uint64_t* __restrict__ seeds, const uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  if (b >= B) return;
  if (mask && !mask[b]) return;
  const uint64_t inst = (uint64_t)(E.first_instance + b);
  const uint64_t seed = seeds ? seeds[b] : next_episode_seed(lane_seed(S, R), inst);
  env_reset_lane(T, R, E, S, inst, seed);
}

// Rows that follow from a checkpoint but are not part of it (gs_set_state): (e, f) from (|V|, angle), so that a
// warm-started sweep solver resumes from the checkpointed voltages, and the uncurtailed renewable powers of the
// observation (grid_env.py:773-777) from the checkpointed clock and weather.
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
  for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power(T, R, S, g);
}

// ---- rollout collector ------------------------------------------------------------------------------------------
// Uniform random actions in (-1, 1) for T steps (the reference samples env.action_space, algorithms/base.py:280, from
// python's global `random`; here: Philox keyed by the caller's policy seed, counter = (global instance, step of the
// rollout, action quad, 'ACTN'), every 32-bit word one action 2 (r + 1/2) 2^-32 - 1).  One thread per (t, b, quad).
extern "C" __global__ void __launch_bounds__(256)
gs_k_rollout_actions(double* __restrict__ act, int T, int B, int A, uint64_t seed, int64_t first_instance, uint32_t t0) {
  const int Aq = (A + 3) >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)T * B * Aq) return;
  const int q = (int)(idx % Aq);
  const long long tb = idx / Aq;
  const int b = (int)(tb % B), t = (int)(tb / B);
  const U4 r = philox((uint32_t)(first_instance + b), t0 + (uint32_t)t, (uint32_t)q, 0x4143544Eu, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t w[4] = {r.a, r.b, r.c, r.d};
  double* o = act + ((size_t)t * B + b) * A + 4 * q;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (4 * q + k < A) o[k] = 2.0 * (((double)w[k] + 0.5) * (1.0 / 4294967296.0)) - 1.0;
}

// The constant columns [skip0, skip1) of an observation (static load powers, grid_env.py:769-770) for `slots`
// consecutive [B][obs_dim] blocks: written once when the rollout buffers are allocated -- the step kernel only ever
// writes the columns that change.
extern "C" __global__ void __launch_bounds__(256)
gs_k_fill_const_columns(double* __restrict__ out, long long rows, int obs_dim, int skip0, int skip1,
                        const int32_t* __restrict__ map, const double* __restrict__ cst) {
  const int w = skip1 - skip0;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w <= 0 || idx >= rows * w) return;
  const int c = skip0 + (int)(idx % w);
  out[(idx / w) * obs_dim + c] = cst[-map[c] - 1];
}

// After step t of a rollout (lane = instance): reward and done flags of the step into the [T][B] arrays; an instance
// that finished (terminated or truncated) has its terminal observation -- row b of obs_next, which the step kernel has
// just written -- moved to the side list, is reset in place with the next seed of its chain (algorithms/base.py:289-290:
// `obs, _ = env.reset()`), and its fresh observation takes the row's place, so that obs_next is what step t + 1 starts
// from for every instance.  With random actions an episode lasts 10-20 steps (truncation), so the row moves are done by
// the whole workgroup, coalesced (per-lane column loops cost 1.5 ms per 8192 finished instances).
// (arguments: GsRolloutPostArgs, gs_internal.h)
extern "C" __global__ void __launch_bounds__(256)
gs_k_rollout_post(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, GsRolloutPostArgs A) {
  // 256 threads per 64-instance group: threads 0-63 are the instances (flags, list entries, reset), all of them move the
  // observation rows, consecutive threads on consecutive columns
  __shared__ int fin[GS_LANES];          // entry of the side list (>= 0), -1 list full, -2 not finished
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * GS_LANES + lane;
  const GsLaneRows S = gs_lane_rows(slab, blockIdx.x, R.total, lane);
  const bool mine = threadIdx.x < GS_LANES && b < A.B;
  if (threadIdx.x < GS_LANES) {
    int k = -2;
    if (mine) {
      const double rew = ROW(R.REWARD), te = ROW(R.TERM), tr = ROW(R.TRUNC);
      const int d = (te != 0.0 ? 1 : 0) | (tr != 0.0 ? 2 : 0);
      if (A.rew) {                          // (NULL: the step kernel has written them itself)
        A.rew[(size_t)A.t * A.B + b] = rew;
        A.done[(size_t)A.t * A.B + b] = (uint8_t)d;
      }
      if (d) {
        k = atomicAdd(A.term_count, 1);
        if (k < A.term_cap) { A.term_idx[2 * k] = A.t; A.term_idx[2 * k + 1] = b; }
        else k = -1;
      }
    }
    fin[lane] = k;
  }
  __syncthreads();
  bool any = false;
  for (int q = 0; q < GS_LANES; ++q) {
    const int k = fin[q];
    any |= k != -2;
    if (k >= 0) {
      const double* row = A.obs_next + (size_t)(blockIdx.x * GS_LANES + q) * A.obs_dim;
      double* dst = A.term_obs + (size_t)k * A.obs_dim;
      for (int c = threadIdx.x; c < A.obs_dim; c += blockDim.x) dst[c] = row[c];
    }
  }
  if (!any) return;                       // (uniform: fin is shared)
  if (mine && fin[lane] != -2) {
    const uint64_t inst = (uint64_t)(E.first_instance + b);
    env_reset_lane_scalars(T, R, E, S, inst, next_episode_seed(lane_seed(S, R), inst));
  }
  for (int q = 0; q < GS_LANES; ++q) {
    if (fin[q] == -2) continue;
    const GsLaneRows Sq = gs_lane_rows(slab, blockIdx.x, R.total, q);
    for (int j = threadIdx.x; j < T.n + T.m; j += blockDim.x) env_reset_element(T, R, Sq, j);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the reset rows are in memory before the other waves gather them
  for (int q = 0; q < GS_LANES; ++q) {
    if (fin[q] == -2) continue;
    const GsLaneRows Sq = gs_lane_rows(slab, blockIdx.x, R.total, q);
    double* row = A.obs_next + (size_t)(blockIdx.x * GS_LANES + q) * A.obs_dim;
    for (int c = threadIdx.x; c < A.obs_dim; c += blockDim.x) {
      const int s = A.map[c];
      row[c] = (s >= 0) ? Sq.lane_row((size_t)s * GS_LANES).get() : A.cst[-s - 1];
    }
  }
}
