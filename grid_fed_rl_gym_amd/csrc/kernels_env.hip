// kernels_env.hip -- the non-solver part of GridEnvironment.step()/reset() for 64 instances per
// wavefront (lane = instance).  One wavefront per group: this work is O(n + m + devices) scalar
// arithmetic per instance, a few microseconds per batched step next to the load flow.
//
// Reference arithmetic restated (paths relative to /root/reference/grid_fed_rl/environments/):
//   reset                 grid_env.py:360-408
//   _apply_actions        grid_env.py:621-651, dynamics.py:189-220, 304-324
//   _update_weather       grid_env.py:653-681         (Philox stream instead of python `random`)
//   renewable models      dynamics.py:120-142, 158-170
//   load model            dynamics.py:54-75           (Philox stream instead of np.random)
//   injections            grid_env.py:683-720, power_flow.py:112-121
//   grid state, dynamics  grid_env.py:722-751, base.py:261-264, dynamics.py:260-273
//   reward                grid_env.py:785-826
//   constraints / flags   base.py:140-167, grid_env.py:563-608
#include <hip/hip_runtime.h>
#include <math.h>

#include "gs_internal.h"

#define ROW(r) S[(size_t)(r) * GS_LANES]

template <typename X>
__device__ __forceinline__ X cld(const X* p, int i) {
  return ((const GS_CONST X*)p)[i];
}

// ---- Philox4x32-10; same stream as oracle/oracle_np.py::philox4x32 ---------------------------
struct U4 { uint32_t a, b, c, d; };

__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  U4 o; o.a = c0; o.b = c1; o.c = c2; o.d = c3; return o;
}

__device__ __forceinline__ void rng_uniform_pair(uint64_t seed, uint64_t instance, uint32_t step, uint32_t draw,
                                                 double* u0, double* u1) {
  const U4 r = philox((uint32_t)instance, step, draw, 0x47535450u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t x0 = ((uint64_t)r.a << 32) | r.b, x1 = ((uint64_t)r.c << 32) | r.d;
  *u0 = (double)(x0 >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0);
  *u1 = (double)(x1 >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0);
}

__device__ __forceinline__ double rng_normal(uint64_t seed, uint64_t instance, uint32_t step, uint32_t draw) {
  double u0, u1;
  rng_uniform_pair(seed, instance, step, draw, &u0, &u1);
  return sqrt(-2.0 * log(u0)) * cos(2.0 * M_PI * u1);
}

enum { DRAW_IRRADIANCE = 0, DRAW_WIND = 1, DRAW_TEMP = 2, DRAW_CLOUD = 3, DRAW_LOAD0 = 16 };

__device__ const double kDailyProfile[24] = {0.5, 0.4, 0.4, 0.4, 0.4, 0.5, 0.7, 0.9, 0.8, 0.7, 0.6, 0.6,
                                             0.7, 0.7, 0.6, 0.6, 0.7, 0.9, 1.0, 0.9, 0.8, 0.7, 0.6, 0.5};

__device__ __forceinline__ uint64_t lane_seed(double* S, const GsRows& R) {
  return ((uint64_t)(uint32_t)ROW(R.SEEDHI) << 32) | (uint64_t)(uint32_t)ROW(R.SEEDLO);
}

// grid_env.py:653-681
__device__ __forceinline__ void weather_update(const GsRows& R, const GsEnvCfg& E, double* S, uint64_t inst) {
  if (!E.weather_variation) return;
  const uint64_t seed = lane_seed(S, R);
  const uint32_t step = (uint32_t)ROW(R.STEP);
  const double hour = fmod(ROW(R.TIME) / 3600.0, 24.0);
  const double base = (hour >= 6.0 && hour <= 18.0) ? 1000.0 * sin(M_PI * (hour - 6.0) / 12.0) : 0.0;
  double u, u_unused;
  rng_uniform_pair(seed, inst, step, DRAW_IRRADIANCE, &u, &u_unused);
  ROW(R.IRR) = base * (0.8 + 0.4 * u);
  ROW(R.WIND) = fmax(0.0, fmin(30.0, ROW(R.WIND) + 0.5 * rng_normal(seed, inst, step, DRAW_WIND)));
  ROW(R.TEMP) = 25.0 + 10.0 * sin(2.0 * M_PI * (hour - 12.0) / 24.0) + 2.0 * rng_normal(seed, inst, step, DRAW_TEMP);
  ROW(R.CLOUD) = fmax(0.0, fmin(1.0, ROW(R.CLOUD) + 0.1 * rng_normal(seed, inst, step, DRAW_CLOUD)));
}

// dynamics.py:120-142 / 158-170
__device__ __forceinline__ double renewable_power(const GsTables& T, const GsRows& R, double* S, int g) {
  const double cap = cld(T.gen_cap, g), p0 = cld(T.gen_p0, g), p1 = cld(T.gen_p1, g), p2 = cld(T.gen_p2, g);
  if (cld(T.gen_kind, g) == 0) {
    const double hour = fmod(ROW(R.TIME) / 3600.0, 24.0);
    const double elev = (hour >= 6.0 && hour <= 18.0) ? sin(M_PI * (hour - 6.0) / 12.0) : 0.0;
    const double irr = 1000.0 * elev * (1.0 - 0.8 * ROW(R.CLOUD));
    const double tf = 1.0 - 0.004 * fmax(0.0, ROW(R.TEMP) - 25.0);
    return fmin(irr * p1 * p0 * tf, cap);
  }
  const double w = ROW(R.WIND);
  if (w < p0 || w > p2) return 0.0;
  if (w <= p1) { const double q = (w - p0) / (p1 - p0); return cap * (q * q * q); }
  return cap;
}

// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64)
gs_k_env_reset(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B,
               const uint64_t* __restrict__ seeds, const uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  if (b >= B) return;
  if (mask && !mask[b]) return;
  const uint64_t seed = seeds ? seeds[b] : 0ull;
  ROW(R.SEEDLO) = (double)(uint32_t)seed;
  ROW(R.SEEDHI) = (double)(uint32_t)(seed >> 32);
  ROW(R.TIME) = 0.0; ROW(R.STEP) = 0.0; ROW(R.VIOL) = 0.0; ROW(R.TOTLOSS) = 0.0; ROW(R.EPREW) = 0.0;
  ROW(R.FREQ) = 60.0;                                           // grid_env.py:394
  ROW(R.IRR) = 0.0; ROW(R.WIND) = 5.0; ROW(R.TEMP) = 25.0; ROW(R.CLOUD) = 0.3;   // grid_env.py:213-218
  for (int i = 0; i < T.n; ++i) { ROW(R.VM + i) = 1.0; ROW(R.VA + i) = 0.0; }
  for (int k = 0; k < T.m; ++k) { ROW(R.FLOW + k) = 0.0; ROW(R.ENVLOAD + k) = 0.0; ROW(R.LOAD + k) = 0.0; }
  for (int q = 0; q < T.n_bats; ++q) { ROW(R.SOC + q) = 0.5; ROW(R.BATP + q) = 0.0; }      // grid_env.py:397-399
  for (int g = 0; g < T.n_gens; ++g) ROW(R.CURT + g) = 1.0;
  weather_update(R, E, S, (uint64_t)(E.first_instance + b));      // grid_env.py:402
  for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power(T, R, S, g);
  ROW(R.REWARD) = 0.0; ROW(R.TERM) = 0.0; ROW(R.TRUNC) = 0.0; ROW(R.VMAX) = 1.0; ROW(R.VMIN) = 1.0;
  for (int v = 0; v < 4; ++v) ROW(R.VFLAGS + v) = 0.0;
  ROW(R.LOSSES) = 0.0; ROW(R.MAXMIS) = 0.0; ROW(R.ITERS) = 0.0; ROW(R.CONV) = 0.0; ROW(R.STATUS) = 0.0;
}

// Everything of step() that precedes the load flow: actions, clock, weather, injections.
extern "C" __global__ void __launch_bounds__(64)
gs_k_env_pre(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  if (b >= B) return;
  const double dt = E.timestep;
  // _apply_actions: batteries first, then curtailment factors
  for (int q = 0; q < T.n_bats; ++q) {
    const double rating = cld(T.bat_rating, q), cap = cld(T.bat_cap, q), eff = cld(T.bat_eff, q);
    const double cmd = ROW(R.ACT + q) * rating;
    double soc = ROW(R.SOC + q);
    if (cmd > 0.0) {                                  // discharge, dynamics.py:206-220
      const double p = fmin(cmd, rating);
      const double e = fmin(p * dt / 3600.0, soc * cap * eff);
      soc -= e / (cap * eff);
      ROW(R.SOC + q) = soc;
      ROW(R.BATP + q) = e * 3600.0 / dt;
    } else if (cmd < 0.0) {                           // charge, dynamics.py:189-204
      const double p = fmin(-cmd, rating);
      const double max_e = (1.0 - soc) * cap;
      const double e = fmin(p * dt / 3600.0, max_e / eff);
      soc += e * eff / cap;
      ROW(R.SOC + q) = soc;
      ROW(R.BATP + q) = -(e * 3600.0 / dt);
    }
  }
  for (int g = 0; g < T.n_gens; ++g) ROW(R.CURT + g) = (ROW(R.ACT + T.n_bats + g) + 1.0) / 2.0;
  ROW(R.TIME) = ROW(R.TIME) + dt;                    // grid_env.py:470-471
  ROW(R.STEP) = ROW(R.STEP) + 1.0;
  const uint64_t inst = (uint64_t)(E.first_instance + b);
  weather_update(R, E, S, inst);
  for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power(T, R, S, g);
  // realised load power per load (dynamics.py:54-75 when stochastic, base_power otherwise)
  if (E.stochastic_loads) {
    const uint64_t seed = lane_seed(S, R);
    const uint32_t step = (uint32_t)ROW(R.STEP);
    const double hour = fmod(ROW(R.TIME) / 3600.0, 24.0);
    const int hi = (int)hour;
    const double frac = hour - (double)hi;
    const double prof = kDailyProfile[hi] * (1.0 - frac) + kDailyProfile[(hi + 1) % 24] * frac;
    for (int l = 0; l < T.n_loads; ++l) {
      const double z = rng_normal(seed, inst, step, DRAW_LOAD0 + l);
      const double mult = prof * (1.0 + 0.1 * z);
      ROW(R.LOADP + l) = fmax(0.0, cld(T.load_base, l) * mult * 1.0);
    }
  }
  // per-bus injection, in the reference's accumulation order (grid_env.py:689-718), then
  // P_spec = (0 - loads) + generation (power_flow.py:112-121)
  const double inv_base = E.power_base;
  for (int i = 0; i < T.n; ++i) {
    double ls = 0.0, gs = 0.0;
    for (int p = cld(T.bl_ptr, i); p < cld(T.bl_ptr, i + 1); ++p) {
      const int l = cld(T.bl_idx, p);
      ls += E.stochastic_loads ? ROW(R.LOADP + l) : cld(T.load_base, l);
    }
    for (int p = cld(T.bg_ptr, i); p < cld(T.bg_ptr, i + 1); ++p) {
      const int g = cld(T.bg_idx, p);
      gs += ROW(R.GENP + g) * ROW(R.CURT + g);
    }
    for (int p = cld(T.bb_ptr, i); p < cld(T.bb_ptr, i + 1); ++p) {
      const double bp = ROW(R.BATP + cld(T.bb_idx, p));
      if (bp > 0.0) gs += bp; else if (bp < 0.0) ls += fabs(bp);
    }
    ROW(R.P + i) = (0.0 - ls / inv_base) + gs / inv_base;
    ROW(R.Q + i) = 0.0;                                // the reference never injects Q (power_flow.py:107)
  }
}

// Everything of step() that follows the load flow.
extern "C" __global__ void __launch_bounds__(64)
gs_k_env_post(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B, double total_load) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x * GS_LANES + lane;
  double* S = slab + (size_t)blockIdx.x * R.total * GS_LANES + lane;
  if (b >= B) return;
  const double dt = E.timestep;
  const double losses = ROW(R.LOSSES);
  // _update_grid_state (grid_env.py:722-739); Line.update_state overwrites loading with |P|/rating
  int overloaded = 0;
  for (int k = 0; k < T.m; ++k) {
    const double rating = cld(T.lrating, k);
    const double ld = (rating > 0.0) ? fabs(ROW(R.FLOW + k)) / rating : 0.0;
    ROW(R.ENVLOAD + k) = ld;
    overloaded += (ld > 0.8) ? 1 : 0;
  }
  const double totloss = ROW(R.TOTLOSS) + losses * dt / 3600.0;
  ROW(R.TOTLOSS) = totloss;
  // _update_dynamics (grid_env.py:741-751) + swing equation (dynamics.py:260-273)
  double total_gen = 0.0, total_curt = 0.0;
  for (int g = 0; g < T.n_gens; ++g) {
    const double p = ROW(R.GENP + g);
    total_gen += p;
    total_curt += p * (1.0 - ROW(R.CURT + g));
  }
  const double imbalance = (total_gen - total_load - losses * E.power_base) / 1e6;
  double f = ROW(R.FREQ);
  f += ((imbalance - E.D * (f - E.f0)) / (2.0 * E.H * E.f0)) * dt;
  f = fmax(55.0, fmin(65.0, f));
  ROW(R.FREQ) = f;
  // reward (grid_env.py:785-826) and constraint flags (base.py:153-167) in one pass over the buses
  double dev = 0.0, vmax = -INFINITY, vmin = INFINITY;
  int vhigh = 0, vlow = 0;
  for (int i = 0; i < T.n; ++i) {
    const double v = ROW(R.VM + i);
    dev += fabs(v - 1.0);
    vmax = fmax(vmax, v); vmin = fmin(vmin, v);
    vhigh |= (v > E.v_max) ? 1 : 0;
    vlow |= (v < E.v_min) ? 1 : 0;
  }
  double reward = 0.0;
  reward -= dev * 10.0;
  reward -= fabs(f - 60.0) * 20.0;
  reward -= (double)(overloaded * 50);
  reward -= totloss * 0.1;
  reward += (total_gen - total_curt) * 1e-5;
  for (int q = 0; q < T.n_bats; ++q) {
    const double soc = ROW(R.SOC + q);
    reward += (soc >= 0.2 && soc <= 0.8) ? 1.0 : -5.0;
  }
  const int fhigh = f > E.f_max, flow_ = f < E.f_min;
  double viol = ROW(R.VIOL);
  double trunc = 0.0;
  if (vhigh | vlow | fhigh | flow_) {
    viol += 1.0;
    if (viol > 10.0) { trunc = 1.0; reward -= E.safety_penalty; }     // grid_env.py:604-606
  }
  ROW(R.VIOL) = viol;
  ROW(R.TRUNC) = trunc;
  ROW(R.TERM) = (ROW(R.STEP) >= (double)E.episode_length) ? 1.0 : 0.0;   // base.py:140-142
  ROW(R.REWARD) = reward;
  ROW(R.EPREW) = ROW(R.EPREW) + reward;
  ROW(R.VMAX) = vmax; ROW(R.VMIN) = vmin;
  ROW(R.VFLAGS + 0) = (double)vhigh; ROW(R.VFLAGS + 1) = (double)vlow;
  ROW(R.VFLAGS + 2) = (double)fhigh; ROW(R.VFLAGS + 3) = (double)flow_;
}
