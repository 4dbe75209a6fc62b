// env_device.h -- device-side pieces of GridEnvironment.step()/reset() shared by the reset kernel
// and the fused step kernels.  lane = instance; every function works on the caller's lane.
//
// Reference arithmetic restated (paths relative to /root/reference/grid_fed_rl/environments/):
//   _apply_actions        grid_env.py:621-651, dynamics.py:189-220, 304-324
//   _update_weather       grid_env.py:653-681         (Philox stream instead of python `random`)
//   renewable models      dynamics.py:120-142, 158-170
//   load model            dynamics.py:54-75           (Philox stream instead of np.random)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "gs_internal.h"
#include "fastmath.h"

#ifndef ROW
#define ROW(r) S[(size_t)(r) * GS_LANES]
#endif

template <typename X>
__device__ __forceinline__ X cld(const X* p, int i) {
  return ((const GS_CONST X*)p)[i];
}

// fmod(time / 3600, 24), exactly as libm's division and fmod give it (fastmath.h)
__device__ __forceinline__ double hour_of_day(double time_s) {
  return gs_fmod_pos(gs_div_by(time_s, 3600.0, 1.0 / 3600.0), 24.0, 1.0 / 24.0);
}

// ---- Philox4x32-10 (Salmon et al. 2011), key = seed, counter = (instance, step, draw, tag) ------
struct U4 { uint32_t a, b, c, d; };

__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one: integer multiplies are
    // quarter rate, and the forty of a Philox call were most of the load-noise phase
    const uint64_t m0 = (uint64_t)0xD2511F53u * (uint64_t)c0, m1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
    const uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
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

// Four standard normals from ONE Philox call: every 32-bit output word is a uniform (r + 1/2) 2^-32, words (0, 1) and
// (2, 3) are one Box-Muller pair each (cosine, sine).  The load noise takes them four at a time (load l: draw
// DRAW_LOAD0 + l / 4, component l & 3), the weather takes wind / temperature / cloud from one call: the integer
// multiplies of Philox were half the vector work of the load-noise phase with one call per pair.
__device__ __forceinline__ void rng_normal_quad(uint64_t seed, uint64_t instance, uint32_t step, uint32_t draw, double (&z)[4]) {
  const U4 r = philox((uint32_t)instance, step, draw, 0x47535450u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u0 = ((double)r.a + 0.5) * (1.0 / 4294967296.0), u1 = ((double)r.b + 0.5) * (1.0 / 4294967296.0);
  const double u2 = ((double)r.c + 0.5) * (1.0 / 4294967296.0), u3 = ((double)r.d + 0.5) * (1.0 / 4294967296.0);
  const double ra = sqrt(-2.0 * gs_log01(u0)), rb = sqrt(-2.0 * gs_log01(u2));
  double sn, cs;
  gs_sincos_turns(u1, &sn, &cs); z[0] = ra * cs; z[1] = ra * sn;
  gs_sincos_turns(u3, &sn, &cs); z[2] = rb * cs; z[3] = rb * sn;
}

enum { DRAW_IRRADIANCE = 0, DRAW_WEATHER = 1, DRAW_LOAD0 = 16 };

// Seed of an instance's NEXT episode when reset() is called without one (gs_reset with seeds == NULL, and the
// automatic resets inside gs_rollout).  The reference's reset(seed=None) does not re-seed: its global `random` /
// np.random streams simply run on (grid_env.py:366-369), so consecutive episodes see different noise.  Here the
// stream is a pure function of (seed, instance, step, draw); "running on" = a new seed derived from the old one,
// one Philox call with its own tag word.  reset(seed=k) stays exactly reproducible, and the seed is part of the
// checkpoint, so a resumed run continues the same chain.
__device__ __forceinline__ uint64_t next_episode_seed(uint64_t seed, uint64_t instance) {
  const U4 r = philox((uint32_t)instance, (uint32_t)(instance >> 32), 0u, 0x52534544u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return ((uint64_t)r.b << 32) | (uint64_t)r.a;
}

__device__ const double kDailyProfile[24] = {0.5, 0.4, 0.4, 0.4, 0.4, 0.5, 0.7, 0.9, 0.8, 0.7, 0.6, 0.6,
                                             0.7, 0.7, 0.6, 0.6, 0.7, 0.9, 1.0, 0.9, 0.8, 0.7, 0.6, 0.5};

__device__ __forceinline__ uint64_t lane_seed(GsLaneRows S, const GsRows& R) {
  return ((uint64_t)(uint32_t)ROW(R.SEEDHI) << 32) | (uint64_t)(uint32_t)ROW(R.SEEDLO);
}

// grid_env.py:653-681
// time_s / step: the clock AFTER this step's advance (the rows may be updated by another wave meanwhile)
__device__ __forceinline__ void weather_update_at(const GsRows& R, const GsEnvCfg& E, GsLaneRows S, uint64_t inst, double time_s, uint32_t step) {
  if (!E.weather_variation) return;
  const uint64_t seed = lane_seed(S, R);
  const double hour = hour_of_day(time_s);
  double sn, cs;
  gs_sincos_turns((hour - 6.0) * (1.0 / 24.0), &sn, &cs);       // sin(pi (hour - 6) / 12)
  const double base = (hour >= 6.0 && hour <= 18.0) ? 1000.0 * sn : 0.0;
  double u, u_unused;
  rng_uniform_pair(seed, inst, step, DRAW_IRRADIANCE, &u, &u_unused);
  ROW(R.IRR) = base * (0.8 + 0.4 * u);
  double z[4];
  rng_normal_quad(seed, inst, step, DRAW_WEATHER, z);
  ROW(R.WIND) = fmax(0.0, fmin(30.0, ROW(R.WIND) + 0.5 * z[0]));
  gs_sincos_turns((hour - 12.0) * (1.0 / 24.0), &sn, &cs);      // sin(2 pi (hour - 12) / 24)
  ROW(R.TEMP) = 25.0 + 10.0 * sn + 2.0 * z[1];
  ROW(R.CLOUD) = fmax(0.0, fmin(1.0, ROW(R.CLOUD) + 0.1 * z[2]));
}
// The same update with the weather state handed on in registers: the three rows are requested together, updated where
// `update` holds (a padded lane keeps its rows), and what the renewables need comes back without reading them again.
struct GsWeather { double wind, temp, cloud; };
__device__ __forceinline__ GsWeather weather_step(const GsRows& R, const GsEnvCfg& E, GsLaneRows S, uint64_t inst, double time_s, uint32_t step, bool update) {
  GsWeather w;
  w.wind = ROW(R.WIND); w.temp = ROW(R.TEMP); w.cloud = ROW(R.CLOUD);
  if (!E.weather_variation) return w;
  const uint64_t seed = lane_seed(S, R);
  const double hour = hour_of_day(time_s);
  double sn, cs;
  gs_sincos_turns((hour - 6.0) * (1.0 / 24.0), &sn, &cs);       // sin(pi (hour - 6) / 12)
  const double base = (hour >= 6.0 && hour <= 18.0) ? 1000.0 * sn : 0.0;
  double u, u_unused;
  rng_uniform_pair(seed, inst, step, DRAW_IRRADIANCE, &u, &u_unused);
  const double irr = base * (0.8 + 0.4 * u);
  double z[4];
  rng_normal_quad(seed, inst, step, DRAW_WEATHER, z);
  const double wind = fmax(0.0, fmin(30.0, w.wind + 0.5 * z[0]));
  gs_sincos_turns((hour - 12.0) * (1.0 / 24.0), &sn, &cs);      // sin(2 pi (hour - 12) / 24)
  const double temp = 25.0 + 10.0 * sn + 2.0 * z[1];
  const double cloud = fmax(0.0, fmin(1.0, w.cloud + 0.1 * z[2]));
  if (update) {
    ROW(R.IRR) = irr; ROW(R.WIND) = wind; ROW(R.TEMP) = temp; ROW(R.CLOUD) = cloud;
    w.wind = wind; w.temp = temp; w.cloud = cloud;
  }
  return w;
}
__device__ __forceinline__ double renewable_power_w(const GsTables& T, int g, double elev, const GsWeather& w) {
  const double cap = cld(T.gen_cap, g), p0 = cld(T.gen_p0, g), p1 = cld(T.gen_p1, g), p2 = cld(T.gen_p2, g);
  if (cld(T.gen_kind, g) == 0) {
    const double irr = 1000.0 * elev * (1.0 - 0.8 * w.cloud);
    const double tf = 1.0 - 0.004 * fmax(0.0, w.temp - 25.0);
    return fmin(irr * p1 * p0 * tf, cap);
  }
  if (w.wind < p0 || w.wind > p2) return 0.0;
  if (w.wind <= p1) { const double q = (w.wind - p0) / (p1 - p0); return cap * (q * q * q); }
  return cap;
}
__device__ __forceinline__ void weather_update(const GsRows& R, const GsEnvCfg& E, GsLaneRows S, uint64_t inst) {
  weather_update_at(R, E, S, inst, ROW(R.TIME), (uint32_t)ROW(R.STEP));
}

// dynamics.py:120-142 / 158-170
// sin of the solar elevation proxy of the reference's solar model: sin(pi (hour - 6) / 12) between 6 h and 18 h
__device__ __forceinline__ double solar_elevation(double time_s) {
  const double hour = hour_of_day(time_s);
  double sn, cs;
  gs_sincos_turns((hour - 6.0) * (1.0 / 24.0), &sn, &cs);
  return (hour >= 6.0 && hour <= 18.0) ? sn : 0.0;
}
__device__ __forceinline__ double renewable_power_e(const GsTables& T, const GsRows& R, GsLaneRows S, int g, double elev) {
  const double cap = cld(T.gen_cap, g), p0 = cld(T.gen_p0, g), p1 = cld(T.gen_p1, g), p2 = cld(T.gen_p2, g);
  if (cld(T.gen_kind, g) == 0) {
    const double irr = 1000.0 * elev * (1.0 - 0.8 * ROW(R.CLOUD));
    const double tf = 1.0 - 0.004 * fmax(0.0, ROW(R.TEMP) - 25.0);
    return fmin(irr * p1 * p0 * tf, cap);
  }
  const double w = ROW(R.WIND);
  if (w < p0 || w > p2) return 0.0;
  if (w <= p1) { const double q = (w - p0) / (p1 - p0); return cap * (q * q * q); }
  return cap;
}
__device__ __forceinline__ double renewable_power(const GsTables& T, const GsRows& R, GsLaneRows S, int g) {
  return renewable_power_e(T, R, S, g, solar_elevation(ROW(R.TIME)));
}

// _apply_actions + clock + weather for this lane: the part of step() that mutates scalar state
// before the injections are formed (grid_env.py:621-651, 470-474).  `act` points at this
// instance's action row (batch-major [B][A]).
__device__ __forceinline__ void env_actions_clock(const GsTables& T, const GsRows& R, const GsEnvCfg& E,
                                                  GsLaneRows S, const double* __restrict__ act) {
  const double dt = E.timestep;
  // the action entries and state-of-charge rows of the first four batteries and the actions of the first eight
  // generators are requested before anything is computed: each is otherwise a round trip of its own
  double a_b[4], s_b[4], a_g[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a_b[q] = act[min(q, max(T.n_bats - 1, 0))]; s_b[q] = ROW(R.SOC + min(q, max(T.n_bats - 1, 0))); }
#pragma unroll
  for (int g = 0; g < 8; ++g) a_g[g] = act[T.n_bats + min(g, max(T.n_gens - 1, 0))];
  const double t_old = ROW(R.TIME), k_old = ROW(R.STEP);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= T.n_bats) break;
    const double rating = cld(T.bat_rating, q), cap = cld(T.bat_cap, q), eff = cld(T.bat_eff, q);
    const double cmd = a_b[q] * rating;
    double soc = s_b[q];
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
  for (int q = 4; q < T.n_bats; ++q) {
    const double rating = cld(T.bat_rating, q), cap = cld(T.bat_cap, q), eff = cld(T.bat_eff, q);
    const double cmd = act[q] * rating;
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
#pragma unroll
  for (int g = 0; g < 8; ++g) { if (g >= T.n_gens) break; ROW(R.CURT + g) = (a_g[g] + 1.0) / 2.0; }
  for (int g = 8; g < T.n_gens; ++g) ROW(R.CURT + g) = (act[T.n_bats + g] + 1.0) / 2.0;
  ROW(R.TIME) = t_old + dt;                          // grid_env.py:470-471
  ROW(R.STEP) = k_old + 1.0;
}

// realised power of load l (dynamics.py:54-75 when stochastic, base_power otherwise)
__device__ __forceinline__ double load_power_z(const GsTables& T, int l, double z, double prof) {
  return fmax(0.0, cld(T.load_base, l) * (prof * (1.0 + 0.1 * z)) * 1.0);
}
// (the table through any pointer: the second-generation step kernels keep a copy in LDS, kernels_flow2.hip)
template <typename TablePtr>
__device__ __forceinline__ double daily_profile_from(TablePtr table, double time_s) {
  const double hour = hour_of_day(time_s);
  const int hi = (int)hour;
  const double frac = hour - (double)hi;
  return table[hi] * (1.0 - frac) + table[(hi + 1) % 24] * frac;
}
__device__ __forceinline__ double daily_profile(double time_s) { return daily_profile_from(kDailyProfile, time_s); }

// per-bus injection in the reference's accumulation order (grid_env.py:689-718), then
// P_spec = (0 - loads) + generation (power_flow.py:112-121); Q_spec = 0 (power_flow.py:107)
__device__ __forceinline__ void bus_injection(const GsTables& T, const GsRows& R, const GsEnvCfg& E, GsLaneRows S, int i) {
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
  ROW(R.P + i) = (0.0 - gs_div_by(ls, E.power_base, E.inv_power_base)) + gs_div_by(gs, E.power_base, E.inv_power_base);
  ROW(R.Q + i) = 0.0;
}

// reset() of this lane's instance with the given seed (grid_env.py:360-408), in two parts so that a workgroup can share
// the second: the per-instance scalars / devices / weather (one lane per instance), and the per-bus and per-line rows
// (any thread, any instance: `Sx` addresses the instance, j in [0, n + m) picks the bus or line).
__device__ __forceinline__ void env_reset_lane_scalars(const GsTables& T, const GsRows& R, const GsEnvCfg& E, GsLaneRows S,
                                                       uint64_t inst, uint64_t seed) {
  ROW(R.SEEDLO) = (double)(uint32_t)seed;
  ROW(R.SEEDHI) = (double)(uint32_t)(seed >> 32);
  ROW(R.TIME) = 0.0; ROW(R.STEP) = 0.0; ROW(R.VIOL) = 0.0; ROW(R.TOTLOSS) = 0.0; ROW(R.EPREW) = 0.0;
  ROW(R.FREQ) = 60.0;                                           // grid_env.py:394
  ROW(R.IRR) = 0.0; ROW(R.WIND) = 5.0; ROW(R.TEMP) = 25.0; ROW(R.CLOUD) = 0.3;   // grid_env.py:213-218
  for (int q = 0; q < T.n_bats; ++q) { ROW(R.SOC + q) = 0.5; ROW(R.BATP + q) = 0.0; }      // grid_env.py:397-399
  for (int g = 0; g < T.n_gens; ++g) ROW(R.CURT + g) = 1.0;
  weather_update(R, E, S, inst);                                  // grid_env.py:402
  for (int g = 0; g < T.n_gens; ++g) ROW(R.GENP + g) = renewable_power(T, R, S, g);
  ROW(R.REWARD) = 0.0; ROW(R.TERM) = 0.0; ROW(R.TRUNC) = 0.0; ROW(R.VMAX) = 1.0; ROW(R.VMIN) = 1.0;
  for (int v = 0; v < 4; ++v) ROW(R.VFLAGS + v) = 0.0;
  ROW(R.LOSSES) = 0.0; ROW(R.MAXMIS) = 0.0; ROW(R.ITERS) = 0.0; ROW(R.CONV) = 0.0; ROW(R.STATUS) = 0.0;
}
__device__ __forceinline__ void env_reset_element(const GsTables& T, const GsRows& R, GsLaneRows Sx, int j) {
  if (j < T.n) {
    const double e0 = T.fixed_v[j] ? T.v_set[j] : 1.0;            // the flat start, for a warm-started sweep solver
    Sx.lane_row((size_t)(R.VM + j) * GS_LANES).put(1.0); Sx.lane_row((size_t)(R.VA + j) * GS_LANES).put(0.0);
    Sx.lane_row((size_t)(R.E + j) * GS_LANES).put(e0); Sx.lane_row((size_t)(R.F + j) * GS_LANES).put(0.0);
  } else if (j < T.n + T.m) {
    const int k = j - T.n;
    Sx.lane_row((size_t)(R.FLOW + k) * GS_LANES).put(0.0); Sx.lane_row((size_t)(R.ENVLOAD + k) * GS_LANES).put(0.0);
    Sx.lane_row((size_t)(R.LOAD + k) * GS_LANES).put(0.0);
  }
}
__device__ __forceinline__ void env_reset_lane(const GsTables& T, const GsRows& R, const GsEnvCfg& E, GsLaneRows S,
                                               uint64_t inst, uint64_t seed) {
  env_reset_lane_scalars(T, R, E, S, inst, seed);
  for (int j = 0; j < T.n + T.m; ++j) env_reset_element(T, R, S, j);
}

