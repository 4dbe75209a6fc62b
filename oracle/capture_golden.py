#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference IN THE BUILD CONTAINER.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 oracle/capture_golden.py

The reference never travels: only the numeric inputs/outputs written here (plain arrays,
``allow_pickle=False``) are committed, together with this script.  Two tiers (SURVEY.md
section 8(c)):

* Tier A -- the reference exactly as coded (J11-diagonal sign and all), function by function
  and for ``solve()`` truncated at 1..3 iterations, plus the deterministic environment
  trajectory.
* Tier B -- the reference's own ``solve()`` driven through a capture-only subclass that
  corrects the single J11-diagonal term (``ExactNR`` below).  Everything else in the loop is
  the reference's code, unchanged; its converged answers are the physics anchor.

Capture hygiene: NewtonRaphsonSolver is called directly (no solution caches); the env is
built with an explicit solver, stochastic_loads=False, weather_variation=False and the
process-global power_flow_cache is cleared before every step.
"""
import hashlib
import json
import logging
import os
import sys
import warnings

import numpy as np

logging.disable(logging.CRITICAL)
warnings.simplefilter("ignore")

from grid_fed_rl.environments.base import Bus, Line, Load                       # noqa: E402
from grid_fed_rl.environments.power_flow import NewtonRaphsonSolver             # noqa: E402
from grid_fed_rl.environments import dynamics as ref_dyn                        # noqa: E402
from grid_fed_rl.environments.grid_env import GridEnvironment                   # noqa: E402
from grid_fed_rl.feeders.base import SimpleRadialFeeder                         # noqa: E402
from grid_fed_rl.feeders.ieee_feeders import IEEE13Bus, IEEE123Bus              # noqa: E402
from grid_fed_rl.feeders.synthetic import ScalableFeeder                        # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from grid_fed_rl_gym_amd import feeders as F                                    # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
MANIFEST = {"numpy": np.__version__, "python": sys.version.split()[0], "files": {},
            "reference": "danieleschmidt/grid-fed-rl-gym @ 2025-08-29 (/root/reference)"}


class ExactNR(NewtonRaphsonSolver):
    """Reference solver with the J11-diagonal sign corrected -- capture-only (Tier B)."""

    def _build_jacobian(self, Y, V, buses, slack_bus, pv_buses, pq_buses):
        J = super()._build_jacobian(Y, V, buses, slack_bus, pv_buses, pq_buses)
        Vm, B = np.abs(V), Y.imag
        non_slack = [i for i in range(len(buses)) if i != slack_bus]
        for row, i in enumerate(non_slack):
            J[row, row] -= 2.0 * Vm[i] * Vm[i] * B[i, i]
        return J


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    clean = {k: np.asarray(v) for k, v in arrays.items()}
    for k, v in clean.items():
        assert v.dtype != object, (name, k)
    np.savez_compressed(path, **clean)
    h = hashlib.sha256()
    for k in sorted(clean):
        h.update(k.encode()); h.update(np.ascontiguousarray(clean[k]).tobytes())
    MANIFEST["files"][name + ".npz"] = {"arrays": sorted(clean), "sha256_of_arrays": h.hexdigest()}


def objects_from_spec(spec):
    names = {0: "pq", 1: "pv", 2: "slack"}
    buses = []
    for i, bid in enumerate(spec.bus_ids):
        b = Bus(id=bid, voltage_level=4160.0, bus_type=names[int(spec.bus_type[i])])
        b.voltage_magnitude = float(spec.v_set[i])
        buses.append(b)
    lines = [Line(id=f"l{k}", from_bus=spec.bus_ids[int(spec.frm[k])], to_bus=spec.bus_ids[int(spec.to[k])],
                  resistance=float(spec.r[k]), reactance=float(spec.x[k]), rating=float(spec.rating[k]))
             for k in range(spec.m)]
    return buses, lines


def net_arrays(buses, lines):
    spec_like = F.flatten_network(buses, lines)
    _, bus_type, v_set, frm, to, r, x, rating = spec_like
    return dict(bus_type=bus_type, v_set=v_set, frm=frm, to=to, r=r, x=x, rating=rating)


def classify(buses):
    slack, pv, pq = None, [], []
    for i, b in enumerate(buses):
        if b.bus_type == "slack":
            slack = i
        elif b.bus_type == "pv":
            pv.append(i)
        else:
            pq.append(i)
    return (0 if slack is None else slack), pv, pq


def function_level(name, buses, lines, P_spec, seed, n_points=3, keep_J=True):
    """G1-G5 for one network: Y, then mismatch / Jacobian / dx / updated V / flows / losses at
    the flat start and at ``n_points`` seeded perturbed voltage vectors."""
    s = NewtonRaphsonSolver()
    n = len(buses)
    Y = s.build_admittance_matrix(buses, lines)
    slack, pv, pq = classify(buses)
    bus_map = {b.id: i for i, b in enumerate(buses)}
    rng = np.random.default_rng(seed)
    out = dict(net_arrays(buses, lines), Y_re=Y.real, Y_im=Y.imag, P_spec=P_spec)
    pts = [np.ones(n, dtype=complex)]
    for _ in range(n_points):
        vm = 1.0 + 0.05 * rng.standard_normal(n)
        va = 0.1 * rng.standard_normal(n)
        V = vm * np.exp(1j * va)
        V[slack] = 1.0 + 0j
        pts.append(V)
    non_slack = [i for i in range(n) if i != slack]
    for p, V in enumerate(pts):
        S = V * np.conj(Y @ V)
        dP = np.zeros(n); dQ = np.zeros(n)
        for i in non_slack:
            dP[i] = P_spec[i] - S.real[i]
        for i in pq:
            dQ[i] = 0.0 - S.imag[i]
        J = s._build_jacobian(Y, V, buses, slack, pv, pq)
        rhs = np.concatenate([dP[non_slack], dQ[pq]])
        out[f"V{p}_re"], out[f"V{p}_im"] = V.real, V.imag
        out[f"S{p}_re"], out[f"S{p}_im"] = S.real, S.imag
        out[f"dP{p}"], out[f"dQ{p}"] = dP, dQ
        out[f"mm{p}"] = max(np.max(np.abs(dP)), np.max(np.abs(dQ)))
        if keep_J or p == 0:
            out[f"J{p}"] = J
        try:
            dx = np.linalg.solve(J, rhs)
            V2 = V.copy()
            s._apply_corrections(dx, V2, buses, slack, pv, pq)
            out[f"dx{p}"] = dx
            out[f"Vnew{p}_re"], out[f"Vnew{p}_im"] = V2.real, V2.imag
        except np.linalg.LinAlgError:
            out[f"singular{p}"] = np.array(1)
        fl, ld = s._calculate_line_flows(V, Y, lines, bus_map)
        out[f"flow{p}"], out[f"loading{p}"] = fl, ld
        out[f"loss{p}"] = np.sum(V * np.conj(Y @ V)).real
    out["n_points"] = np.array(len(pts))
    save("fn_" + name, **out)


def solve_record(prefix, out, sol):
    out[prefix + "converged"] = np.array(bool(sol.converged))
    out[prefix + "iterations"] = np.array(int(sol.iterations))
    out[prefix + "Vm"] = sol.bus_voltages
    out[prefix + "Va"] = sol.bus_angles
    out[prefix + "flow"] = sol.line_flows
    out[prefix + "loading"] = sol.line_loadings
    out[prefix + "losses"] = np.array(float(sol.losses))
    out[prefix + "max_mismatch"] = np.array(float(sol.max_mismatch))


def solve_level(name, buses, lines, loads, gens, its=(1, 2, 3), exact_scales=()):
    """G6 (as-coded solve at iteration caps) and G11 (ExactNR to convergence) for one network."""
    bus_map = {b.id: i for i, b in enumerate(buses)}
    P_spec = np.zeros(len(buses))
    for k, v in loads.items():
        P_spec[bus_map[k]] -= v
    for k, v in gens.items():
        P_spec[bus_map[k]] += v
    out = dict(net_arrays(buses, lines), P_spec=P_spec, its=np.array(its), exact_scales=np.array(exact_scales, dtype=float))
    types0 = [b.bus_type for b in buses]

    def restore():
        # solve() re-types bus 0 as "slack" when no bus is (power_flow.py:141); undo that
        # between captures so every record is a first call on the network as described.
        for b, t in zip(buses, types0):
            b.bus_type = t

    for k in its:
        restore()
        sol = NewtonRaphsonSolver(tolerance=1e-6, max_iterations=k).solve(buses, lines, dict(loads), dict(gens))
        solve_record(f"A{k}_", out, sol)
    for q, lam in enumerate(exact_scales):
        restore()
        sol = ExactNR(tolerance=1e-6, max_iterations=50).solve(
            buses, lines, {k: v * lam for k, v in loads.items()}, {k: v * lam for k, v in gens.items()})
        solve_record(f"B{q}_", out, sol)
        # one more pass at 1e-12 so that the anchor is converged far below the 1e-6 acceptance band
        restore()
        sol = ExactNR(tolerance=1e-12, max_iterations=50).solve(
            buses, lines, {k: v * lam for k, v in loads.items()}, {k: v * lam for k, v in gens.items()})
        solve_record(f"C{q}_", out, sol)
    save("solve_" + name, **out)


def main():
    # ---------------- networks --------------------------------------------------------
    env3_b = [Bus(1, 12.47e3, "slack"), Bus(2, 12.47e3, "pq"), Bus(3, 12.47e3, "pq")]
    env3_l = [Line("line_1_2", 1, 2, 0.01, 0.02, 5e6), Line("line_2_3", 2, 3, 0.015, 0.025, 3e6)]

    def radial(n):
        f = SimpleRadialFeeder(n)
        return f.buses, f.lines

    ieee13 = IEEE13Bus()
    np.random.seed(0)
    ieee123 = IEEE123Bus()
    scal20 = ScalableFeeder(20, seed=3)
    scal123 = ScalableFeeder(123, seed=1)
    spec13 = F.ieee13_like("epsilon")
    b13e, l13e = objects_from_spec(spec13)
    spec123 = F.ieee123_like()
    b123t, l123t = objects_from_spec(spec123)
    specm = F.random_meshed(30, 12, seed=7)
    bm, lm = objects_from_spec(specm)
    # a network with pv buses and a non-unit slack set-point
    specpv = F.random_meshed(12, 4, seed=11)
    specpv.bus_type[[3, 7]] = 1
    specpv.v_set[[0, 3, 7]] = [1.02, 1.01, 0.99]
    bpv, lpv = objects_from_spec(specpv)
    # no bus typed slack (reference defaults bus 0 but keeps it in the pq list)
    specns = F.simple_radial(5)
    specns.bus_type[0] = 0
    bns, lns = objects_from_spec(specns)

    # ---------------- G1-G5 function level ---------------------------------------------
    def pspec(buses, scale, seed):
        rng = np.random.default_rng(seed)
        p = -scale * rng.uniform(0.2, 1.0, len(buses))
        return p

    function_level("env3", env3_b, env3_l, np.array([0.0, -0.2, -0.1]), 1)
    for n in (5, 13):
        b, l = radial(n)
        function_level(f"radial{n}", b, l, pspec(b, 0.02, n), 2)
    b, l = radial(123)
    function_level("radial123", b, l, pspec(b, 0.004, 123), 3, n_points=1, keep_J=False)
    function_level("ieee13_as_coded", ieee13.buses, ieee13.lines, pspec(ieee13.buses, 0.03, 13), 4)
    function_level("ieee13_eps", b13e, l13e, pspec(b13e, 0.03, 13), 5)
    function_level("ieee123_as_coded", ieee123.buses, ieee123.lines, pspec(ieee123.buses, 0.004, 9), 6,
                   n_points=1, keep_J=False)
    function_level("scal20", scal20.buses, scal20.lines, pspec(scal20.buses, 0.02, 20), 7)
    function_level("scal123", scal123.buses, scal123.lines, pspec(scal123.buses, 0.004, 21), 8,
                   n_points=1, keep_J=False)
    function_level("tree123", b123t, l123t, pspec(b123t, 0.008, 22), 9, n_points=1, keep_J=False)
    function_level("meshed30", bm, lm, pspec(bm, 0.02, 23), 10)
    function_level("pv12", bpv, lpv, pspec(bpv, 0.03, 24), 11)
    function_level("noslack5", bns, lns, pspec(bns, 0.02, 25), 12)

    # ---------------- G6-G8, G11 solve level -------------------------------------------
    solve_level("env3", env3_b, env3_l, {2: 0.2, 3: 0.15}, {3: 0.05}, exact_scales=(0.5, 1.0, 2.0))
    solve_level("env3_zero", env3_b, env3_l, {}, {})                                   # G7
    solve_level("env3_watts", env3_b, env3_l, {2: 2e6, 3: 1.5e6}, {})                 # F4 as coded
    for n, ld in ((5, 0.02), (13, 0.02), (123, 0.0008)):
        b, l = radial(n)
        solve_level(f"radial{n}", b, l, {bb.id: ld for bb in b[1:]}, {}, exact_scales=(0.5, 1.0, 1.5))
    pu13 = {ld.bus: ld.base_power / 10e6 for ld in ieee13.loads}
    solve_level("ieee13_as_coded", ieee13.buses, ieee13.lines, pu13, {})              # G8 singular
    solve_level("ieee13_eps", b13e, l13e, pu13, {}, exact_scales=(0.5, 1.0, 1.5))
    pu123 = {}
    for i, p in zip(spec123.load_bus, spec123.load_base):
        bid = spec123.bus_ids[int(i)]
        pu123[bid] = pu123.get(bid, 0.0) + p / 10e6
    solve_level("tree123", b123t, l123t, pu123, {spec123.bus_ids[int(spec123.gen_bus[0])]: 0.03},
                exact_scales=(0.5, 1.0, 1.5))
    pum = {specm.bus_ids[int(i)]: p / 10e6 for i, p in zip(specm.load_bus, specm.load_base)}
    solve_level("meshed30", bm, lm, pum, {}, exact_scales=(1.0, 3.0))
    pus20 = {ld.bus: ld.base_power / 10e6 for ld in scal20.loads}
    solve_level("scal20", scal20.buses, scal20.lines, pus20, {}, exact_scales=(1.0,))
    pus123 = {ld.bus: ld.base_power / 10e6 for ld in scal123.loads}
    solve_level("scal123", scal123.buses, scal123.lines, pus123, {}, its=(1, 2), exact_scales=(1.0,))
    pupv = {specpv.bus_ids[int(i)]: p / 10e6 * 3 for i, p in zip(specpv.load_bus, specpv.load_base)}
    solve_level("pv12", bpv, lpv, pupv, {specpv.bus_ids[3]: 0.05, specpv.bus_ids[7]: 0.04}, exact_scales=(1.0,))
    solve_level("noslack5", bns, lns, {bb.id: 0.02 for bb in bns[1:]}, {}, its=(1, 2, 3))

    # ---------------- G9 dynamics known answers ----------------------------------------
    rng = np.random.default_rng(99)
    K = 64
    bat = dict(soc=rng.uniform(0, 1, K), cmd=rng.uniform(-8e5, 8e5, K), dt=rng.choice([1.0, 60.0, 900.0], K),
               cap=rng.uniform(500, 2000, K), rating=rng.uniform(2e5, 6e5, K), eff=rng.uniform(0.85, 0.99, K),
               power0=rng.uniform(-1e5, 1e5, K))
    bat["cmd"][:4] = 0.0
    soc1, pow1 = np.zeros(K), np.zeros(K)
    for k in range(K):
        gd = ref_dyn.GridDynamics()
        bm_ = ref_dyn.BatteryModel(bat["cap"][k], bat["rating"][k], bat["eff"][k], bat["soc"][k])
        bm_.current_power = bat["power0"][k]
        gd.add_battery_model("b", bm_)
        gd.update_batteries({"b": float(bat["cmd"][k])}, float(bat["dt"][k]))
        soc1[k], pow1[k] = bm_.soc, bm_.current_power
    t = rng.uniform(0, 3 * 86400, K)
    basep = rng.uniform(1e4, 3e6, K)
    pf = rng.uniform(0.85, 0.99, K)
    lp, lq = np.zeros(K), np.zeros(K)
    for k in range(K):
        lp[k], lq[k] = ref_dyn.TimeVaryingLoadModel().get_power(float(t[k]), float(basep[k]), noise_factor=0,
                                                               power_factor=float(pf[k]))
    cloud, temp, wind = rng.uniform(0, 1, K), rng.uniform(5, 45, K), rng.uniform(0, 30, K)
    cap = rng.uniform(2e5, 2e6, K); eff = rng.uniform(0.15, 0.22, K); area = rng.uniform(800, 8000, K)
    sol_p, wind_p = np.zeros(K), np.zeros(K)
    for k in range(K):
        w = ref_dyn.WeatherData(800.0, float(wind[k]), float(temp[k]), float(cloud[k]))
        sol_p[k] = ref_dyn.SolarPVModel(float(eff[k]), float(area[k])).get_power(float(t[k]), w, float(cap[k]))
        wind_p[k] = ref_dyn.WindTurbineModel().get_power(float(t[k]), w, float(cap[k]))
    f0 = rng.uniform(58, 62, K); imb = rng.uniform(-5, 5, K)
    f0[:3] = [60.0, 55.001, 64.999]; imb[:3] = [0.1, -500.0, 500.0]
    f1 = np.zeros(K)
    for k in range(K):
        gd = ref_dyn.GridDynamics(); gd.frequency = float(f0[k]); gd.update_frequency(float(imb[k]), float(bat["dt"][k]))
        f1[k] = gd.frequency
    save("dynamics", bat_soc=bat["soc"], bat_cmd=bat["cmd"], bat_dt=bat["dt"], bat_cap=bat["cap"],
         bat_rating=bat["rating"], bat_eff=bat["eff"], bat_power0=bat["power0"], bat_soc1=soc1, bat_power1=pow1,
         t=t, base_power=basep, pf=pf, load_p=lp, load_q=lq, cloud=cloud, temp=temp, wind=wind, cap=cap,
         eff=eff, area=area, solar_p=sol_p, wind_p=wind_p, f0=f0, imb=imb, f1=f1)

    # ---------------- G10 deterministic env trajectories -------------------------------
    from grid_fed_rl.utils.performance_optimization import power_flow_cache

    def trajectory(name, sources, max_it, actions, t0=0.0, exact=False, wind_speed=None):
        cls = ExactNR if exact else NewtonRaphsonSolver
        env = GridEnvironment(feeder=None, stochastic_loads=False, weather_variation=False,
                              renewable_sources=list(sources), episode_length=len(actions) - 2,
                              power_flow_solver=cls(tolerance=1e-6, max_iterations=max_it))
        obs0, _ = env.reset(seed=0)
        env.current_time = t0
        if wind_speed is not None:
            env.weather.wind_speed = wind_speed
        obs, rew, term, trunc = [np.asarray(obs0, dtype=float)], [], [], []
        conv, its, mm, loss, vmax, vmin, viol = [], [], [], [], [], [], []
        for a in actions:
            power_flow_cache.cache.clear()
            o, r, te, tr, info = env.step(np.asarray(a, dtype=float))
            obs.append(np.asarray(o, dtype=float)); rew.append(r); term.append(te); trunc.append(tr)
            conv.append(info["power_flow_converged"]); loss.append(info["total_losses"])
            vmax.append(info["max_voltage"]); vmin.append(info["min_voltage"])
            v = info["constraint_violations"]
            viol.append([v["voltage_high"], v["voltage_low"], v["frequency_high"], v["frequency_low"]])
        save("env_" + name, actions=np.asarray(actions, dtype=float), obs=np.stack(obs), reward=np.array(rew),
             terminated=np.array(term), truncated=np.array(trunc), converged=np.array(conv),
             losses=np.array(loss), vmax=np.array(vmax), vmin=np.array(vmin), violations=np.array(viol),
             t0=np.array(t0), max_it=np.array(max_it), wind_speed=np.array(-1.0 if wind_speed is None else wind_speed),
             episode_length=np.array(len(actions) - 2))

    rng = np.random.default_rng(5)
    acts1 = [[0.0], [0.5], [-1.0]] + rng.uniform(-1, 1, (12, 1)).tolist()
    trajectory("ref3_norenew_it1", [], 1, acts1)
    acts3 = [[0.0, 0.0, 0.0], [0.5, -0.2, 0.9], [-1.0, 1.0, -1.0]] + rng.uniform(-1, 1, (12, 3)).tolist()
    trajectory("ref3_solarwind_it1", ["solar", "wind"], 1, acts3, t0=10 * 3600.0, wind_speed=8.0)
    trajectory("ref3_solarwind_it2", ["solar", "wind"], 2, acts3, t0=17.5 * 3600.0, wind_speed=14.0)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
    print("wrote", len(MANIFEST["files"]), "fixture files to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
