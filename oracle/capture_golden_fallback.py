#!/usr/bin/env python3
"""Golden vectors for the linear-approximation fallback, captured by importing the
reference in THIS container only:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 oracle/capture_golden_fallback.py

Writes tests/golden/fallback_linear.npz: K seeded cases (inputs flattened per case, padded to the largest) and what
LinearApproximationSolver.solve (robust_power_flow.py:336-398) returned for each, plus the quality score
AdvancedRobustPowerFlowSolver._assess_solution_quality (:615-657) gives that answer.
"""
import hashlib, json, logging, os
import numpy as np

logging.disable(logging.CRITICAL)
from grid_fed_rl.environments.robust_power_flow import LinearApproximationSolver, AdvancedRobustPowerFlowSolver   # noqa: E402
from grid_fed_rl.environments.base import Bus, Line                                                                 # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    rng = np.random.default_rng(20250829)
    K, N, M = 40, 16, 20
    n_a = np.zeros(K, np.int32); m_a = np.zeros(K, np.int32)
    is_slack = np.zeros((K, N), np.int32); loads = np.zeros((K, N)); gens = np.zeros((K, N))
    tl = np.zeros(K); tg = np.zeros(K)
    lf = np.zeros((K, M), np.int32); lt = np.zeros((K, M), np.int32); lx = np.zeros((K, M)); lr = np.zeros((K, M))
    V = np.zeros((K, N)); TH = np.zeros((K, N)); FL = np.zeros((K, M)); LD = np.zeros((K, M)); LO = np.zeros(K); Q = np.zeros(K)
    solver = LinearApproximationSolver()
    gate = AdvancedRobustPowerFlowSolver.__new__(AdvancedRobustPowerFlowSolver)
    gate.tolerance = 1e-6
    for k in range(K):
        n = int(rng.integers(1, N + 1)) if k else 1                      # case 0: a single bus
        m = int(rng.integers(0, M + 1)) if n > 1 else 0
        slack = int(rng.integers(0, n))
        buses = [Bus(i, 4160.0, "slack" if i == slack else ("pv" if rng.random() < 0.15 else "pq")) for i in range(n)]
        lines = []
        for j in range(m):
            a, b = rng.integers(0, n, 2)
            lines.append(Line(j, int(a), int(b), float(rng.uniform(0.001, 0.05)),
                              float(rng.choice([0.0, rng.uniform(0.001, 0.1)], p=[0.15, 0.85])),
                              float(rng.choice([0.0, rng.uniform(1e5, 5e6)], p=[0.1, 0.9]))))
        scale = 10.0 ** rng.uniform(4, 8.3)                               # up to 200 MW per bus: the clips engage
        ld, gn = {}, {}
        for i in rng.permutation(n)[: int(rng.integers(0, n + 1))]:
            ld[int(i)] = float(rng.uniform(0.0, 1.0) * scale)
        for i in rng.permutation(n)[: int(rng.integers(0, n + 1))]:
            gn[int(i)] = float(rng.uniform(0.0, 1.0) * scale * rng.choice([1.0, 30.0]))
        if k % 7 == 3:
            ld = {}                                                       # empty dict: total 0, no losses
        sol = solver.solve(buses, lines, ld, gn)
        n_a[k], m_a[k] = n, m
        is_slack[k, slack] = 1
        for i, v in ld.items(): loads[k, i] = v
        for i, v in gn.items(): gens[k, i] = v
        tl[k] = sum(ld.values()) if ld else 0; tg[k] = sum(gn.values()) if gn else 0
        for j, L in enumerate(lines):
            lf[k, j], lt[k, j], lx[k, j], lr[k, j] = L.from_bus, L.to_bus, L.reactance, L.rating
        V[k, :n] = sol.bus_voltages; TH[k, :n] = sol.bus_angles
        FL[k, :m] = sol.line_flows; LD[k, :m] = sol.line_loadings; LO[k] = sol.losses
        assert sol.converged and sol.iterations == 1 and sol.max_mismatch == 0.0
        Q[k] = gate._assess_solution_quality(sol)
    arrays = dict(n=n_a, m=m_a, is_slack=is_slack, loads=loads, gens=gens, total_load=tl, total_gen=tg, line_from=lf,
                  line_to=lt, line_x=lx, line_rating=lr, bus_voltages=V, bus_angles=TH, line_flows=FL, line_loadings=LD,
                  losses=LO, quality=Q)
    np.savez_compressed(os.path.join(OUT, "fallback_linear.npz"), **arrays)
    h = hashlib.sha256()
    for key in sorted(arrays):
        h.update(key.encode()); h.update(np.ascontiguousarray(arrays[key]).tobytes())
    mp = os.path.join(OUT, "manifest.json")
    manifest = json.load(open(mp)) if os.path.exists(mp) else {"files": {}}
    manifest["files"]["fallback_linear.npz"] = {"arrays": sorted(arrays), "sha256_of_arrays": h.hexdigest()}
    json.dump(manifest, open(mp, "w"), indent=1, sort_keys=True)
    print("wrote fallback_linear.npz", K, "cases; quality range", Q.min(), Q.max(), "accepted", int((Q > 0.7).sum()))


if __name__ == "__main__":
    main()
