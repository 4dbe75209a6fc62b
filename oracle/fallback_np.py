"""CPU restatement of the reference's fallback policy for one load flow -- TEST
INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

  linear_approximation   LinearApproximationSolver.solve (environments/robust_power_flow.py:336-398), literally: the
                         per-bus voltage rules (5 % per 10 MW of load, 2 % per 20 MW of generation, clipped), the
                         angle recurrence over the lines in list order, one flow value for every line, 3 % losses
  accept_or_fall_back    the accept rule of AdvancedRobustPowerFlowSolver.solve (:567-572): the first answer whose
                         quality score (oracle/checks_np.py quality()) exceeds 0.7 is taken

Pinned by tests/golden/fallback_linear.npz (oracle/capture_golden_fallback.py ran the reference class itself).
Arrays are per instance: loads[n] / gens[n] are the totals of the reference's two dicts by bus index (0 = no entry),
total_load / total_gen the sums of the dict values IN DICT ORDER (the caller owns that order).
"""
import numpy as np


def linear_approximation(is_slack, loads, gens, total_load, total_gen, line_from, line_to, line_x, line_rating):
    n, m = len(is_slack), len(line_from)
    V = np.ones(n)
    theta = np.zeros(n)
    for i in range(n):                                      # robust_power_flow.py:356-370
        if loads[i] != 0 and not is_slack[i]:
            V[i] = 1.0 - (loads[i] / 10e6) * 0.05
            V[i] = min(max(V[i], 0.85), 1.15)
    for i in range(n):
        if gens[i] != 0 and not is_slack[i]:
            V[i] = min(V[i] + (gens[i] / 20e6) * 0.02, 1.10)
    pfl = 0.0
    if n > 1:                                               # :373-382
        pfl = (total_gen - total_load) / max(m, 1)
        for k in range(m):
            i, j = int(line_from[k]), int(line_to[k])
            if line_x[k] > 0:
                theta[j] = theta[i] - pfl * line_x[k] / 100
    flows = np.full(m, abs(pfl))                            # :385-386
    loadings = np.array([f / r if r > 0 else 0.0 for f, r in zip(flows, line_rating)], dtype=np.float64).reshape(m)
    losses = total_load * 0.03 if total_load > 0 else 0.0   # :388
    return dict(converged=True, iterations=1, bus_voltages=V, bus_angles=theta, line_flows=flows,
                line_loadings=loadings, losses=losses, max_mismatch=0.0)


def accept_or_fall_back(quality_primary, quality_fallback, accept=0.7):
    """method per instance: 0 = the primary solver's answer stands, 1 = the linear approximation replaces it,
    -1 = neither passes the gate (the reference raises PowerFlowError there)."""
    qp, qf = np.asarray(quality_primary), np.asarray(quality_fallback)
    return np.where(qp > accept, 0, np.where(qf > accept, 1, -1)).astype(np.int32)
