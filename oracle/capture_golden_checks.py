#!/usr/bin/env python3
"""Golden vectors for the post-step checks, captured by importing the
reference in THIS container only:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 oracle/capture_golden_checks.py

Writes tests/golden/checks_*.npz and adds them to tests/golden/manifest.json.  Test infrastructure: nothing
under grid_fed_rl_gym_amd/ imports this file, and the fixtures hold inputs and expected outputs only.

  checks_safety_seq.npz   SafetyChecker.check_constraints / is_safe / get_violation_severity (utils/safety.py:97-203)
                          and SafetyMonitor.check_constraints (utils/safety.py:313-394) driven step by step over
                          seeded sequences -- both classes are stateful (previous state, consecutive violations,
                          sticky emergency mode), so the sequences are the fixture
  checks_quality.npz      AdvancedRobustPowerFlowSolver._assess_solution_quality (robust_power_flow.py:615-657)
"""
import hashlib, json, logging, os, sys, types
import numpy as np

logging.disable(logging.CRITICAL)
from grid_fed_rl.utils.safety import SafetyChecker, SafetyMonitor                          # noqa: E402
from grid_fed_rl.environments.robust_power_flow import AdvancedRobustPowerFlowSolver        # noqa: E402
from grid_fed_rl.environments.power_flow import PowerFlowSolution                           # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEVERITY = {"safe": 0, "low": 1, "medium": 2, "high": 3, "critical": 4}


def save(manifest, name, **arrays):
    clean = {k: np.asarray(v) for k, v in arrays.items()}
    for k, v in clean.items():
        assert v.dtype != object, (name, k)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **clean)
    h = hashlib.sha256()
    for k in sorted(clean):
        h.update(k.encode()); h.update(np.ascontiguousarray(clean[k]).tobytes())
    manifest["files"][name + ".npz"] = {"arrays": sorted(clean), "sha256_of_arrays": h.hexdigest()}


def sequences(rng, K, T, n, m):
    """[T, K, n] voltages, [T, K] frequency, [T, K, m] loadings; instance k exercises one family of branches."""
    v = 1.0 + 0.01 * rng.standard_normal((T, K, n))
    f = 60.0 + 0.05 * rng.standard_normal((T, K))
    ld = rng.uniform(0.1, 0.9, (T, K, m))
    v[:, 1, :4] -= 0.07                                       # mild low voltage
    v[:, 2, 3:9] += 0.08                                      # high voltage, six buses
    v[3:, 3, 5] = 0.78; v[6:, 3, 6] = 1.22                    # emergency low, later emergency high
    ld[:, 4, :] = rng.uniform(0.9, 1.3, (T, m))               # many overloads (total > 10 on some steps)
    v[:, 5, :] = 1.0 + np.linspace(0, 1.2, T)[:, None] * 0.12 * np.sign(rng.standard_normal(n))[None, :]   # ramps: rate of change
    f[:, 6] = [60.0, 59.3, 60.7, 60.6, 59.0, 56.5, 60.0, 63.5, 60.2, 59.8][:T]
    v[:, 7, :] = 1.0 + 0.06 * rng.standard_normal((T, n)); ld[:, 7, :] = rng.uniform(0.2, 1.2, (T, m))
    v[4, 0, 2] = 0.95; v[5, 0, 2] = 1.05; ld[4, 0, 1] = 1.0    # exactly on the limits: strict inequalities
    return v, f, ld


def run_safety(v, f, ld, checker_kw, monitor_kw, dt):
    T, K, n = v.shape; m = ld.shape[2]
    o = {k: np.zeros((T, K), dtype=np.int64) for k in
         ("c_voltage_low", "c_voltage_high", "c_freq_low", "c_freq_high", "c_line_overload", "c_voltage_rate", "c_freq_rate",
          "c_total", "c_is_safe", "c_severity", "m_emergency_count", "m_freq_high", "m_freq_low", "m_freq_emergency",
          "m_total", "m_action_required", "m_consecutive", "m_emergency_mode")}
    o["c_voltage_rate_value"] = np.zeros((T, K)); o["c_freq_rate_value"] = np.zeros((T, K))
    o["c_mask_low"] = np.zeros((T, K, n), dtype=np.uint8); o["c_mask_high"] = np.zeros((T, K, n), dtype=np.uint8)
    o["c_mask_overload"] = np.zeros((T, K, m), dtype=np.uint8)
    o["m_mask_high"] = np.zeros((T, K, n), dtype=np.uint8); o["m_mask_low"] = np.zeros((T, K, n), dtype=np.uint8)
    o["m_mask_overload"] = np.zeros((T, K, m), dtype=np.uint8)
    for k in range(K):
        chk, mon = SafetyChecker(**checker_kw), SafetyMonitor(**monitor_kw)
        for t in range(T):
            viol = chk.check_constraints(v[t, k].copy(), float(f[t, k]), ld[t, k].copy(), None, dt)
            for x in viol["voltage"]:
                (o["c_mask_low"] if x.violation_type == "voltage_low" else o["c_mask_high"])[t, k, x.location] = 1
            o["c_voltage_low"][t, k] = sum(x.violation_type == "voltage_low" for x in viol["voltage"])
            o["c_voltage_high"][t, k] = sum(x.violation_type == "voltage_high" for x in viol["voltage"])
            o["c_freq_low"][t, k] = sum(x.violation_type == "frequency_low" for x in viol["frequency"])
            o["c_freq_high"][t, k] = sum(x.violation_type == "frequency_high" for x in viol["frequency"])
            for x in viol["line_loading"]:
                o["c_mask_overload"][t, k, x.location] = 1
            o["c_line_overload"][t, k] = len(viol["line_loading"])
            for x in viol["rate_of_change"]:
                if x.violation_type == "voltage_rate": o["c_voltage_rate"][t, k] = 1; o["c_voltage_rate_value"][t, k] = x.value
                else: o["c_freq_rate"][t, k] = 1; o["c_freq_rate_value"][t, k] = x.value
            assert not viol["thermal"]
            o["c_total"][t, k] = sum(len(x) for x in viol.values())
            o["c_is_safe"][t, k] = chk.is_safe(viol)
            o["c_severity"][t, k] = SEVERITY[chk.get_violation_severity(viol)]
            r = mon.check_constraints(v[t, k].copy(), float(f[t, k]), ld[t, k].copy(), t)
            o["m_mask_high"][t, k, r["voltage_high"]] = 1; o["m_mask_low"][t, k, r["voltage_low"]] = 1
            o["m_emergency_count"][t, k] = len(r["voltage_emergency"])
            o["m_freq_high"][t, k] = r["frequency_high"]; o["m_freq_low"][t, k] = r["frequency_low"]
            o["m_freq_emergency"][t, k] = r["frequency_emergency"]
            o["m_mask_overload"][t, k, r["line_overload"]] = 1
            o["m_total"][t, k] = r["total_violations"]; o["m_action_required"][t, k] = r["emergency_action_required"]
            o["m_consecutive"][t, k] = mon.consecutive_violations; o["m_emergency_mode"][t, k] = mon.emergency_mode
    return o


def main():
    mpath = os.path.join(OUT, "manifest.json")
    manifest = json.load(open(mpath))
    rng = np.random.default_rng(20240607)
    K, T, n, m = 8, 10, 12, 11
    v, f, ld = sequences(rng, K, T, n, m)
    arrays = {"voltages": v, "frequency": f, "loadings": ld}
    cfgs = {"default": ({}, {}, 1.0),
            "custom": (dict(voltage_limits=(0.97, 1.03), frequency_limits=(59.8, 60.2), line_loading_limit=0.8,
                            rate_of_change_limits={"voltage": 0.02, "frequency": 0.1}),
                       dict(voltage_limits=(0.96, 1.04), frequency_limits=(59.7, 60.3), line_loading_limit=0.85,
                            emergency_voltage_limits=(0.9, 1.1), emergency_frequency_limits=(59.2, 60.8)), 0.5)}
    for name, (ckw, mkw, dt) in cfgs.items():
        for key, val in run_safety(v, f, ld, ckw, mkw, dt).items():
            arrays[f"{name}_{key}"] = val
        arrays[f"{name}_dt"] = np.array(dt)
    arrays["custom_checker_limits"] = np.array([0.97, 1.03, 59.8, 60.2, 0.8, 0.02, 0.1])
    arrays["custom_monitor_limits"] = np.array([0.96, 1.04, 59.7, 60.3, 0.85, 0.9, 1.1, 59.2, 60.8])
    save(manifest, "checks_safety_seq", **arrays)

    # quality gate
    Q, nb, nl = 96, 10, 9
    conv = rng.random(Q) < 0.85
    its = rng.integers(1, 31, Q)
    mm = 10.0 ** rng.uniform(-9, -2, Q)
    vm = 1.0 + 0.04 * rng.standard_normal((Q, nb))
    vm[rng.random(Q) < 0.25, 0] = 0.85; vm[rng.random(Q) < 0.15, 1] = 0.75; vm[rng.random(Q) < 0.15, 2] = 1.15; vm[rng.random(Q) < 0.1, 3] = 1.25
    lo = rng.uniform(0, 0.9, (Q, nl)); lo[rng.random(Q) < 0.3, 0] = 1.5; lo[rng.random(Q) < 0.15, 1] = 2.5
    fl = rng.standard_normal((Q, nl))
    vm[5, 4] = np.nan; vm[9, 4] = np.inf; fl[13, 2] = np.nan; fl[17, 3] = -np.inf
    conv[[5, 9, 13, 17]] = True
    vm[20, 0] = 0.8; vm[21, 0] = 0.9; vm[22, 0] = 1.1; vm[23, 0] = 1.2; lo[24, 0] = 1.0; lo[25, 0] = 2.0; its[26] = 5; its[27] = 20; its[28] = 21; its[29] = 6
    mm[30] = 1e-4; mm[31] = 1.0000001e-4
    dummy = types.SimpleNamespace(tolerance=1e-6)
    q = np.zeros(Q)
    for k in range(Q):
        sol = PowerFlowSolution(converged=bool(conv[k]), iterations=int(its[k]), bus_voltages=vm[k].copy(), bus_angles=np.zeros(nb),
                                line_flows=fl[k].copy(), line_loadings=lo[k].copy(), losses=0.0, max_mismatch=float(mm[k]))
        q[k] = AdvancedRobustPowerFlowSolver._assess_solution_quality(dummy, sol)
    save(manifest, "checks_quality", converged=conv, iterations=its, max_mismatch=mm, bus_voltages=vm, line_loadings=lo, line_flows=fl,
         tolerance=np.array(1e-6), quality=q)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote checks_safety_seq.npz, checks_quality.npz")


if __name__ == "__main__":
    main()
