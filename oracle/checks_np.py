"""CPU restatement of the reference's post-step checks -- TEST INFRASTRUCTURE (oracle), never imported by the
product.  Batched over instances along axis 0; stateful exactly where the reference classes are.

  SafetyChecker.check_constraints / is_safe / get_violation_severity   utils/safety.py:97-203
  SafetyMonitor.check_constraints                                      utils/safety.py:313-394
  AdvancedRobustPowerFlowSolver._assess_solution_quality               robust_power_flow.py:615-657

Pinned by tests/golden/checks_safety_seq.npz and checks_quality.npz (oracle/capture_golden_checks.py ran the
reference classes themselves).  `thermal_data` is not modelled (the environment has no temperatures), so the
'critical' severity, which only thermal violations can reach, does not occur.
"""
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

SEVERITY_NAMES = ("safe", "low", "medium", "high", "critical")


@dataclass
class CheckerConfig:            # SafetyChecker.__init__ defaults, safety.py:100-112
    voltage_limits: Tuple[float, float] = (0.95, 1.05)
    frequency_limits: Tuple[float, float] = (59.5, 60.5)
    line_loading_limit: float = 1.0
    rate_voltage: float = 0.1
    rate_frequency: float = 0.5


@dataclass
class MonitorConfig:            # SafetyMonitor.__init__ defaults, safety.py:296-311
    voltage_limits: Tuple[float, float] = (0.90, 1.10)
    frequency_limits: Tuple[float, float] = (59.0, 61.0)
    line_loading_limit: float = 1.0
    emergency_voltage_limits: Tuple[float, float] = (0.80, 1.20)
    emergency_frequency_limits: Tuple[float, float] = (57.0, 63.0)


@dataclass
class CheckerState:
    has_prev: Optional[np.ndarray] = None      # [B] bool
    prev_v: Optional[np.ndarray] = None        # [B, n]
    prev_f: Optional[np.ndarray] = None        # [B]


@dataclass
class MonitorState:
    consecutive: Optional[np.ndarray] = None   # [B] int
    emergency_mode: Optional[np.ndarray] = None  # [B] bool


def checker_step(cfg: CheckerConfig, st: CheckerState, v: np.ndarray, f: np.ndarray, ld: np.ndarray, dt: float = 1.0) -> Dict[str, np.ndarray]:
    """One SafetyChecker.check_constraints call per instance (safety.py:114-182)."""
    v = np.asarray(v, float); f = np.asarray(f, float); ld = np.asarray(ld, float)
    B = v.shape[0]
    if st.has_prev is None:
        st.has_prev = np.zeros(B, bool); st.prev_v = np.zeros_like(v); st.prev_f = np.zeros(B)
    low = v < cfg.voltage_limits[0]                       # :129-137  (elif: a bus is never both)
    high = ~low & (v > cfg.voltage_limits[1])
    f_low = f < cfg.frequency_limits[0]                   # :140-147
    f_high = ~f_low & (f > cfg.frequency_limits[1])
    over = ld > cfg.line_loading_limit                    # :150-154
    with np.errstate(invalid="ignore"):
        vrate = np.max(np.abs(v - st.prev_v), axis=1) / dt   # :168 (np.max propagates NaN)
        frate = np.abs(f - st.prev_f) / dt                    # :174
    v_viol = st.has_prev & (vrate > cfg.rate_voltage)
    f_viol = st.has_prev & (frate > cfg.rate_frequency)
    st.has_prev = np.ones(B, bool); st.prev_v = v.copy(); st.prev_f = f.copy()   # :181-184
    total = low.sum(1) + high.sum(1) + f_low + f_high + over.sum(1) + v_viol + f_viol
    severity = np.where(total > 5, 3, np.where(total > 2, 2, np.where(total > 0, 1, 0)))   # :199-206 without thermal
    return dict(voltage_low=low, voltage_high=high, n_voltage_low=low.sum(1), n_voltage_high=high.sum(1),
                frequency_low=f_low, frequency_high=f_high, line_overload=over, n_line_overload=over.sum(1),
                voltage_rate=vrate, voltage_rate_violation=v_viol, frequency_rate=frate, frequency_rate_violation=f_viol,
                total=total, is_safe=total == 0, severity=severity)


def monitor_step(cfg: MonitorConfig, st: MonitorState, v: np.ndarray, f: np.ndarray, ld: np.ndarray) -> Dict[str, np.ndarray]:
    """One SafetyMonitor.check_constraints call per instance (safety.py:313-394)."""
    v = np.asarray(v, float); f = np.asarray(f, float); ld = np.asarray(ld, float)
    B = v.shape[0]
    if st.consecutive is None:
        st.consecutive = np.zeros(B, np.int64); st.emergency_mode = np.zeros(B, bool)
    high = v > cfg.voltage_limits[1]; low = v < cfg.voltage_limits[0]                       # :333-337
    em_high = v > cfg.emergency_voltage_limits[1]; em_low = v < cfg.emergency_voltage_limits[0]   # :340-349
    n_em = em_high.sum(1) + em_low.sum(1)
    action = n_em > 0
    f_high = f > cfg.frequency_limits[1]                                                     # :352-355 (elif)
    f_low = ~f_high & (f < cfg.frequency_limits[0])
    f_em = (f > cfg.emergency_frequency_limits[1]) | (f < cfg.emergency_frequency_limits[0])  # :358-361
    action = action | f_em
    over = ld > cfg.line_loading_limit                                                       # :364-365
    total = high.sum(1) + low.sum(1) + n_em + f_high + f_low + f_em + over.sum(1)            # :368-376
    st.consecutive = np.where(total > 0, st.consecutive + 1, 0)                              # :379-383
    trigger = action | (st.consecutive > 5) | (total > 10)                                   # :386-390
    st.emergency_mode = st.emergency_mode | trigger
    return dict(voltage_high=high, voltage_low=low, n_voltage_high=high.sum(1), n_voltage_low=low.sum(1), n_voltage_emergency=n_em,
                frequency_high=f_high, frequency_low=f_low, frequency_emergency=f_em, line_overload=over, n_line_overload=over.sum(1),
                total_violations=total, emergency_action_required=trigger, consecutive_violations=st.consecutive.copy(),
                emergency_mode=st.emergency_mode.copy())


def quality(converged, iterations, max_mismatch, bus_voltages, line_loadings, line_flows, tolerance: float) -> np.ndarray:
    """_assess_solution_quality per instance (robust_power_flow.py:615-657); np.min / np.max propagate NaN, so a NaN
    voltage or loading switches its penalty off -- and a non-finite voltage or flow zeroes the score anyway."""
    v = np.asarray(bus_voltages, float); ld = np.asarray(line_loadings, float); fl = np.asarray(line_flows, float)
    B = v.shape[0]
    q = np.ones(B)
    with np.errstate(invalid="ignore"):
        if v.shape[1] > 0:
            mn, mx = np.min(v, axis=1), np.max(v, axis=1)
            severe = (mn < 0.8) | (mx > 1.2)
            moderate = ~severe & ((mn < 0.9) | (mx > 1.1))
            q = np.where(severe, q * 0.3, np.where(moderate, q * 0.7, q))
        if ld.shape[1] > 0:
            ml = np.max(ld, axis=1)
            q = np.where(ml > 2.0, q * 0.2, np.where(ml > 1.0, q * 0.5, q))
    q = np.where(np.asarray(max_mismatch) > tolerance * 100, q * 0.6, q)
    bad = ~np.isfinite(v).all(axis=1) | ~np.isfinite(fl).all(axis=1)
    its = np.asarray(iterations)
    q = np.where(its <= 5, q * 1.1, np.where(its > 20, q * 0.9, q))
    q = np.minimum(q, 1.0)
    return np.where(np.asarray(converged, bool) & ~bad, q, 0.0)
