"""Batched, device-side mirror of the reference's safety checks and solution-quality gate
 over the gs_checks_* entry points of libgridstep.so.

Reference classes (grid_fed_rl/utils/safety.py, environments/robust_power_flow.py):
  SafetyChecker(voltage_limits, frequency_limits, line_loading_limit, thermal_limits, rate_of_change_limits)
      .check_constraints(bus_voltages, frequency, line_loadings, thermal_data, timestep) -> {kind: [ConstraintViolation]}
      .is_safe(violations) / .get_violation_severity(violations)                              safety.py:97-203
  SafetyMonitor(voltage_limits, frequency_limits, line_loading_limit, emergency_*_limits)
      .check_constraints(bus_voltages, frequency, line_loadings, timestep) -> dict            safety.py:293-394
  AdvancedRobustPowerFlowSolver._assess_solution_quality(solution) -> float                    robust_power_flow.py:615-657

Here the arrays never leave the GPU: one kernel reads the voltages, loadings and frequency the last
``step()`` / ``solve_device()`` left in HBM and returns per-instance counts, flags and (on request) bit
masks.  The objects are stateful exactly where the reference classes are (previous state for the rate
limits, consecutive violations, sticky emergency mode).  ``thermal_data`` is not modelled -- the
environment has no temperatures -- so severity never reaches ``'critical'``.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

SEVERITY_NAMES = ("safe", "low", "medium", "high", "critical")
CI = {name: k for k, name in enumerate(
    ["c_n_voltage_low", "c_n_voltage_high", "c_frequency_low", "c_frequency_high", "c_n_line_overload", "c_voltage_rate_violation",
     "c_frequency_rate_violation", "c_total", "c_severity", "m_n_voltage_high", "m_n_voltage_low", "m_n_voltage_emergency",
     "m_frequency_high", "m_frequency_low", "m_frequency_emergency", "m_n_line_overload", "m_total_violations",
     "m_emergency_action_required", "m_consecutive_violations", "m_emergency_mode"])}
CF = {"voltage_rate": 0, "frequency_rate": 1, "quality": 2}
BM_C_LOW, BM_C_HIGH, BM_M_LOW, BM_M_HIGH, BM_M_EMERGENCY = 1, 2, 4, 8, 16
LM_C_OVERLOAD, LM_M_OVERLOAD = 1, 2

_dp, _ip, _up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class gs_checks_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("loading_source", C.c_int32),
                ("voltage_limits", C.c_double * 2), ("frequency_limits", C.c_double * 2), ("line_loading_limit", C.c_double),
                ("rate_voltage", C.c_double), ("rate_frequency", C.c_double), ("timestep", C.c_double),
                ("mon_voltage_limits", C.c_double * 2), ("mon_frequency_limits", C.c_double * 2), ("mon_line_loading_limit", C.c_double),
                ("mon_emergency_voltage", C.c_double * 2), ("mon_emergency_frequency", C.c_double * 2),
                ("quality_tolerance", C.c_double)]


class gs_checks_view(C.Structure):
    _fields_ = [("ints", _ip), ("reals", _dp), ("bus_mask", _up), ("line_mask", _up)]


_CK = C.c_void_p
CHECKS_SYMBOLS = [
    ("gs_checks_create", C.c_int, [C.c_void_p, C.POINTER(gs_checks_config), C.POINTER(_CK)]),
    ("gs_checks_destroy", None, [_CK]),
    ("gs_checks_set_frequency", C.c_int, [_CK, _dp]),
    ("gs_checks_run", C.c_int, [_CK]),
    ("gs_checks_download", C.c_int, [_CK, C.POINTER(gs_checks_view)]),
    ("gs_checks_reset", C.c_int, [_CK, _up]),
    ("gs_checks_timing_enable", C.c_int, [_CK, C.c_int32]),
    ("gs_checks_timing_read", C.c_int, [_CK, _dp, C.POINTER(C.c_int64)]),
    ("gs_checks_set_fused", C.c_int, [_CK, C.c_int32, C.c_int32]),
]
_bound = False


def _bind():
    global _bound
    lib = _lib.load()
    if not _bound:
        for name, res, args in CHECKS_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return lib


def _native_handle(source: Any) -> "_lib.Handle":
    """BatchedGridEnvironment (``.handle``), a raw _lib.Handle, or anything exposing one."""
    h = getattr(source, "handle", source)
    if not isinstance(h, _lib.Handle):
        raise TypeError("checks need a BatchedGridEnvironment or a native Handle (pass solver.handle_for(spec, batch))")
    return h


class PostStepChecks:
    """One gs_checks object: SafetyChecker + SafetyMonitor + quality gate evaluated together on the device."""

    def __init__(self, source: Any, *, checker: Optional[Dict[str, Any]] = None, monitor: Optional[Dict[str, Any]] = None,
                 quality_tolerance: float = 1e-6, loading: str = "environment", timestep: float = 1.0, fused: bool = False,
                 fused_masks: bool = True):
        self._lib = _bind()
        self._handle = _native_handle(source)            # keeps the gs_handle alive for as long as the checks exist
        ck = dict(voltage_limits=(0.95, 1.05), frequency_limits=(59.5, 60.5), line_loading_limit=1.0,
                  rate_of_change_limits={"voltage": 0.1, "frequency": 0.5})
        ck.update(checker or {})
        mo = dict(voltage_limits=(0.90, 1.10), frequency_limits=(59.0, 61.0), line_loading_limit=1.0,
                  emergency_voltage_limits=(0.80, 1.20), emergency_frequency_limits=(57.0, 63.0))
        mo.update(monitor or {})
        if loading not in ("environment", "solution"):
            raise ValueError("loading must be 'environment' (|P|/rating, Line.update_state) or 'solution' (|S|/rating)")
        cfg = gs_checks_config()
        cfg.struct_size = C.sizeof(gs_checks_config)
        cfg.loading_source = 1 if loading == "environment" else 0
        cfg.voltage_limits[:] = ck["voltage_limits"]; cfg.frequency_limits[:] = ck["frequency_limits"]
        cfg.line_loading_limit = ck["line_loading_limit"]
        cfg.rate_voltage = ck["rate_of_change_limits"]["voltage"]; cfg.rate_frequency = ck["rate_of_change_limits"]["frequency"]
        cfg.timestep = float(timestep)
        cfg.mon_voltage_limits[:] = mo["voltage_limits"]; cfg.mon_frequency_limits[:] = mo["frequency_limits"]
        cfg.mon_line_loading_limit = mo["line_loading_limit"]
        cfg.mon_emergency_voltage[:] = mo["emergency_voltage_limits"]; cfg.mon_emergency_frequency[:] = mo["emergency_frequency_limits"]
        cfg.quality_tolerance = float(quality_tolerance)
        self.checker_config, self.monitor_config = ck, mo
        self._c = _CK()
        rc = self._lib.gs_checks_create(self._handle._h, C.byref(cfg), C.byref(self._c))
        if rc != _lib.GS_OK:
            raise RuntimeError(f"gs_checks_create failed ({rc}): {self._handle.last_error()}")
        self.B, self.n, self.m = self._handle.B, self._handle.spec.n, self._handle.spec.m
        self._handle._adopt(self)                        # the handle closes its checks before itself
        self.fused = False
        if fused:
            self.set_fused(True, fused_masks)

    def _check(self, rc: int) -> None:
        if rc != _lib.GS_OK:
            raise RuntimeError(f"libgridstep checks call failed ({rc}): {self._handle.last_error()}")

    def close(self) -> None:
        if getattr(self, "_c", None) is not None and self._c:
            self._lib.gs_checks_destroy(self._c)
            self._c = _CK()
            self._handle._release(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_frequency(self, frequency_hz: Optional[np.ndarray]) -> None:
        if frequency_hz is None:
            self._check(self._lib.gs_checks_set_frequency(self._c, None)); return
        f = np.ascontiguousarray(np.broadcast_to(np.asarray(frequency_hz, dtype=np.float64), (self.B,)))
        self._check(self._lib.gs_checks_set_frequency(self._c, f.ctypes.data_as(_dp)))

    def set_fused(self, on: bool = True, masks: bool = True) -> None:
        """Evaluate the checks inside every later ``step()`` (the step kernel's epilogue already holds the voltages and
        loadings): ``download()`` then returns the checks of the last step and ``run()`` must not be called as well."""
        self._check(self._lib.gs_checks_set_fused(self._c, int(bool(on)), int(bool(masks))))
        self.fused = bool(on)

    def run(self) -> None:
        """One check_constraints call per instance on the handle's current device state (asynchronous)."""
        if self.fused:
            raise RuntimeError("these checks are fused into step(): download() has the last step's result")
        self._check(self._lib.gs_checks_run(self._c))

    def download(self, masks: bool = False) -> Dict[str, np.ndarray]:
        ints = np.zeros((len(CI), self.B), dtype=np.int32); reals = np.zeros((len(CF), self.B))
        view = gs_checks_view(ints.ctypes.data_as(_ip), reals.ctypes.data_as(_dp), None, None)
        bm = lm = None
        if masks:
            bm = np.zeros((self.B, self.n), dtype=np.uint8); lm = np.zeros((self.B, self.m), dtype=np.uint8)
            view.bus_mask, view.line_mask = bm.ctypes.data_as(_up), lm.ctypes.data_as(_up)
        self._check(self._lib.gs_checks_download(self._c, C.byref(view)))
        out: Dict[str, np.ndarray] = {name: ints[k] for name, k in CI.items()}
        out.update({name: reals[k] for name, k in CF.items()})
        if masks:
            out["bus_mask"], out["line_mask"] = bm, lm
        return out

    def reset(self, mask: Optional[np.ndarray] = None) -> None:
        if mask is None:
            self._check(self._lib.gs_checks_reset(self._c, None)); return
        mk = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if mk.shape != (self.B,):
            raise ValueError(f"mask must have shape ({self.B},)")
        self._check(self._lib.gs_checks_reset(self._c, mk.ctypes.data_as(_up)))

    def timing_enable(self, on: bool = True) -> None:
        """HIP-event pairs around every later ``run()`` (off by default: nothing is recorded unless somebody reads it)."""
        self._check(self._lib.gs_checks_timing_enable(self._c, int(bool(on))))

    def timing_read(self) -> Tuple[float, int]:
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.gs_checks_timing_read(self._c, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


class BatchedSafetyChecker:
    """``SafetyChecker`` (utils/safety.py:97-203) for every instance of a batched environment at once.
    Same constructor keywords; ``check_constraints()`` takes no arrays -- it checks the state the last
    ``step()`` left on the GPU -- and returns arrays over the batch instead of lists of ConstraintViolation.
    ``violations(b)`` rebuilds the reference-shaped dict for one instance."""

    def __init__(self, source: Any, voltage_limits: Tuple[float, float] = (0.95, 1.05), frequency_limits: Tuple[float, float] = (59.5, 60.5),
                 line_loading_limit: float = 1.0, thermal_limits: Optional[Dict[str, float]] = None,
                 rate_of_change_limits: Optional[Dict[str, float]] = None, *, timestep: float = 1.0, loading: str = "environment"):
        self.voltage_limits, self.frequency_limits, self.line_loading_limit = tuple(voltage_limits), tuple(frequency_limits), line_loading_limit
        self.thermal_limits = thermal_limits or {"transformer": 100.0, "generator": 150.0}      # kept for API parity; not evaluated
        self.rate_of_change_limits = rate_of_change_limits or {"voltage": 0.1, "frequency": 0.5}
        self._source = source
        self._checks = PostStepChecks(source, checker=dict(voltage_limits=self.voltage_limits, frequency_limits=self.frequency_limits,
                                                           line_loading_limit=line_loading_limit,
                                                           rate_of_change_limits=self.rate_of_change_limits),
                                      timestep=timestep, loading=loading)
        self._last: Optional[Dict[str, np.ndarray]] = None

    def check_constraints(self, masks: bool = False) -> Dict[str, np.ndarray]:
        self._checks.run()
        d = self._checks.download(masks=masks)
        out = {k[2:]: v for k, v in d.items() if k.startswith("c_")}
        out["voltage_rate"], out["frequency_rate"] = d["voltage_rate"], d["frequency_rate"]
        out["is_safe"] = out["total"] == 0
        if masks:
            out["voltage_low"] = (d["bus_mask"] & BM_C_LOW) != 0; out["voltage_high"] = (d["bus_mask"] & BM_C_HIGH) != 0
            out["line_overload"] = (d["line_mask"] & LM_C_OVERLOAD) != 0
        self._last = out
        return out

    @staticmethod
    def is_safe(violations: Dict[str, np.ndarray]) -> np.ndarray:
        return violations["total"] == 0

    @staticmethod
    def get_violation_severity(violations: Dict[str, np.ndarray]) -> List[str]:
        return [SEVERITY_NAMES[int(s)] for s in violations["severity"]]

    def violations(self, b: int, bus_voltages: np.ndarray, frequency: float, line_loadings: np.ndarray) -> Dict[str, List[Tuple[str, Any, float, float]]]:
        """The reference's ``{kind: [ConstraintViolation]}`` for instance ``b`` as (type, location, value, limit) tuples, rebuilt
        from the last ``check_constraints(masks=True)`` and the instance's host-side arrays."""
        v = self._last
        if v is None or "voltage_low" not in v:
            raise RuntimeError("call check_constraints(masks=True) first")
        out: Dict[str, List[Tuple[str, Any, float, float]]] = {"voltage": [], "frequency": [], "line_loading": [], "thermal": [], "rate_of_change": []}
        for i in range(len(bus_voltages)):
            if v["voltage_low"][b, i]: out["voltage"].append(("voltage_low", i, float(bus_voltages[i]), self.voltage_limits[0]))
            elif v["voltage_high"][b, i]: out["voltage"].append(("voltage_high", i, float(bus_voltages[i]), self.voltage_limits[1]))
        if v["frequency_low"][b]: out["frequency"].append(("frequency_low", "system", float(frequency), self.frequency_limits[0]))
        elif v["frequency_high"][b]: out["frequency"].append(("frequency_high", "system", float(frequency), self.frequency_limits[1]))
        for k in np.nonzero(v["line_overload"][b])[0]:
            out["line_loading"].append(("line_overload", int(k), float(line_loadings[k]), self.line_loading_limit))
        if v["voltage_rate_violation"][b]: out["rate_of_change"].append(("voltage_rate", "system", float(v["voltage_rate"][b]), self.rate_of_change_limits["voltage"]))
        if v["frequency_rate_violation"][b]: out["rate_of_change"].append(("frequency_rate", "system", float(v["frequency_rate"][b]), self.rate_of_change_limits["frequency"]))
        return out

    def reset(self, mask: Optional[np.ndarray] = None) -> None:
        self._checks.reset(mask)

    def close(self) -> None:
        self._checks.close()


class BatchedSafetyMonitor:
    """``SafetyMonitor.check_constraints`` (utils/safety.py:293-394) over the batch: same keywords, array results, the consecutive-violation
    counter and the sticky ``emergency_mode`` per instance."""

    def __init__(self, source: Any, voltage_limits: Tuple[float, float] = (0.90, 1.10), frequency_limits: Tuple[float, float] = (59.0, 61.0),
                 line_loading_limit: float = 1.0, emergency_voltage_limits: Tuple[float, float] = (0.80, 1.20),
                 emergency_frequency_limits: Tuple[float, float] = (57.0, 63.0), *, loading: str = "environment"):
        self._checks = PostStepChecks(source, monitor=dict(voltage_limits=tuple(voltage_limits), frequency_limits=tuple(frequency_limits),
                                                           line_loading_limit=line_loading_limit,
                                                           emergency_voltage_limits=tuple(emergency_voltage_limits),
                                                           emergency_frequency_limits=tuple(emergency_frequency_limits)), loading=loading)
        self.emergency_mode = np.zeros(self._checks.B, dtype=bool)
        self.consecutive_violations = np.zeros(self._checks.B, dtype=np.int64)

    def check_constraints(self, masks: bool = False) -> Dict[str, np.ndarray]:
        self._checks.run()
        d = self._checks.download(masks=masks)
        out = {k[2:]: v for k, v in d.items() if k.startswith("m_")}
        for k in ("frequency_high", "frequency_low", "frequency_emergency", "emergency_action_required", "emergency_mode"):
            out[k] = out[k].astype(bool)
        if masks:
            out["voltage_high"] = (d["bus_mask"] & BM_M_HIGH) != 0; out["voltage_low"] = (d["bus_mask"] & BM_M_LOW) != 0
            out["voltage_emergency"] = (d["bus_mask"] & BM_M_EMERGENCY) != 0
            out["line_overload"] = (d["line_mask"] & LM_M_OVERLOAD) != 0
        self.emergency_mode = out["emergency_mode"].copy()
        self.consecutive_violations = out["consecutive_violations"].astype(np.int64)
        return out

    def reset(self, mask: Optional[np.ndarray] = None) -> None:
        self._checks.reset(mask)
        if mask is None:
            self.emergency_mode[:] = False; self.consecutive_violations[:] = 0
        else:
            mk = np.asarray(mask).astype(bool)
            self.emergency_mode[mk] = False; self.consecutive_violations[mk] = 0

    def close(self) -> None:
        self._checks.close()


def device_quality_score(source: Any, tolerance: float = 1e-6, loading: str = "solution") -> np.ndarray:
    """``_assess_solution_quality`` (robust_power_flow.py:615-657) of the solution / step currently on the device, per
    instance; > 0.7 is what the reference's fallback chain accepts."""
    ck = PostStepChecks(source, quality_tolerance=tolerance, loading=loading)
    try:
        ck.run()
        return ck.download()["quality"].copy()
    finally:
        ck.close()
