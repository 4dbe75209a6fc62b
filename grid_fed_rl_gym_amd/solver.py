"""Batched power-flow solvers behind the reference's solver plug point.

The reference's plug point is any object with
``solve(buses, lines, loads, generation) -> PowerFlowSolution``
(reference environments/power_flow.py:25-46; passed as ``GridEnvironment(power_flow_solver=...)``,
grid_env.py:169, or to ``parallel_power_flow_batch(solver, configs)``, utils/distributed.py:835).
``BatchedNewtonRaphsonSolver`` keeps that method with the same argument meaning and the same
"never raises for non-convergence" behaviour, and adds ``solve_batch`` for thousands of
instances of one topology per call.  All arithmetic happens in libgridstep's HIP kernels.

``jacobian`` selects between the reference's Jacobian exactly as coded (``"as_coded"``: the
J11 diagonal carries the sign written at power_flow.py:248, with which the reference never
converges under load) and the true derivative (``"exact"``, the default, which converges
quadratically).  ``"as_coded"`` exists so that results can be compared with the reference
number for number.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .components import BatchedPowerFlowSolution, PowerFlowError, PowerFlowSolution
from .feeders import FeederSpec, flatten_network

_EMPTY_I = np.zeros(0, dtype=np.int32)
_EMPTY_F = np.zeros(0, dtype=np.float64)


def _network_spec(buses: Sequence[Any], lines: Sequence[Any]) -> FeederSpec:
    bus_ids, bus_type, v_set, frm, to, r, x, rating = flatten_network(buses, lines)
    return FeederSpec(name="network", bus_ids=bus_ids, bus_type=bus_type, v_set=v_set, frm=frm, to=to,
                      r=r, x=x, rating=rating)


def injections_from_dicts(spec: FeederSpec, loads: Dict[Any, float], generation: Dict[Any, float]) -> np.ndarray:
    """P_spec row from the reference's two dicts keyed by bus id; unknown ids are ignored
    (power_flow.py:112-121)."""
    idx = spec.bus_index()
    p = np.zeros(spec.n)
    for bus_id, power in loads.items():
        if bus_id in idx:
            p[idx[bus_id]] -= power
    for bus_id, power in generation.items():
        if bus_id in idx:
            p[idx[bus_id]] += power
    return p


class _BatchedSolverBase:
    solver_kind = "nr"

    def __init__(self, tolerance: float = 1e-6, max_iterations: int = 50, acceleration_factor: float = 1.0,
                 jacobian: str = "exact", zero_z: str = "open", linear_solver: str = "auto", device: int = 0,
                 waves_per_group: int = 0, **kwargs: Any) -> None:
        if jacobian not in _lib.JACOBIAN:
            raise ValueError(f"jacobian must be one of {sorted(_lib.JACOBIAN)}")
        if zero_z not in _lib.ZERO_Z:
            raise ValueError(f"zero_z must be one of {sorted(_lib.ZERO_Z)}")
        if linear_solver not in _lib.LINSOLVE:
            raise ValueError(f"linear_solver must be one of {sorted(_lib.LINSOLVE)}")
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.acceleration_factor = acceleration_factor
        self.jacobian = jacobian
        self.zero_z = zero_z
        self.linear_solver = linear_solver
        self.device = device
        self.waves_per_group = waves_per_group
        self._handles: Dict[Tuple, _lib.Handle] = {}

    # -- handle cache: one compiled topology per (network, batch) --------------------------
    def _config(self) -> "_lib.gs_config":
        return _lib.make_config(solver_kind=_lib.SOLVER[self.solver_kind], jacobian_mode=_lib.JACOBIAN[self.jacobian],
                                zero_z_mode=_lib.ZERO_Z[self.zero_z], linear_solver=_lib.LINSOLVE[self.linear_solver],
                                max_iterations=int(self.max_iterations), tolerance=float(self.tolerance),
                                acceleration_factor=float(self.acceleration_factor),
                                waves_per_group=int(self.waves_per_group))

    def handle_for(self, spec: FeederSpec, batch: int) -> "_lib.Handle":
        """The native handle that serves (spec, batch): what the device-side checks (safety.py) bind to."""
        return self._handle(spec, batch)

    def _handle(self, spec: FeederSpec, batch: int) -> "_lib.Handle":
        key = (spec.sha256(), int(batch), self.tolerance, self.max_iterations, self.acceleration_factor,
               self.jacobian, self.zero_z, self.linear_solver, self.device, self.waves_per_group)
        h = self._handles.get(key)
        if h is None:
            if len(self._handles) >= 8:
                _, old = self._handles.popitem()
                old.close()
            h = _lib.Handle(spec, self._config(), batch, self.device)
            self._handles[key] = h
        return h

    def close(self) -> None:
        for h in self._handles.values():
            h.close()
        self._handles.clear()

    # -- batched API ------------------------------------------------------------------------
    def solve_batch(self, network: Union[FeederSpec, Tuple[Sequence[Any], Sequence[Any]]], P_spec,
                    Q_spec=None) -> BatchedPowerFlowSolution:
        """Solve B instances of one topology.  ``P_spec[B, n]`` is the net injection per bus
        (generation - load) in the units of the line impedances (per unit)."""
        spec = network if isinstance(network, FeederSpec) else _network_spec(*network)
        P = np.ascontiguousarray(P_spec, dtype=np.float64)
        if P.ndim != 2 or P.shape[1] != spec.n:
            raise PowerFlowError(f"P_spec must have shape (B, {spec.n}), got {P.shape}")
        out = self._handle(spec, P.shape[0]).solve(P, Q_spec)
        return BatchedPowerFlowSolution(converged=out["converged"].astype(bool), iterations=out["iterations"],
                                        bus_voltages=out["bus_voltages"], bus_angles=out["bus_angles"],
                                        line_flows=out["line_flows"], line_loadings=out["line_loadings"],
                                        losses=out["losses"], max_mismatch=out["max_mismatch"], status=out["status"])

    # -- the reference's plug-point signature (power_flow.py:38-46) ---------------------------
    def solve(self, buses: List[Any], lines: List[Any], loads: Dict[Any, float],
              generation: Dict[Any, float]) -> PowerFlowSolution:
        spec = _network_spec(buses, lines)
        has_slack = any(getattr(b, "bus_type", "pq") == "slack" for b in buses)
        P = injections_from_dicts(spec, loads, generation)[None, :]
        sol = self.solve_batch(spec, P)[0]
        if not has_slack and len(buses) > 0:
            buses[0].bus_type = "slack"          # the reference re-types bus 0 in place (power_flow.py:141)
        return sol

    def quality_score(self, sol: BatchedPowerFlowSolution) -> np.ndarray:
        """Vectorised ``_assess_solution_quality`` (robust_power_flow.py:615-657): the accept
        signal (> 0.7) the reference's fallback chain applies to each solver's answer."""
        v = sol.bus_voltages
        q = np.ones(len(sol))
        bad = ((v < 0.8) | (v > 1.2)).any(axis=1)
        warn = ((v < 0.9) | (v > 1.1)).any(axis=1) & ~bad
        q = np.where(bad, q * 0.3, np.where(warn, q * 0.7, q))
        if sol.line_loadings.shape[1] > 0:
            mx = sol.line_loadings.max(axis=1)
            q = np.where(mx > 2.0, q * 0.2, np.where(mx > 1.0, q * 0.5, q))
        q = np.where(sol.max_mismatch > self.tolerance * 100, q * 0.6, q)
        q = np.where(sol.iterations <= 5, q * 1.1, np.where(sol.iterations > 20, q * 0.9, q))
        return np.where(sol.converged, np.minimum(1.0, q), 0.0)


class BatchedNewtonRaphsonSolver(_BatchedSolverBase):
    """Newton-Raphson load flow (reference ``NewtonRaphsonSolver``, power_flow.py:76-358) for
    batches of one topology; radial networks use the forest elimination kernel, meshed ones
    the statically scheduled sparse block LU."""
    solver_kind = "nr"


class NewtonRaphsonSolver(BatchedNewtonRaphsonSolver):
    """Reference class name; same constructor keywords (tolerance, max_iterations,
    acceleration_factor)."""


class FastDecoupledSolver(BatchedNewtonRaphsonSolver):
    """The reference's FastDecoupledSolver delegates to Newton-Raphson with the same
    tolerance / max_iterations (power_flow.py:361-378); so does this one."""

    def __init__(self, tolerance: float = 1e-6, max_iterations: int = 50, **kwargs: Any) -> None:
        super().__init__(tolerance=tolerance, max_iterations=max_iterations, **kwargs)


class BatchedForwardBackwardSweepSolver(_BatchedSolverBase):
    """Forward/backward sweep for radial feeders with pq buses.  The reference advertises this
    solver (README.md:187-197, ``DistributionPowerFlow``) without implementing it; parity is
    defined against Newton-Raphson with the exact Jacobian on the same feeder."""
    solver_kind = "fbs"

    def __init__(self, tolerance: float = 1e-6, max_iterations: int = 100, **kwargs: Any) -> None:
        kwargs.pop("jacobian", None)
        super().__init__(tolerance=tolerance, max_iterations=max_iterations, jacobian="exact", **kwargs)


DistributionPowerFlow = BatchedForwardBackwardSweepSolver


def parallel_power_flow_batch(solver: _BatchedSolverBase, network_configs: List[Tuple], num_workers: Optional[int] = None
                              ) -> List[PowerFlowSolution]:
    """Batched form of the reference helper of the same name (utils/distributed.py:835-888):
    ``network_configs`` is a list of ``(buses, lines, loads, generation)``.  Configurations that
    share a topology are solved in one kernel launch; results come back in input order.
    ``num_workers`` is accepted and ignored (there is no thread pool)."""
    if not network_configs:
        return []
    groups: Dict[str, List[int]] = {}
    specs: Dict[str, FeederSpec] = {}
    for i, (buses, lines, _, _) in enumerate(network_configs):
        spec = _network_spec(buses, lines)
        key = spec.sha256()
        groups.setdefault(key, []).append(i)
        specs[key] = spec
    results: List[Optional[PowerFlowSolution]] = [None] * len(network_configs)
    for key, members in groups.items():
        spec = specs[key]
        P = np.stack([injections_from_dicts(spec, network_configs[i][2], network_configs[i][3]) for i in members])
        sol = solver.solve_batch(spec, P)
        for row, i in enumerate(members):
            results[i] = sol[row]
    return results  # type: ignore[return-value]


class BatchedRobustPowerFlowSolver:
    """The accept / fall-back policy of the reference's ``AdvancedRobustPowerFlowSolver.solve``
    (robust_power_flow.py:523-613) for a batch: the primary solver answers every instance, the quality gate
    (``_assess_solution_quality``, :615-657) is evaluated on the device, and the instances it rejects (score <= 0.7)
    get the reference's linear approximation (``LinearApproximationSolver``, :336-398) written over their rows by
    ``gs_fallback_linear`` -- the other instances are not touched and nothing but the verdicts crosses the host boundary
    in between.  What the reference's chain has in between (Gauss-Seidel as coded diverges, the "fast decoupled" class is
    Newton-Raphson again) and behind (a constant-voltage guess) is not reproduced; caches, circuit breakers
    and health scores are host bookkeeping outside the path.

    ``solve_batch(network, loads_w, gens_w)`` takes the totals of the reference's two dicts per bus, ``[B, n]`` each, in
    the reference's units (W); the net injection handed to the primary solver is ``(gens_w - loads_w) / power_base``.
    Returns the solution with two more per-instance fields: ``method`` (0 primary, 1 linear approximation, -1 neither
    passed the gate -- the reference raises PowerFlowError there) and ``quality``.
    """

    METHODS = {0: "newton_raphson", 1: "linear_approximation", -1: "failed"}

    def __init__(self, primary: Optional[_BatchedSolverBase] = None, accept_quality: float = 0.7, power_base: float = 1.0,
                 **primary_kwargs: Any) -> None:
        self.primary = primary if primary is not None else BatchedNewtonRaphsonSolver(**primary_kwargs)
        self.accept_quality = float(accept_quality)
        self.power_base = float(power_base)

    def close(self) -> None:
        self.primary.close()

    def solve_batch(self, network: Union[FeederSpec, Tuple[Sequence[Any], Sequence[Any]]], loads_w, gens_w,
                    total_load=None, total_gen=None) -> BatchedPowerFlowSolution:
        from .safety import PostStepChecks
        spec = network if isinstance(network, FeederSpec) else _network_spec(*network)
        L = np.ascontiguousarray(loads_w, dtype=np.float64); G = np.ascontiguousarray(gens_w, dtype=np.float64)
        if L.ndim != 2 or L.shape[1] != spec.n or G.shape != L.shape:
            raise PowerFlowError(f"loads_w / gens_w must have shape (B, {spec.n}), got {L.shape} / {G.shape}")
        h = self.primary.handle_for(spec, L.shape[0])
        h.upload_injections((G - L) / self.power_base)
        h.solve_device()
        gate = PostStepChecks(h, quality_tolerance=self.primary.tolerance, loading="solution")
        try:
            gate.run(); q0 = gate.download()["quality"]
            rejected = ~(q0 > self.accept_quality)
            method = np.zeros(L.shape[0], dtype=np.int32)
            quality = q0.copy()
            if rejected.any():
                applied = h.fallback_linear(L, G, total_load, total_gen, mask=rejected)
                gate.run(); q1 = gate.download()["quality"]
                quality = np.where(applied, q1, q0)
                method = np.where(applied, np.where(q1 > self.accept_quality, 1, -1), 0).astype(np.int32)
        finally:
            gate.close()
        out = h.download_solution()
        sol = BatchedPowerFlowSolution(converged=out["converged"].astype(bool), iterations=out["iterations"],
                                       bus_voltages=out["bus_voltages"], bus_angles=out["bus_angles"],
                                       line_flows=out["line_flows"], line_loadings=out["line_loadings"],
                                       losses=out["losses"], max_mismatch=out["max_mismatch"], status=out["status"])
        sol.method = method
        sol.quality = quality
        return sol

    def solve(self, buses: List[Any], lines: List[Any], loads: Dict[Any, float], generation: Dict[Any, float]):
        """The reference's signature; returns (solution, method_name) like ``SolverResult.solution / .method_used``.
        Totals are summed in the dicts' own order, as the reference's linear solver does."""
        spec = _network_spec(buses, lines)
        idx = spec.bus_index()
        L = np.zeros((1, spec.n)); G = np.zeros((1, spec.n))
        for bus_id, p in loads.items():
            if bus_id in idx:
                L[0, idx[bus_id]] += p
        for bus_id, p in generation.items():
            if bus_id in idx:
                G[0, idx[bus_id]] += p
        tl = np.array([float(sum(loads.values())) if loads else 0.0]); tg = np.array([float(sum(generation.values())) if generation else 0.0])
        sol = self.solve_batch(spec, L, G, tl, tg)
        if sol.method[0] < 0:
            raise PowerFlowError("All power flow solvers failed: neither the primary answer nor the linear approximation passed the quality gate")
        return sol[0], self.METHODS[int(sol.method[0])]
