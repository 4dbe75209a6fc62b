"""Host-side network containers with the reference's attribute names.

These mirror the *interface* of the reference's ``Bus`` / ``Line`` / ``Load`` attribute bags
(reference grid_fed_rl/environments/base.py:197-295) and of ``PowerFlowSolution``
(environments/power_flow.py:12-22) so that code written against the reference -- and the
reference's own feeder objects, which are duck-typed -- can be handed to the batched solver
and environment unchanged.  They carry no behaviour of their own: the batched path flattens
them once into structure-of-arrays form (feeders.FeederSpec) and never touches them again.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Union

import numpy as np

BusId = Union[int, str]


class Bus:
    """Electrical bus: ``id``, ``bus_type`` in {"slack","pv","pq"}, set-point magnitude."""

    def __init__(self, id: BusId, voltage_level: float = 12.47e3, bus_type: str = "pq",
                 base_voltage: float = 1.0, **kwargs: Any) -> None:
        self.id = id
        self.voltage_level = voltage_level
        self.bus_type = bus_type
        self.base_voltage = base_voltage
        self.voltage_magnitude = 1.0
        self.voltage_angle = 0.0
        self.parameters = kwargs


class Line:
    """Series branch between two bus ids; impedance in per unit, rating in VA."""

    def __init__(self, id: BusId, from_bus: BusId, to_bus: BusId, resistance: float,
                 reactance: float, rating: float, **kwargs: Any) -> None:
        self.id = id
        self.from_bus = from_bus
        self.to_bus = to_bus
        self.resistance = resistance
        self.reactance = reactance
        self.rating = rating
        self.power_flow = 0.0
        self.loading = 0.0
        self.parameters = kwargs


class Load:
    """Load at a bus; ``active_power``/``reactive_power`` are the static values the
    reference reports in observations (base.py:282-283)."""

    def __init__(self, id: BusId, bus: BusId, base_power: float, power_factor: float = 0.95,
                 **kwargs: Any) -> None:
        self.id = id
        self.bus = bus
        self.base_power = base_power
        self.power_factor = power_factor
        self.active_power = base_power
        self.reactive_power = base_power * math.tan(math.acos(power_factor))
        self.parameters = kwargs


@dataclass
class PowerFlowSolution:
    """Single-instance result record; field names as reference power_flow.py:12-22."""
    converged: bool
    iterations: int
    bus_voltages: np.ndarray
    bus_angles: np.ndarray
    line_flows: np.ndarray
    line_loadings: np.ndarray
    losses: float
    max_mismatch: float


@dataclass
class BatchedPowerFlowSolution:
    """The same record with a leading batch axis on every field, plus per-instance status
    (0 converged, 1 iteration cap, 2 singular Jacobian, 3 non-finite mismatch, 4 replaced by the linear-approximation
    fallback)."""
    converged: np.ndarray        # bool  [B]
    iterations: np.ndarray       # int32 [B]
    bus_voltages: np.ndarray     # f64   [B, n]
    bus_angles: np.ndarray       # f64   [B, n]
    line_flows: np.ndarray       # f64   [B, m]
    line_loadings: np.ndarray    # f64   [B, m]
    losses: np.ndarray           # f64   [B]
    max_mismatch: np.ndarray     # f64   [B]
    status: np.ndarray           # int32 [B]

    def __len__(self) -> int:
        return int(self.converged.shape[0])

    def __getitem__(self, b: int) -> PowerFlowSolution:
        return PowerFlowSolution(bool(self.converged[b]), int(self.iterations[b]),
                                 self.bus_voltages[b], self.bus_angles[b], self.line_flows[b],
                                 self.line_loadings[b], float(self.losses[b]),
                                 float(self.max_mismatch[b]))


class PowerFlowError(Exception):
    """Raised for call-level failures of the batched solver (bad shapes, device errors).
    Per-instance numerical failures never raise; they set ``status`` (reference convention:
    solve() returns converged=False, power_flow.py:186-190)."""


class InvalidActionError(Exception):
    """Raised when an action batch has the wrong shape or non-finite entries
    (reference name: utils/exceptions.py:68)."""
