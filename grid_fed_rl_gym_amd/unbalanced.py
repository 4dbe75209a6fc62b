"""Three-phase unbalanced radial load flow (BASELINE.json config 5) -- host side of the gs3_* ABI.

The reference names ``UnbalancedPowerFlow`` (README.md:187-197, API_REFERENCE.md:420) but
contains no implementation: this is new functionality.  The class keeps the solver plug-point
conventions (tolerance / max_iterations constructor, result record with converged / iterations /
losses / max_mismatch, never raising for non-convergence) and adds a phase axis.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .components import PowerFlowError

_dp, _ip, _up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class gs3_topology(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("n", C.c_int32), ("source", C.c_int32), ("reserved", C.c_int32),
                ("parent", _ip), ("phases", _up), ("z_re", _dp), ("z_im", _dp), ("v_source", _dp)]


class gs3_solution_view(C.Structure):
    _fields_ = [("v_re", _dp), ("v_im", _dp), ("losses", _dp), ("max_mismatch", _dp), ("iterations", _ip), ("converged", _up)]


GS3_SYMBOLS = [
    ("gs3_create", C.c_int, [C.POINTER(gs3_topology), C.c_double, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("gs3_destroy", None, [C.c_void_p]),
    ("gs3_last_error", C.c_char_p, [C.c_void_p]),
    ("gs3_solve", C.c_int, [C.c_void_p, _dp, _dp, C.POINTER(gs3_solution_view)]),
    ("gs3_upload_injections", C.c_int, [C.c_void_p, _dp, _dp]),
    ("gs3_solve_device", C.c_int, [C.c_void_p]),
    ("gs3_download_solution", C.c_int, [C.c_void_p, C.POINTER(gs3_solution_view)]),
    ("gs3_synchronize", C.c_int, [C.c_void_p]),
    ("gs3_timing_read", C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    ("gs3_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
]


def _lib3():
    lib = _lib.load()
    if not getattr(lib, "_gs3_bound", False):
        for name, res, args in GS3_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib._gs3_bound = True
    return lib


@dataclass
class UnbalancedFeederSpec:
    """Radial three-phase feeder: node ``source`` is the substation; every other node hangs off
    ``parent[i]`` through a line with 3x3 series impedance ``z[i]`` (per unit) carrying the phases
    in ``phases[i]`` (bit mask a=1, b=2, c=4, a subset of the parent's)."""
    name: str
    parent: np.ndarray          # i32 [n]
    phases: np.ndarray          # u8  [n]
    z: np.ndarray               # c128 [n, 3, 3]
    source: int = 0
    v_source: tuple = (1.0, 1.0, 1.0)

    @property
    def n(self) -> int:
        return int(len(self.parent))


@dataclass
class UnbalancedSolution:
    converged: np.ndarray       # bool [B]
    iterations: np.ndarray      # i32  [B]
    voltages: np.ndarray        # c128 [B, n, 3]; absent phases are 0
    losses: np.ndarray          # f64  [B]
    max_mismatch: np.ndarray    # f64  [B]

    @property
    def bus_voltages(self) -> np.ndarray:
        return np.abs(self.voltages)

    @property
    def bus_angles(self) -> np.ndarray:
        return np.angle(self.voltages)


class UnbalancedPowerFlow:
    """Batched three-phase forward/backward sweep on one GPU."""

    def __init__(self, tolerance: float = 1e-6, max_iterations: int = 100, device: int = 0) -> None:
        self.tolerance, self.max_iterations, self.device = float(tolerance), int(max_iterations), int(device)
        self._h = C.c_void_p()
        self._key = None
        self._keep = None

    def _handle(self, spec: UnbalancedFeederSpec, batch: int):
        key = (id(spec), batch)
        if self._key == key and self._h.value:
            return self._h
        self.close()
        lib = _lib3()
        keep = dict(parent=np.ascontiguousarray(spec.parent, dtype=np.int32), phases=np.ascontiguousarray(spec.phases, dtype=np.uint8),
                    zre=np.ascontiguousarray(spec.z.real, dtype=np.float64), zim=np.ascontiguousarray(spec.z.imag, dtype=np.float64),
                    vs=np.ascontiguousarray(spec.v_source, dtype=np.float64))
        t = gs3_topology(C.sizeof(gs3_topology), spec.n, int(spec.source), 0, keep["parent"].ctypes.data_as(_ip),
                         keep["phases"].ctypes.data_as(_up), keep["zre"].ctypes.data_as(_dp), keep["zim"].ctypes.data_as(_dp),
                         keep["vs"].ctypes.data_as(_dp))
        h = C.c_void_p()
        rc = lib.gs3_create(C.byref(t), self.tolerance, self.max_iterations, int(batch), self.device, C.byref(h))
        if rc != 0:
            raise PowerFlowError(f"gs3_create failed ({rc}): {lib.gs3_last_error(None).decode()}")
        self._h, self._key, self._keep, self._n, self._B = h, key, keep, spec.n, int(batch)
        return h

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise PowerFlowError(f"libgridstep error {rc}: {_lib3().gs3_last_error(self._h).decode()}")

    def close(self) -> None:
        if self._h.value:
            _lib3().gs3_destroy(self._h)
            self._h = C.c_void_p()
            self._key = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buffers(self):
        B, n = self._B, self._n
        out = dict(v_re=np.empty((B, n, 3)), v_im=np.empty((B, n, 3)), losses=np.empty(B), max_mismatch=np.empty(B),
                   iterations=np.empty(B, dtype=np.int32), converged=np.empty(B, dtype=np.uint8))
        view = gs3_solution_view(out["v_re"].ctypes.data_as(_dp), out["v_im"].ctypes.data_as(_dp), out["losses"].ctypes.data_as(_dp),
                                 out["max_mismatch"].ctypes.data_as(_dp), out["iterations"].ctypes.data_as(_ip),
                                 out["converged"].ctypes.data_as(_up))
        return out, view

    @staticmethod
    def _wrap(out) -> UnbalancedSolution:
        return UnbalancedSolution(out["converged"].astype(bool), out["iterations"], out["v_re"] + 1j * out["v_im"],
                                  out["losses"], out["max_mismatch"])

    def solve_batch(self, spec: UnbalancedFeederSpec, P_spec, Q_spec=None) -> UnbalancedSolution:
        """``P_spec`` / ``Q_spec`` [B, n, 3]: net injection per node and phase (generation - load), per unit."""
        P = np.ascontiguousarray(P_spec, dtype=np.float64)
        if P.ndim != 3 or P.shape[1:] != (spec.n, 3):
            raise PowerFlowError(f"P_spec must have shape (B, {spec.n}, 3), got {P.shape}")
        Q = None if Q_spec is None else np.ascontiguousarray(Q_spec, dtype=np.float64)
        if Q is not None and Q.shape != P.shape:
            raise PowerFlowError("Q_spec shape differs from P_spec")
        h = self._handle(spec, P.shape[0])
        out, view = self._buffers()
        self._check(_lib3().gs3_solve(h, P.ctypes.data_as(_dp), None if Q is None else Q.ctypes.data_as(_dp), C.byref(view)))
        return self._wrap(out)

    # device-resident variant for measurement
    def upload(self, spec: UnbalancedFeederSpec, P_spec, Q_spec=None) -> None:
        P = np.ascontiguousarray(P_spec, dtype=np.float64)
        Q = None if Q_spec is None else np.ascontiguousarray(Q_spec, dtype=np.float64)
        h = self._handle(spec, P.shape[0])
        self._check(_lib3().gs3_upload_injections(h, P.ctypes.data_as(_dp), None if Q is None else Q.ctypes.data_as(_dp)))

    def solve_device(self) -> None:
        self._check(_lib3().gs3_solve_device(self._h))

    def synchronize(self) -> None:
        self._check(_lib3().gs3_synchronize(self._h))

    def download(self) -> UnbalancedSolution:
        out, view = self._buffers()
        self._check(_lib3().gs3_download_solution(self._h, C.byref(view)))
        return self._wrap(out)

    def timing_read(self):
        ms, cnt = C.c_double(), C.c_int64()
        self._check(_lib3().gs3_timing_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def describe(self) -> dict:
        buf = C.create_string_buffer(512)
        self._check(_lib3().gs3_describe(self._h, buf, 512))
        return json.loads(buf.value.decode())


def unbalanced_from_single_phase(fs, coupling: float = 0.0) -> UnbalancedFeederSpec:
    """Three-phase feeder with the topology of a radial single-phase FeederSpec: every line gets
    Z = z I + coupling * z (1 - I); all nodes carry all three phases.  With ``coupling = 0`` and
    balanced loads each phase reproduces the single-phase solution."""
    n = fs.n
    adj = [[] for _ in range(n)]
    for k in range(fs.m):
        a, b = int(fs.frm[k]), int(fs.to[k])
        adj[a].append((b, k)); adj[b].append((a, k))
    src = int(np.argmax(fs.bus_type == 2))
    parent = np.full(n, -1, dtype=np.int32)
    z = np.zeros((n, 3, 3), dtype=complex)
    seen, order = {src}, [src]
    for u in order:
        for v, k in adj[u]:
            if v not in seen:
                seen.add(v); parent[v] = u; order.append(v)
                zz = complex(fs.r[k], fs.x[k])
                z[v] = zz * np.eye(3) + coupling * zz * (1 - np.eye(3))
    if len(order) != n:
        raise ValueError("feeder is not connected")
    return UnbalancedFeederSpec(name=fs.name + "_3ph", parent=parent, phases=np.full(n, 7, dtype=np.uint8), z=z, source=src)


def ieee8500_like(n: int = 8500, seed: int = 8500, lateral_probability: float = 0.3):
    """Seeded radial three-phase feeder in the spirit of the IEEE 8500-node case:
    node b > 0 hangs off a uniformly drawn lower-numbered node; Z = Zs I + 0.3 Zs (1 - I) per line;
    30 % of the nodes whose parent is three-phase start a one- or two-phase lateral (descendants
    inherit); loads per present phase U(1, 15) kW on a 100 MVA base.  Returns (spec, P[n, 3], Q[n, 3])."""
    rng = np.random.default_rng(seed)
    parent = np.full(n, -1, dtype=np.int32)
    phases = np.full(n, 7, dtype=np.uint8)
    z = np.zeros((n, 3, 3), dtype=complex)
    subsets = [1, 2, 4, 3, 5, 6]
    for b in range(1, n):
        p = int(rng.integers(max(0, b - 400), b))        # local attachment keeps the tree deep like a real feeder
        parent[b] = p
        m = int(phases[p])
        if m == 7 and rng.random() < lateral_probability:
            m = subsets[int(rng.integers(0, 6))]
        elif m in (3, 5, 6) and rng.random() < 0.2:
            m = [q for q in (1, 2, 4) if q & m][int(rng.integers(0, 2))]
        phases[b] = m
        zs = complex(rng.uniform(0.0045, 0.009), rng.uniform(0.009, 0.018))   # gives V_min ~ 0.95 pu at nominal load
        z[b] = zs * np.eye(3) + 0.3 * zs * (1 - np.eye(3))
    kw = rng.uniform(1.0, 15.0, (n, 3))
    present = ((phases[:, None] >> np.arange(3)[None, :]) & 1).astype(bool)
    P = np.where(present, -kw * 1e3 / 100e6, 0.0)
    pf = rng.uniform(0.9, 0.98, (n, 3))
    Q = P * np.tan(np.arccos(pf))
    P[0] = 0.0; Q[0] = 0.0
    return UnbalancedFeederSpec(name=f"ieee8500_like_{n}_seed{seed}", parent=parent, phases=phases, z=z), P, Q
