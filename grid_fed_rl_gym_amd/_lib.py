"""ctypes binding of libgridstep.so (include/gridstep.h).

This is the only place Python touches the device path.  There is no fallback of any kind:
if the shared library has not been built (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C grid_fed_rl_gym_amd/csrc``) loading raises, and if no GPU is visible
``gs_create`` fails with GS_E_NO_DEVICE and ``Handle`` raises PowerFlowError.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
from typing import Optional

import numpy as np

from .components import PowerFlowError
from .feeders import FeederSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgridstep.so")

GS_ABI_VERSION = 1
GS_OK, GS_E_INVALID, GS_E_NO_DEVICE, GS_E_HIP, GS_E_TOPOLOGY, GS_E_STATE, GS_E_COMM, GS_E_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7
JACOBIAN = {"as_coded": 0, "exact": 1}
ZERO_Z = {"open": 0, "epsilon": 1}
SOLVER = {"nr": 0, "newton_raphson": 0, "fbs": 1}
LINSOLVE = {"auto": 0, "tree": 1, "sparse_lu": 2, "dense_pivot": 3, "dense_mfma": 4, "sparse_lds": 5}
KERNEL_NAMES = ["unpack", "env_pre", "solve", "env_post", "pack"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)


class gs_topology(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("from_bus", _ip), ("to_bus", _ip), ("r", _dp), ("x", _dp), ("rating", _dp),
                ("bus_type", _up), ("v_set", _dp),
                ("n_loads", C.c_int32), ("load_bus", _ip), ("load_base", _dp), ("load_pf", _dp),
                ("n_gens", C.c_int32), ("gen_bus", _ip), ("gen_kind", _ip), ("gen_cap", _dp),
                ("gen_p0", _dp), ("gen_p1", _dp), ("gen_p2", _dp),
                ("n_bats", C.c_int32), ("bat_bus", _ip), ("bat_cap", _dp), ("bat_rating", _dp), ("bat_eff", _dp)]


class gs_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("solver_kind", C.c_int32), ("jacobian_mode", C.c_int32),
                ("zero_z_mode", C.c_int32), ("linear_solver", C.c_int32), ("max_iterations", C.c_int32),
                ("episode_length", C.c_int32), ("stochastic_loads", C.c_int32), ("weather_variation", C.c_int32),
                ("waves_per_group", C.c_int32), ("fbs_warm_start", C.c_int32),
                ("tolerance", C.c_double), ("acceleration_factor", C.c_double), ("timestep", C.c_double),
                ("v_min", C.c_double), ("v_max", C.c_double), ("f_min", C.c_double), ("f_max", C.c_double),
                ("safety_penalty", C.c_double), ("inertia_H", C.c_double), ("damping_D", C.c_double),
                ("f_nominal", C.c_double), ("power_base", C.c_double)]


class gs_solution_view(C.Structure):
    _fields_ = [("bus_voltages", _dp), ("bus_angles", _dp), ("line_flows", _dp), ("line_loadings", _dp),
                ("losses", _dp), ("max_mismatch", _dp), ("iterations", _ip), ("converged", _up), ("status", _ip)]


class gs_info_view(C.Structure):
    _fields_ = [("power_flow_converged", _up), ("max_voltage", _dp), ("min_voltage", _dp), ("total_losses", _dp),
                ("violations", _up), ("constraint_violations", _ip), ("current_step", _ip),
                ("episode_reward", _dp), ("iterations", _ip), ("status", _ip)]


class gs_rollout_view(C.Structure):
    _fields_ = [("observations", _dp), ("actions", _dp), ("rewards", _dp), ("next_observations", _dp), ("terminals", _up),
                ("final_observation", _dp), ("n_terminal", _ip)]


class gs_rollout_device(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("obs_dim", C.c_int32), ("action_dim", C.c_int32),
                ("obs_seq", C.c_void_p), ("actions", C.c_void_p), ("rewards", C.c_void_p), ("terminals", C.c_void_p),
                ("n_terminal", C.c_int32), ("reserved", C.c_int32), ("terminal_index", C.c_void_p), ("terminal_obs", C.c_void_p)]


POLICY = {"uploaded": 0, "random": 1}

# every symbol include/gridstep.h declares: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("gs_version", C.c_int, []),
    ("gs_build_experiments", C.c_int, []),
    ("gs_device_count", C.c_int, []),
    ("gs_last_error", C.c_char_p, [_H]),
    ("gs_create", C.c_int, [C.POINTER(gs_topology), C.POINTER(gs_config), C.c_int32, C.c_int32, C.c_int64, C.POINTER(_H)]),
    ("gs_destroy", None, [_H]),
    ("gs_dims", C.c_int, [_H, _ip, _ip, _ip, _ip, _ip, _ip]),
    ("gs_describe", C.c_int, [_H, C.c_char_p, C.c_int32]),
    ("gs_synchronize", C.c_int, [_H]),
    ("gs_solve", C.c_int, [_H, _dp, _dp, C.POINTER(gs_solution_view)]),
    ("gs_upload_injections", C.c_int, [_H, _dp, _dp]),
    ("gs_solve_device", C.c_int, [_H]),
    ("gs_download_solution", C.c_int, [_H, C.POINTER(gs_solution_view)]),
    ("gs_reset", C.c_int, [_H, C.POINTER(C.c_uint64), _up, _dp]),
    ("gs_step", C.c_int, [_H, _dp, _dp, _dp, _up, _up, C.POINTER(gs_info_view)]),
    ("gs_upload_actions", C.c_int, [_H, _dp, C.c_int32]),
    ("gs_step_device", C.c_int, [_H, C.c_int32]),
    ("gs_download_step", C.c_int, [_H, _dp, _dp, _up, _up, C.POINTER(gs_info_view)]),
    ("gs_step_f32", C.c_int, [_H, _dp, C.POINTER(C.c_float), _dp, _up, _up, C.POINTER(gs_info_view)]),
    ("gs_download_step_f32", C.c_int, [_H, C.POINTER(C.c_float), _dp, _up, _up, C.POINTER(gs_info_view)]),
    ("gs_rollout", C.c_int, [_H, C.c_int32, C.c_int32, C.c_uint64, _dp]),
    ("gs_rollout_download", C.c_int, [_H, C.POINTER(gs_rollout_view)]),
    ("gs_rollout_device_view", C.c_int, [_H, C.POINTER(gs_rollout_device)]),
    ("gs_get_state", C.c_int, [_H, _dp]),
    ("gs_set_state", C.c_int, [_H, _dp]),
    ("gs_comm_unique_id", C.c_int, [_up]),
    ("gs_comm_init", C.c_int, [_H, _up, C.c_int32, C.c_int32]),
    ("gs_allgather_obs", C.c_int, [_H, _dp]),
    ("gs_comm_destroy", C.c_int, [_H]),
    ("gs_comm_info", C.c_int, [_H, C.c_void_p]),
    ("gs_host_obs_bind", C.c_int, [_H, _dp]),
    ("gs_host_obs_unbind", C.c_int, [_H, _dp]),
    ("gs_flat_newton_map_dump", C.c_int, [C.c_void_p, C.c_int32, _dp]),
    ("gs_mesh_schedule_dump_packed", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip, _ip, _ip, _dp, _ip]),
    ("gs_mesh_schedule_dump", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip, C.c_char_p, C.c_int32,
                                        C.c_void_p, _ip, _ip, _dp]),
    ("gs_comm_init_loopback", C.c_int, [C.POINTER(_H), C.c_int32]),
    ("gs_allgather_obs_shards", C.c_int, [C.POINTER(_H), C.c_int32, _dp]),
    ("gs_allgather_obs_view", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("gs_allgather_obs_download", C.c_int, [_H, _dp]),
    ("gs_timing_enable", C.c_int, [_H, C.c_int32]),
    ("gs_timing_read", C.c_int, [_H, _dp, C.POINTER(C.c_int64)]),
    ("gs_step_device_ptr", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("gs_step_device_view", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("gs_host_alloc", C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    ("gs_host_free", C.c_int, [C.c_void_p]),
    ("gs_debug_stamps", C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    ("gs_debug_block_times", C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    ("gs_debug_write_rows", C.c_int, [_H, C.c_int32, _dp]),
    ("gs_debug_read_rows", C.c_int, [_H, C.c_int32, _dp]),
    ("gs_fallback_linear", C.c_int, [_H, _dp, _dp, _dp, _dp, _up, _up, C.POINTER(C.c_int32)]),
]
# the gs3_* entry points (three-phase solver) are bound in unbalanced.py


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libgridstep.so and bind every declared symbol; raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PowerFlowError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build() or `make -C grid_fed_rl_gym_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gs_version() != GS_ABI_VERSION:
        raise PowerFlowError(f"libgridstep ABI {lib.gs_version()} != binding {GS_ABI_VERSION}")
    _lib = lib
    return lib


def experiments() -> bool:
    """True if libgridstep.so was built with ``make EXPERIMENTS=1`` (gs_build_experiments)."""
    return bool(load().gs_build_experiments())


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


def make_config(**kw) -> gs_config:
    """gs_config with the reference's constructor defaults (power_flow.py:79-84,
    grid_env.py:161-173, dynamics.py:233-238)."""
    cfg = gs_config()
    cfg.struct_size = C.sizeof(gs_config)
    d = dict(solver_kind=0, jacobian_mode=0, zero_z_mode=0, linear_solver=0, max_iterations=50,
             episode_length=86400, stochastic_loads=0, weather_variation=0, waves_per_group=0, fbs_warm_start=0,
             tolerance=1e-6, acceleration_factor=1.0, timestep=1.0, v_min=0.95, v_max=1.05, f_min=59.5,
             f_max=60.5, safety_penalty=100.0, inertia_H=5.0, damping_D=1.0, f_nominal=60.0, power_base=1.0)
    for k, v in kw.items():
        if k not in d:
            raise TypeError(f"unknown config field {k!r}")
        d[k] = v
    for k, v in d.items():
        setattr(cfg, k, v)
    return cfg


class gs_comm_info_t(C.Structure):
    _fields_ = [("transport", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32),
                ("comm_device", C.c_int32), ("rccl_version", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32),
                ("device_uuid", C.c_uint8 * 16)]


class gs_gathered_obs(C.Structure):
    _fields_ = [("observations", C.c_void_p), ("rows", C.c_int64), ("obs_dim", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("reserved", C.c_int32)]


class gs_step_device_out(C.Structure):
    _fields_ = [("observations", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
                ("B", C.c_int32), ("obs_dim", C.c_int32)]


class DeviceArray:
    """A view of device memory owned by a handle: address, shape, dtype, exposed through ``__cuda_array_interface__`` (which
    PyTorch-ROCm and CuPy read as well) so that a consumer on the GPU wraps it without a copy."""

    def __init__(self, ptr: int, shape, typestr: str) -> None:
        self.ptr, self.shape, self.typestr = int(ptr or 0), tuple(int(x) for x in shape), typestr

    @property
    def __cuda_array_interface__(self) -> dict:
        return {"shape": self.shape, "typestr": self.typestr, "data": (self.ptr, False), "version": 3, "strides": None}

    def __repr__(self) -> str:
        return f"DeviceArray(0x{self.ptr:x}, shape={self.shape}, typestr={self.typestr!r})"


def _stream_arg(stream):
    """A consumer's / producer's ``hipStream_t`` for the C ABI: ``None`` = no stream (the caller synchronises itself); an
    integer handle otherwise.  Handle 0 is the LEGACY DEFAULT stream -- what ``torch.cuda.current_stream().cuda_stream`` is
    unless the caller made a stream of its own -- and goes over as ``hipStreamLegacy`` (1), since NULL means "no stream" at
    the boundary.  (Round 3: 0 used to be taken for "no stream", which left a step free to read its actions before the
    default stream had written them.)"""
    if stream is None:
        return None
    return C.c_void_p(int(stream) if int(stream) != 0 else 1)


def _device_address(x, shape) -> int:
    """Device address of a float64 C-contiguous array of ``shape`` given as an int, a torch tensor or anything with
    ``__cuda_array_interface__``."""
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):                      # torch
        if tuple(x.shape) != tuple(shape) or str(x.dtype) not in ("torch.float64",) or not x.is_contiguous():
            raise PowerFlowError(f"device actions must be a contiguous float64 tensor of shape {tuple(shape)}, got {tuple(x.shape)} {x.dtype}")
        return int(x.data_ptr())
    cai = getattr(x, "__cuda_array_interface__", None)
    if cai is None:
        raise PowerFlowError("device actions: an address, a torch tensor or an object with __cuda_array_interface__")
    if tuple(cai["shape"]) != tuple(shape) or cai["typestr"] not in ("<f8", "=f8") or cai.get("strides") not in (None,):
        raise PowerFlowError(f"device actions must be C-contiguous float64 of shape {tuple(shape)}")
    return int(cai["data"][0])


def _topology_of(spec: FeederSpec):
    """gs_topology of a FeederSpec and the arrays it points into (keep them alive while the struct is in use)."""
    keep = dict(frm=_i32(spec.frm), to=_i32(spec.to), r=_f64(spec.r), x=_f64(spec.x), rating=_f64(spec.rating),
                bus_type=np.ascontiguousarray(spec.bus_type, dtype=np.uint8), v_set=_f64(spec.v_set),
                load_bus=_i32(spec.load_bus), load_base=_f64(spec.load_base), load_pf=_f64(spec.load_pf),
                gen_bus=_i32(spec.gen_bus), gen_kind=_i32(spec.gen_kind), gen_cap=_f64(spec.gen_cap),
                gen_p0=_f64(spec.gen_p0), gen_p1=_f64(spec.gen_p1), gen_p2=_f64(spec.gen_p2),
                bat_bus=_i32(spec.bat_bus), bat_cap=_f64(spec.bat_cap), bat_rating=_f64(spec.bat_rating),
                bat_eff=_f64(spec.bat_eff))
    t = gs_topology()
    t.struct_size = C.sizeof(gs_topology)
    t.n, t.m = spec.n, spec.m
    t.from_bus, t.to_bus = _ptr(keep["frm"], _ip), _ptr(keep["to"], _ip)
    t.r, t.x, t.rating = _ptr(keep["r"], _dp), _ptr(keep["x"], _dp), _ptr(keep["rating"], _dp)
    t.bus_type, t.v_set = _ptr(keep["bus_type"], _up), _ptr(keep["v_set"], _dp)
    t.n_loads = spec.n_loads
    t.load_bus, t.load_base, t.load_pf = _ptr(keep["load_bus"], _ip), _ptr(keep["load_base"], _dp), _ptr(keep["load_pf"], _dp)
    t.n_gens = spec.n_gens
    t.gen_bus, t.gen_kind, t.gen_cap = _ptr(keep["gen_bus"], _ip), _ptr(keep["gen_kind"], _ip), _ptr(keep["gen_cap"], _dp)
    t.gen_p0, t.gen_p1, t.gen_p2 = _ptr(keep["gen_p0"], _dp), _ptr(keep["gen_p1"], _dp), _ptr(keep["gen_p2"], _dp)
    t.n_bats = spec.n_bats
    t.bat_bus, t.bat_cap = _ptr(keep["bat_bus"], _ip), _ptr(keep["bat_cap"], _dp)
    t.bat_rating, t.bat_eff = _ptr(keep["bat_rating"], _dp), _ptr(keep["bat_eff"], _dp)
    return t, keep


MESH_ITEM_DTYPE = np.dtype([("vk_off", "<i4"), ("vj_off", "<i4"), ("xk_off", "<i4"), ("xj_off", "<i4"), ("flags", "<i4"), ("cq_off", "<i4"),
                            ("adj_ptr", "<i4"), ("bus", "<i4"), ("ykj_g", "<f8"), ("ykj_b", "<f8"), ("ykk_g", "<f8"), ("ykk_b", "<f8"),
                            ("mout", "<i4", (8,)), ("cq_in", "<i4", (4,)), ("rw_in", "<i4", (4,)), ("cl_in", "<i4", (4,)),
                            ("nbr", "<i4"), ("pos", "<i4"), ("pad", "<i4", (10,))])      # GsMeshItem, csrc/gs_internal.h


def flat_newton_map(spec, zero_z="open") -> np.ndarray:
    """gs_flat_newton_map_dump: W [2 (n - 1), n] with x = W [P_spec (non-slack buses, bus order); 1] -- the first Newton step from the flat
    start as the meshed step kernel takes it (host arithmetic, no GPU)."""
    lib = load()
    t, keep = _topology_of(spec)
    out = np.zeros((2 * (spec.n - 1), spec.n))
    rc = lib.gs_flat_newton_map_dump(C.byref(t), ZERO_Z[zero_z], _ptr(out, _dp))
    if rc != GS_OK:
        raise PowerFlowError(f"gs_flat_newton_map_dump failed ({rc}): {lib.gs_last_error(None).decode()}")
    return out


def mesh_schedule(spec: FeederSpec, nw: int = 4, ni: int = 10, acc_cap: int = 4, region_base: int = 0, slot_bytes: int = 144,
                  zero_z: str = "open", unit_budget: int = 0) -> dict:
    """gs_mesh_schedule_dump: the host-side schedule of the meshed Newton-Raphson step kernel for ``spec`` (no device needed).
    Returns the header fields, ``why`` (when not eligible) and the tables as NumPy arrays (items: MESH_ITEM_DTYPE)."""
    lib = load()
    t, keep = _topology_of(spec)
    hd = np.zeros(16, dtype=np.int32)
    why = C.create_string_buffer(256)
    args = (C.byref(t), ZERO_Z[zero_z], int(nw), int(ni), int(acc_cap), int(unit_budget), int(region_base), int(slot_bytes))
    rc = lib.gs_mesh_schedule_dump(*args, _ptr(hd, _ip), why, 256, None, None, None, None)
    if rc != GS_OK:
        raise PowerFlowError(f"gs_mesh_schedule_dump failed ({rc}): {lib.gs_last_error(None).decode()}")
    names = ("ok", "n_levels", "n_rows", "max_rows_per_wave", "n_pivots", "msg_units", "n_messages", "n_accumulators", "max_degree",
             "unit_bytes", "zero_off", "dummy_off", "body_off", "region_bytes", "item_bytes", "n_adj")
    out = {k: int(v) for k, v in zip(names, hd)}
    out["why"] = why.value.decode(); out["nw"], out["ni"] = int(nw), int(ni)
    if not out["ok"]:
        return out
    if out["item_bytes"] != MESH_ITEM_DTYPE.itemsize:
        raise PowerFlowError(f"GsMeshItem is {out['item_bytes']} bytes in the library, {MESH_ITEM_DTYPE.itemsize} here")
    items = np.zeros(nw * ni * 8, dtype=MESH_ITEM_DTYPE)
    rowinfo = np.zeros((nw * ni, 4), dtype=np.int32)
    adj_off = np.zeros(max(out["n_adj"], 1), dtype=np.int32)
    adj_y = np.zeros((max(out["n_adj"], 1), 2))
    rc = lib.gs_mesh_schedule_dump(*args, _ptr(hd, _ip), why, 256, items.ctypes.data_as(C.c_void_p), _ptr(rowinfo, _ip),
                                   _ptr(adj_off, _ip), _ptr(adj_y, _dp))
    if rc != GS_OK:
        raise PowerFlowError(f"gs_mesh_schedule_dump failed ({rc}): {lib.gs_last_error(None).decode()}")
    out.update(items=items.reshape(nw, ni, 8), rowinfo=rowinfo.reshape(nw, ni, 4), adj_off=adj_off, adj_y=adj_y)
    # the packed form the kernel reads
    cnt = np.zeros(4, dtype=np.int32)
    rc = lib.gs_mesh_schedule_dump_packed(*args, _ptr(cnt, _ip), None, None, None, None)
    if rc != GS_OK:
        raise PowerFlowError(f"gs_mesh_schedule_dump_packed failed ({rc}): {lib.gs_last_error(None).decode()}")
    packed = np.zeros((nw * ni * 8, int(cnt[3])), dtype=np.int32)
    rowinfo_p = np.zeros((nw * ni, 4), dtype=np.int32)
    ytab = np.zeros((int(cnt[1]) // 2, 2))
    adj_ent = np.zeros(int(cnt[2]), dtype=np.int32)
    rc = lib.gs_mesh_schedule_dump_packed(*args, _ptr(cnt, _ip), _ptr(packed, _ip), _ptr(rowinfo_p, _ip), _ptr(ytab, _dp), _ptr(adj_ent, _ip))
    if rc != GS_OK:
        raise PowerFlowError(f"gs_mesh_schedule_dump_packed failed ({rc}): {lib.gs_last_error(None).decode()}")
    out.update(n_pairs=int(cnt[0]), packed=packed.reshape(nw, ni, 8, -1), rowinfo_packed=rowinfo_p.reshape(nw, ni, 4), ytab=ytab, adj_ent=adj_ent)
    return out


class Handle:
    """Owns one gs_handle (one GPU, one stream).  All array arguments are NumPy, batch-major."""

    def __init__(self, spec: FeederSpec, cfg: gs_config, batch: int, device: int = 0, first_instance: int = 0):
        self._lib = load()
        self._h = _H()
        self.spec = spec
        self.B = int(batch)
        self.obs_dtype = np.dtype(np.float64)      # np.float32: step() / download_step() hand out the block rounded on the device (gs_step_f32)
        t, keep = _topology_of(spec)
        rc = self._lib.gs_create(C.byref(t), C.byref(cfg), self.B, int(device), int(first_instance), C.byref(self._h))
        if rc != GS_OK:
            self._h = _H()
            raise PowerFlowError(f"gs_create failed ({rc}): {self._lib.gs_last_error(None).decode()}")
        dims = [C.c_int32() for _ in range(6)]
        self._check(self._lib.gs_dims(self._h, *[C.byref(d) for d in dims]))
        self.n, self.m, self.obs_dim, self.action_dim, self.state_dim, _ = [d.value for d in dims]

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != GS_OK:
            raise PowerFlowError(f"libgridstep error {rc}: {self._lib.gs_last_error(self._h).decode()}")

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            for child in list(getattr(self, "_children", [])):     # objects bound to this handle (safety.PostStepChecks) go first
                child.close()
            self._lib.gs_destroy(self._h)
            self._h = _H()
            self._free_pinned()

    def last_error(self) -> str:
        return self._lib.gs_last_error(self._h).decode()

    def _adopt(self, child) -> None:
        if not hasattr(self, "_children"):
            self._children = []
        self._children.append(child)

    def _release(self, child) -> None:
        if child in getattr(self, "_children", []):
            self._children.remove(child)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self) -> dict:
        buf = C.create_string_buffer(4096)
        self._check(self._lib.gs_describe(self._h, buf, 4096))
        return json.loads(buf.value.decode())

    def synchronize(self) -> None:
        self._check(self._lib.gs_synchronize(self._h))

    # -- solver ---------------------------------------------------------------------------
    def _solution_buffers(self):
        B, n, m = self.B, self.n, self.m
        out = dict(bus_voltages=np.empty((B, n)), bus_angles=np.empty((B, n)), line_flows=np.empty((B, m)),
                   line_loadings=np.empty((B, m)), losses=np.empty(B), max_mismatch=np.empty(B),
                   iterations=np.empty(B, dtype=np.int32), converged=np.empty(B, dtype=np.uint8),
                   status=np.empty(B, dtype=np.int32))
        v = gs_solution_view(_ptr(out["bus_voltages"], _dp), _ptr(out["bus_angles"], _dp), _ptr(out["line_flows"], _dp),
                             _ptr(out["line_loadings"], _dp), _ptr(out["losses"], _dp), _ptr(out["max_mismatch"], _dp),
                             _ptr(out["iterations"], _ip), _ptr(out["converged"], _up), _ptr(out["status"], _ip))
        return out, v

    def _pq(self, P, Q):
        P = _f64(P)
        if P.shape != (self.B, self.n):
            raise PowerFlowError(f"P_spec shape {P.shape} != ({self.B}, {self.n})")
        if Q is not None:
            Q = _f64(Q)
            if Q.shape != P.shape:
                raise PowerFlowError(f"Q_spec shape {Q.shape} != {P.shape}")
        return P, Q

    def solve(self, P, Q=None) -> dict:
        P, Q = self._pq(P, Q)
        out, v = self._solution_buffers()
        self._check(self._lib.gs_solve(self._h, _ptr(P, _dp), _ptr(Q, _dp), C.byref(v)))
        return out

    def fallback_linear(self, load_w=None, gen_w=None, total_load=None, total_gen=None, mask=None) -> np.ndarray:
        """gs_fallback_linear: replace the solution of the masked (default: non-converged) instances by the reference's
        linear approximation (robust_power_flow.py:336-398); returns the boolean [B] array of replaced instances.
        Without load_w / gen_w the per-bus totals come from the environment state on the device."""
        def arr(a, shape):
            if a is None:
                return None
            a = _f64(a)
            if a.shape != shape:
                raise PowerFlowError(f"expected shape {shape}, got {a.shape}")
            return a
        lw, gw = arr(load_w, (self.B, self.n)), arr(gen_w, (self.B, self.n))
        tl, tg = arr(total_load, (self.B,)), arr(total_gen, (self.B,))
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8).reshape(self.B))
        applied = np.zeros(self.B, dtype=np.uint8)
        cnt = C.c_int32(0)
        self._check(self._lib.gs_fallback_linear(self._h, _ptr(lw, _dp), _ptr(gw, _dp), _ptr(tl, _dp), _ptr(tg, _dp), _ptr(mk, _up),
                                                 _ptr(applied, _up), C.byref(cnt)))
        return applied.astype(bool)

    def upload_injections(self, P, Q=None) -> None:
        P, Q = self._pq(P, Q)
        self._check(self._lib.gs_upload_injections(self._h, _ptr(P, _dp), _ptr(Q, _dp)))

    def solve_device(self) -> None:
        self._check(self._lib.gs_solve_device(self._h))

    def download_solution(self) -> dict:
        out, v = self._solution_buffers()
        self._check(self._lib.gs_download_solution(self._h, C.byref(v)))
        return out

    # -- env ------------------------------------------------------------------------------
    def reset(self, seeds=None, mask=None, want_obs: bool = True):
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        k = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if s is not None and s.shape != (self.B,):
            raise PowerFlowError(f"seeds shape {s.shape} != ({self.B},)")
        if k is not None and k.shape != (self.B,):
            raise PowerFlowError(f"mask shape {k.shape} != ({self.B},)")
        obs = np.empty((self.B, self.obs_dim)) if want_obs else None
        self._check(self._lib.gs_reset(self._h, _ptr(s, C.POINTER(C.c_uint64)), _ptr(k, _up), _ptr(obs, _dp)))
        return obs

    # -- page-locked output buffers (opt-in) ----------------------------------------------------
    def use_pinned_outputs(self, sets: int = 2) -> None:
        """From now on step() / download_step() return arrays that live in ``sets`` rotating sets of page-locked host
        buffers (gs_host_alloc): the device-to-host copy runs at the link's rate and no 45 MB array is page-faulted in
        per step.  The price is the reference's "fresh arrays every step": what step() returned is overwritten ``sets``
        steps later -- copy what has to live longer (a replay buffer does that anyway)."""
        if getattr(self, "_pinned", None):
            return
        self._pinned, self._pinned_ptrs, self._pinned_turn = [], [], 0
        for _ in range(max(2, int(sets))):
            self._pinned.append(self._alloc_step_set(self._pinned_array))

    def _pinned_array(self, shape, dtype=np.float64):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        rc = self._lib.gs_host_alloc(C.byref(p), max(n, 8))
        if rc != GS_OK:
            raise PowerFlowError(f"gs_host_alloc({n}) failed ({rc}): {self._lib.gs_last_error(None).decode()}")
        self._pinned_ptrs.append(p)
        return np.frombuffer((C.c_char * max(n, 8)).from_address(p.value), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def _alloc_step_set(self, new, want_obs=True):
        B = self.B
        return dict(obs=new((B, self.obs_dim), self.obs_dtype) if want_obs else None, reward=new((B,)),
                    terminated=new((B,), np.uint8), truncated=new((B,), np.uint8),
                    power_flow_converged=new((B,), np.uint8), max_voltage=new((B,)), min_voltage=new((B,)),
                    total_losses=new((B,)), violations=new((B, 4), np.uint8),
                    constraint_violations=new((B,), np.int32), current_step=new((B,), np.int32),
                    episode_reward=new((B,)), iterations=new((B,), np.int32), status=new((B,), np.int32))

    def _free_pinned(self) -> None:
        self._pinned = None
        rec = getattr(self, "_recycle", None)
        if rec and getattr(self, "_h", None) and self._h.value:
            for st in rec["sets"]:
                self._unbind_obs(st)
        self._recycle = None
        keep = set()
        if rec:       # a set the caller still holds arrays of is not freed: those arrays stay valid (the memory goes with the process)
            for st in rec["sets"]:
                if not self._set_is_free(st):
                    keep.update(st["ptrs"])
        for p in getattr(self, "_pinned_ptrs", []):
            if p.value not in keep:
                self._lib.gs_host_free(p)
        self._pinned_ptrs = []

    # -- recycled output buffers (what BatchedGridEnvironment uses by default) -------------------------
    def use_recycled_outputs(self, max_sets: int = 4) -> None:
        """step() / download_step() return arrays that are views of page-locked buffer sets owned by the handle, and a set is
        taken again only once NOTHING the caller got from it is alive any more (the reference counts of its root arrays: every
        view, slice or reshape of a returned array holds one).  So the contract stays the reference's -- what step() returned is
        never overwritten behind the caller's back -- while the usual loop (consume the observation, drop it, step again) neither
        page-faults a fresh 45 MB array per step nor copies through pageable memory: 1.9 M -> ~7 M env-steps/s at B = 8192.
        A caller that keeps everything makes the pool grow to ``max_sets`` sets and then gets fresh pageable arrays as before."""
        if getattr(self, "_recycle", None) or getattr(self, "_pinned", None):
            return
        if not hasattr(self, "_pinned_ptrs"):
            self._pinned_ptrs = []
        self._recycle = {"sets": [], "max": max(1, int(max_sets))}

    @staticmethod
    def _anchor_counts(st):
        return [sys.getrefcount(a) for a in st["anchors"]]

    def _set_is_free(self, st) -> bool:
        # (NumPy collapses the base of a view of a view to the first array of the chain: the anchors are those first arrays --
        # what every view, slice and reshape of a returned array keeps alive; counted the same way as the baseline was)
        return all(c <= b for c, b in zip(self._anchor_counts(st), st["base"]))

    def _recycled_set(self, want_obs):
        rec = self._recycle
        for st in rec["sets"]:
            if st["want_obs"] == want_obs and self._set_is_free(st):
                return st
        if len(rec["sets"]) >= rec["max"]:
            return None
        # Only the observation block is page-locked and recycled: it is the 45 MB whose page faults and staged copy the sets exist to
        # avoid.  The dozen per-instance arrays (half a megabyte together) are fresh pageable arrays every step, so a caller that keeps an
        # `info` array does not keep a 45 MB block from being taken again -- the pool's bound is max_sets x B x obs_dim x 8 bytes of
        # page-locked memory, reached only by a caller that keeps max_sets observation arrays alive.
        before = len(self._pinned_ptrs)
        try:
            roots = {"obs": self._pinned_array((self.B, self.obs_dim), self.obs_dtype)}
        except PowerFlowError:                        # no page-locked memory to be had: an ordinary array, still reused
            for p in self._pinned_ptrs[before:]:
                self._lib.gs_host_free(p)
            del self._pinned_ptrs[before:]
            roots = {"obs": np.empty((self.B, self.obs_dim), dtype=self.obs_dtype)}
        st = {"roots": roots, "want_obs": want_obs, "ptrs": {p.value for p in self._pinned_ptrs[before:]}, "anchors": [], "base": []}
        def first_array(a):
            while isinstance(a.base, np.ndarray):
                a = a.base
            return a
        st["anchors"] = [first_array(r) for r in roots.values() if r is not None]
        st["base"] = self._anchor_counts(st)                                      # no views alive yet
        rec["sets"].append(st)
        return st

    # -- constant observation columns of a recycled / pinned set: written once (gs_host_obs_bind), not moved again ------------
    def _const_block(self):
        s = self.spec
        c0 = 2 * s.n + 2 * s.m + 1
        return c0, c0 + 2 * s.n_loads

    def _bind_obs(self, st) -> None:
        """Called once a set's observation array holds a whole observation: from now on downloads into it move the changing
        columns only.  A caller that edits returned arrays IN PLACE would spoil the constant columns for the set's next use: a
        sample of them is checked before every reuse (``_obs_intact``), and a spoilt set is bound again."""
        obs = st["roots"].get("obs") if st else None
        c0, c1 = self._const_block()
        if obs is None or c1 <= c0 or obs.dtype != np.float64:      # (a float32 block is converted and copied whole)
            return
        if self._lib.gs_host_obs_bind(self._h, _ptr(obs, _dp)) != GS_OK:
            return
        rng = np.random.default_rng(12345)
        k = min(128, obs.shape[0] * (c1 - c0))
        rows, cols = rng.integers(0, obs.shape[0], k), rng.integers(c0, c1, k)
        st["probe"] = (rows, cols, obs[rows, cols].copy())

    def _obs_intact(self, st) -> bool:
        pr = st.get("probe")
        if pr is None:
            return True
        rows, cols, vals = pr
        return bool(np.array_equal(st["roots"]["obs"][rows, cols], vals))

    def _unbind_obs(self, st) -> None:
        if st.pop("probe", None) is not None:
            self._lib.gs_host_obs_unbind(self._h, _ptr(st["roots"]["obs"], _dp))

    def _step_buffers(self, want_obs=True):
        st = self._recycled_set(want_obs) if (getattr(self, "_recycle", None) and want_obs) else None
        self._cur_set = st
        if st is not None and not self._obs_intact(st):
            self._unbind_obs(st)                   # (this download writes whole rows again; bound anew behind it)
        if getattr(self, "_pinned", None):
            out = dict(self._pinned[self._pinned_turn])
            self._pinned_turn = (self._pinned_turn + 1) % len(self._pinned)
            if not want_obs:
                out["obs"] = None
        elif st is not None:
            out = self._alloc_step_set(lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype), False)
            out["obs"] = st["roots"]["obs"].view()
        else:
            out = self._alloc_step_set(lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype), want_obs)
        info = gs_info_view(_ptr(out["power_flow_converged"], _up), _ptr(out["max_voltage"], _dp),
                            _ptr(out["min_voltage"], _dp), _ptr(out["total_losses"], _dp), _ptr(out["violations"], _up),
                            _ptr(out["constraint_violations"], _ip), _ptr(out["current_step"], _ip),
                            _ptr(out["episode_reward"], _dp), _ptr(out["iterations"], _ip), _ptr(out["status"], _ip))
        return out, info

    def step(self, actions) -> dict:
        a = _f64(actions)
        if a.shape != (self.B, self.action_dim):
            raise PowerFlowError(f"actions shape {a.shape} != ({self.B}, {self.action_dim})")
        out, info = self._step_buffers()
        if self.obs_dtype == np.float32:
            self._check(self._lib.gs_step_f32(self._h, _ptr(a, _dp), _ptr(out["obs"], C.POINTER(C.c_float)), _ptr(out["reward"], _dp),
                                              _ptr(out["terminated"], _up), _ptr(out["truncated"], _up), C.byref(info)))
            return out
        self._check(self._lib.gs_step(self._h, _ptr(a, _dp), _ptr(out["obs"], _dp), _ptr(out["reward"], _dp),
                                      _ptr(out["terminated"], _up), _ptr(out["truncated"], _up), C.byref(info)))
        self._after_download()
        return out

    def _after_download(self) -> None:
        st = getattr(self, "_cur_set", None)
        if st is not None and st["roots"].get("obs") is not None and "probe" not in st and st["ptrs"]:
            self._bind_obs(st)

    def upload_actions(self, actions) -> None:
        a = _f64(actions)
        if a.ndim != 3 or a.shape[1:] != (self.B, self.action_dim):
            raise PowerFlowError(f"actions shape {a.shape} != (K, {self.B}, {self.action_dim})")
        self._check(self._lib.gs_upload_actions(self._h, _ptr(a, _dp), a.shape[0]))

    def step_device(self, k: int) -> None:
        self._check(self._lib.gs_step_device(self._h, int(k)))

    # -- a consumer on the same GPU: device pointers in, device pointers out -------------------------
    def step_device_ptr(self, actions, stream=None) -> None:
        """gs_step_device_ptr: one step with ``actions`` [B, action_dim] float64 in DEVICE memory -- an integer address, or
        any object with ``__cuda_array_interface__`` / ``data_ptr()`` (a torch tensor on this GPU).  ``stream``: the
        producer's ``hipStream_t`` as an integer (torch: ``torch.cuda.current_stream().cuda_stream``); the step then waits
        on the device for what is queued there.  None: the caller has synchronised."""
        self._check(self._lib.gs_step_device_ptr(self._h, C.c_void_p(_device_address(actions, (self.B, self.action_dim))),
                                                 _stream_arg(stream)))

    def step_device_view(self, stream=None) -> dict:
        """gs_step_device_view: the last step's observation block, rewards and flags as ``DeviceArray`` objects (zero-copy:
        ``torch.as_tensor(x, device="cuda")`` / ``cupy.asarray(x)``).  ``stream``: the consumer's stream, made to wait on the
        device for the step; None: the call returns when the step has finished.  The observation block is one of the handle's
        two buffers (valid until the next-but-one step), the other arrays are refreshed by every call."""
        out = gs_step_device_out()
        self._check(self._lib.gs_step_device_view(self._h, C.byref(out), _stream_arg(stream)))
        B = int(out.B)
        return dict(obs=DeviceArray(out.observations, (B, int(out.obs_dim)), "<f8"), reward=DeviceArray(out.reward, (B,), "<f8"),
                    terminated=DeviceArray(out.terminated, (B,), "|u1"), truncated=DeviceArray(out.truncated, (B,), "|u1"))

    def download_step(self, want_obs: bool = True) -> dict:
        out, info = self._step_buffers(want_obs)
        if want_obs and self.obs_dtype == np.float32:
            self._check(self._lib.gs_download_step_f32(self._h, _ptr(out["obs"], C.POINTER(C.c_float)), _ptr(out["reward"], _dp),
                                                       _ptr(out["terminated"], _up), _ptr(out["truncated"], _up), C.byref(info)))
            return out
        self._check(self._lib.gs_download_step(self._h, _ptr(out["obs"], _dp), _ptr(out["reward"], _dp),
                                               _ptr(out["terminated"], _up), _ptr(out["truncated"], _up), C.byref(info)))
        if want_obs:
            self._after_download()
        return out

    # -- rollout collection ------------------------------------------------------------------
    def rollout(self, T: int, policy: str = "random", seed: int = 0, actions=None) -> None:
        """gs_rollout: T env steps back to back on the device (asynchronous).  policy "random": uniform actions in
        (-1, 1) drawn on the device from ``seed``; "uploaded": ``actions`` [T, B, A]."""
        a = None
        if policy == "uploaded":
            a = _f64(actions)
            if a.shape != (int(T), self.B, self.action_dim):
                raise PowerFlowError(f"actions shape {a.shape} != ({T}, {self.B}, {self.action_dim})")
        self._check(self._lib.gs_rollout(self._h, int(T), POLICY[policy], C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(a, _dp)))
        self._rollout_T = int(T)

    def rollout_download(self, want=("observations", "actions", "rewards", "next_observations", "terminals")) -> dict:
        """Host copies of the last rollout, [T, B, ...] each (gs_rollout_download); ``terminals`` is the raw uint8 flag
        array (bit 0 terminated, bit 1 truncated)."""
        T, B = self._rollout_T, self.B
        out = {}
        if "observations" in want: out["observations"] = np.empty((T, B, self.obs_dim))
        if "actions" in want: out["actions"] = np.empty((T, B, self.action_dim))
        if "rewards" in want: out["rewards"] = np.empty((T, B))
        if "next_observations" in want: out["next_observations"] = np.empty((T, B, self.obs_dim))
        if "terminals" in want: out["terminals"] = np.empty((T, B), dtype=np.uint8)
        if "final_observation" in want: out["final_observation"] = np.empty((B, self.obs_dim))
        n = C.c_int32(0)
        v = gs_rollout_view(_ptr(out.get("observations"), _dp), _ptr(out.get("actions"), _dp), _ptr(out.get("rewards"), _dp),
                            _ptr(out.get("next_observations"), _dp), _ptr(out.get("terminals"), _up),
                            _ptr(out.get("final_observation"), _dp), C.pointer(n))
        self._check(self._lib.gs_rollout_download(self._h, C.byref(v)))
        out["n_terminal"] = int(n.value)
        return out

    def rollout_device_view(self) -> gs_rollout_device:
        """Device pointers of the last rollout (gs_rollout_device_view) for a consumer that stays on the GPU."""
        v = gs_rollout_device()
        self._check(self._lib.gs_rollout_device_view(self._h, C.byref(v)))
        return v

    def rollout_device_arrays(self) -> dict:
        """The last rollout as zero-copy ``DeviceArray`` views (``torch.as_tensor(x, device="cuda")``): ``obs_seq`` [T + 1, B, obs_dim]
        (slot t = what step t started from; ``obs_seq[1:]`` are the next observations except where an episode ended),
        ``actions`` [T, B, A], ``rewards`` [T, B], ``terminals`` [T, B] uint8 (bit 0 terminated, bit 1 truncated), and for the
        transitions that ended an episode ``terminal_index`` [n, 2] int32 (t, b) with their ``terminal_obs`` [n, obs_dim].
        Valid until the next rollout on the handle."""
        v = self.rollout_device_view()
        T, B, D, A, n = int(v.T), int(v.B), int(v.obs_dim), int(v.action_dim), int(v.n_terminal)
        return dict(obs_seq=DeviceArray(v.obs_seq, (T + 1, B, D), "<f8"), actions=DeviceArray(v.actions, (T, B, A), "<f8"),
                    rewards=DeviceArray(v.rewards, (T, B), "<f8"), terminals=DeviceArray(v.terminals, (T, B), "|u1"),
                    terminal_index=DeviceArray(v.terminal_index, (n, 2), "<i4"), terminal_obs=DeviceArray(v.terminal_obs, (n, D), "<f8"))

    def get_state(self) -> np.ndarray:
        st = np.empty((self.B, self.state_dim))
        self._check(self._lib.gs_get_state(self._h, _ptr(st, _dp)))
        return st

    def set_state(self, state) -> None:
        st = _f64(state)
        if st.shape != (self.B, self.state_dim):
            raise PowerFlowError(f"state shape {st.shape} != ({self.B}, {self.state_dim})")
        self._check(self._lib.gs_set_state(self._h, _ptr(st, _dp)))

    # -- multi-GPU ------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        lib = load()
        buf = (C.c_uint8 * 128)()
        rc = lib.gs_comm_unique_id(buf)
        if rc != GS_OK:
            raise PowerFlowError(f"gs_comm_unique_id failed ({rc}): {lib.gs_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, uid: bytes, rank: int, world: int) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self._lib.gs_comm_init(self._h, buf, int(rank), int(world)))
        self.world = int(world)

    def allgather_obs(self, to_host: bool = False):
        full = np.empty((self.world * self.B, self.obs_dim)) if to_host else None
        self._check(self._lib.gs_allgather_obs(self._h, _ptr(full, _dp)))
        return full

    def comm_destroy(self) -> None:
        self._check(self._lib.gs_comm_destroy(self._h))

    def comm_info(self) -> dict:
        """gs_comm_info: what the communicator itself reports about this member (RCCL's rank count, rank, device and version;
        -1 where the library lacks the entry point) and the HIP device's UUID."""
        v = gs_comm_info_t()
        self._check(self._lib.gs_comm_info(self._h, C.byref(v)))
        return {"transport": {1: "rccl", 2: "loopback"}.get(v.transport, "?"), "nranks": int(v.nranks), "rank": int(v.rank),
                "device": int(v.device), "comm_device": int(v.comm_device), "rccl_version": int(v.rccl_version),
                "device_uuid": bytes(v.device_uuid).hex()}

    @staticmethod
    def comm_init_loopback(handles) -> None:
        """gs_comm_init_loopback: the handles of this process (shard r built with first_instance = r * B) become ranks
        0 .. len - 1 of an in-process communicator whose all-gather moves the blocks by device-to-device copies."""
        lib = load()
        arr = (_H * len(handles))(*[h._h for h in handles])
        rc = lib.gs_comm_init_loopback(arr, len(handles))
        if rc != GS_OK:
            raise PowerFlowError(f"gs_comm_init_loopback failed ({rc}): {lib.gs_last_error(None).decode()}")
        for h in handles:
            h.world = len(handles)

    @staticmethod
    def allgather_obs_shards(handles, to_host: bool = False):
        """gs_allgather_obs_shards: one gather for every member; returns the [world * B, obs_dim] block (to_host) or None."""
        lib = load()
        arr = (_H * len(handles))(*[h._h for h in handles])
        full = np.empty((len(handles) * handles[0].B, handles[0].obs_dim)) if to_host else None
        rc = lib.gs_allgather_obs_shards(arr, len(handles), _ptr(full, _dp))
        if rc != GS_OK:
            raise PowerFlowError(f"gs_allgather_obs_shards failed ({rc}): {lib.gs_last_error(None).decode()} / {handles[0].last_error()}")
        return full

    def allgather_obs_view(self, stream=None) -> "DeviceArray":
        """gs_allgather_obs_view: this member's gathered block [world * B, obs_dim] as a zero-copy ``DeviceArray``."""
        v = gs_gathered_obs()
        self._check(self._lib.gs_allgather_obs_view(self._h, C.byref(v), _stream_arg(stream)))
        return DeviceArray(v.observations, (int(v.rows), int(v.obs_dim)), "<f8")

    def allgather_obs_download(self) -> np.ndarray:
        full = np.empty((self.world * self.B, self.obs_dim))
        self._check(self._lib.gs_allgather_obs_download(self._h, _ptr(full, _dp)))
        return full

    # -- measurement ----------------------------------------------------------------------
    def timing_enable(self, on: bool = True, span: bool = False) -> None:
        """Per-launch HIP event pairs (on), or one pair around the whole region up to the next timing_read() (span)."""
        self._check(self._lib.gs_timing_enable(self._h, (2 if span else 1) if on else 0))

    STAMP_NAMES = ["prologue_inject", "init", "mismatch", "bottom_up", "flag", "top_down", "final_mismatch", "epilogue_pack",
                   "prologue_scalars_rng", "prologue_chains", "epi_buses", "epi_lines", "epi_reduce", "epi_scalars", "nr_apply", "startup"]

    def debug_stamps(self) -> dict:
        buf = (C.c_uint64 * 16)()
        self._check(self._lib.gs_debug_stamps(self._h, buf, 16))
        return {n: int(buf[k]) for k, n in enumerate(self.STAMP_NAMES)}

    def debug_block_times(self, n_blocks: int) -> np.ndarray:
        """[n_blocks, 2] (start, end) of the workgroups of the last step launch, 100 MHz ticks (gs_debug_block_times)."""
        out = np.zeros((int(n_blocks), 2), dtype=np.uint64)
        self._check(self._lib.gs_debug_block_times(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(n_blocks)))
        return out

    ROW_FAMILIES = {"VM": 0, "LOAD": 1, "ENVLOAD": 2, "FLOW": 3, "FREQ": 4, "CONV": 5, "ITERS": 6, "MAXMIS": 7, "LOADP": 8}

    def debug_read_rows(self, name: str) -> np.ndarray:
        """Test aid (gs_debug_read_rows): one family of device rows as a [B, width] array."""
        width = {"VM": self.n, "LOAD": self.m, "ENVLOAD": self.m, "FLOW": self.m, "LOADP": self.spec.n_loads}.get(name, 1)
        out = np.empty((self.B, width))
        if width:
            self._check(self._lib.gs_debug_read_rows(self._h, self.ROW_FAMILIES[name], out.ctypes.data_as(_dp)))
        return out

    def debug_write_rows(self, rows: dict) -> None:
        """Test aid (gs_debug_write_rows): overwrite families of device rows with [B, width] arrays; None entries are skipped."""
        for name, val in rows.items():
            if val is None:
                continue
            a = np.ascontiguousarray(np.asarray(val, dtype=np.float64).reshape(self.B, -1))
            self._check(self._lib.gs_debug_write_rows(self._h, self.ROW_FAMILIES[name], a.ctypes.data_as(_dp)))

    def timing_read(self) -> dict:
        ms = (C.c_double * 5)()
        cnt = (C.c_int64 * 5)()
        self._check(self._lib.gs_timing_read(self._h, ms, cnt))
        return {KERNEL_NAMES[k]: {"total_ms": ms[k], "launches": int(cnt[k])} for k in range(5)}
