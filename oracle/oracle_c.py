"""ctypes wrapper of oracle/liboracle_cpu.so (the C/OpenMP restatement, oracle_cpu.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/ and by bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_cpu.so")
_dp, _ip, _up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class orc_net(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("frm", _ip), ("to", _ip), ("r", _dp), ("x", _dp), ("rating", _dp),
                ("bus_type", _up), ("v_set", _dp), ("n_loads", C.c_int32), ("load_bus", _ip), ("load_base", _dp),
                ("load_pf", _dp), ("n_gens", C.c_int32), ("gen_bus", _ip), ("gen_kind", _ip), ("gen_cap", _dp),
                ("gen_p0", _dp), ("gen_p1", _dp), ("gen_p2", _dp), ("n_bats", C.c_int32), ("bat_bus", _ip),
                ("bat_cap", _dp), ("bat_rating", _dp), ("bat_eff", _dp)]


class orc_cfg(C.Structure):
    _fields_ = [("solver_fbs", C.c_int32), ("jacobian_exact", C.c_int32), ("zero_z_eps", C.c_int32),
                ("max_iterations", C.c_int32), ("episode_length", C.c_int32), ("stochastic_loads", C.c_int32),
                ("weather_variation", C.c_int32), ("threads", C.c_int32), ("tolerance", C.c_double), ("alpha", C.c_double),
                ("timestep", C.c_double), ("v_min", C.c_double), ("v_max", C.c_double), ("f_min", C.c_double),
                ("f_max", C.c_double), ("safety_penalty", C.c_double), ("H", C.c_double), ("D", C.c_double),
                ("f0", C.c_double), ("power_base", C.c_double), ("first_instance", C.c_int64)]


_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.orc_state_dim.restype = C.c_int
        _lib.orc_obs_dim.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Net:
    """Keeps the arrays of a FeederSpec-like object alive next to the C struct."""

    def __init__(self, fs):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)     # noqa: E731
        self.keep = dict(frm=i32(fs.frm), to=i32(fs.to), r=f64(fs.r), x=f64(fs.x), rating=f64(fs.rating),
                         bus_type=np.ascontiguousarray(fs.bus_type, dtype=np.uint8), v_set=f64(fs.v_set),
                         load_bus=i32(fs.load_bus), load_base=f64(fs.load_base), load_pf=f64(fs.load_pf),
                         gen_bus=i32(fs.gen_bus), gen_kind=i32(fs.gen_kind), gen_cap=f64(fs.gen_cap),
                         gen_p0=f64(fs.gen_p0), gen_p1=f64(fs.gen_p1), gen_p2=f64(fs.gen_p2), bat_bus=i32(fs.bat_bus),
                         bat_cap=f64(fs.bat_cap), bat_rating=f64(fs.bat_rating), bat_eff=f64(fs.bat_eff))
        k = self.keep
        self.n, self.m = len(k["bus_type"]), len(k["frm"])
        self.c = orc_net(self.n, self.m, _p(k["frm"], _ip), _p(k["to"], _ip), _p(k["r"], _dp), _p(k["x"], _dp),
                         _p(k["rating"], _dp), _p(k["bus_type"], _up), _p(k["v_set"], _dp), len(k["load_bus"]),
                         _p(k["load_bus"], _ip), _p(k["load_base"], _dp), _p(k["load_pf"], _dp), len(k["gen_bus"]),
                         _p(k["gen_bus"], _ip), _p(k["gen_kind"], _ip), _p(k["gen_cap"], _dp), _p(k["gen_p0"], _dp),
                         _p(k["gen_p1"], _dp), _p(k["gen_p2"], _dp), len(k["bat_bus"]), _p(k["bat_bus"], _ip),
                         _p(k["bat_cap"], _dp), _p(k["bat_rating"], _dp), _p(k["bat_eff"], _dp))
        self.state_dim = lib().orc_state_dim(C.byref(self.c))
        self.obs_dim = lib().orc_obs_dim(C.byref(self.c))
        self.action_dim = len(k["bat_bus"]) + len(k["gen_bus"])


def config(solver="nr", jacobian="exact", zero_z="open", max_iterations=50, tolerance=1e-6, alpha=1.0, timestep=1.0,
           episode_length=86400, stochastic_loads=False, weather_variation=False, v_lim=(0.95, 1.05), f_lim=(59.5, 60.5),
           safety_penalty=100.0, H=5.0, D=1.0, f0=60.0, power_base=1.0, first_instance=0, threads=0) -> orc_cfg:
    return orc_cfg(int(solver == "fbs"), int(jacobian == "exact"), int(zero_z == "epsilon"), max_iterations,
                   episode_length, int(stochastic_loads), int(weather_variation), threads, tolerance, alpha, timestep,
                   v_lim[0], v_lim[1], f_lim[0], f_lim[1], safety_penalty, H, D, f0, power_base, first_instance)


def solve_batch(net: Net, cfg: orc_cfg, P, Q=None) -> dict:
    P = np.ascontiguousarray(P, dtype=np.float64)
    B = P.shape[0]
    Q = None if Q is None else np.ascontiguousarray(Q, dtype=np.float64)
    out = dict(bus_voltages=np.empty((B, net.n)), bus_angles=np.empty((B, net.n)), line_flows=np.empty((B, net.m)),
               line_loadings=np.empty((B, net.m)), losses=np.empty(B), max_mismatch=np.empty(B),
               iterations=np.empty(B, dtype=np.int32), converged=np.empty(B, dtype=np.uint8),
               status=np.empty(B, dtype=np.int32))
    rc = lib().orc_solve_batch(C.byref(net.c), C.byref(cfg), B, _p(P, _dp), _p(Q, _dp), _p(out["bus_voltages"], _dp),
                               _p(out["bus_angles"], _dp), _p(out["line_flows"], _dp), _p(out["line_loadings"], _dp),
                               _p(out["losses"], _dp), _p(out["max_mismatch"], _dp), _p(out["iterations"], _ip),
                               _p(out["converged"], _up), _p(out["status"], _ip))
    if rc != 0:
        raise RuntimeError(f"orc_solve_batch failed ({rc})")
    return out


def env_reset(net: Net, cfg: orc_cfg, B: int, seeds=None):
    state = np.zeros((B, net.state_dim))
    obs = np.empty((B, net.obs_dim))
    s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
    lib().orc_env_reset(C.byref(net.c), C.byref(cfg), B, _p(s, C.POINTER(C.c_uint64)), _p(state, _dp), _p(obs, _dp))
    return obs, state


def env_step(net: Net, cfg: orc_cfg, state, actions) -> dict:
    B = state.shape[0]
    a = np.ascontiguousarray(actions, dtype=np.float64)
    out = dict(obs=np.empty((B, net.obs_dim)), reward=np.empty(B), terminated=np.empty(B, dtype=np.uint8),
               truncated=np.empty(B, dtype=np.uint8), converged=np.empty(B, dtype=np.uint8),
               iterations=np.empty(B, dtype=np.int32), losses=np.empty(B), vmax=np.empty(B), vmin=np.empty(B),
               violations=np.empty((B, 4), dtype=np.uint8))
    rc = lib().orc_env_step(C.byref(net.c), C.byref(cfg), B, _p(a, _dp), _p(state, _dp), _p(out["obs"], _dp),
                            _p(out["reward"], _dp), _p(out["terminated"], _up), _p(out["truncated"], _up),
                            _p(out["converged"], _up), _p(out["iterations"], _ip), _p(out["losses"], _dp),
                            _p(out["vmax"], _dp), _p(out["vmin"], _dp), _p(out["violations"], _up))
    if rc != 0:
        raise RuntimeError(f"orc_env_step failed ({rc})")
    return out


def cpu_share() -> dict:
    """CPUs this process may actually use: the affinity mask, capped by the container's CPU quota (cgroup v2 ``cpu.max``, v1
    ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``).  A GPU box exposes every host thread in the mask but grants a share of them
    per GPU: a pool sized by the mask alone is time-sliced inside the quota (round 3 measured 256 threads three times SLOWER
    than 16 for exactly that reason)."""
    mask = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = mask if quota is None else max(1, min(mask, int(quota + 0.999)))
    return {"affinity_mask": mask, "cgroup_quota_cpus": quota, "usable": usable}


def bench_env_steps(fs, env_kwargs, budget_s=15.0, threads=None) -> dict:
    """cpu_baseline leg of bench.py: the C/OpenMP port on all host cores (or on `threads`), bounded sample."""
    net = Net(fs)
    # the GPU box exposes all host threads but grants a 16-core share per GPU; never oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else lib().orc_max_threads()
    share = cpu_share()
    if threads == "all":                      # every CPU this process may USE ("at 1 core and at all cores"): mask capped by the cgroup quota
        threads = max(1, share["usable"])
    threads = max(1, min(lib().orc_max_threads(), avail, 16)) if threads is None else int(threads)
    cfg = config(solver=env_kwargs["solver"], jacobian="exact", max_iterations=env_kwargs["max_iterations"],
                 tolerance=env_kwargs["tolerance"], stochastic_loads=env_kwargs["stochastic_loads"],
                 weather_variation=env_kwargs["weather_variation"], power_base=fs.base_power_va, threads=threads)
    B = 16 * threads
    rng = np.random.default_rng(5678)
    _, state = env_reset(net, cfg, B, np.arange(B, dtype=np.uint64))
    state[:, 0] = 11.5 * 3600.0
    env_step(net, cfg, state, rng.uniform(-1, 1, (B, net.action_dim)))     # warm-up
    n_done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env_step(net, cfg, state, rng.uniform(-1, 1, (B, net.action_dim)))
        n_done += B
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "env_steps/s", "cores": threads, "kind": "port", "cpu_share": share,
            "sample": f"C/OpenMP oracle ({'dense Newton-Raphson as the reference codes it' if env_kwargs['solver'] == 'nr' else 'forward/backward sweep, dense mismatch'}), {n_done} env-steps of the same workload "
                      f"(batches of {B}) in {dt:.1f} s on {threads} threads"}


def solve3_batch(spec, P, Q, tolerance=1e-6, max_iterations=100, threads=0) -> dict:
    """Three-phase FBS on the CPU (orc3_solve_batch).  ``spec`` is an UnbalancedFeederSpec-like object
    whose nodes satisfy parent[i] < i with the source at 0 (the generators in unbalanced.py do)."""
    n = len(spec.parent)
    parent = np.ascontiguousarray(spec.parent, dtype=np.int32)
    phases = np.ascontiguousarray(spec.phases, dtype=np.uint8)
    z = np.asarray(spec.z)
    pres = ((phases[:, None] >> np.arange(3)[None, :]) & 1).astype(bool)
    y = np.zeros_like(z)
    zz = z.copy()
    for i in range(1, n):
        idx = np.nonzero(pres[i])[0]
        m = np.zeros((3, 3), dtype=complex); m[np.ix_(idx, idx)] = z[i][np.ix_(idx, idx)]
        zz[i] = m
        y[i][np.ix_(idx, idx)] = np.linalg.inv(z[i][np.ix_(idx, idx)])
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
    zre, zim, yre, yim = f(zz.real.reshape(n, 9)), f(zz.imag.reshape(n, 9)), f(y.real.reshape(n, 9)), f(y.imag.reshape(n, 9))
    P = f(P); Q = f(np.zeros_like(P) if Q is None else Q)
    B = P.shape[0]
    out = dict(v_re=np.empty((B, n, 3)), v_im=np.empty((B, n, 3)), losses=np.empty(B), max_mismatch=np.empty(B),
               iterations=np.empty(B, dtype=np.int32), converged=np.empty(B, dtype=np.uint8))
    vs = f(spec.v_source)
    rc = lib().orc3_solve_batch(C.c_int32(n), _p(parent, _ip), _p(phases, _up), _p(zre, _dp), _p(zim, _dp), _p(yre, _dp),
                                _p(yim, _dp), _p(vs, _dp), C.c_int32(B), _p(P, _dp), _p(Q, _dp), C.c_double(tolerance),
                                C.c_int32(max_iterations), C.c_int32(threads), _p(out["v_re"], _dp), _p(out["v_im"], _dp),
                                _p(out["losses"], _dp), _p(out["max_mismatch"], _dp), _p(out["iterations"], _ip),
                                _p(out["converged"], _up))
    if rc != 0:
        raise RuntimeError(f"orc3_solve_batch failed ({rc})")
    out["voltages"] = out["v_re"] + 1j * out["v_im"]
    return out
