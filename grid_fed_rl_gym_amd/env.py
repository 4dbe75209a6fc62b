"""Batched GridEnvironment: B independent feeder instances stepped by one kernel sequence.

Mirrors the reference's Gym-style API -- ``reset(seed, options) -> (obs, info)``,
``step(action) -> (obs, reward, terminated, truncated, info)`` (reference
environments/grid_env.py:360, 410; base.py:109-121) -- with a leading batch axis on every
return value, and its list-based batched form ``VectorizedEnvironment.reset(seeds)/step(actions)``
(utils/parallel_environment.py:309-355).  Observation and action layouts are the reference's
(grid_env.py:753-783, 621-651).

Differences that are deliberate, all of them documented in DESIGN.md:
* the feeder passed in is actually used (the reference ignores it and always builds a fixed
  3-bus network, grid_env.py:243-298; ``reference_default()`` builds that network);
* the load flow is Newton-Raphson (or FBS) on the GPU, not the five-solver heuristic
  fallback chain;
* stochastic loads / weather draw from a counter-based Philox stream per (seed, instance,
  step) instead of the process-global ``random`` / ``np.random`` streams;
* a non-finite action row is replaced by the neutral action for that instance (the
  reference's intent at grid_env.py:427) and out-of-range values are applied unclipped,
  as the reference does.
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .components import InvalidActionError
from .feeders import FeederSpec, flatten_feeder, reference_env_network, with_reference_env_renewables


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (the reference ships its own, base.py:37-63)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).copy() if shape is not None else np.asarray(low, dtype=np.float64)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).copy() if shape is not None else np.asarray(high, dtype=np.float64)
        self.shape = tuple(shape) if shape is not None else self.low.shape
        self.dtype = dtype

    def sample(self, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        rng = rng or np.random.default_rng()
        return rng.uniform(self.low, self.high)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class BatchedGridEnvironment:
    """``num_envs`` instances of one feeder on one GPU.

    Parameters follow ``GridEnvironment.__init__`` (grid_env.py:161-174); additions:
    ``num_envs``, ``solver`` ("nr" | "fbs"), ``jacobian``, ``zero_z``, ``tolerance``,
    ``max_iterations``, ``power_base`` (injections are divided by it before the solve; the
    reference passes watts straight into a per-unit network, i.e. 1.0), ``device``,
    ``first_instance`` (global index of instance 0 when a batch is sharded across GPUs).
    """

    def __init__(self, feeder: Union[FeederSpec, Any], num_envs: int = 1, timestep: float = 1.0,
                 episode_length: int = 86400, stochastic_loads: bool = True, renewable_sources: Optional[List[str]] = None,
                 weather_variation: bool = True, power_flow_solver: Optional[Any] = None,
                 voltage_limits: Tuple[float, float] = (0.95, 1.05), frequency_limits: Tuple[float, float] = (59.5, 60.5),
                 safety_penalty: float = 100.0, *, solver: str = "nr", jacobian: str = "exact", zero_z: str = "open",
                 tolerance: float = 1e-6, max_iterations: int = 50, acceleration_factor: float = 1.0,
                 linear_solver: str = "auto", power_base: Optional[float] = None, device: int = 0,
                 first_instance: int = 0, waves_per_group: int = 0, warm_start: bool = False,
                 pinned_host_buffers: bool = False, recycle_host_buffers: bool = True, obs_dtype: Any = np.float64, **kwargs: Any) -> None:
        spec = feeder if isinstance(feeder, FeederSpec) else flatten_feeder(feeder)
        if renewable_sources is not None:
            keep = [g for g in range(spec.n_gens)
                    if ("solar" in renewable_sources and spec.gen_kind[g] == 0) or ("wind" in renewable_sources and spec.gen_kind[g] == 1)]
            if len(keep) != spec.n_gens:
                import copy
                spec = copy.copy(spec)
                for f in ("gen_bus", "gen_kind", "gen_cap", "gen_p0", "gen_p1", "gen_p2"):
                    setattr(spec, f, getattr(spec, f)[keep])
        if power_flow_solver is not None:
            # a solver object configured like the reference's: take its numerical settings
            solver = getattr(power_flow_solver, "solver_kind", solver)
            jacobian = getattr(power_flow_solver, "jacobian", jacobian)
            zero_z = getattr(power_flow_solver, "zero_z", zero_z)
            tolerance = getattr(power_flow_solver, "tolerance", tolerance)
            max_iterations = getattr(power_flow_solver, "max_iterations", max_iterations)
            acceleration_factor = getattr(power_flow_solver, "acceleration_factor", acceleration_factor)
            linear_solver = getattr(power_flow_solver, "linear_solver", linear_solver)
        self.spec = spec
        self.num_envs = int(num_envs)
        self.timestep = float(timestep)
        self.episode_length = int(episode_length)
        self.stochastic_loads = bool(stochastic_loads)
        self.weather_variation = bool(weather_variation)
        self.voltage_limits = tuple(voltage_limits)
        self.frequency_limits = tuple(frequency_limits)
        self.safety_penalty = float(safety_penalty)
        self.power_base = float(spec.base_power_va if power_base is None else power_base)
        cfg = _lib.make_config(solver_kind=_lib.SOLVER[solver], jacobian_mode=_lib.JACOBIAN[jacobian],
                               zero_z_mode=_lib.ZERO_Z[zero_z], linear_solver=_lib.LINSOLVE[linear_solver],
                               max_iterations=int(max_iterations), tolerance=float(tolerance),
                               acceleration_factor=float(acceleration_factor), episode_length=self.episode_length,
                               stochastic_loads=int(self.stochastic_loads), weather_variation=int(self.weather_variation),
                               timestep=self.timestep, v_min=float(voltage_limits[0]), v_max=float(voltage_limits[1]),
                               f_min=float(frequency_limits[0]), f_max=float(frequency_limits[1]),
                               safety_penalty=self.safety_penalty, power_base=self.power_base,
                               waves_per_group=int(waves_per_group), fbs_warm_start=int(bool(warm_start) and solver == "fbs"))
        self._h = _lib.Handle(spec, cfg, self.num_envs, device, first_instance)
        # obs_dtype=np.float32 (opt-in): step() returns the observation block in the dtype the reference DECLARES for its observation
        # space (grid_env.py:346) -- rounded on the device, half the bytes over PCIe; reset(), step_device() and the rollout collector
        # stay float64, and so does every parity test
        if np.dtype(obs_dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("obs_dtype must be float64 or float32")
        self._h.obs_dtype = np.dtype(obs_dtype)
        if pinned_host_buffers and self._h.obs_dtype == np.float32:
            raise ValueError("pinned_host_buffers and obs_dtype=float32 are not combined: the recycled pool (the default) serves float32")
        if pinned_host_buffers:
            # step() then returns views of two rotating page-locked buffer sets instead of fresh arrays (valid until the
            # next-but-one step): the 45 MB observation copy of a B = 8192 batch runs at the link's rate
            self._h.use_pinned_outputs()
        elif recycle_host_buffers:
            # the default: step() returns views of page-locked buffer sets that are reused only once the caller has let go of
            # everything it got from them -- fresh-array semantics without a fresh 45 MB allocation and a pageable copy per step
            self._h.use_recycled_outputs()
        self.obs_dim, self.action_dim, self.state_dim = self._h.obs_dim, self._h.action_dim, self._h.state_dim
        big = np.finfo(np.float64).max
        self.obs_dtype = self._h.obs_dtype
        self.single_observation_space = Box(-big, big, shape=(self.obs_dim,), dtype=np.float64)
        self.single_action_space = Box(-1.0, 1.0, shape=(self.action_dim,), dtype=np.float32)    # grid_env.py:353-358
        self.observation_space = Box(-big, big, shape=(self.num_envs, self.obs_dim), dtype=np.float64)
        self.action_space = Box(-1.0, 1.0, shape=(self.num_envs, self.action_dim), dtype=np.float32)
        self._needs_reset = True

    @classmethod
    def reference_default(cls, num_envs: int = 1, renewable_sources: Optional[List[str]] = None, **kw: Any
                          ) -> "BatchedGridEnvironment":
        """The network every reference ``GridEnvironment`` actually simulates (grid_env.py:243-298),
        with the reference's unit handling (watts into a per-unit solve, power_base = 1)."""
        spec = with_reference_env_renewables(reference_env_network(), renewable_sources or [])
        kw.setdefault("power_base", 1.0)
        return cls(spec, num_envs=num_envs, **kw)

    # -- Gym API ------------------------------------------------------------------------------
    def reset(self, seed: Union[None, int, Sequence[int]] = None, options: Optional[Dict[str, Any]] = None,
              mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Dict[str, Any]]:
        """Reset all instances (or those where ``mask`` is true).  ``seed`` may be one int
        (instance b gets seed + b) or one seed per instance."""
        if seed is None:
            seeds = None
        elif np.isscalar(seed):
            seeds = (np.uint64(int(seed)) + np.arange(self.num_envs, dtype=np.uint64))
        else:
            seeds = np.asarray(seed, dtype=np.uint64)
        obs = self._h.reset(seeds, mask)
        self._needs_reset = False
        return obs, self._base_info(np.zeros(self.num_envs, dtype=np.int32), np.zeros(self.num_envs),
                                    np.zeros(self.num_envs, dtype=np.int32))

    def step(self, actions) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, Dict[str, Any]]:
        a = np.asarray(actions, dtype=np.float64)
        if a.ndim == 1 and self.num_envs == 1:
            a = a[None, :]
        if a.shape != (self.num_envs, self.action_dim):
            raise InvalidActionError(f"actions must have shape ({self.num_envs}, {self.action_dim}), got {a.shape}")
        # a non-finite action counts as "no action" (zeros) for its instance.  One pass decides the common case: a finite sum means
        # every entry is finite (NaN and +-inf survive any sum); the per-row test runs only when the sum is not.
        if np.isfinite(np.add.reduce(a, axis=None)):
            bad = np.zeros(self.num_envs, dtype=bool)
        else:
            bad = ~np.isfinite(a).all(axis=1)
            if bad.any():
                a = np.where(bad[:, None], 0.0, a)
        out = self._h.step(a)
        info = self._base_info(out["current_step"], out["episode_reward"], out["constraint_violations"])
        v = out["violations"].astype(bool)
        info.update(power_flow_converged=out["power_flow_converged"].astype(bool), max_voltage=out["max_voltage"],
                    min_voltage=out["min_voltage"], total_losses=out["total_losses"],
                    constraint_violations={"voltage_high": v[:, 0], "voltage_low": v[:, 1],
                                           "frequency_high": v[:, 2], "frequency_low": v[:, 3]},
                    iterations=out["iterations"], status=out["status"], action_invalid=bad)
        return out["obs"], out["reward"], out["terminated"].astype(bool), out["truncated"].astype(bool), info

    def step_device(self, actions, stream=None):
        """``step()`` for a policy that lives on the same GPU: ``actions`` is a float64 [num_envs, action_dim] array in
        device memory (a torch tensor, anything with ``__cuda_array_interface__``, or an address), the result is
        ``(obs, reward, terminated, truncated)`` as zero-copy ``DeviceArray`` views (``torch.as_tensor(obs, device="cuda")``).
        ``stream``: the caller's ``hipStream_t`` (torch: ``torch.cuda.current_stream().cuda_stream``) -- the step waits on
        the device for the actions queued there, and the stream waits for the step before it reads the results; None: the
        caller synchronises.  No host copy, no validation of the action values (non-finite actions are the caller's)."""
        if self._needs_reset:
            raise RuntimeError("reset() before step_device()")
        self._h.step_device_ptr(actions, stream)
        out = self._h.step_device_view(stream)
        return out["obs"], out["reward"], out["terminated"], out["truncated"]

    def _base_info(self, step, ep_reward, viol) -> Dict[str, Any]:
        # get_info(), base.py:169-176
        return {"current_step": step, "episode_reward": ep_reward, "constraint_violations_count": viol,
                "timestep": self.timestep}

    def render(self, mode: str = "human") -> None:
        return None

    def close(self) -> None:
        self._h.close()

    # -- checkpoint / resume ---------------------------------------------------------------------
    def get_state(self) -> np.ndarray:
        """[B, state_dim] blob; layout documented at gs_get_state in include/gridstep.h."""
        return self._h.get_state()

    def set_state(self, state: np.ndarray) -> None:
        self._h.set_state(state)
        self._needs_reset = False

    STATE_FIELDS = ("time", "step", "constraint_violations", "total_losses", "episode_reward", "frequency",
                    "irradiance", "wind", "temperature", "cloud", "seed_lo", "seed_hi")

    def state_column(self, name: str) -> int:
        return self.STATE_FIELDS.index(name)

    def state_layout(self) -> Dict[str, Any]:
        """Column index / slice of every field of the ``get_state()`` blob (layout: gs_get_state in include/gridstep.h)."""
        sp = self.spec
        lay: Dict[str, Any] = {name: k for k, name in enumerate(self.STATE_FIELDS)}
        o = len(self.STATE_FIELDS)
        for name, width in (("soc", sp.n_bats), ("battery_power", sp.n_bats), ("curtailment", sp.n_gens), ("vm", sp.n), ("va", sp.n),
                            ("line_flow", sp.m), ("line_loading", sp.m)):
            lay[name] = slice(o, o + width); o += width
        return lay

    def last_solution(self) -> Dict[str, np.ndarray]:
        """The load-flow solution of the last step as it stands on the device (PowerFlowSolution fields, batched)."""
        return self._h.download_solution()

    @property
    def handle(self) -> "_lib.Handle":
        return self._h


class VectorizedEnvironment:
    """List-in / list-out adapter with the reference's batched-env surface
    (utils/parallel_environment.py:283-379) over one ``BatchedGridEnvironment``."""

    def __init__(self, env: BatchedGridEnvironment) -> None:
        self.env = env
        self.num_envs = env.num_envs
        self.step_count = 0
        self.total_step_time = 0.0
        self.reset_count = 0
        self.total_reset_time = 0.0

    def reset(self, seeds: Optional[List[int]] = None) -> Tuple[List[np.ndarray], List[Dict[str, Any]]]:
        t0 = time.time()
        obs, info = self.env.reset(seed=None if seeds is None else list(seeds))
        self.reset_count += 1
        self.total_reset_time += time.time() - t0
        return [obs[b] for b in range(self.num_envs)], [self._row(info, b) for b in range(self.num_envs)]

    def step(self, actions: List[Any]):
        if len(actions) != self.num_envs:
            raise ValueError(f"Expected {self.num_envs} actions, got {len(actions)}")
        t0 = time.time()
        obs, rew, term, trunc, info = self.env.step(np.asarray([np.atleast_1d(a) for a in actions], dtype=np.float64))
        self.step_count += 1
        self.total_step_time += time.time() - t0
        B = self.num_envs
        return ([obs[b] for b in range(B)], [float(rew[b]) for b in range(B)], [bool(term[b]) for b in range(B)],
                [bool(trunc[b]) for b in range(B)], [self._row(info, b) for b in range(B)])

    @staticmethod
    def _row(info: Dict[str, Any], b: int) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for k, v in info.items():
            if isinstance(v, dict):
                out[k] = {kk: bool(vv[b]) for kk, vv in v.items()}
            elif isinstance(v, np.ndarray):
                out[k] = v[b].item()
            else:
                out[k] = v
        return out

    def close(self) -> None:
        self.env.close()

    def get_performance_stats(self) -> Dict[str, Any]:
        avg_step = self.total_step_time / self.step_count if self.step_count else 0.0
        avg_reset = self.total_reset_time / self.reset_count if self.reset_count else 0.0
        return {"num_environments": self.num_envs, "total_steps": self.step_count, "total_resets": self.reset_count,
                "avg_step_time_ms": avg_step * 1000, "avg_reset_time_ms": avg_reset * 1000,
                "steps_per_second": self.num_envs / avg_step if avg_step > 0 else 0.0,
                "parallel_enabled": True, "max_workers": 1}
