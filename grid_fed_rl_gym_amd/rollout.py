"""Batched rollout collection -- the immediate consumer of env.step().

Mirrors the reference's ``collect_random_data(env, num_steps)`` and ``GridDataset``
(reference algorithms/base.py:268-298, 180-266): same dictionary keys, same normalisation
arithmetic.  The loop itself runs on the device (``gs_rollout``, include/gridstep.h): ``num_steps``
fused step kernels back to back, each writing its observation block straight into the next slot of a
``[T + 1, B, obs_dim]`` device buffer, random actions drawn on the device, finished instances reset in
place where the reference calls ``env.reset()`` (base.py:289-290) -- one host copy at the end, or none
(``rollout_device``).  NumPy only -- the learner side (torch) is out of scope.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .env import BatchedGridEnvironment


def collect_random_data(env: BatchedGridEnvironment, num_steps: int, seed: int = 0,
                        actions: Optional[np.ndarray] = None, reset: bool = True) -> Dict[str, np.ndarray]:
    """``num_steps`` batched steps with uniform random actions in (-1, 1) (the reference samples
    ``env.action_space``, base.py:280) -> ``num_steps * B`` transitions, time-major
    (transition index = t * B + b).  ``actions`` ([num_steps, B, A]) overrides the sampling.
    ``reset=False`` continues from where the environment stands (the reference always resets first, base.py:277)."""
    rollout_device(env, num_steps, seed=seed, actions=actions, reset=reset)
    d = env.handle.rollout_download()
    T, B = int(num_steps), env.num_envs
    out = {k: d[k].reshape((T * B,) + d[k].shape[2:]) for k in ("observations", "actions", "rewards", "next_observations")}
    out["terminals"] = d["terminals"].reshape(T * B) != 0
    return out


def rollout_device(env: BatchedGridEnvironment, num_steps: int, seed: int = 0, actions: Optional[np.ndarray] = None,
                   reset: bool = True):
    """The same collection left on the GPU: returns the ``gs_rollout_device`` view (device pointers to
    ``obs_seq[T + 1, B, obs_dim]``, actions, rewards, done flags and the list of terminal observations) for a learner
    that consumes it there.  Valid until the next rollout on the environment."""
    if reset or env._needs_reset:
        env.reset(seed=seed)
    if actions is None:
        env.handle.rollout(int(num_steps), "random", seed=seed)
    else:
        env.handle.rollout(int(num_steps), "uploaded", actions=np.asarray(actions, dtype=np.float64))
    return env.handle.rollout_device_view()


class GridDataset:
    """Transition store with the reference's normalisation (algorithms/base.py:207-224):
    observations / actions standardised per column with ``std + 1e-6``, rewards by their scalar
    mean / std; ``next_observations`` use the observation statistics."""

    def __init__(self, observations, actions, rewards, next_observations, terminals, normalize: bool = True) -> None:
        self.observations = np.asarray(observations, dtype=np.float64)
        self.actions = np.asarray(actions, dtype=np.float64)
        self.rewards = np.asarray(rewards, dtype=np.float64)
        self.next_observations = np.asarray(next_observations, dtype=np.float64)
        self.terminals = np.asarray(terminals)
        if normalize:
            self._normalize_data()
        self.size = len(self.observations)

    def _normalize_data(self) -> None:
        self.obs_mean = np.mean(self.observations, axis=0)
        self.obs_std = np.std(self.observations, axis=0) + 1e-6
        self.observations = (self.observations - self.obs_mean) / self.obs_std
        self.next_observations = (self.next_observations - self.obs_mean) / self.obs_std
        self.action_mean = np.mean(self.actions, axis=0)
        self.action_std = np.std(self.actions, axis=0) + 1e-6
        self.actions = (self.actions - self.action_mean) / self.action_std
        self.reward_mean = np.mean(self.rewards)
        self.reward_std = np.std(self.rewards) + 1e-6
        self.rewards = (self.rewards - self.reward_mean) / self.reward_std

    def sample_batch(self, batch_size: int, rng: Optional[np.random.Generator] = None) -> Dict[str, np.ndarray]:
        rng = rng or np.random.default_rng()
        idx = rng.integers(0, self.size, batch_size)
        return {"observations": self.observations[idx], "actions": self.actions[idx], "rewards": self.rewards[idx],
                "next_observations": self.next_observations[idx], "terminals": self.terminals[idx].astype(np.float64)}

    def get_all_data(self) -> Dict[str, np.ndarray]:
        return {"observations": self.observations, "actions": self.actions, "rewards": self.rewards,
                "next_observations": self.next_observations, "terminals": self.terminals.astype(np.float64)}

    def denormalize_action(self, action: np.ndarray) -> np.ndarray:
        return action * self.action_std + self.action_mean if hasattr(self, "action_mean") else action

    def denormalize_observation(self, obs: np.ndarray) -> np.ndarray:
        return obs * self.obs_std + self.obs_mean if hasattr(self, "obs_mean") else obs
