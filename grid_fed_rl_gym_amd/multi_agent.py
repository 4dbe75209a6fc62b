"""Batched mirror of the reference's ``MultiAgentEnvironmentWrapper`` (algorithms/multi_agent.py:36-134): per-agent
views of one ``BatchedGridEnvironment`` -- observation slices, joint-action assembly, reward split -- with the batch
axis in front of everything.  Host-side array plumbing only; the step itself is the fused HIP kernel."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class AgentConfig:
    """multi_agent.py:25-33 (same field names and defaults)."""
    agent_id: str
    observation_dim: int
    action_dim: int
    agent_type: str = "continuous"
    learning_rate: float = 1e-3
    hidden_dims: Optional[List[int]] = None


class BatchedMultiAgentWrapper:
    """``reset()`` -> {agent: obs[B, obs_dim_a]};  ``step({agent: act[B, act_dim_a]})`` ->
    ({agent: obs}, {agent: reward[B]}, {agent: done[B]}, {agent: info}).

    Semantics per instance are the reference's: observations are consecutive slices of the global observation in agent
    order, zero-padded where the global observation is shorter (:82-99); an agent that supplies no action contributes
    zeros, scalars and higher-rank actions are flattened (:101-116); the reward is split equally and
    ``info["<agent>_reward_bonus"]`` (scalar or [B]) is added to that agent's share (:118-133); done = terminated or
    truncated for every agent (:75)."""

    def __init__(self, base_env: Any, agent_configs: Sequence[AgentConfig]):
        self.base_env = base_env
        self.agent_configs = {c.agent_id: c for c in agent_configs}
        self.n_agents = len(agent_configs)
        self.agent_obs_dims = {c.agent_id: c.observation_dim for c in agent_configs}
        self.agent_action_dims = {c.agent_id: c.action_dim for c in agent_configs}

    def reset(self, **kwargs: Any) -> Dict[str, np.ndarray]:
        global_obs, _info = self.base_env.reset(**kwargs)
        return self._split_observation(global_obs)

    def step(self, actions: Dict[str, np.ndarray]):
        joint = self._combine_actions(actions)
        global_obs, global_reward, terminated, truncated, info = self.base_env.step(joint)
        done = np.asarray(terminated, dtype=bool) | np.asarray(truncated, dtype=bool)
        return (self._split_observation(global_obs), self._split_reward(global_reward, info),
                {a: done.copy() for a in self.agent_configs}, {a: info for a in self.agent_configs})

    def _batch(self) -> int:
        return int(getattr(self.base_env, "num_envs"))

    def _split_observation(self, global_obs: np.ndarray) -> Dict[str, np.ndarray]:
        g = np.asarray(global_obs)
        B, D = g.shape
        out, start = {}, 0
        for a, d in self.agent_obs_dims.items():
            end = start + d
            if end <= D:
                out[a] = g[:, start:end]
            else:
                o = np.zeros((B, d))
                if start < D:
                    o[:, :D - start] = g[:, start:]
                out[a] = o
            start = end
        return out

    def _combine_actions(self, actions: Dict[str, np.ndarray]) -> np.ndarray:
        B = self._batch()
        parts = []
        for a in self.agent_configs:
            if a in actions:
                act = np.asarray(actions[a], dtype=np.float64)
                if act.ndim == 0:                       # one scalar for the whole batch
                    act = np.full((B, 1), float(act))
                elif act.ndim == 1 and act.shape[0] == B and self.agent_action_dims[a] == 1:
                    act = act.reshape(B, 1)             # one scalar per instance
                parts.append(act.reshape(B, -1))
            else:
                parts.append(np.zeros((B, self.agent_action_dims[a])))
        return np.concatenate(parts, axis=1)

    def _split_reward(self, global_reward: np.ndarray, info: Dict[str, Any]) -> Dict[str, np.ndarray]:
        base = np.asarray(global_reward, dtype=np.float64) / self.n_agents
        out = {}
        for a in self.agent_configs:
            r = base.copy()
            key = f"{a}_reward_bonus"
            if key in info:
                r = r + np.asarray(info[key], dtype=np.float64)
            out[a] = r
        return out
