#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8(f) row 4, captured by importing the reference in THIS container only:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 oracle/capture_golden_wrappers.py

  wrappers_multi_agent.npz   MultiAgentEnvironmentWrapper (algorithms/multi_agent.py:36-134) around a deterministic
                             stand-in environment: observation split (with zero padding), joint action assembly
                             (missing agent -> zeros, scalar action), equal reward split + per-agent bonus, done flags
  wrappers_feeder_dict.json  CustomFeeder.from_dict -> to_dict (feeders/base.py:170-253) on a network dict that omits
                             every optional key, so the defaults the reference fills in are part of the fixture
Test infrastructure; fixtures hold inputs and expected outputs only.
"""
import hashlib, json, logging, os
import numpy as np

logging.disable(logging.CRITICAL)
from grid_fed_rl.algorithms.multi_agent import MultiAgentEnvironmentWrapper, AgentConfig        # noqa: E402
from grid_fed_rl.feeders.base import CustomFeeder                                                  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class StandIn:
    """reset/step with the reference env's signatures; every quantity a fixed function of the step count."""
    def __init__(self, obs_dim, rng):
        self.obs_dim, self.t, self.rng = obs_dim, 0, rng
        self.seen = []
    def reset(self):
        self.t = 0
        return np.arange(self.obs_dim, dtype=float) * 0.5, {}
    def step(self, action):
        self.t += 1
        self.seen.append(np.asarray(action, dtype=float).copy())
        obs = np.arange(self.obs_dim, dtype=float) * 0.5 + self.t + 0.01 * float(np.sum(action))
        info = {"a1_reward_bonus": 0.25 * self.t} if self.t % 2 else {}
        return obs, -3.0 * self.t + float(np.sum(action)), self.t == 3, self.t == 4, info


def main():
    mpath = os.path.join(OUT, "manifest.json")
    manifest = json.load(open(mpath))
    rng = np.random.default_rng(7)
    cfgs = [AgentConfig("a0", 4, 2), AgentConfig("a1", 3, 1), AgentConfig("a2", 5, 3)]      # 12 > obs_dim 10: last agent padded
    env = StandIn(10, rng)
    w = MultiAgentEnvironmentWrapper(env, cfgs)
    obs0 = w.reset()
    arrays = {"agent_obs_dims": np.array([4, 3, 5]), "agent_action_dims": np.array([2, 1, 3]), "obs_dim": np.array(10)}
    for a in ("a0", "a1", "a2"):
        arrays[f"reset_obs_{a}"] = obs0[a]
    acts = [{"a0": np.array([0.1, -0.2]), "a1": np.array([0.3]), "a2": np.array([0.5, 0.6, -0.7])},
            {"a0": np.array([[0.4], [0.9]]), "a2": np.array([1.0, 0.0, -1.0])},                      # a1 missing; a0 2-D
            {"a0": np.array([0.0, 0.0]), "a1": np.float64(0.8), "a2": np.array([0.2, 0.2, 0.2])},    # scalar action
            {"a1": np.array([-0.5])}]
    for t, act in enumerate(acts):
        o, r, d, info = w.step(act)
        for a in ("a0", "a1", "a2"):
            arrays[f"step{t}_obs_{a}"] = o[a]; arrays[f"step{t}_reward_{a}"] = np.array(r[a]); arrays[f"step{t}_done_{a}"] = np.array(d[a])
            if a in act:
                arrays[f"step{t}_action_{a}"] = np.asarray(act[a], dtype=float).reshape(-1)
        arrays[f"step{t}_joint_action"] = env.seen[-1]
    clean = {k: np.asarray(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, "wrappers_multi_agent.npz"), **clean)
    h = hashlib.sha256()
    for k in sorted(clean):
        h.update(k.encode()); h.update(np.ascontiguousarray(clean[k]).tobytes())
    manifest["files"]["wrappers_multi_agent.npz"] = {"arrays": sorted(clean), "sha256_of_arrays": h.hexdigest()}

    net = {"buses": [{"id": 1, "type": "slack"}, {"id": 2}, {"id": "b3", "voltage_level": 480.0, "base_voltage": 0.48}, {"id": 4, "type": "pv"}],
           "lines": [{"id": "l12", "from_bus": 1, "to_bus": 2, "resistance": 0.01, "reactance": 0.02},
                     {"id": "l23", "from_bus": 2, "to_bus": "b3", "resistance": 0.015, "reactance": 0.03, "rating": 2e6},
                     {"id": 7, "from_bus": 2, "to_bus": 4, "resistance": 0.02, "reactance": 0.025}],
           "loads": [{"id": "ld2", "bus": 2, "power": 1.5e5}, {"id": "ld3", "bus": "b3", "power": 2.5e5, "power_factor": 0.9}],
           "generators": [{"id": "pv1", "type": "solar", "bus": 4, "capacity": 5e5},
                          {"id": "w1", "type": "wind", "bus": "b3", "capacity": 8e5, "rated_speed": 11.0}]}
    f = CustomFeeder("dict_case")
    f.from_dict(net)
    out = f.to_dict()
    json.dump({"input": net, "to_dict": out}, open(os.path.join(OUT, "wrappers_feeder_dict.json"), "w"), indent=1, sort_keys=True)
    manifest["files"]["wrappers_feeder_dict.json"] = {"sha256": hashlib.sha256(json.dumps(out, sort_keys=True).encode()).hexdigest()}
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote wrappers_multi_agent.npz, wrappers_feeder_dict.json")


if __name__ == "__main__":
    main()
