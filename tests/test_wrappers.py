"""the batched multi-agent wrapper and the dictionary network format, against fixtures
captured from the reference's MultiAgentEnvironmentWrapper and CustomFeeder (oracle/capture_golden_wrappers.py)."""
import json
import os

import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd import feeders as F
from grid_fed_rl_gym_amd.multi_agent import AgentConfig, BatchedMultiAgentWrapper

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class BatchedStandIn:
    """The capture script's stand-in environment, B copies side by side (instance b = the fixture's single environment)."""
    def __init__(self, obs_dim, B):
        self.obs_dim, self.num_envs, self.t, self.seen = obs_dim, B, 0, []
    def reset(self):
        self.t = 0
        return np.tile(np.arange(self.obs_dim, dtype=float) * 0.5, (self.num_envs, 1)), {}
    def step(self, action):
        self.t += 1
        self.seen.append(np.asarray(action, dtype=float).copy())
        s = action.sum(axis=1)
        obs = np.arange(self.obs_dim, dtype=float)[None, :] * 0.5 + self.t + 0.01 * s[:, None]
        info = {"a1_reward_bonus": np.full(self.num_envs, 0.25 * self.t)} if self.t % 2 else {}
        B = self.num_envs
        return obs, -3.0 * self.t + s, np.full(B, self.t == 3), np.full(B, self.t == 4), info


def test_multi_agent_wrapper_matches_reference_per_instance():
    d = np.load(os.path.join(GOLD, "wrappers_multi_agent.npz"))
    B = 5
    cfgs = [AgentConfig(f"a{k}", int(d["agent_obs_dims"][k]), int(d["agent_action_dims"][k])) for k in range(3)]
    env = BatchedStandIn(int(d["obs_dim"]), B)
    w = BatchedMultiAgentWrapper(env, cfgs)
    o = w.reset()
    for a in ("a0", "a1", "a2"):
        np.testing.assert_array_equal(o[a], np.tile(d[f"reset_obs_{a}"], (B, 1)))
    for t in range(4):
        act = {}
        for a in ("a0", "a1", "a2"):
            key = f"step{t}_action_{a}"
            if key in d.files:
                v = d[key]
                act[a] = float(v[0]) if (t == 2 and a == "a1") else np.tile(v, (B, 1))     # step 2: a1 gives one scalar
        if t == 1:
            act["a0"] = act["a0"].reshape(B, 2, 1)                                          # higher-rank action, flattened per instance
        obs, rew, done, info = w.step(act)
        np.testing.assert_array_equal(env.seen[-1], np.tile(d[f"step{t}_joint_action"], (B, 1)))
        for a in ("a0", "a1", "a2"):
            np.testing.assert_array_equal(obs[a], np.tile(d[f"step{t}_obs_{a}"], (B, 1)))
            np.testing.assert_array_equal(rew[a], np.full(B, float(d[f"step{t}_reward_{a}"])))
            np.testing.assert_array_equal(done[a], np.full(B, bool(d[f"step{t}_done_{a}"])))
            assert info[a] is info["a0"]
    # per-instance scalars for a one-dimensional agent
    w2 = BatchedMultiAgentWrapper(BatchedStandIn(10, B), cfgs)
    w2.reset()
    w2.step({"a1": np.linspace(-1, 1, B)})
    np.testing.assert_array_equal(w2.base_env.seen[-1][:, 2], np.linspace(-1, 1, B))
    assert w2.base_env.seen[-1].shape == (B, 6)


def test_dictionary_network_format_matches_custom_feeder():
    d = json.load(open(os.path.join(GOLD, "wrappers_feeder_dict.json")))
    spec = F.feeder_from_dict(d["input"], name="dict_case")
    assert F.feeder_to_dict(spec) == d["to_dict"]                        # defaults filled exactly as CustomFeeder fills them
    assert F.network_dict_normalized(d["input"], "dict_case") == d["to_dict"]
    assert spec.n == 4 and spec.m == 3 and spec.n_loads == 2 and spec.n_gens == 2
    assert spec.bus_ids == [1, 2, "b3", 4] and spec.bus_type.tolist() == [F.SLACK, F.PQ, F.PQ, F.PV]
    np.testing.assert_array_equal(spec.rating, [1e6, 2e6, 1e6])
    np.testing.assert_array_equal(spec.load_pf, [0.95, 0.9])
    assert spec.gen_p1[1] == 11.0                                         # wind rated_speed came through the generator entry
    # a second pass through the normalised form is the identity
    assert F.feeder_to_dict(F.feeder_from_dict(d["to_dict"], name="dict_case")) == d["to_dict"]


@pytest.mark.parametrize("make", [F.ieee13_like, F.ieee123_like, lambda: F.with_reference_env_renewables(F.reference_env_network(), ["solar", "wind"]),
                                  lambda: F.random_meshed(20, 30, seed=3)])
def test_any_feeder_round_trips_through_the_dictionary_format(make):
    spec = make()
    again = F.feeder_from_dict(json.loads(json.dumps(F.feeder_to_dict(spec))))
    assert again.sha256() == spec.sha256() and again.bus_ids == spec.bus_ids
    assert again.obs_dim == spec.obs_dim and again.action_dim == spec.action_dim


@pytest.mark.gpu
def test_multi_agent_wrapper_over_the_device_environment():
    spec = P.ieee123_like(); B = 64
    env = P.BatchedGridEnvironment(spec, num_envs=B, solver="fbs")
    ref = P.BatchedGridEnvironment(spec, num_envs=B, solver="fbs")
    dims = [2 * spec.n, 2 * spec.m + 1, spec.obs_dim - (2 * spec.n + 2 * spec.m + 1)]
    cfgs = [AgentConfig("volt", dims[0], spec.n_bats), AgentConfig("flow", dims[1], 0), AgentConfig("der", dims[2], spec.n_gens)]
    w = BatchedMultiAgentWrapper(env, cfgs)
    o = w.reset(seed=11); g0, _ = ref.reset(seed=11)
    np.testing.assert_array_equal(np.concatenate([o["volt"], o["flow"], o["der"]], axis=1), g0)
    rng = np.random.default_rng(1)
    for _ in range(3):
        a_b, a_g = rng.uniform(-1, 1, (B, spec.n_bats)), rng.uniform(-1, 1, (B, spec.n_gens))
        obs, rew, done, info = w.step({"volt": a_b, "der": a_g})
        g, r, te, tr, _ = ref.step(np.concatenate([a_b, a_g], axis=1))
        np.testing.assert_array_equal(np.concatenate([obs["volt"], obs["flow"], obs["der"]], axis=1), g)
        np.testing.assert_array_equal(rew["volt"] + rew["flow"] + rew["der"], 3 * (r / 3))
        np.testing.assert_array_equal(done["flow"], te | tr)
    env.close(); ref.close()
