"""Soak of the two round-4 meshed solvers through the device rollout collector (T fused steps per call, random actions drawn on the
device, in-place resets of finished episodes): hundreds of steps of a few thousand instances must leave the same rewards, terminal
flags and final observations as the first-generation sparse block LU they replace -- the kind of run in which an LDS hazard that fires
once in a million rows shows.  (Each step starts its Newton iteration from the flat start, so the two builds cannot drift apart: a
difference is a wrong solve, not chaos.)"""
import numpy as np
import pytest

import grid_fed_rl_gym_amd as P

pytestmark = pytest.mark.gpu


def _rollouts(spec, B, calls, monkeypatch, no_mesh=False, **kw):
    if no_mesh:
        monkeypatch.setenv("GS_NO_MESH2", "1")
    env = P.BatchedGridEnvironment(spec, num_envs=B, solver="nr", stochastic_loads=True, weather_variation=True, jacobian="exact",
                                   episode_length=40, **kw)
    monkeypatch.delenv("GS_NO_MESH2", raising=False)
    kernel = env.handle.describe()["kernel"]
    env.reset(seed=np.arange(B, dtype=np.uint64) + 1)
    outs = []
    for r in range(calls):
        env.handle.rollout(50, "random", seed=100 + r)
        d = env.handle.rollout_download(want=("rewards", "terminals", "final_observation"))
        outs.append((d["rewards"].copy(), d["terminals"].copy(), d["final_observation"].copy(), d["n_terminal"]))
    env.close()
    return kernel, outs


def _same(a, b, tol):
    finished = 0
    for (r1, t1, o1, n1), (r2, t2, o2, n2) in zip(a, b):
        assert np.array_equal(t1, t2) and n1 == n2
        assert np.isfinite(o1).all()
        assert np.max(np.abs(o1 - o2) / np.maximum(1.0, np.abs(o2))) < tol
        assert np.max(np.abs(r1 - r2) / np.maximum(1.0, np.abs(r2))) < tol
        finished += n1
    return finished


def test_meshed_member_soak_against_the_slab_row_sparse_lu(monkeypatch):
    spec = P.random_meshed(123, 26, seed=1)
    k1, a = _rollouts(spec, 2048, 4, monkeypatch)
    k2, b = _rollouts(spec, 2048, 4, monkeypatch, no_mesh=True)
    assert (k1, k2) == ("nr_mesh2", "nr_sparse_lu")
    assert _same(a, b, 1e-10) > 5000               # episodes of 40 steps: thousands of in-place resets on the way


def test_dense_block_row_soak_against_the_sparse_lu(monkeypatch):
    spec = P.scalable_like(60, seed=4)
    k1, a = _rollouts(spec, 700, 3, monkeypatch, linear_solver="dense_mfma")
    k2, b = _rollouts(spec, 700, 3, monkeypatch, linear_solver="sparse_lu")
    assert (k1, k2) == ("nr_dense_mfma", "nr_sparse_lu")
    assert _same(a, b, 1e-10) > 1000
