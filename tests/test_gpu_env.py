"""GPU parity for the batched environment step, through the C ABI.

* G10: the reference's own deterministic trajectories of its hard-coded 3-bus env
  (tests/golden/env_*.npz) -- observation, reward, flags, step by step, as coded.
* seeded batches on the 13- and 123-bus feeders against the NumPy oracle (exact Jacobian,
  per-unit scaling), including stochastic loads / weather on the shared Philox stream,
  ragged batch sizes, masked reset and checkpoint round trips.
"""
import os

import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from oracle import oracle_np as O
from tests.helpers import golden, oracle_spec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,sources", [("env_ref3_norenew_it1", []), ("env_ref3_solarwind_it1", ["solar", "wind"]),
                                          ("env_ref3_solarwind_it2", ["solar", "wind"])])
def test_reference_trajectory_as_coded(name, sources):
    d = golden(name)
    B = 5     # identical instances; all must reproduce the reference trajectory
    env = P.BatchedGridEnvironment.reference_default(
        num_envs=B, renewable_sources=sources, stochastic_loads=False, weather_variation=False,
        episode_length=int(d["episode_length"]), jacobian="as_coded", zero_z="open", tolerance=1e-6,
        max_iterations=int(d["max_it"]))
    obs, info = env.reset(seed=0)
    st = env.get_state()
    st[:, env.state_column("time")] = float(d["t0"])
    if float(d["wind_speed"]) >= 0:
        st[:, env.state_column("wind")] = float(d["wind_speed"])
    env.set_state(st)
    assert obs.shape == (B, d["obs"].shape[1])
    for k, a in enumerate(d["actions"]):
        obs, rew, term, trunc, info = env.step(np.tile(a, (B, 1)))
        ref = d["obs"][k + 1]
        scale = np.maximum(1.0, np.abs(ref))
        err = np.max(np.abs(obs - ref[None, :]) / scale[None, :])
        # iterates are O(1e4..1e9) here (watts into a per-unit solve, F4); 1e-9 relative per entry
        assert err < 1e-9, (k, err)
        assert np.all(np.abs(rew - d["reward"][k]) <= 1e-9 * max(1.0, abs(d["reward"][k])))
        assert np.all(term == bool(d["terminated"][k])) and np.all(trunc == bool(d["truncated"][k]))
        assert np.all(info["power_flow_converged"] == bool(d["converged"][k]))
        v = info["constraint_violations"]
        got = [v["voltage_high"][0], v["voltage_low"][0], v["frequency_high"][0], v["frequency_low"][0]]
        assert got == [bool(z) for z in d["violations"][k]]
        assert np.all(np.abs(info["max_voltage"] - d["vmax"][k]) <= 1e-9 * max(1.0, abs(d["vmax"][k])))
        assert np.all(np.abs(info["total_losses"] - d["losses"][k]) <= 1e-9 * max(1.0, abs(d["losses"][k])))
    env.close()


def _oracle_rollout(fs, cfg, actions, seeds, first_instance=0, t0=None):
    """actions [T, B, A] -> lists of per-step (obs, rew, term, trunc, info) per instance."""
    T, B, _ = actions.shape
    spec = oracle_spec(fs, **cfg)
    out = []
    for b in range(B):
        _, st = O.env_reset(spec, seed=int(seeds[b]), instance=first_instance + b)
        if t0 is not None:
            st.time = t0
        traj = [O.env_step(spec, st, actions[t, b]) for t in range(T)]
        out.append(traj)
    return out


@pytest.mark.parametrize("maker,B,T,solver,stoch", [
    (lambda: P.ieee13_like("epsilon"), 70, 4, "nr", False),
    (lambda: P.ieee13_like("epsilon"), 33, 3, "nr", True),
    (lambda: P.ieee123_like(), 66, 3, "nr", False),
    (lambda: P.ieee123_like(), 20, 2, "fbs", True),
])
def test_seeded_rollouts_against_oracle(maker, B, T, solver, stoch):
    fs = maker()
    rng = np.random.default_rng(5678)
    actions = rng.uniform(-1, 1, (T, B, fs.action_dim))
    seeds = np.arange(100, 100 + B, dtype=np.uint64)
    t0 = 11.5 * 3600.0
    env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=stoch, weather_variation=stoch, solver=solver,
                                   jacobian="exact", tolerance=1e-9, max_iterations=100, first_instance=1000)
    env.reset(seed=seeds)
    st = env.get_state()
    st[:, env.state_column("time")] = t0
    env.set_state(st)
    cfg = dict(stochastic_loads=stoch, weather_variation=stoch, power_base=fs.base_power_va, solver=solver,
               tolerance=1e-9, max_iterations=100, jacobian_mode="exact", zero_z="open")
    ref = _oracle_rollout(fs, cfg, actions, seeds, first_instance=1000, t0=t0)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(actions[t])
        for b in range(0, B, 3):
            o, r, te, tr, inf = ref[b][t]
            scale = np.maximum(1.0, np.abs(o))
            # voltages/angles/flows: 1e-8 (north-star bar 1e-6 pu); FBS and NR stop at mismatch < 1e-9
            assert np.max(np.abs(obs[b] - o) / scale) < 1e-8, (t, b, int(np.argmax(np.abs(obs[b] - o) / scale)))
            assert abs(rew[b] - r) <= 1e-7 * max(1.0, abs(r))
            assert bool(term[b]) == te and bool(trunc[b]) == tr
            assert bool(info["power_flow_converged"][b]) == inf["power_flow_converged"]
            assert abs(info["total_losses"][b] - inf["total_losses"]) < 1e-8
        assert info["power_flow_converged"].all()
    env.close()


@pytest.mark.parametrize("maker,B,solver,cap", [(lambda: P.ieee123_like(), 37, "fbs", 2), (lambda: P.ieee123_like(), 37, "nr", 1),
                                                (lambda: P.ieee123_like(), 1, "nr", 2), (lambda: P.ieee13_like("epsilon"), 9, "fbs", 3),
                                                (lambda: P.ieee13_like("epsilon"), 1, "nr", 2), (lambda: P.random_meshed(200, 0, seed=5), 5, "fbs", 2)])
def test_iteration_cap_in_the_fused_step_kernels_matches_the_oracle(maker, B, solver, cap):
    """power_flow.py:143-211 under an iteration cap: the solve stops after `cap` checks, reports converged = False, the
    iteration count and the last mismatch, and the step goes on with the voltages it has (the reference's behaviour; the
    environment then counts whatever limits they break).  Also the smallest batches (one instance, one partial workgroup)."""
    fs = maker()
    rng = np.random.default_rng(11)
    T = 2
    actions = rng.uniform(-1, 1, (T, B, fs.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 5
    env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=True, weather_variation=True, solver=solver, jacobian="exact",
                                   tolerance=1e-9, max_iterations=cap, first_instance=40)
    assert env.handle.describe()["kernel"] in ("fbs_flow2h", "fbs_flow2s", "fbs_flow2x", "nr_flow2", "nr_flow2s")
    env.reset(seed=seeds)
    cfg = dict(stochastic_loads=True, weather_variation=True, power_base=fs.base_power_va, solver=solver, tolerance=1e-9,
               max_iterations=cap, jacobian_mode="exact", zero_z="open")
    ref = _oracle_rollout(fs, cfg, actions, seeds, first_instance=40)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(actions[t])
        assert not info["power_flow_converged"].any() and (info["iterations"] == cap).all() and (info["status"] == 1).all()
        for b in range(B):
            o, r, te, tr, inf = ref[b][t]
            assert np.max(np.abs(obs[b] - o) / np.maximum(1.0, np.abs(o))) < 1e-9, (t, b)
            assert abs(rew[b] - r) <= 1e-8 * max(1.0, abs(r)) and bool(term[b]) == te and bool(trunc[b]) == tr
            assert inf["power_flow_converged"] is False or not inf["power_flow_converged"]
    env.close()


def test_masked_reset_checkpoint_and_list_adapter():
    fs = P.ieee13_like("epsilon")
    B = 9
    env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=False, weather_variation=False, tolerance=1e-9)
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (3, B, env.action_dim))
    env.reset(seed=7)
    env.step(a[0])
    snap = env.get_state()
    o1 = env.step(a[1])
    env.set_state(snap)                      # resume from the checkpoint: identical continuation
    o2 = env.step(a[1])
    assert np.array_equal(o1[0], o2[0]) and np.array_equal(o1[1], o2[1])
    mask = np.zeros(B, dtype=np.uint8); mask[[2, 5]] = 1
    obs_r, _ = env.reset(seed=7, mask=mask)
    st = env.get_state()
    assert np.all(st[[2, 5], env.state_column("step")] == 0) and np.all(st[[0, 1, 3], env.state_column("step")] == 2)
    assert np.all(obs_r[[2, 5], 0] == 1.0)
    vec = P.VectorizedEnvironment(env)
    obs_l, infos = vec.reset(seeds=list(range(B)))
    assert len(obs_l) == B and obs_l[0].shape == (env.obs_dim,)
    o, r, te, tr, inf = vec.step([a[2, b] for b in range(B)])
    assert len(o) == len(r) == len(te) == len(tr) == len(inf) == B and isinstance(r[0], float)
    assert set(inf[0]["constraint_violations"]) == {"voltage_high", "voltage_low", "frequency_high", "frequency_low"}
    assert vec.get_performance_stats()["steps_per_second"] > 0
    with pytest.raises(P.InvalidActionError):
        env.step(np.zeros((B, env.action_dim + 1)))
    bad = a[2].copy(); bad[4, 0] = np.nan
    *_, info = env.step(bad)
    assert info["action_invalid"][4] and not info["action_invalid"][0]
    env.close()


def test_truncation_after_ten_violating_steps_and_termination():
    """base.py:140-167 / grid_env.py:604-606 through properties that do not depend on the oracle."""
    fs = P.ieee13_like("epsilon")
    env = P.BatchedGridEnvironment(fs, num_envs=4, stochastic_loads=False, weather_variation=False,
                                   voltage_limits=(0.999, 1.05), episode_length=12, safety_penalty=100.0)
    env.reset(seed=1)
    z = np.zeros((4, env.action_dim))
    for k in range(13):
        obs, rew, term, trunc, info = env.step(z)
        assert np.all(info["constraint_violations"]["voltage_low"])
        assert np.all(trunc == (k + 1 > 10)) and np.all(term == (k + 1 >= 12))
        assert np.all(info["constraint_violations_count"] == k + 1)
    env.close()


def test_rccl_allgather_single_rank():
    """gs_comm_* through RCCL with world_size 1 (all a 1-GPU box allows): the gathered block is the
    step's observation block; exercises dlopen(librccl), communicator setup and the device-side gather."""
    from grid_fed_rl_gym_amd.sharding import ShardedGridEnvironment
    from grid_fed_rl_gym_amd._lib import Handle
    fs = P.ieee13_like("epsilon")
    env = ShardedGridEnvironment(fs, global_num_envs=48, rank=0, world=1, device=0, transport="rccl",
                                 stochastic_loads=True, weather_variation=True)
    env.init_rccl(Handle.comm_unique_id())
    env.reset(seed=3)
    obs, *_ = env.step(np.random.default_rng(1).uniform(-1, 1, (48, fs.action_dim)))
    full = env.gather_observations()
    assert full.shape == (48, fs.obs_dim) and np.array_equal(full, obs)
    env.close()


def test_rccl_allgather_two_ranks_on_one_device(tmp_path):
    """The N = 2 data path through RCCL: two processes, file rendezvous, compact all-gather, against one process that
    owns the whole batch.  A one-GPU box has one device for both ranks; RCCL refuses a communicator with two ranks on one
    device ("Duplicate GPU detected"), in which case the test SKIPS with RCCL's message -- the multi-GPU runs of
    bench.py are then the only execution of this path."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_two_rank_worker.py")
    outs = [str(tmp_path / f"full{r}.npy") for r in range(2)]
    env = dict(os.environ, NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path / "rdzv"), outs[r]], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    texts, codes = [], []
    for p in procs:
        try:
            t, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            t, _ = p.communicate()
            t += "\n[killed after 150 s]"
        texts.append(t); codes.append(p.returncode)
    if any(c == 77 for c in codes) or any("Duplicate GPU" in t or "invalid usage" in t.lower() for t in texts):
        reason = next((ln for t in texts for ln in t.splitlines() if "REFUSED" in ln or "Duplicate GPU" in ln), "RCCL refused")
        pytest.skip("RCCL does not form a 2-rank communicator on one device: " + reason.strip()[:200])
    assert codes == [0, 0], "\n".join(texts)
    from grid_fed_rl_gym_amd.sharding import ShardedGridEnvironment
    fs = P.ieee13_like("epsilon")
    one = ShardedGridEnvironment(fs, global_num_envs=96, rank=0, world=1, device=0, stochastic_loads=True, weather_variation=True)
    one.reset(seed=3)
    acts = np.random.default_rng(1).uniform(-1, 1, (96, fs.action_dim))
    for _ in range(3):
        obs, *_ = one.step(acts)
    one.close()
    for o in outs:
        assert np.array_equal(np.load(o), obs)


def _oracle_collect(fs, cfg, actions, seeds, first_instance, policy_seed=None):
    """collect_random_data (algorithms/base.py:268-298) per instance with the oracle: reset, then T steps; where the
    reference calls env.reset() after a finished transition the instance's seed moves one step along its chain."""
    T = actions.shape[0] if actions is not None else cfg.pop("T")
    B = len(seeds)
    spec = oracle_spec(fs, **cfg)
    out = dict(observations=np.empty((T, B, fs.obs_dim)), actions=np.empty((T, B, fs.action_dim)), rewards=np.empty((T, B)),
               next_observations=np.empty((T, B, fs.obs_dim)), terminals=np.zeros((T, B), dtype=bool))
    for b in range(B):
        seed = int(seeds[b])
        obs, st = O.env_reset(spec, seed=seed, instance=first_instance + b)
        for t in range(T):
            a = actions[t, b] if actions is not None else O.rollout_random_actions(policy_seed, first_instance + b, t, fs.action_dim)
            nxt, r, te, tr, _ = O.env_step(spec, st, a)
            out["observations"][t, b] = obs; out["actions"][t, b] = a; out["rewards"][t, b] = r
            out["next_observations"][t, b] = nxt; out["terminals"][t, b] = te or tr
            if te or tr:
                seed = O.next_episode_seed(seed, first_instance + b)
                obs, st = O.env_reset(spec, seed=seed, instance=first_instance + b)
            else:
                obs = nxt
    return out


@pytest.mark.parametrize("solver,policy", [("nr", "uploaded"), ("fbs", "uploaded"), ("fbs", "random")])
def test_device_rollout_equals_the_reference_loop_across_episode_boundaries(solver, policy):
    """gs_rollout: the [T, B, ...] buffers equal T oracle steps + the reset rule of algorithms/base.py:283-292 (1e-8),
    including two episode boundaries (episode_length 3, T 7) and the instances' seed chains."""
    fs = P.ieee13_like("epsilon")
    B, T, first = 37, 7, 2000
    env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=True, weather_variation=True, solver=solver, episode_length=3,
                                   tolerance=1e-9, max_iterations=100, first_instance=first)
    cfg = dict(stochastic_loads=True, weather_variation=True, power_base=fs.base_power_va, solver=solver, tolerance=1e-9,
               max_iterations=100, jacobian_mode="exact", zero_z="open", episode_length=3)
    acts = np.random.default_rng(4).uniform(-1, 1, (T, B, fs.action_dim)) if policy == "uploaded" else None
    data = P.collect_random_data(env, T, seed=21, actions=acts)
    if policy == "random":
        cfg["T"] = T
    ref = _oracle_collect(fs, cfg, acts, np.uint64(21) + np.arange(B, dtype=np.uint64), first, policy_seed=21)
    term = data["terminals"].reshape(T, B)
    assert np.array_equal(term, ref["terminals"]) and term[2].all() and term[5].all() and term.sum() == 2 * B
    assert np.array_equal(data["actions"].reshape(T, B, -1), ref["actions"]) if policy == "uploaded" else \
        np.allclose(data["actions"].reshape(T, B, -1), ref["actions"], rtol=0, atol=1e-15)
    for k in ("observations", "next_observations"):
        got, want = data[k].reshape(T, B, -1), ref[k]
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert err.max() < 1e-8, (k, np.unravel_index(np.argmax(err), err.shape))
    assert np.max(np.abs(data["rewards"].reshape(T, B) - ref["rewards"]) / np.maximum(1.0, np.abs(ref["rewards"]))) < 1e-7
    obs = data["observations"].reshape(T, B, -1); nxt = data["next_observations"].reshape(T, B, -1)
    assert np.array_equal(obs[1:3], nxt[0:2]) and np.array_equal(obs[4:6], nxt[3:5])      # chained inside an episode
    assert not np.array_equal(obs[3], nxt[2]) and np.all(obs[3][:, 0] == 1.0)            # fresh episode after the in-place reset
    # the environment stands where the loop left it: one more step continues the third episode
    a1 = np.random.default_rng(9).uniform(-1, 1, (B, fs.action_dim))
    o_gpu, *_ = env.step(a1)
    view = env.handle.rollout_device_view()
    assert (view.T, view.B, view.obs_dim, view.n_terminal) == (T, B, fs.obs_dim, 2 * B)
    ds = P.GridDataset(**data)
    a = ds.sample_batch(16, np.random.default_rng(0))
    assert a["observations"].shape == (16, fs.obs_dim)
    raw = data["actions"]
    assert np.allclose(ds.denormalize_action(ds.actions), raw)
    assert np.allclose(ds.actions, (raw - raw.mean(0)) / (raw.std(0) + 1e-6))
    assert abs(ds.rewards.mean()) < 1e-9 and abs(ds.rewards.std() - 1.0) < 1e-3
    assert o_gpu.shape == (B, fs.obs_dim) and np.isfinite(o_gpu).all()
    env.close()


def test_device_rollout_equals_stepping_through_the_abi():
    """The same actions through gs_step one call at a time and through gs_rollout: bit for bit (123-bus feeder, both
    solvers' default kernels, no episode boundary), and the rollout can be continued without a reset."""
    fs = P.ieee123_like()
    B, T = 130, 5
    acts = np.random.default_rng(2).uniform(-1, 1, (2 * T, B, fs.action_dim))
    for solver in ("fbs", "nr"):
        a = P.BatchedGridEnvironment(fs, num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True)
        b = P.BatchedGridEnvironment(fs, num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True)
        o0, _ = a.reset(seed=5)
        stepped = [a.step(acts[t]) for t in range(2 * T)]
        d1 = P.collect_random_data(b, T, seed=5, actions=acts[:T])
        d2 = P.collect_random_data(b, T, seed=5, actions=acts[T:], reset=False)
        for d, off in ((d1, 0), (d2, T)):
            nxt = d["next_observations"].reshape(T, B, -1); rew = d["rewards"].reshape(T, B)
            for t in range(T):
                assert np.array_equal(nxt[t], stepped[off + t][0]), (solver, off + t)
                assert np.array_equal(rew[t], stepped[off + t][1])
        assert np.array_equal(d1["observations"].reshape(T, B, -1)[0], o0)
        assert np.array_equal(d2["observations"].reshape(T, B, -1)[0], stepped[T - 1][0])
        assert not d1["terminals"].any()
        a.close(); b.close()


def test_set_state_on_a_fresh_handle_restores_the_whole_observation():
    """Resume in a new process: a handle that was never reset gets a checkpoint; its next observation -- constant
    columns and renewable powers included -- equals the uninterrupted run's."""
    fs = P.ieee123_like()
    B = 66
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    a = P.BatchedGridEnvironment(fs, **kw)
    acts = np.random.default_rng(8).uniform(-1, 1, (3, B, fs.action_dim))
    a.reset(seed=77)
    st = a.get_state(); st[:, a.state_column("time")] = 10 * 3600.0; a.set_state(st)
    a.step(acts[0])
    snap = a.get_state()
    want1 = a.step(acts[1])
    want2 = a.step(acts[2])
    b = P.BatchedGridEnvironment(fs, **kw)            # never reset
    b.set_state(snap)
    got1 = b.step(acts[1])
    got2 = b.step(acts[2])
    for w, g in ((want1, got1), (want2, got2)):
        assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1])
    lo = 2 * fs.n + 2 * fs.m + 1
    assert np.array_equal(got1[0][:, lo:lo + 2 * fs.n_loads:2], np.tile(fs.load_base, (B, 1)))     # the static load columns are there
    a.close(); b.close()


def test_pinned_host_buffers_give_the_same_arrays_and_rotate():
    """`pinned_host_buffers=True`: step() returns views of two rotating page-locked buffer sets (gs_host_alloc) -- same
    numbers as the fresh-array path, an array stays intact through the next step and is reused by the one after."""
    fs = P.ieee13_like("epsilon"); B = 70
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    plain, pinned = P.BatchedGridEnvironment(fs, **kw), P.BatchedGridEnvironment(fs, pinned_host_buffers=True, **kw)
    plain.reset(seed=5); pinned.reset(seed=5)
    rng = np.random.default_rng(2)
    kept = []
    for t in range(4):
        a = rng.uniform(-1, 1, (B, fs.action_dim))
        o0, r0, te0, tr0, i0 = plain.step(a)
        o1, r1, te1, tr1, i1 = pinned.step(a)
        assert np.array_equal(o0, o1) and np.array_equal(r0, r1) and np.array_equal(te0, te1) and np.array_equal(tr0, tr1)
        for k in ("max_voltage", "total_losses", "iterations", "episode_reward"):
            assert np.array_equal(i0[k], i1[k]), k
        kept.append((o1, o0.copy()))
        if t >= 1:
            assert np.array_equal(kept[t - 1][0], kept[t - 1][1])          # the previous step's array is still intact
        if t >= 2:
            assert np.shares_memory(kept[t - 2][0], o1)      # ... and the one before that is being reused
    plain.close(); pinned.close()


def test_handles_with_different_lds_footprints_coexist():
    """A 123-bus NR handle (120 KB of dynamic LDS) keeps working after a small FBS handle was created."""
    big = P.BatchedGridEnvironment(P.ieee123_like(), num_envs=4, stochastic_loads=False, weather_variation=False)
    big.reset(seed=0)
    o1 = big.step(np.zeros((4, big.action_dim)))[0]
    small = P.BatchedGridEnvironment(P.ieee13_like("epsilon"), num_envs=4, solver="fbs", stochastic_loads=False,
                                     weather_variation=False)
    small.reset(seed=0)
    small.step(np.zeros((4, small.action_dim)))
    big.reset(seed=0)
    o2 = big.step(np.zeros((4, big.action_dim)))[0]
    assert np.array_equal(o1, o2)
    big.close(); small.close()


@pytest.mark.parametrize("solver", ["nr", "fbs"])
def test_repeated_steps_never_lose_a_cross_wave_write(solver):
    """Stress for memory-ordering slips between the waves of a group (a flat start or an injection written by one wave
    and read by another a barrier later): an instance whose first Newton iteration sees stale rows reports SINGULAR /
    stays at the flat start, which this catches within a few dozen launches."""
    fs = P.ieee123_like(); B = 66
    rng = np.random.default_rng(5678)
    actions = rng.uniform(-1, 1, (3, B, fs.action_dim))
    for rep in range(25):
        env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=False, weather_variation=False, solver=solver,
                                       jacobian="exact", tolerance=1e-9, max_iterations=100)
        env.reset(seed=np.arange(B, dtype=np.uint64))
        for t in range(3):
            obs, rew, term, trunc, info = env.step(actions[t])
            assert info["power_flow_converged"].all(), (rep, t, np.unique(info["status"]))
            assert not ((obs[:, 2] == 1.0) & (obs[:, 3] == 0.0)).any(), (rep, t)      # bus 1 never sits at 1.0 / 0 under load
        env.close()


def test_warm_started_sweep_solver_agrees_with_the_cold_start():
    """`warm_start=True` (option): the sweep solver resumes from the previous step's voltages.  Same tolerance, so the
    trajectories agree to the solver tolerance; slightly fewer sweeps; checkpoints still resume (set_state rebuilds e, f)."""
    fs = P.ieee123_like(); B = 96
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, tolerance=1e-9, max_iterations=100)
    cold, warm = P.BatchedGridEnvironment(fs, **kw), P.BatchedGridEnvironment(fs, warm_start=True, **kw)
    seeds = np.arange(B, dtype=np.uint64) + 17
    cold.reset(seed=seeds); warm.reset(seed=seeds)
    rng = np.random.default_rng(4)
    its_c = its_w = 0
    for t in range(6):
        a = rng.uniform(-1, 1, (B, fs.action_dim))
        oc, rc, *_, ic = cold.step(a)
        ow, rw, *_, iw = warm.step(a)
        assert ic["power_flow_converged"].all() and iw["power_flow_converged"].all()
        assert np.max(np.abs(oc - ow) / np.maximum(1.0, np.abs(oc))) < 1e-7
        its_c += ic["iterations"].sum(); its_w += iw["iterations"].sum()
        if t == 0:
            assert np.max(np.abs(oc - ow) / np.maximum(1.0, np.abs(oc))) < 1e-12     # the first step after reset starts flat either way (two kernels: sums associate differently)
        if t == 2:
            snap = warm.get_state()
    assert its_w < its_c                    # independent 10 % load noise every step leaves little to resume from: ~0.4 sweeps
    resumed = P.BatchedGridEnvironment(fs, warm_start=True, **kw)
    resumed.reset(seed=seeds); resumed.set_state(snap)
    rng2 = np.random.default_rng(4)
    for t in range(3): rng2.uniform(-1, 1, (B, fs.action_dim))
    a3 = rng2.uniform(-1, 1, (B, fs.action_dim))
    orr, *_ = resumed.step(a3)
    w2 = P.BatchedGridEnvironment(fs, warm_start=True, **kw); w2.reset(seed=seeds)
    rng3 = np.random.default_rng(4)
    for t in range(4):
        o4, *_ = w2.step(rng3.uniform(-1, 1, (B, fs.action_dim)))
    assert np.max(np.abs(orr - o4) / np.maximum(1.0, np.abs(o4))) < 1e-7
    for e in (cold, warm, resumed, w2): e.close()


@pytest.mark.parametrize("maker,B", [(P.ieee123_like, 130), (lambda: P.ieee13_like("epsilon"), 70),
                                     (lambda: P.ieee123_like(seed=7, load_seed=5), 96), (lambda: P.ieee123_like(seed=2024, load_seed=9), 64),
                                     (lambda: P.random_meshed(90, 0, seed=3), 64),           # other tree shapes: depth, fan-out, bus count
                                     (lambda: P.random_meshed(200, 0, seed=5), 48),          # 129 ... 256 buses: the eight-buses-per-sub-group member
                                     (lambda: P.random_meshed(256, 0, seed=6), 20)])
def test_dataflow_sweeps_agree_with_the_level_synchronous_kernel(maker, B, monkeypatch):
    """`fbs_flow2h` / `fbs_flow2` (16 / 32 instances per workgroup, sub-groups of a wavefront on different buses, sweeps
    as scans, observation tiles straight from the LDS slots) and `fbs_flow` (64 instances per workgroup; both: per-bus LDS slots + flags, register-resident bus state, no
    level barriers) against `fbs_lds` (level barriers): same iteration counts per instance and the same trajectories to
    rounding (they sum children's currents / per-wave partials in different orders).  Several tolerances, so that at
    some of them the instances of one group stop at different iterations and the frozen lanes are exercised."""
    fs = maker()
    seeds = np.arange(B, dtype=np.uint64) * 7 + 1
    spread = 0
    for tol in (1e-4, 3e-5, 1e-5, 3e-6, 1e-6, 1e-7, 1e-9):
        kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, tolerance=tol, max_iterations=50)
        flow2 = P.BatchedGridEnvironment(fs, **kw)
        monkeypatch.setenv("GS_FLOW2_IW", "32")
        flow2w = P.BatchedGridEnvironment(fs, **kw)
        monkeypatch.delenv("GS_FLOW2_IW")
        monkeypatch.setenv("GS_NO_FLOW2", "1")
        flow = P.BatchedGridEnvironment(fs, **kw)
        monkeypatch.setenv("GS_NO_FLOW", "1")
        sync = P.BatchedGridEnvironment(fs, **kw)
        monkeypatch.delenv("GS_NO_FLOW"); monkeypatch.delenv("GS_NO_FLOW2")
        wide = fs.n - 1 > 128
        assert flow2.handle.describe()["kernel"] in (("fbs_flow2x",) if wide else ("fbs_flow2h", "fbs_flow2s")), flow2.handle.describe()["flow2"]
        # (the 32-instance member exists in a library built with `make EXPERIMENTS=1`; in the default build the switch is not read)
        from grid_fed_rl_gym_amd import _lib
        assert flow2w.handle.describe()["kernel"] in (("fbs_flow2x",) if wide else (("fbs_flow2", "fbs_flow2s") if _lib.experiments() else ("fbs_flow2h", "fbs_flow2s"))), flow2w.handle.describe()["flow2"]
        assert flow.handle.describe()["kernel"] in (("fbs_lds", "fbs") if wide else ("fbs_flow",)) and sync.handle.describe()["kernel"] in ("fbs_lds", "fbs")
        for e in (flow2, flow2w, flow, sync):
            e.reset(seed=seeds)
        rng = np.random.default_rng(99)
        for t in range(4):
            a = rng.uniform(-1, 1, (B, fs.action_dim))
            o2, r2, t2, c2, in2 = flow2.step(a)
            ow, rw, tw, cw, inw = flow2w.step(a)
            of, rf, tf, cf, inf_ = flow.step(a)
            os_, rs, ts, cs, ins = sync.step(a)
            assert in2["power_flow_converged"].all() and inf_["power_flow_converged"].all() and ins["power_flow_converged"].all()
            assert np.array_equal(inf_["iterations"], ins["iterations"]) and np.array_equal(in2["iterations"], ins["iterations"]), tol
            spread += len(np.unique(inf_["iterations"][:64])) > 1
            assert np.array_equal(inw["iterations"], ins["iterations"]) and np.array_equal(tw, ts) and np.array_equal(cw, cs)
            for o, r in ((of, rf), (o2, r2), (ow, rw)):
                assert np.max(np.abs(o - os_) / np.maximum(1.0, np.abs(os_))) < 1e-12, tol
                assert np.allclose(r, rs, rtol=1e-12, atol=1e-12)
            assert np.array_equal(tf, ts) and np.array_equal(cf, cs) and np.array_equal(t2, ts) and np.array_equal(c2, cs)
            for k in ("max_voltage", "min_voltage", "total_losses", "episode_reward"):
                assert np.allclose(in2[k], ins[k], rtol=1e-11, atol=1e-11), k      # losses: a sum of O(1) injections that leaves 1e-3
            assert np.array_equal(in2["constraint_violations_count"], ins["constraint_violations_count"])
        st2, sts = flow2.get_state(), sync.get_state()
        assert np.allclose(st2, sts, rtol=1e-11, atol=1e-11)
        flow2.close(); flow2w.close(); flow.close(); sync.close()
    if fs.n == 123 and fs.name.endswith("seed42"):
        assert spread > 0        # at some tolerance the first group mixes instances that stop one iteration apart


@pytest.mark.parametrize("battery_w, status, iterations", [(1e13, 1, 50), (1e300, 3, 2), (float("inf"), 3, 1)])
def test_a_diverging_instance_is_flagged_and_leaves_its_workgroup_alone(battery_w, status, iterations, monkeypatch):
    """One instance of a workgroup is driven out of the solvable range through its state (a battery delivering 1e13 W, 1e300 W,
    inf): the sweeps of the second-generation kernel stop it where the level-synchronous kernel does (iteration cap / non-finite
    mismatch, seen through the summed mismatch), and the 15 instances that share its workgroup -- and everybody else -- come out
    bit for bit as without it."""
    fs = P.ieee123_like()
    B = 40

    def run(perturb):
        env = P.BatchedGridEnvironment(fs, num_envs=B, solver="fbs", stochastic_loads=False, weather_variation=False)
        env.reset(seed=np.arange(B, dtype=np.uint64))
        st = env.get_state()
        if perturb:
            st[3, env.state_layout()["battery_power"].start] = battery_w
        env.set_state(st)
        obs, rew, term, trunc, info = env.step(np.zeros((B, env.action_dim)))
        kernel = env.handle.describe()["kernel"]
        mm = env.last_solution()["max_mismatch"].copy()
        env.close()
        return kernel, obs.copy(), info["status"].copy(), info["iterations"].copy(), mm

    k_bad, o_bad, s_bad, it_bad, mm_bad = run(True)
    k_ok, o_ok, s_ok, it_ok, _ = run(False)
    assert k_bad == k_ok == "fbs_flow2h"
    others = [b for b in range(B) if b != 3]
    assert s_bad[3] == status and it_bad[3] == iterations and (mm_bad[3] > 1e6 or not np.isfinite(mm_bad[3]))
    assert np.all(s_bad[others] == 0) and np.array_equal(it_bad[others], it_ok[others])
    assert np.array_equal(o_bad[others], o_ok[others])
    monkeypatch.setenv("GS_NO_FLOW2", "1")
    k_lvl, _, s_lvl, it_lvl, _ = run(True)
    monkeypatch.delenv("GS_NO_FLOW2")
    assert k_lvl != k_bad and np.array_equal(s_lvl, s_bad) and np.array_equal(it_lvl, it_bad)


def test_dataflow_kernel_falls_back_when_a_wave_would_own_too_many_buses(monkeypatch):
    """The dataflow kernel keeps the state of at most 8 buses per wave in registers: with 4 waves per group the
    123-bus feeder takes the level-synchronous kernel, and the results do not depend on which one ran."""
    fs = P.ieee123_like(); B = 64
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=False, weather_variation=False, tolerance=1e-9, max_iterations=50)
    a = np.random.default_rng(3).uniform(-1, 1, (B, fs.action_dim))
    wide = P.BatchedGridEnvironment(fs, **kw)
    monkeypatch.setenv("GS_WAVES", "4")
    narrow = P.BatchedGridEnvironment(fs, **kw)
    monkeypatch.delenv("GS_WAVES")
    assert wide.handle.describe()["kernel"] == "fbs_flow2h" and narrow.handle.describe()["kernel"] == "fbs_lds"
    wide.reset(seed=1); narrow.reset(seed=1)
    ow = wide.step(a)[0]; on = narrow.step(a)[0]
    assert np.max(np.abs(ow - on) / np.maximum(1.0, np.abs(on))) < 1e-12
    wide.close(); narrow.close()


@pytest.mark.parametrize("solver", ["nr", "fbs"])
def test_two_builds_of_the_step_kernel_agree_bit_for_bit(solver):
    """The step kernel with the fused post-step checks is a second instantiation of the same code with different
    register allocation and timing.  This comparison is what exposed the store-data hazard of the 16-byte row stores
    (a VALU write to the data registers right behind a buffer_store_dwordx4 whose offset is in an SGPR: lanes 12-15 of
    every 16 store the new register contents; GsPairRef::put in csrc/gs_internal.h): it showed up as a 64-byte sector
    of one build holding different values."""
    from grid_fed_rl_gym_amd.safety import PostStepChecks
    spec = P.ieee123_like(); B = 130
    kw = dict(num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True)
    for rep in range(4):
        a, b = P.BatchedGridEnvironment(spec, **kw), P.BatchedGridEnvironment(spec, **kw)
        seeds = np.arange(B, dtype=np.uint64) * 3 + 1 + rep
        a.reset(seed=seeds); b.reset(seed=seeds)
        fused = PostStepChecks(b, loading="environment", fused=True)
        rng = np.random.default_rng(rep)
        for t in range(6):
            act = rng.uniform(-1, 1, (B, spec.action_dim))
            oa, ra, *_ = a.step(act); ob, rb, *_ = b.step(act)
            assert np.array_equal(oa, ob), (rep, t, np.unique(np.nonzero(oa != ob)[0])[:8])
            assert np.array_equal(ra, rb)
        fused.close(); a.close(); b.close()


def test_store_data_hazard_reproducer_is_clean_with_one_wait_state(tmp_path):
    """tools/store_data_hazard.hip: a 16-byte buffer store with its offset in an SGPR needs one wait state before its data
    registers are overwritten (the compiler does not insert it); GsPairRef::put relies on that being enough."""
    import os, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "store_data_hazard.hip")
    exe = str(tmp_path / "hazard")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", src, "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    lines = {int(l.split()[2].rstrip(":")): int(l.split()[3]) for l in out.splitlines() if l.startswith("wait states")}
    assert set(lines) == {0, 1, 2, 4}, out
    assert lines[1] == 0 and lines[2] == 0 and lines[4] == 0, out        # lines[0] is > 0 on gfx950 (the hazard itself); not asserted
    # the other store forms with zero wait states: the 8-byte stores (GsRowRef::put and plain global stores) are not affected;
    # the bare 16-byte global store is (heavily) -- the compiler never emits it bare, tools/check_store_hazard.py watches that
    forms = {l.split()[1].rstrip(","): int(l.split()[5]) for l in out.splitlines() if l.startswith("form ")}
    assert set(forms) == {"global_store_dwordx4_saddr", "buffer_store_dwordx2_soffset", "global_store_dwordx2_saddr"}, out
    assert forms["buffer_store_dwordx2_soffset"] == 0 and forms["global_store_dwordx2_saddr"] == 0, out


@pytest.mark.parametrize("solver,maker,B", [("fbs", lambda: P.ieee123_like(), 200), ("nr", lambda: P.ieee123_like(), 100),
                                            ("nr", lambda: P.ieee13_like("epsilon"), 70)])
def test_result_rows_restored_on_demand_equal_rows_written_by_every_step(solver, maker, B, monkeypatch):
    """The second-generation step kernels no longer write the (|V|, angle) and (flow, |P| / rating) row pairs -- they are
    the first 2 n + 2 m columns of the observation block the step writes anyway -- and whatever reads or partly rewrites those
    rows restores them from the block first (gridstep_abi.hip, ensure_rows).  Against a handle built with GS_EAGER_ROWS=1
    (every step writes them, as before round 3): checkpoints, load-flow solutions, post-step checks, the linear fallback, a
    masked reset, a device rollout with in-place resets and a checkpoint round trip, all bit for bit."""
    from grid_fed_rl_gym_amd.safety import PostStepChecks
    spec = maker()
    kw = dict(num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True, episode_length=9)
    lean = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.setenv("GS_EAGER_ROWS", "1")
    eager = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.delenv("GS_EAGER_ROWS")
    rng = np.random.default_rng(31)
    acts = rng.uniform(-1, 1, (6, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 5
    mask = (rng.random(B) < 0.4).astype(np.uint8)
    outs = []
    for env in (lean, eager):
        got = []
        h = env.handle
        env.reset(seed=seeds); h.upload_actions(acts)
        ck = PostStepChecks(env)
        for k in range(3):
            h.step_device(k)
        got.append(env.get_state()); got.append(env.last_solution())
        h.step_device(3)
        ck.run(); got.append(ck.download())
        h.step_device(4)
        got.append(dict(applied=h.fallback_linear(mask=mask))); got.append(env.last_solution())
        h.step_device(5)
        h.reset(seeds + np.uint64(3), mask, want_obs=False)
        got.append(env.get_state()); got.append(h.download_step())
        h.rollout(14, "random", seed=2)                     # episodes of 9 steps: in-place resets on the way
        got.append(env.get_state()); got.append(env.last_solution())
        st = env.get_state(); env.set_state(st)
        h.step_device(0)
        got.append(h.download_step()); got.append(env.get_state())
        outs.append(got)
        ck.close()
    for k, (a, b) in enumerate(zip(*outs)):
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, b), k
        else:
            for q in a:
                assert np.array_equal(np.asarray(a[q]), np.asarray(b[q])), (k, q)
    lean.close(); eager.close()


def test_environment_on_a_meshed_feeder_through_the_dense_mfma_solver_equals_the_sparse_lu_path():
    """A step of a handle whose Newton-Raphson is the dense block LU on the matrix cores is three launches (prologue | one workgroup
    per instance | epilogue + observation pack).  Against the same environment on the sparse block LU (one fused launch): steps with
    stochastic loads and weather, the fused post-step checks, a masked reset, a checkpoint round trip and a device rollout with
    in-place resets -- observations within 1e-10 (two linear solvers), discrete outputs equal."""
    from grid_fed_rl_gym_amd.safety import PostStepChecks
    spec = P.scalable_like(40, seed=3)
    B = 150
    kw = dict(num_envs=B, solver="nr", stochastic_loads=True, weather_variation=True, episode_length=6, tolerance=1e-9)
    envs = {ls: P.BatchedGridEnvironment(spec, linear_solver=ls, **kw) for ls in ("dense_mfma", "sparse_lu")}
    assert envs["dense_mfma"].handle.describe()["solve_kernel"] == "nr_dense_mfma"
    rng = np.random.default_rng(17)
    acts = rng.uniform(-1, 1, (5, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 40
    mask = (rng.random(B) < 0.3).astype(np.uint8)
    outs = {}
    for ls, env in envs.items():
        got = []
        h = env.handle
        env.reset(seed=seeds); h.upload_actions(acts)
        for k in range(2):
            h.step_device(k)
        got.append(h.download_step()); got.append(env.last_solution())
        ck = PostStepChecks(env, fused=True)
        h.step_device(2)
        got.append(ck.download()); got.append(h.download_step())
        ck.close()
        h.reset(seeds + np.uint64(1), mask, want_obs=False)
        env.set_state(env.get_state())
        h.step_device(3)
        got.append(h.download_step())
        h.rollout(9, "random", seed=5)
        got.append(h.rollout_download())
        outs[ls] = got
    for k, (a, b) in enumerate(zip(outs["dense_mfma"], outs["sparse_lu"])):
        for q in a:
            x, y = np.asarray(a[q]), np.asarray(b[q])
            if x.dtype.kind == "f":
                assert np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)), initial=0.0) < 1e-10, (k, q)
            else:
                assert np.array_equal(x, y), (k, q)
    assert outs["dense_mfma"][-1]["n_terminal"] > 0
    for env in envs.values():
        env.close()


@pytest.mark.parametrize("maker,B", [(lambda: P.ieee123_like(), 100), (lambda: P.ieee13_like("epsilon"), 70)])
def test_newton_raphson_flat_start_table_changes_nothing_beyond_rounding(maker, B, monkeypatch):
    """nr_flow2 / nr_flow2s: iteration 0 of every solve starts from the flat start, where D_i^-1, T_i and L_i of the tree elimination do
    not depend on the instance; the handle keeps them in a table (written once by the step kernel itself) and iteration 0 only carries
    its right-hand side through.  Against a handle that eliminates for itself every time (GS_NR_NO_FLAT=1): the same iterates, equal
    iteration counts, observations within 1e-12 (the table's values are the kernel's own; only the mismatch at the flat start is
    no longer formed through the K slots)."""
    spec = maker()
    kw = dict(num_envs=B, solver="nr", stochastic_loads=True, weather_variation=True)
    tab = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.setenv("GS_NR_NO_FLAT", "1")
    own = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.delenv("GS_NR_NO_FLAT")
    assert tab.handle.describe()["kernel"].startswith("nr_flow2")
    rng = np.random.default_rng(23)
    acts = rng.uniform(-1, 1, (4, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 9
    for env in (tab, own):
        env.reset(seed=seeds); env.handle.upload_actions(acts)
    for k in range(4):
        outs = []
        for env in (tab, own):
            env.handle.step_device(k)
            outs.append(env.handle.download_step())
        a, b = outs
        assert np.array_equal(a["iterations"], b["iterations"]) and np.array_equal(a["status"], b["status"]) and a["power_flow_converged"].all()
        assert np.max(np.abs(a["obs"] - b["obs"]) / np.maximum(1.0, np.abs(b["obs"]))) < 1e-12, k
        assert np.max(np.abs(a["reward"] - b["reward"]) / np.maximum(1.0, np.abs(b["reward"]))) < 1e-12
    tab.close(); own.close()


def test_step_outputs_are_recycled_only_when_the_caller_has_let_go():
    """BatchedGridEnvironment.step() by default returns views of buffer sets the handle owns (page-locked when it can get them)
    and takes a set again only when nothing the caller got from it is alive: an observation that is kept -- or a slice of it --
    is never overwritten by later steps, and a loop that drops what it got runs on two or three sets for ever."""
    fs = P.ieee13_like("epsilon")
    B = 64
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver="nr")
    env.reset(seed=5)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (12, B, fs.action_dim))
    kept, copies = [], []
    for k in range(6):                       # keep everything: more sets than the pool's cap, then fresh arrays
        obs, rew, *_ = env.step(acts[k])
        kept.append((obs[3:7], rew)); copies.append((obs[3:7].copy(), rew.copy()))
    for (o, r), (oc, rc) in zip(kept, copies):
        assert np.array_equal(o, oc) and np.array_equal(r, rc)
    addrs = {o.__array_interface__["data"][0] for o, _ in kept}
    assert len(addrs) == 6                    # six different buffers while all six are alive
    del kept, obs, rew, o, r
    seen = []
    for k in range(6, 12):                   # drop everything every step: the same few buffers come round again
        obs, *_ = env.step(acts[k])
        seen.append(obs.__array_interface__["data"][0])
        del obs
    assert len(set(seen)) <= 2
    # only the observation block is pooled: keeping every step's small arrays (rewards, info) does not hold its buffer back
    hoard, seen = [], []
    for k in range(6):
        obs, rew, term, trunc, info = env.step(acts[k])
        hoard.append((rew, info["iterations"], info["total_losses"], rew.copy(), info["total_losses"].copy()))
        seen.append(obs.__array_interface__["data"][0])
        del obs
    assert len(set(seen)) <= 2 and len({id(h[0]) for h in hoard}) == 6
    assert all(np.array_equal(h[0], h[3]) and np.array_equal(h[2], h[4]) for h in hoard)      # nothing kept was written again
    env.reset(seed=5); del hoard
    for k in range(12):
        o1, r1, *_ = env.step(acts[k]); del o1
    # the same trajectory with fresh arrays every step
    ref = P.BatchedGridEnvironment(fs, num_envs=B, solver="nr", recycle_host_buffers=False)
    ref.reset(seed=5)
    for k in range(12):
        o2, r2, *_ = ref.step(acts[k])
    o1, r1, *_ = env.step(acts[0]); o2, r2, *_ = ref.step(acts[0])
    assert np.array_equal(o1, o2) and np.array_equal(r1, r2)
    held = o1[:2]
    env.close(); ref.close()
    assert np.array_equal(held, o2[:2])       # an array held across close() stays readable


def test_recycled_observation_arrays_download_the_changing_columns_only_and_stay_whole():
    """A recycled page-locked observation array gets its constant columns (the static load powers, grid_env.py:769-770) once
    (gs_host_obs_bind); later steps move the changing columns only -- and return, bit for bit, what a whole-row download into a
    fresh array returns.  Also after the caller edited a returned array IN PLACE (the constant columns of that set are spoilt:
    the sampled check notices and the set is bound again), after a masked reset, and for download_step()."""
    fs = P.ieee123_like(); B = 200
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    ref = P.BatchedGridEnvironment(fs, num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, recycle_host_buffers=False)
    seeds = np.arange(B, dtype=np.uint64) + 3
    env.reset(seed=seeds); ref.reset(seed=seeds)
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (10, B, fs.action_dim))
    c0 = 2 * fs.n + 2 * fs.m + 1; c1 = c0 + 2 * fs.n_loads
    bound_seen = False
    for k in range(10):
        o1, r1, *_ = env.step(acts[k]); o2, r2, *_ = ref.step(acts[k])
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2), k
        assert (o1[:, c0:c1] != 0).any()
        st = env.handle._cur_set
        bound_seen |= st is not None and "probe" in st
        if k == 4:
            o1 *= 0.5                             # a caller normalising in place: the set's constant columns are gone
        if k == 6:
            m = (rng.random(B) < 0.4).astype(np.uint8)
            env.handle.reset(seeds + np.uint64(9), m, want_obs=False); ref.handle.reset(seeds + np.uint64(9), m, want_obs=False)
        del o1, r1
    assert bound_seen
    d1 = env.handle.download_step(); d2 = ref.handle.download_step()
    assert np.array_equal(d1["obs"], d2["obs"])
    env.close(); ref.close()


def test_float32_observations_are_the_float64_block_rounded_on_the_device():
    """obs_dtype=np.float32 (opt-in; the dtype the reference declares for its observation space, grid_env.py:346): step() and
    download_step() return the float64 observation block rounded to nearest, every other output unchanged; the arrays come from the
    same pool (an observation that is kept is not overwritten)."""
    fs = P.ieee123_like(); B = 77
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    e32 = P.BatchedGridEnvironment(fs, obs_dtype=np.float32, **kw)
    e64 = P.BatchedGridEnvironment(fs, **kw)
    seeds = np.arange(B, dtype=np.uint64) + 17
    o0, _ = e32.reset(seed=seeds); e64.reset(seed=seeds)
    assert o0.dtype == np.float64                      # reset() keeps the reference's values
    rng = np.random.default_rng(2)
    acts = rng.uniform(-1, 1, (4, B, fs.action_dim))
    kept = []
    for k in range(4):
        a32 = e32.step(acts[k]); a64 = e64.step(acts[k])
        assert a32[0].dtype == np.float32 and a32[0].shape == (B, fs.obs_dim)
        assert np.array_equal(a32[0], a64[0].astype(np.float32))
        assert np.array_equal(a32[1], a64[1]) and np.array_equal(a32[2], a64[2]) and np.array_equal(a32[3], a64[3])
        assert np.array_equal(a32[4]["total_losses"], a64[4]["total_losses"])
        kept.append((a32[0], a32[0].copy()))
    assert all(np.array_equal(a, c) for a, c in kept)
    d32 = e32.handle.download_step(); d64 = e64.handle.download_step()
    assert d32["obs"].dtype == np.float32 and np.array_equal(d32["obs"], d64["obs"].astype(np.float32))
    with pytest.raises(ValueError):
        P.BatchedGridEnvironment(fs, num_envs=4, obs_dtype=np.int32)
    e32.close(); e64.close()
