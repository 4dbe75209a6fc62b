"""GPU parity of the meshed Newton-Raphson member of the second-generation step kernels (gs_k_step_nr_mesh2: the block LU of
a feeder with a few loops as rows of lane items, accumulating messages in LDS; csrc/mesh_schedule.h, kernels_flow2.hip) --
the kind of network the reference's own IEEE123Bus generates (26 cycles, feeders/ieee_feeders.py:236-330) and solves densely
(np.linalg.solve, environments/power_flow.py:186-190).  Through the C ABI: against the NumPy oracle's environment (dense
Newton-Raphson, exact Jacobian) on seeded batches, against the first-generation sparse-LU kernel it replaces, under an
iteration cap, and for an instance's independence of the batch it runs in."""
import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from tests.test_gpu_env import _oracle_rollout

pytestmark = pytest.mark.gpu

MESHED = [(lambda: P.random_meshed(123, 26, seed=1), 44),      # the benchmark's feeder: 26 loops on 123 buses
          (lambda: P.random_meshed(60, 10, seed=2), 70),
          (lambda: P.random_meshed(20, 4, seed=1), 9),         # one partial workgroup more than a slab group's worth of them
          (lambda: P.random_meshed(123, 26, seed=5), 20)]      # a wider core: bigger message region, one workgroup per CU


def _env(spec, B, **kw):
    base = dict(num_envs=B, solver="nr", stochastic_loads=True, weather_variation=True, jacobian="exact", tolerance=1e-9, max_iterations=50)
    base.update(kw)
    return P.BatchedGridEnvironment(spec, **base)


@pytest.mark.parametrize("maker,B", MESHED)
def test_meshed_environment_steps_equal_the_oracle(maker, B):
    spec = maker()
    assert not spec.is_radial()
    T = 3
    rng = np.random.default_rng(91)
    actions = rng.uniform(-1, 1, (T, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 300
    t0 = 11.5 * 3600.0
    env = _env(spec, B, first_instance=500)
    d = env.handle.describe()
    assert d["kernel"] == "nr_mesh2", d
    env.reset(seed=seeds)
    st = env.get_state(); st[:, env.state_column("time")] = t0; env.set_state(st)
    cfg = dict(stochastic_loads=True, weather_variation=True, power_base=spec.base_power_va, solver="nr", tolerance=1e-9, max_iterations=50,
               jacobian_mode="exact", zero_z="open")
    ref = _oracle_rollout(spec, cfg, actions, seeds, first_instance=500, t0=t0)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(actions[t])
        assert info["power_flow_converged"].all()
        for b in range(B):
            o, r, te, tr, inf = ref[b][t]
            assert np.max(np.abs(obs[b] - o) / np.maximum(1.0, np.abs(o))) < 1e-8, (t, b, int(np.argmax(np.abs(obs[b] - o))))
            assert abs(rew[b] - r) <= 1e-7 * max(1.0, abs(r)) and bool(term[b]) == te and bool(trunc[b]) == tr
            assert int(info["iterations"][b]) == int(inf["iterations"]), (t, b)
            assert abs(info["total_losses"][b] - inf["total_losses"]) < 1e-8
    env.close()


@pytest.mark.parametrize("maker,B", MESHED[:3])
def test_meshed_member_equals_the_first_generation_sparse_lu_kernel(maker, B, monkeypatch):
    """Steps with stochastic loads and weather, the fused post-step checks, a masked reset, a checkpoint round trip and a device
    rollout with in-place resets: observations within 1e-10 of the slab-row sparse LU (the same pivots in the same order, sums
    associated differently), every discrete output equal."""
    from grid_fed_rl_gym_amd.safety import PostStepChecks
    spec = maker()
    kw = dict(episode_length=6)
    rng = np.random.default_rng(17)
    acts = rng.uniform(-1, 1, (5, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 40
    mask = (rng.random(B) < 0.3).astype(np.uint8)
    outs = {}
    for which in ("mesh2", "sparse_lu"):
        if which == "sparse_lu":
            monkeypatch.setenv("GS_NO_MESH2", "1")
        env = _env(spec, B, **kw)
        monkeypatch.delenv("GS_NO_MESH2", raising=False)
        assert env.handle.describe()["kernel"] == ("nr_mesh2" if which == "mesh2" else "nr_sparse_lu")
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
        outs[which] = got
        env.close()
    for k, (a, b) in enumerate(zip(outs["mesh2"], outs["sparse_lu"])):
        for q in a:
            x, y = np.asarray(a[q]), np.asarray(b[q])
            if x.dtype.kind == "f":
                assert np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)), initial=0.0) < 1e-10, (k, q)
            else:
                assert np.array_equal(x, y), (k, q)
    assert outs["mesh2"][-1]["n_terminal"] > 0


@pytest.mark.parametrize("cap", [1, 2])
def test_meshed_member_under_an_iteration_cap_matches_the_oracle(cap):
    spec = P.random_meshed(60, 10, seed=2); B, T = 21, 2
    rng = np.random.default_rng(11)
    actions = rng.uniform(-1, 1, (T, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 5
    env = _env(spec, B, max_iterations=cap, first_instance=40)
    assert env.handle.describe()["kernel"] == "nr_mesh2"
    env.reset(seed=seeds)
    cfg = dict(stochastic_loads=True, weather_variation=True, power_base=spec.base_power_va, solver="nr", tolerance=1e-9, max_iterations=cap,
               jacobian_mode="exact", zero_z="open")
    ref = _oracle_rollout(spec, cfg, actions, seeds, first_instance=40)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(actions[t])
        assert not info["power_flow_converged"].any() and (info["iterations"] == cap).all() and (info["status"] == 1).all()
        for b in range(B):
            o, r, te, tr, inf = ref[b][t]
            assert np.max(np.abs(obs[b] - o) / np.maximum(1.0, np.abs(o))) < 1e-9, (t, b)
    env.close()


def test_meshed_member_at_bench_size_is_independent_of_the_batch_and_solves_the_equations():
    """B = 8192 on the benchmark's feeder: every instance converged; the voltages satisfy S = V conj(Y V) with the oracle's dense
    Ybus at the injections the step built; and the same instances (global-index seeds) in two smaller handles give the same
    observations bit for bit."""
    from oracle import oracle_np as O
    spec = P.random_meshed(123, 26, seed=1); B = 8192
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (2, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 9
    env = _env(spec, B, tolerance=1e-8)
    d = env.handle.describe()
    assert d["kernel"] == "nr_mesh2" and d["workgroups"] == 1024
    env.reset(seed=seeds)
    for t in range(2):
        obs, rew, term, trunc, info = env.step(acts[t])
    assert info["power_flow_converged"].all() and info["iterations"].max() <= 6
    n = spec.n
    Y = O.admittance_matrix(n, spec.frm, spec.to, spec.r, spec.x)
    sol = env.last_solution()
    for b in (0, 77, 4095, 8191):
        V = sol["bus_voltages"][b] * np.exp(1j * sol["bus_angles"][b])
        S = V * np.conj(Y @ V)
        assert np.max(np.abs(S[1:].imag)) < 1e-7                      # Q_spec = 0 at every PQ bus
        assert abs(S.real.sum() - info["total_losses"][b]) < 1e-9
    parts = []
    for lo, hi in ((0, 5000), (5000, B)):
        e2 = _env(spec, hi - lo, tolerance=1e-8, first_instance=lo)
        e2.reset(seed=seeds[lo:hi])
        for t in range(2):
            o2, *_ = e2.step(acts[t, lo:hi])
        parts.append(o2); e2.close()
    np.testing.assert_array_equal(np.vstack(parts), obs)
    env.close()


@pytest.mark.parametrize("maker,B", MESHED[:2])
def test_flat_start_table_of_the_meshed_member_changes_nothing_beyond_rounding(maker, B, monkeypatch):
    """Iteration 0 of every solve starts from the flat start, where every pivot's D^-1, T and column blocks do not depend on the
    instance: the handle keeps them in a table (written once by the step kernel itself) and iteration 0 only carries the right-hand
    side through.  Against a handle that eliminates for itself every time (GS_NR_NO_FLAT=1): equal iteration counts and flags,
    observations within 1e-12."""
    spec = maker()
    rng = np.random.default_rng(23)
    acts = rng.uniform(-1, 1, (3, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 11
    outs = []
    for no_flat in (False, True):
        if no_flat:
            monkeypatch.setenv("GS_NR_NO_FLAT", "1")
        env = _env(spec, B)
        monkeypatch.delenv("GS_NR_NO_FLAT", raising=False)
        assert env.handle.describe()["kernel"] == "nr_mesh2"
        env.reset(seed=seeds)
        got = []
        for t in range(3):
            obs, rew, term, trunc, info = env.step(acts[t])
            got.append((obs.copy(), rew.copy(), info["iterations"].copy(), info["power_flow_converged"].copy(), info["total_losses"].copy()))
        outs.append(got); env.close()
    for (o1, r1, i1, c1, l1), (o2, r2, i2, c2, l2) in zip(*outs):
        assert np.array_equal(i1, i2) and np.array_equal(c1, c2) and c1.all()
        assert np.max(np.abs(o1 - o2) / np.maximum(1.0, np.abs(o2))) < 1e-12 and np.max(np.abs(l1 - l2)) < 1e-12


@pytest.mark.parametrize("B", [1, 8, 13])
def test_meshed_member_with_one_instance_a_full_workgroup_and_a_ragged_second_one(B):
    """B = 1 (seven idle lanes per sub-group), B = 8 (exactly one workgroup), B = 13 (a second workgroup with five instances): the
    same instances -- global-index seeds, `first_instance` -- as rows 0..B-1 of a larger batch, bit for bit."""
    spec = P.random_meshed(60, 10, seed=2)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (2, 40, spec.action_dim))
    seeds = np.arange(40, dtype=np.uint64) + 21
    big = _env(spec, 40); big.reset(seed=seeds)
    small = _env(spec, B); small.reset(seed=seeds[:B])
    assert small.handle.describe()["kernel"] == "nr_mesh2"
    for t in range(2):
        ob, rb, *_ = big.step(acts[t]); osm, rs, *_, info = small.step(acts[t, :B])
        assert info["power_flow_converged"].all()
        np.testing.assert_array_equal(osm, ob[:B]); np.testing.assert_array_equal(rs, rb[:B])
    big.close(); small.close()


def test_networks_the_meshed_member_does_not_take_say_why_and_run_on_the_kernel_they_ran_on_before(monkeypatch):
    """Eligibility (gs_describe "mesh2"), for handles whose linear solver is the sparse block LU: the ScalableFeeder-like graph has
    pivots with more than eight neighbours left at their elimination; the as-coded Jacobian; a PV bus; a denser 123-bus feeder
    needing more rows per wavefront than a lane holds T / s for; GS_NO_MESH2=1.  Every one of them runs the slab-row sparse LU as
    before, steps, and (exact Jacobian) converges."""
    import dataclasses
    base = P.random_meshed(40, 6, seed=2)
    bt = base.bus_type.copy(); bt[5] = 1
    cases = [(P.scalable_like(40, seed=3), dict(jacobian="exact"), "neighbours"),
             (base, dict(jacobian="as_coded"), "as-coded Jacobian"),
             (dataclasses.replace(base, bus_type=bt), dict(jacobian="exact"), "not a PQ bus"),
             (base, dict(jacobian="exact"), "GS_NO_MESH2")]
    for spec, kw, word in cases:
        if word == "GS_NO_MESH2":
            monkeypatch.setenv("GS_NO_MESH2", "1")
        env = P.BatchedGridEnvironment(spec, num_envs=9, solver="nr", linear_solver="sparse_lu", tolerance=1e-8, max_iterations=50, **kw)
        monkeypatch.delenv("GS_NO_MESH2", raising=False)
        d = env.handle.describe()
        assert d["kernel"] == "nr_sparse_lu" and word in d["mesh2"], d["mesh2"]
        env.reset(seed=3)
        obs, rew, term, trunc, info = env.step(np.zeros((9, spec.action_dim)))
        assert np.isfinite(obs).all()
        if kw.get("jacobian") == "exact":
            assert info["power_flow_converged"].all()
        env.close()


def _relabel(spec, perm):
    """The same network with bus i renamed perm[i] (the slack need not be bus 0: the matrix product of iteration 0 and the
    scheduler both index buses, not "non-slack buses")."""
    import dataclasses
    perm = np.asarray(perm)
    inv = np.argsort(perm)
    m = lambda a: perm[np.asarray(a)].astype(np.int32)
    return dataclasses.replace(spec, name=spec.name + "_relabelled", bus_type=spec.bus_type[inv].copy(), v_set=spec.v_set[inv].copy(),
                               frm=m(spec.frm), to=m(spec.to), load_bus=m(spec.load_bus), gen_bus=m(spec.gen_bus), bat_bus=m(spec.bat_bus),
                               bus_ids=[spec.bus_ids[i] for i in inv])


def test_meshed_member_with_the_slack_in_the_middle_and_with_devices_equals_the_oracle():
    """Iteration 0 is a matrix product x = W [P_spec; 1] over the NON-slack buses in bus order: with the slack renamed to bus 7 of 40,
    and with batteries and generators on the network (their powers enter P_spec), environment steps still equal the NumPy oracle's,
    iteration counts included; and they equal the same handle's without the flat-start shortcuts (GS_NR_NO_FLAT=1) to 1e-12."""
    import dataclasses, os
    base = P.random_meshed(40, 6, seed=2)
    perm = np.arange(40); perm[[0, 7]] = perm[[7, 0]]; perm[[3, 22]] = perm[[22, 3]]
    spec = _relabel(base, perm)
    spec = dataclasses.replace(spec, gen_bus=np.array([5, 31], dtype=np.int32), gen_kind=np.array([0, 1], dtype=np.int32), gen_cap=np.array([300e3, 200e3]),
                               gen_p0=np.array([0.2, 3.0]), gen_p1=np.array([1500.0, 12.0]), gen_p2=np.array([0.0, 25.0]),
                               bat_bus=np.array([11], dtype=np.int32), bat_cap=np.array([500e3]), bat_rating=np.array([100e3]), bat_eff=np.array([0.95]))
    assert int(np.flatnonzero(spec.bus_type == 2)[0]) == 7
    B, T = 19, 3
    rng = np.random.default_rng(8)
    actions = rng.uniform(-1, 1, (T, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 50
    cfg = dict(stochastic_loads=True, weather_variation=True, power_base=spec.base_power_va, solver="nr", tolerance=1e-9, max_iterations=50,
               jacobian_mode="exact", zero_z="open")
    ref = _oracle_rollout(spec, cfg, actions, seeds, first_instance=0)
    outs = []
    for no_flat in (False, True):
        if no_flat:
            os.environ["GS_NR_NO_FLAT"] = "1"
        try:
            env = _env(spec, B)
        finally:
            os.environ.pop("GS_NR_NO_FLAT", None)
        assert env.handle.describe()["kernel"] == "nr_mesh2"
        env.reset(seed=seeds)
        got = []
        for t in range(T):
            obs, rew, term, trunc, info = env.step(actions[t])
            assert info["power_flow_converged"].all()
            for b in range(B):
                o, r, te, tr, inf = ref[b][t]
                assert np.max(np.abs(obs[b] - o) / np.maximum(1.0, np.abs(o))) < 1e-8, (no_flat, t, b)
                assert int(info["iterations"][b]) == int(inf["iterations"])
            got.append(obs.copy())
        outs.append(got); env.close()
    for a, b in zip(*outs):
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-12


def test_meshed_member_beyond_128_buses_keeps_the_flat_start_table_path():
    """The matrix product of iteration 0 is laid out for at most 128 columns (n - 1 buses + the constant): a larger feeder the member
    is still eligible for runs iteration 0 by elimination with the handle's flat-start table, as before -- same answers as the slab-row
    sparse LU, equal iteration counts."""
    spec = P.random_meshed(170, 8, seed=3)
    B = 24
    rng = np.random.default_rng(4)
    acts = rng.uniform(-1, 1, (2, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 2
    outs = []
    import os
    for no_mesh in (False, True):
        if no_mesh:
            os.environ["GS_NO_MESH2"] = "1"
        try:
            env = _env(spec, B)
        finally:
            os.environ.pop("GS_NO_MESH2", None)
        d = env.handle.describe()
        if not no_mesh and d["kernel"] != "nr_mesh2":
            env.close()
            pytest.skip("this feeder is not eligible for the meshed member: " + d["mesh2"])
        env.reset(seed=seeds)
        got = []
        for t in range(2):
            obs, rew, term, trunc, info = env.step(acts[t])
            assert info["power_flow_converged"].all()
            got.append((obs.copy(), info["iterations"].copy()))
        outs.append(got); env.close()
    for (o1, i1), (o2, i2) in zip(*outs):
        assert np.array_equal(i1, i2) and np.max(np.abs(o1 - o2) / np.maximum(1.0, np.abs(o2))) < 1e-10
