"""GPU parity for the load-flow kernels, through the C ABI.

Tier A: HIP Newton-Raphson with the Jacobian as coded, capped at 1..3 iterations, against the
numbers the reference itself produced (tests/golden/solve_*.npz, A* records).
Tier B: HIP NR with the exact Jacobian and HIP FBS against the reference-with-one-sign-fixed
converged answers (B*/C* records) and against the NumPy oracle on seeded batches.
Tolerances are written next to each comparison; the north-star bar is 1e-6 pu.
"""
import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd import _lib
from oracle import oracle_np as O
from tests.helpers import golden, golden_names, net_of

pytestmark = pytest.mark.gpu

SOLVE = golden_names("solve_")


def spec_of(d, name="g"):
    n, frm, to, r, x, rating, bt, vs = net_of(d)
    return P.FeederSpec(name=name, bus_ids=list(range(n)), bus_type=bt.astype(np.uint8), v_set=vs, frm=frm, to=to,
                        r=r, x=x, rating=rating)


def check(sol, b, d, pre, tol, flows_tol=None):
    assert bool(sol.converged[b]) == bool(d[pre + "converged"]), pre
    assert int(sol.iterations[b]) == int(d[pre + "iterations"]), pre
    for got, key in ((sol.bus_voltages[b], "Vm"), (sol.bus_angles[b], "Va"), (sol.line_flows[b], "flow"),
                     (sol.line_loadings[b], "loading")):
        ref = d[pre + key]
        scale = max(1.0, float(np.max(np.abs(ref)))) if len(ref) else 1.0
        t = tol if key in ("Vm", "Va") or flows_tol is None else flows_tol
        assert np.max(np.abs(got - ref), initial=0.0) <= t * scale, (pre, key, float(np.max(np.abs(got - ref))))
    assert abs(sol.losses[b] - float(d[pre + "losses"])) <= (flows_tol or tol) * max(1.0, abs(float(d[pre + "losses"])))


@pytest.mark.parametrize("name", SOLVE)
def test_tier_a_as_coded_iteration_caps(name):
    d = golden(name)
    spec = spec_of(d, name)
    growth = 1.0
    if name == "solve_noslack5":
        # No bus holds its magnitude here (the defaulted slack stays in the pq list, power_flow.py:138-141),
        # so the as-coded Jacobian is numerically singular (cond = 3.5e16 at the flat start): the
        # reference's step has an arbitrary component along "all Vm shift together".  Pinned instead:
        # flags, iteration count, angles, and magnitudes modulo that common shift.
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=1, jacobian="as_coded", zero_z="open")
        sol = s.solve_batch(spec, d["P_spec"][None, :])
        assert not sol.converged[0] and sol.iterations[0] == 1 and sol.status[0] == 1
        dv = sol.bus_voltages[0] - d["A1_Vm"]
        assert np.max(np.abs(dv - dv.mean())) < 1e-9 and abs(dv.mean()) < 1e-3
        assert np.max(np.abs(sol.bus_angles[0] - d["A1_Va"])) < 1e-9
        assert abs(sol.max_mismatch[0] - float(d["A1_max_mismatch"])) < 1e-12
        s.close()
        return
    for k in d["its"]:
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=int(k), jacobian="as_coded", zero_z="open")
        # B = 3 identical instances: also checks that lanes do not interfere
        sol = s.solve_batch(spec, np.tile(d["P_spec"], (3, 1)))
        growth = max(growth, float(d[f"A{k}_max_mismatch"]), float(np.max(np.abs(d[f"A{k}_Vm"]))))
        for b in range(3):
            check(sol, b, d, f"A{k}_", 1e-9 * growth ** 2)
            mmr = float(d[f"A{k}_max_mismatch"])
            assert abs(sol.max_mismatch[b] - mmr) <= 1e-9 * growth ** 2 * max(1.0, abs(mmr))
        s.close()


@pytest.mark.parametrize("name", SOLVE)
def test_tier_b_exact_converged(name):
    d = golden(name)
    if len(d["exact_scales"]) == 0:
        pytest.skip("no Tier-B record")
    spec = spec_of(d, name)
    s = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=50, jacobian="exact")
    P_batch = np.stack([d["P_spec"] * lam for lam in d["exact_scales"]])
    sol = s.solve_batch(spec, P_batch)
    for q in range(len(d["exact_scales"])):
        # same iterate sequence as the reference-with-sign-fixed => same iteration count, 1e-9 agreement
        check(sol, q, d, f"B{q}_", 1e-9)
        assert sol.converged[q] and sol.max_mismatch[q] < 1e-6 and sol.status[q] == 0
    s.close()
    # run to 1e-12: equals the tightly converged anchor
    s = P.BatchedNewtonRaphsonSolver(tolerance=1e-11, max_iterations=50, jacobian="exact")
    sol = s.solve_batch(spec, P_batch)
    for q in range(len(d["exact_scales"])):
        assert np.max(np.abs(sol.bus_voltages[q] - d[f"C{q}_Vm"])) < 1e-9
        assert np.max(np.abs(sol.bus_angles[q] - d[f"C{q}_Va"])) < 1e-9
        assert np.max(np.abs(sol.line_flows[q] - d[f"C{q}_flow"])) < 1e-8
    s.close()


@pytest.mark.parametrize("name", ["solve_env3", "solve_radial5", "solve_radial13", "solve_radial123",
                                  "solve_ieee13_eps", "solve_tree123"])
def test_fbs_matches_reference_anchor(name):
    d = golden(name)
    spec = spec_of(d, name)
    s = P.BatchedForwardBackwardSweepSolver(tolerance=1e-10, max_iterations=200)
    sol = s.solve_batch(spec, np.stack([d["P_spec"] * lam for lam in d["exact_scales"]]))
    for q in range(len(d["exact_scales"])):
        assert sol.converged[q], name
        assert np.max(np.abs(sol.bus_voltages[q] - d[f"C{q}_Vm"])) < 1e-8     # bar: 1e-6 pu
        assert np.max(np.abs(sol.bus_angles[q] - d[f"C{q}_Va"])) < 1e-8
        assert np.max(np.abs(sol.line_flows[q] - d[f"C{q}_flow"])) < 1e-7
    s.close()


def test_singular_as_coded_ieee13_and_zero_load():
    d = golden("solve_ieee13_as_coded")
    s = P.BatchedNewtonRaphsonSolver(max_iterations=3, jacobian="as_coded", zero_z="open")
    sol = s.solve_batch(spec_of(d), d["P_spec"][None, :])
    assert sol.status[0] == 2 and not sol.converged[0] and sol.iterations[0] == 1
    assert np.all(sol.bus_voltages[0] == 1.0) and abs(sol.max_mismatch[0] - 0.1155) < 1e-12
    s.close()
    d = golden("solve_env3_zero")
    s = P.BatchedNewtonRaphsonSolver(max_iterations=5, jacobian="as_coded")
    sol = s.solve_batch(spec_of(d), d["P_spec"][None, :])
    assert sol.converged[0] and sol.iterations[0] == 1 and np.all(sol.bus_voltages[0] == 1.0)
    s.close()


@pytest.mark.parametrize("maker,B,scale", [(lambda: P.ieee13_like("epsilon"), 200, 1.0),
                                           (lambda: P.ieee123_like(), 130, 1.0),
                                           (lambda: P.random_meshed(24, 9, seed=3), 70, 2.0)])
def test_seeded_batches_against_oracle(maker, B, scale):
    """Ragged batch sizes (not multiples of 64), per-instance loading, every instance checked."""
    fs = maker()
    rng = np.random.default_rng(1234)
    base = np.zeros(fs.n)
    np.add.at(base, fs.load_bus, -fs.load_base / 10e6)
    lam = rng.uniform(0.5, 1.5, B) * scale
    Pb = lam[:, None] * base[None, :] * rng.uniform(0.8, 1.2, (B, fs.n))
    s = P.BatchedNewtonRaphsonSolver(tolerance=1e-8, max_iterations=30, jacobian="exact")
    sol = s.solve_batch(fs, Pb)
    worst_v = worst_f = 0.0
    for b in range(0, B, 7):
        ref = O.nr_solve(fs.n, fs.frm, fs.to, fs.r, fs.x, fs.rating, fs.bus_type, fs.v_set, Pb[b], tolerance=1e-8,
                         max_iterations=30, jacobian_mode="exact")
        assert ref["converged"] and sol.converged[b] and sol.iterations[b] == ref["iterations"]
        worst_v = max(worst_v, np.max(np.abs(sol.bus_voltages[b] - ref["bus_voltages"])),
                      np.max(np.abs(sol.bus_angles[b] - ref["bus_angles"])))
        worst_f = max(worst_f, np.max(np.abs(sol.line_flows[b] - ref["line_flows"])))
        assert abs(sol.losses[b] - ref["losses"]) < 1e-9
    assert worst_v < 1e-9 and worst_f < 1e-8, (worst_v, worst_f)
    assert sol.converged.all()
    # size-independent property: the converged V satisfies the reference's own mismatch formula
    Y = O.admittance_matrix(fs.n, fs.frm, fs.to, fs.r, fs.x)
    V = sol.bus_voltages * np.exp(1j * sol.bus_angles)
    S = V * np.conj(V @ Y.T)
    resid = np.abs(S.real - Pb)[:, 1:].max()
    assert resid < 1e-7 and np.abs(S.imag[:, 1:]).max() < 1e-7
    s.close()


def test_reference_plug_point_signature_and_batch_helper():
    """solve(buses, lines, loads, generation) -> PowerFlowSolution, and parallel_power_flow_batch."""
    buses = [P.Bus(1, bus_type="slack"), P.Bus(2), P.Bus(3)]
    lines = [P.Line("line_1_2", 1, 2, 0.01, 0.02, 5e6), P.Line("line_2_3", 2, 3, 0.015, 0.025, 3e6)]
    d = golden("solve_env3")
    s = P.NewtonRaphsonSolver(max_iterations=1, jacobian="as_coded")
    sol = s.solve(buses, lines, {2: 0.2, 3: 0.15, "nope": 9.0}, {3: 0.05})
    assert isinstance(sol, P.PowerFlowSolution) and sol.iterations == 1 and not sol.converged
    assert np.max(np.abs(sol.bus_voltages - d["A1_Vm"])) < 1e-12
    assert abs(sol.losses - float(d["A1_losses"])) < 1e-12
    ex = P.NewtonRaphsonSolver(jacobian="exact")
    cfgs = [(buses, lines, {2: 0.2 * k, 3: 0.15 * k}, {3: 0.05 * k}) for k in (0.5, 1.0, 2.0)]
    res = P.parallel_power_flow_batch(ex, cfgs)
    for q, r in enumerate(res):
        assert r.converged and np.max(np.abs(r.bus_voltages - d[f"B{q}_Vm"])) < 1e-9
    s.close(); ex.close()


@pytest.mark.parametrize("name", ["solve_env3", "solve_radial13", "solve_tree123", "solve_meshed30", "solve_pv12", "solve_scal20", "solve_scal123"])
def test_linear_solver_paths_agree(name):
    """tree elimination, sparse block LU (blocks in slab rows, and blocks in the LDS of a one-wave workgroup per instance,
    kernels_sparse.hip), dense partial-pivot LU and the dense block LU on the matrix cores (one workgroup per instance,
    kernels_dense.hip) are five routes to the same Newton step: identical iteration counts, solutions within 1e-10."""
    d = golden(name)
    spec = spec_of(d, name)
    Pb = np.stack([d["P_spec"] * lam for lam in d["exact_scales"]])
    sols = {}
    for ls in ("tree", "sparse_lu", "sparse_lds", "dense_pivot", "dense_mfma"):
        if ls == "tree" and not spec.is_radial():
            continue
        if ls == "sparse_lds" and (spec.is_radial() or name == "solve_scal123" or not _lib.experiments()):
            continue                                     # (meshed networks only; the 123-bus dense graph's 8600 blocks do not fit LDS; an experiment: `make EXPERIMENTS=1`)
        if ls == "dense_pivot" and spec.n > 100 and not spec.is_radial():
            continue                                     # (the lane-per-instance pivoted LU on a 244 x 244 matrix: minutes)
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30, jacobian="exact", linear_solver=ls)
        sols[ls] = s.solve_batch(spec, Pb)
        if ls == "dense_mfma":
            assert s.handle_for(spec, len(Pb)).describe()["solve_kernel"] == "nr_dense_mfma"
        if ls == "sparse_lds":
            assert s.handle_for(spec, len(Pb)).describe()["solve_kernel"] == "nr_sparse_lds"
        s.close()
    base = sols.get("dense_pivot", sols["sparse_lu"])
    for ls, sol in sols.items():
        assert np.array_equal(sol.iterations, base.iterations), ls
        assert np.max(np.abs(sol.bus_voltages - base.bus_voltages)) < 1e-10, ls
        assert np.max(np.abs(sol.bus_angles - base.bus_angles)) < 1e-10, ls
        assert np.max(np.abs(sol.line_flows - base.line_flows)) < 1e-9, ls
        assert np.max(np.abs(sol.losses - base.losses)) < 1e-10 and np.max(np.abs(sol.max_mismatch - base.max_mismatch)) < 1e-10, ls
        assert sol.converged.all()


@pytest.mark.parametrize("name", ["solve_meshed30", "solve_scal20", "solve_tree123"])
def test_sparse_lu_flat_start_factors_shared_by_the_handle_change_nothing(name, monkeypatch):
    """The sparse block LU keeps the factors of the flat-start Jacobian (the same for every instance) as a table of scalars and
    iteration 0 only carries its right-hand side through them: bit for bit what every solve factoring for itself gives
    (GS_LU_NO_FLAT=1) -- the same blocks, the same operations in the same order."""
    d = golden(name)
    spec = spec_of(d)
    rng = np.random.default_rng(6)
    Pb = d["P_spec"][None, :] * rng.uniform(0.3, 1.5, (70, 1)) * (1.0 + 0.1 * rng.standard_normal((70, spec.n)))
    Pb[3] = 0.0
    outs = []
    for flag in (None, "1"):
        if flag:
            monkeypatch.setenv("GS_LU_NO_FLAT", flag)
        for cap in (1, 30):
            s = P.BatchedNewtonRaphsonSolver(tolerance=1e-10, max_iterations=cap, linear_solver="sparse_lu")
            outs.append(s.solve_batch(spec, Pb))
            s.close()
    monkeypatch.delenv("GS_LU_NO_FLAT")
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert np.array_equal(a.iterations, b.iterations) and np.array_equal(a.status, b.status)
        for f in ("bus_voltages", "bus_angles", "line_flows", "losses", "max_mismatch"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert outs[1].converged.all()


def test_dense_mfma_flat_start_factors_shared_by_the_handle_change_nothing(monkeypatch):
    """Iteration 0 of every solve starts from the flat start, where the Jacobian does not depend on the instance: its block
    factors are computed once per handle (by the solver kernel itself) and only substituted with.  The same iterates as every
    solve factoring for itself (GS_DENSE_NO_FLAT=1) up to the order of the additions in the forward substitution (1e-13)."""
    d = golden("solve_scal123")
    spec = spec_of(d)
    rng = np.random.default_rng(5)
    Pb = d["P_spec"][None, :] * rng.uniform(0.3, 1.6, (40, 1)) * (1.0 + 0.1 * rng.standard_normal((40, spec.n)))
    outs = []
    for flag in (None, "1"):
        if flag:
            monkeypatch.setenv("GS_DENSE_NO_FLAT", flag)
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-10, max_iterations=30, linear_solver="dense_mfma")
        outs.append(s.solve_batch(spec, Pb))
        s.close()
    monkeypatch.delenv("GS_DENSE_NO_FLAT")
    a, b = outs
    assert a.converged.all() and np.array_equal(a.iterations, b.iterations)
    for f, tol in (("bus_voltages", 1e-13), ("bus_angles", 1e-13), ("line_flows", 1e-11), ("losses", 1e-11), ("max_mismatch", 1e-11)):
        assert np.max(np.abs(getattr(a, f) - getattr(b, f))) < tol, f      # (flows: voltage differences times admittances of ~1e3)


@pytest.mark.parametrize("n,B", [(70, 1100), (123, 1030)])
def test_dense_block_row_form_with_three_and_four_block_rows_and_a_grid_smaller_than_the_batch(n, B):
    """The block-row form of the dense LU (one 64 x 64 block at a time through two LDS buffers, two workgroups per CU) with NB = 3 and
    NB = 4 block rows, more instances than the persistent grid has workgroups (2 x CUs) and a ragged last round: iterates of the
    sparse block LU on the same network, which eliminates in a different order (1e-10)."""
    fs = P.scalable_like(n, seed=5)
    rng = np.random.default_rng(n)
    Pd = -rng.uniform(0.0, 0.02, (B, fs.n)); Pd[:, 0] = 0.0
    outs = []
    for ls in ("dense_mfma", "sparse_lu"):
        q = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30, linear_solver=ls)
        outs.append(q.solve_batch(fs, Pd))
        d = q.handle_for(fs, B).describe()
        if ls == "dense_mfma":
            assert d["dense_form"] == "block_row" and d["dense_workgroups"] < B
        q.close()
    a, b = outs
    assert a.converged.all() and np.array_equal(a.iterations, b.iterations) and np.array_equal(a.status, b.status)
    assert np.max(np.abs(a.bus_voltages - b.bus_voltages)) < 1e-10 and np.max(np.abs(a.bus_angles - b.bus_angles)) < 1e-10
    assert np.max(np.abs(a.line_flows - b.line_flows)) < 1e-9


def test_dense_mfma_is_what_auto_takes_when_the_sparse_lu_fills_in_and_handles_the_edge_cases():
    """AUTO: the ScalableFeeder-like graph goes to the dense block LU on the matrix cores, the 26-loop feeder stays sparse.  Edge
    cases through the dense kernel: a ragged batch (more instances than workgroups of the persistent grid would be B > 256; here
    B = 70 with a zero-load instance that converges at the first check), the iteration cap (status 1, iterates equal to the
    other routes'), a non-finite injection (status 3) and PV buses."""
    dense, sparse = P.scalable_like(40, seed=3), P.random_meshed(40, 6, seed=2)
    s = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30)
    rng = np.random.default_rng(0)
    B = 70
    Pd = -rng.uniform(0.0, 0.03, (B, dense.n)); Pd[:, 0] = 0.0; Pd[5] = 0.0
    a = s.solve_batch(dense, Pd)
    d = s.handle_for(dense, B).describe()
    assert d["solve_kernel"] == "nr_dense_mfma" and d["dense_form"] == "block_row" and d["dense_workgroups"] == B
    s.solve_batch(sparse, np.zeros((2, sparse.n)))
    assert s.handle_for(sparse, 2).describe()["solve_kernel"] == "nr_sparse_lu"
    s.close()
    ref = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30, linear_solver="sparse_lu")
    b = ref.solve_batch(dense, Pd)
    ref.close()
    assert a.converged.all() and np.array_equal(a.iterations, b.iterations) and a.iterations[5] == 1
    assert np.all(a.bus_voltages[5] == 1.0) and np.max(np.abs(a.bus_voltages - b.bus_voltages)) < 1e-11
    assert np.max(np.abs(a.bus_angles - b.bus_angles)) < 1e-11 and np.max(np.abs(a.line_flows - b.line_flows)) < 1e-10
    for cap in (1, 2):
        outs = []
        for ls in ("dense_mfma", "sparse_lu"):
            q = P.BatchedNewtonRaphsonSolver(tolerance=1e-12, max_iterations=cap, linear_solver=ls)
            outs.append(q.solve_batch(dense, Pd[:9]))
            q.close()
        assert np.array_equal(outs[0].status, outs[1].status) and np.array_equal(outs[0].iterations, outs[1].iterations)
        assert np.max(np.abs(outs[0].bus_voltages - outs[1].bus_voltages)) < 1e-11
        assert np.max(np.abs(outs[0].max_mismatch - outs[1].max_mismatch)) < 1e-11
    bad = Pd[:4].copy(); bad[2, 7] = np.nan
    q = P.BatchedNewtonRaphsonSolver(linear_solver="dense_mfma")
    o = q.solve_batch(dense, bad)
    q.close()
    assert o.status[2] == 3 and not o.converged[2] and o.converged[[0, 1, 3]].all()
    d = golden("solve_pv12")
    spec = spec_of(d)
    q = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=50, linear_solver="dense_mfma")
    sol = q.solve_batch(spec, np.stack([d["P_spec"] * lam for lam in d["exact_scales"]]))
    q.close()
    for k in range(len(d["exact_scales"])):
        check(sol, k, d, f"B{k}_", 1e-9)


def test_sparse_lds_is_refused_with_a_reason_in_the_default_build():
    if _lib.experiments():
        pytest.skip("library built with EXPERIMENTS=1")
    q = P.BatchedNewtonRaphsonSolver(linear_solver="sparse_lds")
    with pytest.raises(P.PowerFlowError, match="EXPERIMENTS=1"):
        q.solve_batch(P.random_meshed(10, 3, seed=1), np.zeros((2, 10)))
    q.close()


@pytest.mark.skipif("not _lib.experiments()", reason="linear_solver sparse_lds is an experiment: `make EXPERIMENTS=1`")
def test_sparse_lu_in_lds_handles_the_edge_cases(monkeypatch):
    """linear_solver="sparse_lds": the sparse block LU with an instance's blocks in LDS, one wavefront per instance
    (kernels_sparse.hip; an alternative AUTO does not take).  Against the slab-row sparse LU: a ragged batch larger than the persistent grid's stride pattern, a zero-load instance (one iteration),
    iteration caps (status 1, same iterates), a non-finite injection (status 3), PV buses, and the handle's shared flat-start
    factors against every solve factoring for itself (GS_LU_NO_FLAT=1)."""
    fs = P.random_meshed(123, 26, seed=1)
    rng = np.random.default_rng(3)
    B = 333
    Pb = -rng.uniform(0.0, 0.004, (B, fs.n)); Pb[:, 0] = 0.0; Pb[7] = 0.0
    outs = {}
    for ls in ("sparse_lds", "auto"):
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30, linear_solver=ls)
        outs[ls] = s.solve_batch(fs, Pb)
        assert s.handle_for(fs, B).describe()["solve_kernel"] == ("nr_sparse_lds" if ls == "sparse_lds" else "nr_sparse_lu")
        s.close()
    a, b = outs["sparse_lds"], outs["auto"]
    assert a.converged.all() and np.array_equal(a.iterations, b.iterations) and a.iterations[7] == 1
    assert np.all(a.bus_voltages[7] == 1.0) and np.max(np.abs(a.bus_voltages - b.bus_voltages)) < 1e-11
    assert np.max(np.abs(a.bus_angles - b.bus_angles)) < 1e-11 and np.max(np.abs(a.line_flows - b.line_flows)) < 1e-10
    assert np.max(np.abs(a.losses - b.losses)) < 1e-11 and np.max(np.abs(a.max_mismatch - b.max_mismatch)) < 1e-11
    for cap in (1, 2):
        o = []
        for ls in ("sparse_lds", "sparse_lu"):
            q = P.BatchedNewtonRaphsonSolver(tolerance=1e-12, max_iterations=cap, linear_solver=ls)
            o.append(q.solve_batch(fs, Pb[:9]))
            q.close()
        assert np.array_equal(o[0].status, o[1].status) and np.array_equal(o[0].iterations, o[1].iterations)
        assert np.max(np.abs(o[0].bus_voltages - o[1].bus_voltages)) < 1e-11 and np.max(np.abs(o[0].max_mismatch - o[1].max_mismatch)) < 1e-11
    bad = Pb[:4].copy(); bad[2, 9] = np.nan
    q = P.BatchedNewtonRaphsonSolver(linear_solver="sparse_lds")
    o = q.solve_batch(fs, bad)
    q.close()
    assert o.status[2] == 3 and not o.converged[2] and o.converged[[0, 1, 3]].all()
    monkeypatch.setenv("GS_LU_NO_FLAT", "1")
    q = P.BatchedNewtonRaphsonSolver(tolerance=1e-9, max_iterations=30, linear_solver="sparse_lds")
    c = q.solve_batch(fs, Pb[:40])
    q.close()
    monkeypatch.delenv("GS_LU_NO_FLAT")
    assert np.array_equal(c.iterations, a.iterations[:40]) and np.max(np.abs(c.bus_voltages - a.bus_voltages[:40])) < 1e-13
    d = golden("solve_meshed30")
    spec = spec_of(d, "solve_meshed30")
    q = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=50, linear_solver="sparse_lds")
    sol = q.solve_batch(spec, np.stack([d["P_spec"] * lam for lam in d["exact_scales"]]))
    q.close()
    for k in range(len(d["exact_scales"])):
        check(sol, k, d, f"B{k}_", 1e-9)


def test_block_elimination_as_coded_where_its_pivots_are_regular():
    """The 2x2-block paths also reproduce the as-coded reference iterates when no diagonal
    block is singular (x = 2r chain): checks the J11 sign switch inside those kernels."""
    d = golden("solve_radial13")
    spec = spec_of(d)
    for ls in ("tree", "sparse_lu"):
        s = P.BatchedNewtonRaphsonSolver(tolerance=1e-6, max_iterations=2, jacobian="as_coded", linear_solver=ls)
        sol = s.solve_batch(spec, d["P_spec"][None, :])
        check(sol, 0, d, "A2_", 1e-9)
        s.close()


def test_topology_errors():
    fs = P.random_meshed(10, 3, seed=1)
    with pytest.raises(P.PowerFlowError, match="radial"):
        P.BatchedForwardBackwardSweepSolver().solve_batch(fs, np.zeros((1, 10)))
    with pytest.raises(P.PowerFlowError, match="loops"):
        P.BatchedNewtonRaphsonSolver(linear_solver="tree").solve_batch(fs, np.zeros((1, 10)))
    with pytest.raises(P.PowerFlowError, match="shape"):
        P.BatchedNewtonRaphsonSolver().solve_batch(fs, np.zeros((1, 9)))


def test_dataflow_sweeps_on_a_bus_with_many_children(monkeypatch):
    """A bus with 13 children (more than the 8 child slots of an item record: the rest go through the overflow list)
    and a zero-load batch instance (converges at the very first check): dataflow kernel = level-synchronous kernel =
    Newton-Raphson."""
    from grid_fed_rl_gym_amd.feeders import FeederSpec
    n = 20
    frm = [0] + [1] * 13 + [2, 15, 16, 3, 18]
    to = list(range(1, n))
    rng = np.random.default_rng(11)
    spec = FeederSpec(name="star", bus_ids=list(range(n)), bus_type=np.array([2] + [0] * (n - 1), dtype=np.uint8), v_set=np.ones(n),
                      frm=np.asarray(frm, dtype=np.int32), to=np.asarray(to, dtype=np.int32), r=rng.uniform(0.005, 0.02, n - 1),
                      x=rng.uniform(0.005, 0.03, n - 1), rating=np.full(n - 1, 5e6))
    B = 70
    P_spec = -rng.uniform(0.0, 0.04, (B, n)); P_spec[:, 0] = 0.0
    P_spec[3] = 0.0                                        # nothing to solve: stops at the flat start
    flow = P.BatchedForwardBackwardSweepSolver(tolerance=1e-10, max_iterations=100)
    a = flow.solve_batch(spec, P_spec)
    assert flow.handle_for(spec, B).describe()["solve_kernel"] == "fbs_flow"
    monkeypatch.setenv("GS_NO_FLOW", "1")
    sync = P.BatchedForwardBackwardSweepSolver(tolerance=1e-10, max_iterations=100)
    b = sync.solve_batch(spec, P_spec)
    monkeypatch.delenv("GS_NO_FLOW")
    assert sync.handle_for(spec, B).describe()["solve_kernel"] in ("fbs_lds", "fbs")
    nr = P.BatchedNewtonRaphsonSolver(tolerance=1e-12, max_iterations=50).solve_batch(spec, P_spec)
    assert a.converged.all() and b.converged.all() and nr.converged.all()
    assert np.array_equal(a.iterations, b.iterations)
    assert np.max(np.abs(a.bus_voltages - b.bus_voltages)) < 1e-13 and np.max(np.abs(a.bus_angles - b.bus_angles)) < 1e-13
    assert np.max(np.abs(a.bus_voltages - nr.bus_voltages)) < 1e-9 and np.max(np.abs(a.bus_angles - nr.bus_angles)) < 1e-9
    assert np.all(a.bus_voltages[3] == 1.0) and np.all(a.bus_angles[3] == 0.0) and np.all(a.line_flows[3] == 0.0)
    np.testing.assert_allclose(a.line_flows, b.line_flows, rtol=1e-10, atol=1e-14)
    flow.close(); sync.close()
