"""The host-side schedule of the meshed Newton-Raphson step kernel (csrc/mesh_schedule.cpp, dumped through
gs_mesh_schedule_dump: no GPU needed) replayed in NumPy exactly as the kernel walks it -- per (wavefront, row, sub-group)
item, pull lists, accumulating messages in a model of the LDS region, T kept per item, back substitution through the x
slots -- and held against a dense solve of the oracle's exact Jacobian (the reference's np.linalg.solve,
environments/power_flow.py:186-190) and against the oracle's calculated injections."""
import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd import _lib
from oracle import oracle_np as O

F_PIVOT, F_NBR, F_SLACKPOS = 1, 2, 4
SLOT = 144          # bytes of a voltage slot at 8 instances per workgroup: (8 + 1) * 16


def _replay(spec, S, V, P_spec, rng):
    """One Newton step of ONE instance (lane 0) through the tables.  Returns (dx per bus, P_calc + j Q_calc per pivot bus, losses sum)."""
    n = spec.n
    base = S["region_base"]
    lds = np.zeros((base + S["region_bytes"]) // 8 + 64)
    lds[:] = np.nan                                              # whatever is read before it is written shows
    def rd(off, k=2): return lds[off // 8: off // 8 + k].copy()
    def wr(off, v): lds[off // 8: off // 8 + len(v)] = v
    for i in range(n):
        wr(i * SLOT, [V[i].real, V[i].imag])
    wr(n * SLOT, [0.0, 0.0]); wr((n + 1) * SLOT, [1.0, 0.0])     # the ZERO and ONE voltage slots
    U = S["unit_bytes"]
    zero, dummy = base + S["zero_off"], base + S["dummy_off"]
    for u in range(3): wr(zero + u * U, [0.0, 0.0])
    items, rowinfo, adj_off, adj_y = S["items"], S["rowinfo"], S["adj_off"], S["adj_y"]
    NW, NI = S["nw"], S["ni"]
    rows_by_level = {}
    for w in range(NW):
        last = -1
        for j in range(NI):
            lev = int(rowinfo[w, j, 0])
            if lev >= 0:
                assert lev >= last, "a wavefront's rows come in level order"
                last = lev
                rows_by_level.setdefault(lev, []).append((w, j))
    assert sorted(rows_by_level) == list(range(S["n_levels"]))

    def yv(it, nadj):
        acc = 0j
        for u in range(nadj):
            e, f = rd(int(adj_off[it["adj_ptr"] + u]))
            acc += complex(*adj_y[it["adj_ptr"] + u]) * complex(e, f)
        ek, fk = rd(int(it["vk_off"]))
        s = complex(ek, fk) * np.conj(acc)
        return s.real, s.imag

    # ---- mismatch pass ----
    scalc, losses = {}, 0.0
    for lev in rows_by_level:
        for (w, j) in rows_by_level[lev]:
            nadj = int(rowinfo[w, j, 2])
            for hv in range(8):
                it = items[w, j, hv]
                if it["flags"] & (F_PIVOT | F_SLACKPOS):
                    pc, qc = yv(it, nadj)
                    losses += pc
                    if it["flags"] & F_PIVOT:
                        scalc[int(it["bus"])] = complex(pc, qc)

    # ---- bottom-up ----
    T, s_reg = {}, {}
    seen_pivots = set()
    for lev in sorted(rows_by_level):
        rows = list(rows_by_level[lev]); rng.shuffle(rows)
        for (w, j) in rows:
            ri = rowinfo[w, j]
            g_row, ncq, nrw, ncl, nadj = int(ri[1]) & 255, (int(ri[1]) >> 8) & 255, (int(ri[1]) >> 16) & 255, (int(ri[1]) >> 24) & 255, int(ri[2])
            scratch = {}
            lane = [None] * 8
            for hv in range(8):                                   # everything before the exchange of D^-1 and s
                it = items[w, j, hv]; fl = int(it["flags"])
                ek, fk = rd(int(it["vk_off"])); ej, fj = rd(int(it["vj_off"]))
                pc, qc = yv(it, nadj) if int(it["bus"]) >= 0 else (0.0, 0.0)     # every lane of a group: the pivot bus's
                vm2 = ek * ek + fk * fk; vm = np.sqrt(vm2); rvk = 1.0 / vm; rvj = 1.0 / np.sqrt(ej * ej + fj * fj)
                G, B, Gd, Bd = float(it["ykj_g"]), float(it["ykj_b"]), float(it["ykk_g"]), float(it["ykk_b"])
                D = np.array([[-qc - vm2 * Bd, pc * rvk + vm * Gd], [pc - vm2 * Gd, qc * rvk - vm * Bd]])
                bus = int(it["bus"])
                r = np.array([(P_spec[bus] if bus >= 0 else 0.0) - pc, 0.0 - qc])
                a = ek * ej + fk * fj; bk = fk * ej - ek * fj; bj = -bk
                gs, gc = G * bk - B * a, G * a + B * bk
                Akj = np.array([[gs, gc * rvj], [-gc, gs * rvj]])
                gs2, gc2 = G * bj - B * a, G * a + B * bj
                Ajk = np.array([[gs2, gc2 * rvk], [-gc2, gs2 * rvk]])
                for u in range(ncq):
                    o = int(it["cq_in"][u]); c0, c1, q = rd(o), rd(o + U), rd(o + 2 * U)
                    D += np.array([c0, c1]); r += q                 # (every lane of the group: its lists are the pivot's)
                for u in range(nrw):
                    o = int(it["rw_in"][u]); Akj += np.array([rd(o), rd(o + U)])
                for u in range(ncl):
                    o = int(it["cl_in"][u]); Ajk += np.array([rd(o), rd(o + U)])
                lane[hv] = dict(it=it, fl=fl, D=D, r=r, Akj=Akj, Ajk=Ajk)
                if fl & F_PIVOT:
                    assert bus not in seen_pivots; seen_pivots.add(bus)
                    assert np.isfinite(D).all() and np.isfinite(r).all(), (bus, D, r)
            for hv in range(8):                                   # lane 0 of every group: inverse, s
                L = lane[hv]
                Dinv = np.linalg.inv(L["D"]); s = Dinv @ L["r"]
                L["Dinv"], L["s"] = Dinv, s
                scratch[("d", hv)] = (Dinv, s)
            for hv in range(8):
                L = lane[hv]; fl = L["fl"]
                hv0 = (fl >> 4) & 15
                Dinv, s = L["Dinv"], L["s"]                         # formed by the lane itself ...
                if int(L["it"]["bus"]) >= 0 and not fl & F_SLACKPOS:    # ... and equal to what the group's lane 0 formed
                    assert np.array_equal(Dinv, scratch[("d", hv0)][0]) and np.array_equal(s, scratch[("d", hv0)][1])
                L["Dg"], L["sg"] = Dinv, s
                L["T"] = Dinv @ L["Akj"]
                scratch[("t", hv)] = L["T"]
                T[(w, j, hv)] = L["T"]
                if fl & F_PIVOT: s_reg[(w, j, hv)] = s
            writes = []
            for hv in range(8):
                L = lane[hv]; it, fl = L["it"], L["fl"]
                hv0, t, g = (fl >> 4) & 15, (fl >> 8) & 15, (fl >> 12) & 15
                for t2 in range(g_row):
                    if g_row > 1 and t2 >= g:                     # (the kernel forms it from whatever it reads and sends it to the DUMMY slot)
                        assert int(it["mout"][t2]) == dummy
                        continue
                    Tt = scratch[("t", hv0 + t2)] if g_row > 1 else L["T"]
                    M = -(L["Ajk"] @ Tt)
                    o = int(it["mout"][t2])
                    if (fl >> (16 + t2)) & 1: M = M + np.array([rd(o), rd(o + U)])
                    writes.append((o, M[0])); writes.append((o + U, M[1]))
                    if t2 == t:
                        q = -(L["Ajk"] @ L["sg"])
                        if (fl >> (16 + t2)) & 1: q = q + rd(o + 2 * U)
                        writes.append((o + 2 * U, q))
            for o, v in writes:
                if o != dummy and o != dummy + U and o != dummy + 2 * U:
                    assert o >= base + S["body_off"], o
                wr(o, v)
    assert len(seen_pivots) == S["n_pivots"]

    # ---- back substitution: the x slots share the message body ----
    lds[(base + S["body_off"]) // 8:] = np.nan
    x = {}
    for lev in sorted(rows_by_level, reverse=True):
        for (w, j) in rows_by_level[lev]:
            g_row = int(rowinfo[w, j, 1]) & 255
            part = {}
            for hv in range(8):
                it = items[w, j, hv]
                xj = rd(int(it["xj_off"]))
                assert np.isfinite(xj).all(), ("x read before it was written", int(it["bus"]), int(it["nbr"]))
                part[hv] = T[(w, j, hv)] @ xj
            for hv in range(8):
                it = items[w, j, hv]; fl = int(it["flags"])
                if not fl & F_PIVOT: continue
                g = (fl >> 12) & 15
                tot = np.zeros(2)
                for t2 in range(g): tot = tot + part[hv + t2]
                xk = s_reg[(w, j, hv)] - tot
                wr(int(it["xk_off"]), xk); x[int(it["bus"])] = xk
    return x, scalc, losses


FEEDERS = [("loops26", lambda: P.random_meshed(123, 26)), ("radial123", lambda: P.ieee123_like()), ("ieee13", lambda: P.ieee13_like("epsilon")),
           ("loops10_n60", lambda: P.random_meshed(60, 10, seed=2)), ("loops26_seed5", lambda: P.random_meshed(123, 26, seed=5)),
           ("loops4_n20", lambda: P.random_meshed(20, 4, seed=1))]


@pytest.mark.parametrize("name,maker", FEEDERS)
@pytest.mark.parametrize("acc_cap,budget", [(4, 0), (1, 0), (4, 200), (2, 1)])
def test_replay_of_the_schedule_solves_the_newton_step(name, maker, acc_cap, budget):
    spec = maker()
    region_base = (spec.n + 3) * SLOT + 16
    region_base += (-region_base) % 16
    S = _lib.mesh_schedule(spec, nw=4, ni=16, acc_cap=acc_cap, region_base=region_base, slot_bytes=SLOT, unit_budget=budget)
    assert S["ok"], S["why"]
    S["region_base"] = region_base
    n = spec.n
    rng = np.random.default_rng(7)
    Y = O.admittance_matrix(n, spec.frm, spec.to, spec.r, spec.x)
    V = (1.0 + 0.05 * rng.standard_normal(n)) * np.exp(1j * 0.1 * rng.standard_normal(n)); V[0] = 1.0
    slack, pv, pq = O.classify(spec.bus_type)
    P_spec = 0.1 * rng.standard_normal(n)
    x, scalc, losses = _replay(spec, S, V, P_spec, rng)
    Sc = V * np.conj(Y @ V)
    for b, s in scalc.items():
        assert abs(s - Sc[b]) < 1e-11 * max(1.0, abs(Sc[b])), (b, s, Sc[b])
    assert abs(losses - Sc.real.sum()) < 1e-10
    J = O.jacobian(Y, V, slack, pv, pq, mode="exact")
    ns = [i for i in range(n) if i != slack]
    rhs = np.concatenate([[P_spec[i] - Sc[i].real for i in ns], [0.0 - Sc[i].imag for i in pq]])
    dx = np.linalg.solve(J, rhs)
    th = {b: q for q, b in enumerate(ns)}; vm = {b: len(ns) + q for q, b in enumerate(pq)}
    scale = np.abs(dx).max()
    for b in ns:
        assert abs(x[b][0] - dx[th[b]]) < 1e-9 * scale and abs(x[b][1] - dx[vm[b]]) < 1e-9 * scale, (b, x[b], dx[th[b]], dx[vm[b]])


def test_schedule_limits_and_refusals():
    """Pull lists never exceed the accumulator cap; a graph that fills in is refused with a reason, as is a row budget too small."""
    spec = P.random_meshed(123, 26)
    for cap in (1, 2, 4):
        S = _lib.mesh_schedule(spec, ni=16, acc_cap=cap)
        assert S["ok"]
        ri = S["rowinfo"].reshape(-1, 4)
        live = ri[:, 0] >= 0
        assert (((ri[live, 1] >> 8) & 255) <= cap).all() and (((ri[live, 1] >> 16) & 255) <= cap).all() and (((ri[live, 1] >> 24) & 255) <= cap).all()
    S = _lib.mesh_schedule(P.scalable_like(123, 1))
    assert not S["ok"] and "neighbours" in S["why"]
    S = _lib.mesh_schedule(spec, ni=4)
    assert not S["ok"] and "rows per wavefront" in S["why"]


@pytest.mark.parametrize("name,maker", FEEDERS)
def test_packed_items_say_what_the_verbose_items_say(name, maker):
    """The 16-word items the kernel reads (GS_MESH_W_*) against the verbose records the replay above walks: every offset as a slot /
    unit number, the Ybus pair of every neighbour lane, the diagonal entry, and the neighbour list of every pivot bus."""
    spec = maker(); n = spec.n
    base = (n + 3) * SLOT
    S = _lib.mesh_schedule(spec, nw=4, ni=16, acc_cap=4, region_base=base, slot_bytes=SLOT)
    assert S["ok"]
    U = S["unit_bytes"]
    Y = O.admittance_matrix(n, spec.frm, spec.to, spec.r, spec.x)
    unit = lambda off: (int(off) - base) // U
    ytab, ent, npairs = S["ytab"], S["adj_ent"], S["n_pairs"]
    for w in range(4):
        for j in range(16):
            ri, rp = S["rowinfo"][w, j], S["rowinfo_packed"][w, j]
            assert ri[0] == rp[0] and ri[1] == rp[1]
            for hv in range(8):
                it, p = S["items"][w, j, hv], S["packed"][w, j, hv]
                bus, nbr = int(p[0]) & 0xffff, (int(p[0]) >> 16) & 0xffff
                assert bus * SLOT == it["vk_off"] and nbr * SLOT == it["vj_off"]
                fl = int(p[1]) & 0xffffff
                assert fl == int(it["flags"])
                pair, cq = int(p[2]) & 0xffff, (int(p[2]) >> 16) & 0xffff
                assert cq == unit(it["cq_off"])
                assert tuple(ytab[pair]) == (float(it["ykj_g"]), float(it["ykj_b"]))
                dslot, ap = int(p[3]) & 0xffff, (int(p[3]) >> 16) & 0xffff
                nadj = (int(p[1]) >> 24) & 255
                if int(it["bus"]) >= 0:
                    assert dslot == bus and tuple(ytab[npairs + 1 + dslot]) == (Y[bus, bus].real, Y[bus, bus].imag)
                    got = sorted((int(e) >> 16, tuple(ytab[int(e) & 0xffff])) for e in ent[ap:ap + nadj])
                    want = sorted((k, (Y[bus, k].real, Y[bus, k].imag)) for k in range(n) if k != bus and Y[bus, k] != 0)
                    assert got == want and nadj <= rp[2]
                else:
                    assert nadj == 0 and tuple(ytab[npairs + 1 + dslot]) == (0.0, -1.0)
                if fl & F_PIVOT:
                    assert unit(it["xk_off"]) == 6 + bus
                if fl & F_NBR:
                    assert unit(it["xj_off"]) == 6 + nbr
                lists = [it["cq_in"], it["rw_in"], it["cl_in"]]
                for q in range(3):
                    for u in range(4):
                        word = int(p[4 + 2 * q + u // 2]); got = (word >> 16) & 0xffff if u & 1 else word & 0xffff
                        assert got == unit(lists[q][u])
                for t in range(8):
                    word = int(p[10 + t // 2]); got = (word >> 16) & 0xffff if t & 1 else word & 0xffff
                    assert got == unit(it["mout"][t])
    assert (int(ent[-1]) & 0xffff) == npairs and (int(ent[-1]) >> 16) == n          # the list's last entry: no branch, the ZERO slot


def test_level_search_lowers_the_peak_of_the_message_region():
    """Pivots delayed inside their windows (unit_budget > 0): the benchmark's feeder needs a quarter fewer message units, in the
    same number of levels; the packed items carry each group's exchange slot, at most four groups of two or more lanes per row."""
    spec = P.random_meshed(123, 26, seed=1)
    early = _lib.mesh_schedule(spec, ni=12, unit_budget=0)
    late = _lib.mesh_schedule(spec, ni=12, unit_budget=290)
    assert early["ok"] and late["ok"]
    assert late["msg_units"] <= 0.8 * early["msg_units"] and late["n_levels"] == early["n_levels"] and late["n_rows"] <= early["n_rows"]
    p = late["packed"].reshape(-1, 16)
    multi = ((p[:, 1] >> 12) & 15) > 1
    assert p[multi, 14].max() <= 3 and p[:, 14].min() >= 0
    rows = late["packed"].reshape(-1, 8, 16)
    for r in rows:                                      # the lanes of one group share a slot, different groups of a row do not
        g = {}
        for hv in range(8):
            if ((r[hv, 1] >> 12) & 15) > 1 and r[hv, 1] & 3:
                g.setdefault(int((r[hv, 1] >> 4) & 15), set()).add(int(r[hv, 14]))
        assert all(len(v) == 1 for v in g.values()) and len({next(iter(v)) for v in g.values()}) == len(g)


@pytest.mark.parametrize("maker", [lambda: P.random_meshed(123, 26, seed=1), lambda: P.random_meshed(40, 6, seed=2), lambda: P.ieee13_like("epsilon")])
def test_flat_start_newton_map_equals_the_oracles_first_newton_step(maker):
    """Iteration 0 of the meshed step kernel is x = W [P_spec; 1] (gs_flat_newton_map_dump: W inverted once on the host).  Against the
    NumPy oracle's own first Newton step from the flat start -- mismatch, exact Jacobian, np.linalg.solve (power_flow.py:125-190) -- for
    random injections, with the slack renamed into the middle of the numbering as well (no GPU)."""
    from oracle import oracle_np as O
    from grid_fed_rl_gym_amd import _lib
    import dataclasses
    base = maker()
    for relabel in (False, True):
        spec = base
        if relabel:
            perm = np.arange(base.n); perm[[0, base.n // 2]] = perm[[base.n // 2, 0]]
            inv = np.argsort(perm)
            m = lambda a: perm[np.asarray(a)].astype(np.int32)
            spec = dataclasses.replace(base, bus_type=base.bus_type[inv].copy(), v_set=base.v_set[inv].copy(), frm=m(base.frm), to=m(base.to),
                                       load_bus=m(base.load_bus), gen_bus=m(base.gen_bus), bat_bus=m(base.bat_bus), bus_ids=[base.bus_ids[i] for i in inv])
        W = _lib.flat_newton_map(spec)
        n = spec.n
        slack = int(np.flatnonzero(spec.bus_type == 2)[0])
        ns = [i for i in range(n) if i != slack]
        assert W.shape == (2 * (n - 1), n)
        Y = O.admittance_matrix(n, spec.frm, spec.to, spec.r, spec.x)
        V0 = np.ones(n, dtype=complex); V0[slack] = spec.v_set[slack]
        J = O.jacobian(Y, V0, slack, [], ns, mode="exact")
        rng = np.random.default_rng(7)
        for _ in range(3):
            Pn = np.zeros(n); Pn[ns] = -rng.uniform(0.0, 0.05, n - 1)
            S, dP, dQ, mm = O.mismatch(Y, V0, Pn, np.zeros(n), slack, ns)
            dx = np.linalg.solve(J, np.concatenate([dP[ns], dQ[ns]]))          # [d theta (non-slack); d|V| (non-slack)]
            x = W @ np.concatenate([Pn[ns], [1.0]])                            # (d theta, d|V|) interleaved per bus
            assert np.max(np.abs(x[0::2] - dx[:n - 1])) < 1e-11 and np.max(np.abs(x[1::2] - dx[n - 1:])) < 1e-11
