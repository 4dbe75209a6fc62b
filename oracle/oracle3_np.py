"""CPU oracle for the three-phase unbalanced load flow (NumPy, float64).  TEST INFRASTRUCTURE ONLY.

Parity status: the reference contains no three-phase solver, so there is no
reference output to pin against -- **parity unpinned by the reference**.  What anchors it:
(1) in the balanced, uncoupled limit each phase must equal the single-phase solution, which IS
pinned by the reference-derived fixtures (tests/test_unbalanced.py uses solve_tree123 / solve_radial13);
(2) the converged voltages must satisfy S = V conj(Y3 V) with Y3 assembled independently from the
3x3 line blocks (``residual``).
"""
import numpy as np

A120 = np.exp(1j * np.array([0.0, -2.0 * np.pi / 3.0, 2.0 * np.pi / 3.0]))


def _present(mask):
    return np.array([(mask >> ph) & 1 for ph in range(3)], dtype=bool)


def fbs3_solve(parent, phases, z, source, v_source, P, Q, tolerance=1e-6, max_iterations=100):
    """Ladder (forward/backward sweep) solution of one instance.  P, Q: [n, 3] net injections."""
    n = len(parent)
    kids = [[] for _ in range(n)]
    for i in range(n):
        if i != source:
            kids[parent[i]].append(i)
    order = [source]
    for u in order:
        order.extend(kids[u])
    S = np.asarray(P, dtype=float) + 1j * np.asarray(Q, dtype=float)
    pres = np.array([_present(int(m)) for m in phases])
    Vs = np.asarray(v_source, dtype=float) * A120
    V = np.where(pres, Vs[None, :], 0.0).astype(complex)
    Yb = np.zeros((n, 3, 3), dtype=complex)
    for i in range(n):
        if i == source:
            continue
        idx = np.nonzero(pres[i])[0]
        Yb[i][np.ix_(idx, idx)] = np.linalg.inv(z[i][np.ix_(idx, idx)])
    it, mm, conv, losses = 0, np.inf, False, 0.0
    for it in range(max_iterations):
        K = np.zeros((n, 3), dtype=complex)
        J = np.zeros((n, 3), dtype=complex)
        mm, losses = 0.0, 0.0
        for i in reversed(order[1:]):
            K[i] = Yb[i] @ (V[i] - V[parent[i]])
            sk = sum((K[c] for c in kids[i]), np.zeros(3, dtype=complex))
            sj = sum((J[c] for c in kids[i]), np.zeros(3, dtype=complex))
            scalc = V[i] * np.conj(K[i] - sk)
            d = np.where(pres[i], S[i] - scalc, 0.0)
            mm = max(mm, np.max(np.abs(d.real)), np.max(np.abs(d.imag)))
            losses += scalc[pres[i]].real.sum()
            iinj = np.where(pres[i], np.conj(S[i] / np.where(pres[i], V[i], 1.0)), 0.0)
            J[i] = -iinj + sj
        losses += (Vs * np.conj(-sum((K[c] for c in kids[source]), np.zeros(3, dtype=complex)))).real.sum()
        if mm < tolerance:
            conv = True
            break
        for i in order[1:]:
            V[i] = np.where(pres[i], V[parent[i]] - z[i] @ J[i], 0.0)
    return dict(converged=conv, iterations=it + 1, voltages=V, losses=losses, max_mismatch=mm)


def residual(parent, phases, z, source, V, P, Q):
    """max |S_spec - V conj(Y3 V)| over the present phases of the non-source nodes, with Y3 applied
    edge by edge from independently inverted line blocks (vectorised; works at thousands of nodes)."""
    n = len(parent)
    pres = ((np.asarray(phases)[:, None] >> np.arange(3)[None, :]) & 1).astype(bool)
    I = np.zeros((n, 3), dtype=complex)
    for i in range(n):
        if i == source:
            continue
        idx = np.nonzero(pres[i])[0]
        y = np.linalg.inv(z[i][np.ix_(idx, idx)])
        k = np.zeros(3, dtype=complex)
        k[idx] = y @ (V[i][idx] - V[parent[i]][idx])
        I[i] += k
        I[parent[i]] -= k
    Sc = V * np.conj(I)
    d = np.where(pres, (np.asarray(P) + 1j * np.asarray(Q)) - Sc, 0.0)
    d[source] = 0.0
    return float(max(np.max(np.abs(d.real)), np.max(np.abs(d.imag)))), float(Sc[pres].real.sum())
