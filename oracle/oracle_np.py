"""CPU oracle (NumPy, float64) for the batched AC power-flow env.step() path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker* for the HIP path: it may be
imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by
nothing in the product package (grid_fed_rl_gym_amd/ never imports oracle/).

It restates, array-in/array-out and one feeder instance at a time, the arithmetic of the
reference's transition function.  Every function cites the reference file:line it follows
(paths relative to /root/reference/grid_fed_rl/).  The restatement is deliberately dense and
literal (dense Ybus, dense Jacobian, LAPACK solve) -- i.e. it shares no algorithm with the
sparse, batch-innermost HIP kernels, so agreement between the two is meaningful.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which oracle/capture_golden.py produced by importing the reference in
the build container (Tier A: as coded; Tier B: reference solve() with the J11-diagonal sign
corrected in a test-only subclass).  See SURVEY.md section 8(c).

Conventions
-----------
* buses are indexed 0..n-1 in list order (the reference's bus_map, power_flow.py:54).
* bus_type codes: 0 = pq, 1 = pv, 2 = slack.
* ``jacobian`` is "as_coded" (reference, including the J11-diagonal sign at
  power_flow.py:248) or "exact" (true derivative).
* ``zero_z`` is "open" (reference: |z| <= 1e-12 -> y = 0, power_flow.py:63) or "epsilon"
  (|z| <= 1e-12 is replaced by z = 1e-4 + 1e-4j before inversion).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PQ, PV, SLACK = 0, 1, 2

STATUS_OK = 0          # converged
STATUS_MAX_ITER = 1    # ran out of iterations
STATUS_SINGULAR = 2    # linear solve failed (reference: LinAlgError -> break)
STATUS_NAN = 3         # non-finite mismatch


# --------------------------------------------------------------------------------------
# a2  admittance matrix
# --------------------------------------------------------------------------------------
def line_admittance(r: float, x: float, zero_z: str = "open") -> complex:
    """Series admittance of one line -- power_flow.py:62-63 (and :345-346)."""
    z = complex(r, x)
    if abs(z) > 1e-12:
        return 1.0 / z
    if zero_z == "epsilon":
        return 1.0 / complex(1e-4, 1e-4)
    return 0.0 + 0.0j


def admittance_matrix(n, frm, to, r, x, zero_z: str = "open") -> np.ndarray:
    """Dense Ybus, series elements only, accumulated in line order -- power_flow.py:48-73."""
    Y = np.zeros((n, n), dtype=complex)
    for k in range(len(frm)):
        i, j = int(frm[k]), int(to[k])
        y = line_admittance(r[k], x[k], zero_z)
        Y[i, j] -= y
        Y[j, i] -= y
        Y[i, i] += y
        Y[j, j] += y
    return Y


# --------------------------------------------------------------------------------------
# a3  bus classification
# --------------------------------------------------------------------------------------
def classify(bus_type: Sequence[int]) -> Tuple[int, List[int], List[int]]:
    """(slack index, pv list, pq list) -- power_flow.py:123-141.

    The *last* bus typed slack wins (the loop overwrites ``slack_bus``); when no bus is
    typed slack, index 0 becomes the slack for the P equations but stays in ``pq`` (it was
    appended there before the default was applied) -- reproduced as coded.
    """
    slack = None
    pv, pq = [], []
    for i, t in enumerate(bus_type):
        if t == SLACK:
            slack = i
        elif t == PV:
            pv.append(i)
        else:
            pq.append(i)
    if slack is None:
        slack = 0
    return slack, pv, pq


def initial_voltage(bus_type, v_set) -> np.ndarray:
    """Flat start; slack and pv buses at their set magnitude, angle 0 -- power_flow.py:103,128-136."""
    V = np.ones(len(bus_type), dtype=complex)
    for i, t in enumerate(bus_type):
        if t == SLACK or t == PV:
            V[i] = complex(v_set[i], 0.0)
    return V


# --------------------------------------------------------------------------------------
# a4  mismatch
# --------------------------------------------------------------------------------------
def mismatch(Y, V, P_spec, Q_spec, slack, pq):
    """S = V conj(Y V); dP on non-slack, dQ on pq; max |.| -- power_flow.py:150-168."""
    S = V * np.conj(Y @ V)
    n = len(V)
    dP = np.zeros(n)
    dQ = np.zeros(n)
    for i in range(n):
        if i != slack:
            dP[i] = P_spec[i] - S.real[i]
    for i in pq:
        dQ[i] = Q_spec[i] - S.imag[i]
    mm = max(np.max(np.abs(dP)), np.max(np.abs(dQ)))
    return S, dP, dQ, mm


# --------------------------------------------------------------------------------------
# a5  Jacobian
# --------------------------------------------------------------------------------------
def jacobian(Y, V, slack, pv, pq, mode: str = "as_coded") -> np.ndarray:
    """Polar Newton-Raphson Jacobian -- power_flow.py:213-295.

    Unknowns [theta(non-slack, ascending) ; Vm(pq, ascending)], rows [P(non-slack); Q(pq)].
    Built here from whole-matrix outer products instead of the reference's scalar double
    loops; entry formulas are the reference's (:247-287).
    """
    n = len(V)
    G, B = Y.real, Y.imag
    Vm, Va = np.abs(V), np.angle(V)
    d = Va[:, None] - Va[None, :]
    sn, cs = np.sin(d), np.cos(d)
    gs_bc = G * sn - B * cs            # G sin - B cos
    gc_bs = G * cs + B * sn            # G cos + B sin
    VV = Vm[:, None] * Vm[None, :]
    H = VV * gs_bc                     # dP/dtheta off-diagonals (:251)
    N = Vm[:, None] * gc_bs            # dP/dVm    off-diagonals (:263)
    M = -VV * gc_bs                    # dQ/dtheta off-diagonals (:274)
    L = Vm[:, None] * gs_bc            # dQ/dVm    off-diagonals (:287)
    idx = np.arange(n)
    Bd, Gd = np.diag(B), np.diag(G)
    sumH = (VV * gs_bc).sum(axis=1)                  # = Q_i
    sumN = (Vm[None, :] * gc_bs).sum(axis=1)         # = P_i / Vm_i
    sumM = (VV * gc_bs).sum(axis=1)                  # = P_i
    sumL = (Vm[None, :] * gs_bc).sum(axis=1)         # = Q_i / Vm_i
    if mode == "as_coded":
        H[idx, idx] = -sumH + Vm * Vm * Bd           # :247-248 (sign as coded)
    elif mode == "exact":
        H[idx, idx] = -sumH - Vm * Vm * Bd           # true d P_i / d theta_i
    else:
        raise ValueError(mode)
    N[idx, idx] = sumN + Vm * Gd                     # :259-260
    M[idx, idx] = sumM - Vm * Vm * Gd                # :270-271
    L[idx, idx] = sumL - Vm * Bd                     # :283-284
    ns = [i for i in range(n) if i != slack]
    if len(pq) > 0:
        return np.block([[H[np.ix_(ns, ns)], N[np.ix_(ns, pq)]],
                         [M[np.ix_(pq, ns)], L[np.ix_(pq, pq)]]])
    return H[np.ix_(ns, ns)]


# --------------------------------------------------------------------------------------
# a7  corrections
# --------------------------------------------------------------------------------------
def apply_corrections(dx, V, slack, pq, alpha: float = 1.0) -> None:
    """theta += a dtheta (non-slack) then Vm += a dVm (pq), via polar round trips -- power_flow.py:297-327."""
    n = len(V)
    ns = [i for i in range(n) if i != slack]
    dVa = dx[:len(ns)]
    dVm = dx[len(ns):]
    for k, i in enumerate(ns):
        V[i] = np.abs(V[i]) * np.exp(1j * (np.angle(V[i]) + alpha * dVa[k]))
    for k, i in enumerate(pq):
        if k < len(dVm):
            V[i] = (np.abs(V[i]) + alpha * dVm[k]) * np.exp(1j * np.angle(V[i]))


# --------------------------------------------------------------------------------------
# a8 / a9  line flows, losses
# --------------------------------------------------------------------------------------
def line_flows(V, frm, to, r, x, rating, zero_z: str = "open"):
    """P flow from->to and |S|/rating per line -- power_flow.py:329-358."""
    m = len(frm)
    flows = np.zeros(m)
    loadings = np.zeros(m)
    for k in range(m):
        i, j = int(frm[k]), int(to[k])
        y = line_admittance(r[k], x[k], zero_z)
        I = y * (V[i] - V[j])
        S = V[i] * np.conj(I)
        flows[k] = S.real
        loadings[k] = abs(S) / rating[k] if rating[k] > 0 else 0.0
    return flows, loadings


def total_losses(Y, V) -> float:
    """Re sum_i V_i conj((YV)_i) -- power_flow.py:198-200."""
    return float(np.sum(V * np.conj(Y @ V)).real)


# --------------------------------------------------------------------------------------
# a3-a9  full Newton-Raphson solve
# --------------------------------------------------------------------------------------
def nr_solve(n, frm, to, r, x, rating, bus_type, v_set, P_spec, Q_spec=None, *,
             tolerance: float = 1e-6, max_iterations: int = 50, alpha: float = 1.0,
             jacobian_mode: str = "as_coded", zero_z: str = "open") -> Dict:
    """NewtonRaphsonSolver.solve on dense arrays -- power_flow.py:89-211.

    ``iterations`` is last loop index + 1 (:204); ``max_mismatch`` is the value computed at
    the top of the last executed loop pass (:168), i.e. *before* that pass's update.
    """
    Y = admittance_matrix(n, frm, to, r, x, zero_z)
    slack, pv, pq = classify(bus_type)
    V = initial_voltage(bus_type, v_set)
    P_spec = np.asarray(P_spec, dtype=float)
    Q_spec = np.zeros(n) if Q_spec is None else np.asarray(Q_spec, dtype=float)
    converged = False
    it = 0
    mm = float("inf")
    status = STATUS_MAX_ITER
    for it in range(max_iterations):
        _, dP, dQ, mm = mismatch(Y, V, P_spec, Q_spec, slack, pq)
        if mm < tolerance:
            converged = True
            status = STATUS_OK
            break
        J = jacobian(Y, V, slack, pv, pq, jacobian_mode)
        rhs = np.concatenate([np.array([dP[i] for i in range(n) if i != slack]),
                              np.array([dQ[i] for i in pq])])
        try:
            dx = np.linalg.solve(J, rhs)
        except np.linalg.LinAlgError:
            status = STATUS_SINGULAR
            break
        apply_corrections(dx, V, slack, pq, alpha)
    flows, loadings = line_flows(V, frm, to, r, x, rating, zero_z)
    return dict(converged=converged, iterations=it + 1, bus_voltages=np.abs(V),
                bus_angles=np.angle(V), line_flows=flows, line_loadings=loadings,
                losses=total_losses(Y, V), max_mismatch=float(mm), status=status, V=V)


# --------------------------------------------------------------------------------------
# forward/backward sweep (NEW functionality -- the reference names it, README.md:187-197,
# but contains no implementation; parity for it is against nr_solve(jacobian="exact"))
# --------------------------------------------------------------------------------------
def fbs_solve(n, frm, to, r, x, rating, bus_type, v_set, P_spec, Q_spec=None, *,
              tolerance: float = 1e-6, max_iterations: int = 50, zero_z: str = "epsilon") -> Dict:
    """Current-injection forward/backward sweep on a radial feeder (constant-power buses).

    Convergence test: the power mismatch of power_flow.py:150-171, SUMMED over the buses --
    ``sum_i |dP_i| + sum_i |dQ_i| < tolerance / 2`` -- instead of its maximum (the factor 2 covers the second-order
    part of a flow's error, the change of the losses below the line: measured on the 13-bus feeder at 1.5 x loading the
    flows stopped 1.04e-6 pu off with the bare sum at 1e-6).  A converged answer still
    satisfies the reference's own acceptance criterion (the maximum is below the sum), and on a radial
    feeder the sum bounds the error of every line flow: a line carries the injections of its subtree,
    so its flow is off by at most the mismatches summed over that subtree.  The sweeps converge
    linearly and from one side (every bus's mismatch has the same sign), so with the maximum alone
    the head-of-feeder flows stopped up to n_loads x tolerance away from the converged solution
    (6.7e-6 pu at tolerance 1e-6 on the 123-bus feeder; Newton-Raphson, converging quadratically,
    stops ~1e-7 away at the same tolerance).  ``max_mismatch`` reports the maximum, as the reference's field does.
    """
    slack, pv, pq = classify(bus_type)
    if pv:
        raise ValueError("FBS handles pq buses only")
    Y = admittance_matrix(n, frm, to, r, x, zero_z)
    Q_spec = np.zeros(n) if Q_spec is None else np.asarray(Q_spec, dtype=float)
    adj: List[List[Tuple[int, complex]]] = [[] for _ in range(n)]
    for k in range(len(frm)):
        y = line_admittance(r[k], x[k], zero_z)
        adj[int(frm[k])].append((int(to[k]), y))
        adj[int(to[k])].append((int(frm[k]), y))
    parent = [-1] * n
    ypar = [0j] * n
    order = [slack]
    seen = {slack}
    for u in order:
        for v, y in adj[u]:
            if v not in seen:
                seen.add(v)
                parent[v] = u
                ypar[v] = y
                order.append(v)
            elif v != parent[u]:
                raise ValueError("network is not radial")
    V = initial_voltage(bus_type, v_set)
    S_spec = np.asarray(P_spec, dtype=float) + 1j * Q_spec
    converged, it, mm, status = False, 0, float("inf"), STATUS_MAX_ITER
    for it in range(max_iterations):
        _, dP, dQ, mm = mismatch(Y, V, P_spec, Q_spec, slack, pq)
        if 2.0 * float(np.sum(np.abs(dP)) + np.sum(np.abs(dQ))) < tolerance:
            converged, status = True, STATUS_OK
            break
        Iinj = np.conj(S_spec / V)           # current injected INTO the network at each bus
        Iinj[slack] = 0.0
        J = -Iinj.copy()                     # branch current parent->bus = -(injections downstream)
        for v in reversed(order[1:]):
            J[parent[v]] += J[v]
        for v in order[1:]:
            V[v] = V[parent[v]] - J[v] / ypar[v]
    flows, loadings = line_flows(V, frm, to, r, x, rating, zero_z)
    return dict(converged=converged, iterations=it + 1, bus_voltages=np.abs(V),
                bus_angles=np.angle(V), line_flows=flows, line_loadings=loadings,
                losses=total_losses(Y, V), max_mismatch=float(mm), status=status, V=V)


# --------------------------------------------------------------------------------------
# a12  solution quality score
# --------------------------------------------------------------------------------------
def quality_score(sol: Dict, tolerance: float) -> float:
    """_assess_solution_quality -- robust_power_flow.py:615-657."""
    if not sol["converged"]:
        return 0.0
    q = 1.0
    v = sol["bus_voltages"]
    if np.any(v < 0.8) or np.any(v > 1.2):
        q *= 0.3
    elif np.any(v < 0.9) or np.any(v > 1.1):
        q *= 0.7
    if len(sol["line_loadings"]) > 0:
        mx = np.max(sol["line_loadings"])
        if mx > 2.0:
            q *= 0.2
        elif mx > 1.0:
            q *= 0.5
    if sol["max_mismatch"] > tolerance * 100:
        q *= 0.6
    if sol["iterations"] <= 5:
        q *= 1.1
    elif sol["iterations"] > 20:
        q *= 0.9
    return min(1.0, q)


# --------------------------------------------------------------------------------------
# a15-a20  dynamics
# --------------------------------------------------------------------------------------
DAILY_PROFILE = np.array([0.5, 0.4, 0.4, 0.4, 0.4, 0.5, 0.7, 0.9, 0.8, 0.7, 0.6, 0.6,
                          0.7, 0.7, 0.6, 0.6, 0.7, 0.9, 1.0, 0.9, 0.8, 0.7, 0.6, 0.5])  # dynamics.py:43-48


def load_profile_power(time_s: float, base_power: float, noise: float = 0.0,
                       power_factor: float = 0.95, seasonal: float = 1.0):
    """TimeVaryingLoadModel.get_power with an explicit noise draw -- dynamics.py:54-75.

    ``noise`` is the realised N(0, noise_factor) sample (0.0 reproduces noise_factor=0).
    """
    hour = (time_s / 3600) % 24
    hi = int(hour)
    frac = hour - hi
    mult = DAILY_PROFILE[hi] * (1 - frac) + DAILY_PROFILE[(hi + 1) % 24] * frac
    mult *= (1 + noise)
    p = base_power * mult * seasonal
    q = p * math.tan(math.acos(power_factor))
    return max(0, p), q


def solar_power(time_s, cloud_cover, temperature, capacity, efficiency=0.18, panel_area=1000.0) -> float:
    """SolarPVModel.get_power -- dynamics.py:120-142."""
    hour = (time_s / 3600) % 24
    elev = math.sin(math.pi * (hour - 6) / 12) if 6 <= hour <= 18 else 0
    irr = 1000 * elev * (1 - 0.8 * cloud_cover)
    temp_factor = 1 - 0.004 * max(0, temperature - 25)
    return min(irr * panel_area * efficiency * temp_factor, capacity)


def wind_power(wind_speed, capacity, cut_in=3.0, rated=12.0, cut_out=25.0) -> float:
    """WindTurbineModel.get_power -- dynamics.py:158-170."""
    if wind_speed < cut_in or wind_speed > cut_out:
        return 0.0
    if wind_speed <= rated:
        return capacity * ((wind_speed - cut_in) / (rated - cut_in)) ** 3
    return capacity


def battery_update(soc, cur_power, cmd, dt, capacity, rating, eff):
    """GridDynamics.update_batteries for one battery -- dynamics.py:189-220, 304-324.

    cmd > 0 discharges, cmd < 0 charges, cmd == 0 leaves (soc, current_power) untouched.
    Returns (soc, current_power).
    """
    if cmd > 0:
        p = min(cmd, rating)
        e = min(p * dt / 3600, soc * capacity * eff)
        soc = soc - e / (capacity * eff)
        cur_power = e * 3600 / dt
    elif cmd < 0:
        p = min(-cmd, rating)
        max_e = (1.0 - soc) * capacity
        e = min(p * dt / 3600, max_e / eff)
        soc = soc + e * eff / capacity
        cur_power = -(e * 3600 / dt)
    return soc, cur_power


def frequency_update(freq, imbalance_mw, dt, H=5.0, D=1.0, f0=60.0) -> float:
    """GridDynamics.update_frequency -- dynamics.py:260-273."""
    dfdt = (imbalance_mw - D * (freq - f0)) / (2 * H * f0)
    freq = freq + dfdt * dt
    return max(55.0, min(65.0, freq))


# --------------------------------------------------------------------------------------
# counter-based RNG shared by the oracle and the HIP kernels (stochastic mode only).
# The reference draws from python ``random`` / ``np.random`` global streams
# (grid_env.py:669-681, dynamics.py:69); bit parity with those is impossible for a batched
# device implementation, so stochastic mode is defined on Philox4x32-10 (pinned by the Random123
# known answers) and validated against the reference's *distributions* (tests/test_stochastic.py).
# --------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85
_M32 = 0xFFFFFFFF


def philox4x32(counter: Tuple[int, int, int, int], key: Tuple[int, int]) -> Tuple[int, int, int, int]:
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        hi0, lo0 = (p0 >> 32) & _M32, p0 & _M32
        hi1, lo1 = (p1 >> 32) & _M32, p1 & _M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _M32, lo1, (hi0 ^ c3 ^ k1) & _M32, lo0
        k0 = (k0 + _PH_W0) & _M32
        k1 = (k1 + _PH_W1) & _M32
    return c0, c1, c2, c3


def rng_uniform_pair(seed: int, instance: int, step: int, draw: int) -> Tuple[float, float]:
    """Two uniforms in (0,1) from 53-bit mantissas; key=(seed lo, seed hi), counter=(instance, step, draw, 0)."""
    r = philox4x32((instance & _M32, step & _M32, draw & _M32, 0x47535450),
                   (seed & _M32, (seed >> 32) & _M32))
    u0 = (((r[0] << 32) | r[1]) >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0)
    u1 = (((r[2] << 32) | r[3]) >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0)
    return u0, u1


def rng_normal_quad(seed: int, instance: int, step: int, draw: int) -> Tuple[float, float, float, float]:
    """Four standard normals from ONE Philox call: every 32-bit output word is a uniform (r + 1/2) 2^-32, words (0, 1)
    and (2, 3) are one Box-Muller pair each (cosine, sine).  The load noise takes them four at a time (load l: draw
    DRAW_LOAD0 + l // 4, component l & 3), the weather takes wind / temperature / cloud from one call."""
    r = philox4x32((instance & _M32, step & _M32, draw & _M32, 0x47535450), (seed & _M32, (seed >> 32) & _M32))
    u = [(x + 0.5) * (1.0 / 4294967296.0) for x in r]
    ra, rb = math.sqrt(-2.0 * math.log(u[0])), math.sqrt(-2.0 * math.log(u[2]))
    return (ra * math.cos(2.0 * math.pi * u[1]), ra * math.sin(2.0 * math.pi * u[1]),
            rb * math.cos(2.0 * math.pi * u[3]), rb * math.sin(2.0 * math.pi * u[3]))


# draw indices (per instance, per step)
DRAW_IRRADIANCE, DRAW_WEATHER, DRAW_LOAD0 = 0, 1, 16


def next_episode_seed(seed: int, instance: int) -> int:
    """Seed of an instance's next episode when reset() is given none (gs_reset with seeds == NULL and the automatic
    resets of gs_rollout): the reference's reset(seed=None) leaves its global streams running (grid_env.py:366-369);
    here the stream "runs on" through one Philox call keyed by the old seed, counter (instance lo, instance hi, 0, 'RSED')."""
    r = philox4x32((instance & _M32, (instance >> 32) & _M32, 0, 0x52534544), (seed & _M32, (seed >> 32) & _M32))
    return (r[1] << 32) | r[0]


def rollout_random_actions(seed: int, instance: int, t: int, action_dim: int) -> np.ndarray:
    """Uniform actions in (-1, 1) of gs_rollout's GS_POLICY_RANDOM (the reference samples env.action_space,
    algorithms/base.py:280): Philox keyed by the policy seed, counter (instance, t, action // 4, 'ACTN'), word k of a
    call = action 4 q + k = 2 (r + 1/2) 2^-32 - 1."""
    a = np.empty(action_dim)
    for q in range((action_dim + 3) // 4):
        r = philox4x32((instance & _M32, t & _M32, q, 0x4143544E), (seed & _M32, (seed >> 32) & _M32))
        for k in range(4):
            if 4 * q + k < action_dim:
                a[4 * q + k] = 2.0 * ((r[k] + 0.5) * (1.0 / 4294967296.0)) - 1.0
    return a


# --------------------------------------------------------------------------------------
# a13-a24  environment
# --------------------------------------------------------------------------------------
@dataclass
class EnvSpec:
    """Flattened description of one feeder + its controllable devices (host-side SoA)."""
    n: int
    frm: np.ndarray
    to: np.ndarray
    r: np.ndarray
    x: np.ndarray
    rating: np.ndarray
    bus_type: np.ndarray
    v_set: np.ndarray
    load_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    load_base: np.ndarray = field(default_factory=lambda: np.zeros(0))
    load_pf: np.ndarray = field(default_factory=lambda: np.zeros(0))
    gen_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    gen_kind: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))   # 0 solar, 1 wind
    gen_cap: np.ndarray = field(default_factory=lambda: np.zeros(0))
    gen_p0: np.ndarray = field(default_factory=lambda: np.zeros(0))   # solar: efficiency ; wind: cut-in
    gen_p1: np.ndarray = field(default_factory=lambda: np.zeros(0))   # solar: panel area ; wind: rated
    gen_p2: np.ndarray = field(default_factory=lambda: np.zeros(0))   # wind: cut-out
    bat_bus: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    bat_cap: np.ndarray = field(default_factory=lambda: np.zeros(0))
    bat_rating: np.ndarray = field(default_factory=lambda: np.zeros(0))
    bat_eff: np.ndarray = field(default_factory=lambda: np.zeros(0))
    bat_soc0: np.ndarray = field(default_factory=lambda: np.zeros(0))
    # configuration
    timestep: float = 1.0
    episode_length: int = 86400
    v_lim: Tuple[float, float] = (0.95, 1.05)
    f_lim: Tuple[float, float] = (59.5, 60.5)
    safety_penalty: float = 100.0
    H: float = 5.0
    D: float = 1.0
    f0: float = 60.0
    stochastic_loads: bool = False
    weather_variation: bool = False
    power_base: float = 1.0          # injections are divided by this before the solve (1.0 = as coded)
    solver: str = "nr"               # "nr" | "fbs"
    tolerance: float = 1e-6
    max_iterations: int = 50
    alpha: float = 1.0
    jacobian_mode: str = "as_coded"
    zero_z: str = "open"

    @property
    def m(self): return len(self.frm)
    @property
    def L(self): return len(self.load_bus)
    @property
    def G(self): return len(self.gen_bus)
    @property
    def Bt(self): return len(self.bat_bus)
    @property
    def obs_dim(self): return 2 * self.n + 2 * self.m + 1 + 2 * self.L + self.G + 2 * self.Bt
    @property
    def action_dim(self): return self.Bt + self.G


@dataclass
class EnvState:
    """Mutable per-instance state (what the reference scatters over Bus/Line/Battery objects)."""
    time: float = 0.0
    step: int = 0
    episode_reward: float = 0.0
    violations: int = 0
    total_losses: float = 0.0
    freq: float = 60.0
    irradiance: float = 0.0
    wind: float = 5.0
    temp: float = 25.0
    cloud: float = 0.3
    Vm: Optional[np.ndarray] = None
    Va: Optional[np.ndarray] = None
    flow: Optional[np.ndarray] = None
    loading: Optional[np.ndarray] = None
    soc: Optional[np.ndarray] = None
    bat_power: Optional[np.ndarray] = None
    curtail: Optional[np.ndarray] = None
    seed: int = 0
    instance: int = 0


def _weather_update(spec: EnvSpec, st: EnvState) -> None:
    """_update_weather on the Philox stream -- grid_env.py:653-681."""
    if not spec.weather_variation:
        return
    hour = (st.time / 3600) % 24
    base = 1000 * math.sin(math.pi * (hour - 6) / 12) if 6 <= hour <= 18 else 0
    u, _ = rng_uniform_pair(st.seed, st.instance, st.step, DRAW_IRRADIANCE)
    st.irradiance = base * (0.8 + 0.4 * u)
    z = rng_normal_quad(st.seed, st.instance, st.step, DRAW_WEATHER)
    st.wind = max(0, min(30, st.wind + 0.5 * z[0]))
    st.temp = 25 + 10 * math.sin(2 * math.pi * (hour - 12) / 24) + 2 * z[1]
    st.cloud = max(0, min(1, st.cloud + 0.1 * z[2]))


def _renewable(spec: EnvSpec, st: EnvState, g: int) -> float:
    """dynamics.get_renewable_power -- dynamics.py:290-302 dispatching to :120 / :158."""
    if spec.gen_kind[g] == 0:
        return solar_power(st.time, st.cloud, st.temp, spec.gen_cap[g], spec.gen_p0[g], spec.gen_p1[g])
    return wind_power(st.wind, spec.gen_cap[g], spec.gen_p0[g], spec.gen_p1[g], spec.gen_p2[g])


def env_reset(spec: EnvSpec, seed: int = 0, instance: int = 0) -> Tuple[np.ndarray, EnvState]:
    """GridEnvironment.reset -- grid_env.py:360-408."""
    st = EnvState(seed=seed, instance=instance)
    st.Vm = np.ones(spec.n)
    st.Va = np.zeros(spec.n)
    st.flow = np.zeros(spec.m)
    st.loading = np.zeros(spec.m)
    st.soc = np.full(spec.Bt, 0.5)            # :398 (hard 0.5, not initial_soc)
    st.bat_power = np.zeros(spec.Bt)
    st.curtail = np.ones(spec.G)
    _weather_update(spec, st)                  # :402
    return env_observation(spec, st), st


def env_observation(spec: EnvSpec, st: EnvState) -> np.ndarray:
    """get_observation -- grid_env.py:753-783."""
    obs: List[float] = []
    for i in range(spec.n):
        obs.extend([st.Vm[i], st.Va[i]])
    for k in range(spec.m):
        obs.extend([st.flow[k], st.loading[k]])
    obs.append(st.freq)
    for l in range(spec.L):
        obs.extend([spec.load_base[l], spec.load_base[l] * np.tan(np.arccos(spec.load_pf[l]))])  # base.py:282-283
    for g in range(spec.G):
        obs.append(_renewable(spec, st, g))
    for b in range(spec.Bt):
        obs.extend([st.soc[b], st.bat_power[b]])
    return np.array(obs, dtype=float)


def env_injections(spec: EnvSpec, st: EnvState) -> Tuple[np.ndarray, np.ndarray]:
    """_calculate_power_injections as dense per-bus sums -- grid_env.py:683-720.

    Returns (load_sum[n], gen_sum[n]) in the reference's accumulation order: loads in list
    order then charging batteries; generators in dict order then discharging batteries.
    """
    load_sum = np.zeros(spec.n)
    gen_sum = np.zeros(spec.n)
    for l in range(spec.L):
        if spec.stochastic_loads:
            z = rng_normal_quad(st.seed, st.instance, st.step, DRAW_LOAD0 + l // 4)[l & 3]
            p, _ = load_profile_power(st.time, spec.load_base[l], noise=0.1 * z, power_factor=0.95)
        else:
            p = spec.load_base[l]
        load_sum[spec.load_bus[l]] += p
    for g in range(spec.G):
        gen_sum[spec.gen_bus[g]] += _renewable(spec, st, g) * st.curtail[g]
    for b in range(spec.Bt):
        if st.bat_power[b] > 0:
            gen_sum[spec.bat_bus[b]] += st.bat_power[b]
        elif st.bat_power[b] < 0:
            load_sum[spec.bat_bus[b]] += abs(st.bat_power[b])
    return load_sum, gen_sum


def env_step(spec: EnvSpec, st: EnvState, action: np.ndarray):
    """GridEnvironment.step with an explicit solver and all host hooks removed -- grid_env.py:410-619.

    Returns (obs, reward, terminated, truncated, info).
    """
    action = np.asarray(action, dtype=float).reshape(-1)
    # _apply_actions -- :621-651
    for b in range(spec.Bt):
        cmd = action[b] * spec.bat_rating[b] if b < len(action) else 0.0
        st.soc[b], st.bat_power[b] = battery_update(st.soc[b], st.bat_power[b], cmd, spec.timestep,
                                                    spec.bat_cap[b], spec.bat_rating[b], spec.bat_eff[b])
    for g in range(spec.G):
        j = spec.Bt + g
        st.curtail[g] = (action[j] + 1) / 2 if j < len(action) else 1.0
    st.time += spec.timestep                                    # :470
    st.step += 1                                                # :471
    _weather_update(spec, st)                                   # :474
    load_sum, gen_sum = env_injections(spec, st)                # :477
    P_spec = (0.0 - load_sum / spec.power_base) + gen_sum / spec.power_base   # power_flow.py:112-121
    solve = nr_solve if spec.solver == "nr" else fbs_solve
    kw = dict(tolerance=spec.tolerance, max_iterations=spec.max_iterations, zero_z=spec.zero_z)
    if spec.solver == "nr":
        kw.update(alpha=spec.alpha, jacobian_mode=spec.jacobian_mode)
    sol = solve(spec.n, spec.frm, spec.to, spec.r, spec.x, spec.rating, spec.bus_type, spec.v_set,
                P_spec, None, **kw)                             # :495
    # _update_grid_state -- :722-739, base.py:261-264
    st.Vm = sol["bus_voltages"].copy()
    st.Va = sol["bus_angles"].copy()
    st.flow = sol["line_flows"].copy()
    st.loading = np.array([abs(st.flow[k]) / spec.rating[k] if spec.rating[k] > 0 else 0.0
                           for k in range(spec.m)])
    st.total_losses += sol["losses"] * spec.timestep / 3600
    # _update_dynamics -- :741-751
    total_load = sum(float(p) for p in spec.load_base)
    total_gen = sum(_renewable(spec, st, g) for g in range(spec.G))
    imbalance = total_gen - total_load - sol["losses"] * spec.power_base
    st.freq = frequency_update(st.freq, imbalance / 1e6, spec.timestep, spec.H, spec.D, spec.f0)
    obs = env_observation(spec, st)                             # :559
    # get_reward -- :785-826
    reward = 0.0
    reward -= sum(abs(v - 1.0) for v in st.Vm) * 10
    reward -= abs(st.freq - 60.0) * 20
    reward -= sum(l > 0.8 for l in st.loading) * 50
    reward -= st.total_losses * 0.1
    ren = [_renewable(spec, st, g) for g in range(spec.G)]
    total_ren = sum(ren)
    total_curt = sum(ren[g] * (1 - st.curtail[g]) for g in range(spec.G))
    reward += (total_ren - total_curt) * 1e-5
    for b in range(spec.Bt):
        reward += 1.0 if 0.2 <= st.soc[b] <= 0.8 else -5.0
    terminated = st.step >= spec.episode_length                # base.py:140-142
    truncated = False
    viol = dict(voltage_high=bool(any(v > spec.v_lim[1] for v in st.Vm)),      # base.py:144-167
                voltage_low=bool(any(v < spec.v_lim[0] for v in st.Vm)),
                frequency_high=bool(st.freq > spec.f_lim[1]),
                frequency_low=bool(st.freq < spec.f_lim[0]))
    if any(viol.values()):
        st.violations += 1
        if st.violations > 10:                                  # :604-606
            truncated = True
            reward -= spec.safety_penalty
    st.episode_reward += reward
    info = dict(current_step=st.step, episode_reward=st.episode_reward,
                constraint_violations_count=st.violations, timestep=spec.timestep,
                power_flow_converged=sol["converged"], max_voltage=float(np.max(sol["bus_voltages"])),
                min_voltage=float(np.min(sol["bus_voltages"])), total_losses=sol["losses"],
                constraint_violations=viol, iterations=sol["iterations"],
                max_mismatch=sol["max_mismatch"], status=sol["status"])
    return obs, float(reward), bool(terminated), bool(truncated), info
