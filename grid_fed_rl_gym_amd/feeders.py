"""Network data for the batched path: flattening of feeder objects and seeded generators.

``FeederSpec`` is the structure-of-arrays form every batched entry point consumes.  It is
produced either by ``flatten_feeder`` from any object exposing the reference's
``BaseFeeder`` surface (``.buses``, ``.lines``, ``.loads``, ``.generators`` --
reference feeders/base.py:44-47; duck-typed, so the reference's own feeders work) or by
the seeded generators below, which produce *solvable* stand-ins for the reference's IEEE
feeders (the shipped ones are not: SURVEY.md fact F5).
"""
from __future__ import annotations

import hashlib
import json
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .components import Bus, Line, Load

PQ, PV, SLACK = 0, 1, 2
_BUS_TYPE_CODE = {"pq": PQ, "pv": PV, "slack": SLACK}
GEN_SOLAR, GEN_WIND = 0, 1


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


@dataclass
class FeederSpec:
    """One feeder topology + its devices as flat arrays (bus references are 0-based indices)."""
    name: str
    bus_ids: List[Any]
    bus_type: np.ndarray            # u8  [n]  0 pq, 1 pv, 2 slack
    v_set: np.ndarray               # f64 [n]  magnitude set-point of slack / pv buses
    frm: np.ndarray                 # i32 [m]
    to: np.ndarray                  # i32 [m]
    r: np.ndarray                   # f64 [m]  per unit
    x: np.ndarray                   # f64 [m]  per unit
    rating: np.ndarray              # f64 [m]  VA (as the reference stores it)
    load_bus: np.ndarray = field(default_factory=lambda: _i32([]))
    load_base: np.ndarray = field(default_factory=lambda: _f64([]))      # W
    load_pf: np.ndarray = field(default_factory=lambda: _f64([]))
    gen_bus: np.ndarray = field(default_factory=lambda: _i32([]))
    gen_kind: np.ndarray = field(default_factory=lambda: _i32([]))       # 0 solar, 1 wind
    gen_cap: np.ndarray = field(default_factory=lambda: _f64([]))        # W
    gen_p0: np.ndarray = field(default_factory=lambda: _f64([]))         # solar efficiency | wind cut-in
    gen_p1: np.ndarray = field(default_factory=lambda: _f64([]))         # solar panel area | wind rated
    gen_p2: np.ndarray = field(default_factory=lambda: _f64([]))         # -                | wind cut-out
    bat_bus: np.ndarray = field(default_factory=lambda: _i32([]))
    bat_cap: np.ndarray = field(default_factory=lambda: _f64([]))        # energy units of rating*h (dynamics.py:178)
    bat_rating: np.ndarray = field(default_factory=lambda: _f64([]))     # power units of the loads (dynamics.py:179)
    bat_eff: np.ndarray = field(default_factory=lambda: _f64([]))
    base_power_va: float = 10e6

    # sizes -----------------------------------------------------------------------------
    @property
    def n(self) -> int: return int(len(self.bus_type))
    @property
    def m(self) -> int: return int(len(self.frm))
    @property
    def n_loads(self) -> int: return int(len(self.load_bus))
    @property
    def n_gens(self) -> int: return int(len(self.gen_bus))
    @property
    def n_bats(self) -> int: return int(len(self.bat_bus))
    @property
    def obs_dim(self) -> int:
        """Reference layout grid_env.py:307-314."""
        return 2 * self.n + 2 * self.m + 1 + 2 * self.n_loads + self.n_gens + 2 * self.n_bats
    @property
    def action_dim(self) -> int:
        """Reference layout grid_env.py:351."""
        return self.n_bats + self.n_gens

    def bus_index(self) -> Dict[Any, int]:
        return {bid: i for i, bid in enumerate(self.bus_ids)}

    def sha256(self) -> str:
        """Digest over every numeric array -- recorded in fixtures and bench output."""
        h = hashlib.sha256()
        for a in (self.bus_type, self.v_set, self.frm, self.to, self.r, self.x, self.rating,
                  self.load_bus, self.load_base, self.load_pf, self.gen_bus, self.gen_kind,
                  self.gen_cap, self.gen_p0, self.gen_p1, self.gen_p2, self.bat_bus,
                  self.bat_cap, self.bat_rating, self.bat_eff):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    def is_radial(self) -> bool:
        """True when the bus graph (parallel lines merged) is a tree spanning all buses."""
        n = self.n
        pairs = {(min(int(a), int(b)), max(int(a), int(b))) for a, b in zip(self.frm, self.to) if a != b}
        if len(pairs) != n - 1:
            return False
        parent = list(range(n))
        def find(u):
            while parent[u] != u:
                parent[u] = parent[parent[u]]
                u = parent[u]
            return u
        for a, b in pairs:
            ra, rb = find(a), find(b)
            if ra == rb:
                return False
            parent[ra] = rb
        return True


# ----------------------------------------------------------------------------------------
# flattening
# ----------------------------------------------------------------------------------------
def flatten_network(buses: Sequence[Any], lines: Sequence[Any]):
    """(bus_ids, bus_type, v_set, frm, to, r, x, rating) from Bus/Line-shaped objects.

    Index = list position, exactly the reference's ``bus_map`` (power_flow.py:54).  A line
    naming an unknown bus id raises KeyError, as ``bus_map[line.from_bus]`` does (:58).
    """
    bus_ids = [b.id for b in buses]
    idx = {bid: i for i, bid in enumerate(bus_ids)}
    bus_type = np.array([_BUS_TYPE_CODE.get(getattr(b, "bus_type", "pq"), PQ) for b in buses], dtype=np.uint8)
    v_set = _f64([getattr(b, "voltage_magnitude", 1.0) for b in buses])
    frm = _i32([idx[l.from_bus] for l in lines])
    to = _i32([idx[l.to_bus] for l in lines])
    r = _f64([l.resistance for l in lines])
    x = _f64([l.reactance for l in lines])
    rating = _f64([l.rating for l in lines])
    return bus_ids, bus_type, v_set, frm, to, r, x, rating


def flatten_feeder(feeder: Any, batteries: Optional[Iterable[Dict[str, Any]]] = None,
                   name: Optional[str] = None) -> FeederSpec:
    """FeederSpec from a BaseFeeder-shaped object.

    ``feeder.generators`` entries of type "solar"/"wind" become renewable sources; entries
    of type "battery" (the reference's IEEE123 stores them there, ieee_feeders.py:371-379)
    and the optional ``batteries`` iterable become storage units.  Loads or devices naming
    a bus id that is not in the feeder are dropped, matching the solver's silent skip
    (power_flow.py:113,119).
    """
    bus_ids, bus_type, v_set, frm, to, r, x, rating = flatten_network(feeder.buses, feeder.lines)
    idx = {bid: i for i, bid in enumerate(bus_ids)}
    lb, lp, lpf = [], [], []
    for ld in getattr(feeder, "loads", []):
        if ld.bus in idx:
            lb.append(idx[ld.bus]); lp.append(ld.base_power); lpf.append(ld.power_factor)
    gb, gk, gc, g0, g1, g2 = [], [], [], [], [], []
    bb, bc, br, be = [], [], [], []
    gens = getattr(feeder, "generators", {}) or {}
    for info in (gens.values() if isinstance(gens, dict) else gens):
        kind = info.get("type")
        if info.get("bus") not in idx:
            continue
        if kind == "solar":
            eff = float(info.get("efficiency", 0.18))
            cap = float(info["capacity"])
            gb.append(idx[info["bus"]]); gk.append(GEN_SOLAR); gc.append(cap)
            g0.append(eff); g1.append(float(info.get("panel_area", cap / (eff * 1000.0)))); g2.append(0.0)
        elif kind == "wind":
            gb.append(idx[info["bus"]]); gk.append(GEN_WIND); gc.append(float(info["capacity"]))
            g0.append(float(info.get("cut_in_speed", 3.0))); g1.append(float(info.get("rated_speed", 12.0)))
            g2.append(float(info.get("cut_out_speed", 25.0)))
        elif kind == "battery":
            bb.append(idx[info["bus"]]); bc.append(float(info.get("capacity_kwh", info.get("capacity", 1e3))))
            br.append(float(info.get("power_rating_kw", info.get("power_rating", 0.5e6))))
            be.append(float(info.get("efficiency", 0.95)))
    for info in (batteries or []):
        if info.get("bus") in idx:
            bb.append(idx[info["bus"]]); bc.append(float(info["capacity"]))
            br.append(float(info["power_rating"])); be.append(float(info.get("efficiency", 0.95)))
    params = getattr(feeder, "parameters", None)
    base_va = float(getattr(params, "base_power", 10.0)) * 1e6 if params is not None else 10e6
    return FeederSpec(name=name or getattr(feeder, "name", "feeder"), bus_ids=bus_ids,
                      bus_type=bus_type, v_set=v_set, frm=frm, to=to, r=r, x=x, rating=rating,
                      load_bus=_i32(lb), load_base=_f64(lp), load_pf=_f64(lpf),
                      gen_bus=_i32(gb), gen_kind=_i32(gk), gen_cap=_f64(gc), gen_p0=_f64(g0),
                      gen_p1=_f64(g1), gen_p2=_f64(g2), bat_bus=_i32(bb), bat_cap=_f64(bc),
                      bat_rating=_f64(br), bat_eff=_f64(be), base_power_va=base_va)


def to_objects(spec: FeederSpec) -> Tuple[List[Bus], List[Line], List[Load]]:
    """Inverse of flattening: Bus/Line/Load lists (for handing a generated feeder to code
    written against the reference's object API)."""
    names = {PQ: "pq", PV: "pv", SLACK: "slack"}
    buses = []
    for i, bid in enumerate(spec.bus_ids):
        b = Bus(bid, bus_type=names[int(spec.bus_type[i])])
        b.voltage_magnitude = float(spec.v_set[i])
        buses.append(b)
    lines = [Line(f"line_{k}", spec.bus_ids[int(spec.frm[k])], spec.bus_ids[int(spec.to[k])],
                  float(spec.r[k]), float(spec.x[k]), float(spec.rating[k])) for k in range(spec.m)]
    loads = [Load(f"load_{l}", spec.bus_ids[int(spec.load_bus[l])], float(spec.load_base[l]),
                  float(spec.load_pf[l])) for l in range(spec.n_loads)]
    return buses, lines, loads


# ----------------------------------------------------------------------------------------
# generators
# ----------------------------------------------------------------------------------------
def reference_env_network() -> FeederSpec:
    """The 3-bus network GridEnvironment hard-codes whatever feeder it is given
    (reference grid_env.py:243-298): 2 lines, 2 loads, one battery injecting at bus id 2."""
    return FeederSpec(
        name="reference_env_3bus", bus_ids=[1, 2, 3],
        bus_type=np.array([SLACK, PQ, PQ], dtype=np.uint8), v_set=_f64([1, 1, 1]),
        frm=_i32([0, 1]), to=_i32([1, 2]), r=_f64([0.01, 0.015]), x=_f64([0.02, 0.025]),
        rating=_f64([5e6, 3e6]),
        load_bus=_i32([1, 2]), load_base=_f64([2e6, 1.5e6]), load_pf=_f64([0.95, 0.95]),
        bat_bus=_i32([1]), bat_cap=_f64([1e3]), bat_rating=_f64([0.5e6]), bat_eff=_f64([0.95]),
        base_power_va=10e6)


def with_reference_env_renewables(spec: FeederSpec, sources: Sequence[str]) -> FeederSpec:
    """Adds the hard-coded solar_2 / wind_3 units of grid_env.py:273-289 (dict order: solar, wind)."""
    gb, gk, gc, g0, g1, g2 = [], [], [], [], [], []
    if "solar" in sources:
        gb.append(1); gk.append(GEN_SOLAR); gc.append(1e6); g0.append(0.18); g1.append(5556.0); g2.append(0.0)
    if "wind" in sources:
        gb.append(2); gk.append(GEN_WIND); gc.append(2e6); g0.append(3.0); g1.append(12.0); g2.append(25.0)
    spec.gen_bus, spec.gen_kind, spec.gen_cap = _i32(gb), _i32(gk), _f64(gc)
    spec.gen_p0, spec.gen_p1, spec.gen_p2 = _f64(g0), _f64(g1), _f64(g2)
    return spec


def simple_radial(num_buses: int = 5, line_impedance: Tuple[float, float] = (0.01, 0.02),
                  load_power: float = 1e6) -> FeederSpec:
    """Chain feeder with the reference's SimpleRadialFeeder parameters (feeders/base.py:256-304)."""
    n = num_buses
    return FeederSpec(
        name=f"simple_radial_{n}", bus_ids=list(range(1, n + 1)),
        bus_type=np.array([SLACK] + [PQ] * (n - 1), dtype=np.uint8), v_set=np.ones(n),
        frm=_i32(range(0, n - 1)), to=_i32(range(1, n)),
        r=np.full(n - 1, float(line_impedance[0])), x=np.full(n - 1, float(line_impedance[1])),
        rating=np.full(n - 1, 5e6),
        load_bus=_i32(range(1, n)), load_base=np.full(n - 1, float(load_power)),
        load_pf=np.full(n - 1, 0.95))


# IEEE 13-node test feeder as the reference encodes it (ieee_feeders.py:37-113): bus order,
# (from, to, length_ft, config) and per-mile config impedances in ohm; base Z = 4.16^2/10.
_IEEE13_BUSES = [650, 632, 633, 634, 645, 646, 671, 680, 684, 611, 652, 692, 675]
_IEEE13_LINES = [(650, 632, 2000, "601"), (632, 633, 500, "602"), (632, 645, 500, "603"),
                 (632, 671, 2000, "601"), (645, 646, 300, "603"), (671, 680, 1000, "601"),
                 (671, 684, 300, "604"), (633, 634, 0, "XFM1"), (684, 611, 300, "603"),
                 (684, 652, 800, "607"), (671, 692, 0, "SWITCH"), (692, 675, 500, "606")]
_IEEE13_Z = {"601": (0.3465, 1.0179), "602": (0.7526, 1.1814), "603": (1.3238, 1.3569),
             "604": (1.3238, 1.3569), "606": (0.7982, 0.4463), "607": (1.3425, 0.5124),
             "XFM1": (0.0, 0.06), "SWITCH": (0.0001, 0.0001)}
_IEEE13_LOADS = [(634, 400, 290), (645, 170, 125), (646, 230, 132), (652, 128, 86),
                 (671, 1155, 660), (675, 843, 462), (692, 170, 151), (611, 170, 80)]


def ieee13_like(zero_length: str = "epsilon") -> FeederSpec:
    """13-bus feeder with the reference IEEE13Bus topology/impedances/loads.

    ``zero_length="as_coded"`` keeps the two zero-length elements at z = 0 (which the
    reference turns into open circuits, F5); ``"epsilon"`` gives them z = 1e-4 + 1e-4j pu so
    the network is connected and solvable (recorded deviation).
    Devices: solar_671 500 kW, wind_675 1 MW (ieee_feeders.py:126-141) and one storage unit
    at 671 with the reference env's battery parameters (grid_env.py:292-297).
    """
    zb = 4.16 ** 2 / 10.0
    idx = {b: i for i, b in enumerate(_IEEE13_BUSES)}
    frm, to, r, x = [], [], [], []
    for a, b, ft, cfg in _IEEE13_LINES:
        rm, xm = _IEEE13_Z[cfg]
        miles = ft / 5280.0
        rr, xx = (rm * miles) / zb, (xm * miles) / zb
        if ft == 0 and zero_length == "epsilon":
            rr, xx = 1e-4, 1e-4
        frm.append(idx[a]); to.append(idx[b]); r.append(rr); x.append(xx)
    lb = [idx[b] for b, _, _ in _IEEE13_LOADS]
    lp = [kw / 1000.0 * 1e6 for _, kw, _ in _IEEE13_LOADS]
    lpf = [kw / np.sqrt(kw ** 2 + kvar ** 2) for _, kw, kvar in _IEEE13_LOADS]
    return FeederSpec(
        name=f"ieee13_like_{zero_length}", bus_ids=list(_IEEE13_BUSES),
        bus_type=np.array([SLACK] + [PQ] * 12, dtype=np.uint8), v_set=np.ones(13),
        frm=_i32(frm), to=_i32(to), r=_f64(r), x=_f64(x), rating=np.full(12, 5e6),
        load_bus=_i32(lb), load_base=_f64(lp), load_pf=_f64(lpf),
        gen_bus=_i32([idx[671], idx[675]]), gen_kind=_i32([GEN_SOLAR, GEN_WIND]),
        gen_cap=_f64([500e3, 1e6]), gen_p0=_f64([0.18, 3.0]),
        gen_p1=_f64([500e3 / (0.18 * 1000.0), 12.0]), gen_p2=_f64([0.0, 25.0]),
        bat_bus=_i32([idx[671]]), bat_cap=_f64([1e3]), bat_rating=_f64([0.5e6]), bat_eff=_f64([0.95]),
        base_power_va=10e6)


_IEEE123_BACKBONE = [1, 3, 7, 13, 18, 25, 35, 49, 64, 78, 97, 114]


def ieee123_like(seed: int = 42, load_seed: int = 123) -> FeederSpec:
    """Seeded *connected radial* 123-bus feeder (SURVEY.md section 8(d) recipe).

    Uses the reference IEEE123Bus parameter ranges (backbone ids ieee_feeders.py:267; backbone
    z :277-278; lateral z :304-305; 70 % load probability and 10-200 kW :333-346; DG buses
    :350; storage buses :371) but attaches every non-backbone bus b to a uniformly drawn
    lower-numbered bus, which yields a spanning tree (122 lines) instead of the reference's
    77-component random graph.
    """
    rng = np.random.default_rng(seed)
    frm, to, r, x, rating = [], [], [], [], []
    for a, b in zip(_IEEE123_BACKBONE[:-1], _IEEE123_BACKBONE[1:]):
        rr = rng.uniform(0.003, 0.005); xx = rng.uniform(0.006, 0.010)
        frm.append(a - 1); to.append(b - 1); r.append(rr); x.append(xx); rating.append(15e6)
    bb = set(_IEEE123_BACKBONE)
    for b in range(2, 124):
        if b in bb:
            continue
        p = int(rng.integers(1, b))
        rr = rng.uniform(0.008, 0.013); xx = rng.uniform(0.012, 0.020)
        frm.append(p - 1); to.append(b - 1); r.append(rr); x.append(xx); rating.append(5e6)
    r2 = np.random.default_rng(load_seed)
    r3 = np.random.default_rng(load_seed + 1)
    lb, lp, lpf = [], [], []
    for b in range(2, 124):
        if r2.random() < 0.7:
            kw = 10 + 190 * r2.random()
            lb.append(b - 1); lp.append(kw * 1000.0); lpf.append(0.92 + 0.06 * r3.random())
    dg = [25, 49, 78, 97, 114]
    gk = [GEN_SOLAR if i % 2 == 0 else GEN_WIND for i in range(len(dg))]
    gc, g0, g1, g2 = [], [], [], []
    for k in gk:
        if k == GEN_SOLAR:
            cap = (200 + 300 * r3.random()) * 1000; eff = 0.18 + 0.04 * r3.random()
            gc.append(cap); g0.append(eff); g1.append(cap / (eff * 1000.0)); g2.append(0.0)
        else:
            gc.append((500 + 1000 * r3.random()) * 1000); g0.append(3.0); g1.append(12.0); g2.append(25.0)
    st = [35, 64, 97]
    bc = [(500 + 500 * r3.random()) * 1000.0 for _ in st]      # Wh, consistent with W-valued power (E = P*dt/3600)
    br = [(250 + 250 * r3.random()) * 1000.0 for _ in st]     # W, so that action*rating is in W like the loads
    be = [0.90 + 0.05 * r3.random() for _ in st]
    return FeederSpec(
        name=f"ieee123_like_radial_seed{seed}", bus_ids=list(range(1, 124)),
        bus_type=np.array([SLACK] + [PQ] * 122, dtype=np.uint8), v_set=np.ones(123),
        frm=_i32(frm), to=_i32(to), r=_f64(r), x=_f64(x), rating=_f64(rating),
        load_bus=_i32(lb), load_base=_f64(lp), load_pf=_f64(lpf),
        gen_bus=_i32([b - 1 for b in dg]), gen_kind=_i32(gk), gen_cap=_f64(gc), gen_p0=_f64(g0),
        gen_p1=_f64(g1), gen_p2=_f64(g2),
        bat_bus=_i32([b - 1 for b in st]), bat_cap=_f64(bc), bat_rating=_f64(br), bat_eff=_f64(be),
        base_power_va=10e6)


def random_meshed(n: int, extra_lines: int, seed: int = 0) -> FeederSpec:
    """Connected feeder with ``extra_lines`` loop-closing branches on top of a random
    spanning tree (impedance ranges of reference synthetic.py:128-155 scaled to pu)."""
    rng = np.random.default_rng(seed)
    frm, to = [], []
    for b in range(1, n):
        frm.append(int(rng.integers(0, b))); to.append(b)
    have = {(min(a, b), max(a, b)) for a, b in zip(frm, to)}
    tries = 0
    while extra_lines > 0 and tries < 100 * (extra_lines + 1):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n)); tries += 1
        if a == b or (min(a, b), max(a, b)) in have:
            continue
        have.add((min(a, b), max(a, b))); frm.append(a); to.append(b); extra_lines -= 1
    m = len(frm)
    r = rng.uniform(0.005, 0.02, m); x = rng.uniform(0.01, 0.04, m)
    lb = list(range(1, n))
    return FeederSpec(
        name=f"random_meshed_{n}_{m}_seed{seed}", bus_ids=list(range(1, n + 1)),
        bus_type=np.array([SLACK] + [PQ] * (n - 1), dtype=np.uint8), v_set=np.ones(n),
        frm=_i32(frm), to=_i32(to), r=_f64(r), x=_f64(x), rating=np.full(m, 5e6),
        load_bus=_i32(lb), load_base=rng.uniform(10e3, 200e3, n - 1), load_pf=np.full(n - 1, 0.95))


def scalable_like(num_buses: int = 123, seed: int = 1, connectivity: Optional[float] = None) -> FeederSpec:
    """Seeded MESHED feeder following the recipe of the reference's ``ScalableFeeder(num_buses, seed)``
    (feeders/synthetic.py:233-251 on top of ``SyntheticFeeder``, :65-214), drawn from a private
    ``default_rng(seed)`` instead of the process-global NumPy generator: random spanning tree grown from bus 1
    (:75-94), then random extra lines towards ``tree + connectivity * (all pairs - tree)`` with the reference's
    cap of 1000 attempts (:96-124; connectivity = clip(20 / n, 0.1, 0.6), i.e. ~1000 lines at n = 123 -- a graph
    whose block LU fills in almost completely), per-km impedances 0.2-0.5 / 0.3-0.7 ohm over 0.05-1.5 km on the
    12.47 kV / 10 MVA base, ratings 2-10 MVA (:126-155), loads with probability min(0.9, 0.5 + 0.01 n) of 20-300 kW at
    power factor 0.85-1.0 (:157-177), DG with probability min(0.4, 0.1 + 0.005 n), a third each solar / wind / battery
    (:179-213).  ``connectivity`` overrides the recipe's value (0.0 = the spanning tree alone)."""
    n = int(num_buses)
    rng = np.random.default_rng(seed)
    conn = max(0.1, min(0.6, 20.0 / n)) if connectivity is None else float(connectivity)
    base_z = 12.47 ** 2 / 10.0
    frm, to, r, x, rating = [], [], [], [], []

    def add_line(a, b):
        length = 0.05 + 1.45 * rng.random()
        frm.append(a); to.append(b)
        r.append((0.2 + 0.3 * rng.random()) * length / base_z); x.append((0.3 + 0.4 * rng.random()) * length / base_z)
        rating.append((2 + 8 * rng.random()) * 1e6)

    connected, unconnected = [0], list(range(1, n))
    while unconnected:
        a = connected[int(rng.integers(0, len(connected)))]
        b = unconnected.pop(int(rng.integers(0, len(unconnected))))
        add_line(a, b); connected.append(b)
    have = {(a, b) for a, b in zip(frm, to)} | {(b, a) for a, b in zip(frm, to)}
    target = int(len(frm) + conn * (n * (n - 1) // 2 - len(frm)))
    attempts = 0
    while len(frm) < target and attempts < 1000:
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n)); attempts += 1
        if a != b and (a, b) not in have:
            add_line(a, b); have.add((a, b)); have.add((b, a))
    lb, lp, lpf = [], [], []
    p_load = min(0.9, 0.5 + 0.01 * n)
    for b in range(1, n):
        if rng.random() < p_load:
            lb.append(b); lp.append((20 + 280 * rng.random()) * 1000.0); lpf.append(0.85 + 0.15 * rng.random())
    gb, gk, gc, g0, g1, g2, bb, bc, br, be = [], [], [], [], [], [], [], [], [], []
    p_dg = min(0.4, 0.1 + 0.005 * n)
    for b in range(1, n):
        if rng.random() < p_dg:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                cap = (100 + 400 * rng.random()) * 1000.0; eff = 0.15 + 0.10 * rng.random()
                gb.append(b); gk.append(GEN_SOLAR); gc.append(cap); g0.append(eff); g1.append(cap / (eff * 1000.0)); g2.append(0.0)
            elif kind == 1:
                gb.append(b); gk.append(GEN_WIND); gc.append((500 + 1500 * rng.random()) * 1000.0)
                g0.append(2.5 + rng.random()); g1.append(10 + 5 * rng.random()); g2.append(20 + 10 * rng.random())
            else:
                kwh = 200 + 800 * rng.random()
                bb.append(b); bc.append(kwh * 1000.0); br.append(0.5 * kwh * 1000.0); be.append(0.85 + 0.10 * rng.random())
    return FeederSpec(
        name=f"scalable_like_{n}_seed{seed}_m{len(frm)}", bus_ids=list(range(1, n + 1)),
        bus_type=np.array([SLACK] + [PQ] * (n - 1), dtype=np.uint8), v_set=np.ones(n),
        frm=_i32(frm), to=_i32(to), r=_f64(r), x=_f64(x), rating=_f64(rating),
        load_bus=_i32(lb), load_base=_f64(lp), load_pf=_f64(lpf),
        gen_bus=_i32(gb), gen_kind=_i32(gk), gen_cap=_f64(gc), gen_p0=_f64(g0), gen_p1=_f64(g1), gen_p2=_f64(g2),
        bat_bus=_i32(bb), bat_cap=_f64(bc), bat_rating=_f64(br), bat_eff=_f64(be), base_power_va=10e6)


# ----------------------------------------------------------------------------------------
# dictionary (JSON) network format: CustomFeeder.from_dict / to_dict, feeders/base.py:170-253
# ----------------------------------------------------------------------------------------
_FEEDER_PARAMETERS = {"base_voltage": 12.47, "base_power": 10.0, "frequency": 60.0}     # FeederParameters defaults, base.py:18-27


def network_dict_normalized(network_dict: Dict[str, Any], name: str = "Custom",
                            parameters: Optional[Dict[str, float]] = None) -> Dict[str, Any]:
    """What ``CustomFeeder(name).from_dict(d); .to_dict()`` returns: the same schema with every optional key filled
    in the way the reference fills it (voltage_level = base_voltage[kV] * 1000, type "pq", base_voltage 1.0,
    rating 1e6, power_factor 0.95; generator entries verbatim)."""
    par = dict(_FEEDER_PARAMETERS); par.update(parameters or network_dict.get("parameters", {}) or {})
    return {
        "name": name,
        "parameters": {k: par[k] for k in ("base_voltage", "base_power", "frequency")},
        "buses": [{"id": b["id"], "voltage_level": b.get("voltage_level", par["base_voltage"] * 1000), "type": b.get("type", "pq"),
                   "base_voltage": b.get("base_voltage", 1.0)} for b in network_dict.get("buses", [])],
        "lines": [{"id": l["id"], "from_bus": l["from_bus"], "to_bus": l["to_bus"], "resistance": l["resistance"],
                   "reactance": l["reactance"], "rating": l.get("rating", 1e6)} for l in network_dict.get("lines", [])],
        "loads": [{"id": d["id"], "bus": d["bus"], "power": d["power"], "power_factor": d.get("power_factor", 0.95)}
                  for d in network_dict.get("loads", [])],
        "generators": [dict(g) for g in network_dict.get("generators", [])],
    }


def feeder_from_dict(network_dict: Dict[str, Any], name: Optional[str] = None) -> FeederSpec:
    """FeederSpec from the reference's dictionary network format (``CustomFeeder.from_dict``, base.py:170-213).
    The normalised dictionary is kept on the spec (``spec.source_dict``) so that ``feeder_to_dict`` round-trips ids,
    voltage levels and generator entries the flat arrays do not carry."""
    d = network_dict_normalized(network_dict, name or network_dict.get("name", "Custom"))

    class _F:        # BaseFeeder-shaped view for flatten_feeder
        pass
    f = _F()
    f.name = d["name"]
    f.parameters = type("P", (), {"base_power": d["parameters"]["base_power"]})()
    f.buses = [Bus(b["id"], bus_type=b["type"]) for b in d["buses"]]
    f.lines = [Line(l["id"], l["from_bus"], l["to_bus"], l["resistance"], l["reactance"], l["rating"]) for l in d["lines"]]
    f.loads = [Load(q["id"], q["bus"], q["power"], q["power_factor"]) for q in d["loads"]]
    f.generators = {g["id"]: g for g in d["generators"]}
    spec = flatten_feeder(f, name=d["name"])
    spec.source_dict = d
    return spec


def feeder_to_dict(spec: FeederSpec) -> Dict[str, Any]:
    """The reference's ``to_dict`` schema (base.py:215-253) for any FeederSpec.  A spec that came from
    ``feeder_from_dict`` returns its normalised source; others get generated line / load ids (``line_k``,
    ``load_k``) and their renewable / storage units as generator entries ``flatten_feeder`` reads back."""
    src = getattr(spec, "source_dict", None)
    if src is not None:
        return json.loads(json.dumps(src))
    names = {PQ: "pq", PV: "pv", SLACK: "slack"}
    gens: List[Dict[str, Any]] = []
    for g in range(spec.n_gens):
        bus = spec.bus_ids[int(spec.gen_bus[g])]
        if int(spec.gen_kind[g]) == GEN_SOLAR:
            gens.append({"id": f"solar_{g}", "type": "solar", "bus": bus, "capacity": float(spec.gen_cap[g]),
                         "efficiency": float(spec.gen_p0[g]), "panel_area": float(spec.gen_p1[g])})
        else:
            gens.append({"id": f"wind_{g}", "type": "wind", "bus": bus, "capacity": float(spec.gen_cap[g]), "cut_in_speed": float(spec.gen_p0[g]),
                         "rated_speed": float(spec.gen_p1[g]), "cut_out_speed": float(spec.gen_p2[g])})
    for q in range(spec.n_bats):
        gens.append({"id": f"battery_{q}", "type": "battery", "bus": spec.bus_ids[int(spec.bat_bus[q])], "capacity": float(spec.bat_cap[q]),
                     "power_rating": float(spec.bat_rating[q]), "efficiency": float(spec.bat_eff[q])})
    par = dict(_FEEDER_PARAMETERS); par["base_power"] = spec.base_power_va / 1e6
    return {
        "name": spec.name, "parameters": par,
        "buses": [{"id": bid, "voltage_level": par["base_voltage"] * 1000, "type": names[int(spec.bus_type[i])], "base_voltage": 1.0}
                  for i, bid in enumerate(spec.bus_ids)],
        "lines": [{"id": f"line_{k}", "from_bus": spec.bus_ids[int(spec.frm[k])], "to_bus": spec.bus_ids[int(spec.to[k])],
                   "resistance": float(spec.r[k]), "reactance": float(spec.x[k]), "rating": float(spec.rating[k])} for k in range(spec.m)],
        "loads": [{"id": f"load_{l}", "bus": spec.bus_ids[int(spec.load_bus[l])], "power": float(spec.load_base[l]),
                   "power_factor": float(spec.load_pf[l])} for l in range(spec.n_loads)],
        "generators": gens,
    }
