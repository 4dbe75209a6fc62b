"""Shared test plumbing: golden loading and FeederSpec <-> oracle EnvSpec conversion."""
import glob
import os

import numpy as np

from oracle import oracle_np as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def net_of(d):
    """(n, frm, to, r, x, rating, bus_type, v_set) from a fixture."""
    return (len(d["bus_type"]), d["frm"], d["to"], d["r"], d["x"], d["rating"], d["bus_type"], d["v_set"])


def oracle_spec(fs, **cfg):
    """oracle EnvSpec from a product FeederSpec (tests own this mapping; the product never imports oracle/)."""
    return O.EnvSpec(n=fs.n, frm=fs.frm, to=fs.to, r=fs.r, x=fs.x, rating=fs.rating, bus_type=fs.bus_type,
                     v_set=fs.v_set, load_bus=fs.load_bus, load_base=fs.load_base, load_pf=fs.load_pf,
                     gen_bus=fs.gen_bus, gen_kind=fs.gen_kind, gen_cap=fs.gen_cap, gen_p0=fs.gen_p0,
                     gen_p1=fs.gen_p1, gen_p2=fs.gen_p2, bat_bus=fs.bat_bus, bat_cap=fs.bat_cap,
                     bat_rating=fs.bat_rating, bat_eff=fs.bat_eff, bat_soc0=np.full(fs.n_bats, 0.5), **cfg)
