"""CPU-only: the C-ABI library loads and exports every symbol include/gridstep.h declares;
host-side validation paths work without a GPU; nothing on the product path imports oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gridstep.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs3?_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 25
    from grid_fed_rl_gym_amd.unbalanced import GS3_SYMBOLS
    from grid_fed_rl_gym_amd.safety import CHECKS_SYMBOLS
    bound = {name for name, _, _ in _lib.SYMBOLS} | {name for name, _, _ in GS3_SYMBOLS} | {name for name, _, _ in CHECKS_SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gridstep.h but not exported by libgridstep.so"
        assert name in bound, f"{name} declared in gridstep.h but not bound in _lib.SYMBOLS"
    assert lib.gs_version() == _lib.GS_ABI_VERSION


def test_struct_sizes_match_header_layout():
    # natural alignment of the C structs: pointers 8 bytes, int32 padded before a pointer/double
    assert ctypes.sizeof(_lib.gs_config) == 11 * 4 + 4 + 12 * 8
    assert ctypes.sizeof(_lib.gs_solution_view) == 9 * 8
    assert ctypes.sizeof(_lib.gs_info_view) == 10 * 8
    assert ctypes.sizeof(_lib.gs_topology) % 8 == 0


def test_create_fails_loudly_without_gpu_or_with_bad_args():
    lib = _lib.load()
    if lib.gs_device_count() == 0:
        with pytest.raises(P.PowerFlowError, match="no HIP device|no CPU fallback"):
            P.BatchedNewtonRaphsonSolver().solve_batch(P.simple_radial(5), np.zeros((2, 5)))
    cfg = _lib.make_config()
    cfg.struct_size = 3
    h = ctypes.c_void_p()
    t = _lib.gs_topology()
    t.struct_size = ctypes.sizeof(_lib.gs_topology)
    rc = lib.gs_create(ctypes.byref(t), ctypes.byref(cfg), 1, 0, 0, ctypes.byref(h))
    assert rc == _lib.GS_E_INVALID and b"struct_size" in lib.gs_last_error(None)
    assert lib.gs_create(None, None, 1, 0, 0, ctypes.byref(h)) == _lib.GS_E_INVALID


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "grid_fed_rl_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle_cpu" not in text and "liboracle" not in text, f


def test_feeder_generators_and_flattening():
    s = P.ieee123_like()
    assert (s.n, s.m, s.n_loads) == (123, 122, 91) and s.is_radial()
    assert abs(s.load_base.sum() / 10e6 - 0.9434566272596138) < 1e-15
    assert s.obs_dim == 2 * 123 + 2 * 122 + 1 + 2 * 91 + 5 + 2 * 3 and s.action_dim == 8
    buses, lines, loads = P.to_objects(s)
    bus_ids, bt, vs, frm, to, r, x, rating = P.flatten_network(buses, lines)
    assert np.array_equal(frm, s.frm) and np.array_equal(to, s.to) and np.array_equal(r, s.r)
    assert not P.random_meshed(30, 12, seed=7).is_radial()
    e = P.reference_env_network()
    assert (e.obs_dim, e.action_dim) == (17, 1)
    assert P.with_reference_env_renewables(P.reference_env_network(), ["solar", "wind"]).obs_dim == 19
    p = P.injections_from_dicts(e, {2: 0.2, 3: 0.15, 99: 7.0}, {3: 0.05})
    assert np.allclose(p, [0.0, -0.2, -0.1])


def test_device_array_interface_and_device_address_validation():
    """`DeviceArray` speaks `__cuda_array_interface__` (what PyTorch-ROCm / CuPy read); `_device_address` accepts an address, an object
    with that interface or a `data_ptr()` and refuses wrong shapes / dtypes / strides before anything reaches the GPU."""
    from grid_fed_rl_gym_amd._lib import DeviceArray, _device_address
    from grid_fed_rl_gym_amd.components import PowerFlowError
    a = DeviceArray(0x7f0000001000, (4, 3), "<f8")
    cai = a.__cuda_array_interface__
    assert cai["shape"] == (4, 3) and cai["typestr"] == "<f8" and cai["data"] == (0x7f0000001000, False) and cai["version"] == 3
    assert _device_address(a, (4, 3)) == 0x7f0000001000 and _device_address(12345, (1, 1)) == 12345
    with pytest.raises(PowerFlowError):
        _device_address(a, (3, 4))
    with pytest.raises(PowerFlowError):
        _device_address(DeviceArray(1, (4, 3), "<f4"), (4, 3))
    with pytest.raises(PowerFlowError):
        _device_address(object(), (4, 3))

    class FakeTensor:                      # what a torch tensor exposes
        shape, dtype = (4, 3), "torch.float64"
        def is_contiguous(self): return True
        def data_ptr(self): return 4096
    assert _device_address(FakeTensor(), (4, 3)) == 4096
    FakeTensor.dtype = "torch.float32"
    with pytest.raises(PowerFlowError):
        _device_address(FakeTensor(), (4, 3))


def test_scalable_like_generator_is_seeded_meshed_and_in_the_reference_recipes_range():
    """feeders.scalable_like: the recipe of the reference's ScalableFeeder (feeders/synthetic.py:233-251) from a private seeded
    generator -- a connected graph of ~1000 lines on 123 buses (the reference's own instance has 1035, tests/golden/solve_scal123),
    the spanning tree alone with connectivity 0, hash-pinned like the other synthetic feeders."""
    s = P.scalable_like(123, seed=1)
    assert (s.n, s.m, s.n_loads, s.n_gens, s.n_bats) == (123, 1038, 111, 26, 17) and not s.is_radial()
    assert s.sha256() == P.scalable_like(123, seed=1).sha256() != P.scalable_like(123, seed=2).sha256()
    assert s.sha256().startswith("1e19c85ae909a99c")
    pairs = {(min(a, b), max(a, b)) for a, b in zip(s.frm, s.to)}
    assert len(pairs) == s.m and all(a != b for a, b in pairs)                       # no parallel lines, no self loops
    assert 0.0 < s.r.min() and s.r.max() < 0.05 and 0.0 < s.x.min() and s.x.max() < 0.07     # 0.2-0.5 / 0.3-0.7 ohm/km over 0.05-1.5 km on 15.55 ohm
    assert 2e6 <= s.rating.min() and s.rating.max() <= 10e6 and 20e3 <= s.load_base.min() and s.load_base.max() <= 300e3
    t = P.scalable_like(40, seed=3, connectivity=0.0)
    assert t.m == 39 and t.is_radial()


def test_bench_reports_counter_traffic_only_for_the_kernel_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic is a constant from profiles/hbm_traffic.json, not something a bench run measures: bench.py hands it out only
    while the kernel sources hash to what the entry recorded, and says where it came from either way."""
    import importlib, json, sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    cur = bench.csrc_hash()
    assert len(cur) == 16 and cur == bench.csrc_hash()
    assert bench.csrc_hash("ieee8500_3ph_b1024:fbs3") != bench.csrc_hash("ieee123_b8192:fbs")      # tied to the kernel's own sources
    prof = tmp_path / "profiles"; prof.mkdir()
    json.dump({"a:fbs": {"solve_bytes_per_launch": 123, "csrc_sha": cur, "profile": "rXX"},
               "b:nr": {"solve_bytes_per_launch": 456, "csrc_sha": "0" * 16, "profile": "old"}}, open(prof / "hbm_traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_hash", lambda key=None: cur)
    assert bench.traffic_of("a:fbs") == 123 and bench.traffic_of("b:nr") is None and bench.traffic_of("missing") is None
    src = bench.traffic_source("b:nr")
    assert src["stale"] and src["bytes_when_measured"] == 456 and src["profile"] == "old"
    assert not bench.traffic_source("a:fbs")["stale"]
