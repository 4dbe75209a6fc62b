"""Development check of the resident three-phase kernel: resident vs level kernel vs NumPy oracle, and timing.
Usage: python tools/c5_resident_check.py [--time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid_fed_rl_gym_amd.unbalanced import UnbalancedPowerFlow, UnbalancedFeederSpec, ieee8500_like
from oracle import oracle3_np as O3


def random_case(n, seed, lateral=0.35, local=0):
    rng = np.random.default_rng(seed)
    parent = np.full(n, -1, dtype=np.int32); phases = np.full(n, 7, dtype=np.uint8); z = np.zeros((n, 3, 3), dtype=complex)
    for b in range(1, n):
        p = int(rng.integers(max(0, b - local) if local else 0, b)); parent[b] = p
        m = int(phases[p])
        if m == 7 and rng.random() < lateral:
            m = [1, 2, 4, 3, 5, 6][int(rng.integers(0, 6))]
        phases[b] = m
        zs = complex(rng.uniform(0.004, 0.01), rng.uniform(0.008, 0.02)) * (30.0 / n if n > 300 else 1.0)
        z[b] = zs * np.eye(3) + rng.uniform(0.2, 0.4) * zs * (1 - np.eye(3))
    return UnbalancedFeederSpec("rnd", parent, phases, z)


def solve(spec, P, Q, resident, tol=1e-9):
    if resident: os.environ.pop("GS3_NO_RESIDENT", None)
    else: os.environ["GS3_NO_RESIDENT"] = "1"
    s = UnbalancedPowerFlow(tolerance=tol, max_iterations=200)
    sol = s.solve_batch(spec, P, Q)
    d = s.describe()
    s.close()
    return sol, d


ok = True
CASES = [] if "--only-time" in sys.argv else [(40, 5, 1, 0.35), (150, 70, 2, 0.35), (333, 9, 3, 0.35), (900, 3, 4, 0.0), (2500, 3, 5, 0.3), (2300, 2, 6, 0.0), (4000, 2, 7, 0.5)]
for n, B, seed, lat in CASES:
    spec = random_case(n, seed, lat, local=60 if n > 400 else 0)
    rng = np.random.default_rng(seed + 100)
    pres = ((spec.phases[:, None] >> np.arange(3)[None, :]) & 1).astype(bool)
    Pb = np.where(pres[None], -rng.uniform(0.0002, 0.003, (B, n, 3)) * min(1.0, 100.0 / n), 0.0); Pb[:, 0] = 0
    Qb = Pb * rng.uniform(0.2, 0.5, (B, n, 3))
    a, da = solve(spec, Pb, Qb, True)
    b, db = solve(spec, Pb, Qb, False)
    dv = np.max(np.abs(a.voltages - b.voltages))
    line = f"n={n} ns={da['conductors']} kernel={da['kernel']} K={da['positions_per_thread']} thr={da['threads']} conv={a.converged.all()}/{b.converged.all()} " \
           f"it={a.iterations.tolist()}/{b.iterations.tolist()} dV={dv:.2e} dloss={np.max(np.abs(a.losses - b.losses)):.2e} dmm={np.max(np.abs(a.max_mismatch - b.max_mismatch)):.2e}"
    if n <= 1000:
        ref = O3.fbs3_solve(spec.parent, spec.phases, spec.z, 0, spec.v_source, Pb[0], Qb[0], tolerance=1e-9, max_iterations=200)
        line += f" vs oracle {np.max(np.abs(a.voltages[0] - ref['voltages'])):.2e} it_ref={ref['iterations']}"
    print(line, flush=True)
    ok &= bool(dv < 1e-10 and (a.iterations == b.iterations).all() and a.converged.all())
    ok &= bool(np.all(a.voltages[:, ~pres] == 0))
print("OK" if ok else "MISMATCH", flush=True)

if "--time" in sys.argv or "--only-time" in sys.argv:
    spec, Pn, Qn = ieee8500_like()
    for B in ((256,) if os.environ.get("GS3_STAMPS") or "--b256" in sys.argv else (256, 1024, 2048)):
        rng = np.random.default_rng(0)
        lam = rng.uniform(0.5, 1.5, B)
        P = lam[:, None, None] * Pn[None]; Q = lam[:, None, None] * Qn[None]
        for resident in (True, False):
            if resident: os.environ.pop("GS3_NO_RESIDENT", None)
            else: os.environ["GS3_NO_RESIDENT"] = "1"
            s = UnbalancedPowerFlow(tolerance=1e-6, max_iterations=100)
            s.upload(spec, P, Q)
            for _ in range(3): s.solve_device()
            s.synchronize(); s.timing_read()
            for _ in range(20): s.solve_device()
            ms, cnt = s.timing_read()
            sol = s.download()
            print(f"B={B} resident={resident} {s.describe()['kernel']}: {ms / cnt:.4f} ms per launch, {B / (ms / cnt) * 1e3 / 1e6:.3f} M solves/s, "
                  f"iters mean {sol.iterations.mean():.2f}, conv {sol.converged.mean():.3f}", flush=True)
            if resident: keep = sol
            else: print(f"   dV resident vs levels {np.max(np.abs(keep.voltages - sol.voltages)):.2e}, iters equal {(keep.iterations == sol.iterations).all()}", flush=True)
            s.close()
sys.exit(0 if ok else 1)
