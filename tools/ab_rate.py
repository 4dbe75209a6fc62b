#!/usr/bin/env python3
"""A/B of builds of the library on the sustained rate: 2000 steps back to back (no synchronisation in between), median of
five, each build in a process of its own, the builds alternating twice.
    python tools/ab_rate.py libgridstep_a.so libgridstep.so        (file names inside grid_fed_rl_gym_amd/; through gpurun)
How the round-3 changes of DESIGN.md section 3 ("what binds the step kernels") were told apart from noise: the driver's
protocol (20-step regions) carries +-1 % from one run to the next, this figure +-0.3 %.  A second build is made by stashing
the change, `make`, copying libgridstep.so to another name next to it (built .so files travel to the GPU box)."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd import _lib
_lib.LIB_PATH = sys.argv[1]
out = {}
for name, fs, B, solver in (("headline", P.ieee123_like(), 8192, "fbs"), ("nr", P.ieee123_like(), 8192, "nr"), ("c2", P.ieee13_like("epsilon"), 4096, "nr")):
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True)
    h = env.handle
    acts = np.random.default_rng(5678).uniform(-1, 1, (8, B, fs.action_dim)); h.upload_actions(acts)
    env.reset(seed=np.arange(B, dtype=np.uint64))
    for k in range(500): h.step_device(k %% 8)
    h.synchronize()
    best = []
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(2000): h.step_device(k %% 8)
        h.synchronize()
        best.append((time.perf_counter() - t0) / 2000 * 1e6)
    out[name] = sorted(best)[2]
    env.close()
print(out)
''' % ROOT
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "grid_fed_rl_gym_amd", lib)], capture_output=True, text=True)
        print(lib, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
