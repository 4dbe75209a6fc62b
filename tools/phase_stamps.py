#!/usr/bin/env python3
"""In-kernel phase stamps of the step kernel (gs_debug_stamps): cycles per phase and step of one wave of workgroup 0.
    GS_STAMP_WAVE=3 python tools/phase_stamps.py [--solver fbs|nr] [--batch 8192] [--steps 50] [--feeder ieee123|ieee13|loops26|scalable]
Diagnostic only: the stamped build is the shipped build with a buffer armed (one scalar branch per stamp when it is not)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grid_fed_rl_gym_amd as P

ap = argparse.ArgumentParser()
ap.add_argument("--solver", default="fbs"); ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--steps", type=int, default=50); ap.add_argument("--feeder", default="ieee123")
a = ap.parse_args()
fs = {"ieee123": P.ieee123_like, "ieee13": lambda: P.ieee13_like("epsilon"), "loops26": lambda: P.random_meshed(123, 26, seed=1),
      "scalable": lambda: P.scalable_like(123, seed=1)}[a.feeder]()
env = P.BatchedGridEnvironment(fs, num_envs=a.batch, solver=a.solver, stochastic_loads=True, weather_variation=True)
h = env.handle
acts = np.random.default_rng(5678).uniform(-1, 1, (8, a.batch, fs.action_dim))
h.upload_actions(acts)
env.reset(seed=np.arange(a.batch, dtype=np.uint64))
st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
for k in range(5):
    h.step_device(k % 8)
h.synchronize()
h.debug_stamps()                      # arm
for k in range(a.steps):
    h.step_device(k % 8)
h.synchronize()
s = h.debug_stamps()
tot = sum(s.values())
print(json.dumps({"kernel": h.describe()["kernel"], "wave": int(os.environ.get("GS_STAMP_WAVE", "0")), "steps": a.steps,
                  "cycles_per_step": {k: round(v / a.steps) for k, v in s.items() if v}, "total_per_step": round(tot / a.steps)}))
