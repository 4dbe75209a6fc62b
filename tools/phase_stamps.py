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
if h.describe()["kernel"] == "nr_dense_mfma":      # the dense kernel's own slots (kernels_dense.hip GdStamp), summed over the instances workgroup 0 took
    names = ["mismatch", "assembly", "update_products_mfma", "u_blocks_to_scratch", "gauss_jordan", "l_blocks_and_forward_substitution",
             "back_substitution", "corrections", "row_io"]
    vals = list(s.values())
    dense = dict(zip(names, vals[:len(names)]))
    rest = {k: v for k, v in list(s.items())[len(names):]}
    d = h.describe()
    per_wg = a.batch / max(1, d.get("dense_workgroups", 1))
    print(json.dumps({"kernel": "nr_dense_mfma", "dense_form": d.get("dense_form"), "workgroups": d.get("dense_workgroups"), "instances_per_workgroup_and_step": per_wg, "steps": a.steps,
                      "cycles_per_instance": {k: round(v / a.steps / per_wg) for k, v in dense.items()},
                      "total_per_instance": round(sum(dense.values()) / a.steps / per_wg),
                      "pre_and_post_kernels_cycles_per_step_of_their_workgroup_0": {k: round(v / a.steps) for k, v in rest.items() if v}}))
    sys.exit(0)
tot = sum(s.values())
print(json.dumps({"kernel": h.describe()["kernel"], "wave": int(os.environ.get("GS_STAMP_WAVE", "0")), "steps": a.steps,
                  "cycles_per_step": {k: round(v / a.steps) for k, v in s.items() if v}, "total_per_step": round(tot / a.steps)}))
