#!/usr/bin/env python3
"""When do the workgroups of one step launch start and end?  (gs_debug_block_times; 100 MHz real-time clock)
    GS_STAMP_BLOCK_TIMES=1 python tools/block_times.py [--solver fbs] [--batch 8192]
Prints the spread of the start times, the durations, and the launch's span from the first start to the last end."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GS_STAMP_BLOCK_TIMES", "1")
import numpy as np
import grid_fed_rl_gym_amd as P

ap = argparse.ArgumentParser()
ap.add_argument("--solver", default="fbs"); ap.add_argument("--batch", type=int, default=8192)
a = ap.parse_args()
fs = P.ieee123_like()
env = P.BatchedGridEnvironment(fs, num_envs=a.batch, solver=a.solver, stochastic_loads=True, weather_variation=True)
h = env.handle
h.upload_actions(np.random.default_rng(5678).uniform(-1, 1, (8, a.batch, fs.action_dim)))
env.reset(seed=np.arange(a.batch, dtype=np.uint64))
st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
for k in range(20):
    h.step_device(k % 8)
h.synchronize()
h.debug_stamps()                      # arm
nb = min(2048, int(h.describe()["workgroups"]))
rows = []
for rep in range(5):
    for k in range(10):
        h.step_device(k % 8)
    t = h.debug_block_times(nb).astype(np.int64)
    t0 = t[:, 0].min()
    start, dur = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0           # microseconds
    if nb >= 4:
        half = nb // 2
        print(json.dumps({"median_start_us_first_half": float(np.median(start[:half])), "median_start_us_second_half": float(np.median(start[half:])),
                          "median_end_us_first_half": float(np.median(start[:half] + dur[:half])), "median_end_us_second_half": float(np.median(start[half:] + dur[half:]))}))
    if rep == 4:
        by_xcd = [round(float(np.median(dur[x::8])), 2) for x in range(8)]
        order = np.argsort(dur)
        print(json.dumps({"median_duration_by_blockIdx_mod_8": by_xcd, "slowest_blocks": order[-24:].tolist(), "fastest_blocks": order[:24].tolist(),
                          "duration_of_blocks_0_to_31": [round(float(x), 1) for x in dur[:32]]}))
    rows.append({"span_us": float((t[:, 1].max() - t0) / 100.0), "start_us_p50_p90_max": [float(np.percentile(start, q)) for q in (50, 90, 100)],
                 "duration_us_min_p50_p90_max": [float(np.percentile(dur, q)) for q in (0, 50, 90, 100)],
                 "late_starters": int((start > 5.0).sum())})
print(json.dumps({"kernel": h.describe()["kernel"], "workgroups": nb}))
for r in rows:
    print(json.dumps(r))
