import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import grid_fed_rl_gym_amd as P
for name, fs, B, solver in (("c2", P.ieee13_like("epsilon"), 4096, "nr"), ("headline", P.ieee123_like(), 8192, "fbs")):
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver=solver, stochastic_loads=True, weather_variation=True)
    h = env.handle
    acts = np.random.default_rng(5678).uniform(-1, 1, (8, B, fs.action_dim)); h.upload_actions(acts)
    env.reset(seed=np.arange(B, dtype=np.uint64))
    for k in range(200): h.step_device(k % 8)
    h.synchronize()
    for n in (20, 200, 2000):
        t0 = time.perf_counter()
        for k in range(n): h.step_device(k % 8)
        t1 = time.perf_counter()
        h.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: {n} steps: host issue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step", flush=True)
    env.close()
