#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched AC power-flow env.step() on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ieee123_b8192|ieee13_b4096]
                    [--solver nr|fbs] [--batch B] [--no-cpu-baseline]

One "step" = one batched env.step(): actions -> batteries/curtailment -> weather -> injections
-> AC load flow -> line flows -> frequency -> reward/flags -> observation block, for every
instance of the batch, with the K action batches already resident in HBM.  The default
workload is the configuration BASELINE.json's target is quoted on (configs[2]): the 123-bus radial
feeder, 8192 instances per GPU, forward/backward-sweep load flow, reference defaults for stochastic
loads and weather.  The same line also carries the Newton-Raphson measurement ("also"), the accuracy
of the GPU voltages against the CPU oracle's Newton-Raphson, and the CPU baseline.

For N > 1 the driver launches one process per GPU (torch.distributed.run); each rank owns a
contiguous block of instances (weak scaling, per-GPU batch fixed); the ranks are independent, so
`value` has no collective in it.  The one exchange north_star names -- the RCCL all-gather of the
observation block after each step -- is timed in a second pass and reported in the same line
("with_obs_allgather"): it is bound by xGMI, not by the step.  torch is imported only for the
rendezvous (gloo barrier + max-over-ranks), never for compute; libgridstep.so is loaded first
so that the process uses one HIP runtime.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from grid_fed_rl_gym_amd import _lib  # noqa: E402
import grid_fed_rl_gym_amd as P  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # vendor figure for MI355X FP64 vector

WORKLOADS = {
    "ieee123_b8192": dict(feeder="ieee123_like", batch=8192),
    "ieee13_b4096": dict(feeder="ieee13_like", batch=4096),
    "ieee8500_3ph_b1024": dict(feeder="ieee8500_like", batch=1024),      # BASELINE.json config 5 (solver only)
}


def make_feeder(name):
    return P.ieee123_like() if name == "ieee123_like" else P.ieee13_like("epsilon")


def algorithmic_bytes_per_step(fs):
    """B_step = 8 * [A + 2n (P,Q in) + 2n (Vm,Va out) + 2m (flow, loading)
    + obs_dim + 8 scalars] bytes per env-step per instance."""
    return 8 * (fs.action_dim + 2 * fs.n + 2 * fs.n + 2 * fs.m + fs.obs_dim + 8)


def algorithmic_flops_per_iteration(fs):
    """~180 n flops per Newton iteration on a radial feeder (FBS: ~30 n per sweep)."""
    return 180 * fs.n


def cpu_baseline(fs, env_kwargs, budget_s=15.0):
    """The oracle timed on this box's host cores on a bounded sample of the same workload.
    Prefers the C/OpenMP port (oracle/liboracle_cpu.so) when it has been built, else the NumPy
    restatement on one core."""
    try:
        from oracle import oracle_c
        if oracle_c.available():
            return oracle_c.bench_env_steps(fs, env_kwargs, budget_s)
    except Exception as e:  # pragma: no cover - reported, not fatal
        print(f"[bench] C oracle unavailable ({e}); timing the NumPy oracle", file=sys.stderr)
    from oracle import oracle_np as O
    from tests.helpers import oracle_spec
    spec = oracle_spec(fs, stochastic_loads=env_kwargs["stochastic_loads"], weather_variation=env_kwargs["weather_variation"],
                       power_base=fs.base_power_va, solver=env_kwargs["solver"], tolerance=env_kwargs["tolerance"],
                       max_iterations=env_kwargs["max_iterations"], jacobian_mode="exact", zero_z="open")
    rng = np.random.default_rng(5678)
    n_done, t0 = 0, time.perf_counter()
    b = 0
    while time.perf_counter() - t0 < budget_s:
        _, st = O.env_reset(spec, seed=b, instance=b)
        st.time = 11.5 * 3600.0
        for _ in range(4):
            O.env_step(spec, st, rng.uniform(-1, 1, fs.action_dim))
            n_done += 1
        b += 1
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "env_steps/s", "cores": 1, "kind": "port",
            "sample": f"NumPy oracle, {b} instances x 4 steps of the same workload in {dt:.1f} s"}


def bench_unbalanced(args, device):
    """BASELINE.json config 5: 8500-node three-phase unbalanced FBS, batch 1024 -- load-flow solves/s.
    A "step" is one batched solve from a flat start; injections resident in HBM."""
    from grid_fed_rl_gym_amd.unbalanced import UnbalancedPowerFlow, ieee8500_like
    spec, Pn, Qn = ieee8500_like()
    B = args.batch or WORKLOADS[args.workload]["batch"]
    lam = np.random.default_rng(1234).uniform(0.5, 1.5, B)
    Pb, Qb = lam[:, None, None] * Pn[None], lam[:, None, None] * Qn[None]
    s = UnbalancedPowerFlow(tolerance=args.tolerance, max_iterations=args.max_iterations or 100, device=device)
    s.upload(spec, Pb, Qb)
    for _ in range(args.warmup):
        s.solve_device()
    s.synchronize(); s.timing_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.solve_device()
    s.synchronize()
    elapsed = time.perf_counter() - t0
    total_ms, launches = s.timing_read()
    sol = s.download()
    desc = s.describe()
    avg_ms = total_ms / max(launches, 1)
    mean_it = float(sol.iterations.mean())
    # per FBS iteration V and I of every phase conductor read + written once, 4 * 16 B each.
    # The survey's 4 * 48 n assumes three conductors per node; the kernel stores only the conductors that exist
    # (1.1 per node on this feeder), so the algorithmic figure counts those -- the 3-per-node figure is kept beside it.
    alg_bytes = 4 * 16 * desc["conductors"] * mean_it * B
    survey_bytes = 4 * 48 * spec.n * mean_it * B
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    try:
        if B == WORKLOADS[args.workload]["batch"]:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(f"{args.workload}:fbs3", {}).get("solve_bytes_per_launch")
    except Exception:
        traffic = None
    result = {"metric": "three-phase load-flow solves/sec (batched feeders)", "value": B * args.steps / elapsed, "unit": "solves/s",
              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": f"{spec.name}, 3-phase unbalanced FBS, batch={B}, per-instance loading U(0.5,1.5), tolerance {args.tolerance:g}",
                         "n_nodes": spec.n, "phase_conductors": desc["conductors"], "tree_levels": desc["levels"],
                         "max_level_width": desc["max_level_width"],
                         "batch_per_gpu": B, "kernel": "gs3_k_solve"},
              "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": "gs3_k_solve", "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_launch_at_3_conductors_per_node": survey_bytes, "mean_iterations": mean_it},
              "converged_fraction": float(sol.converged.mean()),
              "min_voltage_pu": float(np.abs(sol.voltages)[np.abs(sol.voltages) > 0].min())}
    if not args.no_cpu_baseline:
        try:
            from oracle import oracle_c as OC
            threads = max(1, min(OC.lib().orc_max_threads(), len(os.sched_getaffinity(0)), 16))
            nb = 4 * threads
            t1 = time.perf_counter(); done = 0
            while time.perf_counter() - t1 < 10.0:
                out = OC.solve3_batch(spec, Pb[:nb], Qb[:nb], tolerance=args.tolerance, threads=threads); done += nb
            dt = time.perf_counter() - t1
            ref = OC.solve3_batch(spec, Pb[:4], Qb[:4], tolerance=args.tolerance, threads=threads)
            result["accuracy"] = {"max_abs_dV_pu": float(np.max(np.abs(sol.voltages[:4] - ref["voltages"]))),
                                  "against": "C oracle (same algorithm; the reference has no 3-phase solver: parity unpinned)"}
            result["cpu_baseline"] = {"value": done / dt, "unit": "solves/s", "cores": threads, "kind": "port",
                                      "sample": f"C/OpenMP oracle, {done} solves (batches of {nb}) in {dt:.1f} s on {threads} threads"}
        except Exception as e:
            result["cpu_baseline"] = None
            print(f"[bench] C oracle unavailable: {e}", file=sys.stderr)
    print(json.dumps(result), flush=True)
    s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ieee123_b8192", choices=sorted(WORKLOADS))
    ap.add_argument("--solver", default="fbs", choices=["nr", "fbs"],
                    help="fbs = BASELINE.json config 3 (DistributionPowerFlow); nr = the reference's Newton-Raphson")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second measurement with the other solver")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the workload's)")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per 64-instance group (0 = auto)")
    ap.add_argument("--tolerance", type=float, default=1e-6, help="ablation only; the headline uses 1e-6")
    ap.add_argument("--max-iterations", type=int, default=0, help="ablation only; 0 = 50 (nr) / 100 (fbs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            print("[bench] --gpus > 1 needs `python -m torch.distributed.run --nproc-per-node N bench.py ...`", file=sys.stderr)
            sys.exit(2)

    lib = _lib.load()                # HIP runtime of /opt/rocm first; torch (if any) comes after
    n_dev = max(lib.gs_device_count(), 1)
    device = local_rank % n_dev      # ranks share a device only when rehearsing N > 1 on a smaller box
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    if args.workload == "ieee8500_3ph_b1024":
        if world != 1:
            print("[bench] the 3-phase workload is a single-GPU measurement", file=sys.stderr)
            sys.exit(2)
        bench_unbalanced(args, device)
        return
    wl = WORKLOADS[args.workload]
    fs = make_feeder(wl["feeder"])
    B = args.batch or wl["batch"]
    n_act = 8
    rng = np.random.default_rng(5678 + rank)
    actions = rng.uniform(-1, 1, (n_act, B, fs.action_dim))
    seeds = np.arange(rank * B, (rank + 1) * B, dtype=np.uint64)
    want_gather = world > 1 and not args.no_allgather and n_dev >= world
    kernel_names = {"nr_tree": "nr_tree", "nr_sparse_lu": "nr_lu", "fbs": "fbs", "nr_dense_pivot": "nr_dense",
                    "nr_tree_lds": "nr_tree_lds", "fbs_lds": "fbs_lds", "fbs_flow": "fbs_flow"}

    def env_kwargs_of(solver):
        return dict(stochastic_loads=True, weather_variation=True, solver=solver, tolerance=args.tolerance,
                    max_iterations=args.max_iterations or (50 if solver == "nr" else 100))

    def measure(solver, use_gather):
        """W untimed + K timed batched steps of one solver; returns the measurement as a dict."""
        env = P.BatchedGridEnvironment(fs, num_envs=B, jacobian="exact", zero_z="open", device=device,
                                       first_instance=rank * B, waves_per_group=args.waves, **env_kwargs_of(solver))
        h = env.handle
        desc = h.describe()
        h.upload_actions(actions)                               # inputs resident in HBM before the timed region
        env.reset(seed=seeds)
        st = env.get_state()
        st[:, env.state_column("time")] = 11.5 * 3600.0          # midday: loads near peak, PV producing
        env.set_state(st)
        if use_gather:
            import torch
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(_lib.Handle.comm_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(uid, src=0)
            h.comm_init(bytes(uid.numpy().tobytes()), rank, world)

        def one_step(k):
            h.step_device(k % n_act)
            if use_gather:
                h.allgather_obs(to_host=False)

        def barrier():
            h.synchronize()
            if dist is not None:
                dist.barrier()

        for k in range(args.warmup):
            one_step(k)
        barrier()
        # one HIP event pair on the kernel's stream around the K launches of the timed region (an event pair per launch
        # puts two marker packets between consecutive kernels: +4..5 us per step)
        h.timing_enable(True, span=True)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k)
        timing = h.timing_read()                                # closing event behind the last launch; waits for it
        h.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        h.timing_enable(False)
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        out = h.download_step(want_obs=False)                    # sanity of the timed work: every instance solved
        m = dict(solver=solver, elapsed=elapsed, timing=timing, desc=desc,
                 converged_fraction=float(out["power_flow_converged"].mean()),
                 mean_iterations=float(out["iterations"].mean()))
        if rank == 0 and solver == args.solver:
            # SafetyChecker + SafetyMonitor + quality gate on the device state, outside the timed step
            try:
                from grid_fed_rl_gym_amd.safety import PostStepChecks
                ck = PostStepChecks(env)
                for _ in range(3):
                    ck.run()
                h.synchronize(); ck.timing_read()
                for _ in range(20):
                    ck.run()
                ms, cnt = ck.timing_read()
                byts = B * ((fs.n + 3 * fs.m + 5) + fs.n) * 8 + B * (fs.n + fs.m)      # rows read, previous voltages written, masks written
                m["post_step_checks"] = {"kernel": "gs_k_checks", "avg_launch_us": 1e3 * ms / max(cnt, 1), "bytes_per_launch": byts,
                                         "GB_per_s": byts / (ms / max(cnt, 1) * 1e-3) / 1e9 if ms > 0 else None}
                # the same checks fused into the step kernel's epilogue (gs_checks_set_fused): cost = step time with - without
                def loop(nsteps):
                    for k in range(5):
                        h.step_device(k % n_act)
                    h.synchronize(); t1 = time.perf_counter()
                    for k in range(nsteps):
                        h.step_device(k % n_act)
                    h.synchronize()
                    return 1e6 * (time.perf_counter() - t1) / nsteps
                plain_us = loop(30)
                ck.set_fused(True)
                fused_us = loop(30)
                ck.set_fused(False)
                m["post_step_checks"]["fused_into_step_us"] = fused_us - plain_us
                ck.close()
            except Exception as e:                                 # never let the side measurement break the bench line
                m["post_step_checks"] = {"error": str(e)}
        if use_gather:
            h.comm_destroy()
        env.close()
        return m

    def accuracy(solver):
        """max |V| error of the HIP path against the CPU oracle on a 64-instance, 3-step sample of the workload."""
        try:
            from oracle import oracle_c as OC
            if not OC.available():
                return None
        except Exception:
            return None
        b = 64
        env = P.BatchedGridEnvironment(fs, num_envs=b, jacobian="exact", zero_z="open", device=device, **env_kwargs_of(solver))
        env.reset(seed=np.arange(b, dtype=np.uint64))
        st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
        net = OC.Net(fs)
        kw = env_kwargs_of(solver)
        # the oracle always runs Newton-Raphson (the reference's solver), tightly converged
        cfg = OC.config(solver="nr", jacobian="exact", max_iterations=50, tolerance=1e-10, stochastic_loads=True,
                        weather_variation=True, power_base=fs.base_power_va, threads=8)
        _, cst = OC.env_reset(net, cfg, b, np.arange(b, dtype=np.uint64)); cst[:, 0] = 11.5 * 3600.0
        dv = da = 0.0
        for k in range(3):
            obs, *_ = env.step(actions[k, :b])
            ref = OC.env_step(net, cfg, cst, actions[k, :b])["obs"]
            dv = max(dv, float(np.max(np.abs(obs[:, 0:2 * fs.n:2] - ref[:, 0:2 * fs.n:2]))))
            da = max(da, float(np.max(np.abs(obs[:, 1:2 * fs.n:2] - ref[:, 1:2 * fs.n:2]))))
        env.close()
        return {"max_abs_dVm_pu": dv, "max_abs_dVa_rad": da, "against": "C oracle, Newton-Raphson (reference algorithm) converged to 1e-10",
                "sample": f"{b} instances x 3 steps of the workload", "gpu_tolerance": kw["tolerance"]}

    # `value` is the sharded step itself: the ranks' instances are independent, no collective is on the path.  For N > 1
    # the same steps are then timed WITH north_star's observation all-gather after every step and reported beside it
    # ("with_obs_allgather"); that exchange is bound by xGMI, not by the step.
    main_m = measure(args.solver, False)
    gather_m = measure(args.solver, True) if want_gather else None
    other = None
    if world == 1 and not args.no_secondary:
        other = measure("nr" if args.solver == "fbs" else "fbs", False)

    if rank == 0:
        def summarize(m):
            steps_per_s = world * B * args.steps / m["elapsed"]
            solve = m["timing"]["solve"]
            avg_ms = solve["total_ms"] / max(solve["launches"], 1)
            return steps_per_s, avg_ms

        steps_per_s, avg_solve_ms = summarize(main_m)
        desc = main_m["desc"]
        bytes_step = algorithmic_bytes_per_step(fs)
        achieved_gbs = bytes_step * B / (avg_solve_ms * 1e-3) / 1e9 if avg_solve_ms > 0 else 0.0
        flops_it = algorithmic_flops_per_iteration(fs) if args.solver == "nr" else 30 * fs.n
        tflops = flops_it * main_m["mean_iterations"] * B / (avg_solve_ms * 1e-3) / 1e12 if avg_solve_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"{args.workload}:{args.solver}", {}).get("solve_bytes_per_launch")
            except Exception:
                traffic = None
        solver_text = {"nr": "Newton-Raphson (exact Jacobian)", "fbs": "forward/backward sweep (DistributionPowerFlow)"}
        result = {
            "metric": "env steps/sec (batched feeders)", "value": steps_per_s, "unit": "env_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * main_m["elapsed"] / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{fs.name}, batch={B} per GPU, {solver_text[args.solver]}, "
                                   f"stochastic loads + weather, tolerance {args.tolerance:g}",
                       "feeder_sha256": fs.sha256(), "n_buses": fs.n, "n_lines": fs.m, "obs_dim": fs.obs_dim,
                       "action_dim": fs.action_dim, "batch_per_gpu": B, "global_batch": world * B,
                       "solver": args.solver, "kernel": desc["kernel"], "waves_per_group": desc["waves_per_group"],
                       "tree_levels": desc["levels"], "obs_allgather_in_value": False,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "gs_k_step_" + kernel_names[desc["kernel"]],
                         "avg_launch_ms": avg_solve_ms, "avg_launch_method": "one HIP event pair on the kernel's stream around the K launches of the timed region / K",
                         "algorithmic_bytes_per_launch": bytes_step * B,
                         "fp64_valu": {"achieved_tflops": tflops, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                       "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                                       "mean_iterations": main_m["mean_iterations"]}},
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in main_m["timing"].items()},
            "converged_fraction": main_m["converged_fraction"],
        }
        if "post_step_checks" in main_m:
            result["post_step_checks"] = main_m["post_step_checks"]
        if gather_m is not None:
            g_sps, _ = summarize(gather_m)
            ms_with, ms_without = 1e3 * gather_m["elapsed"] / args.steps, 1e3 * main_m["elapsed"] / args.steps
            obs_bytes = B * fs.obs_dim * 8
            result["with_obs_allgather"] = {
                "value": g_sps, "unit": "env_steps/s", "ms_per_step": ms_with,
                "allgather_ms_per_step": ms_with - ms_without, "bytes_sent_per_rank_per_step": obs_bytes,
                "bytes_received_per_rank_per_step": (world - 1) * obs_bytes,
                "algbw_GB_per_s": world * obs_bytes / max(ms_with - ms_without, 1e-9) / 1e6,
                "note": "RCCL all-gather of the observation block after every step, overlapped with the next step; xGMI-bound"}
        if other is not None:
            o_sps, o_ms = summarize(other)
            result["also"] = {"solver": other["solver"], "kernel": "gs_k_step_" + kernel_names[other["desc"]["kernel"]],
                              "value": o_sps, "unit": "env_steps/s", "ms_per_step": 1e3 * other["elapsed"] / args.steps,
                              "avg_launch_ms": o_ms, "mean_iterations": other["mean_iterations"],
                              "converged_fraction": other["converged_fraction"]}
        if world == 1 and not args.no_cpu_baseline:
            result["accuracy"] = accuracy(args.solver)
            # the reference's CPU path is dense Newton-Raphson: that port is THE baseline; the CPU port of
            # the solver the GPU ran is reported next to it when it differs
            result["cpu_baseline"] = cpu_baseline(fs, env_kwargs_of("nr"), budget_s=12.0)
            if args.solver != "nr":
                result["cpu_baseline_same_solver"] = cpu_baseline(fs, env_kwargs_of(args.solver), budget_s=8.0)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
