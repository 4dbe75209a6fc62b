#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched AC power-flow env.step() on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--workload ieee123_b8192|ieee13_b4096|ieee8500_3ph_b1024]
                    [--solver nr|fbs] [--batch B] [--no-cpu-baseline] [--no-also]

One "step" = one batched env.step(): actions -> batteries/curtailment -> weather -> injections
-> AC load flow -> line flows -> frequency -> reward/flags -> observation block, for every
instance of the batch, with the K action batches already resident in HBM.  The default
workload is the configuration BASELINE.json's target is quoted on (configs[2]): the 123-bus radial
feeder, 8192 instances per GPU, forward/backward-sweep load flow, reference defaults for stochastic
loads and weather.  Protocol: W warm-up steps, then R regions of exactly K
steps, each bracketed by a barrier + device synchronisation on both sides; per region the maximum
over the ranks' clocks; `value` / `ms_per_step` are the MEDIAN region, p10 / p90 beside them.

The same line carries (N = 1): the Newton-Raphson measurement and BASELINE configs 2 and 5 ("also",
each with its own roofline), the device-resident rollout collector and the host-inclusive rate of
plain env.step() ("rollout", "with_host_io"), the accuracy of the GPU voltages / angles / flows
against the CPU oracle's Newton-Raphson, and the CPU baselines (all cores and one core).

For N > 1 the driver launches one process per GPU (torch.distributed.run); each rank owns a
contiguous block of instances (weak scaling, per-GPU batch fixed).  `value` is BASELINE config 4 as
written: the sharded step WITH the RCCL all-gather of the observation blocks after every step
(`config.obs_allgather_in_value`); the rate of the independent ranks without the exchange is
reported beside it ("without_obs_allgather").  The ranks rendezvous through a directory of small
files (grid_fed_rl_gym_amd/rendezvous.py): no torch anywhere in this process.
"""
import argparse
import json
import os
import platform
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
    "ieee123_b8192": dict(feeder="ieee123_like", batch=8192, solver="fbs"),       # BASELINE.json config 3 (and 4 for N > 1)
    "ieee13_b4096": dict(feeder="ieee13_like", batch=4096, solver="nr"),          # config 2
    "ieee8500_3ph_b1024": dict(feeder="ieee8500_like", batch=1024, solver="fbs3"),  # config 5 (solver only)
    # meshed feeders (Newton-Raphson through the sparse block LU): 123 buses with 26 loops, and the ScalableFeeder(123) recipe
    "meshed_loops26_b8192": dict(feeder="meshed_loops26", batch=8192, solver="nr"),
    "meshed_scalable_b8192": dict(feeder="meshed_scalable", batch=8192, solver="nr"),
}
KERNEL_NAMES = {"nr_tree": "nr_tree", "nr_sparse_lu": "nr_lu", "nr_dense_mfma": "nr_dense_mfma", "nr_sparse_lds": "nr_sparse_lds", "fbs": "fbs", "nr_dense_pivot": "nr_dense",
                "nr_tree_lds": "nr_tree_lds", "fbs_lds": "fbs_lds", "fbs_flow": "fbs_flow", "fbs_flow2": "fbs_flow2", "fbs_flow2h": "fbs_flow2h", "fbs_flow2s": "fbs_flow2s", "nr_flow2s": "nr_flow2s",
                "nr_flow2": "nr_flow2", "nr_mesh2": "nr_mesh2"}


def make_feeder(name):
    if name == "meshed_loops26":
        return P.random_meshed(123, 26, seed=1)       # 26 loops: the cycle count of the reference's IEEE123Bus (feeders/ieee_feeders.py:236-330)
    if name == "meshed_scalable":
        return P.scalable_like(123, seed=1)           # the recipe of ScalableFeeder(123) (feeders/synthetic.py:233): ~1000 lines
    return P.ieee123_like() if name == "ieee123_like" else P.ieee13_like("epsilon")


# which kernel sources a workload's step kernel is built from (a counter measurement is tied to these, see traffic_of)
_COMMON_SOURCES = ["gs_internal.h", "env_device.h", "fastmath.h", "kernels.h", "gridstep_abi.hip", "topology.cpp", "topology.h"]
KERNEL_SOURCES = {
    "ieee123_b8192:fbs": ["kernels_flow2.hip"], "ieee123_b8192:nr": ["kernels_flow2.hip"], "ieee13_b4096:nr": ["kernels_flow2.hip"],
    "ieee8500_3ph_b1024:fbs3": ["gridstep3.hip", "gridstep3_resident.h"],
    "meshed_loops26_b8192:nr": ["kernels_flow2.hip", "mesh_schedule.cpp", "mesh_schedule.h"], "meshed_scalable_b8192:nr": ["kernels_dense.hip", "kernels_solve.hip"],
}


def csrc_hash(key=None):
    """SHA-256 over the kernel sources a counter measurement under profiles/ was taken on: the files of `key`'s step kernel plus the
    shared headers and the launch code (every source when the key is unknown)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "grid_fed_rl_gym_amd", "csrc")
    names = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp")))
    if key in KERNEL_SOURCES:
        want = set(KERNEL_SOURCES[key]) | (set() if key.startswith("ieee8500") else set(_COMMON_SOURCES))
        names = [f for f in names if f in want]
    for f in names:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_step(fs):
    """B_step = 8 * [A + 2n (P,Q in) + 2n (Vm,Va out) + 2m (flow, loading)
    + obs_dim + 8 scalars] bytes per env-step per instance."""
    return 8 * (fs.action_dim + 2 * fs.n + 2 * fs.n + 2 * fs.m + fs.obs_dim + 8)


def algorithmic_flops_per_iteration(fs):
    """~180 n flops per Newton iteration on a radial feeder (FBS: ~30 n per sweep)."""
    return 180 * fs.n


def host_description():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return {"cpu_model": model or platform.processor(), "nproc": os.cpu_count(), "cpus_available_to_this_process": avail}


def traffic_of(key):
    """Fabric-side bytes per launch of the workload's step kernel from the PMC passes under profiles/ (tools/profile.sh,
    tools/collect_profile.py) -- a constant this run did not measure, so it is only reported while the kernel sources are the
    ones it was measured on (`csrc_sha` recorded with the entry); otherwise null."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(key, {})
        return e.get("solve_bytes_per_launch") if e.get("csrc_sha") == csrc_hash(key) else None
    except Exception:
        return None


def traffic_source(key):
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(key, {})
        cur = csrc_hash(key)
        return {"profile": e.get("profile"), "measured_on_csrc_sha": e.get("csrc_sha"), "this_build_csrc_sha": cur,
                "stale": e.get("csrc_sha") != cur, "bytes_when_measured": e.get("solve_bytes_per_launch")}
    except Exception:
        return None


def quantiles(xs):
    a = np.sort(np.asarray(xs, dtype=float))
    return float(np.median(a)), float(np.quantile(a, 0.1)), float(np.quantile(a, 0.9))


def cpu_baseline(fs, env_kwargs, budget_s=15.0, threads=None):
    """The oracle timed on this box's host cores on a bounded sample of the same workload.
    Prefers the C/OpenMP port (oracle/liboracle_cpu.so) when it has been built, else the NumPy
    restatement on one core."""
    try:
        from oracle import oracle_c
        if oracle_c.available():
            return oracle_c.bench_env_steps(fs, env_kwargs, budget_s, threads=threads)
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


def measure_unbalanced(args, device, with_cpu):
    """BASELINE.json config 5: 8500-node three-phase unbalanced FBS, batch 1024 -- load-flow solves/s.
    A "step" is one batched solve from a flat start; injections resident in HBM."""
    from grid_fed_rl_gym_amd.unbalanced import UnbalancedPowerFlow, ieee8500_like
    spec, Pn, Qn = ieee8500_like()
    B = (args.batch if args.workload == "ieee8500_3ph_b1024" else 0) or WORKLOADS["ieee8500_3ph_b1024"]["batch"]
    lam = np.random.default_rng(1234).uniform(0.5, 1.5, B)
    Pb, Qb = lam[:, None, None] * Pn[None], lam[:, None, None] * Qn[None]
    s = UnbalancedPowerFlow(tolerance=args.tolerance, max_iterations=args.max_iterations or 100, device=device)
    s.upload(spec, Pb, Qb)
    for _ in range(args.warmup):
        s.solve_device()
    s.synchronize(); s.timing_read()
    regions = []
    total_ms = launches = 0
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s.solve_device()
        s.synchronize()
        regions.append(time.perf_counter() - t0)
        ms, cnt = s.timing_read()
        total_ms += ms; launches += cnt
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
    traffic = traffic_of("ieee8500_3ph_b1024:fbs3") if B == WORKLOADS["ieee8500_3ph_b1024"]["batch"] else None
    kernel = "gs3_k_resident" if desc["kernel"] == "fbs3_resident" else "gs3_k_solve"
    # The resident kernel keeps V, I and the sweep's intermediates of an instance in the registers and LDS of one CU: what it
    # has to move through HBM is S once per sweep (one sweep fewer than the iteration count) and for the flat start, and V once per solve, 16 B each per
    # conductor.  The section-8(d) figure above models a kernel that streams the state every sweep, so against it this
    # kernel reads > 1 of the HBM roofline; what bounds it is FP64 vector issue and LDS gathers (the "valu" entry: flops
    # per conductor and iteration counted from gridstep3_resident.h -- 3 prefix sums 12, J and D 14, V 4, mismatch and
    # next current 44 -- against the 78.6 TFLOP/s FP64 vector peak).
    extra = {}
    if kernel == "gs3_k_resident":
        flops = 74.0 * desc["conductors"] * mean_it * B
        extra = {"bytes_per_launch_resident_design": 16.0 * desc["conductors"] * B * (mean_it + 1.0),
                 "valu": {"bound": "fp64 vector", "achieved": flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0, "peak": 78.6, "unit": "TFLOP/s",
                          "frac": flops / (avg_ms * 1e-3) / 1e12 / 78.6 if avg_ms > 0 else 0.0, "flops_per_conductor_iteration": 74},
                 "note": "frac > 1: the state of an instance never leaves its CU (registers + 156 KB of LDS); the algorithmic bytes are SURVEY 8(d)'s "
                         "streaming model, the design figure beside it is what this kernel must move"}
    med, p10, p90 = quantiles(regions)
    result = {"metric": "three-phase load-flow solves/sec (batched feeders)", "value": B * args.steps / med, "unit": "solves/s",
              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "repeats": args.repeats, "ms_per_step": 1e3 * med / args.steps,
              "ms_per_step_p10_p90": [1e3 * p10 / args.steps, 1e3 * p90 / args.steps],
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": f"{spec.name}, 3-phase unbalanced FBS, batch={B}, per-instance loading U(0.5,1.5), tolerance {args.tolerance:g}",
                         "n_nodes": spec.n, "phase_conductors": desc["conductors"], "tree_levels": desc["levels"],
                         "max_level_width": desc["max_level_width"],
                         "batch_per_gpu": B, "kernel": kernel, "threads_per_instance": desc.get("threads"),
                         "positions_per_thread": desc.get("positions_per_thread"), "lds_bytes": desc.get("lds_bytes"),
                         "parity": "unpinned: the reference has no three-phase solver (README prose only)"},
              "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": kernel, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_launch_at_3_conductors_per_node": survey_bytes, "mean_iterations": mean_it, **extra},
              "converged_fraction": float(sol.converged.mean()),
              "min_voltage_pu": float(np.abs(sol.voltages)[np.abs(sol.voltages) > 0].min())}
    if with_cpu:
        try:
            from oracle import oracle_c as OC
            threads = max(1, min(OC.lib().orc_max_threads(), len(os.sched_getaffinity(0)), 16))
            nb = 4 * threads
            t1 = time.perf_counter(); done = 0
            while time.perf_counter() - t1 < 10.0:
                OC.solve3_batch(spec, Pb[:nb], Qb[:nb], tolerance=args.tolerance, threads=threads); done += nb
            dt = time.perf_counter() - t1
            ref = OC.solve3_batch(spec, Pb[:4], Qb[:4], tolerance=args.tolerance, threads=threads)
            result["accuracy"] = {"max_abs_dV_pu": float(np.max(np.abs(sol.voltages[:4] - ref["voltages"]))),
                                  "against": "C oracle (same algorithm; the reference has no 3-phase solver: parity unpinned)"}
            result["cpu_baseline"] = {"value": done / dt, "unit": "solves/s", "cores": threads, "kind": "port",
                                      "sample": f"C/OpenMP oracle, {done} solves (batches of {nb}) in {dt:.1f} s on {threads} threads"}
        except Exception as e:
            result["cpu_baseline"] = None
            print(f"[bench] C oracle unavailable: {e}", file=sys.stderr)
    s.close()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=500,
                    help="untimed steps before the first timed region (default 500 = 20 ms: the GPU's clocks take that long to settle "
                         "under this load -- with 5 warm-up steps the median over the regions reads 5 %% low and the p10 10 %% low)")
    ap.add_argument("--repeats", type=int, default=120,
                    help="timed regions of exactly --steps steps each, every one bracketed by barrier + device synchronisation; median / p10 / p90 "
                         "are reported, and the median of the FIRST 40 regions beside them (`value_first_40_regions`: rounds 2-3's statistic).  "
                         "120 regions = 0.1 s of GPU time at the driver's `--steps 20 --warmup 5`: the GPU's clocks take ~20 ms to settle under "
                         "this load, so ten regions of 0.9 ms (round 2) all fell into the ramp and read 10 %% below the sustained rate, forty "
                         "(round 3) had their median at its end (5 %% below); with 120 the median is the sustained rate and the p10 shows the ramp")
    ap.add_argument("--workload", default="ieee123_b8192", choices=sorted(WORKLOADS))
    ap.add_argument("--solver", default="", choices=["", "nr", "fbs"],
                    help="fbs = BASELINE.json config 3 (DistributionPowerFlow); nr = the reference's Newton-Raphson; default: the workload's")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second measurement with the other solver")
    ap.add_argument("--no-also", action="store_true", help="skip BASELINE configs 2 and 5, the rollout and the host-inclusive measurements")
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

    lib = _lib.load()
    n_dev = max(lib.gs_device_count(), 1)
    device = local_rank % n_dev      # ranks share a device only when rehearsing N > 1 on a smaller box
    rz = None
    if world > 1:
        from grid_fed_rl_gym_amd.rendezvous import FileRendezvous
        rz = FileRendezvous(rank, world)

    if args.workload == "ieee8500_3ph_b1024":
        if world != 1:
            print("[bench] the 3-phase workload is a single-GPU measurement", file=sys.stderr)
            sys.exit(2)
        print(json.dumps(measure_unbalanced(args, device, not args.no_cpu_baseline)), flush=True)
        return
    wl = WORKLOADS[args.workload]
    solver0 = args.solver or wl["solver"]
    n_act = 8
    # fewer GPUs than ranks (a rehearsal of N > 1 on a smaller box): RCCL refuses two ranks on one device, so rank 0 holds
    # every rank's shard itself and the exchange runs through the in-process transport (gs_comm_init_loopback) -- the same
    # device code with device-to-device copies where ncclAllGather would cross xGMI.  Flagged, never a config-4 number.
    rehearsal = world > 1 and n_dev < world
    want_gather = world > 1 and not args.no_allgather and not rehearsal

    def env_kwargs_of(solver):
        return dict(stochastic_loads=True, weather_variation=True, solver=solver, tolerance=args.tolerance,
                    max_iterations=args.max_iterations or (50 if solver == "nr" else 100))

    def measure(fs, B, solver, use_gather, repeats, extras=False, warmup=None, steps=None, linear_solver=None):
        """W untimed + R x K timed batched steps of one solver; returns the measurement as a dict."""
        warmup = args.warmup if warmup is None else warmup
        steps = args.steps if steps is None else steps
        more = {} if linear_solver is None else {"linear_solver": linear_solver}
        rng = np.random.default_rng(5678 + rank)
        actions = rng.uniform(-1, 1, (n_act, B, fs.action_dim))
        seeds = np.arange(rank * B, (rank + 1) * B, dtype=np.uint64)
        env = P.BatchedGridEnvironment(fs, num_envs=B, jacobian="exact", zero_z="open", device=device,
                                       first_instance=rank * B, waves_per_group=args.waves, **env_kwargs_of(solver), **more)
        h = env.handle
        desc = h.describe()
        h.upload_actions(actions)                               # inputs resident in HBM before the timed region
        env.reset(seed=seeds)
        st = env.get_state()
        st[:, env.state_column("time")] = 11.5 * 3600.0          # midday: loads near peak, PV producing
        env.set_state(st)
        if use_gather:
            uid = rz.broadcast_bytes(_lib.Handle.comm_unique_id() if rank == 0 else None)
            h.comm_init(uid, rank, world)

        def one_step(k):
            h.step_device(k % n_act)
            if use_gather:
                h.allgather_obs(to_host=False)

        def barrier():
            h.synchronize()
            if rz is not None:
                rz.barrier()

        for k in range(warmup):
            one_step(k)
        regions, kernel_ms, kernel_launches = [], 0.0, 0
        for r in range(repeats):
            # one HIP event pair on the kernel's stream around the K launches of the region (an event pair per launch
            # puts two marker packets between consecutive kernels: +4..5 us per step)
            h.timing_enable(True, span=True)
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                one_step(warmup + k)
            timing = h.timing_read()                            # closing event behind the last launch; waits for it
            h.synchronize()
            elapsed = time.perf_counter() - t0
            h.timing_enable(False)
            if rz is not None:
                elapsed = rz.all_reduce_max(elapsed)
            regions.append(elapsed)
            kernel_ms += timing["solve"]["total_ms"]; kernel_launches += timing["solve"]["launches"]
        out = h.download_step(want_obs=False)                    # sanity of the timed work: every instance solved
        m = dict(solver=solver, regions=regions, kernel_ms=kernel_ms, kernel_launches=kernel_launches, desc=desc, B=B, fs=fs, steps=steps,
                 converged_fraction=float(out["power_flow_converged"].mean()),
                 mean_iterations=float(out["iterations"].mean()))
        if extras and rank == 0:
            m.update(side_measurements(env, fs, B, actions, solver))
        if desc.get("step_launches", 1) > 1 and rank == 0 and not use_gather:
            # how long ONE of the step's two dispatches runs (what rocprofv3 --kernel-trace reports per dispatch), from the
            # workgroups' own start / end stamps on the GPU's 100 MHz clock; armed only now, after everything that is timed
            try:
                os.environ["GS_STAMP_BLOCK_TIMES"] = "1"
                h.debug_stamps()
                for k in range(12):
                    h.step_device(k % n_act)
                nwg = int(desc["workgroups"]); t = h.debug_block_times(nwg).astype(np.int64); half = nwg // 2
                spans = [float(t[q, 1].max() - t[q, 0].min()) / 100.0 for q in (slice(0, half), slice(half, nwg))]
                m["dispatch"] = {"in_kernel_us": spans, "second_starts_after_first_us": float(t[half:, 0].min() - t[:half, 0].min()) / 100.0,
                                 "workgroups": [half, nwg - half], "algorithmic_bytes": [algorithmic_bytes_per_step(fs) * B * half // nwg,
                                                                                        algorithmic_bytes_per_step(fs) * B * (nwg - half) // nwg],
                                 "how": "first workgroup start to last workgroup end of each dispatch, last of 12 steps (gs_debug_block_times)"}
            except Exception as e:
                m["dispatch"] = {"error": str(e)}
            finally:
                os.environ.pop("GS_STAMP_BLOCK_TIMES", None)
        if use_gather:
            # the exchange checks itself, outside the timing: one more step and gather, every rank's own block against the
            # slot it landed in on every rank, and what RCCL itself says about the communicator (ranks, devices)
            try:
                from grid_fed_rl_gym_amd.sharding import verify_gathered_block
                h.step_device(0)
                full = h.allgather_obs(to_host=True)
                own = h.download_step()["obs"]
                m["gather_check"] = verify_gathered_block(full, own, rank, world, rz.all_gather_bytes)
                infos = [json.loads(b.decode()) for b in rz.all_gather_bytes(json.dumps(h.comm_info()).encode())]
                m["rccl"] = {"nranks": infos[0]["nranks"], "version": infos[0]["rccl_version"], "ranks": infos,
                             "every_rank_reports_the_same_nranks": len({i["nranks"] for i in infos}) == 1,
                             "distinct_devices": len({i["device_uuid"] for i in infos}),
                             "how": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion and hipDeviceGetUuid on every rank (gs_comm_info)"}
            except Exception as e:
                m["gather_check"] = {"gather_verified": False, "error": str(e)}
            h.comm_destroy()
        env.close()
        return m

    def measure_loopback(fs, B, solver, W, repeats, warmup, steps):
        """Every one of W ranks' shards on THIS device in this process; returns the rates with and without the exchange."""
        envs, hs = [], []
        for r in range(W):
            rng = np.random.default_rng(5678 + r)
            env = P.BatchedGridEnvironment(fs, num_envs=B, jacobian="exact", zero_z="open", device=device,
                                           first_instance=r * B, waves_per_group=args.waves, **env_kwargs_of(solver))
            env.handle.upload_actions(rng.uniform(-1, 1, (n_act, B, fs.action_dim)))
            env.reset(seed=np.arange(r * B, (r + 1) * B, dtype=np.uint64))
            st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
            envs.append(env); hs.append(env.handle)
        _lib.Handle.comm_init_loopback(hs)

        def run(with_gather):
            def one_step(k):
                for h in hs:
                    h.step_device(k % n_act)
                if with_gather:
                    _lib.Handle.allgather_obs_shards(hs)
            for k in range(warmup):
                one_step(k)
            regions = []
            for _ in range(repeats):
                for h in hs:
                    h.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    one_step(k)
                for h in hs:
                    h.synchronize()
                regions.append(time.perf_counter() - t0)
            return regions
        plain, gathered = run(False), run(True)
        out = hs[0].download_step(want_obs=False)
        # the exchanged block against the members' own observations, once, outside the timing
        full = _lib.Handle.allgather_obs_shards(hs, to_host=True)
        same = all(np.array_equal(full[r * B:(r + 1) * B], hs[r].download_step()["obs"]) for r in range(W))
        # the N-rank run's self-check (verify_gathered_block), rehearsed: every member's own gathered block, checksums "exchanged" in-process
        from grid_fed_rl_gym_amd.sharding import block_checksum
        owns = [h.download_step()["obs"] for h in hs]
        sums = [block_checksum(o) for o in owns]
        checks = []
        for r, h in enumerate(hs):
            mine = h.allgather_obs_download()
            bad = [q for q in range(W) if block_checksum(mine[q * B:(q + 1) * B]) != sums[q]]
            checks.append(not bad)
        info0 = hs[0].comm_info()
        desc = hs[0].describe()
        for env in envs:
            env.close()
        return dict(plain=plain, gathered=gathered, desc=desc, converged_fraction=float(out["power_flow_converged"].mean()),
                    gathered_block_equals_member_observations=bool(same), gather_verified=bool(all(checks)) and len(set(sums)) == W,
                    comm_info=info0)

    def loopback_summary(fs, B, W, m, steps):
        obs_bytes = B * (fs.obs_dim - 2 * fs.n_loads) * 8
        gp, gg = quantiles(m["plain"]), quantiles(m["gathered"])
        rate = lambda t: W * B * steps / t
        d_ms = 1e3 * (gg[0] - gp[0]) / steps
        return {"transport": "loopback", "world": W, "devices_used": 1, "batch_per_rank": B,
                "value": rate(gg[0]), "unit": "env_steps/s", "ms_per_step": 1e3 * gg[0] / steps,
                "value_p10_p90": [rate(gg[2]), rate(gg[1])],
                "without_obs_allgather": {"value": rate(gp[0]), "ms_per_step": 1e3 * gp[0] / steps},
                "exchange_ms_per_step": d_ms, "bytes_sent_per_rank_per_step": obs_bytes,
                "bytes_copied_on_the_device_per_step": W * W * obs_bytes,
                "gathered_block_equals_member_observations": m["gathered_block_equals_member_observations"],
                "gather_verified": m["gather_verified"], "comm_info_rank0": m["comm_info"],
                "converged_fraction": m["converged_fraction"],
                "note": f"all {W} ranks' shards are handles of ONE process on ONE GPU; the exchange is the RCCL transport's device code "
                        "(compaction, slot offsets, expansion, constant columns, double-buffer events) with device-to-device copies in "
                        "place of ncclAllGather -- a rehearsal of the N-rank data path, NOT a BASELINE config 4 measurement: there is no "
                        "xGMI in it and one GPU does every rank's work"}

    def side_measurements(env, fs, B, actions, solver):
        """Outside the timed regions: post-step checks, the rollout collector, plain env.step() with its host copies."""
        h = env.handle
        m = {}
        # SafetyChecker + SafetyMonitor + quality gate on the device state
        try:
            from grid_fed_rl_gym_amd.safety import PostStepChecks
            ck = PostStepChecks(env)
            ck.timing_enable(True)
            for _ in range(3):
                ck.run()
            h.synchronize(); ck.timing_read()
            for _ in range(20):
                ck.run()
            ms, cnt = ck.timing_read()
            ck.timing_enable(False)
            byts = B * ((fs.n + 3 * fs.m + 5) + fs.n) * 8 + B * (fs.n + fs.m)      # rows read, previous voltages written, masks written
            m["post_step_checks"] = {"kernel": "gs_k_checks", "avg_launch_us": 1e3 * ms / max(cnt, 1), "bytes_per_launch": byts,
                                     "GB_per_s": byts / (ms / max(cnt, 1) * 1e-3) / 1e9 if ms > 0 else None}

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
        except Exception as e:                                 # never let a side measurement break the bench line
            m["post_step_checks"] = {"error": str(e)}
        if args.no_also:
            return m
        # the device-resident rollout collector (gs_rollout): T steps + bookkeeping + in-place resets,
        # random actions drawn on the device, nothing on the host in between
        try:
            T = max(args.steps, 64)                                               # a rollout length that amortises the per-call fixed cost (two 45 MB slot copies, host wake-up)
            h.rollout(T, "random", seed=1); h.synchronize()                       # allocation + warm-up
            rates = []
            for r in range(5):
                t1 = time.perf_counter()
                h.rollout(T, "random", seed=2 + r)
                h.synchronize()
                rates.append(B * T / (time.perf_counter() - t1))
            med, p10, p90 = quantiles(rates)
            t1 = time.perf_counter()
            d = h.rollout_download()
            dt_dl = time.perf_counter() - t1
            nbytes = sum(v.nbytes for v in d.values() if isinstance(v, np.ndarray))
            m["rollout"] = {"env_steps_per_s": med, "p10_p90": [p10, p90], "T": T, "policy": "random actions drawn on the device",
                            "what": "gs_rollout: T fused steps, each writing its observation block into the next slot of obs_seq[T+1][B][obs_dim]; "
                                    "reward / done arrays, terminal-observation side list and in-place resets of finished instances ride in the same launches "
                                    "(one kernel per step); random actions end an episode by truncation every 10-20 steps, so resets are part of the figure; no host copy",
                            "finished_episodes": d["n_terminal"],
                            "download_once_at_the_end": {"seconds": dt_dl, "bytes": nbytes, "GB_per_s": nbytes / dt_dl / 1e9}}
        except Exception as e:
            m["rollout"] = {"error": str(e)}
        # what a caller of BatchedGridEnvironment.step() gets: actions from host memory in, the whole observation block and
        # the info arrays out, every step (PCIe-bound; never `value`).  The environment's default: views of page-locked buffer
        # sets that are reused once the caller has dropped what it got from them (Handle.use_recycled_outputs)
        def host_io(label, what):
            try:
                env.step(actions[0]); env.step(actions[1])
                ts = []
                for k in range(10):
                    t1 = time.perf_counter()
                    env.step(actions[k % n_act])
                    ts.append(time.perf_counter() - t1)
                med, p10, p90 = quantiles(ts)
                m[label] = {"env_steps_per_s": B / med, "ms_per_step": 1e3 * med, "ms_per_step_p10_p90": [1e3 * p10, 1e3 * p90],
                            "GB_per_s_of_observations": B * fs.obs_dim * 8 / med / 1e9, "what": what}
            except Exception as e:
                m[label] = {"error": str(e)}
        nd_cols = fs.obs_dim - 2 * fs.n_loads
        host_io("with_host_io", f"BatchedGridEnvironment.step() as it is by default; per step: actions [B][{fs.action_dim}] f64 host->device ({B * fs.action_dim * 8} B), "
                                f"observations [B][{fs.obs_dim}] f64: the {nd_cols} changing columns device->host ({B * nd_cols * 8} B, written by a kernel straight into the "
                                f"page-locked array; the {2 * fs.n_loads} constant columns -- static load powers -- were put there once, gs_host_obs_bind), "
                                f"reward / flags / info arrays (~{B * 70} B); the arrays returned are views of page-locked buffer sets, a set reused only when the caller "
                                "holds nothing of it any more (GB_per_s_of_observations counts the whole array the caller sees)")
        saved, h._recycle = getattr(h, "_recycle", None), None
        host_io("with_host_io_fresh_arrays", "recycle_host_buffers=False: a freshly allocated pageable NumPy array per output and step (rounds 1-3's default)")
        h._recycle = saved
        # the same call with the output arrays in page-locked memory (BatchedGridEnvironment(pinned_host_buffers=True):
        # step() returns views of two rotating pinned buffer sets instead of fresh arrays)
        try:
            h.use_pinned_outputs()
            for k in range(4):
                env.step(actions[k % n_act])
            ts = []
            for k in range(10):
                t1 = time.perf_counter()
                env.step(actions[k % n_act])
                ts.append(time.perf_counter() - t1)
            med, p10, p90 = quantiles(ts)
            m["with_host_io_pinned"] = {"env_steps_per_s": B / med, "ms_per_step": 1e3 * med, "ms_per_step_p10_p90": [1e3 * p10, 1e3 * p90],
                                        "GB_per_s_of_observations": B * fs.obs_dim * 8 / med / 1e9,
                                        "what": "as with_host_io, outputs in page-locked host buffers (gs_host_alloc), reused every other step"}
        except Exception as e:
            m["with_host_io_pinned"] = {"error": str(e)}
        # opt-in: the observation block as float32 (the dtype the reference declares for its observation space, grid_env.py:346),
        # rounded on the device -- never the headline, never `with_host_io`
        try:
            e32 = P.BatchedGridEnvironment(fs, num_envs=B, jacobian="exact", zero_z="open", device=device, obs_dtype=np.float32, **env_kwargs_of(solver))
            e32.reset(seed=np.arange(B, dtype=np.uint64))
            for k in range(4):
                o32 = e32.step(actions[k % n_act]); del o32
            ts = []
            for k in range(10):
                t1 = time.perf_counter()
                o32 = e32.step(actions[k % n_act]); del o32
                ts.append(time.perf_counter() - t1)
            e32.close()
            med, p10, p90 = quantiles(ts)
            m["with_host_io_float32_observations"] = {"env_steps_per_s": B / med, "ms_per_step": 1e3 * med, "ms_per_step_p10_p90": [1e3 * p10, 1e3 * p90],
                                                      "what": "BatchedGridEnvironment(obs_dtype=np.float32).step(): as with_host_io, the observation block rounded to float32 on the "
                                                              "device and copied whole (opt-in; every other output unchanged)"}
        except Exception as e:
            m["with_host_io_float32_observations"] = {"error": str(e)}
        return m

    def accuracy(fs, solver):
        """Voltage, angle and line-flow differences of the HIP path on a 64-instance, 3-step sample of the workload against
        (1) the reference's algorithm at the reference's own settings -- Newton-Raphson, tolerance 1e-6 (power_flow.py:81):
        the figure north_star's 1e-6 pu bar is about --, (2) the same algorithm converged to 1e-10 (what the tolerance
        leaves behind, for both), (3) the GPU's own algorithm at the GPU's own tolerance (the implementation alone)."""
        try:
            from oracle import oracle_c as OC
            if not OC.available():
                return None
        except Exception:
            return None
        b = 64
        rng = np.random.default_rng(5678 + rank)
        actions = rng.uniform(-1, 1, (n_act, b, fs.action_dim))
        env = P.BatchedGridEnvironment(fs, num_envs=b, jacobian="exact", zero_z="open", device=device, **env_kwargs_of(solver))
        env.reset(seed=np.arange(b, dtype=np.uint64))
        st = env.get_state(); st[:, env.state_column("time")] = 11.5 * 3600.0; env.set_state(st)
        net = OC.Net(fs)
        kw = env_kwargs_of(solver)
        common = dict(jacobian="exact", stochastic_loads=True, weather_variation=True, power_base=fs.base_power_va, threads=8)
        cfgs = {"ref": OC.config(solver="nr", max_iterations=50, tolerance=1e-6, **common),
                "exact": OC.config(solver="nr", max_iterations=50, tolerance=1e-10, **common),
                "same": OC.config(solver=solver, max_iterations=kw["max_iterations"], tolerance=kw["tolerance"], **common)}
        states = {}
        for k, c in cfgs.items():
            _, cs = OC.env_reset(net, c, b, np.arange(b, dtype=np.uint64)); cs[:, 0] = 11.5 * 3600.0
            states[k] = cs
        o_f = 2 * fs.n
        cols = {"Vm": slice(0, 2 * fs.n, 2), "Va": slice(1, 2 * fs.n, 2), "flow": slice(o_f, o_f + 2 * fs.m, 2)}
        worst = {k: {q: 0.0 for q in cols} for k in ("ref", "exact", "same", "ref_vs_exact")}
        fmax = 0.0
        for t in range(3):
            obs, *_ = env.step(actions[t])
            o = {k: OC.env_step(net, cfgs[k], states[k], actions[t])["obs"] for k in cfgs}
            for q, sl in cols.items():
                for k in cfgs:
                    worst[k][q] = max(worst[k][q], float(np.max(np.abs(obs[:, sl] - o[k][:, sl]))))
                worst["ref_vs_exact"][q] = max(worst["ref_vs_exact"][q], float(np.max(np.abs(o["ref"][:, sl] - o["exact"][:, sl]))))
            fmax = max(fmax, float(np.max(np.abs(o["exact"][:, cols["flow"]]))))
        env.close()
        fmt = lambda w: {"max_abs_dVm_pu": w["Vm"], "max_abs_dVa_rad": w["Va"], "max_abs_dflow_pu": w["flow"]}
        out = fmt(worst["ref"])
        out.update(against="C oracle running the reference's algorithm at the reference's settings: Newton-Raphson, tolerance 1e-6 (power_flow.py:81)",
                   bar_pu=1e-6, largest_flow_pu=fmax,
                   against_converged_solution=dict(fmt(worst["exact"]), against="C oracle, Newton-Raphson converged to 1e-10",
                                                   reference_settings_vs_converged=fmt(worst["ref_vs_exact"])),
                   same_algorithm_same_tolerance=dict(fmt(worst["same"]), against=f"C oracle running the GPU's solver ({solver}) at the GPU's tolerance"),
                   sample=f"{b} instances x 3 steps of the workload", gpu_tolerance=kw["tolerance"])
        # the bar, machine-readable: north_star words it against the reference solver's own output ("vs_reference_settings");
        # for the sweep solver the line flows miss that by the reference iterate's own distance from convergence (the
        # documented deviation, DESIGN.md section 2: tests/test_gpu_fullsize.py keeps the strict form as an expected failure)
        inside = lambda w: {q: bool(w[q] < 1e-6) for q in ("Vm", "Va", "flow")}
        out["within_bar"] = {"vs_reference_settings": inside(worst["ref"]), "vs_converged": inside(worst["exact"]),
                             "reference_settings_vs_converged": inside(worst["ref_vs_exact"])}
        return out

    def summarize(m, n_ranks):
        med, p10, p90 = quantiles(m["regions"])
        K = m.get("steps", args.steps)
        sps = [n_ranks * m["B"] * K / t for t in m["regions"]]
        s_med, s_p10, s_p90 = quantiles(sps)
        return dict(value=s_med, value_p10_p90=[s_p10, s_p90], value_first_40_regions=quantiles(sps[:40])[0], value_first_region=sps[0], ms_per_step=1e3 * med / K,
                    ms_per_step_p10_p90=[1e3 * p10 / K, 1e3 * p90 / K],
                    avg_launch_ms=m["kernel_ms"] / max(m["kernel_launches"], 1))

    def roofline_of(m, s, workload_key):
        fs, B = m["fs"], m["B"]
        bytes_step = algorithmic_bytes_per_step(fs)
        avg = s["avg_launch_ms"]
        achieved = bytes_step * B / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        flops_it = algorithmic_flops_per_iteration(fs) if m["solver"] == "nr" else 30 * fs.n
        tflops = flops_it * m["mean_iterations"] * B / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic_of(f"{workload_key}:{m['solver']}") if B == WORKLOADS.get(workload_key, {}).get("batch") else None,
                "traffic_source": traffic_source(f"{workload_key}:{m['solver']}"),
                "kernel": "gs_k_step_" + KERNEL_NAMES.get(m["desc"]["kernel"], m["desc"]["kernel"]),
                "avg_launch_ms": avg, "avg_launch_method": "one HIP event pair on the kernel's stream around the K launches of each timed region / K, mean over the regions",
                "algorithmic_bytes_per_launch": bytes_step * B,
                **({} if m["desc"].get("step_launches", 1) == 1 else {
                    "dispatches_per_launch": m["desc"]["step_launches"],
                    "note": "one step = TWO concurrent dispatches of the kernel, each half of the workgroups, on two streams that run about half a "
                            "step out of phase (one half's LDS-bound solver beside the other's VALU-bound prologue / epilogue); `avg_launch_ms`, "
                            "`achieved` and `algorithmic_bytes_per_launch` are per STEP (both dispatches); rocprofv3 --kernel-trace lists the "
                            "dispatches separately, each longer than half a step because they overlap (`per_dispatch`)",
                    "per_dispatch": m.get("dispatch")}),
                "fp64_valu": {"achieved_tflops": tflops, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                              "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, "mean_iterations": m["mean_iterations"]}}

    fs = make_feeder(wl["feeder"])
    B = args.batch or wl["batch"]
    if rehearsal:
        if rank == 0:
            m = measure_loopback(fs, B, solver0, world, max(3, args.repeats // 2), min(args.warmup, 50), args.steps)
            lb = loopback_summary(fs, B, world, m, args.steps)
            result = {"metric": "env steps/sec (batched feeders)", "value": lb["value"], "unit": "env_steps/s", "n_gpus": world,
                      "steps": args.steps, "warmup": min(args.warmup, 50), "ms_per_step": lb["ms_per_step"], "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "transport": "loopback", "rehearsal": True,
                      "config": {"workload": f"{fs.name}, batch={B} per rank, REHEARSAL of {world} ranks on {n_dev} device(s) through the in-process transport",
                                 "batch_per_gpu": B, "global_batch": world * B, "solver": solver0, "kernel": m["desc"]["kernel"],
                                 "obs_allgather_in_value": True, "parallelism": f"batch-sharded x{world} (loopback)"},
                      "loopback": lb, "roofline": None, "cpu_baseline": None, "host": host_description()}
            print(json.dumps(result), flush=True)
        rz.barrier()
        rz.close()
        return
    main_m = measure(fs, B, solver0, False, args.repeats, extras=(world == 1))
    # The exchange is the one part of an N-rank run no box of this pool could rehearse over real links: if setting it up fails the same
    # way on every rank (no librccl, a communicator that cannot be formed), the line still goes out -- the sharded steps without the
    # exchange, flagged -- instead of nothing.  The ranks agree on that through fixed-name rendezvous entries (not the sequence-numbered
    # collectives, which a rank that raised half-way is out of step with).
    gather_m, gather_error = None, None
    if want_gather:
        try:
            gather_m = measure(fs, B, solver0, True, args.repeats)
        except Exception as e:                                   # noqa: BLE001 -- reported in the line
            gather_error = f"{type(e).__name__}: {e}"[:300]
        try:
            rz._put(f"gather_status.{rank}", (gather_error or "ok").encode())
            states = [rz._get(f"gather_status.{r}", timeout=180.0).decode() for r in range(world)]
        except Exception as e:                                   # noqa: BLE001
            states = [f"rank {rank}: no status from a peer ({type(e).__name__})"]
        bad = [f"rank {r}: {st}" for r, st in enumerate(states) if st != "ok"]
        if bad:
            gather_m, gather_error = None, "; ".join(bad)[:600]
    other = None
    if world == 1 and not args.no_secondary:
        other = measure(fs, B, "nr" if solver0 == "fbs" else "fbs", False, max(3, args.repeats // 2))

    if rank == 0:
        plain = summarize(main_m, world)
        head_m, head = (gather_m, summarize(gather_m, world)) if gather_m is not None else (main_m, plain)
        desc = main_m["desc"]
        solver_text = {"nr": "Newton-Raphson (exact Jacobian)", "fbs": "forward/backward sweep (DistributionPowerFlow)"}
        result = {
            "metric": "env steps/sec (batched feeders)", "value": head["value"], "unit": "env_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": args.repeats,
            "ms_per_step": head["ms_per_step"], "value_p10_p90": head["value_p10_p90"], "ms_per_step_p10_p90": head["ms_per_step_p10_p90"],
            "value_first_40_regions": head["value_first_40_regions"], "value_first_region": head["value_first_region"],
            "statistic": "median over the timed regions of --steps steps each (max over ranks per region); value_first_40_regions: the median of the "
                         "first 40 of them (the GPU's clocks are still ramping there), value_first_region: the first region alone",
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{fs.name}, batch={B} per GPU, {solver_text[solver0]}, "
                                   f"stochastic loads + weather, tolerance {args.tolerance:g}",
                       "feeder_sha256": fs.sha256(), "n_buses": fs.n, "n_lines": fs.m, "obs_dim": fs.obs_dim,
                       "action_dim": fs.action_dim, "batch_per_gpu": B, "global_batch": world * B,
                       "solver": solver0, "kernel": desc["kernel"], "waves_per_group": desc["waves_per_group"],
                       "instances_per_workgroup": desc.get("instances_per_workgroup", 64), "workgroups": desc.get("workgroups", desc["groups"]),
                       "tree_levels": desc["levels"], "obs_allgather_in_value": gather_m is not None,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": roofline_of(main_m, plain, args.workload),
            "converged_fraction": head_m["converged_fraction"],
            "host": host_description(),
        }
        for k in ("post_step_checks", "rollout", "with_host_io", "with_host_io_fresh_arrays", "with_host_io_pinned", "with_host_io_float32_observations"):
            if k in main_m:
                result[k] = main_m[k]
        if world > 1 and want_gather and gather_m is None:
            result["obs_allgather_error"] = gather_error
            result["gather_verified"] = False
        if world > 1:
            obs_bytes = B * (fs.obs_dim - 2 * fs.n_loads) * 8
            result["without_obs_allgather"] = {"value": plain["value"], "unit": "env_steps/s", "ms_per_step": plain["ms_per_step"],
                                               "value_p10_p90": plain["value_p10_p90"],
                                               "note": "the same sharded steps with no exchange: the ranks are independent (a learner that consumes its own shard)"}
            if gather_m is not None:
                result["gather_verified"] = bool(gather_m.get("gather_check", {}).get("gather_verified", False))
                result["gather_check"] = gather_m.get("gather_check")
                result["rccl"] = gather_m.get("rccl")
                d_ms = head["ms_per_step"] - plain["ms_per_step"]
                result["obs_allgather"] = {"allgather_ms_per_step": d_ms, "bytes_sent_per_rank_per_step": obs_bytes,
                                           "bytes_received_per_rank_per_step": (world - 1) * obs_bytes,
                                           "algbw_GB_per_s": world * obs_bytes / max(d_ms, 1e-9) / 1e6,
                                           "note": "RCCL all-gather of the changing observation columns (the static load columns are rank-independent and never sent) "
                                                   "after every step, on its own stream, overlapped with the next step; xGMI-bound"}
            else:
                result["obs_allgather"] = {"skipped": "fewer devices than ranks (rehearsal on a smaller box) or --no-allgather: value has no exchange in it"}
        if other is not None:
            o = summarize(other, 1)
            result["also"] = {"solver": other["solver"], "kernel": "gs_k_step_" + KERNEL_NAMES.get(other["desc"]["kernel"], other["desc"]["kernel"]),
                              "value": o["value"], "unit": "env_steps/s", "ms_per_step": o["ms_per_step"], "value_p10_p90": o["value_p10_p90"],
                              "value_first_40_regions": o["value_first_40_regions"], "avg_launch_ms": o["avg_launch_ms"], "mean_iterations": other["mean_iterations"],
                              "converged_fraction": other["converged_fraction"], "roofline": roofline_of(other, o, args.workload)}
        if world == 1 and not args.no_also and args.workload == "ieee123_b8192":
            # BASELINE configs 2 and 5 in the same line, each with its own roofline
            try:
                w2 = WORKLOADS["ieee13_b4096"]
                fs2 = make_feeder(w2["feeder"])
                m2 = measure(fs2, w2["batch"], w2["solver"], False, max(3, args.repeats // 2))
                s2 = summarize(m2, 1)
                result["also_config2"] = {"workload": f"{fs2.name}, batch={w2['batch']}, Newton-Raphson (BASELINE config 2)", "value": s2["value"],
                                          "unit": "env_steps/s", "ms_per_step": s2["ms_per_step"], "value_p10_p90": s2["value_p10_p90"],
                                          "value_first_40_regions": s2["value_first_40_regions"], "kernel": m2["desc"]["kernel"], "waves_per_group": m2["desc"]["waves_per_group"],
                                          "workgroups": m2["desc"].get("workgroups", m2["desc"]["groups"]), "mean_iterations": m2["mean_iterations"],
                                          "converged_fraction": m2["converged_fraction"], "roofline": roofline_of(m2, s2, "ieee13_b4096")}
                # the rollout collector on the same feeder (gs_rollout: T fused steps with in-place resets, no host in between)
                try:
                    env2 = P.BatchedGridEnvironment(fs2, num_envs=w2["batch"], jacobian="exact", zero_z="open", device=device, **env_kwargs_of(w2["solver"]))
                    env2.reset(seed=np.arange(w2["batch"], dtype=np.uint64))
                    h2 = env2.handle
                    T2 = 256
                    h2.rollout(T2, "random", seed=1); h2.synchronize()
                    rr = []
                    for r in range(5):
                        t1 = time.perf_counter()
                        h2.rollout(T2, "random", seed=2 + r); h2.synchronize()
                        rr.append(w2["batch"] * T2 / (time.perf_counter() - t1))
                    med2, lo2, hi2 = quantiles(rr)
                    result["also_config2"]["rollout"] = {"env_steps_per_s": med2, "p10_p90": [lo2, hi2], "T": T2, "policy": "random actions drawn on the device",
                                                         "finished_episodes": h2.rollout_download(want=("terminals",))["n_terminal"]}
                    env2.close()
                except Exception as e:
                    result["also_config2"]["rollout"] = {"error": str(e)}
            except Exception as e:
                result["also_config2"] = {"error": str(e)}
            try:
                sub = argparse.Namespace(**vars(args)); sub.workload = "ieee123_b8192"; sub.batch = 0; sub.steps = min(args.steps, 20); sub.repeats = 3
                r5 = measure_unbalanced(sub, device, False)
                result["also_config5"] = {k: r5[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "converged_fraction")}
            except Exception as e:
                result["also_config5"] = {"error": str(e)}
        if world == 1 and not args.no_also and args.workload == "ieee123_b8192":
            # Meshed feeders: Newton-Raphson with a sparse block LU (north_star: "Jacobian build, sparse LU/Cholesky solve";
            # the reference solves them densely, power_flow.py:186-190).  Two graphs of the reference's own generators' kind:
            # 123 buses with 26 loops (IEEE123Bus's cycle count, feeders/ieee_feeders.py:236-330) and the ScalableFeeder(123)
            # recipe (feeders/synthetic.py:233: ~1000 lines, a graph whose block LU fills in almost completely).
            result["also_meshed"] = {}
            for key, mk, Bm, K, W_ in (("loops26", lambda: make_feeder("meshed_loops26"), B, min(args.steps, 20), min(args.warmup, 20)),
                                       ("scalable", lambda: make_feeder("meshed_scalable"), B, 5, 2)):
                try:
                    fsm = mk()
                    mm = measure(fsm, Bm, "nr", False, max(3, args.repeats // (4 if key == "loops26" else 12)), warmup=W_, steps=K)      # (as the headline: the median over the regions; three regions sat on the clock ramp)
                    sm = summarize(mm, 1)
                    rl = roofline_of(mm, sm, f"meshed_{key}_b8192")
                    pairs = int(mm["desc"].get("lu_pairs", 0))
                    # per Newton iteration and instance: one 2x2 (A_ik D^-1) A_kj product and subtraction per scheduled pair
                    # (2 x 12 + 4 flops), one 2x2 inverse per pivot, the Jacobian blocks (~40 flops per Ybus entry)
                    fl_it = 28 * pairs + 30 * fsm.n + 40 * int(mm["desc"]["nnz"])
                    tf = fl_it * mm["mean_iterations"] * Bm / (sm["avg_launch_ms"] * 1e-3) / 1e12 if sm["avg_launch_ms"] > 0 else 0.0
                    rl["fp64_valu"] = {"achieved_tflops": tf, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                                       "flops_per_iteration_per_instance": fl_it, "mean_iterations": mm["mean_iterations"],
                                       "how": "28 flops per scheduled pair update (host schedule, gs_describe lu_pairs) + 30 per pivot + 40 per Ybus entry"}
                    if mm["desc"]["kernel"] == "nr_dense_mfma":
                        # dense block LU on the matrix cores: the roofline that binds is FP64 MFMA.  Algorithmic flops:
                        # (2/3) N^3 + 2 N^2 per Newton solve, N = 2 (n - 1), one solve per iteration but the last (the converged check)
                        N_ = 2 * (fsm.n - 1)
                        fl_solve = (2.0 / 3.0) * N_ ** 3 + 2.0 * N_ ** 2
                        solves = max(mm["mean_iterations"] - 1.0, 0.0)
                        tfm = fl_solve * solves * Bm / (sm["avg_launch_ms"] * 1e-3) / 1e12 if sm["avg_launch_ms"] > 0 else 0.0
                        rl = {"bound": "mfma", "achieved": tfm, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfm / FP64_VECTOR_PEAK_TFLOPS,
                              "traffic": None, "kernel": ("gs_k_nr_dense_mfma2" if mm["desc"].get("dense_form") == "block_row" else "gs_k_nr_dense_mfma") + " (between gs_k_pre_nr_dmfma and gs_k_post_nr_dmfma)",
                              "dense_form": mm["desc"].get("dense_form"), "workgroups": mm["desc"].get("dense_workgroups"), "lds_bytes_per_workgroup": mm["desc"].get("dense_lds_bytes"),
                              "avg_launch_ms": sm["avg_launch_ms"], "avg_launch_method": rl["avg_launch_method"] + "; a step here is three launches (prologue, dense Newton-Raphson, epilogue): the span covers all three",
                              "algorithmic_flops_per_launch": fl_solve * solves * Bm,
                              "how": f"(2/3) N^3 + 2 N^2 = {fl_solve:.3g} flops per Newton solve at N = {N_}, {solves:.2f} solves per step (iterations - 1); the peak is the dense FP64 "
                                     "MFMA rate (= the FP64 vector rate on this part); the first solve of a step reuses the handle's flat-start factors, so the executed flops are about half",
                              "hbm_view": {"achieved_GB_per_s": rl["achieved"], "frac": rl["frac"], "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"]}}
                        # what the matrix cores actually execute: a factorisation is 64^3 block products (padded to 64-wide blocks), the updates
                        # sum_j sum_i min(i, j) of them plus one per off-diagonal block for the scaling; one factorisation per solve but the first
                        NBd = (2 * (fsm.n - 1) + 63) // 64
                        n_prod = sum(min(i, j) for j in range(NBd) for i in range(NBd)) + NBd * (NBd - 1) // 2
                        facts = max(solves - 1.0, 0.0)
                        ex = n_prod * 2.0 * 64 ** 3 * facts * Bm / (sm["avg_launch_ms"] * 1e-3) / 1e12 if sm["avg_launch_ms"] > 0 else 0.0
                        rl["executed_mfma"] = {"achieved_tflops": ex, "frac": ex / FP64_VECTOR_PEAK_TFLOPS, "block_products_per_factorisation": n_prod,
                                               "flops_per_factorisation": n_prod * 2.0 * 64 ** 3, "factorisations_per_step": facts,
                                               "how": "64 x 64 x 64 MFMA block products issued per factorisation (updates + one per off-diagonal block), one factorisation per "
                                                      "solve except the first of a step (flat-start table); the Gauss-Jordan inversions of the diagonal blocks, the substitutions "
                                                      "and the assembly run on the vector units and are not counted"}
                        rl["traffic_split"] = ("none available: FETCH_SIZE / WRITE_SIZE count the L2's fabric-side requests, Infinity-Cache hits included (MI355X guide, HBM "
                                               "section); no gfx950 counter exposed by rocprofv3 separates the workgroups' scratch that stays in the Infinity Cache from HBM")
                    if mm["desc"]["kernel"] == "nr_sparse_lds":
                        rl["kernel"] = "gs_k_nr_sparse_lds (between gs_k_pre_nr_dmfma and gs_k_post_nr_dmfma)"
                        rl["avg_launch_method"] = rl.get("avg_launch_method", "") + "; a step here is three launches (prologue, Newton-Raphson with the sparse LU in LDS, epilogue): the span covers all three"
                    rl["traffic"] = traffic_of(f"meshed_{key}_b8192:nr") if Bm == 8192 else None
                    rl["traffic_source"] = traffic_source(f"meshed_{key}_b8192:nr")
                    entry = {"workload": f"{fsm.name}, batch={Bm}, Newton-Raphson (exact Jacobian), stochastic loads + weather",
                             "feeder_sha256": fsm.sha256(), "n_buses": fsm.n, "n_lines": fsm.m, "obs_dim": fsm.obs_dim, "action_dim": fsm.action_dim,
                             "value": sm["value"], "unit": "env_steps/s", "ms_per_step": sm["ms_per_step"], "value_p10_p90": sm["value_p10_p90"],
                             "steps": K, "kernel": "gs_k_step_" + KERNEL_NAMES.get(mm["desc"]["kernel"], mm["desc"]["kernel"]),
                             "lu_slots": mm["desc"].get("lu_slots"), "lu_original_blocks": mm["desc"].get("lu_orig"), "lu_pair_updates": pairs,
                             "mean_iterations": mm["mean_iterations"], "converged_fraction": mm["converged_fraction"], "roofline": rl}
                    if not args.no_cpu_baseline:
                        entry["cpu_baseline"] = cpu_baseline(fsm, env_kwargs_of("nr"), budget_s=5.0)
                    result["also_meshed"][key] = entry
                except Exception as e:
                    result["also_meshed"][key] = {"error": str(e)}
            # BASELINE config 4's data path rehearsed on this one GPU: W ranks' shards in this process, the exchange through
            # the in-process transport (what an 8-GPU run adds to this is ncclAllGather itself)
            try:
                result["also_config4_loopback"] = {}
                for W in (2, 8):
                    m4 = measure_loopback(fs, B, solver0, W, 3, 20, min(args.steps, 20))
                    result["also_config4_loopback"][f"world_{W}"] = loopback_summary(fs, B, W, m4, min(args.steps, 20))
            except Exception as e:
                result["also_config4_loopback"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["accuracy"] = accuracy(fs, solver0)
            # the reference's CPU path is dense Newton-Raphson: that port is THE baseline; the CPU port of
            # the solver the GPU ran is reported next to it when it differs; and the same on ONE core
            result["cpu_baseline"] = cpu_baseline(fs, env_kwargs_of("nr"), budget_s=10.0)
            result["cpu_baseline_1core"] = cpu_baseline(fs, env_kwargs_of("nr"), budget_s=6.0, threads=1)
            # every CPU in this process's affinity mask (the 16-thread figure above is the share of the host one GPU gets)
            result["cpu_baseline_all_cores"] = cpu_baseline(fs, env_kwargs_of("nr"), budget_s=6.0, threads="all")
            # the baseline the headline is set beside is the FASTER of the two (cores stated in it); the other stays in the line
            a, b16 = result["cpu_baseline_all_cores"], result["cpu_baseline"]
            if a and b16 and a.get("value", 0.0) > b16.get("value", 0.0):
                result["cpu_baseline"], result["cpu_baseline_gpu_share_of_host"] = a, b16
            if solver0 != "nr":
                result["cpu_baseline_same_solver"] = cpu_baseline(fs, env_kwargs_of(solver0), budget_s=6.0)
                result["cpu_baseline_same_solver_1core"] = cpu_baseline(fs, env_kwargs_of(solver0), budget_s=4.0, threads=1)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)

    if rz is not None:
        rz.close()


if __name__ == "__main__":
    main()
