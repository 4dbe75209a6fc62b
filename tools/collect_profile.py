#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile.sh into the tracked summaries under profiles/.

    python tools/collect_profile.py TAG OUT_PREFIX [--traffic-key ieee123_b8192:fbs --kernel gs_k_step_fbs_flow2]

reads   gpurun_out/TAG_trace/**/_kernel_stats.csv            (rocprofv3 --kernel-trace --stats)
        gpurun_out/TAG_pmc_FETCH_SIZE/**/_counter_collection.csv, gpurun_out/TAG_pmc_WRITE_SIZE/**  (separate --pmc passes)
        gpurun_out/TAG_bench.log                             (the un-profiled bench line of the same build)
writes  profiles/OUT_PREFIX_kernel_stats.csv  (rows of the gs_* kernels), profiles/OUT_PREFIX_pmc.csv (kernel, counter, value_kb
        per dispatch), profiles/OUT_PREFIX_bench.json, and -- with --traffic-key -- the entry of profiles/hbm_traffic.json that
        bench.py reads for roofline.traffic: fabric-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE
        doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section; calibrated in round 1 on gs_k_pack).
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(tag_dir, suffix):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag_dir, "**", "*" + suffix), recursive=True))
    return hits[-1] if hits else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag"); ap.add_argument("out_prefix")
    ap.add_argument("--traffic-key", default=""); ap.add_argument("--kernel", default="")
    ap.add_argument("--note", default="")
    ap.add_argument("--dispatches-per-step", type=int, default=1,
                    help="a step that goes out as N dispatches of the kernel (two half-grid launches on two streams): the traffic entry is per STEP")
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    stats = find(a.tag + "_trace", "_kernel_stats.csv")
    if stats:
        rows = list(csv.reader(open(stats)))
        with open(os.path.join(prof, a.out_prefix + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(rows[0])
            for r in rows[1:]:
                if r and (r[0].startswith("gs_") or r[0].startswith("gs3_") or "gs3_k_" in r[0]):
                    w.writerow(r)
        print("kernel stats:", [(r[0], float(r[3]) / 1e3) for r in rows[1:] if r and r[0].startswith("gs_k_step")])
    per_kernel = {}
    out_rows = [("kernel", "counter", "value_kb")]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = find(f"{a.tag}_pmc_{c}", "_counter_collection.csv")
        if not path:
            continue
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            for t in ("gs3_k_solve", "gs3_k_resident"):
                if t in k:
                    k = t                              # (a mangled template name)
            if not k.startswith("gs"):
                continue
            out_rows.append((k, r["Counter_Name"], "%.6f" % float(r["Counter_Value"])))
            per_kernel.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    if len(out_rows) > 1:
        with open(os.path.join(prof, a.out_prefix + "_pmc.csv"), "w", newline="") as f:
            csv.writer(f).writerows(out_rows)
    bench = os.path.join(ROOT, "gpurun_out", a.tag + "_bench.log")
    if os.path.exists(bench):
        for line in open(bench):
            if line.startswith("{"):
                open(os.path.join(prof, a.out_prefix + "_bench.json"), "w").write(line)
    if a.traffic_key and a.kernel:
        fe, wr = per_kernel.get((a.kernel, "FETCH_SIZE")), per_kernel.get((a.kernel, "WRITE_SIZE"))
        if not fe or not wr:
            print("no counters for", a.kernel, file=sys.stderr); sys.exit(1)
        fe_kb, wr_kb = a.dispatches_per_step * sum(fe) / len(fe), a.dispatches_per_step * sum(wr) / len(wr)
        tfile = os.path.join(prof, "hbm_traffic.json")
        d = json.load(open(tfile)) if os.path.exists(tfile) else {}
        sys.path.insert(0, ROOT)
        import bench as _bench
        d[a.traffic_key] = {"kernel": a.kernel, "profile": a.out_prefix, "csrc_sha": _bench.csrc_hash(a.traffic_key), "fetch_size_kb_raw": fe_kb, "write_size_kb": wr_kb,
                            "solve_bytes_per_launch": int((2 * fe_kb + wr_kb) * 1024),
                            "note": "fabric-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled per the gfx950 correction "
                                    "(MI355X_MICROARCH.md, HBM) | " + (a.note or a.out_prefix + " counters") +
                                    (f" | per step = {a.dispatches_per_step} dispatches of the kernel" if a.dispatches_per_step > 1 else "")}
        json.dump(d, open(tfile, "w"), indent=1)
        print(a.traffic_key, d[a.traffic_key]["solve_bytes_per_launch"] / 1e6, "MB per launch (fetch x2 %.1f MB, write %.1f MB)" % (2 * fe_kb / 1024 * 1.048576, wr_kb / 1024 * 1.048576))


if __name__ == "__main__":
    main()
