#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in the gfx950 code objects bundled in a shared library (the .note metadata).

    python tools/kernel_resources.py [grid_fed_rl_gym_amd/libgridstep.so] [name filter]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(lib):
    """{kernel name: dict(vgpr, agpr, sgpr, vspill, sspill, scratch, lds)} for every kernel of the library's gfx950 code objects."""
    tmp = tempfile.mkdtemp(prefix="gs_res_")
    out = {}
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readobj"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                num = lambda v: int(v) if v.isdigit() else -1
                out[g("name")] = dict(vgpr=num(g("vgpr_count")), agpr=num(blk.split()[0]), sgpr=num(g("sgpr_count")), vspill=num(g("vgpr_spill_count")),
                                      sspill=num(g("sgpr_spill_count")), scratch=num(g("private_segment_fixed_size")), lds=num(g("group_segment_fixed_size")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "grid_fed_rl_gym_amd", "libgridstep.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    print("%-44s %5s %5s %5s %7s %7s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for name, r in sorted(resources(lib).items()):
        if flt in name:
            print("%-44s %5d %5d %5d %7d %7d %8d %8d" % (name, r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"], r["scratch"], r["lds"]))


if __name__ == "__main__":
    main()
