"""Static guard for what DESIGN.md section 3 ("what binds the step kernels") found: the step kernels are bound by vector
instruction issue, and kernel arguments taken from the formal parameters are parked in vector lanes (v_writelane / v_readlane
at every use).  The kernels read their arguments in place; this test fails if a change brings the scalar-register spills back
(or spills vector registers inside the headline kernel).  CPU-only: reads the code objects' metadata with llvm-readobj."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as K  # noqa: E402

# kernel: (scalar spills at most, vector spills at most); measured on the round-3 build: 21/2, 23/10, 48/0, 20/0, 35/0, 37/0
LIMITS = {
    "gs_k_step_fbs_flow2h": (60, 8),
    "gs_k_stepc_fbs_flow2h": (60, 16),
    "gs_k_step_nr_flow2": (100, 0),
    "gs_k_step_nr_flow2s": (60, 0),
    "gs_k_step_nr_lu": (80, 0),
    "gs_k_nr_dense_mfma2": (80, 8),        # block-row form (round 4): two workgroups per CU, 256 registers
    "gs_k_step_nr_mesh2": (80, 40),        # meshed member (round 4): T / s of ten rows in registers, 27 spilled outside the row loops
}


def test_step_kernels_keep_their_arguments_out_of_the_vector_lanes():
    so = os.path.join(ROOT, "grid_fed_rl_gym_amd", "libgridstep.so")
    if not os.path.exists(so):
        pytest.skip("libgridstep.so not built")
    if not os.path.exists(os.path.join(K.LLVM_BIN, "llvm-readobj")):
        pytest.skip("no llvm-readobj")
    res = K.resources(so)
    for name, (smax, vmax) in LIMITS.items():
        assert name in res, name
        r = res[name]
        assert 0 <= r["sspill"] <= smax, (name, r)
        assert 0 <= r["vspill"] <= vmax, (name, r)
    # the headline member is built for two workgroups of 8 waves on a CU: 128 registers per lane
    assert res["gs_k_step_fbs_flow2h"]["vgpr"] <= 128
    # the dense block-row form and the meshed member are built for two workgroups of 4 waves on a CU
    assert res["gs_k_nr_dense_mfma2"]["vgpr"] <= 256 and res["gs_k_step_nr_mesh2"]["vgpr"] <= 256
