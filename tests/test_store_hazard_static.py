"""Static guard for the 16-byte store-data hazard: no gfx950 code object of libgridstep.so may hold
a 12/16-byte buffer / global store whose very next instruction is a VALU write into the store's data registers.
CPU-only: hipcc cross-compiles and llvm-objdump disassembles without a GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_store_hazard as H  # noqa: E402

EXPOSED = """
0000000000001000 <gs_k_demo>:
	buffer_store_dwordx4 v[0:3], v77, s[28:31], s27 offen      // 0000000B53DC: E07C1000 1B07004D
	v_mov_b64_e32 v[0:1], s[22:23]                             // 0000000B53E4: 7E007016
	s_nop 3                                                    // 0000000B53E8: BF800003
	global_store_dwordx4 v38, v[18:21], s[14:15]               // 000000006DF4: DC7C8000 000E1226
	v_fmac_f64_e32 v[20:21], v[22:23], v[20:21]                // 000000006E04: 083C2916
"""
CLEAN = """
0000000000001000 <gs_k_demo>:
	buffer_store_dwordx4 v[0:3], v77, s[28:31], s27 offen      // 0000000B53DC: E07C1000 1B07004D
	s_nop 0                                                    // 0000000B53E8: BF800000
	v_mov_b64_e32 v[0:1], s[22:23]                             // 0000000B53E4: 7E007016
	buffer_store_dwordx4 v[4:7], v77, s[28:31], s27 offen      // 0000000B53DC: E07C1000 1B07004D
	v_mov_b64_e32 v[8:9], s[22:23]                             // 0000000B53E4: 7E007016
	global_store_dwordx4 v38, v[18:21], s[14:15]               // 000000006DF4: DC7C8000 000E1226
	ds_write_b64 v68, v[18:19]                                 // 000000006DFC: D89A0000 00001244
	buffer_store_dwordx4 v[0:3], v77, s[28:31], s27 offen      // 0000000B53DC: E07C1000 1B07004D
	v_cmp_le_i32_e32 vcc, s13, v1                              // 000000007BB0: 7D86460D
"""


def test_checker_flags_a_valu_write_right_behind_the_store():
    n, bad = H.scan_disassembly(EXPOSED, "demo")
    assert n == 2 and len(bad) == 2
    assert "v_mov_b64" in bad[0][2] and "v_fmac_f64" in bad[1][2]


def test_checker_accepts_a_wait_state_other_registers_and_non_valu_followers():
    n, bad = H.scan_disassembly(CLEAN, "demo")
    assert n == 4 and bad == []


def test_no_exposed_16_byte_store_in_libgridstep():
    so = os.path.join(ROOT, "grid_fed_rl_gym_amd", "libgridstep.so")
    if not os.path.exists(so):
        pytest.skip("libgridstep.so not built")
    if not (os.path.exists(os.path.join(H.LLVM_BIN, "llvm-objdump")) or __import__("shutil").which("llvm-objdump")):
        pytest.skip("no llvm-objdump")
    n, bad = H.scan_library(so)
    assert n > 100, n                      # the row-pair stores are there at all (GsPairRef::put, the pack, gridstep3)
    assert bad == [], "\n".join(f"{k}: {s} -> {nx}" for k, s, nx in bad)
    assert not [f for f in os.listdir(os.path.dirname(so)) if f.startswith("libgridstep.so.")]   # nothing extracted next to the library
