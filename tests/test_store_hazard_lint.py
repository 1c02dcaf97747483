"""CPU: build-time lint for the 16-byte buffer-store data hazard (DESIGN.md "A hazard worth recording").

The one real corruption this project has seen -- k_rescale_out_lin losing an element in 0.5 % of its polynomials -- came from a
`buffer_store_dwordx4` with an SGPR offset whose first data register was overwritten by the VALU instruction right behind it; it was
invisible below a few hundred ciphertexts.  ALCH_STORE_GUARD is applied by hand behind every such store; tools/lint_store_hazard.py
disassembles the gfx950 code objects of the build and fails when one is missing."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import lint_store_hazard as L  # noqa: E402


def test_scanner_flags_a_valu_write_to_the_store_data_right_behind_the_store():
    bad = """
	buffer_store_dwordx4 v[40:43], v1, s[8:11], s6 offen       // 000000020590: E07C1000 06022801
	v_lshlrev_b32_e32 v40, 2, v38                              // 00000002059C: 24504C82
""".splitlines()
    assert len(L.scan(bad)) == 1
    wide = """
	buffer_store_dwordx4 v[40:43], v1, s[8:11], s6 offen nt
	v_mad_u64_u32 v[42:43], s[0:1], v3, v4, 0
""".splitlines()
    assert len(L.scan(wide)) == 1                                  # a 64-bit destination overlapping the last two data registers
    x3 = ["	buffer_store_dwordx3 v[4:6], v1, s[8:11], s33 offen", "	v_mov_b32_e32 v6, v9"]
    assert len(L.scan(x3)) == 1


def test_scanner_accepts_guarded_and_harmless_sequences():
    ok = """
	buffer_store_dwordx4 v[40:43], v1, s[8:11], s6 offen
	s_nop 0
	v_lshlrev_b32_e32 v40, 2, v38
	buffer_store_dwordx4 v[44:47], v1, s[8:11], s54 offen
	v_lshlrev_b32_e32 v48, 2, v38
	buffer_store_dwordx4 v[44:47], v1, s[8:11], 0 offen
	v_mov_b32_e32 v44, v9
	buffer_store_dwordx4 v[44:47], off, s[8:11], 0
	v_mov_b32_e32 v44, v9
	buffer_store_dwordx2 v[44:45], v1, s[8:11], s6 offen
	v_mov_b32_e32 v44, v9
	buffer_store_dwordx4 v[44:47], v1, s[8:11], s6 offen
	v_cmp_lt_u32_e32 vcc, s37, v47
	buffer_store_dwordx4 v[44:47], v1, s[8:11], s6 offen
	ds_read_b128 v[44:47], v40 offset:8192
""".splitlines()
    assert L.scan(ok) == []                                        # guard, disjoint registers, constant offsets, 8-byte stores, non-VALU writers


@pytest.fixture(scope="module")
def built_objects():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    objs = sorted(glob.glob(os.path.join(ROOT, "alchemy_amd", "csrc", "build", "*.o")))
    assert len(objs) >= 10
    return objs


def test_every_16_byte_buffer_store_of_the_build_is_guarded(built_objects):
    bad, stores = L.lint_objects(built_objects)
    assert stores > 500, "the disassembly pipeline found no stores: the lint is not looking at device code"
    assert bad == [], "unguarded 16-byte buffer stores (add ALCH_STORE_GUARD behind them):\n" + "\n".join(f"{n}:{no}: {s} <- {x}" for n, no, s, x in bad)


def test_the_lint_sees_the_hazard_in_real_code_when_the_guard_is_compiled_out(tmp_path, built_objects):
    """Positive control on the real sources: the n = 2^15 kernels compiled with -DALCH_NO_STORE_GUARD (a macro that exists for this
    test only) contain the sequence, so a store written without the guard would be caught."""
    obj = str(tmp_path / "noguard.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DALCH_NO_STORE_GUARD", "-c",
                    os.path.join(ROOT, "alchemy_amd", "csrc", "inst_32_15.hip"), "-o", obj], check=True, capture_output=True)
    bad, stores = L.lint_objects([obj])
    assert stores > 100 and len(bad) >= 1
    makefile = open(os.path.join(ROOT, "alchemy_amd", "csrc", "Makefile")).read()
    assert "ALCH_NO_STORE_GUARD" not in makefile
