"""CPU: no C++ exception crosses the C ABI (include/alchemy_hip.h: "no exceptions across the ABI").

The callers of this library are C, Haskell's FFI and ctypes: an exception that escapes an `extern "C"` function is undefined
behaviour there (std::terminate at best).  Every `extern "C" int` entry point of libalchemy_hip.so and libalchemy_rccl.so is
therefore a function-try-block ending in abi_catch().  Checked two ways:
  * through the real ABI: the test hook alch_debug_throw raises std::bad_alloc / std::length_error / a non-std exception inside
    such a block -- the call returns ALCH_E_NOMEM / ALCH_E_INTERNAL with a message, the process lives;
  * lexically: no entry-point definition in the sources lacks the block (a new entry point written without it fails here)."""
import ctypes
import glob
import os
import re

import alchemy_amd
from alchemy_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alchemy_amd", "csrc")


def test_exceptions_are_caught_at_the_abi():
    lib = alchemy_amd.load_library()
    lib.alch_debug_throw.argtypes = [ctypes.c_int]
    lib.alch_last_error.restype = ctypes.c_char_p
    assert lib.alch_debug_throw(0) == capi.ALCH_OK
    assert lib.alch_debug_throw(1) == capi.ALCH_E_NOMEM and b"bad_alloc" in lib.alch_last_error()
    assert lib.alch_debug_throw(2) == capi.ALCH_E_INTERNAL and b"alch_debug_throw" in lib.alch_last_error()
    assert lib.alch_debug_throw(3) == capi.ALCH_E_INTERNAL and b"unknown exception" in lib.alch_last_error()
    assert lib.alch_version() >> 16 == 1                      # still alive and answering


def test_every_entry_point_is_a_function_try_block():
    missing, seen = [], 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.cpp"))):
        lines = open(path).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r'^extern "C" (?:__attribute__\(\(.*?\)\) )?int (alch_\w+)\(', ln)
            if not m or m.group(1).startswith("alch_debug_stamps"):   # diagnostic builds only (-DALCH_STAMPS), never in the product
                continue
            seen += 1
            j = i
            while not (lines[j].rstrip().endswith("{") or lines[j].rstrip().endswith("}")):
                j += 1                                            # the header of the definition may span several lines
            if " try {" not in lines[j]:
                missing.append(f"{os.path.basename(path)}:{i + 1} {m.group(1)}")
    assert seen >= 85, seen                                       # 81 + 6 entry points and the hook: the scan found them
    assert not missing, "entry points without the ABI guard: " + ", ".join(missing)
    # every guarded block ends in the handler
    for path in (os.path.join(CSRC, "alchemy_hip.hip"), os.path.join(CSRC, "tensor_ext.inc.hpp"), os.path.join(CSRC, "alchemy_rccl.cpp")):
        src = open(path).read()
        assert src.count(") try {") == src.count("catch (...) { return abi_catch(); }") > 0, path


def test_header_documents_the_internal_status():
    hdr = open(os.path.join(ROOT, "include", "alchemy_hip.h")).read()
    assert "#define ALCH_E_INTERNAL (-8)" in hdr and capi.ALCH_E_INTERNAL == -8
