"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the host-only code of the PRODUCT library.

GPU sanitizers are refused on this pool, and until round 4 only the C oracle ran under ASan.  The host side of
alchemy_amd/csrc/alchemy_hip.hip -- index plans and extension tables (gen_host.hpp), CRT sets over GF(p^d) (crtset_host.hpp), the
root rule (ring_host.hpp), alch_select_limbs, the argument classification of alch_ring_create* -- compiles without device code
(`hipcc --offload-host-only`), so it is built with -fsanitize=address,undefined and driven by tests/sanitize/host_harness.cpp over
the index pairs of tests/test_tensor_ext.py, the reference's rings and modulus lists.  The harness supplies "no device" definitions
of the HIP runtime: no hot-path arithmetic exists in that program."""
import os
import subprocess

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_host_only_code_is_clean_under_asan_and_ubsan(tmp_path):
    lib_o, har_o, exe = str(tmp_path / "alchemy_host.o"), str(tmp_path / "harness.o"), str(tmp_path / "harness")
    common = [HIPCC, "--offload-arch=gfx950", "--offload-host-only", "-std=c++17", "-w"] + SAN
    subprocess.run(common + ["-c", os.path.join(ROOT, "alchemy_amd", "csrc", "alchemy_hip.hip"), "-o", lib_o], check=True)
    subprocess.run(common + ["-c", os.path.join(ROOT, "tests", "sanitize", "host_harness.cpp"), "-o", har_o], check=True)
    # the host object refers to its (absent) device bundle by a per-compilation symbol
    nm = subprocess.run(["nm", lib_o], check=True, capture_output=True, text=True).stdout
    fat = [l.split()[-1] for l in nm.splitlines() if " U __hip_fatbin_" in l]
    assert len(fat) == 1
    subprocess.run([CLANGXX, "-fsanitize=address,undefined", lib_o, har_o, f"-Wl,--defsym={fat[0]}=0", "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.strip().endswith("OK: 0 failed expectation(s)")
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
