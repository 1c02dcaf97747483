"""GPU: the C++ host mirror of Lol's Cyc / SymmSHE (alchemy_amd/host/symmshe.hpp) replaying
examples/Arithmetic.hs through the C ABI -- the reference's own end-to-end check (decrypt the homomorphic
result, compare with the plaintext evaluation, print PASS; examples/Arithmetic.hs:73-75)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def replay_binary():
    exe = os.path.join(ROOT, "examples", "arithmetic_replay")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "examples", "arithmetic_replay.cpp"),
                    "-L" + os.path.join(ROOT, "alchemy_amd", "lib"), "-lalchemy_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "alchemy_amd", "lib")], check=True)
    return exe


@pytest.mark.parametrize("index", ["512", "32"])     # the file's F512, and BASELINE config 1's "m=32"
def test_arithmetic_example_prints_pass(replay_binary, index):
    out = subprocess.run([replay_binary, index], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fused path == per-op path: yes" in out.stdout
    assert "fused full mul_ (2 -> 3 -> 1 limbs) == per-op path: yes" in out.stdout
    assert out.stdout.strip().endswith("PASS")
