"""The native multi-GPU route (include/alchemy_rccl.h, alchemy_amd/lib/libalchemy_rccl.so): RCCL broadcast of
the hint sources and all-gather of result ranges on the library's own device buffers, for hosts that are not Python.

CPU: the library loads, exports every symbol its header declares, and rejects bad arguments before touching a device.
GPU: the collectives with ONE rank through real RCCL calls -- from ctypes, and from the compiled C++ driver
examples/ringround_multi.cpp (no torch in that process), whose shard is checked against the oracle's per-ciphertext checksums.
More than one peer needs a multi-GPU node: unmeasured on hardware."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from alchemy_amd import capi
from conftest import ROOT

LIB = os.path.join(ROOT, "alchemy_amd", "lib", "libalchemy_rccl.so")
SYMS = ["alch_rccl_last_error", "alch_comm_init_all", "alch_comm_destroy", "alch_comm_size", "alch_hint_broadcast", "alch_buf_all_gather",
        "alch_comm_unique_id", "alch_comm_init_rank", "alch_comm_local"]
ID_BYTES = 128                                               # ALCH_COMM_ID_BYTES


@pytest.fixture(scope="module")
def rccl():
    capi.load_library()                                      # libalchemy_hip.so first: the RCCL route links against it
    l = C.CDLL(LIB)
    l.alch_rccl_last_error.restype = C.c_char_p
    l.alch_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    l.alch_comm_destroy.argtypes = [C.c_void_p]
    l.alch_comm_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.alch_comm_unique_id.argtypes = [C.c_char_p]
    l.alch_comm_init_rank.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    l.alch_comm_local.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    l.alch_hint_broadcast.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]
    l.alch_buf_all_gather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    return l


def test_library_exports_what_its_header_declares(rccl):
    header = open(os.path.join(ROOT, "include", "alchemy_rccl.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = re.findall(r"^\s*(?:const\s+)?\w+\s*\*?\s*(alch_\w+)\s*\(", header, flags=re.M)
    assert sorted(declared) == sorted(SYMS)
    for s in SYMS:
        assert hasattr(rccl, s), s


def test_arguments_are_checked_before_any_device_is_touched(rccl):
    h = C.c_void_p()
    assert rccl.alch_comm_init_all(0, C.byref(h)) == capi.ALCH_E_INVALID
    assert rccl.alch_comm_init_all(-3, C.byref(h)) == capi.ALCH_E_INVALID
    assert rccl.alch_comm_init_all(1, None) == capi.ALCH_E_INVALID
    n = C.c_int()
    assert rccl.alch_comm_size(None, C.byref(n)) == capi.ALCH_E_INVALID
    assert rccl.alch_hint_broadcast(None, 0, None, 0, 1) == capi.ALCH_E_INVALID
    assert rccl.alch_buf_all_gather(None, None, 0, 1, None) == capi.ALCH_E_INVALID
    # one process per GPU: rank / id are checked before RCCL or a device is touched
    zero_id = bytes(ID_BYTES)
    assert rccl.alch_comm_unique_id(None) == capi.ALCH_E_INVALID
    assert rccl.alch_comm_init_rank(2, 0, zero_id, None) == capi.ALCH_E_INVALID
    assert rccl.alch_comm_init_rank(2, 0, None, C.byref(h)) == capi.ALCH_E_INVALID
    for n_ranks, rank in ((0, 0), (2, 2), (2, -1), (-1, 0)):
        assert rccl.alch_comm_init_rank(n_ranks, rank, zero_id, C.byref(h)) == capi.ALCH_E_INVALID
    a, b = C.c_int(), C.c_int()
    assert rccl.alch_comm_local(None, C.byref(a), C.byref(b)) == capi.ALCH_E_INVALID
    assert f"#define ALCH_COMM_ID_BYTES {ID_BYTES}" in open(os.path.join(ROOT, "include", "alchemy_rccl.h")).read()
    assert rccl.alch_comm_destroy(None) == capi.ALCH_OK
    rccl.alch_comm_size(None, C.byref(n))
    assert b"null" in rccl.alch_rccl_last_error() or b"n_dev" in rccl.alch_rccl_last_error()


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device error path")
def test_without_a_device_the_route_fails_loudly(rccl):
    h = C.c_void_p()
    assert rccl.alch_comm_init_all(1, C.byref(h)) == capi.ALCH_E_NO_DEVICE
    assert rccl.alch_comm_init_rank(1, 0, bytes(ID_BYTES), C.byref(h)) == capi.ALCH_E_NO_DEVICE
    one = subprocess.run([os.path.join(ROOT, "examples", "ringround_multi"), "--world", "1", "--rank", "0", "--id-file", "/tmp/alch_id_none"],
                         capture_output=True, text=True, cwd=ROOT) if os.path.exists(os.path.join(ROOT, "examples", "ringround_multi")) else None
    assert one is None or (one.returncode == 2 and "no HIP device" in one.stderr)
    subprocess.run([os.path.join(ROOT, "tools", "build_examples.sh")], check=True, capture_output=True)
    out = subprocess.run([os.path.join(ROOT, "examples", "ringround_multi"), "--gpus", "1", "--batch", "4"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 2 and "no HIP device" in out.stderr


@pytest.mark.gpu
def test_one_rank_collectives_through_rccl(rccl):
    import alchemy_amd as A
    ring = A.Ring(11648, [1543651201, 689270401, 718099201])
    rng = np.random.default_rng(1)
    xs = np.stack([np.stack([rng.integers(0, q, size=ring.n, dtype=np.int64) for q in ring.qs], axis=1) for _ in range(6)])
    src, dst = ring.upload(xs), ring.alloc(4)
    comm = C.c_void_p()
    assert rccl.alch_comm_init_all(1, C.byref(comm)) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    n = C.c_int()
    assert rccl.alch_comm_size(comm, C.byref(n)) == capi.ALCH_OK and n.value == 1
    bufs = (C.c_void_p * 1)(src._h)
    assert rccl.alch_hint_broadcast(comm, 0, bufs, 1, 3) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    assert np.array_equal(src.download(), xs)                # a one-rank broadcast leaves the root's data in place
    dsts = (C.c_void_p * 1)(dst._h)
    assert rccl.alch_buf_all_gather(comm, bufs, 2, 4, dsts) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    assert np.array_equal(dst.download(), xs[2:6])           # ordered on the ring's stream: the download waits for the collective
    # argument errors with a live communicator
    assert rccl.alch_hint_broadcast(comm, 1, bufs, 0, 1) == capi.ALCH_E_INVALID              # root out of range
    assert rccl.alch_hint_broadcast(comm, 0, bufs, 5, 2) == capi.ALCH_E_INVALID              # range out of bounds
    assert rccl.alch_buf_all_gather(comm, bufs, 0, 5, dsts) == capi.ALCH_E_INVALID           # dst too small
    big = C.c_size_t(2**64 - 1)
    assert rccl.alch_hint_broadcast(comm, 0, bufs, big, 2) == capi.ALCH_E_INVALID            # first + count wraps: still out of bounds
    assert rccl.alch_buf_all_gather(comm, bufs, big, 2, dsts) == capi.ALCH_E_INVALID
    other = A.Ring(11648, [1543651201, 689270401])
    ob = other.alloc(4)
    assert rccl.alch_buf_all_gather(comm, bufs, 0, 1, (C.c_void_p * 1)(ob._h)) == capi.ALCH_E_INVALID   # different rings
    h2 = C.c_void_p()
    assert rccl.alch_comm_init_all(2, C.byref(h2)) == capi.ALCH_E_NO_DEVICE                  # one rank per GPU: the box has one
    assert rccl.alch_comm_destroy(comm) == capi.ALCH_OK


@pytest.mark.gpu
def test_one_process_per_gpu_communicator_with_one_rank(rccl, tmp_path):
    """alch_comm_unique_id + alch_comm_init_rank (the launch model of torchrun, natively): a communicator of ONE process-rank through
    real RCCL calls, the collectives with single-entry buffer arrays, and the compiled driver in --world 1 mode (id through a file).
    More than one process needs more than one GPU (RCCL refuses two ranks on one device): unmeasured on hardware."""
    import alchemy_amd as A
    ring = A.Ring(11648, [1543651201, 689270401, 718099201])
    rng = np.random.default_rng(2)
    xs = np.stack([np.stack([rng.integers(0, q, size=ring.n, dtype=np.int64) for q in ring.qs], axis=1) for _ in range(5)])
    src, dst = ring.upload(xs), ring.alloc(3)
    ident = C.create_string_buffer(ID_BYTES)
    assert rccl.alch_comm_unique_id(ident) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    assert any(ident.raw)
    comm = C.c_void_p()
    assert rccl.alch_comm_init_rank(1, 0, ident.raw, C.byref(comm)) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    n, nl, first = C.c_int(), C.c_int(), C.c_int()
    assert rccl.alch_comm_size(comm, C.byref(n)) == capi.ALCH_OK and n.value == 1
    assert rccl.alch_comm_local(comm, C.byref(nl), C.byref(first)) == capi.ALCH_OK and (nl.value, first.value) == (1, 0)
    bufs, dsts = (C.c_void_p * 1)(src._h), (C.c_void_p * 1)(dst._h)
    assert rccl.alch_hint_broadcast(comm, 0, bufs, 0, 5) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    assert np.array_equal(src.download(), xs)
    assert rccl.alch_buf_all_gather(comm, bufs, 1, 3, dsts) == capi.ALCH_OK, rccl.alch_rccl_last_error()
    assert np.array_equal(dst.download(), xs[1:4])
    assert rccl.alch_hint_broadcast(comm, 1, bufs, 0, 1) == capi.ALCH_E_INVALID              # root is a rank of the communicator
    assert rccl.alch_buf_all_gather(comm, bufs, 0, 4, dsts) == capi.ALCH_E_INVALID           # dst holds ranks * count
    assert rccl.alch_comm_destroy(comm) == capi.ALCH_OK
    subprocess.run([os.path.join(ROOT, "tools", "build_examples.sh")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "examples", "ringround_multi")
    out = subprocess.run([exe, "--world", "1", "--rank", "0", "--id-file", str(tmp_path / "rccl_id"), "--batch", "64", "--gather", "4", "--passes", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["ranks_in_this_process"] == 1 and d["first_rank"] == 0
    assert d["shard_checksums_ok"] == [True] and d["all_gather_slices_ok"] == [True] and d["ciphertexts_checked_per_shard"] == 64
    assert os.path.getsize(tmp_path / "rccl_id") == ID_BYTES
    # the launcher a native host would use for N processes (here N = 1: the box has one GPU)
    out = subprocess.run([os.path.join(ROOT, "tools", "launch_ringround_ranks.sh"), "1", "--batch", "32", "--gather", "2", "--passes", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["shard_checksums_ok"] == [True] and d["all_gather_slices_ok"] == [True]


@pytest.mark.gpu
def test_native_driver_shards_config_4_and_checks_its_shard_against_the_oracle():
    """examples/ringround_multi --gpus 1: C++ threads + RCCL, no torch in the process -- hint sources broadcast, the ringRound
    pipeline on the shard, every result ciphertext checked against tests/golden/batch_checksums.json, results all-gathered."""
    subprocess.run([os.path.join(ROOT, "tools", "build_examples.sh")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "examples", "ringround_multi")
    out = subprocess.run([exe, "--gpus", "1", "--batch", "96", "--gather", "8", "--passes", "2"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["shard_checksums_ok"] == [True] and d["all_gather_slices_ok"] == [True]
    assert d["ciphertexts_checked_per_shard"] == 96 and d["pipelines_per_s"] > 0
    two = subprocess.run([exe, "--gpus", "2", "--batch", "8"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert two.returncode == 2 and "only 1 devices visible" in two.stderr
    assert "torch" not in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
