#!/usr/bin/env python3
"""Whole-batch result checksums at the BENCH shape (n = 2^15, 4 limbs), produced offline by the C restatement:
    python tests/golden/make_batch_checksums.py            (about two minutes on 8 cores)
-> tests/golden/batch_checksums.json.  alch_buf_checksum of the device result must equal these values:
   * bench_mul_relin: bench.py's default step (B = 8192, seeds 2026 / 900000007 / 0xA1C4E5) -- bench.py asserts it;
   * test_mul_relin / test_mul_full: B = 2 * 1024 + 37 (two full chunks on two streams plus a ragged tail at the default
     launch options) -- tests/test_gpu_bench_shape.py asserts them.
A result error confined to any chunk, stream or persistent-workgroup slot changes the sum."""
import json
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import oracle_full_mul          # noqa: E402
from oracle import cref                      # noqa: E402

CFG3_QS = [2147352577, 2146959361, 2146041857, 2145976321]
FULL_EXTRA_Q = 2144796673
N = 1 << 15
SEED_A, SEED_B, SEED_H = 2026, 900_000_007, 0xA1C4E5
THREADS = int(os.environ.get("ALCH_CPU_THREADS", str(os.cpu_count() or 1)))
MASK = (1 << 64) - 1


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def elem_checksum(aos, elem_index, L):
    """sum over the words of one (n, L) element at position elem_index of a buffer of splitmix64(w ^ value << 20)."""
    with np.errstate(over="ignore"):
        lm = np.ascontiguousarray(aos.T).astype(np.uint64).reshape(-1)            # limb-major words
        w = np.uint64(elem_index * L * N) + np.arange(L * N, dtype=np.uint64)
        return int(splitmix64(w ^ (lm << np.uint64(20))).sum(dtype=np.uint64))


def run_ranges(fn, total):
    """fn(first, count) -> partial checksum; split [0, total) over THREADS threads."""
    parts, out = np.array_split(np.arange(total), THREADS), [0] * THREADS

    def work(i):
        if len(parts[i]):
            out[i] = fn(int(parts[i][0]), len(parts[i]))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(THREADS)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return sum(out) & MASK


def relin_range(first, count):
    return cref.Ring(N, CFG3_QS).mul_relin_checksum(SEED_A, SEED_B, SEED_H, first, count)


def full_range(first, count):
    qs_h = [FULL_EXTRA_Q] + CFG3_QS
    o_in, o_h = cref.Ring(N, CFG3_QS), cref.Ring(N, qs_h)
    hint = [o_h.fill_uniform(SEED_H, i) for i in range(2 * len(qs_h))]
    s = 0
    for ct in range(first, first + count):
        a0, a1 = o_in.fill_uniform(SEED_A, 2 * ct), o_in.fill_uniform(SEED_A, 2 * ct + 1)
        b0, b1 = o_in.fill_uniform(SEED_B, 2 * ct), o_in.fill_uniform(SEED_B, 2 * ct + 1)
        w0, w1 = oracle_full_mul(cref, N, qs_h, 4, 3, hint, a0, a1, b0, b1)
        s += elem_checksum(w0, 2 * ct, 3) + elem_checksum(w1, 2 * ct + 1, 3)
    return s & MASK


if __name__ == "__main__":
    cref.build()
    B_TEST, B_BENCH = 2 * 1024 + 37, 8192
    head = run_ranges(relin_range, B_TEST)
    tail = run_ranges(lambda f, c: relin_range(B_TEST + f, c), B_BENCH - B_TEST)
    full = run_ranges(full_range, B_TEST)
    out = {"n": N, "moduli": CFG3_QS, "full_extra_modulus": FULL_EXTRA_Q, "seeds": {"a": SEED_A, "b": SEED_B, "hint": SEED_H},
           "rule": "sum over result words of splitmix64(w ^ value << 20), w = limb-major word position (alch_buf_checksum)",
           "test_mul_relin": {"batch": B_TEST, "checksum": f"{head:016x}"},
           "bench_mul_relin": {"batch": B_BENCH, "checksum": f"{(head + tail) & MASK:016x}"},
           "test_mul_full": {"batch": B_TEST, "limbs": "4 -> 5 -> 3", "checksum": f"{full:016x}"}}
    with open(os.path.join(HERE, "batch_checksums.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
