#!/usr/bin/env python3
"""Whole-batch result checksums at the BENCH shape (n = 2^15, 4 limbs), produced offline by the C restatement:
    python tests/golden/make_batch_checksums.py            (about two minutes on 8 cores)
-> tests/golden/batch_checksums.json.  alch_buf_checksum of the device result must equal these values:
   * bench_mul_relin: bench.py's default step (B = 8192, seeds 2026 / 900000007 / 0xA1C4E5) -- bench.py asserts it;
   * test_mul_relin / test_mul_full: B = 2 * 1024 + 37 (two full chunks on two streams plus a ragged tail at the default
     launch options) -- tests/test_gpu_bench_shape.py asserts them.
   * general_index: bench.py's extra line keySwitchQuadCirc(a*b) on H5' = F20475 with four HomomRLWR moduli (B = 4096, seeds 11 / 12 / 13)
     and a ragged test batch -- general-index kernels (kernel_gen.hpp);
   * homomrlwr: the HomomRLWR ringRound pipeline of alchemy_amd/ringround.py at the reference's indices, moduli and limb counts, one
     checksum per ciphertext of the B = 1024 batch bench.py times (tests/ringround_oracle.py replays the op sequence on the C
     restatement; about an hour of CPU time split over the cores);
   * tunnel_hs: the five BaseBGad-2 hops of alchemy_amd/tunnelhops.py (examples/Tunnel.hs), one checksum per ciphertext of the
     B = 256 batches bench.py times.
   * n16: the headline op at n = 2^16 on six limbs (split transforms), B = 2048 for bench.py's extra line and a ragged test batch;
   * q30: the headline op on four moduli below 2^30 (Harvey-butterfly kernels), B = 8192 for bench.py's extra line and the ragged
     test batch;
   * bench_extra: bench.py's `full_mul` line (PT2CT's whole mul_, 4 -> 5 -> 3 limbs, B = 4096) and its `pow_basis_in_out` line (the headline
     op with Pow-basis operands and result, B = 2048) -- the two driver-timed lines that only printed a checksum until round 4;
   Per-ciphertext lists let any prefix (a ragged test batch) be checked: sums are position-dependent, so they add.
A result error confined to any chunk, stream or persistent-workgroup slot changes the sum.

    python tests/golden/make_batch_checksums.py [two_power] [bench_extra] [q30] [n16] [general] [homomrlwr] [tunnel_hs]     (default: all; sections not
    regenerated are kept from the existing file)"""
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


def relin_range(first, count, qs=CFG3_QS, n=N):
    return cref.Ring(n, qs).mul_relin_checksum(SEED_A, SEED_B, SEED_H, first, count)


SIX_QS_17 = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]   # six primes < 2^31 that are 1 mod 2^17


Q30_QS = [1073479681, 1071513601, 1070727169, 1068236801]          # the four largest primes < 2^30 that are 1 mod 2^17


def full_range(first, count, qs_h=None):
    qs_h = qs_h or [FULL_EXTRA_Q] + CFG3_QS
    o_in, o_h = cref.Ring(N, qs_h[1:]), cref.Ring(N, qs_h)
    hint = [o_h.fill_uniform(SEED_H, i) for i in range(2 * len(qs_h))]
    s = 0
    for ct in range(first, first + count):
        a0, a1 = o_in.fill_uniform(SEED_A, 2 * ct), o_in.fill_uniform(SEED_A, 2 * ct + 1)
        b0, b1 = o_in.fill_uniform(SEED_B, 2 * ct), o_in.fill_uniform(SEED_B, 2 * ct + 1)
        w0, w1 = oracle_full_mul(cref, N, qs_h, 4, 3, hint, a0, a1, b0, b1)
        s += elem_checksum(w0, 2 * ct, 3) + elem_checksum(w1, 2 * ct + 1, 3)
    return s & MASK


def pow_range(first, count):
    """keySwitchQuadCirc(hint, a*b) with Pow-basis operands and result (ALCH_POW_IN | ALCH_POW_OUT; the hint stays CRT-basis data)."""
    o = cref.Ring(N, CFG3_QS)
    hint = [o.fill_uniform(SEED_H, i) for i in range(2 * len(CFG3_QS))]
    s = 0
    for ct in range(first, first + count):
        w0, w1 = o.ct_mul_relin(hint, o.fill_uniform(SEED_A, 2 * ct), o.fill_uniform(SEED_A, 2 * ct + 1),
                                o.fill_uniform(SEED_B, 2 * ct), o.fill_uniform(SEED_B, 2 * ct + 1), pow_basis=True)
        s += elem_checksum(w0, 2 * ct, 4) + elem_checksum(w1, 2 * ct + 1, 4)
    return s & MASK


def elem_checksum_n(aos, elem_index):
    """elem_checksum for a ring of any dimension: aos is the (n, L) element at position elem_index of its buffer."""
    n, L = aos.shape
    with np.errstate(over="ignore"):
        lm = np.ascontiguousarray(aos.T).astype(np.uint64).reshape(-1)
        w = np.uint64(elem_index * L * n) + np.arange(L * n, dtype=np.uint64)
        return int(splitmix64(w ^ (lm << np.uint64(20))).sum(dtype=np.uint64))


def ct_checksum(pair, ct):
    return (elem_checksum_n(pair[0], 2 * ct) + elem_checksum_n(pair[1], 2 * ct + 1)) & MASK


# ---- general index (bench.py general_index_line) ----
GEN_M, GEN_QS, GEN_SEEDS = 20475, [1543651201, 689270401, 718099201, 720720001], (11, 12, 13)
_gen = {}


def general_ct(ct):
    if "o" not in _gen:
        o = cref.GenRing(GEN_M, GEN_QS)
        _gen["o"], _gen["hint"] = o, [o.fill_uniform(GEN_SEEDS[2], e) for e in range(2 * len(GEN_QS))]
    o = _gen["o"]
    sa, sb = GEN_SEEDS[0], GEN_SEEDS[1]
    return ct_checksum(o.ct_mul_relin(_gen["hint"], o.fill_uniform(sa, 2 * ct), o.fill_uniform(sa, 2 * ct + 1),
                                      o.fill_uniform(sb, 2 * ct), o.fill_uniform(sb, 2 * ct + 1)), ct)


# ---- HomomRLWR pipeline (alchemy_amd/ringround.py) ----
_rr = {}


def ringround_ct(ct):
    if "o" not in _rr:
        from ringround_oracle import RingRoundOracle
        _rr["o"] = RingRoundOracle(cref)
    return ct_checksum(_rr["o"].run(ct), ct)


# ---- Tunnel.hs hops (alchemy_amd/tunnelhops.py) ----
_hops = {}


def hop_ct(args):
    k, ct = args
    if k not in _hops:
        from tunnelhops_oracle import HopOracle
        _hops[k] = HopOracle(cref, k)
    return ct_checksum(_hops[k].run(ct), ct)


# ---- BASELINE config 2: n = 2^14, one limb; crt of seeded polynomials and a pointwise product ----
C2_N, C2_QS, C2_SEEDS = 1 << 14, {"q60": 1152921504606748673, "q31": 2147352577}, (2026, 7)
_c2 = {}


def config2_poly(args):
    """(crt(a_e), crt(a_e) * b_e) checksums of polynomial e of the bench batch (a seeded 2026, b seeded 7)."""
    label, e = args
    if label not in _c2:
        _c2[label] = cref.Ring(C2_N, [C2_QS[label]])
    o = _c2[label]
    fa = o.crt(o.fill_uniform(C2_SEEDS[0], e))
    return elem_checksum_n(fa, e), elem_checksum_n(o.mul(fa, o.fill_uniform(C2_SEEDS[1], e)), e)


def pool_map(fn, items):
    import multiprocessing as mp
    with mp.get_context("fork").Pool(THREADS) as pool:
        return pool.map(fn, items, chunksize=max(1, len(items) // (8 * THREADS)))


def prefix(sums, count):
    return f"{sum(sums[:count]) & MASK:016x}"


if __name__ == "__main__":
    cref.build()
    path = os.path.join(HERE, "batch_checksums.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    want = set(sys.argv[1:]) or {"two_power", "bench_extra", "q30", "n16", "general", "homomrlwr", "tunnel_hs", "config2"}
    if "q30" in want:
        B_TEST, B_BENCH = 2 * 1024 + 37, 8192
        head = run_ranges(lambda f, c: relin_range(f, c, Q30_QS), B_TEST)
        tail = run_ranges(lambda f, c: relin_range(B_TEST + f, c, Q30_QS), B_BENCH - B_TEST)
        Q30_FULL = [1065484289] + Q30_QS                               # the hint's extra limb: the next prime of the same family
        fullq = run_ranges(lambda f, c: full_range(f, c, Q30_FULL), 2048)
        out["q30"] = {"what": "the headline op (same seeds, n = 2^15, 4 limbs) on moduli below 2^30: the Harvey-butterfly kernels",
                      "full_mul": {"batch": 2048, "limbs": "4 -> 5 -> 3", "moduli_hint": Q30_FULL, "checksum": f"{fullq:016x}"},
                      "moduli": Q30_QS, "first_2": f"{relin_range(0, 2, Q30_QS):016x}",
                      "test_mul_relin": {"batch": B_TEST, "checksum": f"{head:016x}"},
                      "bench_mul_relin": {"batch": B_BENCH, "checksum": f"{(head + tail) & MASK:016x}"}}
    if "two_power" in want:
        B_TEST, B_BENCH = 2 * 1024 + 37, 8192
        head = run_ranges(relin_range, B_TEST)
        tail = run_ranges(lambda f, c: relin_range(B_TEST + f, c), B_BENCH - B_TEST)
        full = run_ranges(full_range, B_TEST)
        out.update({"n": N, "moduli": CFG3_QS, "full_extra_modulus": FULL_EXTRA_Q, "seeds": {"a": SEED_A, "b": SEED_B, "hint": SEED_H},
                    "first_2": f"{relin_range(0, 2):016x}",
                    "rule": "sum over result words of splitmix64(w ^ value << 20), w = limb-major word position (alch_buf_checksum)",
                    "test_mul_relin": {"batch": B_TEST, "checksum": f"{head:016x}"},
                    "bench_mul_relin": {"batch": B_BENCH, "checksum": f"{(head + tail) & MASK:016x}"},
                    "test_mul_full": {"batch": B_TEST, "limbs": "4 -> 5 -> 3", "checksum": f"{full:016x}"}})
    if "bench_extra" in want:
        B_TEST, B_FULL, B_POW = out["test_mul_full"]["batch"], 4096, 2048          # sums are position dependent: the test batch is a prefix
        tail = run_ranges(lambda f, c: full_range(B_TEST + f, c), B_FULL - B_TEST)
        powc = run_ranges(pow_range, B_POW)
        out["bench_extra"] = {"what": "bench.py's full_mul line (same seeds as the headline, operands on the four config-3 moduli, hint on five, "
                                      "result on three) and its Pow-basis in/out line",
                              "full_mul": {"batch": B_FULL, "limbs": "4 -> 5 -> 3", "checksum": f"{(int(out['test_mul_full']['checksum'], 16) + tail) & MASK:016x}"},
                              "pow_in_out": {"batch": B_POW, "checksum": f"{powc:016x}", "first_2": f"{pow_range(0, 2):016x}"}}
    if "n16" in want:
        B_TEST, B_BENCH = 1024 + 37, 2048
        head = run_ranges(lambda f, c: relin_range(f, c, SIX_QS_17, 1 << 16), B_TEST)
        tail = run_ranges(lambda f, c: relin_range(B_TEST + f, c, SIX_QS_17, 1 << 16), B_BENCH - B_TEST)
        out["n16"] = {"what": "the headline op (same seeds) at n = 2^16 on SURVEY 8d's six primes that are 1 mod 2^17 (the two-power stand-in for "
                              "BASELINE configs 4 / 5's wording): split transforms", "n": 1 << 16, "moduli": SIX_QS_17,
                      "first_2": f"{relin_range(0, 2, SIX_QS_17, 1 << 16):016x}",
                      "test_mul_relin": {"batch": B_TEST, "checksum": f"{head:016x}"},
                      "bench_mul_relin": {"batch": B_BENCH, "checksum": f"{(head + tail) & MASK:016x}"}}
    if "general" in want:
        B_G, B_GT = 4096, 533
        sums = pool_map(general_ct, list(range(B_G)))
        out["general_index"] = {"index": GEN_M, "moduli": GEN_QS, "seeds": {"a": GEN_SEEDS[0], "b": GEN_SEEDS[1], "hint": GEN_SEEDS[2]},
                                "bench": {"batch": B_G, "checksum": prefix(sums, B_G)}, "test": {"batch": B_GT, "checksum": prefix(sums, B_GT)},
                                "first_4": prefix(sums, 4)}
    if "homomrlwr" in want:
        B_P = int(os.environ.get("ALCH_RLWR_BATCH", "1024"))
        sums = pool_map(ringround_ct, list(range(B_P)))
        out["homomrlwr"] = {"what": "alchemy_amd/ringround.py at default options: checksum of result ciphertext ct (elements 2ct, 2ct+1 of the "
                                    "one-limb result buffer over H5'), per ciphertext", "batch": B_P,
                            "per_ciphertext": [f"{v:016x}" for v in sums], "checksum": prefix(sums, B_P)}
    if "tunnel_hs" in want:
        B_T = int(os.environ.get("ALCH_TUNNEL_BATCH", "2048"))
        hops = []
        for k in range(5):
            sums = pool_map(hop_ct, [(k, ct) for ct in range(B_T)])
            hops.append({"hop": k, "batch": B_T, "per_ciphertext": [f"{v:016x}" for v in sums], "checksum": prefix(sums, B_T)})
        out["tunnel_hs"] = {"what": "alchemy_amd/tunnelhops.py: checksum of result ciphertext ct of hop H_k' -> H_k+1', per ciphertext", "hops": hops}
    if "config2" in want:
        B_C = 512
        out["config2"] = {"what": "tools/bench_config2.py / bench.py config2: polynomial e of the batch, a seeded 2026 and b seeded 7; checksum of "
                                  "crt(a) and of crt(a) * b over the first `prefix` polynomials", "n": C2_N, "prefix": B_C}
        for label in C2_QS:
            sums = pool_map(config2_poly, [(label, e) for e in range(B_C)])
            out["config2"][label] = {"modulus": C2_QS[label], "crt": f"{sum(a for a, _ in sums) & MASK:016x}",
                                     "crt_times_b": f"{sum(b for _, b in sums) & MASK:016x}",
                                     "first_2": [f"{sum(a for a, _ in sums[:2]) & MASK:016x}", f"{sum(b for _, b in sums[:2]) & MASK:016x}"]}
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk not in ("per_ciphertext", "hops")}) for k, v in out.items()}))
