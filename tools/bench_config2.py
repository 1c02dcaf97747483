#!/usr/bin/env python3
"""BASELINE config 2: n = 2^14, one RNS limb -- forward NTT, inverse NTT and pointwise multiply on a batch of
65 536 independent polynomials, for the 60-bit prime (64-bit device words) and for a 31-bit prime
(reference-compatible point, 32-bit device words).  Prints one JSON line per configuration.
Every rate is given twice: at the reference's 8-byte word (SURVEY 8d's algorithmic bytes: transform 2*n*8 = 262 144 B, pointwise mul
3*n*8 = 393 216 B) and at the word the device actually moves; only the latter may be compared with the 8 TB/s peak -- a 31-bit ring
stores 4-byte words, so its "8-byte" figure can exceed the peak (it did in round 1: 9.6 TB/s) and says nothing about the kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring

N = 1 << 14
POLYS = int(os.environ.get("ALCH_C2_POLYS", "65536"))
for label, q in (("q60", 1152921504606748673), ("q31", 2147352577)):
    ring = Ring(2 * N, [q])
    a, b, c = ring.alloc(POLYS), ring.alloc(POLYS), ring.alloc(POLYS)
    a.fill_uniform(2026); b.fill_uniform(7)
    ring.sync()

    def best(fn, reps=3):
        fn(); ring.sync()
        t = 1e9
        for _ in range(reps):
            ring.timer_start(); fn(); t = min(t, ring.timer_stop())
        return t * 1e-3

    t_f = best(lambda: a.crt())
    t_i = best(lambda: a.crtinv())
    t_m = best(lambda: c.mul(a, b, POLYS))
    wb = ring.word_bytes
    dev = lambda words, t: POLYS * words * N * wb / t / 1e9
    out = {"config": f"BASELINE config 2, n=2^14, 1 limb, {label} ({q}), {POLYS} polynomials, device words {wb} B",
           "ntt_per_s": POLYS / t_f, "intt_per_s": POLYS / t_i, "pointwise_mul_per_s": POLYS / t_m,
           "at_8_byte_words_GBs": {"ntt": POLYS * 262144 / t_f / 1e9, "intt": POLYS * 262144 / t_i / 1e9, "mul": POLYS * 393216 / t_m / 1e9},
           "at_device_word_GBs": {"ntt": dev(2, t_f), "intt": dev(2, t_i), "mul": dev(3, t_m)},
           "frac_of_8TBs_at_device_word": {"ntt": dev(2, t_f) / 8000, "intt": dev(2, t_i) / 8000, "mul": dev(3, t_m) / 8000}}
    print(json.dumps(out))
