#!/usr/bin/env python3
"""Randomised sweep of the Tensor methods between two rings m | m' (embedPow / embedDec / embedCRT, twacePowDec / twaceCRT, coeffs) on
random divisor pairs (phi(m') <= 400: the checker is the by-definition Python model, oracle/model_gen.py) and two moduli = 1 mod m',
plus the identities crt . embedPow = embedCRT . crt and crt . twacePowDec = twaceCRT . crt through the library's own crt.
usage: tests/sweeps/fuzz_parity_ext.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alchemy_amd as A
from oracle import model_gen as G
from helpers import primes_1_mod

limbs = lambda arr: np.asarray(arr).T.tolist()


def divisors(m):
    return [d for d in range(1, m + 1) if m % d == 0]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    t0, cases, tally = time.time(), 0, {}
    print(f"seed {seed}", flush=True)
    while time.time() - t0 < budget:
        mb = 2 ** rng.choice([0, 0, 2, 3, 4]) * 3 ** rng.choice([0, 1, 2]) * 5 ** rng.choice([0, 1]) * 7 ** rng.choice([0, 1]) * 13 ** rng.choice([0, 0, 1])
        if mb < 3: continue
        b = G.Index(mb)
        if b.n < 2 or b.n > 400: continue
        m = rng.choice([d for d in divisors(mb) if d % 4 != 2])          # indices are not 2 mod 4
        if m == mb and rng.random() < 0.8: continue
        s = G.Index(m)
        qs = primes_1_mod(mb, 2, lo=rng.choice([1 << 20, 1 << 29, 1 << 30]))
        if max(qs) >= 1 << 31: continue
        rs, rb = A.Ring(m, qs), A.Ring(mb, qs)
        x = np.stack([nprng.integers(0, q, size=s.n, dtype=np.int64) for q in qs], axis=1)
        y = np.stack([nprng.integers(0, q, size=b.n, dtype=np.int64) for q in qs], axis=1)
        xl, yl = limbs(x), limbs(y)
        ok = (limbs(rs.embed_pow(rb, x)) == [G.embed_pow(v, s, b) for v in xl]
              and limbs(rs.embed_dec(rb, x)) == [G.embed_dec_def(v, s, b, q) for v, q in zip(xl, qs)]
              and limbs(rs.embed_crt(rb, x)) == [G.embed_crt_def(v, s, b) for v in xl]
              and limbs(rs.twace_pow_dec(rb, y)) == [G.twace_pow_dec(v, s, b) for v in yl]
              and limbs(rs.twace_crt(rb, y)) == [G.twace_crt_def(v, s, b, q) for v, q in zip(yl, qs)]
              and np.array_equal(rb.crt(rs.embed_pow(rb, x)), rs.embed_crt(rb, rs.crt(x)))
              and np.array_equal(rs.crt(rs.twace_pow_dec(rb, y)), rs.twace_crt(rb, rb.crt(y))))
        cs = rs.coeffs(rb, y)
        for j in range(len(qs)):
            ok = ok and [c[:, j].tolist() for c in cs] == G.coeffs(yl[j], s, b)
        if not ok:
            print("MISMATCH", dict(m=m, mb=mb, qs=qs, seed=seed)); return 1
        cases += 1
        key = "m = 1" if m == 1 else ("m = m'" if m == mb else "proper")
        tally[key] = tally.get(key, 0) + 1
        if cases % 50 == 0: print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    for k in sorted(tally): print(k, tally[k])
    print(f"OK: {cases} random ring pairs, 6 methods + 2 identities each, equal to the by-definition model (seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
