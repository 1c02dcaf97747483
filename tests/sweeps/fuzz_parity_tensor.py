#!/usr/bin/env python3
"""Randomised parity sweep of the per-element Tensor methods the Haskell instance binds (host buffers through the C ABI): crt, crtInv,
mulGPow / mulGDec / mulGCRT, divGPow / divGDec / divGCRT (with Lol's Nothing), l, lInv, zipWithT (*) / (+) / (-), on random indices
(two-power and 2^a 3^b 5^c 7^d 13^e, phi <= 4000) and 1..4 moduli = 1 mod m, against the C restatements; crtInv . crt = id, divG . mulG = id
and l . lInv = id are checked on the way.  usage: tests/sweeps/fuzz_parity_tensor.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alchemy_amd as A
from oracle import cref
from helpers import primes_1_mod


def phi(m):
    r, p, t = m, 2, m
    while p * p <= t:
        if t % p == 0:
            r -= r // p
            while t % p == 0: t //= p
        p += 1
    return r - r // t if t > 1 else r


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    cref.build()
    t0, cases, tally = time.time(), 0, {}
    print(f"seed {seed}", flush=True)
    while time.time() - t0 < budget:
        if rng.random() < 0.3:
            m = 2 ** rng.randint(3, 13)
        else:
            m = 2 ** rng.choice([0, 0, 2, 3, 5, 7]) * 3 ** rng.choice([0, 1, 2]) * 5 ** rng.choice([0, 1, 2]) * 7 ** rng.choice([0, 1]) * 13 ** rng.choice([0, 1])
        n = phi(m)
        if m < 3 or n < 2 or n > 4096: continue
        L = rng.randint(1, 4)
        qs = primes_1_mod(m, L, lo=rng.choice([1 << 20, 1 << 28, 1 << 30]))
        if max(qs) >= 1 << 31: continue
        g, o = A.Ring(m, qs), cref.GenRing(m, qs)
        x = np.stack([nprng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1)
        y = np.stack([nprng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1)
        checks = [("crt", g.crt(x), o.crt(x)), ("crtinv", g.crtinv(x), o.crtinv(x)), ("l", g.l(x), o.l(x)), ("linv", g.linv(x), o.linv(x)),
                  ("mulg_pow", g.mulg_pow(x), o.mulg_pow(x)), ("mulg_dec", g.mulg_dec(x), o.mulg_dec(x)), ("mulg_crt", g.mulg_crt(x), o.mulg_crt(x)),
                  ("divg_crt", g.divg_crt(x), o.divg_crt(x)), ("mul", g.mul(x, y), o.mul(x, y)), ("add", g.add(x, y), o.add(x, y)),
                  ("sub", g.sub(x, y), o.sub(x, y))]
        for name, got, want in checks:
            if not np.array_equal(got, want):
                print("MISMATCH", name, dict(m=m, qs=qs, seed=seed)); return 1
        for name in ("divg_pow", "divg_dec"):                          # Maybe: both sides agree on Nothing, and on the quotient otherwise
            got, want = getattr(g, name)(x), getattr(o, name)(x)
            if (got is None) != (want is None) or (got is not None and not np.array_equal(got, want)):
                print("MISMATCH", name, dict(m=m, qs=qs, seed=seed)); return 1
        ok = (np.array_equal(g.crtinv(g.crt(x)), x) and np.array_equal(g.linv(g.l(x)), x) and np.array_equal(g.divg_pow(g.mulg_pow(x)), x)
              and np.array_equal(g.divg_dec(g.mulg_dec(x)), x) and np.array_equal(g.divg_crt(g.mulg_crt(x)), x))
        if not ok:
            print("ROUND TRIP FAILED", dict(m=m, qs=qs, seed=seed)); return 1
        cases += 1
        key = "two-power" if m & (m - 1) == 0 else ("phi < 500" if n < 500 else "phi <= 4096")
        tally[key] = tally.get(key, 0) + 1
        if cases % 50 == 0: print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    for k in sorted(tally): print(k, tally[k])
    print(f"OK: {cases} random rings, 13 Tensor methods + 5 round trips each, bit-exact against the oracle (seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
