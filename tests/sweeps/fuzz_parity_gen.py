#!/usr/bin/env python3
"""Randomised parity sweep on general cyclotomic indices: alch_ct_mul_relin (TrivGad), alch_ct_mul_full (TrivGad) and alch_ct_mod_switch
on random indices m = 2^a 3^b 5^c 7^d 13^e with phi(m) <= 3000, random 1..5 moduli = 1 mod m (29..31 bits), random batches and launch
options (gen_fused, gen_nt, rs_lin), every result word compared with the general C restatement.  usage: tests/sweeps/fuzz_parity_gen.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alchemy_amd as A
from alchemy_amd import capi
from oracle import cref
from helpers import oracle_full_mul_general, primes_1_mod

OPTS = {"gen_fused": [0, 1], "gen_nt": [0, 128, 256, 512], "rs_lin": [0, 1], "scratch_mib": [1, 64]}


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
        m = 2 ** rng.choice([0, 0, 1, 2, 3, 5]) * 3 ** rng.choice([0, 1, 2]) * 5 ** rng.choice([0, 1, 2]) * 7 ** rng.choice([0, 1]) * 13 ** rng.choice([0, 1])
        if m % 2 == 0 and m % 4 != 0: m *= 2                        # indices are not 2 mod 4
        n = phi(m)
        if n < 4 or n > 3000 or m & (m - 1) == 0: continue           # two-power indices have their own sweep
        if n < 200 and rng.random() < 0.85: continue                 # mostly rings of a few hundred to 3000 coefficients
        L = rng.randint(1, 5)
        qs = primes_1_mod(m, L, lo=rng.choice([1 << 28, 1 << 29, (1 << 30) + (1 << 29), 715_000_000]))
        if max(qs) >= 1 << 31: continue
        rng.shuffle(qs)
        batch = rng.randint(1, 9)
        opts = {k: rng.choice(v) for k, v in OPTS.items() if rng.random() < 0.4}
        rnd = lambda c, q_: np.stack([np.stack([nprng.integers(0, q, size=n, dtype=np.int64) for q in q_], axis=1) for _ in range(c)])
        s_pre = None if rng.random() < 0.5 else [rng.randrange(1, q) for q in qs]
        kind = rng.choice(["relin", "full", "modswitch"]) if L >= 2 else "relin"
        try:
            if kind == "relin":
                g, o = A.Ring(m, qs), cref.GenRing(m, qs)
                for k, v in opts.items(): g.set_option(k, v)
                hint, a, b = rnd(2 * L, qs), rnd(2 * batch, qs), rnd(2 * batch, qs)
                out = g.alloc(2 * batch)
                g.ct_mul_relin(g.hint_load(hint), g.upload(a), g.upload(b), out, batch, s_pre=s_pre)
                got = out.download()
                want = [o.ct_mul_relin(list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], s_pre=s_pre) for ct in range(batch)]
            elif kind == "full":
                l_in = rng.randint(1, L - 1)
                l_out = rng.randint(max(1, L - 3), L - 1)
                rh, rin, rout = A.Ring(m, qs), A.Ring(m, qs[L - l_in:]), A.Ring(m, qs[L - l_out:])
                for k, v in opts.items(): rh.set_option(k, v); rin.set_option(k, v); rout.set_option(k, v)
                pow_out = rng.random() < 0.3
                hint, a, b = rnd(2 * L, qs), rnd(2 * batch, qs[L - l_in:]), rnd(2 * batch, qs[L - l_in:])
                sp = None if s_pre is None else s_pre[L - l_in:]
                out = rout.alloc(2 * batch)
                capi.ct_mul_full(rh.hint_load(hint), rin.upload(a), rin.upload(b), out, batch, s_pre=sp, flags=capi.ALCH_POW_OUT if pow_out else 0)
                got = out.download()
                want = [oracle_full_mul_general(cref, m, qs, l_in, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], sp,
                                                pow_out=pow_out) for ct in range(batch)]
            else:
                ddn = rng.randint(1, min(3, L - 1))
                rin, rout = A.Ring(m, qs), A.Ring(m, qs[ddn:])
                for k, v in opts.items(): rin.set_option(k, v); rout.set_option(k, v)
                x = rnd(2 * batch, qs)
                out = rout.alloc(2 * batch)
                capi.ct_mod_switch(rin.upload(x), out, batch)
                got = out.download()
                want = []
                for ct in range(batch):
                    pair = []
                    for comp in range(2):
                        cur = cref.GenRing(m, qs).crtinv(x[2 * ct + comp])
                        if comp == 0: cur = cref.GenRing(m, qs).linv(cur)
                        for u in range(ddn): cur = cref.GenRing(m, qs[u:]).rescale_drop0(cur)
                        o_out = cref.GenRing(m, qs[ddn:])
                        if comp == 0: cur = o_out.l(cur)
                        pair.append(o_out.crt(cur))
                    want.append(tuple(pair))
        except capi.AlchemyError as e:
            if "not served" in str(e) or "UNSUPPORTED" in str(e) or "at most" in str(e): continue
            raise
        for ct in range(batch):
            if not (np.array_equal(got[2 * ct], want[ct][0]) and np.array_equal(got[2 * ct + 1], want[ct][1])):
                print("MISMATCH", kind, dict(m=m, qs=qs, batch=batch, opts=opts, ct=ct, seed=seed)); return 1
        cases += 1
        key = (kind, "phi < 200" if n < 200 else "phi < 1000" if n < 1000 else "phi <= 3000")
        tally[key] = tally.get(key, 0) + 1
        if cases % 25 == 0: print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    for k in sorted(tally): print(k, tally[k])
    print(f"OK: {cases} random general-index cases bit-exact against the oracle (seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
