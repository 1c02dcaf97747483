#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: alch_ct_mul_relin and alch_ct_mul_full on two-power rings of random size, limb count, moduli
class (31-bit, below 2^30), gadget (TrivGad; BaseBGad 2 at small sizes, hint at, above or below the operands' limb count), batch and launch options, every result word compared with the C restatement (oracle/ is the checker, as in
tests/).  usage: tests/sweeps/fuzz_parity.py [seconds] [seed]   -- prints one line per case class and a final tally; exits non-zero on a mismatch."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alchemy_amd as A
from alchemy_amd import capi
from oracle import cref
from helpers import oracle_full_mul, oracle_full_mul_base2_down, oracle_mul_relin_base2

SIX31 = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]          # = 1 mod 2^17
SIX30 = [1073479681, 1071513601, 1070727169, 1068236801, 1065484289, 1064697857]          # = 1 mod 2^17, below 2^30
OPTS = {"chunk": [8, 16, 24], "one_stream": [0, 1], "pipe": [0, 1], "nstreams": [1, 2, 3, 4], "ks_map": [0, 1], "ks_rev": [0, 1],
        "q30": [0, 1], "split_fused": [0, 1, 2], "ks_grid": [8, 24, 4096], "ti_split": [0, 1, 5], "crt_half": [0, 1]}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    cref.build()
    t0, cases, tally = time.time(), 0, {}
    print(f"seed {seed}", flush=True)
    while time.time() - t0 < budget:
        logn = rng.choice([4, 6, 8, 10, 11, 11, 12, 13, 14, 15, 15, 15, 16])
        n = 1 << logn
        pool = rng.choice([SIX31, SIX30])
        L = rng.randint(1, 6)
        qs = rng.sample(pool, L) if rng.random() < 0.5 else pool[:L]
        batch = rng.randint(1, 40 if logn <= 13 else (12 if logn <= 14 else 5))
        if logn >= 15 and rng.random() < 0.25: batch = rng.randint(9, 30)          # several chunks of 8 .. 24 at the fused kernels' sizes
        opts = {k: rng.choice(v) for k, v in OPTS.items() if rng.random() < 0.4}
        full = L >= 2 and rng.random() < 0.4
        rnd = lambda c, q_: np.stack([np.stack([nprng.integers(0, q, size=n, dtype=np.int64) for q in q_], axis=1) for _ in range(c)])
        s_pre = None if rng.random() < 0.5 else [rng.randrange(1, q) for q in qs]
        key = ("full" if full else "relin", logn, "q30" if pool is SIX30 else "q31")
        if logn <= 11 and L <= 3 and rng.random() < 0.25:              # BaseBGad 2 hints (small sizes: 30 digits per limb)
            batch = min(batch, 4)
            kind = rng.choice(["relin", "full_up", "full_down"]) if L >= 2 else "relin"
            key = ("base2_" + kind, logn, key[2])
            if kind == "relin":
                g = A.Ring(2 * n, qs)
                for k, v in opts.items(): g.set_option(k, v)
                D = g.gadget_digits(capi.ALCH_GAD_BASE2)
                hint, a, b = rnd(2 * D, qs), rnd(2 * batch, qs), rnd(2 * batch, qs)
                out = g.alloc(2 * batch)
                g.ct_mul_relin(g.hint_load(hint, gadget=capi.ALCH_GAD_BASE2), g.upload(a), g.upload(b), out, batch, s_pre=s_pre)
                got = out.download()
                want = [oracle_mul_relin_base2(cref, n, qs, list(hint), a[2 * c], a[2 * c + 1], b[2 * c], b[2 * c + 1], s_pre) for c in range(batch)]
            elif kind == "full_up":                                     # hint's ring at least as long as the operands'
                l_in, l_out = rng.randint(1, L), rng.randint(1, L)
                rh, rin, rout = A.Ring(2 * n, qs), A.Ring(2 * n, qs[L - l_in:]), A.Ring(2 * n, qs[L - l_out:])
                D = rh.gadget_digits(capi.ALCH_GAD_BASE2)
                hint, a, b = rnd(2 * D, qs), rnd(2 * batch, qs[L - l_in:]), rnd(2 * batch, qs[L - l_in:])
                sp = None if s_pre is None else s_pre[L - l_in:]
                out = rout.alloc(2 * batch)
                capi.ct_mul_full(rh.hint_load(hint, gadget=capi.ALCH_GAD_BASE2), rin.upload(a), rin.upload(b), out, batch, s_pre=sp)
                got = out.download()
                want = [oracle_full_mul(cref, n, qs, l_in, l_out, list(hint), a[2 * c], a[2 * c + 1], b[2 * c], b[2 * c + 1], sp, gadget="base2")
                        for c in range(batch)]
            else:                                                       # hint on fewer limbs than the operands
                l_h = rng.randint(1, L - 1)
                l_out = rng.randint(1, l_h)
                rin, rh, rout = A.Ring(2 * n, qs), A.Ring(2 * n, qs[L - l_h:]), A.Ring(2 * n, qs[L - l_out:])
                D = rh.gadget_digits(capi.ALCH_GAD_BASE2)
                hint, a, b = rnd(2 * D, qs[L - l_h:]), rnd(2 * batch, qs), rnd(2 * batch, qs)
                out = rout.alloc(2 * batch)
                capi.ct_mul_full(rh.hint_load(hint, gadget=capi.ALCH_GAD_BASE2), rin.upload(a), rin.upload(b), out, batch, s_pre=s_pre)
                got = out.download()
                want = [oracle_full_mul_base2_down(cref, n, qs, l_h, l_out, list(hint), a[2 * c], a[2 * c + 1], b[2 * c], b[2 * c + 1], s_pre)
                        for c in range(batch)]
            for c in range(batch):
                if not (np.array_equal(got[2 * c], want[c][0]) and np.array_equal(got[2 * c + 1], want[c][1])):
                    print("MISMATCH", key, dict(qs=qs, batch=batch, opts=opts, ct=c, seed=seed)); return 1
            cases += 1
            tally[key] = tally.get(key, 0) + 1
            continue
        if not full:
            g, o = A.Ring(2 * n, qs), cref.Ring(n, qs)
            for k, v in opts.items(): g.set_option(k, v)
            pow_basis = rng.random() < 0.3
            hint, a, b = rnd(2 * L, qs), rnd(2 * batch, qs), rnd(2 * batch, qs)
            out = g.alloc(2 * batch)
            g.ct_mul_relin(g.hint_load(hint), g.upload(a), g.upload(b), out, batch, s_pre=s_pre,
                           flags=(capi.ALCH_POW_IN | capi.ALCH_POW_OUT) if pow_basis else 0)
            got = out.download()
            for ct in range(batch):
                w0, w1 = o.ct_mul_relin(list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], s_pre=s_pre, pow_basis=pow_basis)
                if not (np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1)):
                    print("MISMATCH relin", dict(logn=logn, qs=qs, batch=batch, opts=opts, pow_basis=pow_basis, ct=ct, seed=seed)); return 1
        else:
            l_in = rng.randint(1, L - 1) if rng.random() < 0.8 else L
            l_out = rng.randint(max(1, L - 3), L if l_in < L else L - 1) if L > 1 else 1
            if l_out == L and l_in == L: l_out = L - 1
            if L - l_out > 3 or (l_in == L and l_out == L): continue
            rh, rin, rout = A.Ring(2 * n, qs), A.Ring(2 * n, qs[L - l_in:]), A.Ring(2 * n, qs[L - l_out:])
            if l_out == L: continue                                    # TrivGad full mul_ drops at least one limb
            for k, v in opts.items(): rh.set_option(k, v); rin.set_option(k, v); rout.set_option(k, v)
            pow_out = rng.random() < 0.3
            hint, a, b = rnd(2 * L, qs), rnd(2 * batch, qs[L - l_in:]), rnd(2 * batch, qs[L - l_in:])
            sp = None if s_pre is None else s_pre[L - l_in:]
            out = rout.alloc(2 * batch)
            if l_in == L:
                continue                                               # operands on the hint's own ring: alch_ct_mul_relin's case
            capi.ct_mul_full(rh.hint_load(hint), rin.upload(a), rin.upload(b), out, batch, s_pre=sp, flags=capi.ALCH_POW_OUT if pow_out else 0)
            got = out.download()
            for ct in range(batch):
                w0, w1 = oracle_full_mul(cref, n, qs, l_in, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], sp, pow_out=pow_out)
                if not (np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1)):
                    print("MISMATCH full", dict(logn=logn, qs=qs, l_in=l_in, l_out=l_out, batch=batch, opts=opts, pow_out=pow_out, ct=ct, seed=seed)); return 1
        cases += 1
        tally[key] = tally.get(key, 0) + 1
        if cases % 25 == 0: print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    for k in sorted(tally): print(k, tally[k])
    print(f"OK: {cases} random cases bit-exact against the oracle (seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
