#!/usr/bin/env python3
"""Randomised parity sweep of alch_ct_tunnel: random index pairs (r', s') with lcm <= 60000 and phi <= 3000 that satisfy Lol's tunnel
conditions (alch_tunnel_info decides), 1..4 moduli = 1 mod lcm(r', s'), TrivGad or BaseBGad 2 hints, random linear functions, hints,
ciphertexts, encoding scalars, tunnel_ep / tunnel_fused options and the Pow-basis-out flag, against the C restatement's composition
(tests/helpers.py::oracle_tunnel -- the checker of the tunnel tests).  usage: tests/sweeps/fuzz_parity_tunnel.py [seconds] [seed]"""
import math, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alchemy_amd as A
from alchemy_amd import capi
from oracle import cref
from helpers import oracle_tunnel, primes_1_mod


def phi(m):
    r, p, t = m, 2, m
    while p * p <= t:
        if t % p == 0:
            r -= r // p
            while t % p == 0: t //= p
        p += 1
    return r - r // t if t > 1 else r


def rand_index(rng):
    m = 2 ** rng.choice([0, 0, 2, 3]) * 3 ** rng.choice([0, 1, 2]) * 5 ** rng.choice([0, 1]) * 7 ** rng.choice([0, 1]) * 13 ** rng.choice([0, 0, 1])
    return m


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    cref.build()
    t0, cases, refused, tally = time.time(), 0, 0, {}
    print(f"seed {seed}", flush=True)
    while time.time() - t0 < budget:
        rp, sp = rand_index(rng), rand_index(rng)
        if rp == sp or min(rp, sp) < 3: continue
        lcm = rp * sp // math.gcd(rp, sp)
        if lcm > 60000 or max(phi(rp), phi(sp)) > 3000 or min(phi(rp), phi(sp)) < 4: continue
        if (rp & (rp - 1)) == 0 and (sp & (sp - 1)) == 0: continue
        L = rng.randint(1, 4)
        qs = primes_1_mod(lcm, L, lo=rng.choice([1 << 28, 1 << 29, 1 << 30]))
        if max(qs) >= 1 << 31: continue
        try:
            gr, gs = A.Ring(rp, qs), A.Ring(sp, qs)
            ep, d_rel = A.Tunnel.info(gr, gs)
        except capi.AlchemyError:
            refused += 1
            continue
        gadget = rng.choice(["triv", "triv", "base2"])
        D = gs.gadget_digits(capi.ALCH_GAD_BASE2) if gadget == "base2" else L
        if d_rel * D > 400 and gs.n > 500: continue                  # keep the oracle's work per case small
        batch = rng.randint(1, 4)
        rnd = lambda c, n: np.stack([np.stack([nprng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(c)])
        lin, ks, cts = rnd(d_rel, gs.n), rnd(2 * d_rel * D, gs.n), rnd(2 * batch, gr.n)
        s_pre = None if rng.random() < 0.5 else [rng.randrange(1, q) for q in qs]
        pow_out = rng.random() < 0.3
        gs.set_option("tunnel_ep", rng.choice([0, 1]))
        if gadget == "triv": gs.set_option("tunnel_fused", rng.choice([0, 2, 4]))
        tun = A.Tunnel(gr, gs, gs.upload(lin), gs.upload(ks), gadget=capi.ALCH_GAD_BASE2 if gadget == "base2" else capi.ALCH_GAD_TRIV)
        out = gs.alloc(2 * batch)
        tun.apply(gr.upload(cts), out, batch, s_pre=s_pre, flags=capi.ALCH_POW_OUT if pow_out else 0)
        got = out.download()
        for ct in range(batch):
            w0, w1 = oracle_tunnel(cref, rp, sp, qs, list(lin), list(ks), cts[2 * ct], cts[2 * ct + 1], s_pre, pow_out=pow_out, gadget=gadget)
            if not (np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1)):
                print("MISMATCH", dict(rp=rp, sp=sp, qs=qs, gadget=gadget, batch=batch, pow_out=pow_out, ct=ct, seed=seed)); return 1
        cases += 1
        key = (gadget, "d_rel %d" % d_rel)
        tally[key] = tally.get(key, 0) + 1
        if cases % 20 == 0: print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
    for k in sorted(tally): print(k, tally[k])
    print(f"OK: {cases} random tunnels bit-exact against the oracle ({refused} index pairs refused by alch_tunnel_info; seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
