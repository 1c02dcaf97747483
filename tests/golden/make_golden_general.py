#!/usr/bin/env python3
"""Regenerates the general-index fixtures from the exact model (oracle/model_gen.py): run from the repo root,
    python tests/golden/make_golden_general.py
Outputs (all values are Python ints; ring elements limb-major [L][n], Pow basis unless the key says otherwise):
  general_tensor_small.json   per index m: one RNS element and the model's crt, l, lInv, mulGPow/Dec, divGPow/Dec, g_crt
  tunnel_small.json           per tower (r, s, r', s'): linear function, tunnel hint, MSD input ciphertext over R'_q, SymmSHE.tunnel's
                              output over S'_q, and f(pt)
  general_mul_small.json      per (m, m'): a valid SymmSHE instance -- secret key, two LSD encryptions, a TrivGad hint,
                              keySwitchQuadCirc(hint, a*b) on one ring and PT2CT's whole mul_ (2 -> 3 -> 1 limbs), with
                              the decryptions the model obtained
The reference has no vectors for this path: these pin the C restatement and the HIP library to the model."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import primes_1_mod          # noqa: E402
from oracle import model_gen as G         # noqa: E402


def tensor_vectors():
    rng = random.Random(20260401)
    out = []
    for m in (12, 28, 45, 91, 100, 27):
        idx = G.Index(m)
        qs = primes_1_mod(m, 2, 1 << 29)
        a = [[rng.randrange(q) for _ in range(idx.n)] for q in qs]
        rec = {"m": m, "n": idx.n, "qs": qs, "a": a}
        for name, fn in (("crt", G.crt_def), ("l", G.l_def), ("linv", G.linv_def), ("mulg_pow", G.mulg_pow_def),
                         ("mulg_dec", G.mulg_dec_def), ("divg_pow", G.divg_pow_def), ("divg_dec", G.divg_dec_def)):
            rec[name] = [fn(al, idx, q) for al, q in zip(a, qs)]
        rec["g_crt"] = [G.g_crt(idx, q) for q in qs]
        z = [rng.randrange(-999, 1000) for _ in range(idx.n)]
        rec["z"] = z
        rec["z_mulg_pow"] = G.mulg_pow_def(z, idx, None)
        rec["z_mulg_dec"] = G.mulg_dec_def(z, idx, None)
        out.append(rec)
    return out


def mul_vectors():
    rng = random.Random(20260402)
    out = []
    for m, mp, p in ((4, 28, 8), (3, 45, 4)):
        small, big = G.Index(m), G.Index(mp)
        qs = primes_1_mod(mp, 3, 1 << 29)
        sk = G.g_gen_sk(big, rng)
        pa = [rng.randrange(p) for _ in range(small.n)]
        pb = [rng.randrange(p) for _ in range(small.n)]
        hint = G.g_ks_hint(sk, big, qs, rng)
        rec = {"m": m, "mp": mp, "p": p, "n": big.n, "qs": qs, "sk": sk, "pa": pa, "pb": pb,
               "hint": [[h0, h1] for h0, h1 in hint], "product": G.ring_mul_def(pa, pb, small, p)}
        # (1) keySwitchQuadCirc hint (a * b), everything on the three limbs
        a3, b3 = G.g_encrypt(sk, pa, small, big, p, qs, rng), G.g_encrypt(sk, pb, small, big, p, qs, rng)
        r3 = G.g_key_switch(hint, G.g_ct_mul(a3, b3))
        assert G.g_decrypt(sk, r3) == rec["product"]
        rec["relin"] = {"a": a3.c, "b": b3.c, "out": r3.c, "out_k": r3.k, "out_l": r3.l,
                        "s_pre": [pow(p, -1, q) for q in qs]}
        # (2) PT2CT's mul_: operands on the last two limbs, hint on three, result on the last one
        a2, b2 = G.g_encrypt(sk, pa, small, big, p, qs[1:], rng), G.g_encrypt(sk, pb, small, big, p, qs[1:], rng)
        f = G.g_mod_switch_down(G.g_key_switch(hint, G.g_mod_switch_up(G.g_ct_mul(a2, b2), qs[:1])), 2)
        assert G.g_decrypt(sk, f) == rec["product"]
        rec["full"] = {"a": a2.c, "b": b2.c, "out": f.c, "out_k": f.k, "out_l": f.l,
                       "s_pre": [pow(p, -1, q) for q in qs[1:]]}
        out.append(rec)
    return out


def tunnel_vectors():
    """A valid tunnel instance per index tower: E-linear f given on the relative decoding basis, tunnelHint, an LSD
    encryption, modSwitch up -> tunnel -> modSwitch down, and the decryption (= f(pt)) the model obtained."""
    import math
    rng = random.Random(20260403)
    out = []
    for r, s, rp, sp, p in ((8, 12, 40, 60, 4), (4, 6, 28, 42, 8)):
        T = G.tunnel_indices(r, s, rp, sp)
        qs = primes_1_mod(rp * sp // math.gcd(rp, sp), 3, 1 << 29)
        sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
        ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(T.r.n // T.e.n)]
        pt = [rng.randrange(p) for _ in range(T.r.n)]
        ct = G.g_to_msd(G.g_mod_switch_up(G.g_encrypt(sk_in, pt, T.r, T.rp, p, qs[1:], rng), qs[:1]))
        lin_q, hints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng)
        tun = G.g_tunnel(lin_q, hints, ct, T)
        down = G.g_mod_switch_down(tun, 1)
        want = G.eval_lin_dec(ys, G.linv_def(pt, T.r, p), T.e, T.r, T.s, p)
        assert G.g_decrypt(sk_out, down) == want
        out.append({"r": r, "s": s, "rp": rp, "sp": sp, "ep": T.ep.m, "p": p, "qs": qs, "ys": ys, "pt": pt, "f_of_pt": want,
                    "sk_in": sk_in, "sk_out": sk_out, "lin": lin_q, "hints": hints, "ct_in": ct.c, "ct_out": tun.c,
                    "ct_out_l": tun.l})
    return out


if __name__ == "__main__":
    for name, data in (("general_tensor_small.json", tensor_vectors()), ("general_mul_small.json", mul_vectors()),
                       ("tunnel_small.json", tunnel_vectors())):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
