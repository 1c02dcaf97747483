#!/usr/bin/env python3
"""Generate the committed golden fixtures from the exact big-integer model (oracle/model.py).

The reference ships no golden vectors, known-answer tests or fixtures for this path and cannot be run here
(SURVEY.md 4, 8c), so these vectors pin the oracle and the HIP path to the *mathematical definitions*
(direct-evaluation CRT, schoolbook negacyclic products) -- "parity unpinned" with respect to Lol itself.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.json

All ring elements are stored limb-major: elem[j][k] = coefficient/slot k of limb j.
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import model as M  # noqa: E402

ARITH_QS = [268440577, 8392193, 1073750017]           # examples/Arithmetic.hs:31-34
SMALL_QS = [12289, 40961, 65537]                      # all = 1 mod 4096


def rand_elem(rng, n, qs):
    return [[rng.randrange(q) for _ in range(n)] for q in qs]


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def gen_ntt():
    rng = random.Random(20261004)
    cases = []
    for n, qs in [(4, SMALL_QS), (8, SMALL_QS), (16, ARITH_QS), (64, ARITH_QS), (64, SMALL_QS)]:
        a, b = rand_elem(rng, n, qs), rand_elem(rng, n, qs)
        cases.append({
            "n": n, "qs": qs,
            "psi": [M.root_2n(q, n) for q in qs],
            "generator": [M.smallest_generator(q) for q in qs],
            "a": a, "b": b,
            "crt_a": [M.crt_def(al, q) for al, q in zip(a, qs)],
            "crtinv_b": [M.crtinv_def(bl, q) for bl, q in zip(b, qs)],
            "a_times_b": M.rns_mul(a, b, qs),                      # schoolbook negacyclic product
            "a_plus_b": M.rns_add(a, b, qs),
        })
    dump("ntt_small.json", {"cases": cases})


def gen_decompose():
    rng = random.Random(7)
    cases = []
    for n, qs in [(8, SMALL_QS), (16, ARITH_QS)]:
        c = rand_elem(rng, n, qs)
        # plant the extreme residues
        for j, q in enumerate(qs):
            c[j][0], c[j][1], c[j][2], c[j][3] = 0, q - 1, (q - 1) // 2, (q + 1) // 2
        triv = M.decompose_triv(c, qs)
        base2 = M.decompose_baseb(c, qs, 2)
        cases.append({
            "n": n, "qs": qs, "c": c,
            "triv_digits": triv,                                   # signed integer polynomials
            "triv_reduced": [M.rns_reduce(d, qs) for d in triv],
            "base2_count": len(base2),
            "base2_reduced": [M.rns_reduce(d, qs) for d in base2],
            "rescale_drop1": M.rescale_down(c, qs, 1),
        })
    dump("decompose.json", {"cases": cases})


def ct_to_json(ct):
    return {"enc": ct.enc, "k": ct.k, "l": ct.l, "c": ct.c}


def gen_mul_relin():
    """End to end: encrypt two plaintexts, (*) , keySwitchQuadCirc, decrypt == product in R_p."""
    cases = []
    for seed, (n, npt, p, qs) in enumerate([(16, 2, 7, ARITH_QS[:2]), (16, 2, 7, ARITH_QS), (64, 4, 2, SMALL_QS + [ARITH_QS[0]])]):
        rng = random.Random(1000 + seed)
        sk = M.gen_sk(n, 3.0, rng)
        pa = [rng.randrange(p) for _ in range(npt)]
        pb = [rng.randrange(p) for _ in range(npt)]
        cta = M.encrypt(sk, pa, p, qs, 3.0, rng)
        ctb = M.encrypt(sk, pb, p, qs, 3.0, rng)
        hint = M.ks_quad_circ_hint(sk, qs, 3.0, rng, "triv")
        prod = M.ct_mul(cta, ctb)
        out = M.key_switch_quad_circ(hint, prod)
        want_pt = M.negacyclic_mul(pa, pb, p)
        got_pt = M.decrypt(sk, out, npt)
        assert got_pt == want_pt, (got_pt, want_pt)
        assert M.decrypt(sk, prod, npt) == want_pt
        # scalar folded into the tensor product by the device path: both operands LSD -> toMSD = p^-1 mod q
        s_pre = [pow(p, -1, q) for q in qs]
        cases.append({
            "n": n, "npt": npt, "p": p, "qs": qs, "sk": sk, "pa": pa, "pb": pb, "want_pt": want_pt,
            "cta": ct_to_json(cta), "ctb": ct_to_json(ctb),
            "hint": [[h0, h1] for h0, h1 in hint.h],               # Pow basis, limb-major
            "s_pre": s_pre,
            "out": ct_to_json(out),                                # MSD, Pow basis: the bit-exactness contract
        })
    dump("mul_relin_small.json", {"cases": cases})


def gen_mul_relin_base2():
    """The same end-to-end check with the BaseBGad 2 gadget (PT2CT.hs:140): one hint row per signed binary digit."""
    cases = []
    for seed, (n, npt, p, qs) in enumerate([(16, 2, 7, ARITH_QS[:2]), (32, 4, 2, SMALL_QS[:3])]):
        rng = random.Random(7000 + seed)
        sk = M.gen_sk(n, 3.0, rng)
        pa = [rng.randrange(p) for _ in range(npt)]
        pb = [rng.randrange(p) for _ in range(npt)]
        cta = M.encrypt(sk, pa, p, qs, 3.0, rng)
        ctb = M.encrypt(sk, pb, p, qs, 3.0, rng)
        hint = M.ks_quad_circ_hint(sk, qs, 3.0, rng, "baseb")
        out = M.key_switch_quad_circ(hint, M.ct_mul(cta, ctb))
        want_pt = M.negacyclic_mul(pa, pb, p)
        assert M.decrypt(sk, out, npt) == want_pt
        cases.append({"n": n, "npt": npt, "p": p, "qs": qs, "sk": sk, "want_pt": want_pt,
                      "cta": ct_to_json(cta), "ctb": ct_to_json(ctb), "hint": [[h0, h1] for h0, h1 in hint.h],
                      "s_pre": [pow(p, -1, q) for q in qs], "out": ct_to_json(out)})
    dump("mul_relin_base2_small.json", {"cases": cases})


def gen_arithmetic():
    """examples/Arithmetic.hs: (x + y) * y on R_7 (index 4), ciphertext index 32 (n=16, BASELINE wording) --
    limb counts as PT2CT selects them: the product runs on 2 limbs, hint on 2, result on 1."""
    rng = random.Random(512)
    p, npt, n = 7, 2, 16
    qs = [ARITH_QS[1], ARITH_QS[0]]           # nested pair (q2,(q1)): last-taken modulus outermost
    sk = M.gen_sk(n, 3.0, rng)
    x = [rng.randrange(p) for _ in range(npt)]
    y = [rng.randrange(p) for _ in range(npt)]
    ctx = M.encrypt(sk, x, p, qs, 3.0, rng)
    cty = M.encrypt(sk, y, p, qs, 3.0, rng)
    hint = M.ks_quad_circ_hint(sk, qs, 3.0, rng, "triv")
    s = M.ct_add(ctx, cty)
    prod = M.ct_mul(s, cty)
    ks = M.key_switch_quad_circ(hint, prod)
    res = M.mod_switch_down(ks, 1)
    want = M.negacyclic_mul([(a + b) % p for a, b in zip(x, y)], y, p)
    assert M.decrypt(sk, res, npt) == want
    dump("arithmetic_m32.json", {
        "n": n, "npt": npt, "p": p, "qs": qs, "sk": sk, "x": x, "y": y, "want_pt": want,
        "ctx": ct_to_json(ctx), "cty": ct_to_json(cty), "hint": [[h0, h1] for h0, h1 in hint.h],
        "sum": ct_to_json(s), "ks_out": ct_to_json(ks), "result": ct_to_json(res),
    })


def gen_full_mul():
    """The complete PT2CT mul_ (PT2CT.hs:172-177): modSwitch . keySwitchQuad hint . modSwitch $ (x * y) with the
    hint one limb longer than the operands (KSPNoise for TrivGad, PT2CT.hs:139) and the result shorter.
    Limb order: outermost pair component first, so zq_in and zq_out are suffixes of the hint's limb list."""
    cases = []
    for seed, (n, npt, p, qs_h, l_in, l_out) in enumerate([
            (16, 2, 7, [1073750017, 8392193, 268440577], 2, 1),          # Arithmetic.hs moduli: 2 -> 3 -> 1 limbs
            (32, 4, 2, [40961, 65537, 12289, 114689, 147457], 4, 3)]):   # HomomRLWR's shape 4 -> 5 -> 3
        rng = random.Random(4000 + seed)
        qs_in, qs_out = qs_h[len(qs_h) - l_in:], qs_h[len(qs_h) - l_out:]
        sk = M.gen_sk(n, 3.0, rng)
        pa = [rng.randrange(p) for _ in range(npt)]
        pb = [rng.randrange(p) for _ in range(npt)]
        x = M.encrypt(sk, pa, p, qs_in, 3.0, rng)
        y = M.encrypt(sk, pb, p, qs_in, 3.0, rng)
        hint = M.ks_quad_circ_hint(sk, qs_h, 3.0, rng, "triv")
        prod = M.ct_mul(x, y)
        up = M.mod_switch_up(prod, qs_h[:len(qs_h) - l_in])
        ks = M.key_switch_quad_circ(hint, up)
        res = M.mod_switch_down(ks, len(qs_h) - l_out)
        want = M.negacyclic_mul(pa, pb, p)
        assert M.decrypt(sk, res, npt) == want, "model: full mul_ does not decrypt"
        cases.append({"n": n, "npt": npt, "p": p, "qs_hint": qs_h, "l_in": l_in, "l_out": l_out, "sk": sk,
                      "pa": pa, "pb": pb, "want_pt": want, "x": ct_to_json(x), "y": ct_to_json(y),
                      "hint": [[h0, h1] for h0, h1 in hint.h], "up": ct_to_json(up), "ks": ct_to_json(ks),
                      "result": ct_to_json(res)})
    dump("full_mul_small.json", {"cases": cases})


def gen_digests():
    """Full-size digests (SHA-256 of little-endian int64, limb-major) from the C restatement, which the
    small fixtures above pin to the exact model.  Inputs follow the synthetic-residue rule shared by
    orc_fill_uniform and alch_buf_fill_uniform, so no input data need be stored."""
    import numpy as np
    from oracle import cref
    out = []
    cfg3 = [2147352577, 2146959361, 2146041857, 2145976321]
    for logn, qs, seed in [(14, [1152921504606748673], 2026), (14, [2147352577], 2026), (15, cfg3, 2026)]:
        n = 1 << logn
        ring = cref.Ring(n, qs)
        L = len(qs)
        a = [ring.fill_uniform(seed, e) for e in range(4)]                 # a0, a1, b0, b1 as elements 0..3
        entry = {"logn": logn, "qs": qs, "seed": seed}
        entry["crt_elem0_sha256"] = hashlib.sha256(np.ascontiguousarray(ring.crt(a[0]).T).tobytes()).hexdigest()
        entry["crtinv_elem1_sha256"] = hashlib.sha256(np.ascontiguousarray(ring.crtinv(a[1]).T).tobytes()).hexdigest()
        prod = ring.mul(a[0], a[1])
        entry["mul_elem0_elem1_sha256"] = hashlib.sha256(np.ascontiguousarray(prod.T).tobytes()).hexdigest()
        if L > 1:
            hint = [ring.fill_uniform(0xA1C4E5, e) for e in range(2 * L)]
            o0, o1 = ring.ct_mul_relin(hint, a[0], a[1], a[2], a[3])
            entry["mul_relin_crt_sha256"] = hashlib.sha256(
                np.ascontiguousarray(o0.T).tobytes() + np.ascontiguousarray(o1.T).tobytes()).hexdigest()
            p0, p1 = ring.ct_mul_relin(hint, a[0], a[1], a[2], a[3], pow_basis=True)
            entry["mul_relin_pow_sha256"] = hashlib.sha256(
                np.ascontiguousarray(p0.T).tobytes() + np.ascontiguousarray(p1.T).tobytes()).hexdigest()
        out.append(entry)
    dump("digests_full.json", {"cases": out})


if __name__ == "__main__":
    gen_ntt()
    gen_decompose()
    gen_mul_relin()
    gen_mul_relin_base2()
    gen_arithmetic()
    gen_full_mul()
    gen_digests()
