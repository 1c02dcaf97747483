"""Shared helpers for the test-suite: layout conversions between the fixtures (limb-major lists), the C
oracle / C ABI host layout (numpy int64, shape (n, L), Lol's tuple-interleaved order) and SHA-256 digests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def to_aos(elem):
    """limb-major [L][n] python ints -> (n, L) int64 (host layout)."""
    return np.ascontiguousarray(np.array(elem, dtype=np.int64).T)


def from_aos(arr):
    return np.asarray(arr).T.tolist()


def digest_limb_major(*aos_arrays):
    h = hashlib.sha256()
    for a in aos_arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.int64).T).tobytes())
    return h.hexdigest()


def hint_to_crt_aos(ring_oracle, hint_pow):
    """fixture hint [[h0_i, h1_i]] (Pow basis, limb-major) -> list of 2*D CRT-basis (n, L) arrays in the
    order the C ABI wants: h0_0, h1_0, h0_1, h1_1, ..."""
    out = []
    for h0, h1 in hint_pow:
        out.append(ring_oracle.crt(to_aos(h0)))
        out.append(ring_oracle.crt(to_aos(h1)))
    return out


def oracle_full_mul(oracle_lib, n, qs_h, l_in, l_out, hint_crt, a0, a1, b0, b1, s_pre=None, pow_out=False, gadget="triv"):
    """PT2CT's whole mul_ (PT2CT.hs:172-177) composed from the C restatement's primitives:
    modSwitch (keySwitchQuadCirc hint (modSwitch (a * b))) with operands on the last l_in limbs of qs_h, the hint on
    all of qs_h and the result on the last l_out limbs.  Operands / hint / result in the CRT basis ((n, L) int64),
    result in the Pow basis with pow_out.  Pinned to the exact model by tests/golden/full_mul_small.json."""
    L = len(qs_h)
    dup = L - l_in
    o_in, o_h = oracle_lib.Ring(n, qs_h[dup:]), oracle_lib.Ring(n, qs_h)
    s = list(s_pre) if s_pre is not None else [1] * l_in
    # (*) and the encoding scalars
    c = [o_in.mul(a0, b0), o_in.add(o_in.mul(a0, b1), o_in.mul(a1, b0)), o_in.mul(a1, b1)]
    c = [o_in.scale(x, s) for x in c]
    # modSwitch up: Rescale b -> (a, b):  x -> (0, q_a x)
    mult = 1
    for q in qs_h[:dup]:
        mult *= q
    up = []
    for x in c:
        scaled = o_in.scale(x, [mult % q for q in qs_h[dup:]])
        up.append(np.ascontiguousarray(np.concatenate([np.zeros((n, dup), dtype=np.int64), scaled], axis=1)))
    # keySwitchQuadCirc: [c0, c1] + sum_i crt(reduce d_i) * hint_i
    digs = o_h.decompose_triv(o_h.crtinv(up[2])) if gadget == "triv" else o_h.decompose_base2(o_h.crtinv(up[2]))
    assert 2 * len(digs) == len(hint_crt)
    ks = [up[0], up[1]]
    for i, d in enumerate(digs):
        dc = o_h.crt(d)
        ks[0] = o_h.add(ks[0], o_h.mul(dc, hint_crt[2 * i]))
        ks[1] = o_h.add(ks[1], o_h.mul(dc, hint_crt[2 * i + 1]))
    # modSwitch down: Rescale (a, b) -> b in the Pow basis, outermost limb first
    out = []
    for x in ks:
        cur = o_h.crtinv(x)
        for k in range(L, l_out, -1):
            cur = oracle_lib.Ring(n, qs_h[L - k:]).rescale_drop0(cur)
        out.append(cur if pow_out else oracle_lib.Ring(n, qs_h[L - l_out:]).crt(cur))
    return out[0], out[1]


def oracle_full_mul_base2_down(oracle_lib, n, qs_in, l_h, l_out, hint_crt, a0, a1, b0, b1, s_pre=None, pow_out=False, m=None):
    """PT2CT's whole mul_ with a BaseBGad 2 hint on FEWER limbs than the operands (PT2CT.hs:140,164), from the C restatement's
    primitives: (*) on the operands' ring qs_in (mulG on a general index m), modSwitch DOWN of the quadratic ciphertext to the last l_h
    limbs (rescaleDec on c0, rescalePow on c1 and c2), BaseBGad 2 decomposition of c2, hint products, modSwitch down to l_out limbs."""
    L = len(qs_in)
    ring = (lambda qs: oracle_lib.GenRing(m, qs)) if m else (lambda qs: oracle_lib.Ring(n, qs))
    o_in, o_h, o_out = ring(qs_in), ring(qs_in[L - l_h:]), ring(qs_in[L - l_out:])
    s = list(s_pre) if s_pre is not None else [1] * L
    c = [o_in.mul(a0, b0), o_in.add(o_in.mul(a0, b1), o_in.mul(a1, b0)), o_in.mul(a1, b1)]
    c = [o_in.scale(o_in.mulg_crt(x) if m else x, s) for x in c]

    def down(x_pow, l_from, l_to, dec):
        cur = ring(qs_in[L - l_from:]).linv(x_pow) if dec else x_pow
        for k in range(l_from, l_to, -1):
            cur = ring(qs_in[L - k:]).rescale_drop0(cur)
        return ring(qs_in[L - l_to:]).l(cur) if dec else cur

    dec0 = bool(m)                                   # two-power index: the decoding basis is the powerful basis
    low = [down(o_in.crtinv(x), L, l_h, dec0 and comp == 0) for comp, x in enumerate(c)]
    digs = decompose_base2(low[2], qs_in[L - l_h:]) if m else o_h.decompose_base2(low[2])
    assert 2 * len(digs) == len(hint_crt)
    ks = [o_h.crt(low[0]), o_h.crt(low[1])]
    for i, d in enumerate(digs):
        dc = o_h.crt(d)
        ks[0] = o_h.add(ks[0], o_h.mul(dc, hint_crt[2 * i]))
        ks[1] = o_h.add(ks[1], o_h.mul(dc, hint_crt[2 * i + 1]))
    out = []
    for comp, x in enumerate(ks):
        cur = down(o_h.crtinv(x), l_h, l_out, dec0 and comp == 0)
        out.append(cur if pow_out else o_out.crt(cur))
    return out[0], out[1]


def oracle_mul_relin_base2(oracle_lib, n, qs, hint_crt, a0, a1, b0, b1, s_pre=None, pow_out=False):
    """keySwitchQuadCirc hint (a * b) with a BaseBGad 2 hint, composed from the C restatement's primitives
    (CRT-basis operands and hint).  Pinned to the exact model by tests/golden/mul_relin_base2_small.json."""
    o = oracle_lib.Ring(n, qs)
    s = list(s_pre) if s_pre is not None else [1] * len(qs)
    c0 = o.scale(o.mul(a0, b0), s)
    c1 = o.scale(o.add(o.mul(a0, b1), o.mul(a1, b0)), s)
    c2 = o.scale(o.mul(a1, b1), s)
    digs = o.decompose_base2(o.crtinv(c2))
    assert 2 * len(digs) == len(hint_crt)
    for i, d in enumerate(digs):
        dc = o.crt(d)
        c0 = o.add(c0, o.mul(dc, hint_crt[2 * i]))
        c1 = o.add(c1, o.mul(dc, hint_crt[2 * i + 1]))
    return (o.crtinv(c0), o.crtinv(c1)) if pow_out else (c0, c1)


def primes_1_mod(m, count, lo=0):
    """The first `count` primes q > lo with q = 1 (mod m)."""
    from oracle.model import is_prime
    out, q = [], (lo // m) * m + 1
    while len(out) < count:
        if q > max(lo, 2) and is_prime(q):
            out.append(q)
        q += m
    return out


def oracle_full_mul_general(oracle_lib, m, qs_h, l_in, l_out, hint_crt, a0, a1, b0, b1, s_pre=None, pow_out=False):
    """PT2CT's whole mul_ on a GENERAL cyclotomic index, composed from the general C restatement's primitives:
    (*) with mulG on every product coefficient (Eval.hs:65-67), modSwitch up, keySwitchQuadCirc (Eval.hs:133),
    modSwitch down with rescaleDec on c0 and rescalePow on c1 (Eval.hs:130).  Same layout rules as oracle_full_mul."""
    L = len(qs_h)
    dup = L - l_in
    o_in, o_h = oracle_lib.GenRing(m, qs_h[dup:]), oracle_lib.GenRing(m, qs_h)
    n = o_h.n
    s = list(s_pre) if s_pre is not None else [1] * l_in
    c = [o_in.mul(a0, b0), o_in.add(o_in.mul(a0, b1), o_in.mul(a1, b0)), o_in.mul(a1, b1)]
    c = [o_in.scale(o_in.mulg_crt(x), s) for x in c]
    mult = 1
    for q in qs_h[:dup]:
        mult *= q
    up = []
    for x in c:
        scaled = o_in.scale(x, [mult % q for q in qs_h[dup:]])
        up.append(np.ascontiguousarray(np.concatenate([np.zeros((n, dup), dtype=np.int64), scaled], axis=1)))
    digs = o_h.decompose_triv(o_h.crtinv(up[2]))
    ks = [up[0], up[1]]
    for i, d in enumerate(digs):
        dc = o_h.crt(d)
        ks[0] = o_h.add(ks[0], o_h.mul(dc, hint_crt[2 * i]))
        ks[1] = o_h.add(ks[1], o_h.mul(dc, hint_crt[2 * i + 1]))
    out = []
    for comp, x in enumerate(ks):
        cur = o_h.crtinv(x)
        if comp == 0:
            cur = o_h.linv(cur)                                   # c0: rescaleDec
        for k in range(L, l_out, -1):
            cur = oracle_lib.GenRing(m, qs_h[L - k:]).rescale_drop0(cur)
        o_out = oracle_lib.GenRing(m, qs_h[L - l_out:])
        if comp == 0:
            cur = o_out.l(cur)
        out.append(cur if pow_out else o_out.crt(cur))
    return out[0], out[1]


def decompose_base2(elem, qs):
    """BaseBGad 2 decompose + reduce of one Pow-basis element ((n, L) residues), index-agnostic: per limb i the centred lift is
    split into ceil(log2 q_i) balanced binary digits (remainder 1 taken as -1, the top digit absorbs the rest), limb 0's digits
    first; every digit is returned reduced into all limbs.  Checked against the C restatement (cref.Ring.decompose_base2) in
    tests/test_oracle_general.py."""
    out = []
    qv = np.array(qs, dtype=np.int64)
    for i, q in enumerate(qs):
        v = elem[:, i].astype(np.int64)
        v = np.where(v > (q - 1) // 2, v - q, v)
        kd = (q - 1).bit_length()
        for t in range(kd):
            if t + 1 < kd:
                d = -(v & 1)
                v = (v - d) // 2
            else:
                d = v
            out.append(np.ascontiguousarray(np.mod(d[:, None], qv[None, :])))
    return out


def oracle_tunnel(oracle_lib, r_p, s_p, qs, lin_crt, ks_crt, c0, c1, s_pre=None, pow_out=False, gadget="triv"):
    """SymmSHE.tunnel (Eval.hs:134) on one linear ciphertext (k = 0), composed from the general C restatement's primitives and
    the model's index maps -- following the definition literally: full lInv on R', Tensor `coeffs`, l on every E'-coefficient,
    embedPow into S', crt, times f'(d_i); for c1: `coeffs` on the Pow basis, embedPow, gadget decompose (TrivGad or BaseBGad 2,
    D digits), crt, hint products.
    c0, c1: CRT basis over R' ((n_r, L)); lin_crt: d_rel CRT elements of S'; ks_crt: [(i * D + t) * 2 + {0: b, 1: a}]."""
    import math
    from oracle import model_gen as G
    e_p = math.gcd(r_p, s_p)
    ie, ir, isx = G.Index(e_p), G.Index(r_p), G.Index(s_p)
    Or, Os, Oe = oracle_lib.GenRing(r_p, qs), oracle_lib.GenRing(s_p, qs), oracle_lib.GenRing(e_p, qs)
    L = len(qs)
    s = list(s_pre) if s_pre is not None else [1] * L
    c0p, c1p = Or.scale(Or.crtinv(c0), s), Or.scale(Or.crtinv(c1), s)
    c0d = Or.linv(c0p)
    rows, emb = np.array(G.coeffs_indices(ie, ir)), np.array(G.embed_indices(ie, isx))
    acc0, acc1 = np.zeros((isx.n, L), dtype=np.int64), np.zeros((isx.n, L), dtype=np.int64)
    for i, row in enumerate(rows):
        x = np.zeros((isx.n, L), dtype=np.int64)
        x[emb] = Oe.l(np.ascontiguousarray(c0d[row]))
        acc0 = Os.add(acc0, Os.mul(Os.crt(x), lin_crt[i]))
        x1 = np.zeros((isx.n, L), dtype=np.int64)
        x1[emb] = c1p[row]
        digs = Os.decompose_triv(x1) if gadget == "triv" else decompose_base2(x1, qs)
        for t, d in enumerate(digs):
            dc = Os.crt(d)
            acc0 = Os.add(acc0, Os.mul(dc, ks_crt[(i * len(digs) + t) * 2]))
            acc1 = Os.add(acc1, Os.mul(dc, ks_crt[(i * len(digs) + t) * 2 + 1]))
    return (Os.crtinv(acc0), Os.crtinv(acc1)) if pow_out else (acc0, acc1)
