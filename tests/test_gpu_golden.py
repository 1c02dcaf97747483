"""GPU: the HIP path through the C ABI against the committed golden fixtures (exact-model vectors) and, at
BASELINE.json's full sizes, against SHA-256 digests and size-independent properties."""
import numpy as np
import pytest

from conftest import CFG2_Q60, CFG3_QS
from helpers import digest_limb_major, from_aos, load_golden, to_aos

pytestmark = pytest.mark.gpu


def _ring(n, qs):
    import alchemy_amd as A
    return A.Ring(2 * n, qs)


def test_golden_ntt_vectors():
    for case in load_golden("ntt_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        if n < 16:
            continue                      # the device path starts at n = 16 (index 32); smaller n are oracle-only
        g = _ring(n, qs)
        a, b = to_aos(case["a"]), to_aos(case["b"])
        assert from_aos(g.crt(a)) == case["crt_a"]
        assert from_aos(g.crtinv(b)) == case["crtinv_b"]
        assert from_aos(g.crtinv(g.mul(g.crt(a), g.crt(b)))) == case["a_times_b"]     # schoolbook product
        assert from_aos(g.add(a, b)) == case["a_plus_b"]
        assert np.array_equal(g.sub(g.add(a, b), b), a)
        # mulG / divG: identity for a two-power index, divG always succeeds (Lol's Just)
        assert np.array_equal(g.mulg_pow(a), a) and np.array_equal(g.divg_crt(a), a)


def test_golden_decompose_and_rescale():
    import alchemy_amd as A
    for case in load_golden("decompose.json")["cases"]:
        n, qs = case["n"], case["qs"]
        if n < 16:
            continue
        g = _ring(n, qs)
        c = to_aos(case["c"])
        assert [from_aos(d) for d in g.decompose_triv(c)] == case["triv_reduced"]
        b2 = g.decompose_base2(c)
        assert len(b2) == case["base2_count"]
        assert [from_aos(d) for d in b2] == case["base2_reduced"]          # BaseBGad 2
        # device-resident decompose
        src, dst = g.upload(c[None]), g.alloc(len(qs))
        src.decompose_triv_into(0, dst, 0)
        assert [from_aos(d) for d in dst.download()] == case["triv_reduced"]
        # Rescale (a,b) -> b
        small = A.Ring(2 * n, qs[1:])
        out = small.alloc(1)
        src.rescale_drop0_into(out, 1)
        assert from_aos(out.download()[0]) == case["rescale_drop1"]


def test_golden_mul_relin_pow_basis():
    import alchemy_amd as A
    for case in load_golden("mul_relin_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        g = _ring(n, qs)
        hint_pow = np.stack([to_aos(h) for pair in case["hint"] for h in pair])
        hb = g.upload(hint_pow)
        hb.crt()
        hint = g.hint_from_buf(hb)
        a = g.upload(np.stack([to_aos(x) for x in case["cta"]["c"]]))
        b = g.upload(np.stack([to_aos(x) for x in case["ctb"]["c"]]))
        out = g.alloc(2)
        g.ct_mul_relin(hint, a, b, out, 1, s_pre=case["s_pre"], flags=A.capi.ALCH_POW_IN | A.capi.ALCH_POW_OUT)
        got = out.download()
        assert from_aos(got[0]) == case["out"]["c"][0]
        assert from_aos(got[1]) == case["out"]["c"][1]


def test_golden_full_pt2ct_mul_sequence():
    """modSwitch . keySwitchQuad hint . modSwitch $ (x * y)  (PT2CT.hs:172-177) with L_in -> L_in+1 -> L_out limbs,
    replayed with device-resident ops exactly in E's order (Eval.hs:65-67,130,133)."""
    import alchemy_amd as A
    for case in load_golden("full_mul_small.json")["cases"]:
        n, p, qs_h, l_in, l_out = case["n"], case["p"], case["qs_hint"], case["l_in"], case["l_out"]
        L = len(qs_h)
        rings = {k: A.Ring(2 * n, qs_h[L - k:]) for k in range(l_out, L + 1)}
        rin, rh = rings[l_in], rings[L]
        # (*) : both operands LSD already (fresh encryptions); polynomial product in S on the CRT basis
        x = rin.upload(np.stack([to_aos(c) for c in case["x"]["c"]])); x.crt()
        y = rin.upload(np.stack([to_aos(c) for c in case["y"]["c"]])); y.crt()
        prod, tmp = rin.alloc(3), rin.alloc(1)
        _mul(prod, 0, x, 0, y, 0); _mul(prod, 1, x, 0, y, 1); _mul(tmp, 0, x, 1, y, 0)
        _add(prod, 1, prod, 1, tmp, 0); _mul(prod, 2, x, 1, y, 1)
        # modSwitch up: toMSD (c *= p^-1), then Rescale b -> (a,b) one limb at a time
        prod.scale(prod, 3, [pow(p, -1, q) for q in rin.qs])
        cur = prod
        for k in range(l_in + 1, L + 1):
            nxt = rings[k].alloc(3)
            cur.rescale_add0_into(nxt, 3)
            cur = nxt
        cur.crtinv()
        assert [from_aos(e) for e in cur.download()] == case["up"]["c"]
        # keySwitchQuadCirc: decompose c2 (Pow), crt the digits, inner product with the hint, add c0, c1
        digs = rh.alloc(L)
        cur.decompose_triv_into(2, digs, 0)
        digs.crt(); cur.crt()
        hb = rh.upload(np.stack([to_aos(h) for pair in case["hint"] for h in pair])); hb.crt()
        ks = rh.alloc(2)
        _copy(ks, 0, cur, 0, rh); _copy(ks, 1, cur, 1, rh)
        t = rh.alloc(1)
        for i in range(L):
            _mul(t, 0, digs, i, hb, 2 * i); _add(ks, 0, ks, 0, t, 0)
            _mul(t, 0, digs, i, hb, 2 * i + 1); _add(ks, 1, ks, 1, t, 0)
        ks.crtinv()
        assert [from_aos(e) for e in ks.download()] == case["ks"]["c"]
        # modSwitch down: Rescale (a,b) -> b, outermost limb first
        cur = ks
        for k in range(L - 1, l_out - 1, -1):
            nxt = rings[k].alloc(2)
            cur.rescale_drop0_into(nxt, 2)
            cur = nxt
        assert [from_aos(e) for e in cur.download()] == case["result"]["c"]


# helpers for element-wise ops on single elements of device buffers --------------------------------------
def _binop(fn_name, dst, di, a, ai, b, bi):
    ring = dst.ring
    x = a.download(ai, 1); yv = b.download(bi, 1)
    ta, tb, td = ring.upload(x), ring.upload(yv), ring.alloc(1)
    getattr(td, fn_name)(ta, tb, 1)
    dst.upload(td.download(), di)


def _mul(dst, di, a, ai, b, bi): _binop("mul", dst, di, a, ai, b, bi)
def _add(dst, di, a, ai, b, bi): _binop("add", dst, di, a, ai, b, bi)
def _copy(dst, di, src, si, ring): dst.upload(src.download(si, 1), di)


def test_full_size_digests_config2_and_3():
    import alchemy_amd as A
    for case in load_golden("digests_full.json")["cases"]:
        n, qs, seed = 1 << case["logn"], case["qs"], case["seed"]
        g = _ring(n, qs)
        L = len(qs)
        buf = g.alloc(4)
        buf.fill_uniform(seed)
        a = buf.download()
        e0, e1 = g.upload(a[:1]), g.upload(a[1:2])
        e0.crt(); e1.crtinv()
        assert digest_limb_major(e0.download()[0]) == case["crt_elem0_sha256"]
        assert digest_limb_major(e1.download()[0]) == case["crtinv_elem1_sha256"]
        m = g.alloc(1)
        m.mul(g.upload(a[:1]), g.upload(a[1:2]), 1)
        assert digest_limb_major(m.download()[0]) == case["mul_elem0_elem1_sha256"]
        if L > 1:
            hs = g.alloc(2 * L); hs.fill_uniform(0xA1C4E5)
            hint = g.hint_from_buf(hs)
            ca, cb, out = g.upload(a[0:2]), g.upload(a[2:4]), g.alloc(2)
            g.ct_mul_relin(hint, ca, cb, out, 1)
            o = out.download()
            assert digest_limb_major(o[0], o[1]) == case["mul_relin_crt_sha256"]
            g.ct_mul_relin(hint, ca, cb, out, 1, flags=A.capi.ALCH_POW_IN | A.capi.ALCH_POW_OUT)
            o = out.download()
            assert digest_limb_major(o[0], o[1]) == case["mul_relin_pow_sha256"]


def test_full_size_properties_config3():
    """n = 2^15, 4 limbs, a batch larger than one chunk: round trip, linearity of crt, and position /
    chunking independence of ct_mul_relin (checksums are position-sensitive hashes of whole buffers)."""
    g = _ring(1 << 15, CFG3_QS)
    E = 64
    x, y, s = g.alloc(E), g.alloc(E), g.alloc(E)
    x.fill_uniform(11); y.fill_uniform(12)
    c0 = x.checksum()
    x.crt(); x.crtinv()
    assert x.checksum() == c0                                         # crtInv . crt = id
    s.add(x, y, E); s.crt()
    x.crt(); y.crt()
    t = g.alloc(E); t.add(x, y, E)
    assert t.checksum() == s.checksum()                               # crt(a + b) = crt(a) + crt(b)
    # the same ciphertext pair at two batch positions gives the same result
    B = 24
    hs = g.alloc(8); hs.fill_uniform(0xA1C4E5)
    hint = g.hint_from_buf(hs)
    a, b, out = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B)
    a.fill_uniform(21); b.fill_uniform(22)
    first_a, first_b = a.download(0, 2), b.download(0, 2)
    a.upload(first_a, 2 * (B - 1)); b.upload(first_b, 2 * (B - 1))
    g.ct_mul_relin(hint, a, b, out, B)
    assert out.checksum(0, 2) != 0
    assert np.array_equal(out.download(0, 2), out.download(2 * (B - 1), 2))


def test_config2_60bit_pipeline():
    """BASELINE config 2: n = 2^14, one 60-bit limb: forward NTT, pointwise mul, inverse NTT on a batch,
    checked by the convolution theorem against the oracle on one element and by round trip on all."""
    from oracle import cref
    n = 1 << 14
    g, o = _ring(n, [CFG2_Q60]), cref.Ring(n, [CFG2_Q60])
    E = 32
    a, b, c = g.alloc(E), g.alloc(E), g.alloc(E)
    a.fill_uniform(2026); b.fill_uniform(7)
    ha, hb = a.download(0, 1)[0], b.download(0, 1)[0]
    ca = a.checksum()
    a.crt(); b.crt(); c.mul(a, b, E); c.crtinv()
    want = o.crtinv(o.mul(o.crt(ha), o.crt(hb)))
    assert np.array_equal(c.download(0, 1)[0], want)
    a.crtinv()
    assert a.checksum() == ca
