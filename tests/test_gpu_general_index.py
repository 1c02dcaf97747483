"""GPU: Tensor operations and the SymmSHE hot path on GENERAL cyclotomic indices, through the C ABI,
bit-compared with the C restatement (oracle/lol_tensor_gen.c, itself pinned to the by-definition model on the CPU in
tests/test_oracle_general.py).  Indices: small composites covering every pass kind, and the reference's real
ciphertext indices H0' .. H5' (examples/Common.hs:49-54) with the HomomRLWR moduli (examples/HomomRLWR.hs:37-43).
"""
import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi
from helpers import oracle_full_mul_general, primes_1_mod

pytestmark = pytest.mark.gpu

RLWR_QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
H_PRIME = [11648, 29120, 43680, 54600, 27300, 20475]
SMALL = [(m, primes_1_mod(m, 2, lo)) for m, lo in [(12, 0), (9, 1 << 20), (28, 0), (45, 1 << 30), (36, 100), (91, 1 << 29), (225, 1000),
                                                     (27, 1 << 28), (100, 0), (455, 1 << 30), (16, 0), (8, 1 << 30), (6, 0), (125, 1 << 20), (81, 50000)]]


def rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


@pytest.mark.parametrize("m,qs", SMALL + [(m, RLWR_QS) for m in H_PRIME])
def test_tensor_methods_match_the_oracle(oracle_lib, m, qs):
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    assert g.n == o.n
    rng = np.random.default_rng(m)
    x = rand_elems(rng, 3, g.n, qs)
    # host-buffer Tensor methods, one ring element each
    a = x[0]
    assert np.array_equal(g.crt(a), o.crt(a))
    assert np.array_equal(g.crtinv(a), o.crtinv(a))
    assert np.array_equal(g.crtinv(g.crt(a)), a)
    for name in ("l", "linv", "mulg_pow", "mulg_dec", "mulg_crt", "divg_crt"):
        assert np.array_equal(getattr(g, name)(a), getattr(o, name)(a)), name
    for name in ("divg_pow", "divg_dec"):
        want, got = getattr(o, name)(a), getattr(g, name)(a)
        assert (want is None) == (got is None), name
        if want is not None:
            assert np.array_equal(got, want), name
    assert np.array_equal(g.mul(g.crt(x[0]), g.crt(x[1])), o.mul(o.crt(x[0]), o.crt(x[1])))
    # batched device forms
    buf = g.upload(x)
    buf.crt()
    assert np.array_equal(buf.download(), np.stack([o.crt(e) for e in x]))
    buf.mulg(capi.ALCH_BASIS_CRT, 1, 2)
    want = np.stack([o.crt(x[0]), o.mulg_crt(o.crt(x[1])), o.mulg_crt(o.crt(x[2]))])
    assert np.array_equal(buf.download(), want)
    buf.crtinv()
    assert np.array_equal(buf.download(), np.stack([x[0], o.mulg_pow(x[1]), o.mulg_pow(x[2])]))
    assert buf.divg(capi.ALCH_BASIS_POW, 1, 2)
    assert np.array_equal(buf.download(), x)
    buf.linv(0, 2)
    assert np.array_equal(buf.download(), np.stack([o.linv(x[0]), o.linv(x[1]), x[2]]))
    buf.mulg(capi.ALCH_BASIS_DEC, 0, 1)
    buf.l(0, 2)
    assert np.array_equal(buf.download(), np.stack([o.l(o.mulg_dec(o.linv(x[0]))), x[1], x[2]]))


@pytest.mark.parametrize("m", [28, 45, 455, 20475])
def test_divg_over_the_integers_reports_not_divisible(oracle_lib, m):
    """Tensor t m Int64 (what decrypt lifts to): divG succeeds exactly on multiples of g."""
    g, o = A.Ring(m, [0], nocrt=True), oracle_lib.GenRing(m, [0])
    assert g.word_bytes == 8
    rng = np.random.default_rng(m)
    z = rng.integers(-10**6, 10**6, size=(g.n, 1), dtype=np.int64)
    for mul, div in (("mulg_pow", "divg_pow"), ("mulg_dec", "divg_dec")):
        gz = getattr(g, mul)(z)
        assert np.array_equal(gz, getattr(o, mul)(z)), mul
        assert np.array_equal(getattr(g, div)(gz), z), div
        bad = gz.copy()
        bad[g.n // 2, 0] += 1
        assert getattr(o, div)(bad) is None
        assert getattr(g, div)(bad) is None, div          # ALCH_NOT_DIVISIBLE
    assert np.array_equal(g.l(z), o.l(z)) and np.array_equal(g.linv(z), o.linv(z))
    with pytest.raises(A.AlchemyError) as e:
        g.crt(z)
    assert e.value.code == capi.ALCH_E_NO_CRT


def test_divg_in_a_plaintext_ring_whose_modulus_shares_a_factor_with_the_radical(oracle_lib):
    """Z_q with gcd(q, rad(m)) != 1: lol-cpp's divG returns Nothing whatever the input; a unit radical divides."""
    for q, ok in ((7, False), (8, True), (25, True), (65, False)):
        g, o = A.Ring(91 * 4, [q], nocrt=True), oracle_lib.GenRing(91 * 4, [q])
        x = np.random.default_rng(q).integers(0, q, size=(g.n, 1), dtype=np.int64)
        for name in ("mulg_pow", "mulg_dec", "l", "linv"):
            assert np.array_equal(getattr(g, name)(x), getattr(o, name)(x)), (q, name)
        for name in ("divg_pow", "divg_dec"):
            want, got = getattr(o, name)(x), getattr(g, name)(x)
            assert (want is not None) == ok and (got is not None) == ok, (q, name)
            if ok:
                assert np.array_equal(got, want), (q, name)


@pytest.mark.parametrize("m,L,batch", [(28, 2, 3), (45, 3, 2), (455, 3, 2), (20475, 4, 3), (54600, 5, 2), (11648, 6, 2)])
def test_ct_mul_relin_general_index(oracle_lib, m, L, batch):
    qs = RLWR_QS[:L] if m > 1000 else primes_1_mod(m, L, 1 << 29)
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    rng = np.random.default_rng(m + L)
    hint, a, b = rand_elems(rng, 2 * L, g.n, qs), rand_elems(rng, 2 * batch, g.n, qs), rand_elems(rng, 2 * batch, g.n, qs)
    s_pre = [int(rng.integers(1, q)) for q in qs]
    gh, ga, gb, gout = g.hint_load(hint), g.upload(a), g.upload(b), g.alloc(2 * batch)
    g.ct_mul_relin(gh, ga, gb, gout, batch, s_pre=s_pre)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = o.ct_mul_relin(list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], s_pre)
        assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), ct
    # Pow basis in and out
    apow = np.stack([o.crtinv(e) for e in a])
    bpow = np.stack([o.crtinv(e) for e in b])
    g.ct_mul_relin(gh, g.upload(apow), g.upload(bpow), gout, batch, s_pre=s_pre, flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)
    got2 = gout.download()
    assert np.array_equal(got2, np.stack([o.crtinv(e) for e in got]))


@pytest.mark.parametrize("m,l_in,l_h,l_out", [(28, 1, 2, 1), (455, 2, 3, 1), (54600, 4, 5, 3), (20475, 4, 5, 3), (20475, 3, 4, 2), (11648, 5, 6, 5)])
def test_ct_mul_full_general_index(oracle_lib, m, l_in, l_h, l_out):
    """PT2CT's mul_ with the limb counts of SURVEY 3.3 (x(1+x): 4 -> 5 -> 3) on the reference's rings: mulG in (*),
    modSwitch up, keySwitchQuadCirc, modSwitch down with c0 rescaled on the Dec basis and c1 on the Pow basis."""
    qs_h = (RLWR_QS if m > 1000 else primes_1_mod(m, l_h, 1 << 29))[:l_h]
    qs_h = list(reversed(qs_h))                 # the hint's extra limb goes in front (Noise.hs:82-89)
    batch = 2
    rh, rin, rout = A.Ring(m, qs_h), A.Ring(m, qs_h[l_h - l_in:]), A.Ring(m, qs_h[l_h - l_out:])
    rng = np.random.default_rng(m + l_h)
    hint = rand_elems(rng, 2 * l_h, rh.n, qs_h)
    a, b = rand_elems(rng, 2 * batch, rh.n, qs_h[l_h - l_in:]), rand_elems(rng, 2 * batch, rh.n, qs_h[l_h - l_in:])
    s_pre = [int(rng.integers(1, q)) for q in qs_h[l_h - l_in:]]
    gh, ga, gb, gout = rh.hint_load(hint), rin.upload(a), rin.upload(b), rout.alloc(2 * batch)
    want = {po: [oracle_full_mul_general(oracle_lib, m, qs_h, l_in, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1],
                                         s_pre, po) for ct in range(batch)] for po in (False, True)}
    # rs_lin = 1 (default): the closing modSwitch keeps the surviving limbs in the CRT basis; 0: every limb through Pow / Dec
    for pow_out, lin in ((False, 1), (False, 0), (True, 1)):
        rh.set_option("rs_lin", lin)
        capi.ct_mul_full(gh, ga, gb, gout, batch, s_pre=s_pre, flags=capi.ALCH_POW_OUT if pow_out else 0)
        got = gout.download()
        for ct in range(batch):
            w0, w1 = want[pow_out][ct]
            assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), (ct, pow_out, lin)


@pytest.mark.parametrize("m,L,drop", [(20475, 5, 1), (54600, 6, 1), (11648, 5, 2), (455, 4, 3), (1 << 12, 4, 2), (1 << 16, 3, 1)])
def test_ct_mod_switch_down_and_up(oracle_lib, m, L, drop):
    """SymmSHE modSwitch on linear ciphertexts: down = rescaleDec on c0 / rescalePow on c1 one limb at a time, up = times the
    added moduli; general and two-power indices, CRT basis in and out."""
    big = m >= 1000
    if m & (m - 1) == 0:
        qs = primes_1_mod(m, L, 1 << 30)
    else:
        qs = list(reversed(RLWR_QS[:L])) if big else primes_1_mod(m, L, 1 << 29)
    gen = m & (m - 1) != 0
    Rb, Rs = A.Ring(m, qs), A.Ring(m, qs[drop:])
    ob = oracle_lib.GenRing(m, qs) if gen else oracle_lib.Ring(m // 2, qs)
    mk = (lambda q: oracle_lib.GenRing(m, q)) if gen else (lambda q: oracle_lib.Ring(m // 2, q))
    batch = 2
    rng = np.random.default_rng(m + L)
    x = rand_elems(rng, 2 * batch, Rb.n, qs)
    gx, gy = Rb.upload(x), Rs.alloc(2 * batch)
    want = []
    for e in range(2 * batch):
        cur = ob.crtinv(x[e])
        if e % 2 == 0 and gen:
            cur = ob.linv(cur)
        for u in range(drop):
            cur = mk(qs[u:]).rescale_drop0(cur)
        osm = mk(qs[drop:])
        if e % 2 == 0 and gen:
            cur = osm.l(cur)
        want.append(osm.crt(cur))
    for lin in ((1, 0) if gen else (1,)):      # general index: surviving limbs kept in the CRT basis / the limb-at-a-time form
        Rb.set_option("rs_lin", lin)
        capi.ct_mod_switch(gx, gy, batch)
        got = gy.download()
        for e in range(2 * batch):
            assert np.array_equal(got[e], want[e]), (e, lin)
        if gen:                                 # Pow-basis output (the hand-over between two tunnel hops)
            capi.ct_mod_switch(gx, gy, batch, flags=capi.ALCH_POW_OUT)
            gotp = gy.download()
            for e in range(2 * batch):
                assert np.array_equal(gotp[e], mk(qs[drop:]).crtinv(want[e])), (e, lin, "pow")
            capi.ct_mod_switch(gx, gy, batch)   # back to the CRT-basis result for the checks below
    assert np.array_equal(gx.download(), x)                   # input untouched
    # and up again: (0, .., 0, q_a x)
    gz = Rb.alloc(2 * batch)
    capi.ct_mod_switch(gy, gz, batch)
    up = gz.download()
    mult = 1
    for q in qs[:drop]:
        mult *= q
    for e in range(2 * batch):
        assert not up[e][:, :drop].any()
        assert np.array_equal(up[e][:, drop:], mk(qs[drop:]).scale(got[e], [mult % q for q in qs[drop:]])), e


@pytest.mark.parametrize("fused,nt", [(0, 0), (1, 0), (0, 128), (0, 512), (1, 512), (1, 256)])
def test_general_key_switch_launch_options(oracle_lib, fused, nt):
    """gen_fused (1 = the two fused launches, the default since round 3; 0 = the composed path) and gen_nt (threads per workgroup of
    the transform kernels) are launch-structure options: the same bits either way, for keySwitchQuadCirc(a*b) and for PT2CT's
    whole mul_ (4 -> 5 -> 3 limbs) on H5'."""
    m, qs, batch = 20475, RLWR_QS[:4], 3
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    g.set_option("gen_fused", fused); g.set_option("gen_nt", nt)
    rng = np.random.default_rng(77)
    hint, a, b = rand_elems(rng, 8, g.n, qs), rand_elems(rng, 2 * batch, g.n, qs), rand_elems(rng, 2 * batch, g.n, qs)
    gh, ga, gb, gout = g.hint_load(hint), g.upload(a), g.upload(b), g.alloc(2 * batch)
    g.ct_mul_relin(gh, ga, gb, gout, batch)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = o.ct_mul_relin(list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1])
        assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), ct
    from helpers import oracle_full_mul_general
    qh = list(reversed(RLWR_QS[:5]))
    rh, rin, rout = A.Ring(m, qh), A.Ring(m, qh[1:]), A.Ring(m, qh[2:])
    for r in (rh, rin, rout):
        r.set_option("gen_fused", fused); r.set_option("gen_nt", nt)
    hint5 = rand_elems(rng, 10, rh.n, qh)
    a4, b4 = rand_elems(rng, 2 * batch, rh.n, qh[1:]), rand_elems(rng, 2 * batch, rh.n, qh[1:])
    s_pre = [int(rng.integers(1, q)) for q in qh[1:]]
    out = rout.alloc(2 * batch)
    capi.ct_mul_full(rh.hint_load(hint5), rin.upload(a4), rin.upload(b4), out, batch, s_pre=s_pre)
    got = out.download()
    for ct in range(batch):
        w0, w1 = oracle_full_mul_general(oracle_lib, m, qh, 4, 3, list(hint5), a4[2 * ct], a4[2 * ct + 1], b4[2 * ct], b4[2 * ct + 1], s_pre)
        assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), ct


def test_public_ops_general_index(oracle_lib):
    """mulPublic / addPublic (Eval.hs:131-132) on H5': every ciphertext component times one public element; one public
    element added to every c0."""
    m, qs, batch = 20475, RLWR_QS[:3], 3
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    rng = np.random.default_rng(9)
    cts, pub = rand_elems(rng, 2 * batch, g.n, qs), rand_elems(rng, 2, g.n, qs)
    gc, gp, gd = g.upload(cts), g.upload(pub), g.alloc(2 * batch)
    gd.mul_public(gc, gp, 1, 2 * batch)
    assert np.array_equal(gd.download(), np.stack([o.mul(e, pub[1]) for e in cts]))
    gc.add_public(gp, 0, batch)
    want = cts.copy()
    for ct in range(batch):
        want[2 * ct] = o.add(cts[2 * ct], pub[0])
    assert np.array_equal(gc.download(), want)
    # addPublic with its toLSD scalar as one pass (alch_ct_add_public) == scale, then add_public
    sc = [int(rng.integers(1, q)) for q in qs]
    ge = g.alloc(2 * batch)
    ge.ct_add_public(gc, batch, sc, gp, 1)
    ref = g.alloc(2 * batch)
    ref.scale(gc, 2 * batch, sc); ref.add_public(gp, 1, batch)
    assert np.array_equal(ge.download(), ref.download())
    ge.ct_add_public(gc, batch, None, gp, 0)
    gc.add_public(gp, 0, batch)
    assert np.array_equal(ge.download(), gc.download())


def test_sixty_bit_moduli_general_index(oracle_lib):
    m = 455 * 4
    qs = []
    q = (1 << 59) // m * m + 1
    from oracle.model import is_prime
    while len(qs) < 2:
        if is_prime(q):
            qs.append(q)
        q += m
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    assert g.word_bytes == 8
    x = rand_elems(np.random.default_rng(3), 2, g.n, qs)
    buf = g.upload(x)
    buf.crt()
    assert np.array_equal(buf.download(), np.stack([o.crt(e) for e in x]))
    buf.crtinv()
    assert np.array_equal(buf.download(), x)
    assert np.array_equal(g.mulg_dec(x[0]), o.mulg_dec(x[0]))
    assert np.array_equal(g.divg_pow(x[1]), o.divg_pow(x[1]))


def test_model_fixtures_on_the_gpu():
    """The committed model-generated vectors (tests/golden/general_*.json) straight through the C ABI."""
    from helpers import load_golden, to_aos
    lm = lambda x: np.asarray(x).T.tolist()
    for rec in load_golden("general_tensor_small.json"):
        g = A.Ring(rec["m"], rec["qs"])
        a = to_aos(rec["a"])
        for name in ("crt", "l", "linv", "mulg_pow", "mulg_dec", "divg_pow", "divg_dec"):
            assert lm(getattr(g, name)(a)) == rec[name], (rec["m"], name)
        assert lm(g.mulg_crt(np.ones_like(a))) == rec["g_crt"]
        z = A.Ring(rec["m"], [0], nocrt=True)
        zz = np.array(rec["z"], dtype=np.int64).reshape(-1, 1)
        assert z.mulg_pow(zz)[:, 0].tolist() == rec["z_mulg_pow"] and z.mulg_dec(zz)[:, 0].tolist() == rec["z_mulg_dec"]
    for rec in load_golden("general_mul_small.json"):
        mp, qs = rec["mp"], rec["qs"]
        g = A.Ring(mp, qs)
        hint = g.upload(np.stack([to_aos(h) for pair in rec["hint"] for h in pair]))
        hint.crt()
        gh = g.hint_from_buf(hint)
        r = rec["relin"]
        ga, gb, gout = g.upload(np.stack([to_aos(c) for c in r["a"]])), g.upload(np.stack([to_aos(c) for c in r["b"]])), g.alloc(2)
        g.ct_mul_relin(gh, ga, gb, gout, 1, s_pre=r["s_pre"], flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)
        got = gout.download()
        assert lm(got[0]) == r["out"][0] and lm(got[1]) == r["out"][1], mp
        f = rec["full"]
        rin, rout = A.Ring(mp, qs[1:]), A.Ring(mp, qs[2:])
        ga, gb, gout = rin.upload(np.stack([to_aos(c) for c in f["a"]])), rin.upload(np.stack([to_aos(c) for c in f["b"]])), rout.alloc(2)
        ga.crt(); gb.crt()
        capi.ct_mul_full(gh, ga, gb, gout, 1, s_pre=f["s_pre"], flags=capi.ALCH_POW_OUT)
        got = gout.download()
        assert lm(got[0]) == f["out"][0] and lm(got[1]) == f["out"][1], mp
