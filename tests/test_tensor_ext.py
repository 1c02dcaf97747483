"""The Tensor methods between two cyclotomic indices m | m' (embedPow / embedDec / twacePowDec / coeffs,
crtExtFuncs = (twaceCRT, embedCRT), crtSetDec) -- the CRT-order-dependent half of `instance Tensor`, which must follow the same
slot rule as crt / crtInv (include/alchemy_hip.h) for the instance to be sound.

CPU part: the by-definition model of oracle/model_gen.py is checked against its own definitions (embedCRT / twaceCRT against
crt of embedPow / twacePowDec by direct evaluation; the CRT set against its defining property over GF(p^d)), and the library's
host-only tables (alch_ext_table, alch_crt_set_dec: no GPU needed) against the model.
GPU part: every entry point through the C ABI against the model, on small index pairs and on the reference's plaintext /
ciphertext pairs (H_k, H_k') of examples/Common.hs:38-54 with the HomomRLWR moduli (examples/HomomRLWR.hs:37-43).

PARITY UNPINNED against Lol (as everything on this path): the slot order and the order of the CRT set are this library's."""
import math
import random

import numpy as np
import pytest

from alchemy_amd import capi
from helpers import to_aos
from oracle import model_gen as G
from oracle.model import is_prime

PAIRS = [(4, 12), (3, 9), (5, 15), (8, 40), (12, 60), (7, 21), (1, 7), (9, 45), (4, 8), (16, 48), (32, 96), (1, 1), (15, 15)]
H = [128, 448, 2912, 3640, 5460, 4095]                     # plaintext indices H0 .. H5
HP = [11648, 29120, 43680, 54600, 27300, 20475]            # ciphertext indices H0' .. H5'
QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]


def prime_1_mod(m, start=1 << 20):
    q = start - start % m + 1
    while not is_prime(q):
        q += m
    return q


# ---------------------------------------------------------------------------------------------------------------
# CPU: the model against its definitions
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,mb", PAIRS[:10])
def test_model_crt_ext_funcs_by_definition(m, mb):
    s, b = G.Index(m), G.Index(mb)
    q = prime_1_mod(mb)
    rng = random.Random(m * 1000 + mb)
    x = [rng.randrange(q) for _ in range(s.n)]
    y = [rng.randrange(q) for _ in range(b.n)]
    assert G.crt_def(G.embed_pow(x, s, b), b, q) == G.embed_crt_def(G.crt_def(x, s, q), s, b)
    assert G.crt_def(G.twace_pow_dec(y, s, b), s, q) == G.twace_crt_def(G.crt_def(y, b, q), s, b, q)
    # twacePowDec reads the same positions on the decoding basis; twace . embed = id; embedDec is the embedding on Dec coefficients
    assert G.linv_def(G.twace_pow_dec(y, s, b), s, q) == G.twace_pow_dec(G.linv_def(y, b, q), s, b)
    assert G.twace_pow_dec(G.embed_pow(x, s, b), s, b) == x
    assert G.l_def(G.embed_dec_def(x, s, b, q), b, q) == G.embed_pow(G.l_def(x, s, q), s, b)
    # coeffs: x = sum_i embed(c_i) * (relative powerful basis element i)
    cs = G.coeffs(y, s, b)
    rows = G.coeffs_indices(s, b)
    acc = [0] * b.n
    for c, row in zip(cs, rows):
        unit = [0] * b.n
        unit[row[0]] = 1
        acc = [(u + v) % q for u, v in zip(acc, G.ring_mul_def(G.embed_pow(c, s, b), unit, b, q))]
    assert acc == y


@pytest.mark.parametrize("m,mb,p", [(1, 7, 2), (1, 15, 2), (3, 15, 2), (3, 9, 2), (7, 21, 2), (1, 21, 2), (1, 13, 3), (5, 35, 3),
                                    (9, 63, 2), (7, 91, 2), (4, 12, 5), (1, 8, 3), (1, 1, 2), (5, 5, 2)])
def test_model_crt_set_is_the_set_of_idempotents_it_claims(m, mb, p):
    s, b = G.Index(m), G.Index(mb)
    cs = G.crt_set_dec_def(s, b, p)
    F = G.GF(p, G.mult_order(p, mb))
    w = F.root_of_unity(mb)
    Is = G.crt_set_cosets(s, b, p)
    units = [u for u in range(1, mb + 1) if math.gcd(u, mb) == 1] if mb > 1 else [0]
    assert sorted(u for I in Is for u in I) == units                             # a partition of Z_m'^*
    small_cosets = len({min(u * p ** t % m for t in range(G.mult_order(p, m))) if m > 1 else 0 for u in units})
    assert len(cs) * small_cosets == len(units) // G.mult_order(p, mb)           # #primes above / #primes below
    for c, I in zip(cs, Is):
        powc = G.l_def(c, b, p)
        for u in units:
            assert G.eval_mod_p(powc, b, F, w, u) == (F.one if u in I else F.zero)
        assert G.ring_mul_def(powc, powc, b, p) == powc                          # idempotent
        below = {}
        for u in I:                                                              # one <p>-coset above every <p>-coset below
            key = min(u * p ** t % m for t in range(G.mult_order(p, m))) if m > 1 else 0
            below[key] = below.get(key, 0) + 1
        assert len(below) == small_cosets and set(below.values()) == {G.mult_order(p, mb)}
    tot = [sum(c[j] for c in cs) % p for j in range(b.n)]
    assert G.l_def(tot, b, p) == [1] + [0] * (b.n - 1)                            # the set sums to 1


# ---------------------------------------------------------------------------------------------------------------
# CPU: the library's host-only tables against the model
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,mb", PAIRS + list(zip(H, HP)) + [(HP[0] // 91 * 1, HP[0]), (64, 448), (1365, 4095), (2275, 20475)])
def test_ext_tables_follow_the_model(m, mb):
    if mb % m:
        pytest.skip("not an extension")
    s, b = G.Index(m), G.Index(mb)
    assert capi.ext_table(m, mb, capi.ALCH_EXT_POW_POS).tolist() == G.embed_indices(s, b)
    rows = G.coeffs_indices(s, b)
    assert capi.ext_table(m, mb, capi.ALCH_EXT_COEFFS).reshape(len(rows), s.n).tolist() == rows
    su = G.slot_of_unit(s)
    assert capi.ext_table(m, mb, capi.ALCH_EXT_CRT_SLOT).tolist() == [su[b.slot_unit(t) % m] for t in range(b.n)]


def test_ext_table_rejects_non_extensions():
    with pytest.raises(capi.AlchemyError):
        capi.ext_table(12, 40, capi.ALCH_EXT_POW_POS)


@pytest.mark.parametrize("m,mb,p", [(1, 7, 2), (3, 15, 2), (1, 13, 3), (5, 35, 3), (9, 63, 2), (7, 91, 2), (4, 12, 5), (1, 8, 3),
                                    (1, 1, 2), (7, 7, 2)])
def test_crt_set_dec_equals_the_model(m, mb, p):
    b = G.Index(mb)
    got = capi.crt_set_dec(m, mb, p, b.n)
    assert got.tolist() == G.crt_set_dec_def(G.Index(m), b, p)


@pytest.mark.parametrize("k", range(5))
def test_crt_set_of_the_reference_hops_has_the_defining_property(k):
    """decToCRT @H_k (examples/Common.hs:65-75): crtSet of S / E mod 2 for E = H_k cap H_{k+1}, S = H_{k+1}, taken on the odd parts
    (Cyc crtSet works on PFree 2 of both indices).  Full size: checked through the defining property, on a sample of the units."""
    e, s = math.gcd(H[k], H[k + 1]), H[k + 1]
    while e % 2 == 0:
        e //= 2
    while s % 2 == 0:
        s //= 2
    b = G.Index(s)
    got = capi.crt_set_dec(e, s, 2, b.n)
    dim = G.totient(H[k]) // G.totient(math.gcd(H[k], H[k + 1]))
    assert got.shape[0] >= dim                                                    # `take dim crts` must succeed
    F = G.GF(2, G.mult_order(2, s))
    w = F.root_of_unity(s)
    Is = G.crt_set_cosets(G.Index(e), b, 2)
    rng = random.Random(k)
    units = [u for u in range(1, s + 1) if math.gcd(u, s) == 1]
    tot = np.zeros(b.n, dtype=np.int64)
    for c, I in zip(got.tolist(), Is):
        powc = G.l_def(c, b, 2)
        Iset = set(I)
        for u in rng.sample(units, 6) + rng.sample(I, 3):
            assert G.eval_mod_p(powc, b, F, w, u) == (F.one if u in Iset else F.zero)
        tot += np.array(c)
    assert G.l_def((tot % 2).tolist(), b, 2) == [1] + [0] * (b.n - 1)


# ---------------------------------------------------------------------------------------------------------------
# GPU: the entry points against the model
# ---------------------------------------------------------------------------------------------------------------
def rand_elem(rng, n, qs):
    return np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1)


def limbs(arr):
    return np.asarray(arr).T.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("m,mb", PAIRS)
def test_gpu_ext_methods_small(m, mb):
    import alchemy_amd as A
    s, b = G.Index(m), G.Index(mb)
    qs = [prime_1_mod(mb, 1 << 29), prime_1_mod(mb, 1 << 30)]
    rs, rb = A.Ring(m, qs), A.Ring(mb, qs)
    rng = np.random.default_rng(m * 97 + mb)
    x, y = rand_elem(rng, s.n, qs), rand_elem(rng, b.n, qs)
    xl, yl = limbs(x), limbs(y)
    assert limbs(rs.embed_pow(rb, x)) == [G.embed_pow(v, s, b) for v in xl]
    assert limbs(rs.embed_dec(rb, x)) == [G.embed_dec_def(v, s, b, q) for v, q in zip(xl, qs)]
    assert limbs(rs.embed_crt(rb, x)) == [G.embed_crt_def(v, s, b) for v in xl]
    assert limbs(rs.twace_pow_dec(rb, y)) == [G.twace_pow_dec(v, s, b) for v in yl]
    assert limbs(rs.twace_crt(rb, y)) == [G.twace_crt_def(v, s, b, q) for v, q in zip(yl, qs)]
    cs = rs.coeffs(rb, y)
    for j, q in enumerate(qs):
        assert [c[:, j].tolist() for c in cs] == G.coeffs(yl[j], s, b)
    # the identities that make the instance sound: crt . embedPow = embedCRT . crt,  crt . twacePowDec = twaceCRT . crt
    assert np.array_equal(rb.crt(rs.embed_pow(rb, x)), rs.embed_crt(rb, rs.crt(x)))
    assert np.array_equal(rs.crt(rs.twace_pow_dec(rb, y)), rs.twace_crt(rb, rb.crt(y)))
    # and against crt by definition
    assert limbs(rs.embed_crt(rb, rs.crt(x))) == [G.crt_def(G.embed_pow(v, s, b), b, q) for v, q in zip(xl, qs)]
    assert limbs(rs.twace_crt(rb, rb.crt(y))) == [G.crt_def(G.twace_pow_dec(v, s, b), s, q) for v, q in zip(yl, qs)]


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(6))
def test_gpu_crt_ext_funcs_on_the_reference_index_pairs(k, oracle_lib):
    """(H_k, H_k') with three HomomRLWR moduli, batched device forms.  Checker: the C restatement's crt (pinned to the
    by-definition evaluation at these sizes in tests/test_oracle_general.py) and the model's slot-unit definition of embedCRT /
    twaceCRT -- no O(n^2) evaluation needed here."""
    import alchemy_amd as A
    m, mb = H[k], HP[k]
    qs = QS[:3]
    s, b = G.Index(m), G.Index(mb)
    rs, rb = A.Ring(m, qs), A.Ring(mb, qs)
    os_, ob = oracle_lib.GenRing(m, qs), oracle_lib.GenRing(mb, qs)
    B = 3
    rng = np.random.default_rng(k)
    xs = np.stack([rand_elem(rng, s.n, qs) for _ in range(B)])
    ys = np.stack([rand_elem(rng, b.n, qs) for _ in range(B)])
    bx, by = rs.upload(xs), rb.upload(ys)
    big, small = rb.alloc(B), rs.alloc(B)
    # embedPow / embedCRT
    big.embed_from(bx, B, capi.ALCH_BASIS_POW)
    emb = big.download()
    pos = G.embed_indices(s, b)
    want = np.zeros_like(emb)
    want[:, pos, :] = xs
    assert np.array_equal(emb, want)
    bx.crt()
    big.embed_from(bx, B, capi.ALCH_BASIS_CRT)
    got = big.download()
    slot = capi.ext_table(m, mb, capi.ALCH_EXT_CRT_SLOT)
    for e in range(B):
        xc = os_.crt(xs[e])
        assert np.array_equal(got[e], xc[slot, :])                               # the model's definition of embedCRT
        assert np.array_equal(got[e], ob.crt(emb[e]))                            # == crt . embedPow
    # embedDec
    bx.upload(xs)
    big.embed_from(bx, B, capi.ALCH_BASIS_DEC)
    got = big.download()
    for e in range(B):
        assert np.array_equal(ob.l(got[e]), np.asarray(want[e]) * 0 + ob_embed(os_.l(xs[e]), pos, b.n))
    # twacePowDec / twaceCRT
    small.twace_from(by, B, capi.ALCH_BASIS_POW)
    tw = small.download()
    assert np.array_equal(tw, ys[:, pos, :])
    by.crt()
    small.twace_from(by, B, capi.ALCH_BASIS_CRT)
    got = small.download()
    for e in range(B):
        assert np.array_equal(got[e], os_.crt(tw[e]))                            # == crt . twacePowDec
        yc = ob.crt(ys[e])
        for j, q in enumerate(qs):
            assert got[e][:, j].tolist() == G.twace_crt_def(yc[:, j].tolist(), s, b, q)     # the model's definition
    # coeffs
    by.upload(ys)
    rows = np.array(G.coeffs_indices(s, b))
    cbuf = rs.alloc(B * rows.shape[0])
    cbuf.coeffs_from(by, B)
    got = cbuf.download().reshape(B, rows.shape[0], s.n, len(qs))
    for e in range(B):
        assert np.array_equal(got[e], ys[e][rows, :])


def ob_embed(x_pow, pos, n_big):
    out = np.zeros((n_big, x_pow.shape[1]), dtype=np.int64)
    out[pos, :] = x_pow
    return out


@pytest.mark.gpu
def test_gpu_ext_methods_on_plaintext_rings_without_crt():
    """Plaintext rings Z_{2^k} (examples/Common.hs:32) have no CRT basis: the Pow / Dec forms still serve them."""
    import alchemy_amd as A
    m, mb, p = 12, 60, 32
    s, b = G.Index(m), G.Index(mb)
    rs, rb = A.Ring(m, [p], nocrt=True), A.Ring(mb, [p], nocrt=True)
    rng = np.random.default_rng(5)
    x, y = rand_elem(rng, s.n, [p]), rand_elem(rng, b.n, [p])
    assert rs.embed_pow(rb, x)[:, 0].tolist() == G.embed_pow(x[:, 0].tolist(), s, b)
    assert rs.embed_dec(rb, x)[:, 0].tolist() == G.embed_dec_def(x[:, 0].tolist(), s, b, p)
    assert rs.twace_pow_dec(rb, y)[:, 0].tolist() == G.twace_pow_dec(y[:, 0].tolist(), s, b)
    with pytest.raises(capi.AlchemyError):
        rs.embed_crt(rb, x)


@pytest.mark.gpu
def test_gpu_ext_methods_reject_mismatched_rings():
    import alchemy_amd as A
    qs = [prime_1_mod(120, 1 << 29)]
    r12, r40, r60 = A.Ring(12, qs), A.Ring(40, qs), A.Ring(60, [prime_1_mod(120, 1 << 30)])
    x = np.zeros((r12.n, 1), dtype=np.int64)
    with pytest.raises(capi.AlchemyError):
        r12.embed_pow(r40, x)                   # 12 does not divide 40
    with pytest.raises(capi.AlchemyError):
        r12.embed_pow(r60, x)                   # different moduli
