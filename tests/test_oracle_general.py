"""CPU: the general-index oracle.  Three layers pin each other, since the reference holds nothing
that could (SURVEY 8c -- parity unpinned):
  1. oracle/model_gen.py: definitions (direct-evaluation CRT over the whole index, schoolbook ring product reduced to the
     powerful basis) -- checked for internal consistency: Kronecker == whole-index evaluation, crt is a ring homomorphism,
     mulG == product with g, divG inverts it and fails exactly when it must, and a SymmSHE round trip DECRYPTS CORRECTLY
     through mulG / Dec-basis lifting / divG / twace (the only semantic check available, like Arithmetic.hs's PASS);
  2. oracle/lol_tensor_gen.c (sparse decompositions, as lol-cpp) == the model, on small indices and, by whole-index
     direct evaluation, on the reference's real ciphertext indices H0' .. H5' (examples/Common.hs:49-54);
  3. the committed fixtures tests/golden/general_*.json (model-generated) == the C restatement.
"""
import random

import numpy as np
import pytest

from helpers import load_golden, oracle_full_mul_general, primes_1_mod, to_aos
from oracle import model_gen as G

RLWR_QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
H_PRIME = [11648, 29120, 43680, 54600, 27300, 20475]
SMALL = [4, 16, 3, 9, 27, 12, 28, 45, 36, 91, 63, 225, 100, 455, 6, 125]


def lm(x):
    return np.asarray(x).T.tolist()


@pytest.mark.parametrize("m", [4, 8, 3, 9, 12, 28, 45, 36, 91, 63, 225])
def test_model_is_self_consistent(m):
    rng = random.Random(m)
    idx = G.Index(m)
    q = primes_1_mod(m, 1, 1 << 20)[0]
    a = [rng.randrange(q) for _ in range(idx.n)]
    b = [rng.randrange(q) for _ in range(idx.n)]
    ca, cb = G.crt_def(a, idx, q), G.crt_def(b, idx, q)
    assert ca == G.crt_kron(a, idx, q)                                    # Kronecker of the per-axis definitions
    assert G.crtinv_def(ca, idx, q) == a
    assert G.crt_def(G.ring_mul_def(a, b, idx, q), idx, q) == [x * y % q for x, y in zip(ca, cb)]      # ring homomorphism
    assert G.crt_def(G.mulg_pow_def(a, idx, q), idx, q) == [x * y % q for x, y in zip(G.g_crt(idx, q), ca)]
    assert G.divg_pow_def(G.mulg_pow_def(a, idx, q), idx, q) == a
    assert G.divg_dec_def(G.mulg_dec_def(a, idx, q), idx, q) == a
    assert G.linv_def(G.l_def(a, idx, q), idx, q) == a
    z = [rng.randrange(-50, 50) for _ in range(idx.n)]
    gz = G.mulg_pow_def(z, idx, None)
    assert G.divg_pow_def(gz, idx, None) == z
    if G.odd_rad(idx) > 1:
        gz[0] += 1
        assert G.divg_pow_def(gz, idx, None) is None                      # Lol's Nothing


def test_two_power_slot_order_is_the_round_one_rule():
    from oracle import model as M
    idx, q = G.Index(32), primes_1_mod(32, 1, 1 << 20)[0]
    a = [random.Random(1).randrange(q) for _ in range(16)]
    assert G.crt_def(a, idx, q) == M.crt_def(a, q)


@pytest.mark.parametrize("m,mp,p", [(4, 28, 8), (4, 36, 7), (3, 45, 4), (8, 8, 16), (7, 91, 8)])
def test_symmshe_round_trip_decrypts_to_the_plaintext_product(m, mp, p):
    """encrypt -> (*) with mulG -> modSwitch up -> keySwitchQuadCirc -> modSwitch down (c0 on Dec, c1 on Pow) -> decrypt
    (liftDec, divG^k, twace) == product of the plaintexts; then mulPublic / addPublic / modSwitchPT."""
    rng = random.Random(m * 1000 + mp)
    small, big = G.Index(m), G.Index(mp)
    qs = primes_1_mod(mp, 3, 1 << 29)
    sk = G.g_gen_sk(big, rng)
    pa = [rng.randrange(p) for _ in range(small.n)]
    pb = [rng.randrange(p) for _ in range(small.n)]
    ca, cb = G.g_encrypt(sk, pa, small, big, p, qs[1:], rng), G.g_encrypt(sk, pb, small, big, p, qs[1:], rng)
    assert G.g_decrypt(sk, ca) == pa and G.g_decrypt(sk, cb) == pb
    want = G.ring_mul_def(pa, pb, small, p)
    prod = G.g_ct_mul(ca, cb)
    assert prod.k == 1 and G.g_decrypt(sk, prod) == want
    hint = G.g_ks_hint(sk, big, qs, rng)
    full = G.g_mod_switch_down(G.g_key_switch(hint, G.g_mod_switch_up(prod, qs[:1])), 2)
    assert full.qs == qs[2:] and len(full.c) == 2
    assert G.g_decrypt(sk, full) == want
    pub = [rng.randrange(p) for _ in range(small.n)]
    assert G.g_decrypt(sk, G.g_mul_public(pub, full)) == G.ring_mul_def(want, pub, small, p)
    assert G.g_decrypt(sk, G.g_add_public(pub, full)) == [(x + y) % p for x, y in zip(want, pub)]
    if p % 2 == 0:          # div2_: modSwitchPT of an encryption of 2x decrypts to x mod p/2
        x = [rng.randrange(p // 2) for _ in range(small.n)]
        c2x = G.g_encrypt(sk, [2 * v for v in x], small, big, p, qs, rng)
        assert G.g_decrypt(sk, G.g_mod_switch_pt(c2x, p // 2)) == x


@pytest.mark.parametrize("m", SMALL)
def test_c_restatement_matches_the_model(oracle_lib, m):
    rng = random.Random(m + 5)
    idx = G.Index(m)
    qs = primes_1_mod(m, 2, 1 << 28)
    R = oracle_lib.GenRing(m, qs)
    assert R.n == idx.n and R.has_crt
    a = [[rng.randrange(q) for _ in range(idx.n)] for q in qs]
    A = to_aos(a)
    assert lm(R.crt(A)) == [G.crt_def(al, idx, q) for al, q in zip(a, qs)]
    assert np.array_equal(R.crtinv(R.crt(A)), A)
    for cname, fn in (("l", G.l_def), ("linv", G.linv_def), ("mulg_pow", G.mulg_pow_def), ("mulg_dec", G.mulg_dec_def),
                      ("divg_pow", G.divg_pow_def), ("divg_dec", G.divg_dec_def)):
        assert lm(getattr(R, cname)(A)) == [fn(al, idx, q) for al, q in zip(a, qs)], cname
    assert lm(R.mulg_crt(A)) == [[x * y % q for x, y in zip(G.g_crt(idx, q), al)] for al, q in zip(a, qs)]
    assert np.array_equal(R.divg_crt(R.mulg_crt(A)), A)
    Z = oracle_lib.GenRing(m, [0])
    z = [rng.randrange(-1000, 1000) for _ in range(idx.n)]
    ZA = np.array(z, dtype=np.int64).reshape(-1, 1)
    assert Z.mulg_pow(ZA)[:, 0].tolist() == G.mulg_pow_def(z, idx, None)
    assert Z.mulg_dec(ZA)[:, 0].tolist() == G.mulg_dec_def(z, idx, None)
    assert np.array_equal(Z.divg_pow(Z.mulg_pow(ZA)), ZA) and np.array_equal(Z.divg_dec(Z.mulg_dec(ZA)), ZA)
    if G.odd_rad(idx) > 1:
        bad = Z.mulg_pow(ZA)
        bad[0, 0] += 1
        assert Z.divg_pow(bad) is None
    # a modulus that shares a factor with the radical: divG is Nothing whatever the input (lol-cpp)
    if G.odd_rad(idx) > 1:
        p0 = [p for p, _ in idx.pps if p != 2][0]
        P = oracle_lib.GenRing(m, [p0 * 5])
        assert not P.has_crt
        xp = np.array([rng.randrange(p0 * 5) for _ in range(idx.n)], dtype=np.int64).reshape(-1, 1)
        assert P.divg_pow(xp) is None and P.divg_dec(xp) is None
        assert P.mulg_pow(xp)[:, 0].tolist() == G.mulg_pow_def(xp[:, 0].tolist(), idx, p0 * 5)


@pytest.mark.parametrize("m", H_PRIME)
def test_c_restatement_at_the_references_indices_by_direct_evaluation(oracle_lib, m):
    """One limb per index, full size: crt against the whole-index direct evaluation, and the homomorphism / mulG
    properties on all six HomomRLWR moduli (examples/HomomRLWR.hs:37-43; all are 1 mod every H_i')."""
    idx = G.Index(m)
    assert all((q - 1) % m == 0 for q in RLWR_QS)
    R = oracle_lib.GenRing(m, RLWR_QS)
    x, y = R.fill_uniform(m, 0), R.fill_uniform(m, 1)
    j = H_PRIME.index(m)
    cx = R.crt(x)
    assert cx[:, j].tolist() == G.crt_def(x[:, j].tolist(), idx, RLWR_QS[j])
    assert np.array_equal(R.crtinv(cx), x)
    assert np.array_equal(R.crt(R.mulg_pow(x)), R.mulg_crt(cx))
    assert np.array_equal(R.divg_pow(R.mulg_pow(x)), x) and np.array_equal(R.divg_dec(R.mulg_dec(x)), x)
    assert np.array_equal(R.l(R.linv(x)), x)
    assert R.mulg_crt(np.ones_like(x))[:, j].tolist() == G.g_crt(idx, RLWR_QS[j])


def test_fixtures_tensor(oracle_lib):
    for rec in load_golden("general_tensor_small.json"):
        R = oracle_lib.GenRing(rec["m"], rec["qs"])
        A = to_aos(rec["a"])
        for name in ("crt", "l", "linv", "mulg_pow", "mulg_dec", "divg_pow", "divg_dec"):
            assert lm(getattr(R, name)(A)) == rec[name], (rec["m"], name)
        assert lm(R.mulg_crt(np.ones_like(A))) == rec["g_crt"]
        Z = oracle_lib.GenRing(rec["m"], [0])
        z = np.array(rec["z"], dtype=np.int64).reshape(-1, 1)
        assert Z.mulg_pow(z)[:, 0].tolist() == rec["z_mulg_pow"] and Z.mulg_dec(z)[:, 0].tolist() == rec["z_mulg_dec"]


def test_fixtures_mul(oracle_lib):
    """keySwitchQuadCirc(hint, a*b) and PT2CT's whole mul_ of the model-generated SymmSHE instances, recomputed by the C
    restatement (CRT-basis entry points, results brought back to the Pow basis)."""
    for rec in load_golden("general_mul_small.json"):
        mp, qs = rec["mp"], rec["qs"]
        R = oracle_lib.GenRing(mp, qs)
        hint = []
        for h0, h1 in rec["hint"]:
            hint += [R.crt(to_aos(h0)), R.crt(to_aos(h1))]
        r = rec["relin"]
        a, b = [R.crt(to_aos(c)) for c in r["a"]], [R.crt(to_aos(c)) for c in r["b"]]
        o0, o1 = R.ct_mul_relin(hint, a[0], a[1], b[0], b[1], r["s_pre"])
        assert lm(R.crtinv(o0)) == r["out"][0] and lm(R.crtinv(o1)) == r["out"][1], mp
        f = rec["full"]
        R2 = oracle_lib.GenRing(mp, qs[1:])
        a, b = [R2.crt(to_aos(c)) for c in f["a"]], [R2.crt(to_aos(c)) for c in f["b"]]
        w0, w1 = oracle_full_mul_general(oracle_lib, mp, qs, 2, 1, hint, a[0], a[1], b[0], b[1], f["s_pre"], pow_out=True)
        assert lm(w0) == f["out"][0] and lm(w1) == f["out"][1], mp


def test_numpy_base2_decomposition_equals_the_c_restatement(oracle_lib):
    """tests/helpers.decompose_base2 (used to compose BaseBGad 2 tunnels on general indices) against the C restatement's
    decompose_base2 on a two-power ring -- the decomposition is coefficient-wise, so the index does not enter."""
    from helpers import decompose_base2
    n, qs = 64, [537264001, 539884801, 12289]
    o = oracle_lib.Ring(n, qs)
    rng = np.random.default_rng(17)
    x = np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1)
    x[0] = [(q - 1) // 2 for q in qs]           # the extreme centred residues
    x[1] = [(q + 1) // 2 for q in qs]
    x[2] = 0
    want = o.decompose_base2(x)
    got = decompose_base2(x, qs)
    assert len(got) == len(want) == sum((q - 1).bit_length() for q in qs)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
