"""CPU: the oracle (C restatement and the exact model) against the committed golden vectors, and the
property tests Lol's own suites are built from (crtInv . crt = id, convolution theorem, decompose
recomposes, key switch preserves decryption)."""
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from helpers import digest_limb_major, from_aos, hint_to_crt_aos, load_golden, to_aos
from oracle import model as M


def test_model_number_theory():
    assert M.is_prime(2147352577) and not M.is_prime(2147352579)
    assert M.smallest_generator(65537) == 3
    assert M.smallest_generator(12289) == 11
    q, n = 268440577, 256
    psi = M.root_2n(q, n)
    assert pow(psi, 2 * n, q) == 1 and pow(psi, n, q) == q - 1
    assert [M.centred(x, 7) for x in range(7)] == [0, 1, 2, 3, -3, -2, -1]
    with pytest.raises(AssertionError):
        M.root_2n(8392193, 1 << 15)          # q != 1 mod m: no CRT basis


def test_c_oracle_matches_golden_ntt(oracle_lib):
    for case in load_golden("ntt_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        ring = oracle_lib.Ring(n, qs)
        assert [ring.psi(j) for j in range(len(qs))] == case["psi"]
        assert [oracle_lib.smallest_generator(q) for q in qs] == case["generator"]
        a, b = to_aos(case["a"]), to_aos(case["b"])
        assert from_aos(ring.crt(a)) == case["crt_a"]
        assert from_aos(ring.crtinv(b)) == case["crtinv_b"]
        assert from_aos(ring.crtinv(ring.mul(ring.crt(a), ring.crt(b)))) == case["a_times_b"]
        assert from_aos(ring.add(a, b)) == case["a_plus_b"]
        assert np.array_equal(ring.crtinv(ring.crt(a)), a)


def test_c_oracle_matches_golden_decompose(oracle_lib):
    for case in load_golden("decompose.json")["cases"]:
        n, qs = case["n"], case["qs"]
        ring = oracle_lib.Ring(n, qs)
        c = to_aos(case["c"])
        assert [from_aos(d) for d in ring.decompose_triv(c)] == case["triv_reduced"]
        b2 = ring.decompose_base2(c)
        assert len(b2) == case["base2_count"]
        assert [from_aos(d) for d in b2] == case["base2_reduced"]
        assert from_aos(ring.rescale_drop0(c)) == case["rescale_drop1"]
        # decompose recomposes: sum_i d_i * g_i == c, g_i = unit vector of limb i (TrivGad)
        for i, q in enumerate(qs):
            assert case["triv_reduced"][i][i] == case["c"][i]
        for i, q in enumerate(qs):
            assert all(-(q - 1) // 2 <= v <= (q - 1) // 2 for v in case["triv_digits"][i])


def test_c_oracle_matches_golden_mul_relin(oracle_lib):
    for case in load_golden("mul_relin_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        ring = oracle_lib.Ring(n, qs)
        hint = hint_to_crt_aos(ring, case["hint"])
        a0, a1 = (to_aos(x) for x in case["cta"]["c"])
        b0, b1 = (to_aos(x) for x in case["ctb"]["c"])
        o0, o1 = ring.ct_mul_relin(hint, a0, a1, b0, b1, s_pre=case["s_pre"], pow_basis=True)
        assert from_aos(o0) == case["out"]["c"][0]
        assert from_aos(o1) == case["out"]["c"][1]
        # CRT-basis entry point agrees with the Pow-basis one
        e0, e1 = ring.ct_mul_relin(hint, ring.crt(a0), ring.crt(a1), ring.crt(b0), ring.crt(b1), s_pre=case["s_pre"])
        assert np.array_equal(ring.crtinv(e0), o0) and np.array_equal(ring.crtinv(e1), o1)
        # and the result still decrypts to the plaintext product (model decrypt on the oracle's output)
        out = M.CT(M.MSD, case["out"]["k"], case["out"]["l"], [from_aos(o0), from_aos(o1)], case["p"], qs)
        assert M.decrypt(case["sk"], out, case["npt"]) == case["want_pt"]


def test_arithmetic_scenario_fixture_passes():
    """examples/Arithmetic.hs:73-75 prints PASS when decrypt(result) == plaintext result."""
    g = load_golden("arithmetic_m32.json")
    res = g["result"]
    ct = M.CT(res["enc"], res["k"], res["l"], res["c"], g["p"], g["qs"][1:])
    assert M.decrypt(g["sk"], ct, g["npt"]) == g["want_pt"]
    assert len(ct.qs) == 1 and len(ct.c) == 2


def test_full_size_digests(oracle_lib):
    for case in load_golden("digests_full.json")["cases"]:
        n, qs, seed = 1 << case["logn"], case["qs"], case["seed"]
        ring = oracle_lib.Ring(n, qs)
        a = [ring.fill_uniform(seed, e) for e in range(4)]
        assert digest_limb_major(ring.crt(a[0])) == case["crt_elem0_sha256"]
        assert digest_limb_major(ring.crtinv(a[1])) == case["crtinv_elem1_sha256"]
        if len(qs) > 1:
            hint = [ring.fill_uniform(0xA1C4E5, e) for e in range(2 * len(qs))]
            o0, o1 = ring.ct_mul_relin(hint, *a)
            assert digest_limb_major(o0, o1) == case["mul_relin_crt_sha256"]


# ---- properties (mirroring Lol's upstream property tests) ------------------------------------------------

SMALL_QS = [12289, 40961, 65537]


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**32), st.sampled_from([4, 8, 16, 32]))
def test_prop_model_crt_roundtrip_and_convolution(seed, n):
    rng = random.Random(seed)
    q = rng.choice(SMALL_QS)
    a = [rng.randrange(q) for _ in range(n)]
    b = [rng.randrange(q) for _ in range(n)]
    ca, cb = M.crt_def(a, q), M.crt_def(b, q)
    assert M.crtinv_def(ca, q) == a
    assert M.crtinv_def([x * y % q for x, y in zip(ca, cb)], q) == M.negacyclic_mul(a, b, q)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**32), st.sampled_from([16, 64, 256, 1024]))
def test_prop_c_oracle_ring_laws(oracle_lib, seed, n):
    rng = np.random.default_rng(seed)
    qs = [268440577, 8392193] if n <= 256 else [12289, 65537]
    ring = oracle_lib.Ring(n, qs)
    a, b, c = (np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(3))
    assert np.array_equal(ring.crtinv(ring.crt(a)), a)
    # crt is linear, and turns ring products into pointwise products (distributivity through the CRT basis)
    assert np.array_equal(ring.crt(ring.add(a, b)), ring.add(ring.crt(a), ring.crt(b)))
    lhs = ring.mul(ring.crt(a), ring.add(ring.crt(b), ring.crt(c)))
    rhs = ring.add(ring.mul(ring.crt(a), ring.crt(b)), ring.mul(ring.crt(a), ring.crt(c)))
    assert np.array_equal(lhs, rhs)
    # decompose recomposes (TrivGad): digit i reduced mod q_i is limb i again
    digs = ring.decompose_triv(a)
    for i in range(len(qs)):
        assert np.array_equal(digs[i][:, i], a[:, i])


@settings(max_examples=8, deadline=None)
@given(st.integers(0, 2**32))
def test_prop_key_switch_preserves_decryption(seed):
    rng = random.Random(seed)
    n, npt, p, qs = 16, 2, 7, [268440577, 8392193]
    sk = M.gen_sk(n, 3.0, rng)
    pa, pb = ([rng.randrange(p) for _ in range(npt)] for _ in range(2))
    cta, ctb = M.encrypt(sk, pa, p, qs, 3.0, rng), M.encrypt(sk, pb, p, qs, 3.0, rng)
    hint = M.ks_quad_circ_hint(sk, qs, 3.0, rng, "triv")
    out = M.ct_mul_relin(hint, cta, ctb)
    assert len(out.c) == 2 and out.k == 1
    assert M.decrypt(sk, out, npt) == M.negacyclic_mul(pa, pb, p)
    # BaseBGad 2 gives the same plaintext with a different (smaller-noise) ciphertext
    hint2 = M.ks_quad_circ_hint(sk, qs, 3.0, rng, "base2")
    assert M.decrypt(sk, M.key_switch_quad_circ(hint2, M.ct_mul(cta, ctb)), npt) == M.negacyclic_mul(pa, pb, p)
    # toMSD . toLSD = id
    assert M.to_lsd(M.to_msd(cta)).c == cta.c and M.to_lsd(M.to_msd(cta)).l == cta.l


def test_full_mul_composition_matches_model(oracle_lib):
    """helpers.oracle_full_mul (the checker of alch_ct_mul_full on the GPU) against the exact model's fixtures."""
    from helpers import oracle_full_mul
    for case in load_golden("full_mul_small.json")["cases"]:
        n, p, qs_h, l_in, l_out = case["n"], case["p"], case["qs_hint"], case["l_in"], case["l_out"]
        L = len(qs_h)
        o_in, o_h = oracle_lib.Ring(n, qs_h[L - l_in:]), oracle_lib.Ring(n, qs_h)
        x = [o_in.crt(to_aos(c)) for c in case["x"]["c"]]
        y = [o_in.crt(to_aos(c)) for c in case["y"]["c"]]
        hint = hint_to_crt_aos(o_h, case["hint"])
        s = [pow(p, -1, q) for q in qs_h[L - l_in:]]          # fresh encryptions are LSD; modSwitch's toMSD
        r0, r1 = oracle_full_mul(oracle_lib, n, qs_h, l_in, l_out, hint, x[0], x[1], y[0], y[1], s_pre=s, pow_out=True)
        assert [from_aos(r0), from_aos(r1)] == case["result"]["c"]


def test_base2_key_switch_composition_matches_model(oracle_lib):
    """helpers.oracle_mul_relin_base2 (the checker of the BaseBGad device path) against the exact model's fixture."""
    from helpers import oracle_mul_relin_base2
    for case in load_golden("mul_relin_base2_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        o = oracle_lib.Ring(n, qs)
        a = [o.crt(to_aos(c)) for c in case["cta"]["c"]]
        b = [o.crt(to_aos(c)) for c in case["ctb"]["c"]]
        hint = hint_to_crt_aos(o, case["hint"])
        r0, r1 = oracle_mul_relin_base2(oracle_lib, n, qs, hint, a[0], a[1], b[0], b[1], s_pre=case["s_pre"], pow_out=True)
        assert [from_aos(r0), from_aos(r1)] == case["out"]["c"]
