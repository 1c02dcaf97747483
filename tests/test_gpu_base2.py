"""GPU parity of keySwitchQuadCirc hint (a * b) with a BaseBGad 2 hint (PT2CT.hs:140; the gadget Tunnel.hs:24 and
HomomRLWR.hs:46 select): alch_ct_mul_relin with an ALCH_GAD_BASE2 hint against the exact model's fixture and the C
restatement's composition.  Bit-exact."""
import numpy as np
import pytest

from conftest import ARITH_QS, CFG3_QS
from helpers import from_aos, hint_to_crt_aos, load_golden, oracle_full_mul, oracle_mul_relin_base2, to_aos

pytestmark = pytest.mark.gpu


def test_base2_key_switch_golden(oracle_lib):
    import alchemy_amd as A
    for case in load_golden("mul_relin_base2_small.json")["cases"]:
        n, qs = case["n"], case["qs"]
        g, o = A.Ring(2 * n, qs), oracle_lib.Ring(n, qs)
        hint = g.hint_load(np.stack(hint_to_crt_aos(o, case["hint"])), gadget=A.capi.ALCH_GAD_BASE2)
        a = g.upload(np.stack([to_aos(c) for c in case["cta"]["c"]]))
        b = g.upload(np.stack([to_aos(c) for c in case["ctb"]["c"]]))
        out = g.alloc(2)
        g.ct_mul_relin(hint, a, b, out, 1, s_pre=case["s_pre"], flags=A.capi.ALCH_POW_IN | A.capi.ALCH_POW_OUT)
        assert [from_aos(e) for e in out.download()] == case["out"]["c"]


@pytest.mark.parametrize("logn,qs,batch", [
    (4, ARITH_QS[:2], 3), (8, ARITH_QS, 2), (11, CFG3_QS[:2], 3), (11, [65537, 786433, 2147352577], 2),
    (13, CFG3_QS, 1), (15, CFG3_QS, 1), (10, [1152921504606748673, 1152921504606683137], 2),
])
def test_base2_key_switch_matches_oracle(oracle_lib, logn, qs, batch):
    import alchemy_amd as A
    n, L = 1 << logn, len(qs)
    rng = np.random.default_rng(8000 + logn)
    g = A.Ring(2 * n, qs)
    D = g.gadget_digits(A.capi.ALCH_GAD_BASE2)
    assert D == sum((q - 1).bit_length() for q in qs)

    def rand(count):
        return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])

    hint, a, b = rand(2 * D), rand(2 * batch), rand(2 * batch)
    gh = g.hint_load(hint, gadget=A.capi.ALCH_GAD_BASE2)
    ga, gb, gout = g.upload(a), g.upload(b), g.alloc(2 * batch)
    s_pre = [pow(7, -1, q) for q in qs]
    g.ct_mul_relin(gh, ga, gb, gout, batch, s_pre=s_pre)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = oracle_mul_relin_base2(oracle_lib, n, qs, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1],
                                        s_pre=s_pre)
        assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), f"mismatch ct {ct}"


@pytest.mark.parametrize("logn,qs_h,l_in,l_out", [(6, ARITH_QS, 3, 2), (6, ARITH_QS, 2, 1), (11, CFG3_QS[:3], 2, 2), (11, CFG3_QS[:3], 3, 1),
                                                  (13, CFG3_QS[:2], 1, 1)])
def test_full_mul_with_base2_hint(oracle_lib, logn, qs_h, l_in, l_out):
    """PT2CT's whole mul_ (modSwitch . keySwitchQuad hint . modSwitch $ a * b, PT2CT.hs:172-177) with a BaseBGad 2 hint
    (PT2CT.hs:140) whose ring has at least as many limbs as the operands': alch_ct_mul_full against the C restatement's
    op-by-op composition (tensor product on the operands' ring, modSwitch up of all three components, BaseBGad 2 decomposition of c2,
    hint products, modSwitch down)."""
    import alchemy_amd as A
    from alchemy_amd import capi
    n, L, batch = 1 << logn, len(qs_h), 2
    rng = np.random.default_rng(9100 + logn + l_in)
    rh, rin, rout = A.Ring(2 * n, qs_h), A.Ring(2 * n, qs_h[L - l_in:]), A.Ring(2 * n, qs_h[L - l_out:])
    D = rh.gadget_digits(capi.ALCH_GAD_BASE2)

    def rand(count, qs):
        return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])

    hint, a, b = rand(2 * D, qs_h), rand(2 * batch, qs_h[L - l_in:]), rand(2 * batch, qs_h[L - l_in:])
    s_pre = [int(rng.integers(1, q)) for q in qs_h[L - l_in:]]
    gh = rh.hint_load(hint, gadget=capi.ALCH_GAD_BASE2)
    for pow_out in (False, True):
        out = rout.alloc(2 * batch)
        capi.ct_mul_full(gh, rin.upload(a), rin.upload(b), out, batch, s_pre=s_pre, flags=capi.ALCH_POW_OUT if pow_out else 0)
        got = out.download()
        for ct in range(batch):
            w0, w1 = oracle_full_mul(oracle_lib, n, qs_h, l_in, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1],
                                     s_pre, pow_out=pow_out, gadget="base2")
            assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), (ct, pow_out)


@pytest.mark.parametrize("logn,qs_in,l_h,l_out", [(6, ARITH_QS, 2, 2), (6, ARITH_QS, 2, 1), (6, ARITH_QS, 1, 1), (11, CFG3_QS[:3], 2, 1),
                                                  (13, CFG3_QS[:2], 1, 1)])
def test_full_mul_with_base2_hint_on_fewer_limbs(oracle_lib, logn, qs_in, l_h, l_out):
    """The same mul_ when KSPNoise (BaseBGad 2) leaves the hint on FEWER limbs than the product (PT2CT.hs:140 against :164 -- the
    product sits at p + MulPNoise units rounded up to whole limbs): the leading modSwitch then goes DOWN on the quadratic ciphertext,
    c0 on the decoding basis, c1 and c2 on the powerful basis, before the key switch.  alch_ct_mul_full against the C restatement's
    op-by-op composition."""
    import alchemy_amd as A
    from alchemy_amd import capi
    from helpers import oracle_full_mul_base2_down
    n, L, batch = 1 << logn, len(qs_in), 3
    rng = np.random.default_rng(9300 + logn + l_h + l_out)
    rin, rh, rout = A.Ring(2 * n, qs_in), A.Ring(2 * n, qs_in[L - l_h:]), A.Ring(2 * n, qs_in[L - l_out:])
    D = rh.gadget_digits(capi.ALCH_GAD_BASE2)

    def rand(count, qs):
        return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])

    hint, a, b = rand(2 * D, qs_in[L - l_h:]), rand(2 * batch, qs_in), rand(2 * batch, qs_in)
    s_pre = [int(rng.integers(1, q)) for q in qs_in]
    gh = rh.hint_load(hint, gadget=capi.ALCH_GAD_BASE2)
    for pow_out in (False, True):
        out = rout.alloc(2 * batch)
        capi.ct_mul_full(gh, rin.upload(a), rin.upload(b), out, batch, s_pre=s_pre, flags=capi.ALCH_POW_OUT if pow_out else 0)
        got = out.download()
        for ct in range(batch):
            w0, w1 = oracle_full_mul_base2_down(oracle_lib, n, qs_in, l_h, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct],
                                                b[2 * ct + 1], s_pre, pow_out=pow_out)
            assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), (ct, pow_out)


def test_full_mul_with_base2_hint_on_fewer_limbs_general_index(oracle_lib):
    """The down-first mul_ on a general index (m = 2^2 * 3 * 5 * 7 = 420, phi = 96): mulG on the product, c0's rescales on the decoding basis."""
    import alchemy_amd as A
    from alchemy_amd import capi
    from helpers import oracle_full_mul_base2_down, primes_1_mod
    m, batch = 420, 2
    qs_in = primes_1_mod(m, 3, lo=1 << 20)
    rng = np.random.default_rng(9400)
    rin, rh, rout = A.Ring(m, qs_in), A.Ring(m, qs_in[1:]), A.Ring(m, qs_in[2:])
    n = rin.n
    D = rh.gadget_digits(capi.ALCH_GAD_BASE2)

    def rand(count, qs):
        return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])

    hint, a, b = rand(2 * D, qs_in[1:]), rand(2 * batch, qs_in), rand(2 * batch, qs_in)
    s_pre = [int(rng.integers(1, q)) for q in qs_in]
    gh = rh.hint_load(hint, gadget=capi.ALCH_GAD_BASE2)
    for l_out, ro in ((2, rh), (1, rout)):
        for pow_out in (False, True):
            out = ro.alloc(2 * batch)
            capi.ct_mul_full(gh, rin.upload(a), rin.upload(b), out, batch, s_pre=s_pre, flags=capi.ALCH_POW_OUT if pow_out else 0)
            got = out.download()
            for ct in range(batch):
                w0, w1 = oracle_full_mul_base2_down(oracle_lib, n, qs_in, 2, l_out, list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct],
                                                    b[2 * ct + 1], s_pre, pow_out=pow_out, m=m)
                assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), (l_out, ct, pow_out)


def test_full_mul_with_base2_hint_on_fewer_limbs_decrypts_to_the_product():
    """Semantic check of the down-first mul_ on a VALID instance (exact model as the Haskell host: key, encryptions, a BaseBGad 2 hint at
    the last two of three limbs, r = 3.0): the device result equals the model's
        modSwitch . keySwitchQuadCirc hint . modSwitch $ x * y
    (the model rescales every coefficient of the quadratic ciphertext) bit for bit, and the model decrypts it to the product of the
    plaintexts."""
    import random
    import alchemy_amd as A
    from alchemy_amd import capi
    from oracle import model as M
    from helpers import primes_1_mod, to_aos, from_aos
    rng = random.Random(4177)
    n, npt, p = 64, 8, 7
    qs = primes_1_mod(2 * n, 3, lo=1 << 29)
    sk = M.gen_sk(n, 3.0, rng)
    pt1, pt2 = [rng.randrange(p) for _ in range(npt)], [rng.randrange(p) for _ in range(npt)]
    x, y = M.encrypt(sk, pt1, p, qs, 3.0, rng), M.encrypt(sk, pt2, p, qs, 3.0, rng)
    hint = M.ks_quad_circ_hint(sk, qs[1:], 3.0, rng, gadget="base2")
    for l_out in (2, 1):
        want = M.mod_switch_down(M.key_switch_quad_circ(hint, M.mod_switch_down(M.ct_mul(x, y), 1)), 2 - l_out)
        assert M.decrypt(sk, want, npt) == M.negacyclic_mul(pt1, pt2, p)
        rin, rh, rout = A.Ring(2 * n, qs), A.Ring(2 * n, qs[1:]), A.Ring(2 * n, qs[3 - l_out:])
        hb = rh.upload(np.stack([to_aos(h) for pair in hint.h for h in pair])); hb.crt()
        gh = rh.hint_from_buf(hb, gadget=capi.ALCH_GAD_BASE2)
        a = rin.upload(np.stack([to_aos(c) for c in x.c])); a.crt()
        b = rin.upload(np.stack([to_aos(c) for c in y.c])); b.crt()
        out = rout.alloc(2)
        # fresh encryptions are LSD already; the product's toMSD scalar p^-1 is the only one (PT2CT.hs:172-177, SymmSHE toMSD)
        capi.ct_mul_full(gh, a, b, out, 1, s_pre=[pow(p, -1, q) for q in qs], flags=capi.ALCH_POW_OUT)
        got = out.download()
        assert [from_aos(got[0]), from_aos(got[1])] == want.c, l_out
