"""GPU parity of alch_ct_mul_full -- PT2CT's whole mul_ (PT2CT.hs:160-177): (*) , modSwitch to the hint's modulus,
keySwitchQuadCirc, modSwitch to the output modulus -- against the exact model's fixture and against the C
restatement's composition of the same steps (helpers.oracle_full_mul).  Bit-exact."""
import numpy as np
import pytest

from conftest import CFG3_QS, Q30_QS
from helpers import from_aos, hint_to_crt_aos, load_golden, oracle_full_mul, to_aos

pytestmark = pytest.mark.gpu

# all = 1 mod 2^17; the reference's shape is L_in = 4 -> L_h = 5 -> L_out = 3
SIX_QS = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]
Q60S = [1152921504606748673, 1152921504606683137, 1152921504606584833]      # = 1 mod 2^15


def _rand(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


def test_full_mul_golden(oracle_lib):
    import alchemy_amd as A
    for case in load_golden("full_mul_small.json")["cases"]:
        n, p, qs_h, l_in, l_out = case["n"], case["p"], case["qs_hint"], case["l_in"], case["l_out"]
        L = len(qs_h)
        rin, rh, rout = A.Ring(2 * n, qs_h[L - l_in:]), A.Ring(2 * n, qs_h), A.Ring(2 * n, qs_h[L - l_out:])
        o_h = oracle_lib.Ring(n, qs_h)
        x = rin.upload(np.stack([to_aos(c) for c in case["x"]["c"]])); x.crt()
        y = rin.upload(np.stack([to_aos(c) for c in case["y"]["c"]])); y.crt()
        hint = rh.hint_load(np.stack(hint_to_crt_aos(o_h, case["hint"])))
        out = rout.alloc(2)
        s = [pow(p, -1, q) for q in rin.qs]                # fresh encryptions are LSD; the first modSwitch's toMSD
        A.capi.ct_mul_full(hint, x, y, out, 1, s_pre=s, flags=A.capi.ALCH_POW_OUT)
        assert [from_aos(e) for e in out.download()] == case["result"]["c"]
        A.capi.ct_mul_full(hint, x, y, out, 1, s_pre=s)
        out.crtinv()
        assert [from_aos(e) for e in out.download()] == case["result"]["c"]


def _case(oracle_lib, logn, qs_h, l_in, l_out, batch, seed, s_pre=None, pow_out=False):
    import alchemy_amd as A
    n, L = 1 << logn, len(qs_h)
    rng = np.random.default_rng(seed)
    rin, rh, rout = A.Ring(2 * n, qs_h[L - l_in:]), A.Ring(2 * n, qs_h), A.Ring(2 * n, qs_h[L - l_out:])
    hint = _rand(rng, 2 * L, n, qs_h)
    a = _rand(rng, 2 * batch, n, qs_h[L - l_in:])
    b = _rand(rng, 2 * batch, n, qs_h[L - l_in:])
    gh, ga, gb, gout = rh.hint_load(hint), rin.upload(a), rin.upload(b), rout.alloc(2 * batch)
    A.capi.ct_mul_full(gh, ga, gb, gout, batch, s_pre=s_pre, flags=A.capi.ALCH_POW_OUT if pow_out else 0)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = oracle_full_mul(oracle_lib, n, qs_h, l_in, l_out, list(hint), a[2 * ct], a[2 * ct + 1],
                                 b[2 * ct], b[2 * ct + 1], s_pre=s_pre, pow_out=pow_out)
        assert np.array_equal(got[2 * ct], w0), f"c0 mismatch ct {ct}"
        assert np.array_equal(got[2 * ct + 1], w1), f"c1 mismatch ct {ct}"


@pytest.mark.parametrize("logn,qs_h,l_in,l_out,batch", [
    (4, SIX_QS[:3], 2, 1, 3),
    (8, SIX_QS[:5], 4, 3, 5),
    (10, SIX_QS[:5], 4, 3, 3),
    (11, SIX_QS[:5], 4, 3, 9),              # two-workgroup key-switch kernel, the reference's 4 -> 5 -> 3
    (11, SIX_QS, 4, 3, 3),                  # two limbs added, three dropped
    (11, SIX_QS[:3], 2, 2, 2),              # one added, one dropped
    (11, [65537, 786433, 2147352577], 2, 1, 3),     # unbalanced: general reduce in the digits and in the rescale
    (11, [2147352577, 65537, 786433], 2, 1, 3),     # unbalanced the other way (lifted residue exceeds kept moduli)
    (13, SIX_QS[:5], 4, 3, 2),
    (15, CFG3_QS[:1] + SIX_QS[1:5], 4, 3, 2),
    (15, SIX_QS[:5], 4, 3, 9),
    (15, SIX_QS, 4, 3, 2),                  # n = 2^15, three limbs dropped: the half-size rescale kernels (kernel_rescale_half.hpp)
    (15, [2147352577, 65537, 786433, 2146959361], 3, 2, 2),   # n = 2^15, two dropped, unbalanced: same kernels, general reduce
    # n = 2^16 (split transforms): the same entry point composes the op from element-wise kernels and batched transforms
    (16, SIX_QS, 5, 4, 2), (16, SIX_QS[:4], 2, 1, 3),
    # every modulus below 2^30: Harvey-butterfly instantiations of the tensor and key-switch kernels (here with added limbs)
    (11, Q30_QS[:5], 4, 3, 9), (15, Q30_QS[:5], 4, 3, 3), (15, Q30_QS, 4, 3, 2), (15, Q30_QS[:3], 2, 2, 2),
])
def test_full_mul_matches_oracle(oracle_lib, logn, qs_h, l_in, l_out, batch):
    _case(oracle_lib, logn, qs_h, l_in, l_out, batch, seed=5000 + logn)


@pytest.mark.parametrize("logn", [6, 11, 15, 16])
def test_full_mul_pow_out_with_scalar(oracle_lib, logn):
    qs = SIX_QS[:5]
    _case(oracle_lib, logn, qs, 4, 3, 2, seed=6000 + logn, s_pre=[pow(2, -1, q) for q in qs[1:]], pow_out=True)


@pytest.mark.parametrize("logn", [5, 10, 14, 15])
def test_full_mul_60bit(oracle_lib, logn):
    qs = [1152921504606584833, 1152921504598720513, 1152921504597016577] if logn == 15 else Q60S    # = 1 mod 2^16
    _case(oracle_lib, logn, qs, 2, 1, 2, seed=7000 + logn)


def test_full_mul_rejects_bad_rings():
    import alchemy_amd as A
    n = 64
    rh, rin, rout = A.Ring(2 * n, SIX_QS[:3]), A.Ring(2 * n, SIX_QS[1:3]), A.Ring(2 * n, SIX_QS[2:3])
    other = A.Ring(2 * n, SIX_QS[:2])                       # a prefix, not a suffix, of the hint's ring
    hint = rh.hint_load(np.zeros((6, n, 3), dtype=np.int64))
    x, y = rin.alloc(2), rin.alloc(2)
    with pytest.raises(A.capi.AlchemyError):
        A.capi.ct_mul_full(hint, x, y, other.alloc(2), 1)
    with pytest.raises(A.capi.AlchemyError):
        A.capi.ct_mul_full(hint, other.alloc(2), other.alloc(2), rout.alloc(2), 1)
    with pytest.raises(A.capi.AlchemyError):
        A.capi.ct_mul_full(hint, x, y, rh.alloc(2), 1)      # nothing dropped: not this entry point
    A.capi.ct_mul_full(hint, x, y, rout.alloc(2), 1)
