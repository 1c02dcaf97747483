"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact (integer arithmetic)."""
import numpy as np
import pytest

from conftest import ARITH_QS, CFG2_Q60, CFG3_QS, Q30_QS

pytestmark = pytest.mark.gpu

# eight limbs, all = 1 mod 4096: balanced 31-bit primes and small unbalanced ones
EIGHT_QS = [2147389441, 2147377153, 2147352577, 2147295233, 2147217409, 2147205121, 2147196929, 2147082241]
EIGHT_SMALL_QS = [12289, 40961, 61441, 65537, 86017, 114689, 147457, 151553]
# all = 1 mod 2^16, wildly different sizes
UNBAL_QS = [2147352577, 65537, 786433]


def _rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


def _ring_pair(oracle_lib, n, qs):
    import alchemy_amd as A
    return A.Ring(2 * n, qs), oracle_lib.Ring(n, qs)


# = 1 mod 2^17: the synthetic two-power stand-ins for BASELINE configs 4 / 5
SIX_QS_17 = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]


@pytest.mark.parametrize("logn", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_crt_crtinv_match_oracle(oracle_lib, logn):
    n = 1 << logn
    qs = SIX_QS_17[:3] if logn == 16 else CFG3_QS[:3] if logn > 8 else ARITH_QS
    g, o = _ring_pair(oracle_lib, n, qs)
    rng = np.random.default_rng(100 + logn)
    x = _rand_elems(rng, 3, n, qs)
    buf = g.upload(x)
    buf.crt()
    got = buf.download()
    for e in range(3):
        assert np.array_equal(got[e], o.crt(x[e])), f"crt mismatch elem {e}"
    buf.crtinv()
    assert np.array_equal(buf.download(), x), "crtInv . crt != id"
    # crtInv on arbitrary CRT-basis input against the oracle
    y = _rand_elems(rng, 2, n, qs)
    b2 = g.upload(y)
    b2.crtinv()
    got = b2.download()
    for e in range(2):
        assert np.array_equal(got[e], o.crtinv(y[e]))


@pytest.mark.parametrize("logn", [4, 8, 14, 15])
def test_crt_60bit(oracle_lib, logn):
    n = 1 << logn
    qs = [1152921504606584833 if logn == 15 else CFG2_Q60]          # = 1 mod 2^16 for n = 2^15 (split transform)
    g, o = _ring_pair(oracle_lib, n, qs)
    assert g.word_bytes == 8
    rng = np.random.default_rng(7)
    x = _rand_elems(rng, 2, n, qs)
    buf = g.upload(x)
    buf.crt()
    got = buf.download()
    for e in range(2):
        assert np.array_equal(got[e], o.crt(x[e]))
    buf.crtinv()
    assert np.array_equal(buf.download(), x)
    # pointwise product = negacyclic convolution (config 2: fwd NTT + pointwise mul + inverse NTT)
    a, b = g.upload(x[:1]), g.upload(x[1:])
    a.crt(); b.crt()
    c = g.alloc(1)
    c.mul(a, b, 1)
    c.crtinv()
    want = o.crtinv(o.mul(o.crt(x[0]), o.crt(x[1])))
    assert np.array_equal(c.download()[0], want)


def _mul_relin_case(oracle_lib, n, qs, batch, seed, s_pre=None, pow_basis=False):
    import alchemy_amd as A
    g, o = _ring_pair(oracle_lib, n, qs)
    L = len(qs)
    rng = np.random.default_rng(seed)
    hint = _rand_elems(rng, 2 * L, n, qs)
    a = _rand_elems(rng, 2 * batch, n, qs)
    b = _rand_elems(rng, 2 * batch, n, qs)
    gh = g.hint_load(hint)
    ga, gb, gout = g.upload(a), g.upload(b), g.alloc(2 * batch)
    flags = (A.capi.ALCH_POW_IN | A.capi.ALCH_POW_OUT) if pow_basis else 0
    g.ct_mul_relin(gh, ga, gb, gout, batch, s_pre=s_pre, flags=flags)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = o.ct_mul_relin(list(hint), a[2 * ct], a[2 * ct + 1], b[2 * ct], b[2 * ct + 1], s_pre=s_pre,
                                pow_basis=pow_basis)
        assert np.array_equal(got[2 * ct], w0), f"c0 mismatch ct {ct}"
        assert np.array_equal(got[2 * ct + 1], w1), f"c1 mismatch ct {ct}"


@pytest.mark.parametrize("logn,qs,batch", [
    (4, ARITH_QS, 3), (6, ARITH_QS[:2], 9), (8, ARITH_QS, 5), (8, ARITH_QS[:1], 2),
    (10, CFG3_QS, 3), (11, CFG3_QS, 9), (12, CFG3_QS[:2], 2), (13, CFG3_QS, 2), (14, CFG3_QS, 1), (15, CFG3_QS, 2),
    (15, CFG3_QS, 11),
    # unbalanced moduli (a digit of one limb exceeds another limb's modulus): general reduce path
    (11, UNBAL_QS, 3), (15, UNBAL_QS, 2), (8, [1073750017, 8392193], 2),
    # limb-count extremes on the two-workgroup kernel: one limb (no digit transform at all) and the maximum of 8
    (15, CFG3_QS[:1], 3), (11, CFG3_QS[:1], 9), (11, EIGHT_QS, 2), (11, EIGHT_SMALL_QS, 3),
    # n = 2^16: split transforms + unfused key switch (configs 4 / 5 stand-in: six primes = 1 mod 2^17)
    (16, SIX_QS_17, 2), (16, SIX_QS_17[:2], 3),
    (15, [1152921504606584833, 1152921504598720513], 2),      # 60-bit residues, n = 2^15: the same on 8-byte words
    # every modulus below 2^30: the Harvey-butterfly instantiations of the two fused kernels (values lazy in [0,4q), accumulators in [0,2q))
    (11, Q30_QS[:4], 9), (15, Q30_QS[:4], 2), (15, Q30_QS[:4], 11), (15, Q30_QS[:1], 3), (11, Q30_QS, 3), (15, Q30_QS[1:3], 5),
    (11, [786433, 1073479681], 3),                            # below 2^30 but unbalanced: the general kernels
])
def test_ct_mul_relin_crt_basis(oracle_lib, logn, qs, batch):
    _mul_relin_case(oracle_lib, 1 << logn, qs, batch, seed=1000 + logn)


@pytest.mark.parametrize("logn,qs,batch", [(4, ARITH_QS, 2), (8, ARITH_QS, 3), (12, CFG3_QS, 1), (15, CFG3_QS, 1),
                                           (16, SIX_QS_17[:3], 1), (15, Q30_QS[:3], 2)])
def test_ct_mul_relin_pow_basis(oracle_lib, logn, qs, batch):
    _mul_relin_case(oracle_lib, 1 << logn, qs, batch, seed=2000 + logn, pow_basis=True)


def test_harvey_kernels_agree_with_the_general_ones():
    """Option q30 = 0 sends a ring with moduli below 2^30 through the general (lazy [0,2q)) kernels: same words, whole batch."""
    import alchemy_amd as A
    n, qs, batch = 1 << 15, Q30_QS[:4], 40
    outs = []
    for q30 in (1, 0):
        g = A.Ring(2 * n, qs)
        g.set_option("q30", q30)
        g.set_option("chunk", 16)
        rng = np.random.default_rng(4242)
        hint, a, b = _rand_elems(rng, 8, n, qs), _rand_elems(rng, 2 * batch, n, qs), _rand_elems(rng, 2 * batch, n, qs)
        out = g.alloc(2 * batch)
        g.ct_mul_relin(g.hint_load(hint), g.upload(a), g.upload(b), out, batch, s_pre=[3, 5, 7, 11])
        outs.append(out.download())
    assert np.array_equal(outs[0], outs[1])
    assert int(outs[0].max()) < max(qs) and int(outs[0].min()) >= 0


def test_ct_mul_relin_with_encoding_scalar(oracle_lib):
    # toLSD/toMSD scalars folded into the tensor product: s = p^-1 mod q (p = 7, examples/Arithmetic.hs:23)
    qs = ARITH_QS
    s = [pow(7, -1, q) for q in qs]
    _mul_relin_case(oracle_lib, 256, qs, 2, seed=77, s_pre=s)


def test_device_pointer_view_is_the_limb_major_buffer():
    """alch_buf_device_ptr / Buf.as_torch: the zero-copy view RCCL gathers of result batches use (bench.py)."""
    import torch
    import alchemy_amd as A
    n, qs = 64, ARITH_QS
    g = A.Ring(2 * n, qs)
    rng = np.random.default_rng(3)
    x = _rand_elems(rng, 3, n, qs)
    buf = g.upload(x)
    g.sync()
    addr, nbytes = buf.device_ptr()
    assert addr != 0 and nbytes == 3 * n * len(qs) * g.word_bytes
    view = buf.as_torch(1, 2)                                  # elements 1, 2
    want = np.ascontiguousarray(np.transpose(x[1:], (0, 2, 1))).reshape(-1)      # [elem][limb][coefficient]
    assert np.array_equal(view.cpu().numpy().astype(np.int64), want)
    view.zero_()                                               # writes through to the library's buffer
    torch.cuda.synchronize()
    got = buf.download()
    assert np.array_equal(got[0], x[0]) and not got[1:].any()


@pytest.mark.parametrize("logn,qs", [(15, CFG3_QS[:2]), (14, [CFG2_Q60])])
@pytest.mark.parametrize("half", [0, 1])
def test_crt_whole_and_half_forms(oracle_lib, logn, qs, half):
    """A 128-KiB limb-polynomial has two crt kernels (k_crt: whole polynomial in LDS; k_crt_half: two half-size
    sub-transforms, two workgroups per CU; option "crt_half"): both against the oracle, in place and on CRT-basis input."""
    n = 1 << logn
    g, o = _ring_pair(oracle_lib, n, qs)
    g.set_option("crt_half", half)
    rng = np.random.default_rng(300 + logn + half)
    x = _rand_elems(rng, 3, n, qs)
    buf = g.upload(x)
    buf.crt()
    got = buf.download()
    for e in range(3):
        assert np.array_equal(got[e], o.crt(x[e])), f"crt mismatch elem {e}"
    buf.crtinv()
    assert np.array_equal(buf.download(), x), "crtInv . crt != id"
    y = _rand_elems(rng, 2, n, qs)
    b2 = g.upload(y)
    b2.crtinv()
    got = b2.download()
    for e in range(2):
        assert np.array_equal(got[e], o.crtinv(y[e]))
