"""Empty inputs: every batched entry point called with count / batch = 0 answers ALCH_OK, launches nothing that could fault (a zero-block
grid is a HIP launch error) and leaves its buffers untouched -- on a two-power ring (the fused n = 2^15 kernels' dispatcher) and on a
general index (the pass engine).  The reference maps its functions over lists of ciphertexts; an empty list is a legal input there."""
import ctypes as C

import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi

pytestmark = pytest.mark.gpu

CFG3_QS = [2147352577, 2146959361, 2146041857, 2145976321]
RLWR_QS = [1543651201, 689270401, 718099201, 720720001]


def rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


@pytest.mark.parametrize("m,qs", [(1 << 16, CFG3_QS), (1 << 12, CFG3_QS[:3]), (11648, RLWR_QS), (420, RLWR_QS[:1])])
def test_zero_counts_are_no_ops(m, qs):
    r = A.Ring(m, qs)
    rng = np.random.default_rng(m)
    xs = rand_elems(rng, 4, r.n, r.qs)
    a, b, out = r.upload(xs), r.upload(xs[::-1].copy()), r.upload(xs)
    hint = r.hint_load(rand_elems(rng, 2 * len(qs), r.n, r.qs))
    l = capi.load_library()
    ones = capi._pu64([1] * len(qs))
    cs = C.c_uint64(123)
    calls = {
        "crt": l.alch_buf_crt(a._h, 0, 0), "crtinv": l.alch_buf_crtinv(a._h, 4, 0), "l": l.alch_buf_l(a._h, 0, 0), "linv": l.alch_buf_linv(a._h, 0, 0),
        "mulg": l.alch_buf_mulg(a._h, 0, 0, capi.ALCH_BASIS_POW), "divg": l.alch_buf_divg(a._h, 0, 0, capi.ALCH_BASIS_CRT),
        "mul": l.alch_buf_mul(out._h, a._h, b._h, 0), "add": l.alch_buf_add(out._h, a._h, b._h, 0), "sub": l.alch_buf_sub(out._h, a._h, b._h, 0),
        "scale": l.alch_buf_scale(out._h, a._h, 0, ones), "copy": l.alch_buf_copy(out._h, 0, a._h, 0, 0),
        "tensor_op": l.alch_buf_tensor_op(out._h, 0, a._h, 0, 0, capi.ALCH_T_CRT),
        "mul_public": l.alch_buf_mul_public(out._h, a._h, b._h, 0, 0), "add_public": l.alch_buf_add_public(out._h, b._h, 0, 0),
        "ct_add_public": l.alch_ct_add_public(out._h, a._h, 0, ones, b._h, 0),
        "checksum": l.alch_buf_checksum(a._h, 0, 0, C.byref(cs)),
        "mul_relin": l.alch_ct_mul_relin(r._h, hint._h, a._h, b._h, out._h, 0, None, 0),
        "mul_relin_pow": l.alch_ct_mul_relin(r._h, hint._h, a._h, b._h, out._h, 0, None, capi.ALCH_POW_IN | capi.ALCH_POW_OUT),
    }
    bad = {k: v for k, v in calls.items() if v != capi.ALCH_OK}
    assert not bad, (bad, l.alch_last_error())
    assert cs.value == 0                                                     # the checksum of nothing
    r.sync()
    assert np.array_equal(a.download(), xs) and np.array_equal(b.download(), xs[::-1]) and np.array_equal(out.download(), xs)
    # and the library still works afterwards
    out.mul(a, b, 4)
    assert np.array_equal(out.download()[1], r.mul(xs[1], xs[2]))


def test_zero_batches_between_rings():
    """mul_full (4 -> 5 -> 3 limbs), modSwitch between nested rings, a tunnel and the two-ring Tensor methods with an empty batch."""
    qs = [1543651201, 689270401, 718099201, 720720001, 1556755201]
    rin, rh, rout = A.Ring(11648, qs[1:]), A.Ring(11648, qs), A.Ring(11648, qs[2:])
    rng = np.random.default_rng(5)
    xs = rand_elems(rng, 2, rin.n, rin.qs)
    a, b = rin.upload(xs), rin.upload(xs)
    out = rout.alloc(2)
    out.fill_uniform(9)
    before = out.download()
    hint = rh.hint_load(rand_elems(rng, 2 * len(qs), rh.n, rh.qs))
    l = capi.load_library()
    assert l.alch_ct_mul_full(hint._h, a._h, b._h, out._h, 0, None, 0) == capi.ALCH_OK, l.alch_last_error()
    assert l.alch_ct_mod_switch(a._h, out._h, 0, 0) == capi.ALCH_OK, l.alch_last_error()
    small, big = A.Ring(2912, qs[:2]), A.Ring(11648, qs[:2])
    s, g = small.alloc(2), big.alloc(8)
    s.fill_uniform(1); g.fill_uniform(2)
    sb, gb = s.download(), g.download()
    for rc in (l.alch_buf_embed(g._h, s._h, 0, capi.ALCH_BASIS_POW), l.alch_buf_twace(s._h, g._h, 0, capi.ALCH_BASIS_CRT), l.alch_buf_coeffs(s._h, g._h, 0)):
        assert rc == capi.ALCH_OK, l.alch_last_error()
    rout.sync(); small.sync(); big.sync()
    assert np.array_equal(out.download(), before) and np.array_equal(s.download(), sb) and np.array_equal(g.download(), gb)
