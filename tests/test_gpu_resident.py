"""GPU: device-resident Tensor values (include/alchemy_hip.h: alch_buf_tensor_op, alch_buf_copy, alch_buf_view,
alch_ring_share_stream, the pooled small buffers and the pinned staging of small transfers) -- what `GT`'s constructor GTDev and
the Resident mode of alchemy_amd/host/cycgen.hpp call.

Every out-of-place op must equal the host-buffer Tensor method of the same ring (itself pinned to the oracle by the other GPU
tests) and the C restatement directly; recycled buffers must never be observed half-written."""
import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi

pytestmark = pytest.mark.gpu

RLWR_QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
CFG3_QS = [2147352577, 2146959361, 2146041857, 2145976321]
SPLIT_QS = [2147352577, 2146959361]                       # = 1 mod 2^17


def rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) if q else rng.integers(-999, 999, size=n, dtype=np.int64)
                               for q in qs], axis=1) for _ in range(count)])


UNARY = [(capi.ALCH_T_CRT, "crt"), (capi.ALCH_T_CRTINV, "crtinv"), (capi.ALCH_T_L, "l"), (capi.ALCH_T_LINV, "linv"),
         (capi.ALCH_T_MULG_POW, "mulg_pow"), (capi.ALCH_T_MULG_DEC, "mulg_dec"), (capi.ALCH_T_MULG_CRT, "mulg_crt"),
         (capi.ALCH_T_DIVG_CRT, "divg_crt")]


@pytest.mark.parametrize("m,qs", [(1 << 12, CFG3_QS), (1 << 16, CFG3_QS), (1 << 17, SPLIT_QS), (11648, RLWR_QS[:6][::-1]),
                                  (20475, RLWR_QS[:4]), (420, [RLWR_QS[0]]), (64, CFG3_QS[:2])])
def test_unary_tensor_ops_out_of_place_equal_the_host_buffer_methods(oracle_lib, m, qs):
    r = A.Ring(m, qs)
    o = oracle_lib.GenRing(m, qs)
    rng = np.random.default_rng(m)
    xs = rand_elems(rng, 3, r.n, qs)
    src, dst = r.upload(xs), r.alloc(3)
    for op, name in UNARY:
        dst.fill_uniform(99)                                    # stale contents must be overwritten
        assert dst.tensor_op(src, op, count=2, dst_first=1, src_first=0)
        got = dst.download()
        for i in range(2):
            assert np.array_equal(got[1 + i], getattr(r, name)(xs[i])), (name, i)
            assert np.array_equal(got[1 + i], getattr(o, name)(xs[i])), (name, i, "oracle")
        assert np.array_equal(src.download(), xs), name         # the source is untouched
    # in place = the same range
    inp = r.upload(xs)
    assert inp.tensor_op(inp, capi.ALCH_T_CRT, count=3)
    assert np.array_equal(inp.download(), np.stack([o.crt(x) for x in xs]))
    # divG on Pow / Dec: multiples of g divide; the flag comes back per call
    for mul, div, oname in ((capi.ALCH_T_MULG_POW, capi.ALCH_T_DIVG_POW, "divg_pow"), (capi.ALCH_T_MULG_DEC, capi.ALCH_T_DIVG_DEC, "divg_dec")):
        g = r.alloc(3)
        assert g.tensor_op(src, mul, count=3)
        back = r.alloc(3)
        assert back.tensor_op(g, div, count=3)
        assert np.array_equal(back.download(), xs)
    with pytest.raises(capi.AlchemyError):
        dst.tensor_op(src, 10)
    with pytest.raises(capi.AlchemyError):
        dst.tensor_op(dst, capi.ALCH_T_CRT, count=2, dst_first=1, src_first=0)     # overlapping, not identical


def test_divg_not_divisible_and_no_crt_on_plaintext_and_integer_rings(oracle_lib):
    for m, q in ((20475, 0), (4095, 32), (364, 7)):
        r, o = A.Ring(m, [q], nocrt=True), oracle_lib.GenRing(m, [q])
        rng = np.random.default_rng(m + q)
        xs = rand_elems(rng, 2, r.n, [q])
        src, dst = r.upload(xs), r.alloc(2)
        for op, name in ((capi.ALCH_T_L, "l"), (capi.ALCH_T_LINV, "linv"), (capi.ALCH_T_MULG_POW, "mulg_pow"), (capi.ALCH_T_MULG_DEC, "mulg_dec")):
            assert dst.tensor_op(src, op, count=2)
            assert np.array_equal(dst.download(), np.stack([getattr(o, name)(x) for x in xs])), (m, q, name)
        for op, name in ((capi.ALCH_T_DIVG_POW, "divg_pow"), (capi.ALCH_T_DIVG_DEC, "divg_dec")):
            want = [getattr(o, name)(x) for x in xs]
            ok = dst.tensor_op(src, op, count=2)
            assert ok == all(w is not None for w in want), (m, q, name)
            if ok:
                assert np.array_equal(dst.download(), np.stack(want))
        for op in (capi.ALCH_T_CRT, capi.ALCH_T_CRTINV, capi.ALCH_T_MULG_CRT, capi.ALCH_T_DIVG_CRT):
            with pytest.raises(capi.AlchemyError) as e:
                dst.tensor_op(src, op)
            assert e.value.code == capi.ALCH_E_NO_CRT


def test_views_and_copies():
    r = A.Ring(2912 * 4, RLWR_QS[:3])
    rng = np.random.default_rng(1)
    xs = rand_elems(rng, 5, r.n, r.qs)
    big = r.upload(xs)
    v = big.view(3)
    assert np.array_equal(v.download(), xs[3:4])
    one = r.alloc(1)
    one.copy_from(big, 1, src_first=4)
    assert np.array_equal(one.download()[0], xs[4])
    out = r.alloc(1)
    out.mul(v, one, 1)                                           # a view is a buffer like any other
    assert np.array_equal(out.download()[0], r.mul(xs[3], xs[4]))
    v.tensor_op(v, capi.ALCH_T_CRTINV)                           # writes through to the parent
    assert np.array_equal(big.download(3, 1)[0], r.crtinv(xs[3]))
    v.free()
    assert np.array_equal(big.download(0, 3), xs[:3])            # freeing a view releases nothing
    with pytest.raises(capi.AlchemyError):
        big.view(5)
    with pytest.raises(capi.AlchemyError):
        one.copy_from(big, 2)


def test_element_ranges_that_wrap_size_t_are_refused():
    """first + count and 2 * batch are never formed by the range checks: an argument near 2^64 that would wrap past a naive
    `first + count > n_elems` test is ALCH_E_INVALID, and nothing reaches the device (the data are unchanged afterwards)."""
    import ctypes as C
    r = A.Ring(2912 * 4, RLWR_QS[:3])
    rng = np.random.default_rng(3)
    xs = rand_elems(rng, 4, r.n, r.qs)
    a, b = r.upload(xs), r.upload(xs)
    l, huge, half = capi.load_library(), 2**64 - 1, 2**63 + 1                # huge + 2 = 1, 2 * half = 2 (mod 2^64)
    cs = C.c_uint64()
    for rc in (l.alch_buf_crt(a._h, huge, 2), l.alch_buf_crtinv(a._h, huge, 2), l.alch_buf_checksum(a._h, huge, 2, C.byref(cs)),
               l.alch_buf_copy(a._h, huge, b._h, 0, 2), l.alch_buf_copy(a._h, 0, b._h, huge, 2),
               l.alch_buf_tensor_op(a._h, huge, b._h, 0, 2, capi.ALCH_T_CRT), l.alch_buf_tensor_op(a._h, 0, b._h, huge, 2, capi.ALCH_T_CRT),
               l.alch_ct_mod_switch(a._h, b._h, half, 0)):
        assert rc == capi.ALCH_E_INVALID, rc
    v = C.c_void_p()
    assert l.alch_buf_view(a._h, huge, 2, C.byref(v)) == capi.ALCH_E_INVALID
    assert np.array_equal(a.download(), xs) and np.array_equal(b.download(), xs)


def test_recycled_buffers_are_ordered_on_the_stream(oracle_lib):
    """alloc / op / free chains without any synchronisation: a freed element may be handed out again while the kernel that reads it
    is still queued -- the next writer is queued behind it on the same stream, so results never change."""
    m, qs = 11648, RLWR_QS[:5][::-1]
    r, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    rng = np.random.default_rng(3)
    xs = rand_elems(rng, 4, r.n, qs)
    want = [o.crt(o.mulg_pow(o.crtinv(x))) for x in xs]
    for rounds in range(6):
        res = []
        for x in xs:
            a = r.upload(x[None])                                # pooled, pinned upload: no synchronisation
            b = r.alloc(1); b.tensor_op(a, capi.ALCH_T_CRTINV); a.free()
            c = r.alloc(1); c.tensor_op(b, capi.ALCH_T_MULG_POW); b.free()
            d = r.alloc(1); d.tensor_op(c, capi.ALCH_T_CRT); c.free()
            res.append(d)
        for d, w in zip(res, want):
            assert np.array_equal(d.download()[0], w)
            d.free()


def test_shared_stream_between_rings(oracle_lib):
    """embed / twace / coeffs between two rings that queue on one stream (no events) equal the two-stream results."""
    ms, mb, qs = 128 * 7, 11648, RLWR_QS[:4][::-1]
    rs, rb = A.Ring(ms, qs), A.Ring(mb, qs)
    rng = np.random.default_rng(4)
    x, y = rand_elems(rng, 1, rs.n, qs), rand_elems(rng, 1, rb.n, qs)
    before = (rs.embed_pow(rb, x[0]), rs.embed_crt(rb, x[0]), rs.twace_crt(rb, y[0]), rs.coeffs(rb, y[0]))
    rs.share_stream(rb)
    bx, by = rs.upload(x), rb.upload(y)
    big, small, cs = rb.alloc(1), rs.alloc(1), rs.alloc(rb.n // rs.n)
    big.embed_from(bx, 1, capi.ALCH_BASIS_POW)
    assert np.array_equal(big.download()[0], before[0])
    big.embed_from(bx, 1, capi.ALCH_BASIS_CRT)
    assert np.array_equal(big.download()[0], before[1])
    small.twace_from(by, 1, capi.ALCH_BASIS_CRT)
    assert np.array_equal(small.download()[0], before[2])
    cs.coeffs_from(by, 1)
    assert np.array_equal(cs.download(), before[3])
    rs.share_stream(rb)                                          # idempotent


def test_shared_stream_outlives_the_ring_that_created_it():
    """A ring that borrowed another ring's stream keeps working -- and is destroyed cleanly -- after the lender is destroyed (a
    garbage-collected host destroys rings in any order; the stream belongs to its last user)."""
    qs = RLWR_QS[:3][::-1]
    lender, user, third = A.Ring(128 * 7, qs), A.Ring(11648, qs), A.Ring(128 * 7, qs)
    user.share_stream(lender)
    third.share_stream(user)                                     # a borrowed stream can be lent on
    rng = np.random.default_rng(6)
    x = rand_elems(rng, 2, user.n, qs)
    b = user.upload(x)
    lender.close()                                               # the creator goes first
    b.crt(); b.crtinv()
    assert np.array_equal(b.download(), x)
    b.free()
    user.close()
    y = rand_elems(rng, 1, third.n, qs)
    c = third.upload(y)
    c.crt(); c.crtinv()
    assert np.array_equal(c.download(), y)
    c.free()
    third.close()


def test_dedicated_streams_are_capped_and_refused_gracefully():
    """Option stream_dedicated: at most 32 alive per process (the HIP runtime does not survive running out of hardware queues); beyond
    that the call fails with ALCH_E_UNSUPPORTED and the ring keeps working on its ordinary stream; destroying a ring frees its slot."""
    qs = RLWR_QS[:1]
    rings, refused = [], 0
    for _ in range(40):
        r = A.Ring(128 * 7, qs)
        rings.append(r)
        try:
            r.set_option("stream_dedicated", 1)
        except capi.AlchemyError as e:
            assert e.code == capi.ALCH_E_UNSUPPORTED
            refused += 1
    assert 8 <= refused < 40                                       # other tests of this process may hold a few dedicated streams
    rng = np.random.default_rng(8)
    x = rand_elems(rng, 1, rings[0].n, qs)
    for r in (rings[0], rings[-1]):                                # a dedicated one and a refused one
        b = r.upload(x); b.crt(); b.crtinv()
        assert np.array_equal(b.download(), x)
        b.free()
    rings[0].close()                                               # its slot is free again
    rings[-1].set_option("stream_dedicated", 1)
    b = rings[-1].upload(x); b.crt(); b.crtinv()
    assert np.array_equal(b.download(), x)
    b.free()
    for r in rings[1:]:
        r.close()


def test_transfers_small_and_large_round_trip():
    r = A.Ring(1 << 16, CFG3_QS)                                 # one element = 1 MiB of int64
    rng = np.random.default_rng(5)
    for count in (1, 7, 8, 9, 40):                               # below and above the 8 MiB pinned-staging limit
        xs = rand_elems(rng, count, r.n, r.qs)
        b = r.upload(xs)
        assert np.array_equal(b.download(), xs)
        assert np.array_equal(b.download(count - 1, 1)[0], xs[-1])
