"""GPU: parity at the BENCH shape -- n = 2^15, 4 limbs, a batch of two full 1024-ciphertext chunks plus a ragged tail at
the DEFAULT launch options (two streams, 4096-workgroup persistent key-switch grid, split tensor kernel), for both
keySwitchQuadCirc(a*b) and PT2CT's whole mul_ (4 -> 5 -> 3 limbs):
  * alch_buf_checksum of the WHOLE result batch equals the value the C oracle produced offline
    (tests/golden/batch_checksums.json, tests/golden/make_batch_checksums.py), so a wrong word anywhere -- any chunk,
    either stream, any persistent-workgroup slot -- fails;
  * 16 ciphertexts spread over the first / middle / last chunk and both streams are downloaded and compared with the
    oracle computed live."""
import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi
from conftest import CFG3_QS
from helpers import load_golden, oracle_full_mul

pytestmark = pytest.mark.gpu

N = 1 << 15
SPOTS = [0, 1, 7, 511, 1023, 1024, 1025, 1531, 2046, 2047, 2048, 2049, 2060, 2071, 2083, 2084]


def test_mul_relin_bench_shape(oracle_lib):
    ref = load_golden("batch_checksums.json")
    B, seeds = ref["test_mul_relin"]["batch"], ref["seeds"]
    assert ref["moduli"] == CFG3_QS and B == 2 * 1024 + 37
    g, o = A.Ring(2 * N, CFG3_QS), oracle_lib.Ring(N, CFG3_QS)
    a, b, out, hs = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * g.L)
    a.fill_uniform(seeds["a"]); b.fill_uniform(seeds["b"]); hs.fill_uniform(seeds["hint"])
    hint = g.hint_from_buf(hs)
    g.ct_mul_relin(hint, a, b, out, B)
    assert f"{out.checksum():016x}" == ref["test_mul_relin"]["checksum"]
    hint_host = [o.fill_uniform(seeds["hint"], i) for i in range(2 * g.L)]
    for ct in SPOTS:
        got = out.download(2 * ct, 2)
        w0, w1 = o.ct_mul_relin(hint_host, o.fill_uniform(seeds["a"], 2 * ct), o.fill_uniform(seeds["a"], 2 * ct + 1),
                                o.fill_uniform(seeds["b"], 2 * ct), o.fill_uniform(seeds["b"], 2 * ct + 1))
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), ct


def test_mul_relin_bench_shape_moduli_below_2_30(oracle_lib):
    """The same batch on four moduli below 2^30 (bench.py's `moduli_below_2_30` line): the Harvey-butterfly instantiations of both
    fused kernels at the default launch options, whole-batch checksum against the C restatement's, spot ciphertexts word for word."""
    ref = load_golden("batch_checksums.json")
    B, seeds, qs = ref["q30"]["test_mul_relin"]["batch"], ref["seeds"], ref["q30"]["moduli"]
    assert max(qs) < 1 << 30 and B == 2 * 1024 + 37
    g, o = A.Ring(2 * N, qs), oracle_lib.Ring(N, qs)
    a, b, out, hs = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * g.L)
    a.fill_uniform(seeds["a"]); b.fill_uniform(seeds["b"]); hs.fill_uniform(seeds["hint"])
    hint = g.hint_from_buf(hs)
    g.ct_mul_relin(hint, a, b, out, B)
    assert f"{out.checksum():016x}" == ref["q30"]["test_mul_relin"]["checksum"]
    hint_host = [o.fill_uniform(seeds["hint"], i) for i in range(2 * g.L)]
    for ct in SPOTS[::3]:
        got = out.download(2 * ct, 2)
        w0, w1 = o.ct_mul_relin(hint_host, o.fill_uniform(seeds["a"], 2 * ct), o.fill_uniform(seeds["a"], 2 * ct + 1),
                                o.fill_uniform(seeds["b"], 2 * ct), o.fill_uniform(seeds["b"], 2 * ct + 1))
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), ct


def test_mul_relin_n16_six_limbs_ragged_batch(oracle_lib):
    """bench.py's `n16_six_limbs` line at a ragged batch (one full 1024-ciphertext chunk + 37): n = 2^16, six limbs, the split-ring
    kernels (k_tensor_crtinv_split, k_ks_accum_split<FROM_OPS>) on two streams; whole-batch checksum against the C restatement's, three
    ciphertexts word for word, and the composed forms (split_fused = 1, 0) give the same words."""
    ref = load_golden("batch_checksums.json")
    B, seeds, qs, n = ref["n16"]["test_mul_relin"]["batch"], ref["seeds"], ref["n16"]["moduli"], ref["n16"]["n"]
    g, o = A.Ring(2 * n, qs), oracle_lib.Ring(n, qs)
    a, b, out, hs = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * g.L)
    a.fill_uniform(seeds["a"]); b.fill_uniform(seeds["b"]); hs.fill_uniform(seeds["hint"])
    hint = g.hint_from_buf(hs)
    g.ct_mul_relin(hint, a, b, out, B)
    assert f"{out.checksum():016x}" == ref["n16"]["test_mul_relin"]["checksum"]
    hint_host = [o.fill_uniform(seeds["hint"], i) for i in range(2 * g.L)]
    for ct in (0, 1023, 1060):
        got = out.download(2 * ct, 2)
        w0, w1 = o.ct_mul_relin(hint_host, o.fill_uniform(seeds["a"], 2 * ct), o.fill_uniform(seeds["a"], 2 * ct + 1),
                                o.fill_uniform(seeds["b"], 2 * ct), o.fill_uniform(seeds["b"], 2 * ct + 1))
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), ct
    for mode in (1, 0):
        g.set_option("split_fused", mode)
        out2 = g.alloc(2 * 70)
        g.ct_mul_relin(hint, a, b, out2, 70)
        assert out2.checksum() == out.checksum(0, 2 * 70), mode


def test_mul_full_bench_shape(oracle_lib):
    ref = load_golden("batch_checksums.json")
    B, seeds = ref["test_mul_full"]["batch"], ref["seeds"]
    qs_h = [ref["full_extra_modulus"]] + CFG3_QS
    rh, rin, rout = A.Ring(2 * N, qs_h), A.Ring(2 * N, CFG3_QS), A.Ring(2 * N, CFG3_QS[1:])
    a, b, out, hs = rin.alloc(2 * B), rin.alloc(2 * B), rout.alloc(2 * B), rh.alloc(2 * rh.L)
    a.fill_uniform(seeds["a"]); b.fill_uniform(seeds["b"]); hs.fill_uniform(seeds["hint"])
    hint = rh.hint_from_buf(hs)
    capi.ct_mul_full(hint, a, b, out, B)
    assert f"{out.checksum():016x}" == ref["test_mul_full"]["checksum"]
    o_in, o_h = oracle_lib.Ring(N, CFG3_QS), oracle_lib.Ring(N, qs_h)
    hint_host = [o_h.fill_uniform(seeds["hint"], i) for i in range(2 * rh.L)]
    for ct in SPOTS[::2]:
        got = out.download(2 * ct, 2)
        w0, w1 = oracle_full_mul(oracle_lib, N, qs_h, 4, 3, hint_host, o_in.fill_uniform(seeds["a"], 2 * ct),
                                 o_in.fill_uniform(seeds["a"], 2 * ct + 1), o_in.fill_uniform(seeds["b"], 2 * ct),
                                 o_in.fill_uniform(seeds["b"], 2 * ct + 1))
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), ct


def test_general_index_bench_shape(oracle_lib):
    """bench.py's `general_index` line (keySwitchQuadCirc(a*b) on H5' = F20475, four HomomRLWR moduli) at a ragged batch: whole-batch
    checksum against the C restatement's offline value, spot ciphertexts against the oracle run live."""
    ref = load_golden("batch_checksums.json")["general_index"]
    B, seeds, m, qs = ref["test"]["batch"], ref["seeds"], ref["index"], ref["moduli"]
    g, o = A.Ring(m, qs), oracle_lib.GenRing(m, qs)
    a, b, out, hs = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * g.L)
    a.fill_uniform(seeds["a"]); b.fill_uniform(seeds["b"]); hs.fill_uniform(seeds["hint"])
    hint = g.hint_from_buf(hs)
    g.ct_mul_relin(hint, a, b, out, B)
    assert f"{out.checksum():016x}" == ref["test"]["checksum"]
    hint_host = [o.fill_uniform(seeds["hint"], i) for i in range(2 * g.L)]
    for ct in (0, 1, 255, 256, B - 1):
        got = out.download(2 * ct, 2)
        w0, w1 = o.ct_mul_relin(hint_host, o.fill_uniform(seeds["a"], 2 * ct), o.fill_uniform(seeds["a"], 2 * ct + 1),
                                o.fill_uniform(seeds["b"], 2 * ct), o.fill_uniform(seeds["b"], 2 * ct + 1))
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), ct
