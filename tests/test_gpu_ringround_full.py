"""GPU: BASELINE config 4 at the reference's REAL parameters -- the op sequence bench.py times (alchemy_amd/ringround.py:
mulPublic, the five tunnels H0' -> H5' as modSwitch . tunnel hint . modSwitch, x (1 + x), eight leaves, 4 + 2 + 1 mul_ with div2;
indices of examples/Common.hs:49-54, moduli of examples/HomomRLWR.hs:37-43, limb counts from alch_select_limbs) -- against the
same sequence composed from the C restatement's primitives (tests/ringround_oracle.py over oracle/lol_tensor_gen.c).  The
residues are synthetic (the same seeds on both sides); what is pinned is every bit of the final ciphertexts, i.e. that the measured
pipeline computes what the oracle's composition of the reference's ops computes:
  * two ciphertexts word for word against the oracle run live;
  * a ragged batch (70 ciphertexts: the 128- and 256-thread kernel forms, several chunks of the small scratch) by whole-batch
    checksum against the per-ciphertext values the oracle produced offline (tests/golden/batch_checksums.json) -- what bench.py
    asserts for its 1024-ciphertext batch."""
import numpy as np
import pytest

from alchemy_amd.ringround import RingRound, RingRoundLanes
from helpers import load_golden
from ringround_oracle import RingRoundOracle

pytestmark = pytest.mark.gpu
MASK = (1 << 64) - 1


def test_ringround_pipeline_at_the_reference_parameters(oracle_lib):
    B = 2
    rr = RingRound(B)
    out = rr.run()
    rr.sync()
    got = out.download()
    orc = RingRoundOracle(oracle_lib)
    assert (orc.tuns, orc.muls) == (rr.tuns, rr.muls)
    # the seeded device buffers are what the oracle regenerates on the CPU
    assert np.array_equal(rr.pubs["x"].download(0, 1)[0], orc.G(rr.pubs["x"].ring.m, rr.pubs["x"].ring.qs).fill_uniform(1, 0))
    assert np.array_equal(rr.quad_src[0].download(3, 1)[0], orc.seeded(20475, rr.muls[0][1], 300 + rr.muls[0][1], 4)[3])
    for ct in range(B):
        want = orc.run(ct)
        assert np.array_equal(got[2 * ct], want[0]) and np.array_equal(got[2 * ct + 1], want[1]), ct


@pytest.mark.parametrize("B,opts", [(70, ()), (33, (("scratch_mib", 64),)), (41, (("tunnel_mac", 0),))])
def test_ringround_pipeline_ragged_batch_checksum(B, opts):
    ref = load_golden("batch_checksums.json")["homomrlwr"]
    assert ref["batch"] >= B
    rr = RingRound(B, opts)
    out = rr.run()
    rr.sync()
    want = sum(int(x, 16) for x in ref["per_ciphertext"][:B]) & MASK
    assert f"{out.checksum(0, 2 * B):016x}" == f"{want:016x}"
    out2 = rr.run()                       # the second pass replays the buffer pool: same results
    rr.sync()
    assert out2.checksum(0, 2 * B) == want


@pytest.mark.parametrize("B,lanes", [(70, 3), (9, 2), (5, 8)])
def test_ringround_lanes_equal_the_single_chain(B, lanes):
    """The batch as sub-batches on their own streams (what bench.py times): every word equals the one-chain pipeline's, and the
    checksum assembled from the parts at their positions (alch_buf_checksum_at) equals the oracle's for the whole batch."""
    ref = load_golden("batch_checksums.json")["homomrlwr"]
    rl = RingRoundLanes(B, lanes)
    assert sum(rl.sizes) == B and len(rl.lanes) == min(lanes, B) and max(rl.sizes) - min(rl.sizes) <= 1
    outs = rl.run()
    rl.sync()
    want = sum(int(x, 16) for x in ref["per_ciphertext"][:B]) & MASK
    assert f"{rl.checksum(outs):016x}" == f"{want:016x}"
    cut = B - 2                                                         # a prefix that ends inside the last lane
    assert rl.checksum(outs, cut) == sum(int(x, 16) for x in ref["per_ciphertext"][:cut]) & MASK
    if B <= 9:
        one = RingRound(B)
        o = one.run()
        one.sync()
        assert np.array_equal(rl.download(outs), o.download(0, 2 * B))
    outs2 = rl.run()                                                    # second pass: the lanes replay their buffer pools
    rl.sync()
    assert rl.checksum(outs2) == want
