"""GPU: BASELINE config 4 at the reference's REAL parameters -- the op sequence bench.py times (alchemy_amd/ringround.py:
mulPublic, the five tunnels H0' -> H5' as modSwitch . tunnel hint . modSwitch, x (1 + x), eight leaves, 4 + 2 + 1 mul_ with div2;
indices of examples/Common.hs:49-54, moduli of examples/HomomRLWR.hs:37-43, limb counts from alch_select_limbs) -- on one
ciphertext, against the same sequence composed from the C restatement's primitives (oracle/lol_tensor_gen.c through
tests/helpers.py).  The residues are synthetic (seeded on the device and downloaded); what is pinned is every bit of the final
ciphertext, i.e. that the measured pipeline computes what the oracle's composition of the reference's ops computes."""
import numpy as np
import pytest

from alchemy_amd.ringround import HP, P, RingRound, moduli
from helpers import oracle_full_mul_general, oracle_tunnel

pytestmark = pytest.mark.gpu


def test_ringround_pipeline_at_the_reference_parameters(oracle_lib):
    B = 2
    rr = RingRound(B)
    out = rr.run()
    rr.sync()
    got = out.download()

    G = oracle_lib.GenRing
    scal = lambda o, x, vals: o.scale(x, [int(v) for v in vals])
    tuns, muls = rr.tuns, rr.muls
    # ---- mulPublic a, toMSD's scalar
    L0 = tuns[0][0]
    r0 = rr.ring(HP[0], L0)
    o0 = G(HP[0], moduli(L0))
    x = rr.pubs["x"].download()
    pub = rr.public(r0, 2).download()[0]
    for ct in range(B):
        want = replay(rr, oracle_lib, [scal(o0, o0.mul(x[2 * ct + e], pub), [pow(P, -1, q) for q in moduli(L0)]) for e in range(2)])
        assert np.array_equal(got[2 * ct], want[0]) and np.array_equal(got[2 * ct + 1], want[1]), ct


def replay(rr, oracle_lib, cur):
    """The rest of the pass on one ciphertext (CRT basis over H0' after mulPublic), on the oracle."""
    G = oracle_lib.GenRing
    scal = lambda o, x, vals: o.scale(x, [int(v) for v in vals])
    tuns, muls = rr.tuns, rr.muls
    # ---- switch1 .. switch5
    for k in range(5):
        lin_, lh_, lout_ = tuns[k]
        qs = moduli(lh_)
        dup = lh_ - lin_
        mult = 1
        for q in qs[:dup]:
            mult *= q
        o_in = G(HP[k], moduli(lin_))
        up = [np.ascontiguousarray(np.concatenate([np.zeros((o_in.n, dup), dtype=np.int64),
                                                   scal(o_in, c, [mult % q for q in qs[dup:]])], axis=1)) for c in cur]
        lin, ks = (b.download() for b in rr.tunnel_src[k])
        mid = oracle_tunnel(oracle_lib, HP[k], HP[k + 1], qs, list(lin), list(ks), up[0], up[1])
        nxt = []
        for comp, c in enumerate(mid):
            os_ = G(HP[k + 1], qs)
            v = os_.crtinv(c)
            if comp == 0:
                v = os_.linv(v)                                  # c0: rescaleDec
            for u in range(lh_ - lout_):
                v = G(HP[k + 1], qs[u:]).rescale_drop0(v)
            oo = G(HP[k + 1], moduli(lout_))
            if comp == 0:
                v = oo.l(v)
            nxt.append(oo.crt(v))
        cur = nxt
    # ---- rescale tree on H5'
    m5 = HP[5]

    def product(level, a, b):
        lin_, lh_, lout_ = muls[level]
        hint = list(rr.quad_src[level].download())
        return list(oracle_full_mul_general(oracle_lib, m5, moduli(lh_), lin_, lout_, hint, a[0], a[1], b[0], b[1],
                                            [pow(P, -1, q) for q in moduli(lin_)]))

    def plus_public(src, L, seed):
        o = G(m5, moduli(L))
        v = [scal(o, c, [P % q for q in moduli(L)]) for c in src]
        v[0] = o.add(v[0], rr.public(rr.ring(m5, L), seed).download()[0])
        return v

    La = muls[0][0]
    oa = G(m5, moduli(La))
    x_lsd = [scal(oa, c, [P % q for q in moduli(La)]) for c in cur]
    y = product(0, x_lsd, plus_public(cur, La, 50))
    L1 = muls[1][0]
    t = [plus_public(y, L1, 60 + i) for i in range(8)]
    for level in (1, 2, 3):
        t = [product(level, t[2 * i], t[2 * i + 1]) for i in range(len(t) // 2)]
        ol = G(m5, moduli(muls[level][2]))
        t = [[scal(ol, c, [pow(2, -1, q) for q in moduli(muls[level][2])]) for c in ct] for ct in t]
    return t[0]
