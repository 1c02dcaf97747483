"""The op sequence of alchemy_amd/ringround.py (BASELINE config 4: eval (pt2ct ringRound) of examples/HomomRLWR.hs:45-59 at the
reference's indices, moduli and limb counts) replayed on the C restatement (oracle/lol_tensor_gen.c), one ciphertext at a time,
with the SAME seeded synthetic residues and hints (alch_buf_fill_uniform == orcg_fill_uniform: word (e, j, k) =
splitmix64(seed + ((e L + j) n + k)) mod q_j).  Needs no GPU: tests/golden/make_batch_checksums.py runs it offline for the
whole-batch checksums bench.py asserts; tests/test_gpu_ringround_full.py compares it with the device word for word."""
import numpy as np

from alchemy_amd import capi
from alchemy_amd.ringround import HP, P, QS, moduli
from helpers import oracle_full_mul_general, oracle_tunnel


def limb_counts():
    p, muls, tuns = 0, [], []
    for _ in range(4):
        lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_MUL)
        muls.append((lin, lh, lout))
    for _ in range(5):
        lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_TUNNEL)
        tuns.append((lin, lh, lout))
    muls.reverse(); tuns.reverse()
    return tuns, muls


class RingRoundOracle:
    def __init__(self, oracle_lib):
        self.O = oracle_lib
        self.tuns, self.muls = limb_counts()
        self._rings, self._cache = {}, {}

    def G(self, m, qs):
        key = (m, tuple(qs))
        if key not in self._rings:
            self._rings[key] = self.O.GenRing(m, list(qs))
        return self._rings[key]

    def seeded(self, m, L, seed, count):
        """The `count` elements of a buffer of ring (m, moduli(L)) filled with `seed` (cached: hints and publics)."""
        key = (m, L, seed, count)
        if key not in self._cache:
            g = self.G(m, moduli(L))
            self._cache[key] = [g.fill_uniform(seed, e) for e in range(count)]
        return self._cache[key]

    def d_rel(self, k):
        from oracle import model_gen as MG
        import math
        return MG.totient(HP[k]) // MG.totient(math.gcd(HP[k], HP[k + 1]))

    def run(self, ct):
        """Final ciphertext (two (n, 1) CRT-basis arrays over H5') of input ciphertext number `ct` of the batch."""
        O, tuns, muls = self.O, self.tuns, self.muls
        scal = lambda o, x, vals: o.scale(x, [int(v) for v in vals])
        L0 = tuns[0][0]
        o0 = self.G(HP[0], moduli(L0))
        pub = self.seeded(HP[0], L0, 2, 1)[0]
        cur = [scal(o0, o0.mul(o0.fill_uniform(1, 2 * ct + e), pub), [pow(P, -1, q) for q in moduli(L0)]) for e in range(2)]
        for k in range(5):
            lin_, lh_, lout_ = tuns[k]
            qs = moduli(lh_)
            dup = lh_ - lin_
            mult = 1
            for q in qs[:dup]:
                mult *= q
            o_in = self.G(HP[k], moduli(lin_))
            up = [np.ascontiguousarray(np.concatenate([np.zeros((o_in.n, dup), dtype=np.int64),
                                                       scal(o_in, c, [mult % q for q in qs[dup:]])], axis=1)) for c in cur]
            d = self.d_rel(k)
            lin = self.seeded(HP[k + 1], lh_, 100 + k, d)
            ks = self.seeded(HP[k + 1], lh_, 200 + k, 2 * d * lh_)
            mid = oracle_tunnel(O, HP[k], HP[k + 1], qs, lin, ks, up[0], up[1])
            nxt = []
            for comp, c in enumerate(mid):
                os_ = self.G(HP[k + 1], qs)
                v = os_.crtinv(c)
                if comp == 0:
                    v = os_.linv(v)                                  # c0: rescaleDec
                for u in range(lh_ - lout_):
                    v = self.G(HP[k + 1], qs[u:]).rescale_drop0(v)
                oo = self.G(HP[k + 1], moduli(lout_))
                if comp == 0:
                    v = oo.l(v)
                nxt.append(oo.crt(v))
            cur = nxt
        m5 = HP[5]

        def product(level, a, b):
            lin_, lh_, lout_ = muls[level]
            hint = self.seeded(m5, lh_, 300 + lh_, 2 * lh_)
            return list(oracle_full_mul_general(O, m5, moduli(lh_), lin_, lout_, hint, a[0], a[1], b[0], b[1],
                                                [pow(P, -1, q) for q in moduli(lin_)]))

        def plus_public(src, L, seed):
            o = self.G(m5, moduli(L))
            v = [scal(o, c, [P % q for q in moduli(L)]) for c in src]
            v[0] = o.add(v[0], self.seeded(m5, L, seed, 1)[0])
            return v

        La = muls[0][0]
        oa = self.G(m5, moduli(La))
        x_lsd = [scal(oa, c, [P % q for q in moduli(La)]) for c in cur]
        y = product(0, x_lsd, plus_public(cur, La, 50))
        L1 = muls[1][0]
        t = [plus_public(y, L1, 60 + i) for i in range(8)]
        for level in (1, 2, 3):
            t = [product(level, t[2 * i], t[2 * i + 1]) for i in range(len(t) // 2)]
            ol = self.G(m5, moduli(muls[level][2]))
            t = [[scal(ol, c, [pow(2, -1, q) for q in moduli(muls[level][2])]) for c in ct_] for ct_ in t]
        return t[0]
