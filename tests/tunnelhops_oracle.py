"""One hop of alchemy_amd/tunnelhops.py (BASELINE config 5: modSwitch . tunnel hint . modSwitch of examples/Tunnel.hs with BaseBGad 2
hints at the reference's indices, moduli and limb counts) replayed on the C restatement, one ciphertext at a time, with the same
seeded residues and hints.  Needs no GPU (tests/golden/make_batch_checksums.py); tests/test_gpu_tunnel.py compares it with the
device word for word."""
import math

import numpy as np

from alchemy_amd.tunnelhops import HP, SEED_KS, SEED_LIN, SEED_X, limb_counts, moduli
from helpers import oracle_tunnel


class HopOracle:
    def __init__(self, oracle_lib, k):
        from oracle import model_gen as MG
        self.O, self.k = oracle_lib, k
        self.lin, self.lh, self.lout = limb_counts()[k]
        assert self.lh <= self.lin and self.lout == self.lh          # the shapes of this config (alch_select_limbs, BaseBGad rule)
        G = oracle_lib.GenRing
        self.o_in = G(HP[k], moduli(self.lin))
        self.o_s = G(HP[k + 1], moduli(self.lh))
        self.d_rel = MG.totient(HP[k]) // MG.totient(math.gcd(HP[k], HP[k + 1]))
        self.D = sum((q - 1).bit_length() for q in moduli(self.lh))
        self.lin_h = [self.o_s.fill_uniform(SEED_LIN, e) for e in range(self.d_rel)]
        self.ks_h = [self.o_s.fill_uniform(SEED_KS, e) for e in range(2 * self.d_rel * self.D)]

    def run(self, ct):
        G, k, lin, lh = self.O.GenRing, self.k, self.lin, self.lh
        cur = [self.o_in.fill_uniform(SEED_X, 2 * ct), self.o_in.fill_uniform(SEED_X, 2 * ct + 1)]
        if lh < lin:                                                 # modSwitch down in front of the tunnel
            nxt = []
            for comp, c in enumerate(cur):
                v = self.o_in.crtinv(c)
                if comp == 0:
                    v = self.o_in.linv(v)
                for u in range(lin - lh):
                    v = G(HP[k], moduli(lin)[u:]).rescale_drop0(v)
                oo = G(HP[k], moduli(lh))
                if comp == 0:
                    v = oo.l(v)
                nxt.append(oo.crt(v))
            cur = nxt
        return oracle_tunnel(self.O, HP[k], HP[k + 1], moduli(lh), self.lin_h, self.ks_h, cur[0], cur[1], gadget="base2")
