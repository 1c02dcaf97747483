"""The ring tunnels of the reference's Tunnel example (BASELINE config 5) on resident batches.

examples/Tunnel.hs: BaseBGad 2 hints (:24), five ~30-bit moduli (:34-39), the hops switch1..5 of examples/Common.hs:78-95 over
H0' .. H5', each emitted by PT2CT as `modSwitch_ .: tunnel_ hint .: modSwitch_` (PT2CT.hs:224-229).  Limb counts come from
alch_select_limbs with the BaseBGad rule (KSPNoise (BaseBGad 2) = p + KSAccumPNoise, PT2CT.hs:140), resolved backwards from
the output pNoise 0 of the five-hop chain: the hint may sit on FEWER limbs than the input, so the leading modSwitch can go down.
Residues and hints are synthetic (seeds below); every hop starts from a fresh seeded batch over H_k'.
Used by bench.py (extra field `tunnel_hs`), tools/bench_tunnel.py and tests/test_gpu_tunnel.py."""
from . import capi
from .capi import Ring, Tunnel

QS = [537264001, 539884801, 555609601, 560851201, 566092801]          # examples/Tunnel.hs:34-39, Zqs order
HP = [11648, 29120, 43680, 54600, 27300, 20475]                       # H0' .. H5'
SEED_LIN, SEED_KS, SEED_X = 1, 2, 3


def moduli(L):
    return list(reversed(QS[:L]))                                      # last-taken modulus outermost


def limb_counts():
    """[(L_in, L_hint, L_out)] of switch1 .. switch5 (host-only)."""
    p, tuns = 0, []
    for _ in range(5):
        lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_TUNNEL, capi.ALCH_GAD_BASE2)
        tuns.append((lin, lh, lout))
    tuns.reverse()
    return tuns


class Hop:
    """One hop H_k' -> H_k+1' with resident hint and a seeded input batch."""

    def __init__(self, k, batch, ring_opts=()):
        self.k, self.B = k, batch
        self.lin, self.lh, self.lout = limb_counts()[k]
        self._rings = {}
        self.ring_opts = tuple(ring_opts)
        lin, lh, lout = self.lin, self.lh, self.lout
        self.rin, self.rr = self.ring(HP[k], lin), self.ring(HP[k], lh)
        self.rs, self.ro = self.ring(HP[k + 1], lh), self.ring(HP[k + 1], lout)
        _, self.d_rel = Tunnel.info(self.rr, self.rs)
        self.D = self.rs.gadget_digits(capi.ALCH_GAD_BASE2)
        self.lin_buf, self.ks = self.rs.alloc(self.d_rel), self.rs.alloc(2 * self.d_rel * self.D)
        self.lin_buf.fill_uniform(SEED_LIN); self.ks.fill_uniform(SEED_KS)
        self.tun = Tunnel(self.rr, self.rs, self.lin_buf, self.ks, gadget=capi.ALCH_GAD_BASE2)
        self.x = self.rin.alloc(2 * batch)
        self.x.fill_uniform(SEED_X)
        self.up = self.rr.alloc(2 * batch) if lh != lin else None
        self.mid = self.rs.alloc(2 * batch)
        self.out = self.ro.alloc(2 * batch) if lout != lh else None

    def ring(self, m, L):
        if (m, L) not in self._rings:
            r = Ring(m, moduli(L))
            for name, v in self.ring_opts:
                r.set_option(name, v)
            self._rings[(m, L)] = r
        return self._rings[(m, L)]

    def run(self):
        """modSwitch . tunnel hint . modSwitch on the batch; returns the result buffer (CRT basis over H_k+1')."""
        src = self.x
        if self.up is not None:
            capi.ct_mod_switch(self.x, self.up, self.B); src = self.up
        self.tun.apply(src, self.mid, self.B)
        if self.out is not None:
            capi.ct_mod_switch(self.mid, self.out, self.B)
            return self.out
        return self.mid

    def measure(self, reps=1):
        self.run(); self.rs.sync()
        self.rs.timer_start()
        for _ in range(reps):
            res = self.run()
        return reps * self.B / (self.rs.timer_stop() * 1e-3), res

    def algorithmic_bytes(self):
        """Compulsory bytes of one hop at the reference's 8-byte word: one linear ciphertext in (L_in limbs over H_k'), one out
        (L_out limbs over H_k+1')."""
        return 2 * 8 * (self.lin * self.rin.n + self.lout * self.ro.n)
