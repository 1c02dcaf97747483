#!/usr/bin/env python3
"""BASELINE config 5 at the reference's real parameters: the ring tunnels of examples/Tunnel.hs -- BaseBGad 2 hints (:24), its five
~30-bit moduli (:34-39), the hops of examples/Common.hs:78-95 over H0' .. H5' -- as `modSwitch . tunnel hint . modSwitch` (PT2CT.hs:224-229)
on a batch of ciphertexts resident in HBM.  Limb counts from alch_select_limbs with the BaseBGad rule (KSPNoise (BaseBGad 2) = p + KSAccumPNoise,
PT2CT.hs:140), resolved backwards from the output pNoise 0 for the five-hop chain.  Synthetic residues and hints.  One JSON line per hop."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A
from alchemy_amd import capi

QS = [537264001, 539884801, 555609601, 560851201, 566092801]          # examples/Tunnel.hs:34-39, Zqs order
HP = [11648, 29120, 43680, 54600, 27300, 20475]
B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 2048
OPTS = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]        # launch options, e.g. tunnel_mac=0

_rings = {}
def ring(m, L):
    if (m, L) not in _rings:
        _rings[(m, L)] = A.Ring(m, list(reversed(QS[:L])))
        for k_, v_ in OPTS:
            _rings[(m, L)].set_option(k_, v_)
    return _rings[(m, L)]

p, tuns = 0, []
for _ in range(5):
    lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_TUNNEL, capi.ALCH_GAD_BASE2)
    tuns.append((lin, lh, lout))
tuns.reverse()
for k, (lin, lh, lout) in enumerate(tuns):
    rin, rr, rs, ro = ring(HP[k], lin), ring(HP[k], lh), ring(HP[k + 1], lh), ring(HP[k + 1], lout)
    _, d_rel = A.Tunnel.info(rr, rs)
    D = rs.gadget_digits(capi.ALCH_GAD_BASE2)
    lin_buf, ks = rs.alloc(d_rel), rs.alloc(2 * d_rel * D)
    lin_buf.fill_uniform(1); ks.fill_uniform(2)
    tun = A.Tunnel(rr, rs, lin_buf, ks, gadget=capi.ALCH_GAD_BASE2)
    x, up, mid, out = rin.alloc(2 * B), rr.alloc(2 * B), rs.alloc(2 * B), ro.alloc(2 * B)
    x.fill_uniform(3)

    def hop():
        src = x
        if lh != lin:                        # BaseBGad hints can sit on FEWER limbs than the input: modSwitch goes either way
            capi.ct_mod_switch(x, up, B); src = up
        tun.apply(src, mid, B)
        if lout != lh:
            capi.ct_mod_switch(mid, out, B)

    hop(); rs.sync()
    rs.timer_start(); hop(); ms = rs.timer_stop()
    print(json.dumps({"hop": f"H{k}' -> H{k+1}'", "indices": [HP[k], HP[k + 1]], "limbs_in_hint_out": [lin, lh, lout], "d_rel": d_rel,
                      "gadget": "BaseBGad 2", "digits_per_coefficient": D, "digit_transforms_per_ciphertext": d_rel * D * lh,
                      "batch": B, "tunnels_per_s": B / (ms * 1e-3)}), flush=True)
    del tun, x, up, mid, out, lin_buf, ks
