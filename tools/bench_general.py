#!/usr/bin/env python3
"""General-index rates on one MI355X: limb crt / crtInv, keySwitchQuadCirc(a*b), PT2CT's mul_ and one tunnel
hop on the reference's ciphertext indices H0' .. H5' (examples/Common.hs:49-54) with the HomomRLWR moduli
(examples/HomomRLWR.hs:37-43).  One JSON line per index; algorithmic bytes at the reference's 8-byte word."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A
from alchemy_amd import capi

QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
H = [11648, 29120, 43680, 54600, 27300, 20475]
SCR = int(os.environ.get('GEN_SCRATCH_MIB', '1024'))
FUSED = int(os.environ.get('GEN_FUSED', '1'))
NT = int(os.environ.get('GEN_NT', '0'))
only = [int(x) for x in sys.argv[1:]] or H


def timed(ring, fn, reps=3):
    fn(); ring.sync()
    ring.timer_start()
    for _ in range(reps):
        fn()
    return ring.timer_stop() * 1e-3 / reps


for m in only:
    L = 4
    qs = QS[:L]
    g = A.Ring(m, qs)
    g.set_option('scratch_mib', SCR); g.set_option('gen_fused', FUSED); g.set_option('gen_nt', NT)
    n, E = g.n, 8192
    buf = g.alloc(E); buf.fill_uniform(1)
    t_f, t_i = timed(g, buf.crt), timed(g, buf.crtinv)
    B = 2048
    a, b, out, hs = g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * B), g.alloc(2 * L)
    a.fill_uniform(2); b.fill_uniform(3); hs.fill_uniform(4)
    hint = g.hint_from_buf(hs)
    t_r = timed(g, lambda: g.ct_mul_relin(hint, a, b, out, B))
    qh = list(reversed(QS[:5]))
    rh, rin, rout = A.Ring(m, qh), A.Ring(m, qh[1:]), A.Ring(m, qh[2:])
    rh.set_option('scratch_mib', SCR); rh.set_option('gen_fused', FUSED)
    for r_ in (rh, rin, rout): r_.set_option('gen_nt', NT)
    hs5 = rh.alloc(10); hs5.fill_uniform(5); hint5 = rh.hint_from_buf(hs5)
    a4, b4, o3 = rin.alloc(2 * B), rin.alloc(2 * B), rout.alloc(2 * B)
    a4.fill_uniform(6); b4.fill_uniform(7)
    t_m = timed(rh, lambda: capi.ct_mul_full(hint5, a4, b4, o3, B))
    line = {"index": m, "phi": n, "limbs": L, "crt_ns_per_limb_poly": t_f / (E * L) * 1e9, "crtinv_ns_per_limb_poly": t_i / (E * L) * 1e9,
            "crt_algorithmic_GBs": E * L * 2 * n * 8 / t_f / 1e9, "mul_relin_ops_per_s": B / t_r,
            "mul_relin_algorithmic_GBs": B * 6 * L * n * 8 / t_r / 1e9, "mul_full_4_5_3_ops_per_s": B / t_m}
    if m != H[-1]:
        ms = H[H.index(m) + 1]
        gs = A.Ring(ms, qs)
        gs.set_option('gen_nt', NT)
        ep, d_rel = A.Tunnel.info(g, gs)
        lin, ks = gs.alloc(d_rel), gs.alloc(2 * d_rel * L)
        lin.fill_uniform(8); ks.fill_uniform(9)
        tun = A.Tunnel(g, gs, lin, ks)
        Bt = 512
        tout = gs.alloc(2 * Bt)
        t_t = timed(gs, lambda: tun.apply(a, tout, Bt))
        line.update({"tunnel_to": ms, "tunnel_d_rel": d_rel, "tunnel_ops_per_s": Bt / t_t})
    print(json.dumps(line), flush=True)
