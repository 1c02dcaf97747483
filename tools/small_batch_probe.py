"""Probe (recorded dead end): alch_ct_mul_relin / alch_ct_mul_full at n = 2^15 on batches of 256 .. 2048 with the launch option
`chunk` 128 .. 1024 -- the persistent-grid kernels have no tails to overlap: +-2 %."""
import sys, json, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A
from alchemy_amd import capi
qs = [2147352577, 2146959361, 2146041857, 2145976321]
qh = [2144796673] + qs
def rate(fn, ring, B, reps=20):
    fn(); ring.sync()
    ring.timer_start()
    for _ in range(reps): fn()
    return reps * B / (ring.timer_stop() * 1e-3)
for B in (256, 512, 1024, 2048):
    for chunk in (1024, 512, 256, 128):
        if chunk > B: continue
        r = A.Ring(1 << 16, qs); r.set_option("chunk", chunk)
        a, b, o = r.alloc(2 * B), r.alloc(2 * B), r.alloc(2 * B)
        a.fill_uniform(1); b.fill_uniform(2)
        h = r.alloc(8); h.fill_uniform(3); hint = r.hint_from_buf(h)
        v = rate(lambda: r.ct_mul_relin(hint, a, b, o, B), r, B)
        cs = o.checksum()
        # full mul_
        rh = A.Ring(1 << 16, qh); rh.set_option("chunk", chunk)
        ro = A.Ring(1 << 16, qs[1:])
        hh = rh.alloc(10); hh.fill_uniform(4); hinth = rh.hint_from_buf(hh)
        of = ro.alloc(2 * B)
        vf = rate(lambda: capi.ct_mul_full(hinth, a, b, of, B), rh, B)
        print(json.dumps({"B": B, "chunk": chunk, "mul_relin": round(v), "mul_full": round(vf), "cs": f"{cs:016x}", "csf": f"{of.checksum():016x}"}), flush=True)
