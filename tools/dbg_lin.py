import sys, numpy as np
sys.path.insert(0, '.')
import alchemy_amd as A
from alchemy_amd import capi
qs_h = [2144796673, 2147352577, 2146959361, 2146041857, 2145976321]
N = 1 << 15
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2085
import json
opts = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
rh, rin, rout = A.Ring(2*N, qs_h), A.Ring(2*N, qs_h[1:]), A.Ring(2*N, qs_h[2:])
a, b, o1, o2, hs = rin.alloc(2*B), rin.alloc(2*B), rout.alloc(2*B), rout.alloc(2*B), rh.alloc(10)
for k, v in opts.items():
    rh.set_option(k, v)
a.fill_uniform(2026); b.fill_uniform(900000007); hs.fill_uniform(0xA1C4E5)
hint = rh.hint_from_buf(hs)
rh.set_option("rs_lin", 0); capi.ct_mul_full(hint, a, b, o1, B); rh.sync()
rh.set_option("rs_lin", 1); capi.ct_mul_full(hint, a, b, o2, B); rh.sync()
print("checksums", hex(o1.checksum()), hex(o2.checksum()))
bad = 0
for ct in range(B):
    x, y = o1.download(2*ct, 2), o2.download(2*ct, 2)
    if not np.array_equal(x, y):
        d = np.argwhere(x != y)
        if bad < 4: print("ct", ct, "mismatches", len(d), "first", d[:5].tolist(), "comp/limbs", sorted(set((int(e[0]), int(e[2])) for e in d)))
        bad += 1
print("bad cts:", bad)
