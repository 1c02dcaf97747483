#!/usr/bin/env python3
"""Threads per workgroup of the general-index transform kernels (launch option gen_nt) against the ring size: limb crt / crtInv rates on
every index the reference's tunnels and products run on (H0' .. H5' and the E' rings of the five hops), for 128 / 256 / 512 threads and
the library's own choice (gen_threads in kernel_gen.hpp).  One JSON line per index."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A

QS = [537264001, 539884801]
IDX = [5824, 6825, 10920, 11648, 14560, 27300, 20475, 29120, 43680, 54600]
for m in IDX:
    row = {"index": m}
    for nt in (0, 128, 256, 512):
        g = A.Ring(m, QS)
        g.set_option("gen_nt", nt)
        row["phi"] = g.n
        E = 16384
        buf = g.alloc(E); buf.fill_uniform(1)
        res = {}
        for name, fn in (("crt", buf.crt), ("crtinv", buf.crtinv)):
            fn(); g.sync()
            g.timer_start()
            for _ in range(3):
                fn()
            res[name] = round(g.timer_stop() * 1e6 / 3 / (E * len(QS)), 2)      # ns per limb-polynomial
        row["own" if nt == 0 else str(nt)] = res
        del buf, g
    print(json.dumps(row), flush=True)
