#!/usr/bin/env python3
"""crt / crtInv of 128-KiB limb-polynomials: k_crt (one 16-wave workgroup per CU) against k_crt_half (two half-size
workgroups per CU).  Prints one JSON line per ring; the checksums of both forms must agree."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring

CASES = [("n=2^15, 31-bit", 1 << 16, [2147352577, 2146959361, 2146041857, 2145976321], 4096),
         ("n=2^14, 60-bit", 1 << 15, [1152921504606748673], 32768)]
for label, m, qs, elems in CASES:
    ring = Ring(m, qs)
    a = ring.alloc(elems)
    out = {"ring": label, "limb_polynomials": elems * len(qs)}
    sums = {}
    for half in (0, 1):
        ring.set_option("crt_half", half)
        a.fill_uniform(99)
        ring.sync()
        a.crt(); s1 = a.checksum(); a.crtinv(); s2 = a.checksum()
        sums[half] = (s1, s2)
        def best(fn, reps=5):
            t = 1e9
            for _ in range(reps):
                ring.timer_start(); fn(); t = min(t, ring.timer_stop())
            return t * 1e-3
        tf, ti = best(a.crt), best(a.crtinv)
        npoly = elems * len(qs)
        out[f"crt_half={half}"] = {"crt_us_per_transform": tf / npoly * 1e6, "crtinv_us_per_transform": ti / npoly * 1e6}
    out["same_results"] = sums[0] == sums[1]
    print(json.dumps(out))
    assert sums[0] == sums[1], "k_crt_half differs from k_crt"
