#!/usr/bin/env python3
"""Time ct_mul_relin's two kernels separately (single stream, one chunk) under the ALCH_EXP_FLAGS ablations of a -DALCH_ABLATE build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring
QS = [2147352577, 2146959361, 2146041857, 2145976321]
ring = Ring(1 << 16, QS)
ring.set_option("one_stream", 1)
B = 2048
a, b, out, hs = ring.alloc(2*B), ring.alloc(2*B), ring.alloc(2*B), ring.alloc(8)
a.fill_uniform(2); b.fill_uniform(3); hs.fill_uniform(4)
hint = ring.hint_from_buf(hs)
ring.ct_mul_relin(hint, a, b, out, B); ring.sync()
best = 1e9
for _ in range(4):
    ring.timer_start(); ring.ct_mul_relin(hint, a, b, out, B); best = min(best, ring.timer_stop())
print(f"flags={os.environ.get('ALCH_EXP_FLAGS','0'):>5s}  {best*1e3/B:7.3f} us per ciphertext (A+B)")
