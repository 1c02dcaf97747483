#!/usr/bin/env python3
"""Per-kernel timing probe on the GPU: batched crt / crtInv and ct_mul_relin at config 3 (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring

QS = [2147352577, 2146959361, 2146041857, 2145976321]
ring = Ring(1 << 16, QS)
E = 4096                     # ring elements = 16384 limb-polys
buf = ring.alloc(E)
buf.fill_uniform(1)
ring.sync()

def timeit(fn, reps=5):
    fn(); ring.sync()
    best = 1e9
    for _ in range(reps):
        ring.timer_start(); fn(); best = min(best, ring.timer_stop())
    return best

t = timeit(lambda: buf.crt())
print(f"crt    : {t:8.3f} ms for {E*4} limb-transforms -> {E*4/t/1e3:8.2f} M transforms/s, {E*4*245760/t/1e9:6.3f} T bfly/s")
t = timeit(lambda: buf.crtinv())
print(f"crtInv : {t:8.3f} ms for {E*4} limb-transforms -> {E*4/t/1e3:8.2f} M transforms/s, {E*4*245760/t/1e9:6.3f} T bfly/s")
B = 2048
a, b, out, hs = ring.alloc(2*B), ring.alloc(2*B), ring.alloc(2*B), ring.alloc(8)
a.fill_uniform(2); b.fill_uniform(3); hs.fill_uniform(4)
hint = ring.hint_from_buf(hs)
t = timeit(lambda: ring.ct_mul_relin(hint, a, b, out, B))
print(f"mul_relin (CRT in/out): {t:8.3f} ms for {B} -> {B/t*1e3:10.0f} op/s")
