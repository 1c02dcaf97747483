#!/usr/bin/env python3
"""Phase breakdown of k_tensor_intt from a -DALCH_STAMPS build (see tools/stamp_report.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring, load_library
QS = [2147352577, 2146959361, 2146041857, 2145976321]
ring = Ring(1 << 16, QS)
ring.set_option("one_stream", 1); ring.set_option("chunk", 2048)
B = 2048
a, b, out, hs = ring.alloc(2*B), ring.alloc(2*B), ring.alloc(2*B), ring.alloc(8)
a.fill_uniform(2); b.fill_uniform(3); hs.fill_uniform(4)
hint = ring.hint_from_buf(hs)
lib = load_library()
buf = (C.c_ulonglong * 16)()
ring.ct_mul_relin(hint, a, b, out, B); ring.sync(); lib.alch_debug_stamps_a(buf)
ring.ct_mul_relin(hint, a, b, out, B); ring.sync(); lib.alch_debug_stamps_a(buf)
names = ["loads + c2 + LDS write", "barrier", "pass 3 (stages 3-6, scalar twiddles)", "barriers between passes", "last pass + centred lift + stores", "final barrier", "pass 1 (stages 11-14, per-lane twiddles)", "pass 2 (stages 7-10)"]
wgs = B * 4
tot = sum(buf[i] for i in range(8))
print(f"k_tensor_intt: mean cycles per workgroup (stamped wave) = {tot/wgs:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {buf[i]/wgs:9.0f} cyc  {100.0*buf[i]/tot:5.1f} %")
