#!/usr/bin/env python3
"""Phase breakdown of k_ks_accum_half from the diagnostic (-DALCH_STAMPS) build:
   ALCH_LIB_PATH=alchemy_amd/lib/variants/stamps.so python tools/stamp_report.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring, load_library
QS = [2147352577, 2146959361, 2146041857, 2145976321]
ring = Ring(1 << 16, QS)
B = 8192
a, b, out, hs = ring.alloc(2*B), ring.alloc(2*B), ring.alloc(2*B), ring.alloc(8)
a.fill_uniform(2); b.fill_uniform(3); hs.fill_uniform(4)
hint = ring.hint_from_buf(hs)
lib = load_library()
buf = (C.c_ulonglong * 16)()
ring.ct_mul_relin(hint, a, b, out, B); ring.sync()
lib.alch_debug_stamps(buf)
ring.timer_start(); ring.ct_mul_relin(hint, a, b, out, B); ms = ring.timer_stop()
lib.alch_debug_stamps(buf)
names = ["tensor init (loads+pointwise)", "barrier before G", "pass G (global ld, st 0-2, LDS wr)", "barrier after G", "LDS pass 1",
         "barrier", "LDS pass 2", "barrier", "last pass + hint MAC", "(loop exit)", "result stores"]
tot = sum(buf[i] for i in range(11))
wgs = B * 8
print(f"{ms:.3f} ms for {B} cts; {wgs} workgroups; mean cycles per workgroup (wave 0) = {tot/wgs:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:38s} {buf[i]/wgs:10.0f} cyc  {100.0*buf[i]/tot:5.1f} %")
if buf[12]:
    print(f"  shader clock inside the kernel: {buf[11] / (buf[12] / 100e6) / 1e9:.3f} GHz  (s_memtime / s_memrealtime)")
