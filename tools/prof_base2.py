"""rocprofv3 target: three BaseBGad-2 key switches at the config-3 shape (per-kernel breakdown of the unfused path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring, capi
ring = Ring(1 << 16, [2147352577, 2146959361, 2146041857, 2145976321])
D = ring.gadget_digits(capi.ALCH_GAD_BASE2)
B = 32
a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * D)
a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
hint = ring.hint_from_buf(hs, capi.ALCH_GAD_BASE2)
for _ in range(3):
    ring.ct_mul_relin(hint, a, b, out, B)
ring.sync()
