"""Split rings (a limb-polynomial = two LDS-resident transforms): alch_ct_mul_relin under split_fused = 2 (two launches per chunk, tensor
product in the loaders), 1 (element-wise tensor + crtInv + fused digit kernel), 0 (every step its own kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring
CASES = [("n=2^16, six 31-bit limbs", 1 << 17, [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241], 1024),
         ("n=2^15, two 60-bit limbs", 1 << 16, [1152921504606584833, 1152921504598720513], 1024)]
for name, m, qs, B in CASES:
    res = {}
    for mode in (2, 1, 0):
        ring = Ring(m, qs)
        ring.set_option("split_fused", mode)
        a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
        a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
        hint = ring.hint_from_buf(hs)
        ring.ct_mul_relin(hint, a, b, out, B); ring.sync()
        ring.timer_start()
        for _ in range(3): ring.ct_mul_relin(hint, a, b, out, B)
        res[mode] = {"ops_per_s": round(3 * B / (ring.timer_stop() * 1e-3)), "checksum": f"{out.checksum():016x}"}
        del a, b, out, hs, hint, ring
    print(json.dumps({"case": name, "by_split_fused": res}), flush=True)
