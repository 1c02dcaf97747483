"""BASELINE config 3's shape (n = 2^15, 4 limbs, TrivGad, CRT in/out, B = 8192) on moduli below 2^30 -- the size of the reference's
Tunnel.hs moduli and of HomomRLWR's rounding moduli -- where 4q fits a word and the library runs Harvey's butterflies (option q30).
Prints one JSON line per variant: q30 = 1 (default on such rings) and q30 = 0 (the general kernels on the same ring)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring

QS = [1073479681, 1071513601, 1070727169, 1068236801]
B, n = 8192, 1 << 15
for q30 in (1, 0, 1):
    ring = Ring(2 * n, QS)
    ring.set_option("q30", q30)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); ring.set_option(k, int(v))
    a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
    a.fill_uniform(2026); b.fill_uniform(2027); hs.fill_uniform(0xA1C4E5)
    hint = ring.hint_from_buf(hs)
    ring.ct_mul_relin(hint, a, b, out, B); ring.ct_mul_relin(hint, a, b, out, B)
    ring.sync()
    ring.timer_start()
    for _ in range(10):
        ring.ct_mul_relin(hint, a, b, out, B)
    ms = ring.timer_stop() / 10
    ops = B / (ms * 1e-3)
    print(json.dumps({"workload": "n=2^15, L=4, moduli < 2^30, TrivGad, CRT in/out", "q30": q30, "moduli": QS, "batch": B, "ops_per_s": ops,
                      "ms_per_step": ms, "frac_of_hbm_peak": ops * 6 * 4 * n * 8 / 8e12, "out_checksum": f"{out.checksum():016x}"}), flush=True)
    del a, b, out, hs, hint, ring
