#!/usr/bin/env python3
"""Rates of the paths beside the headline op (one JSON line each):
  * n = 2^16, six primes = 1 mod 2^17 (two-power stand-in for BASELINE configs 4 / 5): crt, crtInv and
    keySwitchQuadCirc(a*b) with the split transforms + unfused key switch;
  * BASELINE config 3 with a BaseBGad 2 hint (124 digits) instead of TrivGad."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd import Ring, capi

def best(ring, fn, reps=3):
    fn(); ring.sync()
    t = 1e9
    for _ in range(reps):
        ring.timer_start(); fn(); t = min(t, ring.timer_stop())
    return t * 1e-3

SIX = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]
ring = Ring(1 << 17, SIX)
B = 512
a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
hint = ring.hint_from_buf(hs)
t_f = best(ring, lambda: a.crt())
t_i = best(ring, lambda: a.crtinv())
t_m = best(ring, lambda: ring.ct_mul_relin(hint, a, b, out, B))
ring.set_option("split_fused", 0)
t_m0 = best(ring, lambda: ring.ct_mul_relin(hint, a, b, out, B))
ring.set_option("split_fused", 1)
polys = 2 * B * ring.L
algo = 6 * ring.L * (1 << 16) * 8
print(json.dumps({"config": "n=2^16, 6 limbs (31-bit, = 1 mod 2^17), TrivGad, CRT in/out, split transforms, digit transforms + hint products fused (k_ks_accum_split)",
                  "limb_ntt_per_s": polys / t_f, "limb_intt_per_s": polys / t_i,
                  "ntt_algorithmic_GBs": polys * 2 * 65536 * 8 / t_f / 1e9, "intt_algorithmic_GBs": polys * 2 * 65536 * 8 / t_i / 1e9,
                  "mul_relin_per_s": B / t_m, "mul_relin_per_s_composed_path": B / t_m0, "mul_relin_algorithmic_bytes": algo, "mul_relin_algorithmic_GBs": B * algo / t_m / 1e9}))
del a, b, out, hs, hint, ring

CFG3 = [2147352577, 2146959361, 2146041857, 2145976321]
ring = Ring(1 << 16, CFG3)
D = ring.gadget_digits(capi.ALCH_GAD_BASE2)
B = 64
a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * D)
a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
hint = ring.hint_from_buf(hs, capi.ALCH_GAD_BASE2)
t_m = best(ring, lambda: ring.ct_mul_relin(hint, a, b, out, B))
print(json.dumps({"config": f"BASELINE config 3 shape with a BaseBGad 2 hint ({D} digits), unfused device path",
                  "mul_relin_per_s": B / t_m, "digit_transforms_per_op": D * ring.L}))
