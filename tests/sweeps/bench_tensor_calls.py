"""Per-call rate of single Tensor methods -- crt, mulGPow, the pointwise product -- as `instance Tensor GT` issues them, in its two
representations:
  host      one ring element as a host vector, staged through the GPU on every call (alch_crt, alch_mulg_pow, alch_mul)
  resident  one ring element in HBM (alch_buf_tensor_op, alch_buf_mul on pooled single-element buffers); a call is one
            asynchronous launch, the stream is synchronised once after the timed calls
next to the CPU restatement (oracle/lol_tensor*.c: the Lol-like scalar algorithm, one thread) on the same element.
Rings: H0' = F11648 with five HomomRLWR moduli (the first hop's ciphertext ring) and n = 2^15 with the four config-3 moduli.
One JSON line per ring.  Run on the GPU box:  python tests/sweeps/bench_tensor_calls.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: F401,E402  (one HIP runtime per process: before the library)
import alchemy_amd as A  # noqa: E402
from alchemy_amd import capi  # noqa: E402
from oracle import cref  # noqa: E402

RLWR = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
CFG3 = [2147352577, 2146959361, 2146041857, 2145976321]


def rate(fn, reps, sync=None):
    fn()
    if sync:
        sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        sync()
    return reps / (time.perf_counter() - t0)


def main():
    for name, m, qs in (("H0' (phi 4608), 5 limbs", 11648, list(reversed(RLWR[:5]))), ("n = 2^15, 4 limbs", 1 << 16, CFG3)):
        r = A.Ring(m, qs)
        o = cref.GenRing(m, qs) if m != 1 << 16 else cref.Ring(1 << 15, qs)
        rng = np.random.default_rng(1)
        x = np.stack([rng.integers(0, q, size=r.n, dtype=np.int64) for q in qs], axis=1)
        y = np.stack([rng.integers(0, q, size=r.n, dtype=np.int64) for q in qs], axis=1)
        bx, by, bz = r.upload(x[None]), r.upload(y[None]), r.alloc(1)
        out = {"ring": name, "element_bytes_host": int(x.nbytes)}
        gen = m != 1 << 16
        ops = [("crt", lambda: r.crt(x), lambda: bz.tensor_op(bx, capi.ALCH_T_CRT), lambda: o.crt(x)),
               ("mul", lambda: r.mul(x, y), lambda: bz.mul(bx, by, 1), lambda: o.mul(x, y))]
        if gen:
            ops.insert(1, ("mulGPow", lambda: r.mulg_pow(x), lambda: bz.tensor_op(bx, capi.ALCH_T_MULG_POW), lambda: o.mulg_pow(x)))
        for op, host, res, cpu in ops:
            out[op] = {"host_buffer_calls_per_s": round(rate(host, 200), 1),
                       "resident_calls_per_s": round(rate(res, 5000, r.sync), 1),
                       "cpu_restatement_calls_per_s": round(rate(cpu, 20), 1)}
        # a resident call including its pooled allocation and release (what a GTDev method does)
        def chain():
            t = r.alloc(1)
            t.tensor_op(bx, capi.ALCH_T_CRT)
            t.free()
        out["crt"]["resident_with_alloc_free_calls_per_s"] = round(rate(chain, 5000, r.sync), 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
