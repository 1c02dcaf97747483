#!/usr/bin/env python3
"""bench.py -- ctxt-mul + relinearize throughput on MI355X (BASELINE.json's metric).

One step = one pass of the hot path (alch_ct_mul_relin: SymmSHE (*) + keySwitchQuadCirc, CRT basis in and
out) over one batch of synthetic ciphertext pairs that is already resident in HBM.
Workload: n = 2^15, 4 RNS limbs (the four largest primes < 2^31 that are
1 mod 2^16), TrivGad hint at the same modulus, B ciphertext pairs per GPU per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = ciphertext pairs all ranks processed / max-over-ranks wall time.
roofline.achieved = value/n_gpus x 6,291,456 algorithmic bytes per op (read two linear ciphertexts,
write one, at the reference's 8-byte word) measured with HIP events on the library's stream.
cpu_baseline = the C restatement of Lol's algorithm (oracle/, kind "port"), one thread, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG3_QS = [2147352577, 2146959361, 2146041857, 2145976321]
FULL_EXTRA_Q = 2144796673                              # next prime = 1 mod 2^16 below CFG3_QS: the hint's extra limb (--full)
LOGN = 15
ALGO_BYTES_PER_OP = 6 * 4 * (1 << LOGN) * 8          # 6,291,456 B  (the reference's 8-byte ZqBasic q Int64 word)
HINT_BYTES = 2 * 4 * 4 * (1 << LOGN) * 8              # 8 MiB, counted once per batch
DEVICE_WORD_BYTES_PER_OP = 6 * 4 * (1 << LOGN) * 4   # 3,145,728 B: the same six ciphertext components at the 4-byte word the device stores
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def kernel_src_sha16() -> str:
    """Fingerprint of the HIP sources the library is built from (what a committed PMC summary is valid for)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "alchemy_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(sample_ops: int):
    from oracle import cref
    ring = cref.Ring(1 << LOGN, CFG3_QS)
    secs = ring.bench_mul_relin(sample_ops, 2026)
    return {"value": sample_ops / secs, "unit": "ctxt-mul+relin/s", "cores": 1, "kind": "port",
            "sample": f"{sample_ops} ops of keySwitchQuadCirc(a*b), n=2^15, L=4, CRT basis in/out, "
                      f"single thread C restatement of Lol's CT algorithm ({secs:.1f} s) on {_cpu_model()}, "
                      f"{os.cpu_count()} logical CPUs visible"}


def cpu_baseline_all_cores(ops_per_thread: int):
    """the same restatement on every host core this process may use, one independent ciphertext
    stream per thread (ctypes releases the GIL).  Extra field only; `cpu_baseline` stays the single-thread figure."""
    import threading
    from oracle import cref
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(threads, int(os.environ.get("ALCH_CPU_THREADS", "16")))     # a one-GPU box's CPU share is 16 cores
    rings = [cref.Ring(1 << LOGN, CFG3_QS) for _ in range(threads)]
    secs = [0.0] * threads

    def work(i):
        secs[i] = rings[i].bench_mul_relin(ops_per_thread, 2026 + i)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return {"value": threads * ops_per_thread / max(secs), "unit": "ctxt-mul+relin/s", "cores": threads, "kind": "port",
            "sample": f"{ops_per_thread} ops on each of {threads} threads, slowest thread {max(secs):.1f} s"}


RING_OPTS = []
TUNNEL_HS_BATCH = 2048
MASK64 = (1 << 64) - 1


def totient(m: int) -> int:
    r, p, t = m, 2, m
    while p * p <= t:
        if t % p == 0:
            r -= r // p
            while t % p == 0:
                t //= p
        p += 1
    return r - r // t if t > 1 else r


_GOLDEN = None


def golden_checksums():
    """tests/golden/batch_checksums.json: what the C restatement computed offline for the seeded batches timed here
    (tests/golden/make_batch_checksums.py).  Every driver-timed line asserts its results against it: a wrong word in any chunk
    of any timed line ends the run.  The fixture is committed: a missing or unreadable file is a broken checkout and ends the run
    too, instead of silently turning every check off."""
    global _GOLDEN
    if _GOLDEN is None:
        path = os.path.join(ROOT, "tests", "golden", "batch_checksums.json")
        try:
            _GOLDEN = json.load(open(path))
        except (OSError, ValueError) as e:
            raise SystemExit(f"[bench] {path} is missing or unreadable ({e}): the timed lines cannot be checked against the oracle")
    return _GOLDEN


def assert_checksum(what, got, expected, detail):
    rec = {"expected": expected, "got": f"{got:016x}", "ok": f"{got:016x}" == expected, **detail}
    if not rec["ok"]:
        raise SystemExit(f"[bench] {what}: result batch differs from the oracle's: {rec}")
    return rec


def check_batch(what, ref, batch, checksum_fn, key="checksum"):
    """Assert a timed line's result batch against its fixture entry, or say WHY it was not checked (a non-default --batch, an entry the
    fixture does not hold) -- never a silent null."""
    if ref is None:
        return {"skipped": f"tests/golden/batch_checksums.json holds no entry for {what}"}
    if ref.get("batch") != batch:
        return {"skipped": f"the fixture's {what} batch is {ref.get('batch')}, this run's is {batch} (non-default --batch)"}
    return assert_checksum(what, checksum_fn(), ref[key], {"batch": batch})


def hbm_copy_GBs(ring_cls):
    """Measured streaming bandwidth of this box next to the 8 TB/s specification ("confirm on the box with a
    device-to-device copy benchmark, and report both"): alch_buf_copy of 1 GiB (hipMemcpyAsync D2D on the library's stream),
    read + write bytes over the HIP-event time, best of five."""
    ring = ring_cls(2 << LOGN, CFG3_QS)
    elems = 2048                                            # 2048 elements of 512 KiB = 1 GiB
    src, dst = ring.alloc(elems), ring.alloc(elems)
    src.fill_uniform(1)
    dst.copy_from(src, elems)
    ring.sync()
    best = 1e9
    for _ in range(5):
        ring.timer_start()
        dst.copy_from(src, elems)
        best = min(best, ring.timer_stop())
    nbytes = elems * ring.n * ring.L * ring.word_bytes
    return {"GBs": 2 * nbytes / (best * 1e-3) / 1e9, "bytes_copied": nbytes, "method": "hipMemcpyAsync device-to-device, read + write bytes, best of 5"}


def apply_opts(*rings):
    for r in rings:
        for k, v in RING_OPTS:
            r.set_option(k, v)


def general_index_line():
    """Extra line: the same op on the ring the reference's HomomRLWR example multiplies in -- H5' = F20475 =
    3^2 5^2 7 13 (examples/Common.hs:54), phi = 8640, the first four HomomRLWR moduli (examples/HomomRLWR.hs:37-43),
    TrivGad, CRT basis in/out -- through the general-index kernels (kernel_gen.hpp)."""
    from alchemy_amd import Ring
    m, qs = 20475, [1543651201, 689270401, 718099201, 720720001]
    ring = Ring(m, qs)
    apply_opts(ring)
    n, Bg = ring.n, 4096
    a, b, out, hs = ring.alloc(2 * Bg), ring.alloc(2 * Bg), ring.alloc(2 * Bg), ring.alloc(2 * ring.L)
    a.fill_uniform(11); b.fill_uniform(12); hs.fill_uniform(13)
    hint = ring.hint_from_buf(hs)
    ring.ct_mul_relin(hint, a, b, out, Bg)
    ring.sync()
    ring.timer_start()
    for _ in range(3):
        ring.ct_mul_relin(hint, a, b, out, Bg)
    ops = 3 * Bg / (ring.timer_stop() * 1e-3)
    algo = 6 * len(qs) * n * 8
    tr = ring.alloc(4096 * 2)
    tr.fill_uniform(5)
    tr.crt(); ring.sync()
    ring.timer_start()
    for _ in range(3):
        tr.crt()
    crt_s = 3 * tr.n_elems * ring.L / (ring.timer_stop() * 1e-3)
    check = check_batch("general_index", golden_checksums().get("general_index", {}).get("bench"), Bg, out.checksum)
    return {"workload": "keySwitchQuadCirc(hint, a*b), index m'=20475 (phi=8640), L=4 HomomRLWR moduli, TrivGad, CRT in/out",
            "ops_per_s": ops, "batch": Bg, "batch_checksum": check, "algorithmic_bytes_per_op": algo, "achieved_GBs": ops * algo / 1e9,
            "frac_of_hbm_peak": ops * algo / 1e9 / HBM_PEAK_GBS, "frac_at_device_word": ops * algo / 2 / 1e9 / HBM_PEAK_GBS,
            "limb_crt_per_s": crt_s, "limb_crt_algorithmic_GBs": crt_s * 2 * n * 8 / 1e9,
            "out_checksum": f"{out.checksum(0, 2):016x}"}


def q30_line(B):
    """Extra line: the headline op (same shape, same seeds) on four moduli below 2^30 -- the size of the reference's Tunnel.hs moduli
    ("good moduli, ~ 30 bits", examples/Tunnel.hs:34-39) and of HomomRLWR's rounding moduli (examples/HomomRLWR.hs:38-40).  4q fits a
    32-bit word there, so the two fused kernels run Harvey's butterflies (8 VALU instructions instead of 10, values lazy in [0,4q),
    accumulators in [0,2q)); `general_kernels_ops_per_s` is the same ring through the kernels the headline uses (option q30 = 0)."""
    from alchemy_amd import Ring
    qs = [1073479681, 1071513601, 1070727169, 1068236801]
    n = 1 << LOGN
    rates, check = {}, None
    for q30 in (1, 0):
        ring = Ring(2 * n, qs)
        apply_opts(ring)
        ring.set_option("q30", q30)
        a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
        a.fill_uniform(2026); b.fill_uniform(900_000_007); hs.fill_uniform(0xA1C4E5)
        hint = ring.hint_from_buf(hs)
        ring.ct_mul_relin(hint, a, b, out, B)
        ring.sync()
        ring.timer_start()
        for _ in range(5):
            ring.ct_mul_relin(hint, a, b, out, B)
        rates[q30] = 5 * B / (ring.timer_stop() * 1e-3)
        rec = check_batch(f"moduli below 2^30 (q30 = {q30})", golden_checksums().get("q30", {}).get("bench_mul_relin"), B, out.checksum)
        check = rec if q30 else check
        del a, b, out, hs, hint, ring
    # PT2CT's whole mul_ (4 -> 5 -> 3 limbs) on the same moduli: the UP instantiation of the Harvey key-switch kernel
    full, fref = None, golden_checksums().get("q30", {}).get("full_mul")
    if fref is not None:
        from alchemy_amd import capi
        qs_h, Bf = fref["moduli_hint"], fref["batch"]
        rh, rin, rout = Ring(2 * n, qs_h), Ring(2 * n, qs_h[1:]), Ring(2 * n, qs_h[2:])
        apply_opts(rh, rin, rout)
        a, b, fout, hs = rin.alloc(2 * Bf), rin.alloc(2 * Bf), rout.alloc(2 * Bf), rh.alloc(2 * rh.L)
        a.fill_uniform(2026); b.fill_uniform(900_000_007); hs.fill_uniform(0xA1C4E5)
        hint_h = rh.hint_from_buf(hs)
        capi.ct_mul_full(hint_h, a, b, fout, Bf)
        rh.sync()
        rh.timer_start()
        for _ in range(3):
            capi.ct_mul_full(hint_h, a, b, fout, Bf)
        fops = 3 * Bf / (rh.timer_stop() * 1e-3)
        full = {"ops_per_s": fops, "limbs": fref["limbs"], "batch": Bf,
                "batch_checksum": assert_checksum("mul_ on moduli below 2^30", fout.checksum(), fref["checksum"], {"batch": Bf})}
        del a, b, fout, hs, hint_h, rh, rin, rout
    algo = 6 * len(qs) * n * 8
    return {"full_mul": full,
            "workload": "BASELINE config 3's op and shape (n=2^15, 4 limbs, TrivGad, CRT in/out) on moduli below 2^30: Harvey butterflies",
            "moduli": qs, "batch": B, "ops_per_s": rates[1], "general_kernels_ops_per_s": rates[0], "batch_checksum": check,
            "algorithmic_bytes_per_op": algo, "frac_of_hbm_peak": rates[1] * algo / 1e9 / HBM_PEAK_GBS,
            "frac_at_device_word": rates[1] * algo / 2 / 1e9 / HBM_PEAK_GBS}


def n16_line():
    """Extra line: the headline op at n = 2^16 on six limbs (SURVEY 8d's two-power stand-in for BASELINE configs 4 / 5's wording, six primes
    that are 1 mod 2^17): a limb-polynomial is 256 KiB, twice an LDS-resident transform, so the transforms are split (kernel_crt_split.hpp);
    two launches per chunk since round 3 (k_tensor_crtinv_split, k_ks_accum_split<FROM_OPS>).  Whole batch asserted against the oracle."""
    from alchemy_amd import Ring
    qs = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]
    n, B = 1 << 16, 2048
    ring = Ring(2 * n, qs)
    apply_opts(ring)
    a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
    a.fill_uniform(2026); b.fill_uniform(900_000_007); hs.fill_uniform(0xA1C4E5)
    hint = ring.hint_from_buf(hs)
    ring.ct_mul_relin(hint, a, b, out, B)
    ring.sync()
    ring.timer_start()
    for _ in range(3):
        ring.ct_mul_relin(hint, a, b, out, B)
    ops = 3 * B / (ring.timer_stop() * 1e-3)
    check = check_batch("n = 2^16, six limbs", golden_checksums().get("n16", {}).get("bench_mul_relin"), B, out.checksum)
    algo = 6 * len(qs) * n * 8
    return {"workload": "keySwitchQuadCirc(hint, a*b), n=2^16, 6 limbs (31-bit primes = 1 mod 2^17), TrivGad, CRT in/out, split transforms",
            "ops_per_s": ops, "batch": B, "batch_checksum": check, "algorithmic_bytes_per_op": algo, "achieved_GBs": ops * algo / 1e9,
            "frac_of_hbm_peak": ops * algo / 1e9 / HBM_PEAK_GBS, "frac_at_device_word": ops * algo / 2 / 1e9 / HBM_PEAK_GBS}


def tunnel_hs_line():
    """Extra line: BASELINE config 5 at the reference's real parameters -- the five hops of examples/Tunnel.hs (BaseBGad 2 hints,
    its moduli, H0' .. H5'), each as modSwitch . tunnel hint . modSwitch on 2048 resident ciphertexts (alchemy_amd/tunnelhops.py);
    every hop's whole result batch is checked against the C restatement's checksums.  (Round 3 first timed 256 ciphertexts, one pass:
    that batch leaves the element-wise kernels of a hop launch-bound -- 248/141/330/417/629 k hops/s against 297/162/434/485/778 k at
    2048 on the same box, tools/hop_batch_probe.py.)"""
    from alchemy_amd.tunnelhops import Hop
    ref = {h["hop"]: h for h in golden_checksums().get("tunnel_hs", {}).get("hops", [])}
    hops = []
    for k in range(5):
        hop = Hop(k, TUNNEL_HS_BATCH, RING_OPTS)
        rate, res = hop.measure(reps=3)
        check = {"skipped": f"the fixture holds no hop {k}"}
        if k in ref:                                        # per-ciphertext checksums are position dependent, so any prefix adds up
            cnt = min(hop.B, ref[k]["batch"])
            want = sum(int(x, 16) for x in ref[k]["per_ciphertext"][:cnt]) & MASK64
            check = assert_checksum(f"tunnel_hs hop {k}", res.checksum(0, 2 * cnt), f"{want:016x}", {"ciphertexts_checked": cnt, "batch": hop.B})
        algo = hop.algorithmic_bytes()
        hops.append({"hop": f"H{k}' -> H{k + 1}'", "limbs_in_hint_out": [hop.lin, hop.lh, hop.lout], "d_rel": hop.d_rel,
                     "digits_per_coefficient": hop.D, "tunnels_per_s": rate, "algorithmic_bytes_per_tunnel": algo,
                     "frac_of_hbm_peak": rate * algo / 1e9 / HBM_PEAK_GBS, "batch_checksum": check})
        del hop
    return {"workload": "examples/Tunnel.hs hops: modSwitch . tunnel hint . modSwitch, BaseBGad 2 hints, Tunnel.hs moduli, limb counts "
                        "from alch_select_limbs, %d ciphertexts resident in HBM, synthetic residues" % TUNNEL_HS_BATCH, "hops": hops}


def config2_line():
    """Extra line: BASELINE config 2 -- n = 2^14, one limb (60-bit q: 64-bit device words; and the 31-bit reference-compatible point):
    forward / inverse transform and pointwise product rates on a resident batch, fractions at the word the device moves.
    Checked: crt and the pointwise product of the first polynomials against the C restatement, crtInv(crt(x)) == x on the batch."""
    from alchemy_amd import Ring
    n, polys = 1 << 14, 32768
    ref = golden_checksums().get("config2", {})
    out = {"workload": f"BASELINE config 2: n=2^14, 1 limb, {polys} polynomials resident in HBM; crt, crtInv, pointwise product",
           "algorithmic_bytes_at_8_byte_words": {"transform": 2 * n * 8, "pointwise_mul": 3 * n * 8}}
    for label, q in (("q60", 1152921504606748673), ("q31", 2147352577)):
        ring = Ring(2 * n, [q])
        apply_opts(ring)
        a, b, c = ring.alloc(polys), ring.alloc(polys), ring.alloc(polys)
        a.fill_uniform(2026); b.fill_uniform(7)
        before = a.checksum()

        def best(fn, reps=3):
            fn(); ring.sync()
            t = 1e9
            for _ in range(reps):
                ring.timer_start(); fn(); t = min(t, ring.timer_stop())
            return t * 1e-3
        # timing leaves `a` transformed an even number of times in each direction: 4 forward, then 4 inverse
        t_f = best(lambda: a.crt())
        t_i = best(lambda: a.crtinv())
        if a.checksum() != before:
            raise SystemExit(f"[bench] config2 {label}: crtInv^4(crt^4(x)) != x")
        a.crt()
        t_m = best(lambda: c.mul(a, b, polys))
        check = {"skipped": f"the fixture holds no config2 entry for {label}"}
        pre = ref.get("prefix")
        if label in ref and pre and pre <= polys:
            check = {"crt": assert_checksum(f"config2 {label} crt", a.checksum(0, pre), ref[label]["crt"], {"polynomials": pre}),
                     "crt_times_b": assert_checksum(f"config2 {label} mul", c.checksum(0, pre), ref[label]["crt_times_b"], {"polynomials": pre})}
        wb = ring.word_bytes
        dev = lambda words, t: polys * words * n * wb / t / 1e9
        out[label] = {"modulus": q, "device_word_bytes": wb, "ntt_per_s": polys / t_f, "intt_per_s": polys / t_i,
                      "pointwise_mul_per_s": polys / t_m,
                      "GBs_at_device_word": {"ntt": dev(2, t_f), "intt": dev(2, t_i), "mul": dev(3, t_m)},
                      "frac_of_hbm_peak_at_device_word": {"ntt": dev(2, t_f) / HBM_PEAK_GBS, "intt": dev(2, t_i) / HBM_PEAK_GBS,
                                                          "mul": dev(3, t_m) / HBM_PEAK_GBS},
                      "checks": check}
        del a, b, c, ring
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192, help="ciphertext pairs per GPU per step (weak scaling)")
    ap.add_argument("--cpu-ops", type=int, default=768, help="ops in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-pow", dest="pow", action="store_false",
                    help="skip the Pow-basis in/out variant (SURVEY 8d's COEFF line; extra field, after the timed region)")
    ap.add_argument("--no-full", dest="full", action="store_false",
                    help="skip PT2CT's whole mul_ (modSwitch . keySwitchQuad . modSwitch . (*)), 4 -> 5 -> 3 limbs (extra field)")
    ap.add_argument("--no-general", dest="general", action="store_false",
                    help="skip the general-index line (keySwitchQuadCirc(a*b) on the reference's H5' = F20475 ring)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="skip the HomomRLWR ringRound pipeline (BASELINE config 4 at the reference's indices and moduli; every rank "
                         "runs its own 1024-ciphertext shard, extra field `homomrlwr`)")
    ap.add_argument("--pipeline-batch", type=int, default=1024, help="ciphertexts per GPU in the HomomRLWR pipeline")
    ap.add_argument("--pipeline-lanes", type=int, default=2, help="sub-batches (own streams) the pipeline's shard runs as")
    ap.add_argument("--no-tunnel-hs", dest="tunnel_hs", action="store_false",
                    help="skip the examples/Tunnel.hs hops (BASELINE config 5: BaseBGad 2 hints, extra field `tunnel_hs`)")
    ap.add_argument("--no-n16", dest="n16", action="store_false",
                    help="skip the extra line: the headline op at n = 2^16 on six limbs (split transforms)")
    ap.add_argument("--no-q30", dest="q30", action="store_false",
                    help="skip the extra line: the headline op on moduli below 2^30 (Harvey-butterfly kernels)")
    ap.add_argument("--no-config2", dest="config2", action="store_false",
                    help="skip BASELINE config 2 (n = 2^14, one limb: transform and pointwise rates, extra field `config2`)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="launch-structure option for every ring (alch_ring_set_option), e.g. one_stream=1 for kernel traces "
                         "whose durations add up to the step time")
    args = ap.parse_args()
    global RING_OPTS
    RING_OPTS = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt]

    import torch
    from alchemy_amd import Ring, shard

    # ALCH_DIST_BACKEND=gloo + ALCH_FORCE_DEVICE=0 rehearse the multi-rank path on a one-GPU box (several ranks
    # sharing cuda:0); the driver's real runs use nccl (= RCCL) with one rank per GPU.
    rank, local_rank, world, dist = shard.init_distributed(os.environ.get("ALCH_DIST_BACKEND"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearsal = os.environ.get("ALCH_DIST_BACKEND") == "gloo"
    if world > 1 and not rehearsal and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks but only {torch.cuda.device_count()} GPUs visible: one rank per GPU")
    dev_index = int(os.environ.get("ALCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    torch.empty(1, device=dev)          # make this rank's device current in the shared HIP runtime before the library binds to it
    red_dev = None if os.environ.get("ALCH_DIST_BACKEND") == "gloo" else dev

    ring = Ring(2 << LOGN, CFG3_QS)
    apply_opts(ring)
    B = args.batch
    a, b, out = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B)
    hint_src = ring.alloc(2 * ring.L)
    # seeds: 2026 + global ciphertext offset of this rank, hint 0xA1C4E5 (identical on every rank)
    sh = shard.partition(B * world, world, rank)
    a.fill_uniform(2026 + 2 * sh.first * 7919)
    b.fill_uniform(900_000_007 + 2 * sh.first * 7919)
    # The hint is generated once (rank 0, seed 0xA1C4E5) and broadcast over RCCL before anything is timed -- the
    # one collective of the path.  A failing broadcast is fatal: a bench line from ranks that do not
    # share one hint would not be a measurement of the path.
    hint_dist = "single rank"
    hint_src.fill_uniform(0xA1C4E5)
    if dist is not None:
        h = hint_src.download()
        if rank != 0:
            h[...] = 0
        if os.environ.get("ALCH_TEST_FAIL_BROADCAST"):      # tests: prove that a broken collective ends the run
            raise SystemExit("[bench] hint broadcast failed (forced by ALCH_TEST_FAIL_BROADCAST)")
        shard.broadcast_array(h, dist, src=0, device=red_dev)
        if rank != 0 and not h.any():
            raise SystemExit("[bench] hint broadcast delivered zeros")
        hint_src.upload(h)
        hint_dist = "rccl broadcast from rank 0" if red_dev is not None else "gloo broadcast from rank 0"
    hint = ring.hint_from_buf(hint_src)
    ring.sync()

    def step():
        ring.ct_mul_relin(hint, a, b, out, B)

    for _ in range(args.warmup):
        step()
    ring.sync()
    torch.cuda.synchronize()
    shard.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ring.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ring.timer_stop()
    torch.cuda.synchronize()
    shard.barrier(dist)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall_max = shard.max_over_ranks(wall, dist, red_dev)
    ev_ms_max = shard.max_over_ranks(ev_ms, dist, red_dev)
    ev_ms_ranks = shard.gather_scalars(ev_ms, dist, red_dev)     # one entry per rank: a straggler is visible
    checksum = out.checksum(0, 2)
    # whole-batch result check on rank 0 (its shard always starts at ciphertext 0, so the seeds are those of the
    # committed fixture): alch_buf_checksum of every result word against the value the C oracle produced offline
    # (tests/golden/batch_checksums.json, generated by tests/golden/make_batch_checksums.py)
    batch_check = None
    if rank == 0:
        batch_check = check_batch("the headline batch", golden_checksums().get("bench_mul_relin"), B, out.checksum)

    # The batch gather (north star: "RCCL over xGMI for the batch gather only"): after timing, every rank
    # contributes a slice of its result batch to an all-gather, zero-copy from the library's buffer.  Reported next
    # to the throughput, never inside it: gathering every result would be bound by xGMI ingress (7 x ~153 GB/s per
    # GPU), far below what the ranks produce.
    gather = None
    if dist is not None:            # any exception here ends the run with a non-zero exit code: no healthy-looking line from a broken RCCL path
        G_CTS = min(B, 256)                               # 256 ciphertexts = 256 MiB of device words per rank
        mine = out.as_torch(0, 2 * G_CTS)
        everyone = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        ring.sync()

        def all_gather():
            if red_dev is not None:
                dist.all_gather_into_tensor(everyone, mine)                  # RCCL
            else:
                dist.all_gather(list(everyone.chunk(world)), mine)           # gloo rehearsal

        all_gather()
        torch.cuda.synchronize()
        ok = bool(torch.equal(everyone[rank * mine.numel():(rank + 1) * mine.numel()], mine))
        if not ok:
            raise SystemExit("[bench] all-gather returned a different slice than this rank contributed")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        shard.barrier(dist)
        e0.record()
        for _ in range(5):
            all_gather()
        e1.record()
        torch.cuda.synchronize()
        ms = shard.max_over_ranks(e0.elapsed_time(e1) / 5, dist, red_dev)
        nbytes = mine.numel() * mine.element_size()
        gather = {"ciphertexts_per_rank": G_CTS, "bytes_per_rank": nbytes, "ms": ms,
                  "ingress_GBs_per_gpu": (world - 1) * nbytes / (ms * 1e-3) / 1e9,
                  "xgmi_ingress_bound_GBs": 7 * 153.0, "own_slice_intact": ok,
                  "backend": "rccl" if red_dev is not None else "gloo"}
        del everyone

    pow_ops = pow_check = None
    if args.pow and rank == 0:
        from alchemy_amd.capi import ALCH_POW_IN, ALCH_POW_OUT
        Bp = min(B, 2048)
        ring.ct_mul_relin(hint, a, b, out, Bp, flags=ALCH_POW_IN | ALCH_POW_OUT)
        ring.sync()
        ring.timer_start()
        for _ in range(3):
            ring.ct_mul_relin(hint, a, b, out, Bp, flags=ALCH_POW_IN | ALCH_POW_OUT)
        pow_ops = 3 * Bp / (ring.timer_stop() * 1e-3)
        pow_check = check_batch("Pow-basis in/out", golden_checksums().get("bench_extra", {}).get("pow_in_out"), Bp, lambda: out.checksum(0, 2 * Bp))

    full = None
    if args.full and rank == 0:
        # operands on 4 limbs, TrivGad hint one limb longer (KSPNoise, PT2CT.hs:139), result on 3
        from alchemy_amd import capi
        qs_h = [FULL_EXTRA_Q] + CFG3_QS
        rh, rout = Ring(2 << LOGN, qs_h), Ring(2 << LOGN, CFG3_QS[1:])
        apply_opts(rh, rout)
        Bf = min(B, 4096)
        hsrc = rh.alloc(2 * rh.L)
        hsrc.fill_uniform(0xA1C4E5)
        hint_h = rh.hint_from_buf(hsrc)
        fout = rout.alloc(2 * Bf)
        rh.sync()
        capi.ct_mul_full(hint_h, a, b, fout, Bf)
        rh.sync()
        rh.timer_start()
        for _ in range(3):
            capi.ct_mul_full(hint_h, a, b, fout, Bf)
        ops = 3 * Bf / (rh.timer_stop() * 1e-3)
        # compulsory bytes at 8-byte words: two 4-limb linear ciphertexts in, one 3-limb out
        algo = (2 * 2 * 4 + 2 * 3) * (1 << LOGN) * 8
        full = {"ops_per_s": ops, "workload": "PT2CT mul_: (*) on 4 limbs, modSwitch to the 5-limb hint modulus, "
                "keySwitchQuadCirc, modSwitch to 3 limbs; CRT-basis in/out", "moduli_hint": qs_h, "batch": Bf,
                "algorithmic_bytes_per_op": algo, "achieved_GBs": ops * algo / 1e9,
                "frac_of_hbm_peak": ops * algo / 1e9 / HBM_PEAK_GBS,
                "frac_at_device_word": ops * (algo // 2) / 1e9 / HBM_PEAK_GBS,
                "batch_checksum": check_batch("full_mul", golden_checksums().get("bench_extra", {}).get("full_mul"), Bf, fout.checksum),
                "out_checksum": f"{fout.checksum(0, 2):016x}"}
        del fout, hint_h, hsrc

    general = None
    if args.general and rank == 0:
        general = general_index_line()

    tunnel_hs = config2 = q30 = n16 = None
    if rank == 0 and (args.tunnel_hs or args.config2 or args.q30 or args.n16):
        del a, b, out
        a = b = out = None
        if args.q30:
            q30 = q30_line(B)
        if args.n16:
            n16 = n16_line()
        if args.tunnel_hs:
            tunnel_hs = tunnel_hs_line()
        if args.config2:
            config2 = config2_line()

    homomrlwr = None
    if args.pipeline:
        # BASELINE config 4: "examples/HomomRLWR.hs pipeline, 8192-ciphertext batch sharded over 8 GPUs" = 1024 ciphertexts per
        # GPU.  Every rank runs the whole op sequence on its own shard (no collective inside), bracketed like the headline.
        from alchemy_amd.ringround import RingRoundLanes
        a = b = out = None
        Bp = args.pipeline_batch
        # the shard runs as two sub-batches on two streams (RingRoundLanes: +11 % over one chain of 1024, same result words)
        rr = RingRoundLanes(Bp, args.pipeline_lanes, RING_OPTS)
        rr.run(); rr.sync()                                    # allocations, first touch
        torch.cuda.synchronize()
        shard.barrier(dist)
        t0 = time.perf_counter()
        PASSES = 4
        for _ in range(PASSES):
            res = rr.run()
        rr.sync()
        torch.cuda.synchronize()
        shard.barrier(dist)
        secs = shard.max_over_ranks((time.perf_counter() - t0) / PASSES, dist, red_dev)
        if rank == 0:
            # whole-batch check of the timed pass: rank 0's shard starts at ciphertext 0, so the committed per-ciphertext checksums
            # of the C restatement's replay (tests/ringround_oracle.py) apply to its first min(Bp, fixture batch) results
            ref = golden_checksums().get("homomrlwr")
            pipe_check = {"skipped": "the fixture holds no homomrlwr entry"}
            if ref is not None:
                cnt = min(Bp, ref["batch"])
                want = sum(int(x, 16) for x in ref["per_ciphertext"][:cnt]) & MASK64
                pipe_check = assert_checksum("homomrlwr", rr.checksum(res, cnt), f"{want:016x}", {"ciphertexts_checked": cnt, "batch": Bp})
            lane0 = rr.lanes[0]
            lane0.stages.clear()
            lane0.run(stage_times=True)
            # compulsory bytes at the reference's 8-byte word: end to end (one linear ciphertext in over H0', one out over H5') and
            # summed over the 13 heavy ops (each tunnel / mul_ reads its operands and writes its result once)
            from alchemy_amd.ringround import HP as _HP
            phi = {m: totient(m) for m in _HP}
            e2e = 2 * 8 * (rr.tuns[0][0] * phi[_HP[0]] + rr.muls[3][2] * phi[_HP[5]])
            per_op = sum(2 * 8 * (li * phi[_HP[k]] + lo * phi[_HP[k + 1]]) for k, (li, _, lo) in enumerate(rr.tuns))
            mult = [1, 4, 2, 1]
            per_op += sum(cnt_ * 2 * 8 * (2 * li + lo) * phi[_HP[5]] for cnt_, (li, _, lo) in zip(mult, rr.muls))
            rate = Bp * world / secs
            homomrlwr = {"workload": "HomomRLWR ringRound op sequence: mulPublic, 5 ring tunnels H0' -> H5', rescale tree with 8 mul_ "
                         "(examples/HomomRLWR.hs:45-59), real indices and moduli, limb counts from alch_select_limbs, synthetic residues",
                         "ciphertexts_per_gpu": Bp, "sub_batches": len(rr.lanes), "n_gpus": world, "pipelines_per_s": rate, "ms_per_batch": secs * 1e3,
                         "batch_checksum": pipe_check,
                         "algorithmic_bytes_per_pipeline": {"end_to_end": e2e, "sum_over_the_13_heavy_ops": per_op},
                         "frac_of_hbm_peak": {"end_to_end": rate / world * e2e / 1e9 / HBM_PEAK_GBS,
                                              "sum_over_the_13_heavy_ops": rate / world * per_op / 1e9 / HBM_PEAK_GBS},
                         "tunnel_limbs": rr.tuns, "mul_limbs": rr.muls,
                         "stage_ms_rank0_first_sub_batch_alone": {k: round(v * 1e3, 3) for k, v in lane0.stages.items()},
                         "out_checksum": f"{res[0].checksum(0, 2):016x}"}
        del rr

    if rank == 0:
        copy_bw = hbm_copy_GBs(Ring)
        total_ops = B * world * args.steps
        value = total_ops / wall_max
        # per-GPU achieved algorithmic bandwidth from the HIP-event time of the K launches on the stream
        per_gpu_ops_s = B * args.steps / (ev_ms_max * 1e-3)
        achieved = per_gpu_ops_s * (ALGO_BYTES_PER_OP + HINT_BYTES / B) / 1e9
        # physical HBM-side bytes per op from the committed PMC passes (profiles/traffic_latest.json: separate
        # FETCH_SIZE and WRITE_SIZE runs of this script, gfx950 FETCH_SIZE correction applied), turned into
        # a rate with this run's op rate so that it compares with `achieved`
        traffic = traffic_per_op = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # only a PMC summary taken on these very kernel sources counts; a stale one is reported as null
                if tj.get("kernel_src_sha16") == kernel_src_sha16():
                    traffic_per_op = float(tj["hbm_bytes_per_op"])
                    traffic = per_gpu_ops_s * traffic_per_op / 1e9
                    traffic_src = {"file": "profiles/traffic_latest.json", "kernel_src_sha16": tj["kernel_src_sha16"],
                                   "collected": tj.get("collected")}
                else:
                    traffic_src = {"file": "profiles/traffic_latest.json", "stale": True,
                                   "kernel_src_sha16_of_file": tj.get("kernel_src_sha16"), "kernel_src_sha16_now": kernel_src_sha16()}
            except Exception:
                traffic = traffic_per_op = None
        line = {
            "metric": "ctxt-mul+relinearize/sec at n=2^15, 4 RNS limbs",
            "value": value, "unit": "ctxt-mul+relin/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: keySwitchQuadCirc(hint, a*b) on linear ciphertexts, "
                                   "n=2^15 (m'=2^16), L=4 primes<2^31, TrivGad hint at the same modulus, "
                                   "CRT-basis in/out, inputs resident in HBM",
                       "batch_per_gpu": B, "global_batch": B * world, "sharding": f"dp{world} by ciphertext, no collective in the timed region", "hint": hint_dist,
                       "moduli": CFG3_QS, "device_word_bytes": ring.word_bytes},
            "roofline": {"bound": "hbm", "limiter": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "hbm_copy_GBs": copy_bw["GBs"], "hbm_copy": copy_bw,
                         "frac_of_measured_copy": achieved / copy_bw["GBs"],
                         "frac_at_device_word": per_gpu_ops_s * (DEVICE_WORD_BYTES_PER_OP + HINT_BYTES / 2 / B) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_bytes_per_op": traffic_per_op, "traffic_source": traffic_src,
                         "note": "frac prices an op at SURVEY 8d's 6,291,456 B (six ciphertext components at the reference's "
                                 "8-byte Int64 word); frac_at_device_word prices the same components at the 4-byte word the "
                                 "device actually moves (3,145,728 B) -- the honest fraction of the 8 TB/s peak for this "
                                 "build.  bound names the roofline SURVEY 8d prescribes; limiter says what the kernels are "
                                 "actually bound by (integer VALU issue, DESIGN.md 4).  hbm_copy_GBs = what a plain device-to-device "
                                 "copy reaches on this very box in this very run (SURVEY 8d asks for it next to the 8 TB/s "
                                 "specification); frac_of_measured_copy = achieved / that.  traffic = physical GB/s = PMC "
                                 "(FETCH_SIZE x2 + WRITE_SIZE) bytes per op x this run's ops/s, null when the committed PMC "
                                 "summary was taken on other kernel sources"},
            "hip_event_ms_per_step": ev_ms_max / args.steps,
            "hip_event_ms_per_step_by_rank": [x / args.steps for x in ev_ms_ranks],
            "batch_checksum": batch_check,
            "out_checksum": f"{checksum:016x}",
        }
        if pow_ops is not None:
            line["pow_basis_in_out_ops_per_s"] = pow_ops
            line["pow_basis_in_out"] = {"ops_per_s": pow_ops, "batch": min(B, 2048), "batch_checksum": pow_check}
        if full is not None:
            line["full_mul"] = full
        if general is not None:
            line["general_index"] = general
        if homomrlwr is not None:
            line["homomrlwr"] = homomrlwr
        if tunnel_hs is not None:
            line["tunnel_hs"] = tunnel_hs
        if config2 is not None:
            line["config2"] = config2
        if q30 is not None:
            line["moduli_below_2_30"] = q30
        if n16 is not None:
            line["n16_six_limbs"] = n16
        line["result_gather"] = gather
        if world == 1 and args.cpu_ops > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_ops)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(max(8, args.cpu_ops // 4))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
