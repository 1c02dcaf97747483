"""GPU: the launch-structure knobs of ct_mul_relin must not change results -- multi-chunk batches on two
streams, tiny chunks, persistent workgroup grids smaller than the item count, persistent tensor kernel.
Each configuration sets the library's launch options through alch_ring_set_option (the library reads no
environment variable), runs in its own process and is compared with the oracle."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %r)
    import alchemy_amd as A
    from oracle import cref
    qs = [2147352577, 2146959361, 2146041857, 2145976321]
    logn, batch, opts = int(sys.argv[1]), int(sys.argv[2]), __import__("json").loads(sys.argv[3])
    n = 1 << logn
    g, o = A.Ring(2 * n, qs), cref.Ring(n, qs)
    for k, v in opts.items():
        g.set_option(k, v)
    rng = np.random.default_rng(5)
    rnd = lambda c: np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(c)])
    hint, a, b = rnd(8), rnd(2 * batch), rnd(2 * batch)
    gh, ga, gb, gout = g.hint_load(hint), g.upload(a), g.upload(b), g.alloc(2 * batch)
    g.ct_mul_relin(gh, ga, gb, gout, batch)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = o.ct_mul_relin(list(hint), a[2*ct], a[2*ct+1], b[2*ct], b[2*ct+1])
        assert np.array_equal(got[2*ct], w0) and np.array_equal(got[2*ct+1], w1), ct
    print("OK")
""" % ROOT)


@pytest.mark.parametrize("opts,logn,batch", [
    ({"chunk": 8}, 11, 37),                              # 5 chunks, two streams, ragged last chunk
    ({"chunk": 8, "one_stream": 1}, 11, 21),
    ({"chunk": 16, "ks_grid": 24}, 11, 40),       # persistent grid smaller than the item count
    ({"chunk": 8, "ks_grid": 8, "ti_grid": 8}, 15, 9),
    ({"ti_grid": -1}, 11, 5),
    ({}, 15, 3),
    ({"ti_split": 0}, 15, 5),                            # whole-polynomial tensor kernel instead of the split one
    ({"ti_split": 0, "ti_grid": 8, "chunk": 8}, 15, 9),
    ({"ti_split": 7, "chunk": 8}, 15, 9),         # split kernel on 7 persistent workgroups
    ({"pipe": 1, "chunk": 8}, 11, 37),                   # tensor kernels one chunk ahead of the key-switch kernels, 5 chunks, ragged tail
    ({"pipe": 1, "chunk": 8}, 15, 17),
    ({"ks_map": 1, "chunk": 16}, 11, 37), ({"ks_map": 1}, 15, 11),        # limb-per-XCD item numbering (L = 4 divides 8)
    ({"ks_rev": 1, "chunk": 16, "ks_grid": 24}, 11, 37), ({"ks_rev": 1}, 15, 11),   # items from the chunk's last ciphertext to its first
    ({"nstreams": 3, "chunk": 8}, 11, 37), ({"nstreams": 4, "chunk": 8}, 15, 33), ({"nstreams": 1, "chunk": 8}, 11, 21),   # pipelines the chunks rotate over
])
def test_launch_structure_does_not_change_results(opts, logn, batch):
    out = subprocess.run([sys.executable, "-c", WORKER, str(logn), str(batch), json.dumps(opts)], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("OK")


WORKER_FULL = textwrap.dedent("""
    import os, sys, numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import alchemy_amd as A
    from oracle import cref
    from helpers import oracle_full_mul
    qs_h = [2144468993, 2147352577, 2146959361, 2146041857, 2145976321]      # all = 1 mod 2^16
    logn, batch, opts = int(sys.argv[1]), int(sys.argv[2]), __import__("json").loads(sys.argv[3])
    n = 1 << logn
    rin, rh, rout = A.Ring(2 * n, qs_h[1:]), A.Ring(2 * n, qs_h), A.Ring(2 * n, qs_h[2:])
    for k, v in opts.items():
        rh.set_option(k, v); rin.set_option(k, v)
    rng = np.random.default_rng(6)
    rnd = lambda c, qs: np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(c)])
    hint, a, b = rnd(10, qs_h), rnd(2 * batch, qs_h[1:]), rnd(2 * batch, qs_h[1:])
    gh, ga, gb, gout = rh.hint_load(hint), rin.upload(a), rin.upload(b), rout.alloc(2 * batch)
    A.capi.ct_mul_full(gh, ga, gb, gout, batch)
    got = gout.download()
    for ct in range(batch):
        w0, w1 = oracle_full_mul(cref, n, qs_h, 4, 3, list(hint), a[2*ct], a[2*ct+1], b[2*ct], b[2*ct+1])
        assert np.array_equal(got[2*ct], w0) and np.array_equal(got[2*ct+1], w1), ct
    print("OK")
""" % (ROOT, ROOT))


@pytest.mark.parametrize("opts,logn,batch", [
    ({"chunk": 8}, 11, 37),                              # 5 chunks on two streams, ragged last chunk
    ({"chunk": 8, "one_stream": 1}, 11, 21),
    ({"chunk": 16, "ks_grid": 24, "rs_slots": 5}, 11, 40),   # persistent grids smaller than the item counts
    ({"chunk": 8, "rs_slots": 3, "ti_grid": 8}, 15, 9),
    ({"rs_half": 1}, 15, 5),                                 # closing rescale as two launches of half-size workgroups (kernel_rescale_half.hpp)
    ({"rs_half": 1, "chunk": 8}, 15, 11),                    # the same over two chunks + ragged tail
    ({"rs_lin": 0, "rs_slots": 3}, 15, 3),                   # every limb through the Pow basis (k_rescale_out)
])
def test_full_mul_launch_structure_does_not_change_results(opts, logn, batch):
    out = subprocess.run([sys.executable, "-c", WORKER_FULL, str(logn), str(batch), json.dumps(opts)], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("OK")
