"""GPU: rings are bound to their HIP device and may be driven from any host thread (HIP's current device is per thread; a
Haskell `safe` FFI call on a -threaded RTS arrives on an arbitrary OS thread).  A ring created on the main thread is used
from worker threads (one ring per thread at a time, as the header requires), and rings created on worker threads are used
from the main thread; results are compared with the oracle."""
import threading

import numpy as np
import pytest

import alchemy_amd as A
from conftest import CFG3_QS

pytestmark = pytest.mark.gpu


def test_rings_work_from_other_threads(oracle_lib):
    n, qs = 1 << 11, CFG3_QS
    o = oracle_lib.Ring(n, qs)
    rng = np.random.default_rng(8)
    rand = lambda c: np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(c)])
    hint, a, b = rand(8), rand(4), rand(4)
    want = [o.ct_mul_relin(list(hint), a[2 * c], a[2 * c + 1], b[2 * c], b[2 * c + 1]) for c in range(2)]
    main_ring = A.Ring(2 * n, qs)
    results, made, errors = {}, {}, []

    def use(tag, ring):
        try:
            gh, ga, gb, gout = ring.hint_load(hint), ring.upload(a), ring.upload(b), ring.alloc(4)
            ring.ct_mul_relin(gh, ga, gb, gout, 2)
            results[tag] = gout.download()
            x = ring.crt(a[0])                       # host-buffer Tensor method (per-ring scratch) from this thread
            assert np.array_equal(ring.crtinv(x), a[0])
        except Exception as e:                       # noqa: BLE001
            errors.append((tag, repr(e)))

    def make_and_use(tag):
        try:
            made[tag] = A.Ring(2 * n, qs)
        except Exception as e:                       # noqa: BLE001
            errors.append((tag, repr(e)))
            return
        use(tag, made[tag])

    t1 = threading.Thread(target=use, args=("main-ring-on-worker", main_ring))
    t1.start(); t1.join()
    ts = [threading.Thread(target=make_and_use, args=(f"worker-{i}",)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for tag, ring in list(made.items()):
        use(tag + "-on-main", ring)
    assert not errors, errors
    assert len(results) == 7
    for tag, got in results.items():
        for c in range(2):
            assert np.array_equal(got[2 * c], want[c][0]) and np.array_equal(got[2 * c + 1], want[c][1]), tag
