"""GPU: the entry points are thread-safe (include/alchemy_hip.h: a call holds the lock of its ring's device until it returns).

A -threaded Haskell RTS forces tensors from any thread, and `GT` keeps one ring per (index, modulus list) for the whole process, so
calls on ONE ring arrive from several OS threads at once; a ring's scratch, pinned staging and buffer pool are plain members.  ctypes
releases the GIL during a call, so Python threads do overlap inside the library here.  Every thread checks its own results; without
the per-device lock the pinned staging of concurrent small transfers is the first thing to tear."""
import threading

import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi

pytestmark = pytest.mark.gpu
QS = [1543651201, 689270401, 718099201]


def rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


def _hammer(ring, seed, rounds, errors):
    try:
        rng = np.random.default_rng(seed)
        for _ in range(rounds):
            x = rand_elems(rng, 1, ring.n, ring.qs)
            b = ring.upload(x)                                   # pinned staging of a small transfer
            c = ring.alloc(1)
            c.tensor_op(b, capi.ALCH_T_CRT)                      # pooled one-element buffers
            d = ring.alloc(1)
            d.tensor_op(c, capi.ALCH_T_CRTINV)
            if not np.array_equal(d.download(), x):
                errors.append(f"thread {seed}: crtInv(crt(x)) != x")
                return
            b.free(); c.free(); d.free()
    except Exception as e:                                       # noqa: BLE001 -- reported by the main thread
        errors.append(f"thread {seed}: {e!r}")


def test_one_ring_from_eight_threads():
    ring = A.Ring(11648, QS)
    errors = []
    threads = [threading.Thread(target=_hammer, args=(ring, 100 + t, 40, errors)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_two_rings_on_one_stream_from_four_threads():
    """Two rings that share a stream (what `GT` sets up) driven from four threads, with calls between the rings mixed in."""
    small, big = A.Ring(128 * 7, QS), A.Ring(11648, QS)
    small.share_stream(big)
    rng = np.random.default_rng(7)
    x = rand_elems(rng, 1, small.n, QS)
    want = small.embed_pow(big, x[0])
    errors = []

    def embedder(seed):
        try:
            for _ in range(30):
                bx = small.upload(x)
                out = big.alloc(1)
                out.embed_from(bx, 1, capi.ALCH_BASIS_POW)
                if not np.array_equal(out.download()[0], want):
                    errors.append(f"embedder {seed}: embedPow differs")
                    return
                bx.free(); out.free()
        except Exception as e:                                   # noqa: BLE001
            errors.append(f"embedder {seed}: {e!r}")

    threads = [threading.Thread(target=_hammer, args=(big, 200, 30, errors)), threading.Thread(target=_hammer, args=(small, 201, 30, errors)),
               threading.Thread(target=embedder, args=(1,)), threading.Thread(target=embedder, args=(2,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
