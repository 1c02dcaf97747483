"""Shared helpers for the test-suite: layout conversions between the fixtures (limb-major lists), the C
oracle / C ABI host layout (numpy int64, shape (n, L), Lol's tuple-interleaved order) and SHA-256 digests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def to_aos(elem):
    """limb-major [L][n] python ints -> (n, L) int64 (host layout)."""
    return np.ascontiguousarray(np.array(elem, dtype=np.int64).T)


def from_aos(arr):
    return np.asarray(arr).T.tolist()


def digest_limb_major(*aos_arrays):
    h = hashlib.sha256()
    for a in aos_arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.int64).T).tobytes())
    return h.hexdigest()


def hint_to_crt_aos(ring_oracle, hint_pow):
    """fixture hint [[h0_i, h1_i]] (Pow basis, limb-major) -> list of 2*D CRT-basis (n, L) arrays in the
    order the C ABI wants: h0_0, h1_0, h0_1, h1_1, ..."""
    out = []
    for h0, h1 in hint_pow:
        out.append(ring_oracle.crt(to_aos(h0)))
        out.append(ring_oracle.crt(to_aos(h1)))
    return out
