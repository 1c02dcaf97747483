"""Probe: how many rings can ask for a dedicated stream (option stream_dedicated).  Without a cap the HIP runtime segfaulted somewhere
after 250 such streams in one process; the library now refuses beyond 32 (ALCH_E_UNSUPPORTED, the ring keeps its ordinary stream)."""
import sys, os
sys.path.insert(0, os.getcwd())
import alchemy_amd as A
rings = []
try:
    for i in range(300):
        r = A.Ring(128 * 7, [1543651201])
        rings.append(r)
        r.set_option("stream_dedicated", 1)
        if i % 50 == 49: print("created", i + 1, flush=True)
except Exception as e:
    print("stopped at", len(rings), repr(e)[:200], flush=True)
# all still usable
import numpy as np
x = np.arange(rings[0].n, dtype=np.int64).reshape(1, -1, 1) % 97
for r in (rings[0], rings[-1]):
    b = r.upload(x); b.crt(); b.crtinv()
    assert np.array_equal(b.download(), x)
print("ok", len(rings))
