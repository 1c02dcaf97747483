import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import alchemy_amd as A
rings = [A.Ring(128 * 7, [1543651201]) for _ in range(400)]
for r in rings[1:]:
    r.share_stream(rings[0])
x = np.arange(rings[0].n, dtype=np.int64).reshape(1, -1, 1) % 97
for r in (rings[0], rings[199], rings[-1]):
    b = r.upload(x); b.crt(); b.crtinv()
    assert np.array_equal(b.download(), x)
print("400 rings on one stream: ok")
