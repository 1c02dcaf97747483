#!/usr/bin/env python3
"""Profiling target: forward / inverse crt of 8192 ring elements (4 limbs) on one general index (default H1' = 29120)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A
m = int(sys.argv[1]) if len(sys.argv) > 1 else 29120
g = A.Ring(m, [1543651201, 689270401, 718099201, 720720001])
buf = g.alloc(8192); buf.fill_uniform(1)
for _ in range(2):
    buf.crt(); buf.crtinv()
g.sync()
