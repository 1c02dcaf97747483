#!/usr/bin/env python3
"""Experiment: HomomRLWR pipeline rate against the scratch budget (= ciphertexts per chunk of the tunnel / mul_ kernels):
do small chunks, whose digit scratch stays inside the 256 MB Infinity Cache, beat large ones?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.ringround import RingRound
for mib in [int(x) for x in sys.argv[1:]] or (4096, 1024, 512, 256, 128, 4096):
    rr = RingRound(1024, (("scratch_mib", abs(mib)),), pow_handoff=mib > 0)      # negative budget: CRT-basis hand-off between hops
    secs, out = rr.measure(passes=2)
    print(mib, round(1024 / secs), f"{out.checksum(0, 2):016x}", flush=True)
    del rr
