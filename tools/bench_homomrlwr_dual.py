#!/usr/bin/env python3
"""Experiment: the HomomRLWR ringRound pipeline as K independent sub-batches on K sets of rings (= K HIP streams), so that the
memory-bound passes and kernel tails of one sub-batch run under the VALU-bound transforms of another.
Usage: tools/bench_homomrlwr_dual.py [total_batch] [K] [passes]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.ringround import RingRound

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 3
OS = os.environ.get('ONE_STREAM', '1') == '1'
rrs = [RingRound(B // K, one_stream=OS) for _ in range(K)]
for rr in rrs:
    rr.run()
for rr in rrs:
    rr.sync()
t0 = time.perf_counter()
for _ in range(P):
    outs = [rr.run() for rr in rrs]
for rr in rrs:
    rr.sync()
secs = (time.perf_counter() - t0) / P
print(json.dumps({"batch": B, "sub_batches": K, "one_stream": OS, "pipelines_per_s": B / secs, "ms_per_batch": secs * 1e3,
                  "out_checksums": [f"{o.checksum(0, 2):016x}" for o in outs]}))
