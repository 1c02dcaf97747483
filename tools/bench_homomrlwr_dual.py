#!/usr/bin/env python3
"""The HomomRLWR ringRound pipeline as K sub-batches on K HIP streams (alchemy_amd.ringround.RingRoundLanes: what bench.py times
with K = 2), so that the memory-bound passes and kernel tails of one dependency chain run under the VALU-bound transforms of
another.  One JSON line per run.
Usage: tools/bench_homomrlwr_dual.py [total_batch] [K] [passes]
Measured on one MI355X (profiles/r04_pipeline_lanes.jsonl): 1024 ciphertexts: K = 1 46.6 k, 2 51.8 k, 4 48.9 k, 8 48.2 k pipelines/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.ringround import RingRoundLanes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rl = RingRoundLanes(B, K)
secs, outs = rl.measure(P)
print(json.dumps({"batch": B, "sub_batches": len(rl.lanes), "passes": P, "pipelines_per_s": B / secs, "ms_per_batch": secs * 1e3,
                  "checksum_at_positions": f"{rl.checksum(outs):016x}"}))
