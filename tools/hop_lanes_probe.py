"""Probe (recorded dead end): a Tunnel.hs hop as K sub-batches on K streams.  usage: tools/hop_lanes_probe.py BATCH K SHARE_STREAM (0 = a stream per ring, 1 = one stream per sub-batch, 2 = one DEDICATED stream per sub-batch)
Measured: 378 -> 383 / 180 -> 189 / 538 -> 535 / 543 -> 547 / 899 -> 868 k tunnels/s for K = 1 -> 2: two big VALU-bound kernels dominate a hop."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.tunnelhops import Hop
B = int(sys.argv[1]); K = int(sys.argv[2]); share = int(sys.argv[3]); reps = 4
for k in range(5):
    hops = [Hop(k, B // K) for _ in range(K)]
    if share:
        for h in hops:
            rs = list(h._rings.values())
            if share == 2:
                rs[0].set_option("stream_dedicated", 1)          # a hardware queue per sub-batch
            for r in rs[1:]:
                r.share_stream(rs[0])
    for h in hops: h.run()
    for h in hops:
        for r in h._rings.values(): r.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for h in hops: h.run()
    for h in hops:
        for r in h._rings.values(): r.sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"hop": k, "B": B, "K": K, "share": share, "tunnels_per_s": B / dt}), flush=True)
    del hops
