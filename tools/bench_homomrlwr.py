#!/usr/bin/env python3
"""BASELINE config 4 at the reference's real parameters: the HomomRLWR ringRound op sequence (alchemy_amd/ringround.py) on a
batch of ciphertexts resident in HBM, as `lanes` sub-batches on their own streams (RingRoundLanes; lanes=1: one dependency chain).
One JSON line: pipelines per second, and the time per stage of the first sub-batch run alone.
Usage: tools/bench_homomrlwr.py [batch] [lanes=K] [name=value ...]      (launch options, e.g. tunnel_mac=0)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.ringround import RingRoundLanes

B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 1024
NT = int(os.environ.get('GEN_NT', '0'))
TF = os.environ.get('TUNNEL_FUSED')
opts = ((('gen_nt', NT),) if NT else ()) + ((('tunnel_fused', int(TF)),) if TF is not None else ())
kv = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
lanes = dict(kv).get("lanes", 2)
opts += tuple(x for x in kv if x[0] != "lanes")
rl = RingRoundLanes(B, lanes, opts)
secs, outs = rl.measure(passes=4)
lane0 = rl.lanes[0]
lane0.stages.clear()
lane0.run(stage_times=True)
print(json.dumps({"workload": "HomomRLWR ringRound op sequence (mulPublic, 5 tunnels H0'->H5', rescale tree with 8 mul_), real indices and moduli, "
                              "synthetic residues; limb counts from alch_select_limbs", "batch": B, "sub_batches": len(rl.lanes),
                  "tunnel_limbs": rl.tuns, "mul_limbs": rl.muls,
                  "pipelines_per_s": B / secs, "ms_per_batch": secs * 1e3,
                  "stage_ms_first_sub_batch_alone": {k: v * 1e3 for k, v in lane0.stages.items()},
                  "checksum_at_positions": f"{rl.checksum(outs):016x}", "out_checksum": f"{outs[0].checksum(0, 2):016x}"}))
