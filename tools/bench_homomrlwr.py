#!/usr/bin/env python3
"""BASELINE config 4 at the reference's real parameters: the HomomRLWR ringRound op sequence (alchemy_amd/ringround.py) on a
batch of ciphertexts resident in HBM.  One JSON line: pipelines per second and the time per stage.
Usage: tools/bench_homomrlwr.py [batch] [name=value ...]      (launch options, e.g. tunnel_mac=0)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alchemy_amd.ringround import RingRound

B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 1024
import os
NT = int(os.environ.get('GEN_NT', '0'))
TF = os.environ.get('TUNNEL_FUSED')
opts = ((('gen_nt', NT),) if NT else ()) + ((('tunnel_fused', int(TF)),) if TF is not None else ())
opts += tuple((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a)
rr = RingRound(B, opts)
secs, out = rr.measure(passes=2)
rr.stages.clear()
rr.run(stage_times=True)
print(json.dumps({"workload": "HomomRLWR ringRound op sequence (mulPublic, 5 tunnels H0'->H5', rescale tree with 8 mul_), real indices and moduli, "
                              "synthetic residues; limb counts from alch_select_limbs", "batch": B, "tunnel_limbs": rr.tuns, "mul_limbs": rr.muls,
                  "pipelines_per_s": B / secs, "ms_per_batch": secs * 1e3, "stage_ms": {k: v * 1e3 for k, v in rr.stages.items()},
                  "out_checksum": f"{out.checksum(0, 2):016x}"}))
