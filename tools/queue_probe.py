"""How the two sub-batches of the HomomRLWR pipeline fare with D other rings (2 D more HIP streams) alive in the process.  With ordinary
streams the lanes shared a hardware queue for every odd D (44.7 k instead of 50.5 k pipelines/s); with the dedicated streams RingRoundLanes
asks for (option stream_dedicated) every D gives 50.5 k (profiles/r04_queue_probe.txt)."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alchemy_amd as A
from alchemy_amd.ringround import RingRoundLanes
QS = [1543651201, 689270401, 718099201, 720720001]
for D in range(0, 6):
    dummies = [A.Ring(20475, QS[:2]) for _ in range(D)]
    rl = RingRoundLanes(1024, 2)
    secs, outs = rl.measure(4)
    print(json.dumps({"dummy_rings_alive": D, "pipelines_per_s": round(1024 / secs)}), flush=True)
    del rl, outs

# the headline's two chunk pipelines (the ring's stream and its aux stream, created back to back) under the same variation
CFG3 = [2147352577, 2146959361, 2146041857, 2145976321]
for D in range(0, 4):
    dummies = [A.Ring(20475, QS[:2]) for _ in range(D)]
    r = A.Ring(1 << 16, CFG3)
    B = 4096
    a, b, o, hs = r.alloc(2 * B), r.alloc(2 * B), r.alloc(2 * B), r.alloc(8)
    a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
    hint = r.hint_from_buf(hs)
    r.ct_mul_relin(hint, a, b, o, B); r.sync()
    r.timer_start()
    for _ in range(6):
        r.ct_mul_relin(hint, a, b, o, B)
    t = r.timer_stop() * 1e-3 / 6
    print(json.dumps({"dummy_rings_alive": D, "headline_shape_ops_per_s": round(B / t)}), flush=True)
    del a, b, o, hs, hint, r
