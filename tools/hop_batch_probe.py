import sys, json
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from alchemy_amd.tunnelhops import Hop
for B in (256, 1024, 2048):
    rates = []
    for k in range(5):
        hop = Hop(k, B)
        hop.run(); hop.rs.sync()
        hop.rs.timer_start()
        for _ in range(3): hop.run()
        rates.append(round(3 * B / (hop.rs.timer_stop() * 1e-3)))
        del hop
    print(B, rates, flush=True)
