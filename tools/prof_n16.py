import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alchemy_amd import Ring
SIX = [2147352577, 2146959361, 2146041857, 2144468993, 2142502913, 2135818241]
ring = Ring(1 << 17, SIX)
B = 256
a, b, out, hs = ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * B), ring.alloc(2 * ring.L)
a.fill_uniform(1); b.fill_uniform(2); hs.fill_uniform(3)
hint = ring.hint_from_buf(hs)
for _ in range(3):
    ring.ct_mul_relin(hint, a, b, out, B)
ring.sync()
