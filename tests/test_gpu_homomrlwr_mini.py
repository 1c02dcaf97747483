"""GPU: the op sequence of examples/HomomRLWR.hs (BASELINE config 4) on the DEVICE, with miniature parameters so that the
exact model can play the Haskell host (key generation, hints, encryption, decryption):

    f = eval (pt2ct ringRound) . (`mulPublic` enc(s))          (examples/HomomRLWR.hs:52-59)
    ringRound = rescaleTree .: switch                           (:45-50; here: one hop, then the tree's first product x * (1 + x))

i.e.  mulPublic a  ->  modSwitch up . tunnel hint . modSwitch down  (PT2CT.hs:224-229)  ->  addPublic 1  ->
      modSwitch . keySwitchQuad hint . modSwitch $ x * (1 + x)  (PT2CT.hs:172-177)  ->  modSwitchPT (div2_, PT2CT.hs:179-189).
Every ring operation runs through the C ABI on a batch of ciphertexts; after every stage the device ciphertexts are compared
bit for bit with the model's, and at the end the model DECRYPTS the device result: it must equal the plaintext evaluation
(the example's PASS).  Indices: R = O_8 -> S = O_12 with R' = O_40 -> S' = O_60 (same shape as H0 -> H1: E = R cap S)."""
import math
import random

import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi
from helpers import primes_1_mod, to_aos
from oracle import model_gen as G
from oracle.model import LSD, MSD

pytestmark = pytest.mark.gpu


def up(ring, elems):          # list of RNS Pow elements (limb-major) -> device buffer
    return ring.upload(np.stack([to_aos(e) for e in elems]))


def lm(arr):
    return np.asarray(arr).T.tolist()


def test_mini_homomrlwr_pipeline_on_the_device():
    rng = random.Random(2026)
    r, s, rp, sp, p, B = 8, 12, 40, 60, 8, 3
    T = G.tunnel_indices(r, s, rp, sp)
    qs = primes_1_mod(rp * sp // math.gcd(rp, sp), 4, 1 << 29)        # q0 (hint's extra limb) .. q3
    q_ct = qs[1:]
    sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
    ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(T.r.n // T.e.n)]     # the E-linear function (linearDec)
    lin_q, thints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng)
    qhint = G.g_ks_hint(sk_out, T.sp, qs, rng)
    secrets = [[rng.randrange(p) for _ in range(T.r.n)] for _ in range(B)]             # RLWR secrets s (plaintexts)
    a_pub = [rng.randrange(p) for _ in range(T.r.n)]                                   # the public a

    # ---- host side (model): fresh encryptions of s
    cts = [G.g_encrypt(sk_in, sv, T.r, T.rp, p, q_ct, rng) for sv in secrets]

    # ---- device: rings and resident hints
    R3, R4 = A.Ring(rp, q_ct), A.Ring(rp, qs)
    S4, S3, S2 = A.Ring(sp, qs), A.Ring(sp, q_ct), A.Ring(sp, qs[2:])
    lin = up(S4, lin_q); lin.crt()
    tks = up(S4, [x for hint_i in thints for pair in hint_i for x in pair]); tks.crt()
    tunnel = A.Tunnel(R4, S4, lin, tks)
    qh = up(S4, [x for pair in qhint for x in pair]); qh.crt()
    quad = S4.hint_from_buf(qh)

    def check(buf, model_cts, what):
        got = buf.download()
        for b, ct in enumerate(model_cts):
            for comp in range(2):
                assert lm(got[2 * b + comp]) == ct.c[comp], (what, b, comp)

    # 1. mulPublic a (every component times embed(reduce(liftPow a))), CRT basis
    x = up(R3, [c for ct in cts for c in ct.c]); x.crt()
    a_emb = G.embed_pow([G.centred(v, p) for v in a_pub], T.r, T.rp)
    pub = up(R3, [[[v % q for v in a_emb] for q in q_ct]]); pub.crt()
    x1 = R3.alloc(2 * B)
    x1.mul_public(x, pub, 0, 2 * B)
    m1 = [G.g_mul_public(a_pub, ct) for ct in cts]
    x1.crtinv(); check(x1, m1, "mulPublic")
    # 2. modSwitch up to the hint modulus: toMSD (p^-1 per limb), then Rescale b -> (a, b)
    x1.scale(x1, 2 * B, [pow(p, -1, q) for q in q_ct])
    x2 = R4.alloc(2 * B)
    capi.ct_mod_switch(x1, x2, B)                            # alch_ct_mod_switch, up (any basis)
    m2 = [G.g_mod_switch_up(ct, qs[:1]) for ct in m1]
    check(x2, m2, "modSwitch up")
    # 3. tunnel (Pow in, Pow out)
    y = S4.alloc(2 * B)
    tunnel.apply(x2, y, B, flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)
    m3 = [G.g_tunnel(lin_q, thints, ct, T) for ct in m2]
    check(y, m3, "tunnel")
    # 4. modSwitch down: rescaleDec on c0, rescalePow on c1
    y3 = S3.alloc(2 * B)
    capi.ct_mod_switch(y, y3, B, flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)       # alch_ct_mod_switch, down
    m4 = [G.g_mod_switch_down(ct, 1) for ct in m3]
    check(y3, m4, "modSwitch down")
    assert all(G.g_decrypt(sk_out, ct) == G.eval_lin_dec(ys, G.linv_def(G.ring_mul_def(sv, a_pub, T.r, p), T.r, p), T.e, T.r, T.s, p)
               for ct, sv in zip(m4, secrets))
    # 5. 1 + x: addPublic on the LSD form (toLSD = p per limb)
    one = [1] + [0] * (T.s.n - 1)
    m5 = [G.g_add_public(one, ct) for ct in m4]
    y3.scale(y3, 2 * B, [p % q for q in q_ct])
    lsd = G.g_to_lsd(m4[0])
    pub1 = G.embed_pow([G.centred(v * pow(lsd.l, -1, p) % p, p) for v in one], T.s, T.sp)       # k = 0: no mulG
    pb = up(S3, [[[v % q for v in pub1] for q in q_ct]])
    y_lsd = S3.alloc(2 * B)                                  # x (LSD) is needed again for the product
    y_lsd.scale(y3, 2 * B, [1] * len(q_ct))
    y3.add_public(pb, 0, B)
    check(y3, m5, "addPublic")
    # 6. the product: modSwitch . keySwitchQuad hint . modSwitch $ x * (1 + x), operands on 3 limbs, hint on 4, result on 2
    y_lsd.crt(); y3.crt()
    z = S2.alloc(2 * B)
    capi.ct_mul_full(quad, y_lsd, y3, z, B, s_pre=[pow(p, -1, q) for q in q_ct], flags=capi.ALCH_POW_OUT)
    m6 = [G.g_mod_switch_down(G.g_key_switch(qhint, G.g_mod_switch_up(G.g_ct_mul(G.g_to_lsd(c4), c5), qs[:1])), 2)
          for c4, c5 in zip(m4, m5)]
    check(z, m6, "mul_")
    # 7. decrypt the DEVICE result on the host: t (1 + t) with t = f(a s)
    got = z.download()
    for b, (sv, mct) in enumerate(zip(secrets, m6)):
        dev_ct = G.GCT(MSD, mct.k, mct.l, [lm(got[2 * b]), lm(got[2 * b + 1])], p, qs[2:], T.sp, T.s)
        t = G.eval_lin_dec(ys, G.linv_def(G.ring_mul_def(sv, a_pub, T.r, p), T.r, p), T.e, T.r, T.s, p)
        want = G.ring_mul_def(t, [(v + (1 if i == 0 else 0)) % p for i, v in enumerate(t)], T.s, p)
        assert G.g_decrypt(sk_out, dev_ct) == want, b                    # PASS
        half = G.g_mod_switch_pt(dev_ct, p // 2)                          # div2_: ring elements unchanged, plaintext modulus halves
        assert half.c == dev_ct.c and half.p == p // 2
