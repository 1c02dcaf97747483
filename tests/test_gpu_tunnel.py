"""GPU: ring tunnelling through the C ABI (alch_tunnel_create / alch_ct_tunnel): the model-generated fixture
(valid hints, the output decrypts to f(pt)) bit for bit, and the reference's first hop switch1 = H0' -> H1'
(F11648 -> F29120, examples/Common.hs:49-50,78-80) at full size against the C restatement's composition."""
import math

import numpy as np
import pytest

import alchemy_amd as A
from alchemy_amd import capi
from helpers import load_golden, oracle_tunnel, primes_1_mod, to_aos

pytestmark = pytest.mark.gpu

RLWR_QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]


def rand_elems(rng, count, n, qs):
    return np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.int64) for q in qs], axis=1) for _ in range(count)])


def test_tunnel_fixture():
    lm = lambda a: np.asarray(a).T.tolist()
    for rec in load_golden("tunnel_small.json"):
        qs = rec["qs"]
        gr, gs = A.Ring(rec["rp"], qs), A.Ring(rec["sp"], qs)
        assert A.Tunnel.info(gr, gs) == (rec["ep"], len(rec["lin"]))
        lin = gs.upload(np.stack([to_aos(y) for y in rec["lin"]]))
        ks = gs.upload(np.stack([to_aos(x) for hint_i in rec["hints"] for pair in hint_i for x in pair]))
        lin.crt(); ks.crt()
        tun = A.Tunnel(gr, gs, lin, ks)
        cin = gr.upload(np.stack([to_aos(c) for c in rec["ct_in"]]))
        cout = gs.alloc(2)
        tun.apply(cin, cout, 1, flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)
        got = cout.download()
        assert lm(got[0]) == rec["ct_out"][0] and lm(got[1]) == rec["ct_out"][1], (rec["rp"], rec["sp"])
        assert np.array_equal(cin.download(), np.stack([to_aos(c) for c in rec["ct_in"]]))      # input untouched


@pytest.mark.parametrize("rp,sp,L,batch", [(40, 60, 2, 3), (63, 105, 3, 2), (11648, 29120, 6, 3), (27300, 20475, 5, 2)])
def test_tunnel_hop_matches_the_oracle(oracle_lib, rp, sp, L, batch):
    """switch1 (H0' -> H1', six limbs: SURVEY 3.3 'tunnels switch1-4: hint on 6') and switch5 (H4' -> H5', five limbs) at
    full size, random linear function / hints / ciphertexts (parity does not need them valid)."""
    qs = RLWR_QS[:L] if rp > 1000 else primes_1_mod(rp * sp // math.gcd(rp, sp), L, 1 << 29)
    gr, gs = A.Ring(rp, qs), A.Ring(sp, qs)
    ep, d_rel = A.Tunnel.info(gr, gs)
    assert ep == math.gcd(rp, sp) and d_rel * (gs.n * 0 + 1) >= 1
    rng = np.random.default_rng(rp)
    lin, ks = rand_elems(rng, d_rel, gs.n, qs), rand_elems(rng, 2 * d_rel * L, gs.n, qs)
    cts = rand_elems(rng, 2 * batch, gr.n, qs)
    s_pre = [int(rng.integers(1, q)) for q in qs]
    want = [oracle_tunnel(oracle_lib, rp, sp, qs, list(lin), list(ks), cts[2 * ct], cts[2 * ct + 1], s_pre) for ct in range(batch)]
    gin, gout = gr.upload(cts), gs.alloc(2 * batch)
    # tunnel_ep = 1 (default): the embedded E'-coefficients are transformed at dimension phi(e') and read through the embedCRT slot
    # table; 0: embedded into S' first and transformed there
    # tunnel_fused: digit transforms + hint products in one kernel (2 or 4 digits side by side), or through HBM (0)
    # tunnel_mac (round 4): the hint inner product with lazy 64-bit groups starting from evalLin's constant term (1, default), or
    # k_tunnel_lin + one Montgomery product at a time (0)
    for ep_level, fused, mac in ((1, 2, 1), (1, 4, 1), (1, 0, 1), (1, 0, 0), (0, 2, 1)):
        gs.set_option("tunnel_ep", ep_level)
        gs.set_option("tunnel_fused", fused)
        gs.set_option("tunnel_mac", mac)
        tun = A.Tunnel(gr, gs, gs.upload(lin), gs.upload(ks))
        tun.apply(gin, gout, batch, s_pre=s_pre)
        got = gout.download()
        for ct in range(batch):
            assert np.array_equal(got[2 * ct], want[ct][0]) and np.array_equal(got[2 * ct + 1], want[ct][1]), (ct, ep_level, fused, mac)


@pytest.mark.parametrize("rp,sp,L,dup,gadget", [(40, 60, 3, 1, "triv"), (63, 105, 4, 2, "triv"), (11648, 29120, 6, 1, "triv"),
                                                (43680, 54600, 6, 1, "triv"), (40, 60, 3, 1, "base2")])
def test_tunnel_below_the_hint_ring_equals_mod_switch_then_tunnel(oracle_lib, rp, sp, L, dup, gadget):
    """PT2CT's modSwitch_ .: tunnel_ hint (PT2CT.hs:224-229) as one call: ciphertexts on the last L - dup limbs of the tunnel's
    ring go straight into alch_ct_tunnel (zero limbs skipped); same residues as alch_ct_mod_switch (up) + alch_ct_tunnel, and
    as the oracle's composition."""
    qs = RLWR_QS[:L] if rp > 1000 else primes_1_mod(rp * sp // math.gcd(rp, sp), L, 1 << 29)
    gr, gs, gsmall = A.Ring(rp, qs), A.Ring(sp, qs), A.Ring(rp, qs[dup:])
    _, d_rel = A.Tunnel.info(gr, gs)
    rng = np.random.default_rng(rp + dup)
    D = gs.gadget_digits(capi.ALCH_GAD_BASE2) if gadget == "base2" else L
    lin, ks = rand_elems(rng, d_rel, gs.n, qs), rand_elems(rng, 2 * d_rel * D, gs.n, qs)
    batch = 2
    cts = rand_elems(rng, 2 * batch, gr.n, qs[dup:])
    s_pre = [int(rng.integers(1, q)) for q in qs]
    tun = A.Tunnel(gr, gs, gs.upload(lin), gs.upload(ks), gadget=capi.ALCH_GAD_BASE2 if gadget == "base2" else capi.ALCH_GAD_TRIV)
    gin, gup, g1, g2 = gsmall.upload(cts), gr.alloc(2 * batch), gs.alloc(2 * batch), gs.alloc(2 * batch)
    capi.ct_mod_switch(gin, gup, batch)
    tun.apply(gup, g1, batch, s_pre=s_pre)
    tun.apply(gin, g2, batch, s_pre=s_pre)
    two_calls, one_call = g1.download(), g2.download()
    assert np.array_equal(one_call, two_calls)
    assert np.array_equal(gin.download(), cts)               # input untouched
    if gadget == "triv":
        up = gup.download()
        for ct in range(batch):
            w0, w1 = oracle_tunnel(oracle_lib, rp, sp, qs, list(lin), list(ks), up[2 * ct], up[2 * ct + 1], s_pre)
            assert np.array_equal(one_call[2 * ct], w0) and np.array_equal(one_call[2 * ct + 1], w1), ct


@pytest.mark.parametrize("k", range(5))
def test_tunnel_hs_hop_with_base2_hints_at_full_size(oracle_lib, k):
    """BASELINE config 5 at the reference's real parameters (what bench.py's `tunnel_hs` line and tools/bench_tunnel.py time,
    alchemy_amd/tunnelhops.py): every hop of examples/Tunnel.hs -- BaseBGad 2 hints (:24), its moduli (:34-39), indices
    H_k' -> H_k+1' (examples/Common.hs:49-54), the limb counts alch_select_limbs derives (the hint may sit on FEWER limbs than the
    input: the leading modSwitch goes down) -- as modSwitch . tunnel hint . modSwitch on the device:
      * two ciphertexts word for word against the C restatement's composition run live (tests/tunnelhops_oracle.py),
      * a ragged batch of 37 by whole-batch checksum against the values the oracle produced offline (what bench.py asserts
        for its 256-ciphertext batches)."""
    from alchemy_amd.tunnelhops import Hop
    from helpers import load_golden
    from tunnelhops_oracle import HopOracle
    B = 37
    old = Hop(k, B, (("tunnel_mac", 0),))                 # the round-3 kernels (k_tunnel_lin + k_hint_mac_e): same words
    old_sum = old.run().checksum(0, 2 * B)
    del old
    hop = Hop(k, B)
    res = hop.run()
    hop.rs.sync()
    assert res.checksum(0, 2 * B) == old_sum
    orc = HopOracle(oracle_lib, k)
    assert (orc.lin, orc.lh, orc.lout, orc.d_rel, orc.D) == (hop.lin, hop.lh, hop.lout, hop.d_rel, hop.D)
    got = res.download(0, 4)
    for ct in range(2):
        w0, w1 = orc.run(ct)
        assert np.array_equal(got[2 * ct], w0) and np.array_equal(got[2 * ct + 1], w1), ct
    ref = load_golden("batch_checksums.json")["tunnel_hs"]["hops"][k]
    want = sum(int(x, 16) for x in ref["per_ciphertext"][:B]) & ((1 << 64) - 1)
    assert f"{res.checksum(0, 2 * B):016x}" == f"{want:016x}"


def test_tunnel_with_base2_hints_decrypts_to_f_of_pt():
    """BaseBGad 2 tunnel hints (the gadget of examples/Tunnel.hs:24) with its moduli (examples/Tunnel.hs:34-39) on a small tower:
    a valid model instance through the device, compared bit for bit with the model and decrypted by it."""
    import random
    from oracle import model_gen as G
    rng = random.Random(24)
    r, s, rp, sp, p = 8, 12, 40, 60, 8
    T = G.tunnel_indices(r, s, rp, sp)
    qs = [537264001, 539884801, 555609601]                          # first three of Tunnel.hs's moduli: all = 1 mod 120
    assert all((q - 1) % 120 == 0 for q in qs)
    sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
    ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(T.r.n // T.e.n)]
    pt = [rng.randrange(p) for _ in range(T.r.n)]
    ct = G.g_mod_switch_up(G.g_encrypt(sk_in, pt, T.r, T.rp, p, qs[1:], rng), qs[:1])
    lin_q, hints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng, gadget="base2")
    want = G.g_tunnel(lin_q, hints, ct, T, gadget="base2")
    gr, gs = A.Ring(rp, qs), A.Ring(sp, qs)
    D = gs.gadget_digits(capi.ALCH_GAD_BASE2)
    assert D == len(hints[0]) == sum(q.bit_length() for q in qs)
    lin = gs.upload(np.stack([to_aos(y) for y in lin_q]))
    ks = gs.upload(np.stack([to_aos(x) for hint_i in hints for pair in hint_i for x in pair]))
    lin.crt(); ks.crt()
    tun = A.Tunnel(gr, gs, lin, ks, gadget=capi.ALCH_GAD_BASE2)
    cin, cout = gr.upload(np.stack([to_aos(c) for c in ct.c])), gs.alloc(2)
    tun.apply(cin, cout, 1, flags=capi.ALCH_POW_IN | capi.ALCH_POW_OUT)
    got = cout.download()
    lm = lambda a: np.asarray(a).T.tolist()
    assert lm(got[0]) == want.c[0] and lm(got[1]) == want.c[1]
    dev = G.GCT(want.enc, want.k, want.l, [lm(got[0]), lm(got[1])], p, qs, T.sp, T.s)
    assert G.g_decrypt(sk_out, G.g_mod_switch_down(dev, 1)) == G.eval_lin_dec(ys, G.linv_def(pt, T.r, p), T.e, T.r, T.s, p)


def test_tunnel_argument_checks():
    qs = primes_1_mod(40 * 60, 3, 1 << 29)
    gr, gs, gs2 = A.Ring(40, qs[:2]), A.Ring(60, qs[:2]), A.Ring(60, qs[1:])
    assert A.Tunnel.info(gr, gs) == (20, 2)
    lin, ks = gs.alloc(2), gs.alloc(8)
    with pytest.raises(A.AlchemyError):
        A.Tunnel(gr, gs2, gs2.alloc(2), gs2.alloc(8))         # different moduli
    with pytest.raises(A.AlchemyError):
        A.Tunnel(gr, gs, lin, gs.alloc(7))                    # too few hint elements
    t = A.Tunnel(gr, gs, lin, ks)
    with pytest.raises(A.AlchemyError):
        t.apply(gs.alloc(2), gs.alloc(2), 1)                  # input must live in R'


def test_tunnel_on_a_ciphertext_with_g_factors():
    """SymmSHE `tunnel` on a ciphertext with k > 0 (after a product): Lol runs `absorbGFactors` first -- every component times
    reduce(liftPow(g^-k mod p)), k <- 0 -- composed here from entry points that exist (alch_divg_pow on a ring without CRT over Z_p,
    alch_crt, alch_buf_mul_public), then alch_ct_tunnel.  A valid model instance: product of two encryptions (k = 1), key switch,
    absorb, tunnel; the device result equals the model's bit for bit and the model decrypts it to f(pt1 * pt2).  No ALCHEMY example
    reaches this path (every tunnel of the reference sits in front of the first product)."""
    import random
    from oracle import model_gen as G
    rng = random.Random(31)
    r, s, rp, sp, p = 8, 12, 40, 60, 8
    T = G.tunnel_indices(r, s, rp, sp)
    qs = primes_1_mod(rp * sp // math.gcd(rp, sp), 3, 1 << 29)
    sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
    ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(T.r.n // T.e.n)]
    pt1, pt2 = ([rng.randrange(p) for _ in range(T.r.n)] for _ in range(2))
    prod = G.g_key_switch(G.g_ks_hint(sk_in, T.rp, qs, rng), G.g_ct_mul(G.g_encrypt(sk_in, pt1, T.r, T.rp, p, qs, rng),
                                                                    G.g_encrypt(sk_in, pt2, T.r, T.rp, p, qs, rng)))
    assert prod.k == 1 and G.g_decrypt(sk_in, prod) == G.ring_mul_def(pt1, pt2, T.r, p)
    absorbed = G.g_absorb_g_factors(prod)
    assert absorbed.k == 0 and G.g_decrypt(sk_in, absorbed) == G.ring_mul_def(pt1, pt2, T.r, p)
    lin_q, hints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng)
    want = G.g_tunnel(lin_q, hints, absorbed, T)
    # ---- the device
    gr, gs, zp = A.Ring(rp, qs), A.Ring(sp, qs), A.Ring(rp, [p], nocrt=True)
    d = np.zeros((T.rp.n, 1), dtype=np.int64)
    d[0, 0] = 1
    for _ in range(prod.k):
        d = zp.divg_pow(d)                                            # g^-1 over Z_p, once per power
        assert d is not None
    dz = np.where(d[:, 0] > (p - 1) // 2, d[:, 0] - p, d[:, 0])
    rep = gr.upload(np.stack([np.stack([dz % q for q in qs], axis=1)]))
    rep.crt()
    cin = gr.upload(np.stack([to_aos(c) for c in prod.c]))             # MSD, Pow basis
    cin.crt()
    cabs = gr.alloc(2)
    cabs.mul_public(cin, rep, 0, 2)
    lin = gs.upload(np.stack([to_aos(y) for y in lin_q]))
    ks = gs.upload(np.stack([to_aos(x) for hint_i in hints for pair in hint_i for x in pair]))
    lin.crt(); ks.crt()
    tun = A.Tunnel(gr, gs, lin, ks)
    cout = gs.alloc(2)
    tun.apply(cabs, cout, 1, flags=capi.ALCH_POW_OUT)
    got = cout.download()
    lm = lambda a: np.asarray(a).T.tolist()
    assert lm(got[0]) == want.c[0] and lm(got[1]) == want.c[1]
    dev = G.GCT(want.enc, 0, want.l, [lm(got[0]), lm(got[1])], p, qs, T.sp, T.s)
    f_of = G.eval_lin_dec(ys, G.linv_def(G.ring_mul_def(pt1, pt2, T.r, p), T.r, p), T.e, T.r, T.s, p)
    assert G.g_decrypt(sk_out, dev) == f_of
