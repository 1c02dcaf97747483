"""CPU: ring tunnelling in the oracle.  The model (oracle/model_gen.py: coeffs, relative bases, linearDec-style
functions, tunnelHint, tunnel) is pinned semantically -- decrypt(modSwitch(tunnel(modSwitch(ct)))) == f(pt) -- and the
composition of the C restatement's primitives (tests/helpers.py::oracle_tunnel, the checker of the GPU test) reproduces the
model's ciphertexts on the committed fixture."""
import math
import random

import numpy as np
import pytest

from helpers import load_golden, oracle_tunnel, primes_1_mod, to_aos
from oracle import model_gen as G


@pytest.mark.parametrize("r,s,rp,sp,p", [(8, 12, 40, 60, 4), (4, 6, 28, 42, 8), (9, 15, 63, 105, 4), (8, 28, 24, 84, 2)])
def test_tunnel_decrypts_to_the_linear_function_of_the_plaintext(r, s, rp, sp, p):
    rng = random.Random(r * 100 + s)
    T = G.tunnel_indices(r, s, rp, sp)
    qs = primes_1_mod(rp * sp // math.gcd(rp, sp), 3, 1 << 29)
    sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
    ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(T.r.n // T.e.n)]
    pt = [rng.randrange(p) for _ in range(T.r.n)]
    ct = G.g_encrypt(sk_in, pt, T.r, T.rp, p, qs[1:], rng)
    lin_q, hints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng)
    out = G.g_mod_switch_down(G.g_tunnel(lin_q, hints, G.g_mod_switch_up(ct, qs[:1]), T), 1)
    assert out.big.m == sp and out.qs == qs[1:]
    assert G.g_decrypt(sk_out, out) == G.eval_lin_dec(ys, G.linv_def(pt, T.r, p), T.e, T.r, T.s, p)


def test_tunnel_with_base2_hints():
    rng = random.Random(5)
    T = G.tunnel_indices(8, 12, 40, 60)
    qs, p = primes_1_mod(120, 3, 1 << 29), 4
    sk_in, sk_out = G.g_gen_sk(T.rp, rng), G.g_gen_sk(T.sp, rng)
    ys = [[rng.randrange(p) for _ in range(T.s.n)] for _ in range(2)]
    pt = [rng.randrange(p) for _ in range(T.r.n)]
    ct = G.g_encrypt(sk_in, pt, T.r, T.rp, p, qs[1:], rng)
    lin_q, hints = G.g_tunnel_hint(ys, T, p, sk_in, sk_out, qs, rng, gadget="base2")
    out = G.g_mod_switch_down(G.g_tunnel(lin_q, hints, G.g_mod_switch_up(ct, qs[:1]), T, gadget="base2"), 1)
    assert G.g_decrypt(sk_out, out) == G.eval_lin_dec(ys, G.linv_def(pt, T.r, p), T.e, T.r, T.s, p)


def test_reference_hops_are_tunnels():
    """The five hops of examples/Common.hs:78-95 with the index maps of :41-54 satisfy Lol's tunnel conditions."""
    H = [128, 448, 2912, 3640, 5460, 4095]
    Hp = [11648, 29120, 43680, 54600, 27300, 20475]
    for i in range(5):
        T = G.tunnel_indices(H[i], H[i + 1], Hp[i], Hp[i + 1])
        assert T.rp.n % T.ep.n == 0 and T.r.n // T.e.n == T.rp.n // T.ep.n


def test_coeffs_is_the_inverse_of_the_relative_basis_expansion():
    """x = sum_i p_i embed(coeffsPow(x)_i) with p_i the relative powerful basis (by the ring product of the model)."""
    rng = random.Random(3)
    small, big = G.Index(12), G.Index(60)
    q = primes_1_mod(60, 1, 1 << 20)[0]
    x = [rng.randrange(q) for _ in range(big.n)]
    acc = [0] * big.n
    for row, c in zip(G.coeffs_indices(small, big), G.coeffs(x, small, big)):
        pi = [0] * big.n
        pi[row[0]] = 1
        acc = [(u + v) % q for u, v in zip(acc, G.ring_mul_def(pi, G.embed_pow(c, small, big), big, q))]
    assert acc == x


def test_c_restatement_composition_reproduces_the_fixture(oracle_lib):
    lm = lambda a: np.asarray(a).T.tolist()
    for rec in load_golden("tunnel_small.json"):
        qs, L = rec["qs"], len(rec["qs"])
        Or, Os = oracle_lib.GenRing(rec["rp"], qs), oracle_lib.GenRing(rec["sp"], qs)
        lin = [Os.crt(to_aos(y)) for y in rec["lin"]]
        ks = []
        for hint_i in rec["hints"]:
            for b, a in hint_i:
                ks += [Os.crt(to_aos(b)), Os.crt(to_aos(a))]
        c0, c1 = Or.crt(to_aos(rec["ct_in"][0])), Or.crt(to_aos(rec["ct_in"][1]))
        w0, w1 = oracle_tunnel(oracle_lib, rec["rp"], rec["sp"], qs, lin, ks, c0, c1, pow_out=True)
        assert lm(w0) == rec["ct_out"][0] and lm(w1) == rec["ct_out"][1], (rec["rp"], rec["sp"])


def _slot_of_the_subring(ie, isx):
    """The embedCRT slot rule the device's E'-level tunnel path relies on (alchemy_amd/csrc/gen_host.hpp, gen_tunnel_table):
    per prime-power factor the slot index is divided by p^(e_s - e_e); primes that do not divide e' drop out."""
    exp_e = {p: e for p, e in ie.pps}
    out = []
    for s in range(isx.n):
        se = 0
        for (p, es), sf in zip(isx.pps, isx.unravel(s)):
            if p in exp_e:
                ee = exp_e[p]
                se = se * ((p - 1) * p ** (ee - 1)) + sf // p ** (es - ee)
        out.append(se)
    return out


@pytest.mark.parametrize("e_m,s_m", [(20, 60), (8, 24), (9, 45), (12, 36), (40, 120), (28, 364), (16, 32), (5, 25), (50, 100), (21, 63), (24, 360)])
def test_crt_of_an_embedded_element_is_the_small_crt_replicated(e_m, s_m):
    """crt_S'(embedPow d)[s] = crt_E'(d)[slot(s)] by direct evaluation of both sides (Tensor embedCRT): the identity that lets a
    tunnel run the transforms of its embedded E'-coefficients at dimension phi(e')."""
    import random
    from helpers import primes_1_mod
    ie, isx = G.Index(e_m), G.Index(s_m)
    q = primes_1_mod(s_m, 1, 1 << 20)[0]
    rng = random.Random(e_m * 1000 + s_m)
    d = [rng.randrange(q) for _ in range(ie.n)]
    big = G.crt_def(G.embed_pow(d, ie, isx), isx, q)
    small = G.crt_def(d, ie, q)
    slot = _slot_of_the_subring(ie, isx)
    assert all(big[s] == small[slot[s]] for s in range(isx.n))
