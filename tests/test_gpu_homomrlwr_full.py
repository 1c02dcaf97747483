"""GPU: the reference's own acceptance check for BASELINE config 4 at its REAL parameters -- examples/HomomRLWR.hs:62-71:
`decrypt (f a) == eval ringRound (s * a)`, PASS / FAIL -- replayed by the compiled C++ host (examples/homomrlwr_replay.cpp over
alchemy_amd/host/cycgen.hpp + symmshe_gen.hpp, above the C ABI): indices H0 .. H5 / H0' .. H5' (examples/Common.hs:38-54), the
six HomomRLWR moduli (examples/HomomRLWR.hs:37-43), Gaussian parameter 5.0 (:56), TrivGad, plaintext modulus 2^5, linear functions
decToCRT @H_k built from crtSet over Z_32 (Common.hs:65-95), tunnelHint / ksQuadCircHint / encrypt with valid keys, the five hops
and the eight mul_ on the batched device entry points with the limb counts of alch_select_limbs.

The replay itself decrypts after every stage and prints PASS.  Here the ORACLE is the checker once more, independently of the
host layer: the final ciphertexts and the H5' key are decrypted with the C restatement (crt, Horner in s, crtInv, lInv, centred
lift, divG^15 over Z_2, twace, l) and must equal the replay's plaintext results; the first hop's linear function is checked
against the by-definition model (CRT-set idempotents, evalLin on the relative decoding basis).

PARITY UNPINNED against Lol: the error sampler (tweaked Gaussian on the decoding basis) and the order of the CRT set are this
build's; the PASS does not depend on either choice, the noise margin does."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import model_gen as G

pytestmark = pytest.mark.gpu
H = [128, 448, 2912, 3640, 5460, 4095]
HP5 = 20475


@pytest.fixture(scope="module")
def replay(tmp_path_factory):
    exe = os.path.join(ROOT, "examples", "homomrlwr_replay")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "examples", "homomrlwr_replay.cpp"),
                    "-L" + os.path.join(ROOT, "alchemy_amd", "lib"), "-lalchemy_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "alchemy_amd", "lib")], check=True)
    d = tmp_path_factory.mktemp("rlwr")
    out = subprocess.run([exe, "6", "--per-element", "--dump", str(d)], capture_output=True, text=True, timeout=600)
    return out, d


def test_homomrlwr_example_prints_pass_at_the_reference_parameters(replay):
    out, _ = replay
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    assert text.strip().endswith("PASS")
    assert "tunnels 5/6/5 5/6/5 5/6/5 5/6/5 5/5/4; mul_ 4/5/3 3/4/2 2/3/1 1/2/1" in text          # SURVEY 3.3's table
    assert "decrypted results equal to the plaintext results: 6 of 6" in text
    assert "every div2 operand even: yes" in text
    assert "equal to the batched result: yes" in text                                             # per-Tensor-call path == batched path
    stages = re.findall(r"^\s+(.+?)\s+q has (\d) limbs, p = +(\d+), k = +(\d+)\s+error rate ([0-9.e+-]+)\s+decrypts to the plaintext stage: (\S+)",
                        text, flags=re.M)
    assert len(stages) == 11
    assert all(ok in ("yes", "-") for *_, ok in stages)
    assert [ok for *_, ok in stages].count("yes") == 8                                              # mulPublic, 5 hops, x(1+x), final
    rates = [float(r) for *_, r, _ in stages]
    assert rates[:7] == sorted(rates[:7]) and rates[7:] == sorted(rates[7:])      # noise grows stage by stage (the leaf's div2 halves p once)
    assert rates[-1] < 0.5


def test_oracle_decrypts_the_device_result_to_the_plaintext_result(replay, oracle_lib):
    out, d = replay
    assert out.returncode == 0
    meta = np.fromfile(os.path.join(d, "meta.i64"), dtype=np.int64)
    B, n, L, msd, k, l, p, q = (int(x) for x in meta)
    assert (n, L, msd, p) == (G.totient(HP5), 1, 1, 2) and k == 15
    sk = np.fromfile(os.path.join(d, "sk5_pow.i64"), dtype=np.int64)
    cts = np.fromfile(os.path.join(d, "cts_crt.i64"), dtype=np.int64).reshape(B, 2, n, 1)
    expect = np.fromfile(os.path.join(d, "expect_pow.i64"), dtype=np.int64).reshape(B, G.totient(H[5]))
    o = oracle_lib.GenRing(HP5, [q])
    s = o.crt(np.ascontiguousarray((sk % q).reshape(n, 1)))
    z2, zs = oracle_lib.GenRing(HP5, [2]), oracle_lib.GenRing(H[5], [2])
    pos = G.embed_indices(G.Index(H[5]), G.Index(HP5))
    # MSD -> LSD: c * p, l * (-q)^-1 mod p
    l_lsd = l * pow((-q) % p, -1, p) % p
    for b in range(B):
        c0, c1 = (o.scale(cts[b][c], [p % q]) for c in range(2))
        e = o.linv(o.crtinv(o.add(c0, o.mul(c1, s))))[:, 0]
        e = np.where(e > (q - 1) // 2, e - q, e)                                 # liftDec
        assert np.abs(e).max() < q / 2
        x = np.ascontiguousarray((e % p).reshape(n, 1))
        for _ in range(k):
            x = z2.divg_dec(x)
            assert x is not None
        t = np.ascontiguousarray(x[pos, :])                                       # twacePowDec on the decoding basis
        got = (l_lsd * zs.l(t)[:, 0]) % p
        assert got.tolist() == expect[b].tolist(), b


def test_first_hop_linear_function_is_dec_to_crt_by_definition(replay):
    """decToCRT @H0 (examples/Common.hs:65-80): E = O_64, R = O_128, S = O_448.  The dumped linear function must consist of
    idempotents of S / 32 S that are 1 on exactly one prime above 2 for every prime of E (checked mod 2 over GF(2^3) through the
    odd part O_7, and idempotence mod 32), and the dumped plaintext after the hop must be evalLin of the plaintext before it."""
    out, d = replay
    assert out.returncode == 0
    e, r, s = G.Index(64), G.Index(H[0]), G.Index(H[1])
    ys = np.fromfile(os.path.join(d, "lin0_pow.i64"), dtype=np.int64).reshape(-1, s.n)
    assert ys.shape[0] == r.n // e.n == 2
    F = G.GF(2, 3)
    w = F.root_of_unity(7)
    o7, seen = G.Index(7), []
    for y in ys.tolist():
        assert G.ring_mul_def(y, y, s, 32) == y                                  # idempotent mod 2^5
        y7 = [v % 2 for v in G.twace_pow_dec(y, o7, s)]                          # the CRT set lives in O_7 (PFree 2), embedded
        assert G.embed_pow(G.twace_pow_dec(y, o7, s), o7, s) == y
        vals = [G.eval_mod_p(y7, o7, F, w, u) for u in range(1, 7)]
        assert all(v in (F.zero, F.one) for v in vals)
        seen.append(tuple(v == F.one for v in vals))
    assert all(sum(a) == 3 for a in seen) and all(x != y for x, y in zip(*seen))       # the two cosets {1,2,4}, {3,5,6}: disjoint
    x0 = np.fromfile(os.path.join(d, "pt_h0.i64"), dtype=np.int64).tolist()
    x1 = np.fromfile(os.path.join(d, "pt_h1.i64"), dtype=np.int64).tolist()
    assert G.eval_lin_dec(ys.tolist(), G.linv_def(x0, r, 32), e, r, s, 32) == x1


def test_tunnel_example_at_the_reference_parameters():
    """examples/Tunnel.hs (BASELINE config 5) through the compiled C++ host: switch3 = H0 -> H1 -> H2 -> H3 (the file's `tunnel3`),
    BaseBGad 2 hints (:24), its moduli (:34-39), Gaussian parameter 3.0 (:59), plaintext modulus 2^3, limb counts from
    alch_select_limbs with the BaseBGad rule (2/1/1, 1/1/1, 1/1/1: the first modSwitch goes DOWN).  The reference prints error rates
    only; the replay also decrypts after every hop and compares with the plaintext evalLin chain."""
    exe = os.path.join(ROOT, "examples", "tunnel_replay")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "examples", "tunnel_replay.cpp"),
                    "-L" + os.path.join(ROOT, "alchemy_amd", "lib"), "-lalchemy_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "alchemy_amd", "lib")], check=True)
    out = subprocess.run([exe, "3", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "limbs (in/hint/out): 2/1/1 1/1/1 1/1/1" in out.stdout
    assert out.stdout.count("decrypts to the plaintext evaluation: 4 of 4") == 3
    assert out.stdout.strip().endswith("PASS")


def test_ring_round_is_coefficientwise_rounding(replay):
    """What ringRound MEANS (independent of Lol's internals and of this build's conventions): the five switches move the 64
    decoding-basis coefficients c_j in Z_32 of x = s * a over H0 = O_128 into mod-2^5 CRT slots of H5 = O_4095, and the rescale tree
    rounds every slot, v -> [8 <= v < 24] (the RLWR rounding floor((v + 8) / 16) mod 2; Language/RescaleTree.hs:64-87 on scalars).
    So the plaintext result, reduced mod 2, must take the value 0 or 1 at every prime above 2 of O_4095 (144 of them, residue field
    GF(2^12)), and the multiset of those values must be {round(c_j)} plus zeros.  This pins crtSet / decToCRT / evalLin and the tree
    of the host layer to the function the example computes, not merely to each other."""
    out, d = replay
    assert out.returncode == 0
    h5 = G.Index(H[5])
    x0 = np.fromfile(os.path.join(d, "pt_h0.i64"), dtype=np.int64).tolist()           # x = s * a: Pow = Dec coefficients over O_128
    res = np.fromfile(os.path.join(d, "expect_pow.i64"), dtype=np.int64).reshape(-1, h5.n)[0].tolist()
    F = G.GF(2, G.mult_order(2, H[5]))
    w = F.root_of_unity(H[5])
    pw, acc = [], F.one                                      # powers of w as bit masks: addition in GF(2^12) is XOR
    for _ in range(H[5]):
        pw.append(sum(b << i for i, b in enumerate(acc)))
        acc = F.mul(acc, w)
    ex = [h5.pow_exponent(j) for j, c in enumerate(res) if c % 2]
    reps = [c[0] for c in _cosets(H[5])]
    assert len(reps) == 144
    vals = []
    for u in reps:
        v = 0
        for e in ex:
            v ^= pw[u * e % H[5]]
        vals.append(v)
    assert all(v in (0, 1) for v in vals)
    ones = sum(vals)
    assert ones == sum(1 for c in x0 if 8 <= c % 32 < 24)
    assert len(x0) == 64 and 0 < ones < 64


def _cosets(m):
    import math
    seen = set()
    for u in range(1, m):
        if math.gcd(u, m) == 1 and u not in seen:
            c, x = [], u
            while x not in seen:
                seen.add(x); c.append(x); x = x * 2 % m
            yield c
