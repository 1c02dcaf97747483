"""Exact big-integer model of ALCHEMY's ciphertext multiply + relinearize path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``alchemy_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use ``oracle/``.

PARITY UNPINNED.  The arithmetic on this path is not in /root/reference at all: it lives in the
third-party Lol packages (``lol``, ``lol-apps``, ``lol-cpp`` from github.com/cpeikert/lol), pinned
by *branch name* ``alchemy-args-debruijn-monad`` in the reference's stack.yaml:54-60 (cabal bounds
``lol >= 0.7``, ``lol-apps >= 0.2``, alchemy.cabal:51-52) and not vendored.  No Haskell toolchain
exists in this pipeline and the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4, 8c).  This model therefore restates the *published mathematics* of Lol's
SymmSHE (Crockett & Peikert, "Lambda-o-lambda: Functional Lattice Cryptography", CCS'16) and is
anchored on the reference's own call sites:

  * ``(*)`` on ``CT``                  -- Crypto/Alchemy/Interpreter/Eval.hs:65-67
  * ``modSwitch``                      -- Eval.hs:130   (context :123)
  * ``keySwitchQuadCirc``              -- Eval.hs:133   (context :126)
  * op order ``modSwitch . keySwitchQuad hint . modSwitch $ x*y``
                                       -- Crypto/Alchemy/Interpreter/PT2CT.hs:172-177
  * ``ksQuadCircHint`` / ``genSK``     -- Crypto/Alchemy/Interpreter/KeysHints.hs:86-96,101-113
  * ``encrypt`` / ``decrypt``          -- PT2CT.hs:84-87, 91-99
  * storage ``ZqBasic q Int64``        -- examples/Common.hs:35
  * gadgets ``TrivGad``, ``BaseBGad 2``-- PT2CT.hs:139-140

Everything here uses Python integers and *definitions* (direct evaluation for the CRT, O(n^2)
schoolbook for ring products), so that it is independent of every fast algorithm it checks.

Conventions shared with the C restatement (oracle/lol_tensor.c) and the HIP library:

  ring            R'_q = Z_q[X]/(X^n + 1), cyclotomic index m' = 2n a power of two
  Pow basis       coefficient vector (a_0 .. a_{n-1}), residues stored reduced to [0, q)
  Dec basis       identical to Pow for a two-power index (Lol's L matrix is 1x1 for p = 2)
  root rule       g = smallest generator of Z_q^*;  psi = g^((q-1)/(2n))  (primitive 2n-th root)
  CRT slot order  slot k holds a(psi^(2*brev_n(k)+1)),  brev_n = bit reversal on log2(n) bits
  RNS             independent limbs q_0 .. q_{L-1}; limb 0 is the outermost component of Lol's
                  nested pair  (q_0,(q_1,(...,q_{L-1})))   [Noise.hs:82-89,130]
  centred lift    x in [0,q)  ->  x if 2x < q else x - q     (q odd: no ties)
  g_m             = 1 for two-power m, so mulG/divG are the identity on every basis
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import List, Sequence

# --------------------------------------------------------------------------------------
# number theory
# --------------------------------------------------------------------------------------

_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 3.3e24."""
    if n < 2:
        return False
    for p in _MR_BASES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _MR_BASES:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _pollard_rho(n: int) -> int:
    if n % 2 == 0:
        return 2
    c = 1
    while True:
        x = y = 2
        d = 1
        while d == 1:
            x = (x * x + c) % n
            y = (y * y + c) % n
            y = (y * y + c) % n
            d = math.gcd(abs(x - y), n)
        if d != n:
            return d
        c += 1


def prime_factors(n: int) -> List[int]:
    """Distinct prime factors of n."""
    out = set()
    stack = [n]
    while stack:
        v = stack.pop()
        if v == 1:
            continue
        if is_prime(v):
            out.add(v)
            continue
        for p in (2, 3, 5, 7, 11, 13):
            if v % p == 0:
                out.add(p)
                while v % p == 0:
                    v //= p
                stack.append(v)
                break
        else:
            d = _pollard_rho(v)
            stack.extend((d, v // d))
    return sorted(out)


def smallest_generator(q: int) -> int:
    """Smallest generator of Z_q^* (q prime).  This is the documented root rule."""
    assert is_prime(q)
    fs = prime_factors(q - 1)
    g = 2
    while True:
        if all(pow(g, (q - 1) // f, q) != 1 for f in fs):
            return g
        g += 1


def root_2n(q: int, n: int) -> int:
    """psi: the primitive 2n-th root of unity mod q fixed by the root rule."""
    assert (q - 1) % (2 * n) == 0, "q must be 1 mod m' = 2n  (Lol: CRTrans fails otherwise)"
    return pow(smallest_generator(q), (q - 1) // (2 * n), q)


def brev(k: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (k & 1)
        k >>= 1
    return r


def centred(x: int, q: int) -> int:
    """Lol's ``lift`` on ZqBasic: representative in [-(q-1)/2, (q-1)/2]."""
    x %= q
    return x if 2 * x < q else x - q


# --------------------------------------------------------------------------------------
# single-limb ring arithmetic by definition
# --------------------------------------------------------------------------------------

def crt_def(a: Sequence[int], q: int) -> List[int]:
    """Tensor ``crt`` for a two-power index, by direct evaluation (O(n^2)).
    slot k = a(psi^(2*brev(k)+1))."""
    n = len(a)
    lg = n.bit_length() - 1
    psi = root_2n(q, n)
    out = []
    for k in range(n):
        x = pow(psi, 2 * brev(k, lg) + 1, q)
        acc = 0
        for c in reversed(a):          # Horner
            acc = (acc * x + c) % q
        out.append(acc)
    return out


def crtinv_def(v: Sequence[int], q: int) -> List[int]:
    """Tensor ``crtInv``: the unique a with crt_def(a) == v, by the inverse-DFT formula."""
    n = len(v)
    lg = n.bit_length() - 1
    psi = root_2n(q, n)
    ninv = pow(n, q - 2, q)
    xs = [pow(psi, 2 * brev(k, lg) + 1, q) for k in range(n)]
    xinv = [pow(x, q - 2, q) for x in xs]
    out = []
    for i in range(n):
        acc = 0
        for k in range(n):
            acc += v[k] * pow(xinv[k], i, q)
        out.append(acc % q * ninv % q)
    return out


def negacyclic_mul(a: Sequence[int], b: Sequence[int], q: int) -> List[int]:
    """Schoolbook product in Z_q[X]/(X^n+1)."""
    n = len(a)
    out = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                out[k] += ai * bj
            else:
                out[k - n] -= ai * bj
    return [x % q for x in out]


# --------------------------------------------------------------------------------------
# RNS ring elements (Pow basis): list of L limbs, each a list of n residues
# --------------------------------------------------------------------------------------

RnsPoly = List[List[int]]


def rns_reduce(z: Sequence[int], qs: Sequence[int]) -> RnsPoly:
    """``reduce`` an integer polynomial into every limb."""
    return [[c % q for c in z] for q in qs]


def rns_add(a: RnsPoly, b: RnsPoly, qs) -> RnsPoly:
    return [[(x + y) % q for x, y in zip(al, bl)] for al, bl, q in zip(a, b, qs)]


def rns_neg(a: RnsPoly, qs) -> RnsPoly:
    return [[(-x) % q for x in al] for al, q in zip(a, qs)]


def rns_mul(a: RnsPoly, b: RnsPoly, qs) -> RnsPoly:
    return [negacyclic_mul(al, bl, q) for al, bl, q in zip(a, b, qs)]


def rns_scale(a: RnsPoly, s: Sequence[int], qs) -> RnsPoly:
    """Multiply limb j by the scalar s[j]."""
    return [[x * sj % q for x in al] for al, sj, q in zip(a, s, qs)]


def rns_zero(n: int, qs) -> RnsPoly:
    return [[0] * n for _ in qs]


# --------------------------------------------------------------------------------------
# gadgets  (Lol Gadget/Decompose on a product ring: per-limb gadgets concatenated)
# --------------------------------------------------------------------------------------

def decompose_triv(c: RnsPoly, qs) -> List[List[int]]:
    """TrivGad: one digit per limb; digit i = centred lift of limb i (integer polynomial)."""
    return [[centred(x, q) for x in cl] for cl, q in zip(c, qs)]


def baseb_digits(q: int, b: int = 2) -> int:
    """Number of base-b digits Lol's BaseBGad uses for modulus q: ceil(log_b q)."""
    k, v = 0, 1
    while v < q:
        v *= b
        k += 1
    return k


def decompose_baseb(c: RnsPoly, qs, b: int = 2) -> List[List[int]]:
    """BaseBGad b: per limb, signed base-b digits d_0..d_{k-1} of the centred lift with
    sum d_t b^t == lift, each digit in [-b/2, b/2) (balanced remainder, Lol's divModCent)."""
    out = []
    for cl, q in zip(c, qs):
        k = baseb_digits(q, b)
        digs = [[0] * len(cl) for _ in range(k)]
        for pos, x in enumerate(cl):
            v = centred(x, q)
            for t in range(k - 1):
                r = v % b
                if 2 * r >= b:
                    r -= b
                digs[t][pos] = r
                v = (v - r) // b
            digs[k - 1][pos] = v       # the most significant digit absorbs the rest
        out.extend(digs)
    return out


def gadget_triv(qs) -> List[List[int]]:
    """Gadget vector entries as per-limb scalars: g_i = (0,..,1 at limb i,..,0)."""
    L = len(qs)
    return [[1 if j == i else 0 for j in range(L)] for i in range(L)]


def gadget_baseb(qs, b: int = 2) -> List[List[int]]:
    out = []
    for i, q in enumerate(qs):
        for t in range(baseb_digits(q, b)):
            out.append([pow(b, t, qj) if j == i else 0 for j, qj in enumerate(qs)])
    return out


# --------------------------------------------------------------------------------------
# SymmSHE
# --------------------------------------------------------------------------------------

MSD, LSD = "MSD", "LSD"


@dataclass
class CT:
    """SymmSHE ciphertext ``CT enc k l c``: c is a polynomial in the secret S over R'_q
    (list of RNS ring elements in the Pow basis), k the accumulated g-power (numerically inert
    for a two-power index), l the accumulated Z_p scalar."""
    enc: str
    k: int
    l: int
    c: List[RnsPoly]
    p: int
    qs: List[int]

    @property
    def n(self) -> int:
        return len(self.c[0][0])


def _qprod(qs) -> int:
    r = 1
    for q in qs:
        r *= q
    return r


def to_lsd(ct: CT) -> CT:
    """MSD -> LSD: l *= (-q)^-1 mod p ; c *= p (mod q)."""
    if ct.enc == LSD:
        return ct
    Q = _qprod(ct.qs)
    zp = pow((-Q) % ct.p, -1, ct.p)
    s = [ct.p % q for q in ct.qs]
    return CT(LSD, ct.k, ct.l * zp % ct.p, [rns_scale(x, s, ct.qs) for x in ct.c], ct.p, ct.qs)


def to_msd(ct: CT) -> CT:
    """LSD -> MSD: l *= (-q) mod p ; c *= p^-1 (mod q)."""
    if ct.enc == MSD:
        return ct
    Q = _qprod(ct.qs)
    s = [pow(ct.p, -1, q) for q in ct.qs]
    return CT(MSD, ct.k, ct.l * ((-Q) % ct.p) % ct.p, [rns_scale(x, s, ct.qs) for x in ct.c], ct.p, ct.qs)


def gaussian_poly(n: int, r: float, rng: random.Random) -> List[int]:
    """Rounded Gaussian with parameter r per (decoding = power basis) coefficient
    (sigma = r/sqrt(2 pi)); setup-time only, exact distribution is not a parity matter."""
    sigma = r / math.sqrt(2 * math.pi)
    return [int(round(rng.gauss(0.0, sigma))) for _ in range(n)]


def gen_sk(n: int, r: float, rng: random.Random) -> List[int]:
    return gaussian_poly(n, r, rng)


def embed_pow(pt: Sequence[int], n: int) -> List[int]:
    """embed R_m -> R_m' on the power basis (two-power indices): X_m -> X_m'^(n'/n_pt)."""
    d = n // len(pt)
    out = [0] * n
    for i, c in enumerate(pt):
        out[i * d] = c
    return out


def twace_pow(x: Sequence[int], npt: int) -> List[int]:
    d = len(x) // npt
    return [x[i * d] for i in range(npt)]


def encrypt(sk: Sequence[int], pt: Sequence[int], p: int, qs, r: float, rng: random.Random) -> CT:
    """LSD encryption: c0 + c1*s = e with e = embed(pt) (mod p), k = 0, l = 1."""
    n = len(sk)
    m = embed_pow(pt, n)
    e = [mi + p * gi for mi, gi in zip(m, gaussian_poly(n, r, rng))]
    c1 = [[rng.randrange(q) for _ in range(n)] for q in qs]
    sq = rns_reduce(sk, qs)
    c0 = rns_add(rns_reduce(e, qs), rns_neg(rns_mul(c1, sq, qs), qs), qs)
    return CT(LSD, 0, 1, [c0, c1], p, list(qs))


def _crt_lift(res: Sequence[int], qs) -> int:
    """Centred representative mod Q of an RNS value."""
    Q = _qprod(qs)
    acc = 0
    for x, q in zip(res, qs):
        Qi = Q // q
        acc += x * Qi * pow(Qi, -1, q)
    acc %= Q
    return acc if 2 * acc < Q else acc - Q


def error_term(sk: Sequence[int], ct: CT) -> List[int]:
    """Centred lift of c(s) for the LSD form of ct."""
    ct = to_lsd(ct)
    sq = rns_reduce(sk, ct.qs)
    acc = rns_zero(ct.n, ct.qs)
    for comp in reversed(ct.c):            # Horner in S
        acc = rns_add(rns_mul(acc, sq, ct.qs), comp, ct.qs)
    return [_crt_lift([acc[j][i] for j in range(len(ct.qs))], ct.qs) for i in range(ct.n)]


def decrypt(sk: Sequence[int], ct: CT, npt: int) -> List[int]:
    """mu = l * g^-k * (c(s) mod p), twaced to the plaintext ring (g = 1 here)."""
    lsd = to_lsd(ct)
    e = error_term(sk, lsd)
    return [lsd.l * x % ct.p for x in twace_pow(e, npt)]


def ct_mul(a: CT, b: CT) -> CT:
    """SymmSHE ``(*)``: both to LSD, polynomial product in S, mulG on every coefficient
    (identity here), k = k1+k2+1, l = l1*l2.  [Eval.hs:65-67]"""
    a, b = to_lsd(a), to_lsd(b)
    qs = a.qs
    out = [rns_zero(a.n, qs) for _ in range(len(a.c) + len(b.c) - 1)]
    for i, x in enumerate(a.c):
        for j, y in enumerate(b.c):
            out[i + j] = rns_add(out[i + j], rns_mul(x, y, qs), qs)
    return CT(LSD, a.k + b.k + 1, a.l * b.l % a.p, out, a.p, qs)


def ct_add(a: CT, b: CT) -> CT:
    """SymmSHE ``(+)`` for operands that already agree in k and l (all this path needs)."""
    assert a.k == b.k and a.l == b.l, "operands must be aligned"
    if a.enc != b.enc:
        a, b = to_msd(a), to_msd(b)
    m = max(len(a.c), len(b.c))
    z = rns_zero(a.n, a.qs)
    c = [rns_add(a.c[i] if i < len(a.c) else z, b.c[i] if i < len(b.c) else z, a.qs) for i in range(m)]
    return CT(a.enc, a.k, a.l, c, a.p, a.qs)


@dataclass
class KSHint:
    """KSQuadCircHint: one degree-1 polynomial (h0, h1) over R'_q per gadget digit."""
    gadget: str
    h: List[List[RnsPoly]] = field(default_factory=list)   # h[i] = [h0_i, h1_i]


def ks_quad_circ_hint(sk: Sequence[int], qs, r: float, rng: random.Random, gadget: str = "triv") -> KSHint:
    """hint_i = g_i * s^2 + (LWE sample under s):  h0_i + h1_i*s = g_i s^2 + e_i.
    [KeysHints.hs:101-113 -> Lol ksQuadCircHint]"""
    n = len(sk)
    sq = rns_reduce(sk, qs)
    s2 = rns_mul(sq, sq, qs)
    gs = gadget_triv(qs) if gadget == "triv" else gadget_baseb(qs, 2)
    hint = KSHint(gadget)
    for g in gs:
        e = rns_reduce(gaussian_poly(n, r, rng), qs)
        h1 = [[rng.randrange(q) for _ in range(n)] for q in qs]
        h0 = rns_add(rns_add(rns_scale(s2, g, qs), e, qs), rns_neg(rns_mul(h1, sq, qs), qs), qs)
        hint.h.append([h0, h1])
    return hint


def key_switch_quad_circ(hint: KSHint, ct: CT) -> CT:
    """keySwitchQuadCirc: toMSD; [c0,c1] + sum_i reduce(d_i) * hint_i, d = decompose c2.
    [Eval.hs:133]"""
    ct = to_msd(ct)
    if len(ct.c) < 3:
        return ct
    assert len(ct.c) == 3
    qs = ct.qs
    digs = decompose_triv(ct.c[2], qs) if hint.gadget == "triv" else decompose_baseb(ct.c[2], qs, 2)
    assert len(digs) == len(hint.h)
    c0, c1 = ct.c[0], ct.c[1]
    for d, (h0, h1) in zip(digs, hint.h):
        dr = rns_reduce(d, qs)
        c0 = rns_add(c0, rns_mul(dr, h0, qs), qs)
        c1 = rns_add(c1, rns_mul(dr, h1, qs), qs)
    return CT(MSD, ct.k, ct.l, [c0, c1], ct.p, qs)


def ct_mul_relin(hint: KSHint, a: CT, b: CT) -> CT:
    """The hot path named by BASELINE.json: keySwitchQuadCirc hint (a * b)."""
    return key_switch_quad_circ(hint, ct_mul(a, b))


# --- RNS rescale (modSwitch), SURVEY section 8(f) N1 ---------------------------------

def rescale_up(x: RnsPoly, qs_old, qs_new_front) -> RnsPoly:
    """Rescale a -> (b, a): new limbs in front hold 0, old limbs are multiplied by the new
    primes.  (x -> x * prod(new) in the bigger modulus.)"""
    n = len(x[0])
    mult = _qprod(qs_new_front)
    return [[0] * n for _ in qs_new_front] + [[c * mult % q for c in xl] for xl, q in zip(x, qs_old)]


def rescale_down(x: RnsPoly, qs, drop: int) -> RnsPoly:
    """Rescale (a, b) -> b, dropping the first ``drop`` limbs one at a time (outermost first):
    b' = q_a^-1 * (b - reduce(lift a))   computed coefficient-wise in the Pow (= Dec) basis."""
    qs = list(qs)
    x = [list(l) for l in x]
    for _ in range(drop):
        qa, a = qs[0], x[0]
        lifted = [centred(v, qa) for v in a]
        x = [[(v - z) * pow(qa, -1, q) % q for v, z in zip(xl, lifted)] for xl, q in zip(x[1:], qs[1:])]
        qs = qs[1:]
    return x


def mod_switch_down(ct: CT, drop: int) -> CT:
    ct = to_msd(ct)
    return CT(MSD, ct.k, ct.l, [rescale_down(c, ct.qs, drop) for c in ct.c], ct.p, ct.qs[drop:])


def mod_switch_up(ct: CT, qs_new_front) -> CT:
    ct = to_msd(ct)
    return CT(MSD, ct.k, ct.l, [rescale_up(c, ct.qs, qs_new_front) for c in ct.c], ct.p,
              list(qs_new_front) + list(ct.qs))
