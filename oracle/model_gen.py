"""Exact model of Lol's ``Tensor`` operations for an ARBITRARY cyclotomic index, by definition.

TEST INFRASTRUCTURE ONLY (see oracle/model.py for the rules: nothing under ``alchemy_amd/`` imports this).

PARITY UNPINNED.  The code these definitions stand for is Lol's (``lol`` / ``lol-cpp``, un-vendored, pinned by
branch name in /root/reference/stack.yaml:54-60); the reference holds no tests or vectors for it.  What is restated
here is the *published mathematics* of the ring-LWE toolkit (Lyubashevsky, Peikert, Regev, "A Toolkit for Ring-LWE
Cryptography", Eurocrypt'13, sections 2-6) as Lol fixes it (Crockett & Peikert, CCS'16, section 3 and appendix C):

  * the reference's composite indices: examples/Common.hs:38-54 (H0' = F11648 ... H5' = F20475), used by
    examples/HomomRLWR.hs:29-35 and examples/Tunnel.hs:26-32;
  * the Tensor methods concerned: ``crt``/``crtInv``, ``mulGPow/Dec/CRT``, ``divGPow/Dec/CRT``, ``l``/``lInv``,
    ``twacePowDec``, ``embedPow``, reached from SymmSHE's ``(*)`` (Eval.hs:65-67: ``mulG`` on every
    product coefficient), ``keySwitchQuadCirc`` (Eval.hs:133), ``modSwitch`` (Eval.hs:130: ``rescaleDec`` on c0,
    ``rescalePow`` on c1) and ``decrypt`` (PT2CT.hs:91-99: lift in the decoding basis, ``divG`` k times).

Definitions (m = prod_l p_l^e_l, primes ascending; m_l = p_l^e_l; m'_l = m_l / p_l; phi = Euler's totient):

  ring          R = Z[zeta_m];  zeta_{m_l} := zeta_m^(m/m_l);  R = tensor product of the Z[zeta_{m_l}]
  Pow basis     p_j = prod_l zeta_{m_l}^{j_l},  j_l in [phi(m_l)];  linear index = mixed radix, first factor outermost
                (for a prime power the powerful basis is the power basis; toolkit Def. 4.1)
  Dec basis     d^T = p^T L,  L = kron_l (L_{p_l} (x) I_{m'_l}),  L_p = lower-triangular all-ones (p-1 x p-1)
                (decoding basis of R = (mhat/g) x decoding basis of R^dual; toolkit 6.3, Lol appendix C);  L_2 = (1)
  l / lInv      Dec coefficients -> Pow coefficients = L c  (prefix sums along j0 of every odd-prime axis), and back
  g             g_m = prod_{odd p | m} (1 - zeta_p)
  CRT basis     slot s = (s_1..s_k) (first factor outermost) holds sigma_u(x), sigma_u: zeta_m -> omega_m^u, with
                omega_m = gen^((q-1)/m), gen = smallest generator of Z_q^* (the root rule of oracle/model.py), and
                u = i0 + p i1 (mod m_l) on axis l where s_l = (i0 - 1) m'_l + digitrev_p(i1), i0 in [1, p-1],
                i1 in [m'_l].  For m = 2^k this is slot k <-> psi^(2 brev(k) + 1), the two-power rule of model.py.
                (The slot ORDER is instance-internal in Lol -- only crtInv . crt = id and the ring homomorphism
                are observable; this order is the one the sparse decomposition CRT_{p^e} = (DFT_{m'} (x) I) T (I (x)
                CRT_p) with decimation-in-frequency DFTs produces in place.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .model import centred, is_prime, smallest_generator


# --------------------------------------------------------------------------------------
# index bookkeeping
# --------------------------------------------------------------------------------------

def factor(m: int):
    """[(p, e)] with primes ascending."""
    out, p = [], 2
    while p * p <= m:
        if m % p == 0:
            e = 0
            while m % p == 0:
                m //= p
                e += 1
            out.append((p, e))
        p += 1 if p == 2 else 2
    if m > 1:
        out.append((m, 1))
    return out


def totient(m: int) -> int:
    r = 1
    for p, e in factor(m):
        r *= (p - 1) * p ** (e - 1)
    return r


def digitrev(x: int, p: int, digits: int) -> int:
    r = 0
    for _ in range(digits):
        r = r * p + x % p
        x //= p
    return r


class Index:
    """Factored cyclotomic index with the mixed-radix bookkeeping of the Pow / Dec / CRT bases."""

    def __init__(self, m: int):
        assert m >= 1
        self.m = m
        self.pps = factor(m)
        self.mls = [p ** e for p, e in self.pps]
        self.dims = [(p - 1) * p ** (e - 1) for p, e in self.pps]
        self.n = 1
        for d in self.dims:
            self.n *= d
        # strides of the mixed-radix linear index (first factor outermost)
        self.strides = []
        s = self.n
        for d in self.dims:
            s //= d
            self.strides.append(s)

    def unravel(self, lin: int):
        return [(lin // s) % d for s, d in zip(self.strides, self.dims)]

    def ravel(self, idx) -> int:
        return sum(i * s for i, s in zip(idx, self.strides))

    def pow_exponent(self, lin: int) -> int:
        """p_j = zeta_m^e(j)."""
        return sum(j * (self.m // ml) for j, ml in zip(self.unravel(lin), self.mls)) % self.m

    def axis_unit(self, l: int, s: int) -> int:
        """Unit of Z_{m_l}^* of slot s of axis l."""
        p, e = self.pps[l]
        mp = p ** (e - 1)
        i0, r = s // mp + 1, s % mp
        i1 = digitrev(r, p, e - 1)
        return (i0 + p * i1) % (p ** e)

    def slot_unit(self, lin: int) -> int:
        """Unit u of Z_m^* of CRT slot `lin` (Chinese remaindering of the per-axis units)."""
        u, mod = 0, 1
        for l, s in enumerate(self.unravel(lin)):
            ml = self.mls[l]
            ul = self.axis_unit(l, s)
            # solve x = u (mod mod), x = ul (mod ml)
            t = (ul - u) * pow(mod, -1, ml) % ml if mod > 1 else ul
            u, mod = u + mod * t, mod * ml
        return u % self.m


def omega_m(q: int, m: int) -> int:
    """omega_m = gen^((q-1)/m): the root rule (same as model.root_2n for m = 2n a power of two)."""
    assert is_prime(q) and (q - 1) % m == 0, "q must be prime and 1 mod m (Lol: crtFuncs = Nothing otherwise)"
    return pow(smallest_generator(q), (q - 1) // m, q)


# --------------------------------------------------------------------------------------
# crt / crtInv by definition
# --------------------------------------------------------------------------------------

def crt_def(a: Sequence[int], idx: Index, q: int) -> List[int]:
    """slot s = sum_j a_j omega_m^(u(s) e(j)): direct evaluation over the whole index (O(n^2), numpy rows)."""
    n, m = idx.n, idx.m
    w = omega_m(q, m)
    powers = np.empty(m, dtype=np.uint64)
    acc = 1
    for t in range(m):
        powers[t] = acc
        acc = acc * w % q
    ex = np.array([idx.pow_exponent(j) for j in range(n)], dtype=np.int64)
    av = np.array([x % q for x in a], dtype=np.uint64)
    out = []
    for s in range(n):
        u = idx.slot_unit(s)
        row = powers[(ex * u) % m]
        out.append(int(((row * av) % np.uint64(q)).sum(dtype=np.uint64) % np.uint64(q)))   # n * q < 2^64 for q < 2^31, n < 2^33
    return out


def _mat_inv_mod(M: List[List[int]], q: int) -> List[List[int]]:
    n = len(M)
    A = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c] % q)
        A[c], A[piv] = A[piv], A[c]
        inv = pow(A[c][c], -1, q)
        A[c] = [x * inv % q for x in A[c]]
        for r in range(n):
            if r != c and A[r][c]:
                f = A[r][c]
                A[r] = [(x - f * y) % q for x, y in zip(A[r], A[c])]
    return [r[n:] for r in A]


def axis_crt_matrix(idx: Index, l: int, q: int) -> List[List[int]]:
    """CRT_{m_l}[s][j] = omega_{m_l}^(u(s) j), by definition."""
    ml, d = idx.mls[l], idx.dims[l]
    w = pow(omega_m(q, idx.m), idx.m // ml, q)
    return [[pow(w, idx.axis_unit(l, s) * j % ml, q) for j in range(d)] for s in range(d)]


def apply_axis(a: Sequence[int], idx: Index, l: int, M: List[List[int]], q: Optional[int]) -> List[int]:
    """out[.., s, ..] = sum_j M[s][j] a[.., j, ..] along axis l (q = None: over the integers)."""
    n, d, st = idx.n, idx.dims[l], idx.strides[l]
    out = [0] * n
    for base in range(n):
        if (base // st) % d:
            continue
        col = [a[base + j * st] for j in range(d)]
        for s in range(d):
            v = sum(M[s][j] * col[j] for j in range(d))
            out[base + s * st] = v % q if q else v
    return out


def crt_kron(a: Sequence[int], idx: Index, q: int) -> List[int]:
    """The same transform as the Kronecker product of the per-axis definitions (must equal crt_def)."""
    x = [v % q for v in a]
    for l in range(len(idx.pps)):
        x = apply_axis(x, idx, l, axis_crt_matrix(idx, l, q), q)
    return x


def crtinv_def(v: Sequence[int], idx: Index, q: int) -> List[int]:
    """crtInv: the unique a with crt(a) = v (per-axis matrix inverses by Gaussian elimination)."""
    x = [t % q for t in v]
    for l in range(len(idx.pps)):
        x = apply_axis(x, idx, l, _mat_inv_mod(axis_crt_matrix(idx, l, q), q), q)
    return x


# --------------------------------------------------------------------------------------
# ring product by definition (schoolbook in Z[X]/(X^m - 1), reduced to the powerful basis)
# --------------------------------------------------------------------------------------

def _reduce_to_pow(coef_by_exp, idx: Index, q: Optional[int]) -> List[int]:
    """sum_t c_t zeta_m^t  ->  Pow-basis coefficients.  zeta_m^t = prod_l zeta_{m_l}^{t_l}; on every axis
    zeta_{p^e}^((p-1) m' + r) = - sum_{i < p-1} zeta_{p^e}^(i m' + r)   (Phi_{p^e}(Y) = 1 + Y^m' + ... + Y^((p-1) m'))."""
    k = len(idx.pps)
    T = np.zeros(idx.mls if k else [1], dtype=object)
    invs = [pow(idx.m // ml, -1, ml) if ml > 1 else 0 for ml in idx.mls]
    for t, c in coef_by_exp.items():
        if c:
            T[tuple(t * inv % ml for inv, ml in zip(invs, idx.mls))] += c
    for l, ((p, e), d) in enumerate(zip(idx.pps, idx.dims)):
        mp = p ** (e - 1)
        T = np.moveaxis(T, l, 0)
        lo = T[:d].copy()
        for r in range(mp):
            top = T[d + r]
            for i in range(p - 1):
                lo[i * mp + r] = lo[i * mp + r] - top
        T = np.moveaxis(lo, 0, l)
    flat = [int(x) for x in T.reshape(-1)]
    return [x % q for x in flat] if q else flat


def ring_mul_def(a: Sequence[int], b: Sequence[int], idx: Index, q: Optional[int]) -> List[int]:
    ex = [idx.pow_exponent(j) for j in range(idx.n)]
    acc = {}
    for i, ai in enumerate(a):
        if not ai:
            continue
        for j, bj in enumerate(b):
            if bj:
                t = (ex[i] + ex[j]) % idx.m
                acc[t] = acc.get(t, 0) + ai * bj
    return _reduce_to_pow(acc, idx, q)


def g_pow(idx: Index) -> List[int]:
    """g_m = prod_{odd p | m} (1 - zeta_p) in the Pow basis (integers)."""
    g = [0] * idx.n
    g[0] = 1
    for l, (p, e) in enumerate(idx.pps):
        if p == 2:
            continue
        f = [0] * idx.n
        f[0] = 1
        f[p ** (e - 1) * idx.strides[l]] -= 1          # zeta_p = zeta_{p^e}^(p^(e-1)): index j_l = m'_l
        g = ring_mul_def(g, f, idx, None)
    return g


# --------------------------------------------------------------------------------------
# l / lInv, mulG / divG by definition
# --------------------------------------------------------------------------------------

def _axis_L(idx: Index, l: int, inverse: bool) -> List[List[int]]:
    p, e = idx.pps[l]
    mp, d = p ** (e - 1), idx.dims[l]
    M = [[0] * d for _ in range(d)]
    for i0 in range(p - 1):
        for j0 in range(p - 1):
            v = (1 if j0 <= i0 else 0) if not inverse else (1 if j0 == i0 else -1 if j0 == i0 - 1 else 0)
            for r in range(mp):
                M[i0 * mp + r][j0 * mp + r] = v
    return M


def l_def(c: Sequence[int], idx: Index, q: Optional[int]) -> List[int]:
    """Dec coefficients -> Pow coefficients."""
    x = list(c)
    for l in range(len(idx.pps)):
        x = apply_axis(x, idx, l, _axis_L(idx, l, False), q)
    return x


def linv_def(a: Sequence[int], idx: Index, q: Optional[int]) -> List[int]:
    x = list(a)
    for l in range(len(idx.pps)):
        x = apply_axis(x, idx, l, _axis_L(idx, l, True), q)
    return x


def mulg_pow_def(a: Sequence[int], idx: Index, q: Optional[int]) -> List[int]:
    return ring_mul_def(g_pow(idx), a, idx, q)


def mulg_dec_def(c: Sequence[int], idx: Index, q: Optional[int]) -> List[int]:
    return linv_def(mulg_pow_def(l_def(c, idx, q), idx, q), idx, q)


def odd_rad(idx: Index) -> int:
    r = 1
    for p, _ in idx.pps:
        if p != 2:
            r *= p
    return r


def divg_pow_def(a: Sequence[int], idx: Index, q: Optional[int]) -> Optional[List[int]]:
    """The b with g b = a, or None (Lol's Nothing).  rad = prod of the odd primes of m;  rad / g is in R, so
    b = (rad / g) a / rad: over Z_q that needs rad invertible mod q, over Z every coefficient divisible by rad."""
    rad = odd_rad(idx)
    h = [0] * idx.n                                   # rad / g = prod_p (p / (1 - zeta_p)) = prod_p sum_i (p-1-i) zeta_p^i
    h[0] = 1
    for l, (p, e) in enumerate(idx.pps):
        if p == 2:
            continue
        f = [0] * idx.n
        for i in range(p - 1):
            f[i * p ** (e - 1) * idx.strides[l]] = p - 1 - i
        h = ring_mul_def(h, f, idx, None)
    t = ring_mul_def(h, a, idx, q)
    if q:
        try:
            inv = pow(rad, -1, q)
        except ValueError:
            return None
        return [x * inv % q for x in t]
    if any(x % rad for x in t):
        return None
    return [x // rad for x in t]


def divg_dec_def(c: Sequence[int], idx: Index, q: Optional[int]) -> Optional[List[int]]:
    r = divg_pow_def(l_def(c, idx, q), idx, q)
    return None if r is None else linv_def(r, idx, q)


def g_crt(idx: Index, q: int) -> List[int]:
    """CRT-basis image of g: slot s = prod_p (1 - omega_p^u(s))."""
    w = omega_m(q, idx.m)
    out = []
    for s in range(idx.n):
        u = idx.slot_unit(s)
        v = 1
        for p, _ in idx.pps:
            if p != 2:
                v = v * (1 - pow(w, (idx.m // p) * u % idx.m, q)) % q
        out.append(v)
    return out


# --------------------------------------------------------------------------------------
# twace / embed on the Pow (and Dec) basis, m | m'
# --------------------------------------------------------------------------------------

def embed_indices(small: Index, big: Index) -> List[int]:
    """Pow-basis index of R_m's basis element j inside R_m' (embedPow is this injection; twacePowDec reads the
    same positions back, in the Pow and in the Dec basis alike)."""
    assert big.m % small.m == 0
    pos = []
    sp = dict(small.pps)
    for j in range(small.n):
        js = dict(zip([p for p, _ in small.pps], small.unravel(j)))
        idxb = []
        for (p, eb) in big.pps:
            es = sp.get(p, 0)
            idxb.append(js[p] * p ** (eb - es) if es else 0)
        pos.append(big.ravel(idxb))
    return pos


def embed_pow(a: Sequence[int], small: Index, big: Index) -> List[int]:
    out = [0] * big.n
    for j, pos in enumerate(embed_indices(small, big)):
        out[pos] = a[j]
    return out


def twace_pow_dec(a: Sequence[int], small: Index, big: Index) -> List[int]:
    return [a[pos] for pos in embed_indices(small, big)]


# --------------------------------------------------------------------------------------
# RNS helpers and SymmSHE on a general index
# --------------------------------------------------------------------------------------

def rns(fn, x, idx, qs, *args):
    return [fn(xl, idx, q, *args) for xl, q in zip(x, qs)]


def rns_ring_mul(a, b, idx: Index, qs):
    return [ring_mul_def(al, bl, idx, q) for al, bl, q in zip(a, b, qs)]


def lift_dec(x, idx: Index, qs) -> List[int]:
    """Lol's liftDec: centred lift (mod Q = prod q) of the Dec-basis coefficients of an RNS ring element (Pow in)."""
    from .model import _crt_lift
    dec = [linv_def(xl, idx, q) for xl, q in zip(x, qs)]
    return [_crt_lift([dec[j][i] for j in range(len(qs))], qs) for i in range(idx.n)]


def rescale_down_basis(x, idx: Index, qs, drop: int, basis: str):
    """Rescale (a, b) -> b, `drop` limbs, coefficient-wise in the Pow or the Dec basis (x given and returned in the
    Pow basis): modSwitch rescales c0 with rescaleDec and c1 with rescalePow."""
    from .model import rescale_down
    if basis == "pow":
        return rescale_down(x, qs, drop)
    dec = [linv_def(xl, idx, q) for xl, q in zip(x, qs)]
    y = rescale_down(dec, qs, drop)
    return [l_def(yl, idx, q) for yl, q in zip(y, qs[drop:])]


# --------------------------------------------------------------------------------------
# SymmSHE on a general index (Lol's CT with the g-power k live): the semantic pin of every convention above --
# decrypt(modSwitch(keySwitchQuadCirc hint (modSwitch (a * b)))) must equal the product of the plaintexts.
# Call sites: encrypt PT2CT.hs:84-87, (*) Eval.hs:65-67, modSwitch Eval.hs:130, keySwitchQuadCirc Eval.hs:133,
# ksQuadCircHint KeysHints.hs:101-113, decrypt PT2CT.hs:91-99, addPublic / mulPublic Eval.hs:131-132,
# modSwitchPT Eval.hs:129.
# --------------------------------------------------------------------------------------
import random
from dataclasses import dataclass

from .model import LSD, MSD, _crt_lift, _qprod, decompose_baseb, decompose_triv, gadget_baseb, gadget_triv, rescale_up


@dataclass
class GCT:
    """CT enc k l c over R'_q, q = prod qs; c = list of RNS ring elements in the Pow basis of the index `big`."""
    enc: str
    k: int
    l: int
    c: list
    p: int
    qs: list
    big: Index
    small: Index


def _small_dec(n: int, bound: int, rng: random.Random) -> List[int]:
    return [rng.randint(-bound, bound) for _ in range(n)]


def g_gen_sk(big: Index, rng: random.Random, bound: int = 2) -> List[int]:
    """Secret key: short in the decoding basis (Lol: genSK = rounded tweaked Gaussian, a Dec-basis object); returned as
    integer Pow coefficients."""
    return l_def(_small_dec(big.n, bound, rng), big, None)


def _rns_scale(x, s, qs):
    return [[v * sj % q for v in xl] for xl, sj, q in zip(x, s, qs)]


def g_to_lsd(ct: GCT) -> GCT:
    if ct.enc == LSD:
        return ct
    Q = _qprod(ct.qs)
    return GCT(LSD, ct.k, ct.l * pow((-Q) % ct.p, -1, ct.p) % ct.p, [_rns_scale(x, [ct.p % q for q in ct.qs], ct.qs) for x in ct.c],
               ct.p, ct.qs, ct.big, ct.small)


def g_to_msd(ct: GCT) -> GCT:
    if ct.enc == MSD:
        return ct
    Q = _qprod(ct.qs)
    return GCT(MSD, ct.k, ct.l * ((-Q) % ct.p) % ct.p, [_rns_scale(x, [pow(ct.p, -1, q) for q in ct.qs], ct.qs) for x in ct.c],
               ct.p, ct.qs, ct.big, ct.small)


def g_encrypt(sk, pt_pow, small: Index, big: Index, p: int, qs, rng: random.Random, bound: int = 2) -> GCT:
    """LSD encryption: c0 + c1 s = e, e = embed(pt) (mod p R'), e short in the decoding basis (Lol: errorCoset)."""
    emb = embed_pow([x % p for x in pt_pow], small, big)
    ptdec = [centred(x, p) for x in linv_def(emb, big, p)]
    e = l_def([v + p * t for v, t in zip(ptdec, _small_dec(big.n, bound, rng))], big, None)
    c1 = [[rng.randrange(q) for _ in range(big.n)] for q in qs]
    c0 = [[(ev - v) % q for ev, v in zip(e, ring_mul_def(c1l, sk, big, q))] for c1l, q in zip(c1, qs)]
    return GCT(LSD, 0, 1, [c0, c1], p, list(qs), big, small)


def g_ct_mul(a: GCT, b: GCT) -> GCT:
    """(*): both to LSD, product of the polynomials in S, mulG on every coefficient, k = k1 + k2 + 1, l = l1 l2."""
    a, b = g_to_lsd(a), g_to_lsd(b)
    qs, big = a.qs, a.big
    out = [[[0] * big.n for _ in qs] for _ in range(len(a.c) + len(b.c) - 1)]
    for i, x in enumerate(a.c):
        for j, y in enumerate(b.c):
            pr = rns_ring_mul(x, y, big, qs)
            out[i + j] = [[(u + v) % q for u, v in zip(ol, pl)] for ol, pl, q in zip(out[i + j], pr, qs)]
    out = [[mulg_pow_def(cl, big, q) for cl, q in zip(c, qs)] for c in out]
    return GCT(LSD, a.k + b.k + 1, a.l * b.l % a.p, out, a.p, qs, big, a.small)


def g_ks_hint(sk, big: Index, qs, rng: random.Random, bound: int = 2):
    """TrivGad KSQuadCircHint: h0_i + h1_i s = g_i s^2 + e_i."""
    hint = []
    for g in gadget_triv(qs):
        e = l_def(_small_dec(big.n, bound, rng), big, None)
        h1 = [[rng.randrange(q) for _ in range(big.n)] for q in qs]
        h0 = []
        for j, q in enumerate(qs):
            s2 = ring_mul_def(sk, sk, big, q)
            h1s = ring_mul_def(h1[j], sk, big, q)
            h0.append([(g[j] * a + ev - b) % q for a, ev, b in zip(s2, e, h1s)])
        hint.append([h0, h1])
    return hint


def g_key_switch(hint, ct: GCT) -> GCT:
    ct = g_to_msd(ct)
    assert len(ct.c) == 3
    qs, big = ct.qs, ct.big
    c0, c1 = ct.c[0], ct.c[1]
    for d, (h0, h1) in zip(decompose_triv(ct.c[2], qs), hint):
        dr = [[v % q for v in d] for q in qs]
        c0 = [[(u + v) % q for u, v in zip(cl, pl)] for cl, pl, q in zip(c0, rns_ring_mul(dr, h0, big, qs), qs)]
        c1 = [[(u + v) % q for u, v in zip(cl, pl)] for cl, pl, q in zip(c1, rns_ring_mul(dr, h1, big, qs), qs)]
    return GCT(MSD, ct.k, ct.l, [c0, c1], ct.p, qs, big, ct.small)


def g_mod_switch_up(ct: GCT, qs_front) -> GCT:
    ct = g_to_msd(ct)
    return GCT(MSD, ct.k, ct.l, [rescale_up(c, ct.qs, qs_front) for c in ct.c], ct.p, list(qs_front) + list(ct.qs), ct.big, ct.small)


def g_mod_switch_down(ct: GCT, drop: int) -> GCT:
    """modSwitch: rescaleDec on c0, rescalePow on the higher coefficients."""
    ct = g_to_msd(ct)
    c = [rescale_down_basis(x, ct.big, ct.qs, drop, "dec" if i == 0 else "pow") for i, x in enumerate(ct.c)]
    return GCT(MSD, ct.k, ct.l, c, ct.p, ct.qs[drop:], ct.big, ct.small)


def g_mod_switch_pt(ct: GCT, p_new: int) -> GCT:
    """modSwitchPT (PT2CT's div2_): MSD form, plaintext modulus p -> p_new | p; the ring elements do not change."""
    ct = g_to_msd(ct)
    assert ct.p % p_new == 0
    return GCT(MSD, ct.k, ct.l % p_new, ct.c, p_new, ct.qs, ct.big, ct.small)


def g_mul_public(pub_pow, ct: GCT) -> GCT:
    """mulPublic a ct: every coefficient times embed(reduce(liftPow a))."""
    a = embed_pow([centred(x, ct.p) for x in pub_pow], ct.small, ct.big)
    return GCT(ct.enc, ct.k, ct.l, [[ring_mul_def(cl, a, ct.big, q) for cl, q in zip(c, ct.qs)] for c in ct.c], ct.p, ct.qs,
               ct.big, ct.small)


def g_add_public(pub_pow, ct: GCT) -> GCT:
    """addPublic b ct: LSD form, c0 += mulG^k (embed (reduce (liftPow (l^-1 b))))."""
    ct = g_to_lsd(ct)
    linv = pow(ct.l, -1, ct.p)
    b = embed_pow([centred(x * linv % ct.p, ct.p) for x in pub_pow], ct.small, ct.big)
    c0 = []
    for cl, q in zip(ct.c[0], ct.qs):
        t = [v % q for v in b]
        for _ in range(ct.k):
            t = mulg_pow_def(t, ct.big, q)
        c0.append([(u + v) % q for u, v in zip(cl, t)])
    return GCT(LSD, ct.k, ct.l, [c0] + ct.c[1:], ct.p, ct.qs, ct.big, ct.small)


def g_absorb_g_factors(ct: GCT) -> GCT:
    """SymmSHE absorbGFactors (what `tunnel` runs first on a ciphertext with k > 0): every component times
    reduce(liftPow d), d = g^-k in R'_p (`iterate divG one !! k`), k <- 0.  d g^k = 1 (mod p R'), so the plaintext is unchanged."""
    if ct.k == 0:
        return ct
    d = [1] + [0] * (ct.big.n - 1)
    for _ in range(ct.k):
        d = divg_pow_def(d, ct.big, ct.p)
        assert d is not None
    dz = [centred(v, ct.p) for v in d]
    c = [[ring_mul_def(cl, dz, ct.big, q) for cl, q in zip(comp, ct.qs)] for comp in ct.c]
    return GCT(ct.enc, 0, ct.l, c, ct.p, ct.qs, ct.big, ct.small)


def g_decrypt(sk, ct: GCT) -> List[int]:
    """Pow coefficients (mod p) of the plaintext: l * twace(g^-k * (liftDec(c(s)) mod p))."""
    ct = g_to_lsd(ct)
    big, qs = ct.big, ct.qs
    acc = [[0] * big.n for _ in qs]
    for comp in reversed(ct.c):                            # Horner in S
        acc = [[(u + v) % q for u, v in zip(ring_mul_def(al, sk, big, q), cl)] for al, cl, q in zip(acc, comp, qs)]
    e = [v % ct.p for v in lift_dec(acc, big, qs)]          # Dec coefficients mod p
    for _ in range(ct.k):
        e = divg_dec_def(e, big, ct.p)
        assert e is not None
    t = twace_pow_dec(e, ct.small, big)                     # Dec coefficients of the plaintext
    return [ct.l * v % ct.p for v in l_def(t, ct.small, ct.p)]


# --------------------------------------------------------------------------------------
# Ring tunnelling: tunnel_ hint between two modSwitch_ (PT2CT.hs:224-229, Eval.hs:134), tunnelHint
# (KeysHints.hs:120-129), linearDec / the five hops of examples/Common.hs:65-95.  Published algorithm: Crockett-Peikert
# CCS'16 section 4.4 ("ring switching") / Lol's SymmSHE.tunnel:
#   E = R cap S, E' = R' cap S'; f: R -> S is E-linear, given by its values y_i on the relative decoding basis of R/E
#   (linearDec); it extends E'-linearly to f': R' -> S' with the same values (embedded).  For ct = (c0, c1) over R'_q (MSD,
#   k = 0):      c0' = f'(c0) = sum_i y_i embed(coeffsDec(c0)_i)
#                c1  = sum_i c1_i p_i   (c1_i in E': coefficients on the relative powerful basis of R'/E')
#                result = (c0', 0) + sum_i switch(hint_i, embed(c1_i)),   hint_i encrypting f'(s_in p_i) under s_out.
# The relative bases: for p^e | p^e', index j' of the big ring's axis = j1 + p^(e'-e) j with j the small ring's index and
# j1 in [p^(e'-e)] the relative index (the same split for the Pow and the Dec basis -- both factor as relative (x) base,
# which is why Lol's Tensor has ONE `coeffs`); a prime that does not divide the small index contributes its whole axis.
# --------------------------------------------------------------------------------------

def coeffs_indices(small: Index, big: Index) -> List[List[int]]:
    """[relative index i][small index j] -> index into the big ring (Pow or Dec alike).  Relative indices are ordered as the
    mixed radix of the per-prime relative indices, first prime outermost."""
    assert big.m % small.m == 0
    sp = dict(small.pps)
    rel_dims = []
    for p, eb in big.pps:
        es = sp.get(p, 0)
        rel_dims.append(p ** (eb - es) if es else (p - 1) * p ** (eb - 1))
    nrel = 1
    for d in rel_dims:
        nrel *= d
    assert nrel * small.n == big.n
    out = []
    for i in range(nrel):
        rel, t = [], i
        for d in reversed(rel_dims):
            rel.append(t % d)
            t //= d
        rel.reverse()
        row = []
        for j in range(small.n):
            js = dict(zip([p for p, _ in small.pps], small.unravel(j)))
            idxb = []
            for (p, eb), j1 in zip(big.pps, rel):
                es = sp.get(p, 0)
                idxb.append(j1 + p ** (eb - es) * js[p] if es else j1)
            row.append(big.ravel(idxb))
        out.append(row)
    return out


def coeffs(a: Sequence[int], small: Index, big: Index) -> List[List[int]]:
    """Tensor `coeffs`: the small-ring coefficient vectors of a big-ring element w.r.t. the relative Pow (or Dec) basis."""
    return [[a[pos] for pos in row] for row in coeffs_indices(small, big)]


def eval_lin_dec(ys_pow, x_dec, e: Index, r: Index, s: Index, q: Optional[int]) -> List[int]:
    """evalLin (linearDec ys) x: sum_i y_i * embed(coeffsDec(x)_i), x given by its Dec coefficients over R, ys (and the
    result) as Pow coefficients over S.  embed of a Dec-basis element = embedPow of its Pow form."""
    acc = [0] * s.n
    for y, c in zip(ys_pow, coeffs(x_dec, e, r)):
        emb = embed_pow(l_def(c, e, q), e, s)
        pr = ring_mul_def(y, emb, s, q)
        acc = [(u + v) % q if q else u + v for u, v in zip(acc, pr)]
    return acc


@dataclass
class TunnelInfo:
    e: Index
    r: Index
    s: Index
    ep: Index
    rp: Index
    sp: Index


def tunnel_indices(r: int, s: int, rp: int, sp: int) -> TunnelInfo:
    """E = R cap S, E' = R' cap S' (index gcds), with Lol's side conditions: e = gcd(r, e') and lcm(r, e') = r', so that the
    relative decoding basis of R/E is also one of R'/E' (extendLin)."""
    import math
    e, ep = math.gcd(r, s), math.gcd(rp, sp)
    assert rp % r == 0 and sp % s == 0 and math.gcd(r, ep) == e and r * ep // e == rp, "indices do not form a tunnel"
    return TunnelInfo(Index(e), Index(r), Index(s), Index(ep), Index(rp), Index(sp))


def g_tunnel_hint(ys_pow_p, T: TunnelInfo, p: int, sk_in, sk_out, qs, rng: random.Random, bound: int = 2, gadget: str = "triv"):
    """tunnelHint f skout skin: (f' mod q as S'-elements, [hint_i]) with hint_i a KSLinearHint (TrivGad, or BaseBGad 2 --
    the gadget of examples/Tunnel.hs:24) for f'(s_in p_i), p_i the relative powerful basis of R'/E'."""
    ysz = [embed_pow([centred(v, p) for v in y], T.s, T.sp) for y in ys_pow_p]           # lift f, extend to S'
    rows = coeffs_indices(T.ep, T.rp)
    hints = []
    for row in rows:
        pi = [0] * T.rp.n
        pi[row[0]] = 1                                                                   # relative Pow basis element
        x = ring_mul_def(sk_in, pi, T.rp, None)                                          # s_in * p_i  over Z
        val = eval_lin_dec(ysz, linv_def(x, T.rp, None), T.ep, T.rp, T.sp, None)         # f'(s_in p_i) over Z
        hint_i = []
        for g in (gadget_triv(qs) if gadget == "triv" else gadget_baseb(qs, 2)):
            err = l_def(_small_dec(T.sp.n, bound, rng), T.sp, None)
            a = [[rng.randrange(q) for _ in range(T.sp.n)] for q in qs]
            b = []
            for j, q in enumerate(qs):
                as_ = ring_mul_def(a[j], sk_out, T.sp, q)
                b.append([(g[j] * v + ev - w) % q for v, ev, w in zip(val, err, as_)])
            hint_i.append([b, a])
        hints.append(hint_i)
    lin_q = [[[v % q for v in y] for q in qs] for y in ysz]
    return lin_q, hints


def g_tunnel(lin_q, hints, ct: GCT, T: TunnelInfo, gadget: str = "triv") -> GCT:
    """SymmSHE.tunnel on a linear ciphertext with k = 0 (the state every ALCHEMY tunnel sees: PT2CT tunnels before it
    multiplies, examples/HomomRLWR.hs:45-50)."""
    ct = g_to_msd(ct)
    assert ct.k == 0 and len(ct.c) == 2 and ct.big.m == T.rp.m
    qs = ct.qs
    c0 = []
    for j, q in enumerate(qs):
        c0.append(eval_lin_dec([y[j] for y in lin_q], linv_def(ct.c[0][j], T.rp, q), T.ep, T.rp, T.sp, q))
    c1 = [[0] * T.sp.n for _ in qs]
    c1parts = [coeffs(ct.c[1][j], T.ep, T.rp) for j in range(len(qs))]                  # [limb][i] -> E'-element (Pow)
    for i, hint_i in enumerate(hints):
        emb = [embed_pow(c1parts[j][i], T.ep, T.sp) for j in range(len(qs))]            # RNS element of S' (Pow)
        digs = decompose_triv(emb, qs) if gadget == "triv" else decompose_baseb(emb, qs, 2)
        assert len(digs) == len(hint_i)
        for d, (b, a) in zip(digs, hint_i):
            dr = [[v % q for v in d] for q in qs]
            c0 = [[(u + v) % q for u, v in zip(cl, pl)] for cl, pl, q in zip(c0, rns_ring_mul(dr, b, T.sp, qs), qs)]
            c1 = [[(u + v) % q for u, v in zip(cl, pl)] for cl, pl, q in zip(c1, rns_ring_mul(dr, a, T.sp, qs), qs)]
    return GCT(MSD, 0, ct.l, [c0, c1], ct.p, qs, T.sp, T.s)


# --------------------------------------------------------------------------------------
# The CRT-order-dependent Tensor methods between two indices m | m' (`crtExtFuncs` = (twaceCRT, embedCRT)),
# embedDec, and the mod-p CRT set `crtSetDec` (used by Cyc's `crtSet`, examples/Common.hs:65-75 `decToCRT`).
# By definition only:
#   embedCRT   sigma_{u'}(embed x) = sigma_{u' mod m}(x)      (omega_m = omega_{m'}^(m'/m) under the root rule)
#   twace      Tw_{m'/m}(y) = (mhat / mhat') Tr_{m'/m}(y g' / g)   (toolkit 2.x / Lol "tweaked trace"), so on the CRT basis
#              Tw(y)[u] = (mhat/mhat') g(omega^u)^-1 sum_{u' = u mod m} g'(omega'^u') y[u']
#   crtSetDec  the idempotents c_k of R_{m'} / p R_{m'} with sigma_i(c_k) = [i in I_k] over GF(p^d), d = ord_{m'}(p), where the
#              I_k partition Z_{m'}^* into unions of <p>-cosets with exactly one coset above every <p>-coset of Z_m^*;
#              returned as coefficient vectors over F_p in the decoding basis.
# --------------------------------------------------------------------------------------

def mhat(m: int) -> int:
    return m // 2 if m % 2 == 0 else m


def slot_of_unit(idx: Index) -> dict:
    return {idx.slot_unit(s): s for s in range(idx.n)}


def embed_crt_def(v: Sequence[int], small: Index, big: Index) -> List[int]:
    su = slot_of_unit(small)
    return [v[su[big.slot_unit(s) % small.m]] for s in range(big.n)]


def twace_crt_def(v: Sequence[int], small: Index, big: Index, q: int) -> List[int]:
    su = slot_of_unit(small)
    gs, gb = g_crt(small, q), g_crt(big, q)
    acc = [0] * small.n
    for s in range(big.n):
        t = su[big.slot_unit(s) % small.m]
        acc[t] = (acc[t] + gb[s] * v[s]) % q
    scale = pow(mhat(big.m) // mhat(small.m), -1, q)
    return [a * pow(g, -1, q) * scale % q for a, g in zip(acc, gs)]


def embed_dec_def(c: Sequence[int], small: Index, big: Index, q: Optional[int]) -> List[int]:
    """embedDec: the Dec coefficients over the big index of an element given by its Dec coefficients over the small one."""
    return linv_def(embed_pow(l_def(c, small, q), small, big), big, q)


class GF:
    """GF(p^d) = F_p[t] / (f), f the lexicographically first monic irreducible polynomial of degree d; elements are tuples of
    d residues (constant term first)."""

    def __init__(self, p: int, d: int):
        assert is_prime(p) and d >= 1
        self.p, self.d = p, d
        self.f = self._first_irreducible()
        self.zero, self.one = tuple([0] * d), tuple([1] + [0] * (d - 1))

    # -- polynomial helpers over F_p (lists, constant term first, no trailing zeros except [0]) --
    def _trim(self, a):
        while len(a) > 1 and a[-1] == 0:
            a.pop()
        return a

    def _pmod(self, a, f):
        a, p = list(a), self.p
        df = len(f) - 1
        inv = pow(f[-1], -1, p)
        while len(a) - 1 >= df and any(a):
            self._trim(a)
            if len(a) - 1 < df:
                break
            c = a[-1] * inv % p
            sh = len(a) - 1 - df
            for i, fi in enumerate(f):
                a[sh + i] = (a[sh + i] - c * fi) % p
            self._trim(a)
        return self._trim(a)

    def _pmul(self, a, b):
        out = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    out[i + j] = (out[i + j] + x * y) % self.p
        return out

    def _pgcd(self, a, b):
        a, b = self._trim(list(a)), self._trim(list(b))
        while any(b):
            a, b = b, self._pmod(a, b)
        return a

    def _powx(self, e, f):
        """t^e mod f."""
        r, b = [1], [0, 1]
        while e:
            if e & 1:
                r = self._pmod(self._pmul(r, b), f)
            b = self._pmod(self._pmul(b, b), f)
            e >>= 1
        return r

    def _irreducible(self, f):
        """Rabin's test."""
        d, p = len(f) - 1, self.p
        x = [0, 1]
        diff = self._powx(p ** d, f)
        diff = self._trim([(u - v) % p for u, v in zip(diff + [0] * 2, x + [0] * len(diff))][:max(len(diff), 2)])
        if any(diff):
            return False
        for r, _ in factor(d):
            h = self._powx(p ** (d // r), f)
            h = self._trim([(u - v) % p for u, v in zip(h + [0] * 2, x + [0] * len(h))][:max(len(h), 2)])
            if len(self._pgcd(f, h)) != 1:
                return False
        return True

    def _first_irreducible(self):
        p, d = self.p, self.d
        if d == 1:
            return [0, 1]
        for code in range(p ** d):
            low = [(code // p ** i) % p for i in range(d)]
            f = low + [1]
            if low[0] and self._irreducible(f):
                return f
        raise AssertionError

    # -- field operations --
    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def neg(self, a):
        return tuple((-x) % self.p for x in a)

    def mul(self, a, b):
        r = self._pmod(self._pmul(list(a), list(b)), self.f)
        return tuple(r + [0] * (self.d - len(r)))

    def pow(self, a, e):
        r, b = self.one, a
        while e:
            if e & 1:
                r = self.mul(r, b)
            b = self.mul(b, b)
            e >>= 1
        return r

    def from_int(self, x):
        return tuple([x % self.p] + [0] * (self.d - 1))

    def root_of_unity(self, m: int):
        """The m-th root rule over GF(p^d): x^((p^d - 1)/m) for the first field element x (counting in base p, constant
        term least significant) for which that power has order exactly m."""
        N = self.p ** self.d - 1
        assert N % m == 0
        if m == 1:
            return self.one
        rs = [r for r, _ in factor(m)]
        for code in range(2, self.p ** self.d):
            x = tuple((code // self.p ** i) % self.p for i in range(self.d))
            w = self.pow(x, N // m)
            if all(self.pow(w, m // r) != self.one for r in rs):
                return w
        raise AssertionError


def mult_order(p: int, m: int) -> int:
    if m == 1:
        return 1
    d, x = 1, p % m
    while x != 1:
        x = x * p % m
        d += 1
    return d


def crt_set_cosets(small: Index, big: Index, p: int) -> List[List[int]]:
    """The index sets I_k (k < r): the <p>-cosets of Z_{m'}^* are grouped by the <p>-coset of Z_m^* they reduce to (groups
    ordered by the smallest element of that coset, cosets inside a group by their smallest element); I_k = union over the groups
    of each group's k-th coset.  (The rule is this library's: Lol's own ordering is not observable here -- parity unpinned.)"""
    import math
    m, mb = small.m, big.m
    assert mb % m == 0 and math.gcd(p, mb) == 1
    units = [u for u in range(1, mb + 1) if math.gcd(u, mb) == 1] if mb > 1 else [0]
    seen, cosets = set(), []
    for u in units:
        if u in seen:
            continue
        c, x = [], u
        while x not in seen:
            seen.add(x)
            c.append(x)
            x = x * p % mb
        cosets.append(sorted(c))
    groups = {}
    for c in cosets:
        red = c[0] % m if m > 1 else 0
        key, x = red, red
        for _ in range(mult_order(p, m)):
            key = min(key, x)
            x = x * p % m if m > 1 else 0
        groups.setdefault(key, []).append(c)
    r = len(cosets) // len(groups)
    assert all(len(g) == r for g in groups.values())
    return [sorted(u for key in sorted(groups) for u in sorted(groups[key], key=lambda c: c[0])[k]) for k in range(r)]


def crt_set_dec_def(small: Index, big: Index, p: int) -> List[List[int]]:
    """crtSetDec: Dec-basis coefficient vectors over F_p of the relative mod-p CRT set of O_{m'} / O_m.
    a_j = mhat'^-1 Tr(c g' conj(p_j)) = mhat'^-1 sum_{i in I} g'(w^i) w^(-i e(j))   (the dual of the decoding basis
    d = (mhat'/g') d^dual-of-conj-powerful is (g'/mhat') conj(p)); checked against the defining property in the tests."""
    mb = big.m
    d = mult_order(p, mb)
    F = GF(p, d)
    w = F.root_of_unity(mb)
    pw = [F.one]
    for _ in range(mb - 1):
        pw.append(F.mul(pw[-1], w))
    odd = [q for q, _ in big.pps if q != 2]

    def g_at(i):
        v = F.one
        for q in odd:
            v = F.mul(v, F.add(F.one, F.neg(pw[(mb // q) * i % mb])))
        return v
    inv_mhat = pow(mhat(mb), -1, p)
    ex = [big.pow_exponent(j) for j in range(big.n)]
    out = []
    for I in crt_set_cosets(small, big, p):
        gi = [(i, g_at(i)) for i in I]
        vec = []
        for j in range(big.n):
            acc = F.zero
            for i, gv in gi:
                acc = F.add(acc, F.mul(gv, pw[(-i * ex[j]) % mb]))
            assert all(x == 0 for x in acc[1:]), "not in the prime field"
            vec.append(acc[0] * inv_mhat % p)
        out.append(vec)
    return out


def eval_mod_p(pow_coeffs: Sequence[int], idx: Index, F: GF, w, unit: int):
    """sigma_unit(x) in GF(p^d) for x given by Pow coefficients over F_p (w: the m-th root of unity)."""
    acc = F.zero
    for j, c in enumerate(pow_coeffs):
        if c % F.p:
            acc = F.add(acc, F.mul(F.from_int(c), F.pow(w, unit * idx.pow_exponent(j) % idx.m)))
    return acc
