"""GPU: every ring the three reference examples instantiate, through the C ABI, following the decision sequence of the shipped
binding (haskell/Crypto/Lol/Cyclotomic/Tensor/GT.hs: `ringFor`, `powRing`, `crtFuncsGT`, `crtExtFuncsGT`) call for call
(VERDICT r03 item 1c).

The rings (reference files):
  examples/Arithmetic.hs:23-34   PT = Cyc F4 (Zq 7); ciphertexts over F512 on the prefixes of its three moduli
  examples/HomomRLWR.hs:29-47    plaintexts over H0 .. H5 on Z_{2^e}, e = 5 .. 1 (Z2E, examples/Common.hs:32; K = P5 and the
                                 rescale tree halves the modulus down to PP2); ciphertexts over H0' .. H5' on every prefix of six moduli
  examples/Tunnel.hs:26-41       PT over H3 on Zq PP8 (tunnel3: H0 .. H3); ciphertexts on the prefixes of five moduli
  keys / lifts                   Cyc t m' Int64 (the integers): getKey, decrypt's lift (KeysHints.hs:86-96, PT2CT.hs:91-99)
plus the ring R'_p = Cyc t m' zp that decrypt divides by g in (PT2CT.hs:91-99).

Contract checked: alch_ring_create answers ALCH_OK or ALCH_E_NO_CRT for every one of them -- never NOT_PRIME / UNSUPPORTED / INVALID
-- ; where it answers NO_CRT (Lol: crtFuncs = Nothing) alch_ring_create_nocrt succeeds and that ring serves l, lInv, mulGPow/Dec,
divGPow/Dec, embedPow/Dec, twacePowDec, coeffs; every status that comes back is in {OK, NO_CRT, NOT_DIVISIBLE}; results equal the
C restatement (oracle.cref.GenRing) and the by-definition model."""
import math

import numpy as np
import pytest

from alchemy_amd import capi
from oracle import model_gen as G

pytestmark = pytest.mark.gpu

H = [128, 448, 2912, 3640, 5460, 4095]
HP = [11648, 29120, 43680, 54600, 27300, 20475]
ARITH_ZQS = [268440577, 8392193, 1073750017]
RLWR_ZQS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]
TUNNEL_ZQS = [537264001, 539884801, 555609601, 560851201, 566092801]
ALLOWED = {capi.ALCH_OK, capi.ALCH_E_NO_CRT, capi.ALCH_NOT_DIVISIBLE}


def nested(zqs, L):
    """PNoise2Zq nests the prefix last-taken-outermost (Noise.hs:82-89,130): limb 0 of the ring is zqs[L-1]."""
    return list(reversed(zqs[:L]))


class Shim:
    """The decision sequence of GT.hs, in Python, with every status recorded."""

    def __init__(self):
        self.l = capi.load_library()
        self.cache = {}
        self.seen = []

    def _create(self, nocrt, m, qs):
        import ctypes as C
        h = C.c_void_p()
        arr = (C.c_uint64 * len(qs))(*qs)
        rc = (self.l.alch_ring_create_nocrt if nocrt else self.l.alch_ring_create)(m, len(qs), arr, C.byref(h))
        self.seen.append((("nocrt" if nocrt else "crt"), m, tuple(qs), rc))
        return rc, h

    def ring_for(self, nocrt, m, qs):
        """GT.hs `ringFor`: ('dev', handle) | ('nocrt-needed',) | ('lolcpp',)"""
        key = (m, tuple(qs), nocrt)
        if key not in self.cache:
            rc, h = self._create(nocrt, m, qs)
            if rc == capi.ALCH_OK:
                self.cache[key] = ("dev", h)
            elif rc == capi.ALCH_E_NO_CRT:
                self.cache[key] = ("nocrt",)
            elif rc == capi.ALCH_E_UNSUPPORTED:
                self.cache[key] = ("lolcpp",)
            else:
                raise AssertionError(f"alch_ring_create({m}, {qs}) -> {rc}: {self.l.alch_last_error().decode()}")
        return self.cache[key]

    def pow_ring(self, m, qs):
        """GT.hs `powRing`: the CRT ring when there is one, else the no-CRT ring; None = the whole (m, r) stays on lol-cpp.
        The integers (modulus 0) are not a GTDispatch type in GT.hs (`Int64` stays on lol-cpp); the compiled C++ host serves them
        from alch_ring_create_nocrt directly (alchemy_amd/host/cycgen.hpp decToPowZ), which is what is replayed for them."""
        if list(qs) == [0]:
            b = self.ring_for(True, m, qs)
            return (b[1], False) if b[0] == "dev" else (None, False)
        a = self.ring_for(False, m, qs)
        if a[0] == "dev":
            return a[1], True
        if a[0] == "lolcpp":
            return None, False
        b = self.ring_for(True, m, qs)
        return (b[1], False) if b[0] == "dev" else (None, False)

    def close(self):
        for v in self.cache.values():
            if v[0] == "dev":
                self.l.alch_ring_destroy(v[1])


class HostRing:
    """Host-buffer Tensor calls on a raw handle (what GT's GTHost constructor issues), statuses recorded."""

    def __init__(self, shim, handle, m, qs):
        self.s, self.h, self.m, self.qs, self.L = shim, handle, m, qs, len(qs)
        self.n = G.totient(m)

    def call(self, name, a):
        import ctypes as C
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        rc = getattr(self.s.l, name)(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)))
        self.s.seen.append((name, self.m, tuple(self.qs), rc))
        assert rc in ALLOWED, (name, self.m, self.qs, rc, self.s.l.alch_last_error().decode())
        return rc, out

    def ext(self, name, big, a, out_shape):
        import ctypes as C
        a = np.ascontiguousarray(a, dtype=np.int64)
        out = np.zeros(out_shape, dtype=np.int64)
        rc = getattr(self.s.l, name)(self.h, big.h, a.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_int64)))
        self.s.seen.append((name, self.m, tuple(self.qs), rc))
        assert rc in ALLOWED, (name, self.m, big.m, self.qs, rc, self.s.l.alch_last_error().decode())
        return rc, out


def rand_elem(rng, n, qs):
    cols = []
    for q in qs:
        cols.append(rng.integers(0, q, size=n, dtype=np.int64) if q else rng.integers(-1000, 1000, size=n, dtype=np.int64))
    return np.stack(cols, axis=1)


def check_pow_dec_methods(shim, oracle_lib, m, qs, expect_crt, seed):
    """powRing's ring serves l, lInv, mulGPow/Dec, divGPow/Dec -- and crtFuncsGT's tuple when a CRT basis exists."""
    handle, has_crt = shim.pow_ring(m, qs)
    assert handle is not None, (m, qs, "the reference's rings are all served")
    assert has_crt == expect_crt, (m, qs)
    r, o = HostRing(shim, handle, m, qs), oracle_lib.GenRing(m, qs)
    assert o.has_crt == expect_crt
    rng = np.random.default_rng(seed)
    x = rand_elem(rng, r.n, qs)
    for name, want in (("alch_l", o.l), ("alch_linv", o.linv), ("alch_mulg_pow", o.mulg_pow), ("alch_mulg_dec", o.mulg_dec)):
        rc, got = r.call(name, x)
        assert rc == capi.ALCH_OK and np.array_equal(got, want(x)), (name, m, qs)
    for name, want, mulg in (("alch_divg_pow", o.divg_pow, o.mulg_pow), ("alch_divg_dec", o.divg_dec, o.mulg_dec)):
        rc, got = r.call(name, x)
        w = want(x)
        assert (rc == capi.ALCH_NOT_DIVISIBLE) == (w is None), (name, m, qs, rc)
        if w is not None:
            assert np.array_equal(got, w)
        gx = mulg(x)                                             # a multiple of g always divides
        rc, got = r.call(name, gx)
        wg = want(gx)
        assert (rc == capi.ALCH_NOT_DIVISIBLE) == (wg is None)
        if wg is not None:
            assert np.array_equal(got, x), (name, m, qs)
    for name, want in (("alch_crt", o.crt), ("alch_crtinv", o.crtinv), ("alch_mulg_crt", o.mulg_crt), ("alch_divg_crt", o.divg_crt)):
        rc, got = r.call(name, x)
        if expect_crt:                                           # crtFuncsGT's device tuple
            assert rc == capi.ALCH_OK and np.array_equal(got, want(x)), (name, m, qs)
        else:                                                    # crtFuncsGT answered Nothing before getting here; the ring itself says so too
            assert rc == capi.ALCH_E_NO_CRT
    return r


def check_ext_methods(small, big, has_crt):
    """between2 / between2' / crtExtFuncsGT on the pair: embedPow / embedDec / twacePowDec / coeffs always, the CRT pair when both
    rings have a CRT basis; against the by-definition model."""
    s, b = G.Index(small.m), G.Index(big.m)
    rng = np.random.default_rng(small.m * 7 + big.m)
    x, y = rand_elem(rng, s.n, small.qs), rand_elem(rng, b.n, big.qs)
    pos = G.embed_indices(s, b)
    rc, got = small.ext("alch_embed_pow", big, x, (b.n, big.L))
    want = np.zeros((b.n, big.L), dtype=np.int64)
    want[pos, :] = x
    assert rc == capi.ALCH_OK and np.array_equal(got, want)
    rc, got = small.ext("alch_twace_pow_dec", big, y, (s.n, small.L))
    assert rc == capi.ALCH_OK and np.array_equal(got, y[pos, :])
    rows = np.array(G.coeffs_indices(s, b))
    rc, got = small.ext("alch_coeffs", big, y, (rows.shape[0], s.n, small.L))
    assert rc == capi.ALCH_OK and np.array_equal(got, y[rows, :])
    rc, got = small.ext("alch_embed_dec", big, x, (b.n, big.L))
    assert rc == capi.ALCH_OK
    for j, q in enumerate(small.qs):
        if q:
            assert got[:, j].tolist() == G.embed_dec_def(x[:, j].tolist(), s, b, q), (small.m, big.m, q)
    rc_e, emb = small.ext("alch_embed_crt", big, x, (b.n, big.L))
    rc_t, tw = small.ext("alch_twace_crt", big, y, (s.n, small.L))
    if has_crt:
        slot = capi.ext_table(small.m, big.m, capi.ALCH_EXT_CRT_SLOT)
        assert rc_e == rc_t == capi.ALCH_OK and np.array_equal(emb, x[slot, :])
        for j, q in enumerate(small.qs):
            assert tw[:, j].tolist() == G.twace_crt_def(y[:, j].tolist(), s, b, q)
    else:
        assert rc_e == rc_t == capi.ALCH_E_NO_CRT                 # Lol: crtExtFuncs = Nothing


@pytest.fixture()
def shim():
    s = Shim()
    yield s
    s.close()


def test_arithmetic_example_rings(shim, oracle_lib):
    pt = check_pow_dec_methods(shim, oracle_lib, 4, [7], False, 1)                  # Zq 7 over F4: 7 != 1 mod 4
    for L in (1, 2, 3):
        ct = check_pow_dec_methods(shim, oracle_lib, 512, nested(ARITH_ZQS, L), True, 10 + L)
    check_pow_dec_methods(shim, oracle_lib, 512, [0], False, 2)                     # the key ring, decrypt's lift
    ptbig = check_pow_dec_methods(shim, oracle_lib, 512, [7], False, 3)             # R'_p: decrypt's divG ring
    check_ext_methods(pt, ptbig, False)                                             # encrypt's embed, decrypt's twace
    assert {rc for *_, rc in shim.seen} <= ALLOWED


@pytest.mark.parametrize("k", range(6))
def test_homomrlwr_example_rings(shim, oracle_lib, k):
    pts = {}
    for e in (5, 4, 3, 2, 1):                                                       # Z2E e, examples/Common.hs:32
        pts[e] = check_pow_dec_methods(shim, oracle_lib, H[k], [1 << e], False, 100 * k + e)
    cts = {}
    for L in range(1, 7):
        cts[L] = check_pow_dec_methods(shim, oracle_lib, HP[k], nested(RLWR_ZQS, L), True, 1000 * k + L)
    check_pow_dec_methods(shim, oracle_lib, HP[k], [0], False, 7)
    for e in (5, 1):
        big = check_pow_dec_methods(shim, oracle_lib, HP[k], [1 << e], False, 50 + e)
        check_ext_methods(pts[e], big, False)
    # the ring switch out of this index: E' = gcd(H_k', H_{k+1}') below R' (coeffs) and below S' (embed), CRT rings
    if k < 5:
        ep = math.gcd(HP[k], HP[k + 1])
        qs = nested(RLWR_ZQS, 6)
        e_ring = check_pow_dec_methods(shim, oracle_lib, ep, qs, True, 77 + k)
        check_ext_methods(e_ring, cts[6], True)
        s_ring = check_pow_dec_methods(shim, oracle_lib, HP[k + 1], qs, True, 78 + k)
        check_ext_methods(e_ring, s_ring, True)
    assert {rc for *_, rc in shim.seen} <= ALLOWED
    assert all(rc == capi.ALCH_OK for kind, *_, rc in shim.seen if kind == "nocrt")  # the fallback ring always exists


@pytest.mark.parametrize("k", range(6))
def test_tunnel_example_rings(shim, oracle_lib, k):
    if k <= 3:                                                                      # PT = Cyc H3 (Zq PP8), tunnel3 = H0 -> H3
        pt = check_pow_dec_methods(shim, oracle_lib, H[k], [8], False, 300 + k)
        big = check_pow_dec_methods(shim, oracle_lib, HP[k], [8], False, 310 + k)
        check_ext_methods(pt, big, False)
    for L in range(1, 6):
        check_pow_dec_methods(shim, oracle_lib, HP[k], nested(TUNNEL_ZQS, L), True, 320 + 10 * k + L)
    check_pow_dec_methods(shim, oracle_lib, HP[k], [0], False, 9)
    assert {rc for *_, rc in shim.seen} <= ALLOWED


def test_unsupported_index_is_reported_after_the_crt_question(shim):
    """Status order: NO_CRT is decided before the index is looked at; UNSUPPORTED only for rings that do have a CRT basis --
    GT.hs then keeps the whole (index, element type) on lol-cpp."""
    assert shim.ring_for(False, 4 * 17, [32]) == ("nocrt",)
    q = next(q for q in range((1 << 20) - (1 << 20) % 68 + 1, 1 << 22, 68) if all(q % d for d in range(2, int(q ** .5) + 1)))
    assert shim.ring_for(False, 68, [q]) == ("lolcpp",)
    assert shim.pow_ring(68, [q]) == (None, False)
    assert shim.ring_for(True, 68, [32]) == ("lolcpp",)                             # alch_ring_create_nocrt: the index itself is not served
