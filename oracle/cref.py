"""ctypes binding of oracle/lol_tensor.c (the C restatement).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblol_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("lol_tensor.c", "lol_tensor_gen.c")]
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
            build()
        l = C.CDLL(_LIB_PATH)
        P64 = C.POINTER(C.c_int64)
        l.orc_ring_new.restype = C.c_void_p
        l.orc_ring_delete.argtypes = [C.c_void_p]
        l.orc_ring_init.argtypes = [C.c_void_p, C.c_int64, C.c_int, P64]
        l.orc_ring_init.restype = C.c_int
        l.orc_ring_psi.argtypes = [C.c_void_p, C.c_int]
        l.orc_ring_psi.restype = C.c_int64
        l.orc_smallest_generator.argtypes = [C.c_int64]
        l.orc_smallest_generator.restype = C.c_int64
        for name in ("orc_crt", "orc_crtinv"):
            getattr(l, name).argtypes = [C.c_void_p, P64]
        for name in ("orc_mul", "orc_add", "orc_sub", "orc_scale"):
            getattr(l, name).argtypes = [C.c_void_p, P64, P64]
        l.orc_decompose_triv.argtypes = [C.c_void_p, P64, C.POINTER(P64)]
        l.orc_decompose_base2.argtypes = [C.c_void_p, P64, C.POINTER(P64)]
        l.orc_baseb_digits.argtypes = [C.c_int64]
        l.orc_baseb_digits.restype = C.c_int
        for name in ("orc_ct_mul_relin_crt", "orc_ct_mul_relin_pow"):
            getattr(l, name).argtypes = [C.c_void_p, C.POINTER(P64), P64, P64, P64, P64, P64, P64, P64]
        l.orc_rescale_drop0.argtypes = [C.c_void_p, P64, P64]
        l.orc_fill_uniform.argtypes = [C.c_void_p, P64, C.c_uint64, C.c_uint64]
        l.orc_bench_mul_relin.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        l.orc_bench_mul_relin.restype = C.c_double
        l.orc_mul_relin_checksum.argtypes = [C.c_void_p] + [C.c_uint64] * 5
        l.orc_mul_relin_checksum.restype = C.c_uint64
        # general-index restatement (lol_tensor_gen.c)
        l.orcg_ring_new.restype = C.c_void_p
        l.orcg_ring_delete.argtypes = [C.c_void_p]
        l.orcg_ring_init.argtypes = [C.c_void_p, C.c_int64, C.c_int, P64]
        l.orcg_ring_init.restype = C.c_int
        l.orcg_n.argtypes = [C.c_void_p]
        l.orcg_n.restype = C.c_int64
        l.orcg_has_crt.argtypes = [C.c_void_p]
        l.orcg_has_crt.restype = C.c_int
        for name in ("orcg_crt", "orcg_crtinv", "orcg_mulg_crt", "orcg_divg_crt", "orcg_divg_pow", "orcg_divg_dec"):
            getattr(l, name).argtypes = [C.c_void_p, P64]
            getattr(l, name).restype = C.c_int
        for name in ("orcg_l", "orcg_linv", "orcg_mulg_pow", "orcg_mulg_dec"):
            getattr(l, name).argtypes = [C.c_void_p, P64]
            getattr(l, name).restype = None
        for name in ("orcg_mul", "orcg_add", "orcg_sub", "orcg_scale"):
            getattr(l, name).argtypes = [C.c_void_p, P64, P64]
            getattr(l, name).restype = None
        l.orcg_decompose_triv.argtypes = [C.c_void_p, P64, C.POINTER(P64)]
        l.orcg_rescale_drop0.argtypes = [C.c_void_p, P64, P64]
        l.orcg_ct_mul_relin_crt.argtypes = [C.c_void_p, C.POINTER(P64), P64, P64, P64, P64, P64, P64, P64]
        l.orcg_ct_mul_relin_crt.restype = C.c_int
        l.orcg_fill_uniform.argtypes = [C.c_void_p, P64, C.c_uint64, C.c_uint64]
        _lib = l
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Ring:
    """Ring context of the C restatement.  Ring elements are numpy int64 arrays of shape (n, L)
    (Lol's tuple-interleaved layout: coefficient-major, limb-minor)."""

    def __init__(self, n: int, qs):
        self.n, self.qs, self.L = int(n), [int(q) for q in qs], len(qs)
        self._h = lib().orc_ring_new()
        q = np.array(self.qs, dtype=np.int64)
        rc = lib().orc_ring_init(self._h, self.n, self.L, _p(q))
        if rc != 0:
            raise ValueError({-1: "bad argument", -2: "modulus not prime", -3: "q != 1 mod 2n"}.get(rc, str(rc)))

    def __del__(self):
        try:
            lib().orc_ring_delete(self._h)
        except Exception:
            pass

    def psi(self, j: int) -> int:
        return int(lib().orc_ring_psi(self._h, j))

    def _unary(self, fn, a):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        assert out.shape == (self.n, self.L)
        fn(self._h, _p(out))
        return out

    def crt(self, a):
        return self._unary(lib().orc_crt, a)

    def crtinv(self, a):
        return self._unary(lib().orc_crtinv, a)

    def _binary(self, fn, a, b):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        bb = np.ascontiguousarray(b, dtype=np.int64)
        fn(self._h, _p(out), _p(bb))
        return out

    def mul(self, a, b):
        return self._binary(lib().orc_mul, a, b)

    def add(self, a, b):
        return self._binary(lib().orc_add, a, b)

    def sub(self, a, b):
        return self._binary(lib().orc_sub, a, b)

    def scale(self, a, s):
        return self._binary(lib().orc_scale, a, np.array(s, dtype=np.int64))

    def decompose_triv(self, c):
        cc = np.ascontiguousarray(c, dtype=np.int64)
        digs = [np.zeros((self.n, self.L), dtype=np.int64) for _ in range(self.L)]
        arr = (C.POINTER(C.c_int64) * self.L)(*[_p(d) for d in digs])
        lib().orc_decompose_triv(self._h, _p(cc), arr)
        return digs

    def decompose_base2(self, c):
        cc = np.ascontiguousarray(c, dtype=np.int64)
        nd = sum(lib().orc_baseb_digits(q) for q in self.qs)
        digs = [np.zeros((self.n, self.L), dtype=np.int64) for _ in range(nd)]
        arr = (C.POINTER(C.c_int64) * nd)(*[_p(d) for d in digs])
        lib().orc_decompose_base2(self._h, _p(cc), arr)
        return digs

    def ct_mul_relin(self, hint, a0, a1, b0, b1, s_pre=None, pow_basis=False):
        """hint: list of 2*L CRT-basis elements [h0_0, h1_0, h0_1, ...].  Returns (out0, out1)."""
        hs = [np.ascontiguousarray(h, dtype=np.int64) for h in hint]
        arr = (C.POINTER(C.c_int64) * len(hs))(*[_p(h) for h in hs])
        ins = [np.ascontiguousarray(x, dtype=np.int64) for x in (a0, a1, b0, b1)]
        s = np.array(s_pre if s_pre is not None else [1] * self.L, dtype=np.int64)
        o0 = np.zeros((self.n, self.L), dtype=np.int64)
        o1 = np.zeros((self.n, self.L), dtype=np.int64)
        fn = lib().orc_ct_mul_relin_pow if pow_basis else lib().orc_ct_mul_relin_crt
        fn(self._h, arr, _p(ins[0]), _p(ins[1]), _p(ins[2]), _p(ins[3]), _p(s), _p(o0), _p(o1))
        return o0, o1

    def rescale_drop0(self, x):
        xx = np.ascontiguousarray(x, dtype=np.int64)
        out = np.zeros((self.n, self.L - 1), dtype=np.int64)
        lib().orc_rescale_drop0(self._h, _p(xx), _p(out))
        return out

    def fill_uniform(self, seed: int, elem: int):
        out = np.zeros((self.n, self.L), dtype=np.int64)
        lib().orc_fill_uniform(self._h, _p(out), C.c_uint64(seed), C.c_uint64(elem))
        return out

    def mul_relin_checksum(self, seed_a: int, seed_b: int, seed_h: int, first: int, count: int) -> int:
        return int(lib().orc_mul_relin_checksum(self._h, seed_a, seed_b, seed_h, first, count))

    def bench_mul_relin(self, ops: int, seed: int = 2026) -> float:
        return float(lib().orc_bench_mul_relin(self._h, int(ops), C.c_uint64(seed)))


class GenRing:
    """Ring context of the general-index C restatement (oracle/lol_tensor_gen.c): any cyclotomic index m; a modulus
    0 stands for the integers (Pow / Dec operations only), a modulus that is not 1 mod m gives a ring without CRT."""

    def __init__(self, m: int, qs):
        self.m, self.qs, self.L = int(m), [int(q) for q in qs], len(qs)
        self._h = lib().orcg_ring_new()
        q = np.array(self.qs, dtype=np.int64)
        rc = lib().orcg_ring_init(self._h, self.m, self.L, _p(q))
        if rc != 0:
            raise ValueError({-1: "bad argument", -2: "modulus not prime"}.get(rc, str(rc)))
        self.n = int(lib().orcg_n(self._h))
        self.has_crt = bool(lib().orcg_has_crt(self._h))

    def __del__(self):
        try:
            lib().orcg_ring_delete(self._h)
        except Exception:
            pass

    def _unary(self, fn, a, maybe=False):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        assert out.shape == (self.n, self.L), (out.shape, self.n, self.L)
        rc = fn(self._h, _p(out))
        if maybe:
            return out if rc == 1 else None             # Lol's Maybe
        if rc not in (None, 0):
            raise ValueError("ring has no CRT basis")
        return out

    def crt(self, a): return self._unary(lib().orcg_crt, a)
    def crtinv(self, a): return self._unary(lib().orcg_crtinv, a)
    def l(self, a): return self._unary(lib().orcg_l, a)
    def linv(self, a): return self._unary(lib().orcg_linv, a)
    def mulg_pow(self, a): return self._unary(lib().orcg_mulg_pow, a)
    def mulg_dec(self, a): return self._unary(lib().orcg_mulg_dec, a)
    def mulg_crt(self, a): return self._unary(lib().orcg_mulg_crt, a)
    def divg_crt(self, a): return self._unary(lib().orcg_divg_crt, a)
    def divg_pow(self, a): return self._unary(lib().orcg_divg_pow, a, maybe=True)
    def divg_dec(self, a): return self._unary(lib().orcg_divg_dec, a, maybe=True)

    def _binary(self, fn, a, b):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        fn(self._h, _p(out), _p(np.ascontiguousarray(b, dtype=np.int64)))
        return out

    def mul(self, a, b): return self._binary(lib().orcg_mul, a, b)
    def add(self, a, b): return self._binary(lib().orcg_add, a, b)
    def sub(self, a, b): return self._binary(lib().orcg_sub, a, b)
    def scale(self, a, s): return self._binary(lib().orcg_scale, a, np.array(s, dtype=np.int64))

    def decompose_triv(self, c):
        cc = np.ascontiguousarray(c, dtype=np.int64)
        digs = [np.zeros((self.n, self.L), dtype=np.int64) for _ in range(self.L)]
        arr = (C.POINTER(C.c_int64) * self.L)(*[_p(d) for d in digs])
        lib().orcg_decompose_triv(self._h, _p(cc), arr)
        return digs

    def rescale_drop0(self, x):
        xx = np.ascontiguousarray(x, dtype=np.int64)
        out = np.zeros((self.n, self.L - 1), dtype=np.int64)
        lib().orcg_rescale_drop0(self._h, _p(xx), _p(out))
        return out

    def ct_mul_relin(self, hint, a0, a1, b0, b1, s_pre=None):
        hs = [np.ascontiguousarray(h, dtype=np.int64) for h in hint]
        arr = (C.POINTER(C.c_int64) * len(hs))(*[_p(h) for h in hs])
        ins = [np.ascontiguousarray(x, dtype=np.int64) for x in (a0, a1, b0, b1)]
        s = np.array(s_pre if s_pre is not None else [1] * self.L, dtype=np.int64)
        o0 = np.zeros((self.n, self.L), dtype=np.int64)
        o1 = np.zeros((self.n, self.L), dtype=np.int64)
        rc = lib().orcg_ct_mul_relin_crt(self._h, arr, _p(ins[0]), _p(ins[1]), _p(ins[2]), _p(ins[3]), _p(s), _p(o0), _p(o1))
        assert rc == 0
        return o0, o1

    def fill_uniform(self, seed: int, elem: int):
        out = np.zeros((self.n, self.L), dtype=np.int64)
        lib().orcg_fill_uniform(self._h, _p(out), C.c_uint64(seed), C.c_uint64(elem))
        return out


def smallest_generator(q: int) -> int:
    return int(lib().orc_smallest_generator(int(q)))
