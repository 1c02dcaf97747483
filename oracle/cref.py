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
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "lol_tensor.c")):
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

    def bench_mul_relin(self, ops: int, seed: int = 2026) -> float:
        return float(lib().orc_bench_mul_relin(self._h, int(ops), C.c_uint64(seed)))


def smallest_generator(q: int) -> int:
    return int(lib().orc_smallest_generator(int(q)))
