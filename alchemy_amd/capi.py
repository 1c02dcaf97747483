"""ctypes binding of include/alchemy_hip.h.  One Python method per C entry point; numpy int64 arrays of
shape (count, n, L) stand for host buffers in Lol's tuple-interleaved layout."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

ALCH_OK = 0
ALCH_E_INVALID, ALCH_E_NOT_PRIME, ALCH_E_NO_CRT, ALCH_E_UNSUPPORTED = -1, -2, -3, -4
ALCH_E_NO_DEVICE, ALCH_E_HIP, ALCH_E_NOMEM, ALCH_E_INTERNAL = -5, -6, -7, -8
ALCH_POW_IN, ALCH_POW_OUT = 1, 2
ALCH_GAD_TRIV, ALCH_GAD_BASE2 = 0, 1
ALCH_NOT_DIVISIBLE = 1
ALCH_BASIS_POW, ALCH_BASIS_DEC, ALCH_BASIS_CRT = 0, 1, 2
# ops of alch_buf_tensor_op (the unary Tensor methods on device-resident elements)
(ALCH_T_CRT, ALCH_T_CRTINV, ALCH_T_L, ALCH_T_LINV, ALCH_T_MULG_POW, ALCH_T_MULG_DEC, ALCH_T_MULG_CRT, ALCH_T_DIVG_POW,
 ALCH_T_DIVG_DEC, ALCH_T_DIVG_CRT) = range(10)

# every symbol include/alchemy_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "alch_last_error", "alch_version", "alch_ring_create", "alch_ring_destroy", "alch_host_root", "alch_ring_n",
    "alch_ring_set_stream", "alch_sync", "alch_timer_start", "alch_timer_stop", "alch_crt", "alch_crtinv",
    "alch_mul", "alch_add", "alch_sub", "alch_scale", "alch_mulg_pow", "alch_mulg_dec", "alch_mulg_crt",
    "alch_divg_pow", "alch_divg_dec", "alch_divg_crt", "alch_decompose_triv", "alch_buf_alloc", "alch_buf_free",
    "alch_buf_elems", "alch_buf_upload", "alch_buf_download", "alch_buf_fill_uniform", "alch_buf_crt",
    "alch_buf_crtinv", "alch_buf_mul", "alch_buf_add", "alch_buf_checksum", "alch_buf_checksum_at", "alch_hint_load", "alch_hint_from_buf",
    "alch_hint_free", "alch_ct_mul_relin", "alch_buf_rescale_drop0", "alch_buf_sub", "alch_buf_scale",
    "alch_buf_decompose_triv", "alch_buf_rescale_add0", "alch_decompose_base2", "alch_ct_mul_full", "alch_buf_device_ptr",
    "alch_ring_set_option", "alch_ring_create_nocrt", "alch_l", "alch_linv", "alch_buf_l", "alch_buf_linv", "alch_buf_mulg",
    "alch_buf_divg", "alch_buf_mul_public", "alch_buf_add_public", "alch_select_limbs", "alch_modulus_units",
    "alch_tunnel_info", "alch_tunnel_create", "alch_tunnel_free", "alch_ct_tunnel", "alch_ct_mod_switch",
    "alch_buf_embed", "alch_buf_twace", "alch_buf_coeffs", "alch_embed_pow", "alch_embed_dec", "alch_embed_crt",
    "alch_twace_pow_dec", "alch_twace_crt", "alch_coeffs", "alch_ext_table", "alch_crt_set_dec", "alch_ct_add_public",
    "alch_ring_share_stream", "alch_buf_copy", "alch_buf_tensor_op", "alch_buf_view", "alch_buf_ring", "alch_ring_device",
]


class AlchemyError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"alchemy_hip error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # ALCH_LIB_PATH selects an experimental build of the same library (tools/build_variant.sh)
    return os.environ.get("ALCH_LIB_PATH") or os.path.join(_HERE, "lib", "libalchemy_hip.so")


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process fight over the device
    (whichever initialises second reports "no HIP device"), so when torch is installed but not imported yet its
    runtime is loaded first: libalchemy_hip.so's libamdhip64.so.7 dependency then resolves to that same object,
    and a later `import torch` reuses it.  Programs without torch (the C++ / Haskell hosts) use ROCm's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise AlchemyError(ALCH_E_NO_DEVICE, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    _share_hip_runtime_with_torch()
    l = C.CDLL(path)
    P64, PU64, VP = C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.c_void_p
    l.alch_last_error.restype = C.c_char_p
    l.alch_version.restype = C.c_uint32
    sig = {
        "alch_ring_create": [C.c_uint32, C.c_int, PU64, C.POINTER(VP)],
        "alch_ring_create_nocrt": [C.c_uint32, C.c_int, PU64, C.POINTER(VP)],
        "alch_ring_destroy": [VP],
        "alch_select_limbs": [PU64, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4,
        "alch_modulus_units": [C.c_uint64],
        "alch_tunnel_info": [VP, VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
        "alch_tunnel_create": [VP, VP, C.c_int, VP, VP, C.POINTER(VP)],
        "alch_tunnel_free": [VP],
        "alch_ct_tunnel": [VP, VP, VP, C.c_size_t, PU64, C.c_uint],
        "alch_ct_mod_switch": [VP, VP, C.c_size_t, C.c_uint],
        "alch_l": [VP, P64], "alch_linv": [VP, P64],
        "alch_buf_l": [VP, C.c_size_t, C.c_size_t], "alch_buf_linv": [VP, C.c_size_t, C.c_size_t],
        "alch_buf_mulg": [VP, C.c_size_t, C.c_size_t, C.c_int], "alch_buf_divg": [VP, C.c_size_t, C.c_size_t, C.c_int],
        "alch_buf_mul_public": [VP, VP, VP, C.c_size_t, C.c_size_t],
        "alch_buf_add_public": [VP, VP, C.c_size_t, C.c_size_t],
        "alch_host_root": [C.c_uint32, C.c_uint64, PU64, PU64],
        "alch_ring_n": [VP, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "alch_ring_set_stream": [VP, VP],
        "alch_ring_share_stream": [VP, VP],
        "alch_buf_ring": [VP, C.POINTER(VP)],
        "alch_ring_device": [VP, C.POINTER(C.c_int), C.POINTER(VP)],
        "alch_buf_copy": [VP, C.c_size_t, VP, C.c_size_t, C.c_size_t],
        "alch_buf_tensor_op": [VP, C.c_size_t, VP, C.c_size_t, C.c_size_t, C.c_int],
        "alch_buf_view": [VP, C.c_size_t, C.c_size_t, C.POINTER(VP)],
        "alch_sync": [VP],
        "alch_ring_set_option": [VP, C.c_char_p, C.c_long],
        "alch_timer_start": [VP],
        "alch_timer_stop": [VP, C.POINTER(C.c_float)],
        "alch_crt": [VP, P64], "alch_crtinv": [VP, P64],
        "alch_mul": [VP, P64, P64], "alch_add": [VP, P64, P64], "alch_sub": [VP, P64, P64],
        "alch_scale": [VP, P64, PU64],
        "alch_mulg_pow": [VP, P64], "alch_mulg_dec": [VP, P64], "alch_mulg_crt": [VP, P64],
        "alch_divg_pow": [VP, P64], "alch_divg_dec": [VP, P64], "alch_divg_crt": [VP, P64],
        "alch_decompose_triv": [VP, P64, P64],
        "alch_decompose_base2": [VP, P64, P64, C.POINTER(C.c_int)],
        "alch_buf_alloc": [VP, C.c_size_t, C.POINTER(VP)],
        "alch_buf_free": [VP],
        "alch_buf_elems": [VP, C.POINTER(C.c_size_t)],
        "alch_buf_upload": [VP, C.c_size_t, C.c_size_t, P64],
        "alch_buf_download": [VP, C.c_size_t, C.c_size_t, P64],
        "alch_buf_fill_uniform": [VP, C.c_uint64],
        "alch_buf_crt": [VP, C.c_size_t, C.c_size_t],
        "alch_buf_crtinv": [VP, C.c_size_t, C.c_size_t],
        "alch_buf_mul": [VP, VP, VP, C.c_size_t],
        "alch_buf_add": [VP, VP, VP, C.c_size_t],
        "alch_buf_checksum": [VP, C.c_size_t, C.c_size_t, PU64],
        "alch_buf_checksum_at": [VP, C.c_size_t, C.c_size_t, C.c_uint64, PU64],
        "alch_hint_load": [VP, C.c_int, P64, C.POINTER(VP)],
        "alch_hint_from_buf": [VP, C.c_int, VP, C.POINTER(VP)],
        "alch_hint_free": [VP],
        "alch_ct_mul_relin": [VP, VP, VP, VP, VP, C.c_size_t, PU64, C.c_uint],
        "alch_ct_mul_full": [VP, VP, VP, VP, C.c_size_t, PU64, C.c_uint],
        "alch_buf_device_ptr": [VP, C.POINTER(VP), C.POINTER(C.c_size_t)],
        "alch_buf_rescale_drop0": [VP, VP, C.c_size_t],
        "alch_buf_rescale_add0": [VP, VP, C.c_size_t],
        "alch_buf_sub": [VP, VP, VP, C.c_size_t],
        "alch_buf_scale": [VP, VP, C.c_size_t, PU64],
        "alch_buf_decompose_triv": [VP, C.c_size_t, VP, C.c_size_t],
        "alch_buf_embed": [VP, VP, C.c_size_t, C.c_int], "alch_buf_twace": [VP, VP, C.c_size_t, C.c_int],
        "alch_buf_coeffs": [VP, VP, C.c_size_t],
        "alch_ct_add_public": [VP, VP, C.c_size_t, PU64, VP, C.c_size_t],
        "alch_embed_pow": [VP, VP, P64, P64], "alch_embed_dec": [VP, VP, P64, P64], "alch_embed_crt": [VP, VP, P64, P64],
        "alch_twace_pow_dec": [VP, VP, P64, P64], "alch_twace_crt": [VP, VP, P64, P64], "alch_coeffs": [VP, VP, P64, P64],
        "alch_ext_table": [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)],
        "alch_crt_set_dec": [C.c_uint32, C.c_uint32, C.c_uint32, P64, C.POINTER(C.c_size_t)],
    }
    for name, args in sig.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = l
    return l


def _check(rc: int):
    if rc < 0:
        raise AlchemyError(rc, load_library().alch_last_error().decode())
    return rc


def _p64(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pu64(vals):
    arr = (C.c_uint64 * len(vals))(*[int(v) for v in vals])
    return arr


ALCH_OP_MUL, ALCH_OP_TUNNEL = 0, 1


def select_limbs(moduli, p_noise_out: int, op: int = ALCH_OP_MUL, gadget: int = ALCH_GAD_TRIV):
    """PT2CT's limb-count selection (host only): (L_in, L_hint, L_out, p_noise_in) of one mul_ / tunnel whose output has
    pNoise p_noise_out, for the circuit's modulus list in `Zqs` order."""
    v = [C.c_int() for _ in range(4)]
    _check(load_library().alch_select_limbs(_pu64(moduli), len(moduli), op, gadget, p_noise_out, *[C.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)


def host_root(m: int, q: int):
    """(psi, generator) of the root rule; host-only, needs no GPU."""
    psi, g = C.c_uint64(), C.c_uint64()
    _check(load_library().alch_host_root(m, q, C.byref(psi), C.byref(g)))
    return int(psi.value), int(g.value)


ALCH_EXT_POW_POS, ALCH_EXT_COEFFS, ALCH_EXT_CRT_SLOT = 0, 1, 2


def ext_table(m_small: int, m_big: int, which: int) -> np.ndarray:
    """Host-only index table of the Tensor methods between two indices (alch_ext_table); needs no GPU."""
    n = C.c_size_t(0)
    _check(load_library().alch_ext_table(m_small, m_big, which, None, C.byref(n)))
    out = np.zeros(n.value, dtype=np.int32)
    _check(load_library().alch_ext_table(m_small, m_big, which, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
    return out


def crt_set_dec(m_small: int, m_big: int, p: int, n_big: int) -> np.ndarray:
    """Tensor crtSetDec (host-only): array (count, n_big) of residues mod p on the decoding basis of index m_big."""
    c = C.c_size_t(0)
    _check(load_library().alch_crt_set_dec(m_small, m_big, p, None, C.byref(c)))
    out = np.zeros((c.value, n_big), dtype=np.int64)
    _check(load_library().alch_crt_set_dec(m_small, m_big, p, _p64(out), C.byref(c)))
    return out


class Ring:
    """alch_ring: one (cyclotomic index, RNS modulus list) context, i.e. one `Cyc t m' zq` type."""

    def __init__(self, m: int, qs, nocrt: bool = False):
        """nocrt: a ring without CRT basis (plaintext ring Z_p, or the integers with modulus 0): Pow / Dec ops only."""
        self._l = load_library()
        self.m, self.qs, self.L = int(m), [int(q) for q in qs], len(qs)
        h = C.c_void_p()
        create = self._l.alch_ring_create_nocrt if nocrt else self._l.alch_ring_create
        _check(create(self.m, self.L, _pu64(self.qs), C.byref(h)))
        self._h = h
        w, n = C.c_int(), C.c_uint32()
        _check(self._l.alch_ring_n(self._h, C.byref(n), None, C.byref(w)))
        self.word_bytes, self.n = int(w.value), int(n.value)          # n = phi(m)

    def close(self):
        if getattr(self, "_h", None):
            self._l.alch_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- stream / timing
    def set_stream(self, hip_stream: int):
        _check(self._l.alch_ring_set_stream(self._h, C.c_void_p(hip_stream)))

    def share_stream(self, other: "Ring"):
        """Queue this ring's work on `other`'s stream from now on (operations between the two rings then need no events)."""
        _check(self._l.alch_ring_share_stream(self._h, other._h))

    def sync(self):
        _check(self._l.alch_sync(self._h))

    def set_option(self, name: str, value: int):
        """Launch-structure option of the fused kernels (chunk, one_stream, ks_grid, ti_grid, ti_split, rs_slots)."""
        _check(self._l.alch_ring_set_option(self._h, name.encode(), int(value)))

    def timer_start(self):
        _check(self._l.alch_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _check(self._l.alch_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    # --- Tensor methods on host buffers (shape (n, L) int64, returns a new array)
    def _host1(self, fn, a):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        assert out.shape == (self.n, self.L), out.shape
        _check(fn(self._h, _p64(out)))
        return out

    def _host2(self, fn, a, b):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        bb = np.ascontiguousarray(b, dtype=np.int64)
        assert out.shape == bb.shape == (self.n, self.L)
        _check(fn(self._h, _p64(out), _p64(bb)))
        return out

    def crt(self, a): return self._host1(self._l.alch_crt, a)
    def crtinv(self, a): return self._host1(self._l.alch_crtinv, a)
    def mul(self, a, b): return self._host2(self._l.alch_mul, a, b)
    def add(self, a, b): return self._host2(self._l.alch_add, a, b)
    def sub(self, a, b): return self._host2(self._l.alch_sub, a, b)
    def mulg_pow(self, a): return self._host1(self._l.alch_mulg_pow, a)
    def mulg_dec(self, a): return self._host1(self._l.alch_mulg_dec, a)
    def mulg_crt(self, a): return self._host1(self._l.alch_mulg_crt, a)
    def l(self, a): return self._host1(self._l.alch_l, a)
    def linv(self, a): return self._host1(self._l.alch_linv, a)

    def _maybe(self, fn, a):
        """divG family: the new array, or None for Lol's Nothing (ALCH_NOT_DIVISIBLE)."""
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        assert out.shape == (self.n, self.L), out.shape
        return None if _check(fn(self._h, _p64(out))) == ALCH_NOT_DIVISIBLE else out

    def divg_pow(self, a): return self._maybe(self._l.alch_divg_pow, a)
    def divg_dec(self, a): return self._maybe(self._l.alch_divg_dec, a)
    def divg_crt(self, a): return self._maybe(self._l.alch_divg_crt, a)

    def scale(self, a, s):
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        _check(self._l.alch_scale(self._h, _p64(out), _pu64(s)))
        return out

    def decompose_triv(self, c_pow):
        c = np.ascontiguousarray(c_pow, dtype=np.int64)
        out = np.zeros((self.L, self.n, self.L), dtype=np.int64)
        _check(self._l.alch_decompose_triv(self._h, _p64(c), _p64(out)))
        return [out[i] for i in range(self.L)]

    def decompose_base2(self, c_pow):
        c = np.ascontiguousarray(c_pow, dtype=np.int64)
        nd = C.c_int()
        _check(self._l.alch_decompose_base2(self._h, None, None, C.byref(nd)))
        out = np.zeros((nd.value, self.n, self.L), dtype=np.int64)
        _check(self._l.alch_decompose_base2(self._h, _p64(c), _p64(out), C.byref(nd)))
        return [out[i] for i in range(nd.value)]

    # --- Tensor methods towards a ring of a multiple index (self = the small ring), host buffers
    def _ext(self, fn, big: "Ring", a, out_shape):
        a = np.ascontiguousarray(a, dtype=np.int64)
        out = np.zeros(out_shape, dtype=np.int64)
        _check(fn(self._h, big._h, _p64(a), _p64(out)))
        return out

    def embed_pow(self, big, a): return self._ext(self._l.alch_embed_pow, big, a, (big.n, big.L))
    def embed_dec(self, big, a): return self._ext(self._l.alch_embed_dec, big, a, (big.n, big.L))
    def embed_crt(self, big, a): return self._ext(self._l.alch_embed_crt, big, a, (big.n, big.L))
    def twace_pow_dec(self, big, a): return self._ext(self._l.alch_twace_pow_dec, big, a, (self.n, self.L))
    def twace_crt(self, big, a): return self._ext(self._l.alch_twace_crt, big, a, (self.n, self.L))
    def coeffs(self, big, a): return self._ext(self._l.alch_coeffs, big, a, (big.n // self.n, self.n, self.L))

    # --- device-resident
    def alloc(self, n_elems: int) -> "Buf":
        return Buf(self, n_elems)

    def upload(self, host) -> "Buf":
        host = np.ascontiguousarray(host, dtype=np.int64)
        assert host.ndim == 3 and host.shape[1:] == (self.n, self.L)
        b = Buf(self, host.shape[0])
        b.upload(host)
        return b

    def hint_load(self, host_crt, gadget: int = ALCH_GAD_TRIV) -> "Hint":
        host = np.ascontiguousarray(host_crt, dtype=np.int64)
        assert host.shape == (2 * self.gadget_digits(gadget), self.n, self.L)
        h = C.c_void_p()
        _check(self._l.alch_hint_load(self._h, gadget, _p64(host), C.byref(h)))
        return Hint(self, h)

    def gadget_digits(self, gadget: int = ALCH_GAD_TRIV) -> int:
        """Digits (= hint rows) of a gadget on this ring: L for TrivGad, sum_i ceil(log2 q_i) for BaseBGad 2."""
        if gadget == ALCH_GAD_TRIV:
            return self.L
        nd = C.c_int()
        _check(self._l.alch_decompose_base2(self._h, None, None, C.byref(nd)))
        return nd.value

    def hint_from_buf(self, buf: "Buf", gadget: int = ALCH_GAD_TRIV) -> "Hint":
        h = C.c_void_p()
        _check(self._l.alch_hint_from_buf(self._h, gadget, buf._h, C.byref(h)))
        return Hint(self, h)

    def ct_mul_relin(self, hint: "Hint", a: "Buf", b: "Buf", out: "Buf", batch: int, s_pre=None, flags: int = 0):
        sp = _pu64(s_pre) if s_pre is not None else None
        _check(self._l.alch_ct_mul_relin(self._h, hint._h, a._h, b._h, out._h, batch, sp, flags))


def ct_mod_switch(src: "Buf", dst: "Buf", batch: int, flags: int = 0):
    """SymmSHE modSwitch of a batch of linear MSD ciphertexts between two rings whose moduli nest (up or down)."""
    _check(load_library().alch_ct_mod_switch(src._h, dst._h, batch, flags))


def ct_mul_full(hint: "Hint", a: "Buf", b: "Buf", out: "Buf", batch: int, s_pre=None, flags: int = 0):
    """PT2CT's whole mul_: modSwitch . keySwitchQuadCirc hint . modSwitch $ a * b (rings come from the handles)."""
    sp = _pu64(s_pre) if s_pre is not None else None
    _check(load_library().alch_ct_mul_full(hint._h, a._h, b._h, out._h, batch, sp, flags))


class Tunnel:
    """alch_tunnel: SymmSHE `tunnel hint` from ring_r = R'_q to ring_s = S'_q (device-resident linear function + hints)."""

    def __init__(self, ring_r: "Ring", ring_s: "Ring", lin_crt: "Buf", ks_crt: "Buf", gadget: int = ALCH_GAD_TRIV):
        self.ring_r, self.ring_s = ring_r, ring_s
        h = C.c_void_p()
        _check(ring_s._l.alch_tunnel_create(ring_r._h, ring_s._h, gadget, lin_crt._h, ks_crt._h, C.byref(h)))
        self._h = h

    @staticmethod
    def info(ring_r: "Ring", ring_s: "Ring"):
        """(index of E' = R' cap S', d_rel = dim R'/E')."""
        e, d = C.c_uint32(), C.c_uint32()
        _check(ring_r._l.alch_tunnel_info(ring_r._h, ring_s._h, C.byref(e), C.byref(d)))
        return int(e.value), int(d.value)

    def apply(self, src: "Buf", dst: "Buf", batch: int, s_pre=None, flags: int = 0):
        sp = _pu64(s_pre) if s_pre is not None else None
        _check(self.ring_s._l.alch_ct_tunnel(self._h, src._h, dst._h, batch, sp, flags))

    def free(self):
        if getattr(self, "_h", None):
            self.ring_s._l.alch_tunnel_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Buf:
    """alch_buf: device-resident array of ring elements (limb-major, 32- or 64-bit words)."""

    def __init__(self, ring: Ring, n_elems: int):
        self.ring, self.n_elems = ring, int(n_elems)
        h = C.c_void_p()
        _check(ring._l.alch_buf_alloc(ring._h, self.n_elems, C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None):
            self.ring._l.alch_buf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, host, first: int = 0):
        host = np.ascontiguousarray(host, dtype=np.int64)
        _check(self.ring._l.alch_buf_upload(self._h, first, host.shape[0], _p64(host)))

    def download(self, first: int = 0, count: int | None = None):
        count = self.n_elems - first if count is None else count
        out = np.zeros((count, self.ring.n, self.ring.L), dtype=np.int64)
        _check(self.ring._l.alch_buf_download(self._h, first, count, _p64(out)))
        return out

    def fill_uniform(self, seed: int):
        _check(self.ring._l.alch_buf_fill_uniform(self._h, C.c_uint64(seed)))

    def copy_from(self, src: "Buf", count: int, dst_first: int = 0, src_first: int = 0):
        _check(self.ring._l.alch_buf_copy(self._h, dst_first, src._h, src_first, count))

    def tensor_op(self, src: "Buf", op: int, count: int = 1, dst_first: int = 0, src_first: int = 0) -> bool:
        """self[dst_first + i] = op(src[src_first + i]) (ALCH_T_*); False = Lol's Nothing (divG family)."""
        return _check(self.ring._l.alch_buf_tensor_op(self._h, dst_first, src._h, src_first, count, op)) == ALCH_OK

    def view(self, first: int, count: int = 1) -> "Buf":
        """Non-owning alias of elements [first, first + count); keeps this buffer alive."""
        h = C.c_void_p()
        _check(self.ring._l.alch_buf_view(self._h, first, count, C.byref(h)))
        v = Buf.__new__(Buf)
        v.ring, v.n_elems, v._h, v._parent = self.ring, int(count), h, self
        return v

    def device_ptr(self):
        """(address, bytes) of the device allocation."""
        p, nb = C.c_void_p(), C.c_size_t()
        _check(self.ring._l.alch_buf_device_ptr(self._h, C.byref(p), C.byref(nb)))
        return int(p.value), int(nb.value)

    def as_torch(self, first: int = 0, count: int | None = None):
        """Zero-copy torch view (1-D, int32 or int64 words, limb-major) of elements [first, first+count) -- for
        torch.distributed collectives on result batches.  The caller orders streams (Ring.sync())."""
        import torch
        count = self.n_elems - first if count is None else count
        addr, _ = self.device_ptr()
        wb = self.ring.word_bytes
        words = count * self.ring.n * self.ring.L

        class _View:                     # minimal __cuda_array_interface__ carrier
            pass

        v = _View()
        v.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i4" if wb == 4 else "<i8",
                                      "data": (addr + first * self.ring.n * self.ring.L * wb, False), "version": 2}
        v._keepalive = self
        return torch.as_tensor(v, device="cuda")

    def crt(self, first: int = 0, count: int | None = None):
        _check(self.ring._l.alch_buf_crt(self._h, first, self.n_elems - first if count is None else count))

    def crtinv(self, first: int = 0, count: int | None = None):
        _check(self.ring._l.alch_buf_crtinv(self._h, first, self.n_elems - first if count is None else count))

    def mul(self, a: "Buf", b: "Buf", count: int):
        _check(self.ring._l.alch_buf_mul(self._h, a._h, b._h, count))

    def add(self, a: "Buf", b: "Buf", count: int):
        _check(self.ring._l.alch_buf_add(self._h, a._h, b._h, count))

    def l(self, first: int = 0, count: int | None = None):
        _check(self.ring._l.alch_buf_l(self._h, first, self.n_elems - first if count is None else count))

    def linv(self, first: int = 0, count: int | None = None):
        _check(self.ring._l.alch_buf_linv(self._h, first, self.n_elems - first if count is None else count))

    def mulg(self, basis: int, first: int = 0, count: int | None = None):
        _check(self.ring._l.alch_buf_mulg(self._h, first, self.n_elems - first if count is None else count, basis))

    def divg(self, basis: int, first: int = 0, count: int | None = None) -> bool:
        """True = divided; False = Lol's Nothing for some element of the range."""
        return _check(self.ring._l.alch_buf_divg(self._h, first, self.n_elems - first if count is None else count, basis)) == ALCH_OK

    def mul_public(self, src: "Buf", pub: "Buf", pub_index: int, count: int):
        _check(self.ring._l.alch_buf_mul_public(self._h, src._h, pub._h, pub_index, count))

    def add_public(self, pub: "Buf", pub_index: int, batch: int):
        _check(self.ring._l.alch_buf_add_public(self._h, pub._h, pub_index, batch))

    def ct_add_public(self, src: "Buf", batch: int, s, pub: "Buf", pub_index: int = 0):
        """self = s * src with pub added to every c0: SymmSHE addPublic including its toLSD scalar, one pass."""
        _check(self.ring._l.alch_ct_add_public(self._h, src._h, batch, _pu64(s) if s is not None else None, pub._h, pub_index))

    def embed_from(self, src_small: "Buf", count: int, basis: int):
        _check(self.ring._l.alch_buf_embed(self._h, src_small._h, count, basis))

    def twace_from(self, src_big: "Buf", count: int, basis: int):
        _check(self.ring._l.alch_buf_twace(self._h, src_big._h, count, basis))

    def coeffs_from(self, src_big: "Buf", count: int):
        _check(self.ring._l.alch_buf_coeffs(self._h, src_big._h, count))

    def checksum(self, first: int = 0, count: int | None = None, position: int = 0) -> int:
        """position: element index of `first` in the whole batch this buffer is a part of (alch_buf_checksum_at)."""
        s = C.c_uint64()
        _check(self.ring._l.alch_buf_checksum_at(self._h, first, self.n_elems - first if count is None else count, position, C.byref(s)))
        return int(s.value)

    def rescale_drop0_into(self, dst: "Buf", count: int):
        _check(self.ring._l.alch_buf_rescale_drop0(self._h, dst._h, count))

    def rescale_add0_into(self, dst: "Buf", count: int):
        _check(self.ring._l.alch_buf_rescale_add0(self._h, dst._h, count))

    def sub(self, a: "Buf", b: "Buf", count: int):
        _check(self.ring._l.alch_buf_sub(self._h, a._h, b._h, count))

    def scale(self, src: "Buf", count: int, s):
        _check(self.ring._l.alch_buf_scale(self._h, src._h, count, _pu64(s)))

    def decompose_triv_into(self, src_index: int, dst: "Buf", dst_first: int = 0):
        _check(self.ring._l.alch_buf_decompose_triv(self._h, src_index, dst._h, dst_first))


class Hint:
    """alch_hint: device-resident KSQuadCircHint (Montgomery form)."""

    def __init__(self, ring: Ring, handle):
        self.ring, self._h = ring, handle

    def free(self):
        if getattr(self, "_h", None):
            self.ring._l.alch_hint_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
