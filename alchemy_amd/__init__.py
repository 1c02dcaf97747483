"""alchemy_amd -- MI355X (gfx950) ciphertext-arithmetic backend for ALCHEMY's PT2CT-lowered evaluator.

The product is the C-ABI shared library ``alchemy_amd/lib/libalchemy_hip.so`` (sources in
``alchemy_amd/csrc``, interface in ``include/alchemy_hip.h``).  This package is the thin ctypes binding
used by the tests and bench.py; the host-side mirror of the SymmSHE operations ALCHEMY's evaluator calls is
the C++ header ``alchemy_amd/host/symmshe.hpp`` (the reference's host code is compiled Haskell, so the mirror is
compiled code too).  There is no CPU fallback: importing works anywhere, but every compute
call raises unless the HIP library is built and a gfx950 device is present.
"""
from .capi import AlchemyError, Buf, Hint, Ring, Tunnel, lib_path, load_library  # noqa: F401
