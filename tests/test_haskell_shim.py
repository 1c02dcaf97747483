"""CPU: the Haskell FFI module shipped as source (haskell/.../GT/Backend.hs) checked mechanically
against include/alchemy_hip.h -- every entry point of the header is imported exactly once, with the header's
arity and C types (no Haskell toolchain exists here, so this is the only check that file gets)."""
import os
import re

from conftest import ROOT

CTYPE = {  # C parameter type (qualifiers stripped) -> Haskell FFI type
    "uint32_t": "Word32", "uint64_t": "Word64", "int": "CInt", "size_t": "CSize", "unsigned": "CUInt", "long": "CLong",
    "char*": "CString",
    "uint64_t*": "Ptr Word64", "uint32_t*": "Ptr Word32", "int*": "Ptr CInt", "size_t*": "Ptr CSize",
    "float*": "Ptr CFloat", "int64_t*": "Ptr Int64", "void*": "Ptr ()", "void**": "Ptr (Ptr ())",
    "alch_ring*": "Ptr AlchRing", "alch_buf*": "Ptr AlchBuf", "alch_hint*": "Ptr AlchHint",
    "alch_ring**": "Ptr (Ptr AlchRing)", "alch_buf**": "Ptr (Ptr AlchBuf)", "alch_hint**": "Ptr (Ptr AlchHint)",
}
RET = {"int": "IO CInt", "uint32_t": "IO Word32", "const char*": "IO CString"}


def header_prototypes():
    text = open(os.path.join(ROOT, "include", "alchemy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for ret, star, name, args in re.findall(r"^((?:const\s+)?\w+)\s*(\*?)\s*(alch_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        params = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a == "void" or not a:
                continue
            a = re.sub(r"\bconst\b", "", a).strip()
            m = re.match(r"(\w+)\s*(\**)\s*\w+$", a)
            assert m, (name, a)
            params.append(m.group(1) + m.group(2))
        protos[name] = ((ret + star).replace("  ", " "), params)
    return protos


def haskell_imports():
    text = open(os.path.join(ROOT, "haskell", "Crypto", "Lol", "Cyclotomic", "Tensor", "GT", "Backend.hs")).read()
    imps = {}
    for safety, cname, sig in re.findall(r'^foreign import ccall (safe|unsafe)\s+"(\w+)"\s+\w+\s*::\s*(.+)$', text, flags=re.M):
        assert cname not in imps, f"{cname} imported twice"
        imps[cname] = (safety, [t.strip() for t in sig.split("->")])
    return imps


def test_every_entry_point_is_imported_with_the_header_signature():
    protos, imps = header_prototypes(), haskell_imports()
    assert len(protos) >= 45
    assert set(imps) == set(protos), (sorted(set(protos) - set(imps)), sorted(set(imps) - set(protos)))
    for name, (ret, params) in protos.items():
        _, sig = imps[name]
        assert sig[-1] == RET[ret], (name, sig[-1], ret)
        assert sig[:-1] == [CTYPE[p] for p in params], (name, sig[:-1], params)


def test_blocking_calls_are_safe_imports():
    imps = haskell_imports()
    for name in ("alch_ct_mul_relin", "alch_ct_mul_full", "alch_sync", "alch_buf_upload", "alch_buf_download",
                 "alch_crt", "alch_crtinv", "alch_hint_load"):
        assert imps[name][0] == "safe", name
