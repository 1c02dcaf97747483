"""CPU: the Haskell FFI module shipped as source (haskell/.../GT/Backend.hs) checked mechanically
against include/alchemy_hip.h -- every entry point of the header is imported exactly once, with the header's
arity and C types (no Haskell toolchain exists here, so this is the only check that file gets)."""
import os
import re

from conftest import ROOT

CTYPE = {  # C parameter type (qualifiers stripped) -> Haskell FFI type
    "uint32_t": "Word32", "uint64_t": "Word64", "int": "CInt", "size_t": "CSize", "unsigned": "CUInt", "long": "CLong",
    "char*": "CString",
    "uint64_t*": "Ptr Word64", "uint32_t*": "Ptr Word32", "int32_t*": "Ptr Int32", "int*": "Ptr CInt", "size_t*": "Ptr CSize",
    "float*": "Ptr CFloat", "int64_t*": "Ptr Int64", "void*": "Ptr ()", "void**": "Ptr (Ptr ())",
    "alch_ring*": "Ptr AlchRing", "alch_buf*": "Ptr AlchBuf", "alch_hint*": "Ptr AlchHint",
    "alch_tunnel*": "Ptr AlchTunnel", "alch_tunnel**": "Ptr (Ptr AlchTunnel)",
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


TENSOR_METHODS = ["scalarPow", "l", "lInv", "mulGPow", "mulGDec", "divGPow", "divGDec", "crtFuncs", "tGaussianDec",
                  "gSqNormDec", "twacePowDec", "embedPow", "embedDec", "crtExtFuncs", "coeffs", "powBasisPow", "crtSetDec",
                  "fmapT", "zipWithT", "unzipT", "entailIndexT", "entailEqT", "entailZTT", "entailNFDataT",
                  "entailRandomT", "entailShowT", "entailModuleT"]        # Lol 0.7's class Tensor


def _gt_source():
    return open(os.path.join(ROOT, "haskell", "Crypto", "Lol", "Cyclotomic", "Tensor", "GT.hs")).read()


def test_tensor_instance_defines_every_method_and_none_is_a_stub():
    text = _gt_source()
    inst = text[text.index("instance Tensor GT where"):text.index("-- | phi(m) as an Int.")]
    for name in TENSOR_METHODS:
        assert re.search(r"^  %s\b[^\n]*=" % re.escape(name), inst, flags=re.M), f"instance Tensor GT lacks {name}"
    code = "\n".join(l.split("--")[0] for l in text.splitlines())            # comments stripped
    assert not re.search(r"=\s*error\b", code) and "undefined" not in code, "stub bodies are not allowed"
    assert not re.search(r"\berror\s+\"(coerce|see|as )", code)


ORDER_DEPENDENT = {   # Tensor method -> the library entry points its definition must reach (directly or through its named helper)
    "l": ["c_l"], "lInv": ["c_lInv"], "mulGPow": ["c_mulGPow"], "mulGDec": ["c_mulGDec"], "divGPow": ["c_divGPow"],
    "divGDec": ["c_divGDec"], "crtFuncs": ["c_crt", "c_crtInv", "c_mulGCRT", "c_divGCRT"],
    "crtExtFuncs": ["c_twaceCRT", "c_embedCRT"], "twacePowDec": ["c_twacePowDec"], "embedPow": ["c_embedPow"],
    "embedDec": ["c_embedDec"], "coeffs": ["c_coeffs"], "powBasisPow": ["c_extTable"], "crtSetDec": ["c_crtSetDec"],
}


def _top_level_body(text, name):
    """Source of the top-level definition `name` (from its type signature to the next blank line followed by a top-level item)."""
    m = re.search(r"^%s\s*::" % re.escape(name), text, flags=re.M)
    assert m, name
    rest = text[m.start():]
    end = re.search(r"\n\n(?=\S)", rest)
    return rest[:end.start()] if end else rest


def test_no_basis_order_dependent_method_is_delegated_to_lol_cpp():
    """`instance Tensor GT` is sound only if every method whose result depends on a basis ORDER (CRT slots,
    relative powerful / decoding bases) comes from the library for the element types the library serves -- mixing lol-cpp's slot
    order with the library's would make `embed` / `twace` of CRT-basis elements silently wrong.  Each such method must reach its
    C entry point; no rewrite RULES are relied on."""
    text = _gt_source()
    inst = text[text.index("instance Tensor GT where"):text.index("-- | phi(m) as an Int.")]
    for method, syms in ORDER_DEPENDENT.items():
        m = re.search(r"^  %s\b[^\n]*=(.*?)(?=^  \w[\w']*\s[^\n]*=|\Z)" % re.escape(method), inst, flags=re.M | re.S)
        assert m, method
        body = m.group(1)
        for helper in re.findall(r"\b(\w+GT)\b", body):
            body += _top_level_body(text, helper)
        for sym in syms:
            assert re.search(r"\b%s\b" % sym, body), f"{method} does not reach {sym}: it would run in lol-cpp's basis order"
    assert "{-# RULES" not in text


def test_every_foreign_symbol_used_by_the_instance_is_imported_by_the_backend():
    text, imps = _gt_source(), haskell_imports()
    hs_names = set(re.findall(r'^foreign import ccall (?:safe|unsafe)\s+"\w+"\s+(\w+)\s*::', open(os.path.join(
        ROOT, "haskell", "Crypto", "Lol", "Cyclotomic", "Tensor", "GT", "Backend.hs")).read(), flags=re.M))
    used = set(re.findall(r"\bc_[A-Za-z0-9]+\b", text))
    assert used and used <= hs_names, sorted(used - hs_names)
    # the hot Tensor methods of SURVEY 8b all cross the FFI
    for sym in ("c_crt", "c_crtInv", "c_mulGPow", "c_mulGDec", "c_mulGCRT", "c_divGPow", "c_divGDec", "c_divGCRT", "c_l", "c_lInv",
                "c_mul", "c_add", "c_ctMulRelin", "c_ctMulFull"):
        assert sym in used, sym
    assert len(imps) >= 55


def test_example_variant_patch_touches_only_the_tensor_type():
    patch = open(os.path.join(ROOT, "haskell", "examples", "Arithmetic-GT.patch")).read()
    minus = [l for l in patch.splitlines() if l.startswith("-") and not l.startswith("---")]
    plus = [l for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert len(minus) == len(plus) == 2
    assert "Tensor.CPP" in minus[0] and "Tensor.GT" in plus[0]
    assert minus[1].replace(" CT ", " GT ") == "-" + plus[1][1:]
