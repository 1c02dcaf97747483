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


def test_rccl_module_imports_every_entry_point_of_its_header():
    """haskell/.../GT/Rccl.hs against include/alchemy_rccl.h: same mechanical check (name, arity, types), every import safe (each of
    these calls can block on RCCL), and the id size equals the header's ALCH_COMM_ID_BYTES."""
    ctype = dict(CTYPE)
    ctype.update({"alch_comm*": "Ptr AlchComm", "alch_comm**": "Ptr (Ptr AlchComm)", "unsigned char*": "Ptr Word8", "alch_buf**": "Ptr (Ptr AlchBuf)"})
    text = open(os.path.join(ROOT, "include", "alchemy_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for ret, star, name, args in re.findall(r"^((?:const\s+)?\w+)\s*(\*?)\s*(alch_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        params = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a == "void" or not a:
                continue
            a = re.sub(r"\*\s+\*", "**", re.sub(r"\bconst\b", "", a)).strip()
            m = re.match(r"((?:unsigned\s+)?\w+)\s*(\**)\s*\w+$", a)
            assert m, (name, a)
            params.append(m.group(1) + m.group(2))
        protos[name] = ((ret + star).replace("  ", " "), params)
    hs = open(os.path.join(ROOT, "haskell", "Crypto", "Lol", "Cyclotomic", "Tensor", "GT", "Rccl.hs")).read()
    imps = {}
    for safety, cname, sig in re.findall(r'^foreign import ccall (safe|unsafe)\s+"(\w+)"\s+\w+\s*::\s*(.+)$', hs, flags=re.M):
        assert cname not in imps and safety == "safe", cname
        imps[cname] = [t.strip() for t in sig.split("->")]
    assert len(protos) == 9 and set(imps) == set(protos), (sorted(set(protos) ^ set(imps)))
    for name, (ret, params) in protos.items():
        assert imps[name][-1] == RET[ret], (name, imps[name][-1], ret)
        assert imps[name][:-1] == [ctype[p] for p in params], (name, imps[name][:-1], params)
    n = re.search(r"#define ALCH_COMM_ID_BYTES (\d+)", open(os.path.join(ROOT, "include", "alchemy_rccl.h")).read()).group(1)
    assert re.search(r"^commIdBytes = " + n + r"$", hs, flags=re.M)


def test_blocking_calls_are_safe_imports():
    imps = haskell_imports()
    for name in ("alch_ct_mul_relin", "alch_ct_mul_full", "alch_sync", "alch_buf_upload", "alch_buf_download",
                 "alch_crt", "alch_crtinv", "alch_hint_load"):
        assert imps[name][0] == "safe", name


def test_unsafe_imports_are_pure_accessors_only():
    """An `unsafe` foreign call cannot be pre-empted: it must never wait.  Since every entry point that touches a ring holds the
    per-device lock (and may synchronise a stream), only accessors whose C body takes no lock and issues no HIP call may be `unsafe`."""
    imps = haskell_imports()
    unsafe = sorted(n for n, (safety, _) in imps.items() if safety == "unsafe")
    src = ""
    for f in ("alchemy_hip.hip", "tensor_ext.inc.hpp"):
        src += open(os.path.join(ROOT, "alchemy_amd", "csrc", f)).read()
    for name in unsafe:
        m = re.search(r'^extern "C" [^\n]*\b%s\([^{]*\{(.*?)^\}' % name, src, flags=re.M | re.S)
        if m is None:                                            # one-line definitions
            m = re.search(r'^extern "C" [^\n]*\b%s\(.*$' % name, src, flags=re.M)
        assert m is not None, name
        body = m.group(0)
        for forbidden in ("BIND(", "ALCH_DEVICE_LOCK", "hipStreamSynchronize", "hipMemcpy", "hipLaunchKernelGGL", "hipMalloc"):
            assert forbidden not in body, f"{name} is imported unsafe but its body contains {forbidden}"


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
    for stub in ("nonsensical",):                                    # the one run-time error Lol's own backends raise too: randomR
        assert len(re.findall(r"\b%s\b" % stub, inst)) == 0, "a Tensor method may not be an error"
    assert not re.search(r"\berror\s+\"(coerce|see|as )", code)


ORDER_DEPENDENT = {   # Tensor method -> what its definition must reach (through its named helpers): entry points and op codes
    "l": ["c_bufTensorOp", "opL"], "lInv": ["c_bufTensorOp", "opLInv"], "mulGPow": ["c_bufTensorOp", "opMulGPow"],
    "mulGDec": ["c_bufTensorOp", "opMulGDec"], "divGPow": ["c_bufTensorOp", "opDivGPow"], "divGDec": ["c_bufTensorOp", "opDivGDec"],
    "crtFuncs": ["c_bufTensorOp", "opCRT", "opCRTInv", "opMulGCRT", "opDivGCRT"],
    "crtExtFuncs": ["c_bufTwace", "c_bufEmbed"], "twacePowDec": ["c_bufTwace"], "embedPow": ["c_bufEmbed"],
    "embedDec": ["c_bufEmbed"], "coeffs": ["c_bufCoeffs"], "powBasisPow": ["c_extTable"], "crtSetDec": ["c_crtSetDec"],
}


def _top_level_body(text, name):
    """Source of the top-level definition `name` (from its type signature to the next blank line followed by a top-level item)."""
    m = re.search(r"^%s\s*::" % re.escape(name), text, flags=re.M)
    if not m:
        return ""
    rest = text[m.start():]
    end = re.search(r"\n\n(?=\S)", rest)
    return rest[:end.start()] if end else rest


def _closure(text, body, depth=3):
    """`body` plus the definitions of every top-level helper of this module it mentions, transitively."""
    seen, out = set(), body
    for _ in range(depth):
        for ident in set(re.findall(r"\b([a-z][A-Za-z0-9']*)\b", out)) - seen:
            seen.add(ident)
            out += _top_level_body(text, ident)
    return out


def test_no_basis_order_dependent_method_is_delegated_to_lol_cpp():
    """`instance Tensor GT` is sound only if every method whose result depends on a basis ORDER (CRT slots,
    relative powerful / decoding bases) comes from the library for the element types the library serves -- mixing lol-cpp's slot
    order with the library's would make `embed` / `twace` of CRT-basis elements silently wrong.  Each such method must reach its
    C entry point (with its op code, for the unary methods); no rewrite RULES are relied on."""
    text = _gt_source()
    inst = text[text.index("instance Tensor GT where"):text.index("-- | phi(m) as an Int.")]
    for method, syms in ORDER_DEPENDENT.items():
        m = re.search(r"^  %s\b[^\n]*=(.*?)(?=^  \w[\w']*\s[^\n]*=|\Z)" % re.escape(method), inst, flags=re.M | re.S)
        assert m, method
        body = _closure(text, m.group(1))
        for sym in syms:
            assert re.search(r"\b%s\b" % sym, body), f"{method} does not reach {sym}: it would run in lol-cpp's basis order"
    assert "{-# RULES" not in text


def test_op_codes_equal_the_header():
    text = _gt_source()
    header = open(os.path.join(ROOT, "include", "alchemy_hip.h")).read()
    for hs, c in (("opCRT", "ALCH_T_CRT"), ("opCRTInv", "ALCH_T_CRTINV"), ("opL", "ALCH_T_L"), ("opLInv", "ALCH_T_LINV"),
                  ("opMulGPow", "ALCH_T_MULG_POW"), ("opMulGDec", "ALCH_T_MULG_DEC"), ("opMulGCRT", "ALCH_T_MULG_CRT"),
                  ("opDivGPow", "ALCH_T_DIVG_POW"), ("opDivGDec", "ALCH_T_DIVG_DEC"), ("opDivGCRT", "ALCH_T_DIVG_CRT")):
        want = int(re.search(r"#define %s (\d+)" % c, header).group(1))
        got = int(re.search(r"\b%s = (\d+)" % hs, text).group(1))
        assert got == want, (hs, got, want)
    # the basis arguments of alch_buf_embed / alch_buf_twace used in the file: 0 Pow, 1 Dec, 2 CRT
    for name, val in (("ALCH_BASIS_POW", 0), ("ALCH_BASIS_DEC", 1), ("ALCH_BASIS_CRT", 2)):
        assert int(re.search(r"#define %s (\d+)" % name, header).group(1)) == val
    assert re.search(r'"twaceCRT" \(\\pd ps -> c_bufTwace pd ps 1 2\)', text) and re.search(r'"embedCRT" \(\\pd ps -> c_bufEmbed pd ps 1 2\)', text)
    assert re.search(r'"embedDec" \(\\pd ps -> c_bufEmbed pd ps 1 1\)', text) and re.search(r'"embedPow" \(\\pd ps -> c_bufEmbed pd ps 1 0\)', text)


def test_every_status_that_is_a_lol_nothing_has_a_branch_that_is_not_an_error():
    """VERDICT r03 item 1d.  include/alchemy_hip.h documents three statuses that stand for a Lol answer rather than a failure:
    ALCH_E_NO_CRT (-3: crtFuncs = Nothing -> the ring without CRT basis), ALCH_E_UNSUPPORTED (-4: the pair stays on lol-cpp) and
    ALCH_NOT_DIVISIBLE (1: divG = Nothing).  Every `case rc of` of GT.hs that can see one of them must map it to a value."""
    text = _gt_source()
    code = "\n".join(l.split("--")[0] for l in text.splitlines())
    blocks = re.findall(r"case rc of\n((?:[ \t]+[^\n]*\n)+)", code)
    assert len(blocks) >= 2
    ring_blocks = [b for b in blocks if "Served" in b]
    assert len(ring_blocks) == 1
    rb = ring_blocks[0]
    branches = {}                                                    # status token -> the text of its branch
    cur = None
    for line in rb.splitlines():
        m = re.match(r"\s*(0|\(-\d+\)|_)\s+->", line)
        if m:
            cur = m.group(1)
            branches[cur] = ""
        if cur is not None:
            branches[cur] += line + "\n"
    for status, ctor in (("0", "Served"), ("(-3)", "NoCRT"), ("(-4)", "NotServed")):
        assert status in branches and ("return " + ctor in branches[status] or "return (" + ctor in branches[status]), (status, branches.get(status))
        assert "rror" not in branches[status], (status, branches[status])
    div_blocks = [b for b in blocks if "Nothing" in b and "Served" not in b]
    assert div_blocks and all(re.search(r"^\s*1\s*->\s*return Nothing", b, flags=re.M) for b in div_blocks)
    # the answers are then used: NoCRT falls back to alch_ring_create_nocrt, NotServed to lol-cpp, for EVERY method family
    pow_ring = _top_level_body(text, "powRing")
    assert "NoCRT" in pow_ring and "ringFor True" in pow_ring and "NotServed" in pow_ring
    crt_ring = _top_level_body(text, "crtRing")
    assert all(k in crt_ring for k in ("CRTServed", "CRTNothing", "CRTLolCpp"))
    for helper in ("crtFuncsGT", "crtExtFuncsGT"):
        assert "-> host" in _top_level_body(text, helper), helper
    for helper in ("twacePowDecGT", "embedPowGT", "embedDecGT", "coeffsGT", "powBasisPowGT"):
        assert re.search(r"_\s+-> host", _top_level_body(text, helper)), helper
    # no pure `error` anywhere: failures that are not Lol answers (no device, HIP errors) raise in IO with the library's message
    assert not re.search(r"(?<![A-Za-z])error\s+\(", code.replace("ioError", "").replace("userError", ""))


def test_gt_is_a_sum_of_a_host_and_a_device_representation():
    text = _gt_source()
    assert re.search(r"^data GT .*where\n\s+GTHost ::.*\n(?:\s+--[^\n]*\n)*\s+GTDev\s+::", text, flags=re.M)
    # downloads happen only in hostOf; the methods that need host data say so by calling it
    assert text.count("c_bufDownload") == 1 and "c_bufDownload" in _top_level_body(text, "hostOf")
    inst = text[text.index("instance Tensor GT where"):text.index("-- | phi(m) as an Int.")]
    for method in ("zipWithT", "fmapT", "unzipT", "gSqNormDec"):
        assert re.search(r"^  %s\b[^\n]*hostOf" % method, inst, flags=re.M), method
    # finalizers release pooled buffers; views keep their parent alive
    assert "FC.newForeignPtr" in _top_level_body(text, "newElems") and "touchForeignPtr parent" in _top_level_body(text, "viewElem")


def test_every_foreign_symbol_used_by_the_instance_is_imported_by_the_backend():
    text, imps = _gt_source(), haskell_imports()
    hs_names = set(re.findall(r'^foreign import ccall (?:safe|unsafe)\s+"\w+"\s+(\w+)\s*::', open(os.path.join(
        ROOT, "haskell", "Crypto", "Lol", "Cyclotomic", "Tensor", "GT", "Backend.hs")).read(), flags=re.M))
    used = set(re.findall(r"\bc_[A-Za-z0-9]+\b", text))
    assert used and used <= hs_names, sorted(used - hs_names)
    # the hot Tensor methods of SURVEY 8b all cross the FFI
    for sym in ("c_bufTensorOp", "c_bufMul", "c_bufAdd", "c_bufSub", "c_bufEmbed", "c_bufTwace", "c_bufCoeffs", "c_bufAlloc", "c_bufFree",
                "c_bufView", "c_bufUpload", "c_bufDownload", "c_ringShareStream", "c_ctMulRelin", "c_ctMulFull", "c_ctTunnel"):
        assert sym in used, sym
    assert len(imps) >= 55


def test_example_variant_patch_touches_only_the_tensor_type():
    patch = open(os.path.join(ROOT, "haskell", "examples", "Arithmetic-GT.patch")).read()
    minus = [l for l in patch.splitlines() if l.startswith("-") and not l.startswith("---")]
    plus = [l for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert len(minus) == len(plus) == 2
    assert "Tensor.CPP" in minus[0] and "Tensor.GT" in plus[0]
    assert minus[1].replace(" CT ", " GT ") == "-" + plus[1][1:]
