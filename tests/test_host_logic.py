"""CPU: host-side logic of the product -- the C-ABI library loads and exports every symbol the header
declares, validates its arguments, applies the documented root rule, and fails loudly without a GPU."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ARITH_QS, CFG2_Q60, CFG3_QS, ROOT

import alchemy_amd as A
from alchemy_amd import capi, shard
from oracle import model as M


def _header_functions():
    text = open(os.path.join(ROOT, "include", "alchemy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alch_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = A.load_library()
    declared = _header_functions()
    assert len(declared) >= 35
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared          # the binding covers the whole header
    assert lib.alch_version() >> 16 == 1


def test_library_links_no_oracle_code():
    """The product must not route through the oracle: no orc_* symbol, no liblol_oracle dependency."""
    out = subprocess.run(["nm", "-D", A.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "orc_" not in out
    ldd = subprocess.run(["ldd", A.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
    for dirpath, _, files in os.walk(os.path.join(ROOT, "alchemy_amd")):
        for f in files:
            if not f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                continue
            for line in open(os.path.join(dirpath, f)):
                st = line.strip()
                if st.startswith(("import ", "from ")):
                    assert "oracle" not in st, (f, st)
                if st.startswith("#include"):
                    assert "oracle" not in st and "lol_tensor" not in st, (f, st)


def test_only_tests_smoke_and_bench_use_the_oracle():
    """oracle/ is test infrastructure: outside oracle/ itself, only tests/, __graft_entry__.py and bench.py may import it or load
    its library; tools/ and examples/ (measurement and replay programs of the product) must not."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    skip = {os.path.join(ROOT, d) for d in ("tests", "oracle", "gpurun_out", ".git", "docs", "profiles")}
    offenders = []
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if os.path.join(dirpath, d) not in skip and d != "__pycache__"]
        for f in files:
            path = os.path.join(dirpath, f)
            if path in allowed or not f.endswith((".py", ".sh", ".cpp", ".hpp", ".hip", ".h", ".c")):
                continue
            for line in open(path, errors="replace"):
                st = line.strip()
                if st.startswith(("import oracle", "from oracle")) or (st.startswith("#include") and ("oracle/" in st or "lol_tensor" in st)) \
                        or "liblol_oracle" in st:
                    offenders.append((os.path.relpath(path, ROOT), st))
    assert not offenders, offenders


@pytest.mark.parametrize("m,q", [(32, ARITH_QS[0]), (512, ARITH_QS[1]), (512, ARITH_QS[2]), (1 << 16, CFG3_QS[0]),
                                 (1 << 16, CFG3_QS[3]), (1 << 15, CFG2_Q60), (4096, 12289), (1 << 16, 65537)])
def test_root_rule_matches_oracle(oracle_lib, m, q):
    psi, g = capi.host_root(m, q)
    assert g == M.smallest_generator(q) == oracle_lib.smallest_generator(q)
    assert psi == M.root_2n(q, m // 2)
    assert pow(psi, m // 2, q) == q - 1               # primitive m-th root


def test_host_root_rejects_bad_arguments():
    with pytest.raises(A.AlchemyError) as e:
        capi.host_root(1 << 16, ARITH_QS[1])          # 8392193 is not 1 mod 2^16
    assert e.value.code == capi.ALCH_E_NO_CRT
    with pytest.raises(A.AlchemyError) as e:
        capi.host_root(512, 268440579)                # not prime
    assert e.value.code == capi.ALCH_E_NOT_PRIME


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device error path")
def test_ring_create_fails_loudly_without_gpu():
    with pytest.raises(A.AlchemyError) as e:
        A.Ring(512, ARITH_QS)
    assert e.value.code == capi.ALCH_E_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_ring_argument_validation_precedes_device_probe():
    """The status ORDER is part of the ABI (include/alchemy_hip.h, alch_ring_create): malformed -> INVALID; no CRT basis over the
    base ring (composite q, q = 2, prime q != 1 mod m: Lol's crtFuncs = Nothing) -> NO_CRT whatever the index; only then
    UNSUPPORTED.  haskell/.../GT.hs `ringFor` maps NO_CRT to the no-CRT ring and UNSUPPORTED to lol-cpp."""
    for args, code in [((512, [268440579]), capi.ALCH_E_NO_CRT),             # composite: Lol's crtInfo needs a prime
                       ((11648, [32]), capi.ALCH_E_NO_CRT),                   # Z_{2^5} over H0': the plaintext ring of HomomRLWR
                       ((128, [2]), capi.ALCH_E_NO_CRT),                      # Z_2 (Z2E 1, examples/Common.hs:32)
                       ((4, [7]), capi.ALCH_E_NO_CRT),                        # Arithmetic's PT = Cyc F4 (Zq 7): 7 != 1 mod 4
                       ((4 * 17, [32]), capi.ALCH_E_NO_CRT),                  # no CRT basis beats "unsupported index"
                       ((4 * 17, [ARITH_QS[1]]), capi.ALCH_E_NO_CRT),         # 8392193 != 1 mod 17
                       ((512, [ARITH_QS[0], 16]), capi.ALCH_E_NO_CRT),        # one bad limb of a pair is enough (CRTrans of a pair needs both)
                       ((512, [0]), capi.ALCH_E_INVALID),                     # the integers are alch_ring_create_nocrt's
                       ((512, [1]), capi.ALCH_E_INVALID),
                       ((1 << 16, [ARITH_QS[1]]), capi.ALCH_E_NO_CRT),        # q != 1 mod m
                       ((48, [ARITH_QS[1]]), capi.ALCH_E_NO_CRT),             # composite index, q != 1 mod 3
                       ((4 * 17, [268435577]), capi.ALCH_E_UNSUPPORTED),         # a CRT basis exists (268435577 = 1 mod 68) but 17 > 13: not served
                       ((3 * 5 * 7 * 11 * 13 * 64, [960961]), capi.ALCH_E_UNSUPPORTED),  # phi = 184320: a limb-polynomial exceeds the LDS
                       ((1 << 17, [CFG3_QS[3]]), capi.ALCH_E_NO_CRT),         # 2145976321 is 1 mod 2^16 only
                       ((1 << 18, [2146959361]), capi.ALCH_E_UNSUPPORTED),    # n = 2^17: beyond the split transform (32-bit words)
                       ((1 << 17, [1152921504606584833]), capi.ALCH_E_UNSUPPORTED),   # n = 2^16 with 64-bit words
                       ((512, [ARITH_QS[0], ARITH_QS[0]]), capi.ALCH_E_INVALID),
                       ((512, []), capi.ALCH_E_INVALID)]:
        with pytest.raises(A.AlchemyError) as e:
            A.Ring(*args)
        assert e.value.code == code, (args, str(e.value))


def test_partition_covers_batch_exactly():
    for total, world in [(8192, 1), (8192, 8), (10, 4), (3, 8), (0, 2), (65536, 3)]:
        shards = [shard.partition(total, world, r) for r in range(world)]
        assert sum(s.count for s in shards) == total
        pos = 0
        for s in shards:
            assert s.first == pos
            pos += s.count
        assert max(s.count for s in shards) - min(s.count for s in shards) <= 1
    with pytest.raises(ValueError):
        shard.partition(4, 2, 2)


def test_graft_entry_build_is_idempotent():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(A.lib_path())


def test_select_limbs_reproduces_pt2ct_type_level_choices():
    """alch_select_limbs against the hand simulation of SURVEY 3.3 / 3.1 (Noise.hs:107-170, PT2CT.hs:132-140,234-249,281-296):
    HomomRLWR resolved backwards from the output pNoise 0 (PNZ, examples/HomomRLWR.hs:47): three tree levels, x(1+x), then
    the five tunnels; Arithmetic's single multiplication."""
    rlwr = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]      # examples/HomomRLWR.hs:37-43
    assert [capi.load_library().alch_modulus_units(q) for q in rlwr] == [5, 4, 4, 4, 5, 5]
    p, got = 0, []
    for _ in range(4):                                   # tree level 3, 2, 1, then x * (1 + x); div2_ keeps the pNoise
        lin, lh, lout, p = capi.select_limbs(rlwr, p, capi.ALCH_OP_MUL)
        got.append((lin, lh, lout))
    assert got == [(1, 2, 1), (2, 3, 1), (3, 4, 2), (4, 5, 3)]
    tunnels = []
    for _ in range(5):                                   # switch5 .. switch1
        lin, lh, lout, p = capi.select_limbs(rlwr, p, capi.ALCH_OP_TUNNEL)
        tunnels.append((lin, lh, lout))
    assert tunnels == [(5, 5, 4), (5, 6, 5), (5, 6, 5), (5, 6, 5), (5, 6, 5)]
    arith = [268440577, 8392193, 1073750017]                                           # examples/Arithmetic.hs:31-34
    assert capi.select_limbs(arith, 0)[:3] == (2, 2, 1)
    # BaseBGad 2 hints need no Max32BitUnits head-room (PT2CT.hs:140)
    assert capi.select_limbs(rlwr, 11, capi.ALCH_OP_MUL, capi.ALCH_GAD_BASE2)[:3] == (4, 3, 3)
    with pytest.raises(A.AlchemyError):
        capi.select_limbs(arith, 9)                      # "You need more/bigger moduli!"


def test_lane_split_of_the_pipeline_batch():
    """alchemy_amd.ringround.lane_split (the sub-batches RingRoundLanes / ringround::Lanes run side by side): contiguous, covering,
    sizes within one of each other, never an empty lane."""
    from alchemy_amd.ringround import lane_split
    for batch in (1, 2, 5, 9, 70, 1024, 1025):
        for lanes in (1, 2, 3, 4, 8, 2000):
            sizes, firsts = lane_split(batch, lanes)
            assert len(sizes) == min(lanes, batch) and sum(sizes) == batch and min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
            assert firsts == [sum(sizes[:i]) for i in range(len(sizes))]
    assert lane_split(1024, 2) == ([512, 512], [0, 512]) and lane_split(70, 3) == ([24, 23, 23], [0, 24, 47])
