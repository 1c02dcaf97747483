"""CPU: tests/golden/batch_checksums.json is what tests/golden/make_batch_checksums.py produces today -- one ciphertext of every
section is recomputed on the C restatement (the generator's own per-ciphertext functions) and compared with the committed value.
Guards against a fixture that went stale when the op sequence, the seeds or the oracle changed."""
import importlib.util
import os

import pytest

from helpers import GOLDEN, load_golden


@pytest.fixture(scope="module")
def gen(oracle_lib):
    spec = importlib.util.spec_from_file_location("make_batch_checksums", os.path.join(GOLDEN, "make_batch_checksums.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_homomrlwr_fixture_is_current(gen):
    ref = load_golden("batch_checksums.json")["homomrlwr"]
    assert len(ref["per_ciphertext"]) == ref["batch"] >= 64
    for ct in (0, ref["batch"] - 1):
        assert f"{gen.ringround_ct(ct):016x}" == ref["per_ciphertext"][ct]


def test_tunnel_hs_fixture_is_current(gen):
    hops = load_golden("batch_checksums.json")["tunnel_hs"]["hops"]
    assert [h["hop"] for h in hops] == list(range(5))
    for h in hops:
        assert f"{gen.hop_ct((h['hop'], 5)):016x}" == h["per_ciphertext"][5]
        assert f"{sum(int(x, 16) for x in h['per_ciphertext']) & gen.MASK:016x}" == h["checksum"]


def test_general_index_and_config2_fixtures_are_current(gen):
    ref = load_golden("batch_checksums.json")
    g = ref["general_index"]
    assert f"{sum(gen.general_ct(ct) for ct in range(4)) & gen.MASK:016x}" == g["first_4"]
    assert g["test"]["batch"] < g["bench"]["batch"]
    c2 = ref["config2"]
    for label in ("q60", "q31"):
        sums = [gen.config2_poly((label, e)) for e in range(2)]
        assert [f"{sum(a for a, _ in sums) & gen.MASK:016x}", f"{sum(b for _, b in sums) & gen.MASK:016x}"] == c2[label]["first_2"]
        assert c2[label]["modulus"] == gen.C2_QS[label]


def test_two_power_fixtures_are_current(gen):
    """The headline, < 2^30-moduli and n = 2^16 sections: the checksum of the first two results, recomputed on the C restatement."""
    ref = load_golden("batch_checksums.json")
    assert f"{gen.relin_range(0, 2):016x}" == ref["first_2"]
    assert ref["q30"]["moduli"] == gen.Q30_QS and max(gen.Q30_QS) < 1 << 30
    assert f"{gen.relin_range(0, 2, gen.Q30_QS):016x}" == ref["q30"]["first_2"]
    assert ref["n16"]["moduli"] == gen.SIX_QS_17 and ref["n16"]["n"] == 1 << 16
    assert f"{gen.relin_range(0, 2, gen.SIX_QS_17, 1 << 16):016x}" == ref["n16"]["first_2"]


def test_bench_extra_fixtures_are_current(gen):
    """bench.py's full_mul (B = 4096) and Pow-basis in/out (B = 2048) lines, asserted since round 4."""
    ref = load_golden("batch_checksums.json")
    ex = ref["bench_extra"]
    assert ex["full_mul"]["batch"] == 4096 and ex["pow_in_out"]["batch"] == 2048
    assert f"{gen.pow_range(0, 2):016x}" == ex["pow_in_out"]["first_2"]
    # the full_mul batch extends the test batch: its checksum is the test batch's plus a tail, so one recomputed ciphertext of the tail pins it
    head = int(ref["test_mul_full"]["checksum"], 16)
    assert ex["full_mul"]["checksum"] != f"{head:016x}" and ref["test_mul_full"]["batch"] < ex["full_mul"]["batch"]


def test_bench_reports_why_a_line_was_not_checked():
    """a stale fixture or a non-default --batch must show up in the line, and a wrong batch must end the run."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert "holds no entry" in bench.check_batch("x", None, 8, lambda: 0)["skipped"]
    assert "non-default --batch" in bench.check_batch("x", {"batch": 16, "checksum": "0" * 16}, 8, lambda: 0)["skipped"]
    assert bench.check_batch("x", {"batch": 8, "checksum": f"{5:016x}"}, 8, lambda: 5)["ok"] is True
    import pytest
    with pytest.raises(SystemExit):
        bench.check_batch("x", {"batch": 8, "checksum": f"{5:016x}"}, 8, lambda: 6)
    assert bench.golden_checksums()["bench_mul_relin"]["batch"] == 8192
