"""GPU: bench.py's multi-rank path rehearsed on the one-GPU box -- two ranks over gloo sharing cuda:0
(ALCH_DIST_BACKEND=gloo, ALCH_FORCE_DEVICE=0; the driver's real runs use nccl = RCCL with one rank per GPU):
hint broadcast, sharded steps, max-over-ranks timing, per-rank event times, result all-gather with the own slice checked.
A broken collective must END the run with a non-zero exit code instead of printing a healthy-looking line."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra_env):
    env = dict(os.environ, ALCH_DIST_BACKEND="gloo", ALCH_FORCE_DEVICE="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "64", "--cpu-ops", "0", "--no-full", "--no-pow", "--no-general", "--pipeline-batch", "48"]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_two_rank_rehearsal_prints_one_line():
    out = _run({})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert d["config"]["hint"].startswith("gloo broadcast")
    assert len(d["hip_event_ms_per_step_by_rank"]) == 2 and all(t > 0 for t in d["hip_event_ms_per_step_by_rank"])
    assert d["result_gather"]["own_slice_intact"] is True
    assert d["cpu_baseline"] is None
    # BASELINE config 4's pipeline: every rank ran its own shard, the rate is the aggregate
    h = d["homomrlwr"]
    assert h["n_gpus"] == 2 and h["ciphertexts_per_gpu"] == 48 and h["pipelines_per_s"] > 0
    assert h["tunnel_limbs"][0] == [5, 6, 5] and h["mul_limbs"][0] == [4, 5, 3]


def test_failed_broadcast_is_fatal():
    out = _run({"ALCH_TEST_FAIL_BROADCAST": "1"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "no result line may be printed"


def test_one_rank_process_group_runs_every_collective_through_rccl():
    """The N > 1 code path with the REAL backend (nccl = RCCL) as far as a one-GPU box allows: a one-rank process group
    (ALCH_DIST_FORCE=1), so the hint broadcast (int64 on the device), the MAX / SUM all-reduces, the barrier and the
    all_gather_into_tensor on a zero-copy view of the library's result buffer all go through RCCL's API."""
    env = dict(os.environ, ALCH_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    env.pop("ALCH_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--cpu-ops", "0",
           "--no-full", "--no-pow", "--no-general", "--no-tunnel-hs", "--no-config2", "--no-q30", "--pipeline-batch", "16"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["hint"].startswith("rccl broadcast"), d["config"]
    assert d["result_gather"]["backend"] == "rccl" and d["result_gather"]["own_slice_intact"] is True
    assert d["homomrlwr"]["pipelines_per_s"] > 0
