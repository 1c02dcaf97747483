"""CPU, world_size 2, gloo: the N > 1 plumbing bench.py uses -- rendezvous on 127.0.0.1, contiguous
sharding of the ciphertext batch with no data-path collective, barrier, MAX-over-ranks of the step time and
the whole-job throughput formula."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, os.environ["ALCH_ROOT"])
    from alchemy_amd import shard
    rank, local_rank, world, dist = shard.init_distributed("gloo")
    assert world == 2 and dist is not None and dist.get_backend() == "gloo"
    total = 1000 + 1                     # odd: ranks get 501 / 500
    sh = shard.partition(total, world, rank)
    # every rank "processes" its own shard; rank 1 is slower
    fake_step_seconds = 0.25 + 0.5 * rank
    shard.barrier(dist)
    t_max = shard.max_over_ranks(fake_step_seconds, dist)
    n_sum = shard.sum_over_ranks(sh.count, dist)
    first_sum = shard.sum_over_ranks(sh.first, dist)
    import numpy as np
    hint = np.arange(24, dtype=np.int64).reshape(2, 3, 4) * (7 if rank == 0 else -1)     # only rank 0 holds the hint
    shard.broadcast_array(hint, dist, src=0)
    assert int(hint.sum()) == 7 * sum(range(24)), hint
    shard.barrier(dist)
    print(json.dumps({"rank": rank, "count": sh.count, "first": sh.first, "t_max": t_max, "n_sum": n_sum,
                      "first_sum": first_sum, "value": n_sum / t_max}), flush=True)
    dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharding_and_timing():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ALCH_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    import json
    res = sorted((json.loads(o.strip().splitlines()[-1]) for o in outs), key=lambda r: r["rank"])
    assert [r["count"] for r in res] == [501, 500]
    assert [r["first"] for r in res] == [0, 501]
    for r in res:
        assert r["n_sum"] == 1001 and r["first_sum"] == 501
        assert abs(r["t_max"] - 0.75) < 1e-9            # the slowest rank sets the job time
        assert abs(r["value"] - 1001 / 0.75) < 1e-6
