"""Sharding of independent ciphertexts over ranks (one process per GPU).

The path has no cross-ciphertext dataflow (every reference op is a pure function of single values,
Crypto/Alchemy/Interpreter/Eval.hs:41-53), so ranks own contiguous chunks of the batch, the hint is
replicated (broadcast once, before timing), and no collective runs inside the timed region.  torch.distributed is used for the
rendezvous, the barrier and the max-over-ranks of the step time; RCCL moves data only when a caller
explicitly gathers results (outside the timed region).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    first: int      # first ciphertext index owned by this rank
    count: int      # ciphertexts owned by this rank


def partition(total: int, world: int, rank: int) -> Shard:
    """Contiguous split of `total` ciphertexts; the first (total % world) ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad partition arguments")
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return Shard(rank, world, first, count)


def env_rank():
    """(rank, local_rank, world) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed when WORLD_SIZE > 1.  Returns (rank, local_rank, world, dist or None)."""
    rank, local_rank, world = env_rank()
    if world == 1 and not os.environ.get("ALCH_DIST_FORCE"):
        return rank, local_rank, world, None
    # ALCH_DIST_FORCE=1 (tests): a one-rank process group, so that every collective of the N > 1 path runs through the real
    # backend (nccl = RCCL) on a one-GPU box
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def broadcast_array(arr, dist, src: int = 0, device=None):
    """Broadcast a numpy array from rank `src` to every rank, in place (the key-switch hint, once per circuit,
    outside any timed region).  With the nccl (= RCCL) backend the bytes travel device to device over xGMI."""
    if dist is None:
        return arr
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    arr[...] = t.cpu().numpy()
    return arr


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist, device=None) -> float:
    """MAX all-reduce of a scalar (the step time of the slowest rank)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scalars(value: float, dist, device=None):
    """All ranks' values of one scalar, in rank order (per-rank step times in the bench line)."""
    if dist is None:
        return [float(value)]
    import torch
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu().tolist()]


def sum_over_ranks(value: float, dist, device=None) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
