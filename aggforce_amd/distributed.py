"""Frame-sharded data parallelism: one process per GPU, one all-reduce of the Gram matrix.

The reference has no distributed code.  The force-map objective is a sum over frames, so
every rank builds the Gram matrix of its own frame shard (K1), the shards are summed with a
single all-reduce (RCCL over xGMI through ``torch.distributed`` backend "nccl"; "gloo" on
CPU tensors in the unit tests), and every rank then runs the identical, replicated solve
(K2) -- no broadcast is needed -- and maps its own frames (K3).
"""
from typing import Tuple

import torch


def resolve_comm(comm):
    """None/False -> no communication; True -> default process group; else the group itself."""
    if comm is None or comm is False:
        return None
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("comm was given but torch.distributed is not initialised")
    if comm is True:
        return dist.group.WORLD
    return comm


def world_size(comm) -> int:
    group = resolve_comm(comm)
    if group is None:
        return 1
    import torch.distributed as dist

    return dist.get_world_size(group)


def all_reduce_sum_(t: torch.Tensor, comm) -> torch.Tensor:
    """In-place sum over the ranks of ``comm`` (no-op without a communicator)."""
    group = resolve_comm(comm)
    if group is None:
        return t
    import torch.distributed as dist

    if dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def frame_shard(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) frame range of ``rank``; sizes differ by at most one frame."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, extra = divmod(n_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)
