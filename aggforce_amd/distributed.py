"""Frame-sharded data parallelism: one process per GPU, one all-reduce of the Gram matrix.

The reference has no distributed code.  The force-map objective is a sum over frames, so
every rank builds the Gram matrix of its own frame shard (K1), the shards are summed with a
single all-reduce (RCCL over xGMI through ``torch.distributed`` backend "nccl"; "gloo" on
CPU tensors in the unit tests), and every rank then runs the identical, replicated solve
(K2) -- no broadcast is needed -- and maps its own frames (K3).
"""
import threading
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


def rank_of(comm) -> int:
    group = resolve_comm(comm)
    if group is None:
        return 0
    import torch.distributed as dist

    return dist.get_rank(group)


def all_reduce_sum_(t: torch.Tensor, comm) -> torch.Tensor:
    """In-place sum over the ranks of ``comm`` (no-op without a communicator)."""
    group = resolve_comm(comm)
    if group is None:
        return t
    import torch.distributed as dist

    if dist.get_world_size(group) > 1:
        _run_overlap_hooks()
        with _stage(t):
            _count(t)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


# Work that does not depend on the reduced matrix and may run BESIDE the collective: project_forces registers the
# slice-map gather of the coordinates here (HBM-bound, a side stream), so the Gram all-reduce -- link-bound, the
# compute units idle -- hides behind it instead of standing alone between K1 and K2.  Each hook runs once, in front of
# the first multi-rank collective after it was registered.
# (per host thread: a hook registered by one thread's fit must not be consumed by another thread's collective)
_tls = threading.local()


def _hooks() -> list:
    if not hasattr(_tls, "hooks"):
        _tls.hooks = []
    return _tls.hooks


def overlap_with_next_collective(thunk) -> None:
    _hooks().append(thunk)


def cancel_overlap(thunk) -> bool:
    """Remove a hook that no collective has consumed; True if it was still waiting."""
    hooks = _hooks()
    if thunk in hooks:
        hooks.remove(thunk)
        return True
    return False


def _run_overlap_hooks() -> None:
    hooks = _hooks()
    while hooks:
        hooks.pop(0)()


# payload bytes of the sum all-reduces since reset_collective_stats() (bench.py: bytes on the wire and bus bandwidth of
# the Gram all-reduce next to its measured time)
_collective = {"allreduce_bytes": 0, "allreduce_calls": 0}


def reset_collective_stats() -> None:
    _collective["allreduce_bytes"] = 0
    _collective["allreduce_calls"] = 0


def collective_stats() -> dict:
    return dict(_collective)


def _count(t: torch.Tensor) -> None:
    _collective["allreduce_bytes"] += t.numel() * t.element_size()
    _collective["allreduce_calls"] += 1


def _stage(t: torch.Tensor):
    """HIP-event bracket "allreduce" (pack + collective + unpack) on the launching stream, when bench.py's stage
    timers are on and the payload lives on the device; torch.distributed makes that stream wait for the collective, so
    the closing event sees its end."""
    import contextlib

    if not t.is_cuda:
        return contextlib.nullcontext()
    from . import _kernels as K

    return K._timed("allreduce")


def all_reduce_sum_sym_(G: torch.Tensor, comm) -> torch.Tensor:
    """In-place sum over the ranks of symmetric matrices G (..., n, n): only the upper triangles travel.

    The Gram matrices are the one large payload of the data-parallel path (134 MB at n = 4096, 4.8 GB for the 64
    sites of the featurised fit); packing halves the bytes on the xGMI links.  Device float64 matrices with
    n >= 256 are packed (aggf_sym_pack_upper / aggf_sym_unpack_upper); anything else goes through
    :func:`all_reduce_sum_` unchanged.  The result is exactly symmetric.
    """
    group = resolve_comm(comm)
    if group is None:
        return G
    import torch.distributed as dist

    if dist.get_world_size(group) == 1:
        return G
    n = G.shape[-1] if G.dim() >= 2 else 0
    if not (G.is_cuda and G.dtype == torch.float64 and G.dim() >= 2 and G.shape[-2] == n and n >= 256
            and G.is_contiguous() and G.numel() // (n * n) <= 65535):
        return all_reduce_sum_(G, comm)
    from . import _kernels as K

    _run_overlap_hooks()
    with _stage(G):
        packed = K.sym_pack_upper(G)
        _count(packed)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        return K.sym_unpack_upper(packed, G)


def all_reduce_minmax_(lo: torch.Tensor, hi: torch.Tensor, comm) -> None:
    """In place: elementwise minimum of ``lo`` and maximum of ``hi`` over the ranks of ``comm``."""
    group = resolve_comm(comm)
    if group is None:
        return
    import torch.distributed as dist

    if dist.get_world_size(group) > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)


def agree_on_min(value: int, comm, device) -> int:
    """The smallest ``value`` over the ranks of ``comm``: for quantities every rank derives from its own
    state (free memory, shard length) but that decide the SHAPE or NUMBER of later collectives."""
    group = resolve_comm(comm)
    if group is None:
        return int(value)
    import torch.distributed as dist

    if dist.get_world_size(group) == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def frame_shard(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) frame range of ``rank``; sizes differ by at most one frame."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, extra = divmod(n_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_extent(n_local: int, comm, device) -> Tuple[int, int]:
    """(first global frame of this rank's shard, total frames): shards are contiguous in rank order."""
    group = resolve_comm(comm)
    if group is None:
        return 0, n_local
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[dist.get_rank(group)] = n_local
    if world > 1:
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = sizes.cpu()
    return int(sizes[: dist.get_rank(group)].sum()), int(sizes.sum())


def agree_on_indices(idx, comm, device):
    """Rank 0's integer index array on every rank (an unseeded draw differs between ranks)."""
    import numpy as np

    group = resolve_comm(comm)
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
    if group is None:
        return idx
    import torch.distributed as dist

    if dist.get_world_size(group) == 1:
        return idx
    t = torch.from_numpy(idx).to(device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
    return t.cpu().numpy()


def take_global_frames(local: torch.Tensor, global_idx, comm) -> torch.Tensor:
    """Frames ``global_idx`` (numbered over the whole frame-sharded trajectory) of the array whose
    shard on this rank is ``local`` -- the same tensor, bit for bit, on every rank.

    Each rank fills in the frames it owns and the zero-filled buffers are summed (x + 0 is exact),
    so e.g. the constraint rows of the featurised fit are built from identical inputs everywhere
    and the replicated solve really is replicated.
    """
    import numpy as np

    idx = np.asarray(global_idx, dtype=np.int64).reshape(-1)
    group = resolve_comm(comm)
    first, total = shard_extent(local.shape[0], comm, local.device)
    if idx.size and (idx.min() < 0 or idx.max() >= total):
        raise IndexError(f"frame index outside 0..{total - 1}")
    if group is None:
        return local[torch.from_numpy(idx).to(local.device)]
    out = torch.zeros((idx.size,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    mine = np.nonzero((idx >= first) & (idx < first + local.shape[0]))[0]
    if mine.size:
        out[torch.from_numpy(mine).to(local.device)] = local[torch.from_numpy(idx[mine] - first).to(local.device)]
    return all_reduce_sum_(out, group)
