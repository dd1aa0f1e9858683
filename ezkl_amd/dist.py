"""Multi-GPU decomposition of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

MSM shards by POINTS: rank r owns the contiguous slice [r*n/G, (r+1)*n/G) of the scalars and of the
resident base set, runs the full Pippenger pipeline on it and produces one 64-byte affine partial.  The
only exchange is an all_gather of those partials (G x 64 B) followed by a fold with the group law on the
host -- RCCL has no elliptic-curve reduction op, so this is gather-then-add, not all_reduce(sum).
NTT shards by COLUMNS (independent polynomials): no collective at all."""
import numpy as np
from . import backend as _b


def shard_range(n, rank, world):
    """contiguous [lo, hi) slice of n items owned by `rank`; remainders go to the low ranks"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_columns(ncols, rank, world):
    """round-robin column ownership for batched NTTs"""
    return list(range(rank, ncols, world))


def fold_points(points):
    """sum of affine partials (k, 8) u64 with the group law (host side of the C ABI)"""
    acc = np.zeros(8, np.uint64)
    for p in points:
        acc = _b.g1_add_affine(acc, p)
    return acc


_bufs = {}


def fold_partials(partial, dist, device):
    """all_gather the per-rank 64-byte partial sums and fold them; every rank returns the full sum.
    Send/receive tensors are allocated once per (device, world size) and reused every step."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return partial
    import torch
    world = dist.get_world_size()
    key = (str(device), world)
    if key not in _bufs:
        _bufs[key] = (torch.empty(64, dtype=torch.uint8, device=device), torch.empty(world * 64, dtype=torch.uint8, device=device))
    send, recv = _bufs[key]
    send.copy_(torch.from_numpy(np.ascontiguousarray(partial).view(np.uint8)))
    dist.all_gather_into_tensor(recv, send)
    parts = recv.cpu().numpy().view(np.uint64).reshape(-1, 8)
    return fold_points(parts)


def allgather_points(partials, dist, device):
    """all_gather a (b, 8) array of per-rank partial points -> (world, b, 8); one collective per commit batch"""
    partials = np.ascontiguousarray(partials, np.uint64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return partials[None]
    import torch
    world = dist.get_world_size()
    send = torch.from_numpy(partials.view(np.uint8).reshape(-1).copy()).to(device)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    return recv.cpu().numpy().view(np.uint64).reshape(world, partials.shape[0], 8)


def fold_columns(gathered):
    """(world, b, 8) partials -> (b, 8) sums"""
    out = np.zeros((gathered.shape[1], 8), np.uint64)
    for j in range(gathered.shape[1]):
        out[j] = fold_points(gathered[:, j])
    return out
