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


# ---- columns -> row windows: the one exchange of the row-sharded quotient sweep -------------------
# After NTTs sharded by columns, rank r holds whole extended columns c with c % world == r.  The sweep sharded by ROWS
# needs, on every rank, rows [lo, hi) of every (column, rotation) pair its gate program reads: one all_to_all in which
# the owner of a column sends each destination the window that destination asked for.  Windows wrap around the domain,
# so a large rotation (the permutation argument reads z at omega^usable) is just another window, not a halo.
def row_windows(queries, n_ext, rank, world):
    """queries: list of (column, row_shift).  Returns [(column, start, length)]: the rows of `column` this rank needs,
    start taken modulo n_ext"""
    lo, hi = shard_range(n_ext, rank, world)
    return [(c, (lo + s) % n_ext, hi - lo) for c, s in queries]


def _window(col, start, length):
    """rows [start, start + length) of a column, wrapping around"""
    n = col.shape[0]
    if start + length <= n:
        return col[start:start + length]
    return np.concatenate([col[start:], col[: start + length - n]])


def reshard_columns_to_rows(owned, queries, n_ext, dist, device, owner=None):
    """owned: {column: (n_ext, 4) u64 array} for the columns this rank owns (shard_columns).  queries: the (column,
    row_shift) pairs of the gate program, the same list on every rank.  Returns {(column, row_shift): (rows, 4) array}
    with this rank's row window of every queried pair.  One all_to_all (byte tensors; RCCL on the GPU box, gloo on CPU)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {(c, s): _window(owned[c], s % n_ext, n_ext) for c, s in queries}
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if owner is None:
        owner = lambda c: c % world                          # round-robin column ownership (shard_columns)
    send, send_sizes = [], []
    for d in range(world):                                   # what destination d needs from the columns I own, in query order
        parts = [_window(owned[c], start, ln) for (c, start, ln) in row_windows(queries, n_ext, d, world) if owner(c) == rank]
        blob = np.concatenate(parts).view(np.uint8).reshape(-1) if parts else np.zeros(0, np.uint8)
        send.append(blob)
        send_sizes.append(blob.size)
    mine = row_windows(queries, n_ext, rank, world)
    recv_sizes = [sum(ln * 32 for (c, _, ln) in mine if owner(c) == s) for s in range(world)]
    sbuf = torch.from_numpy(np.concatenate(send) if sum(send_sizes) else np.zeros(0, np.uint8)).to(device)
    rbuf = torch.empty(sum(recv_sizes), dtype=torch.uint8, device=device)
    dist.all_to_all_single(rbuf, sbuf, output_split_sizes=recv_sizes, input_split_sizes=send_sizes)
    flat = rbuf.cpu().numpy()
    out, off = {}, [sum(recv_sizes[:s]) for s in range(world)]
    for (c, s_), (_, _, ln) in zip(queries, mine):           # unpack in the same per-source query order the senders used
        src = owner(c)
        out[(c, s_)] = flat[off[src]:off[src] + ln * 32].view(np.uint64).reshape(ln, 4).copy()
        off[src] += ln * 32
    return out


def allgather_array(a, dist, device):
    """all_gather a numpy array of the same shape on every rank -> list of world arrays"""
    import torch
    world = dist.get_world_size()
    a = np.ascontiguousarray(a)
    send = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    flat = recv.cpu().numpy()
    return [flat[r * a.nbytes:(r + 1) * a.nbytes].view(a.dtype).reshape(a.shape).copy() for r in range(world)]


def allgather_ints(vals, dist, device):
    """all_gather a list of field elements (ints < 2^256) -> [world][len(vals)]"""
    a = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), np.uint8).copy() if vals else np.zeros(0, np.uint8)
    parts = allgather_array(a, dist, device)
    return [[int.from_bytes(p[32 * i:32 * i + 32].tobytes(), "little") for i in range(len(vals))] for p in parts]


def allgather_rows(buf, lo, hi, n, dist, device):
    """buf: a resident column (backend.DeviceBuffer) of n rows of which this rank computed [lo, hi); afterwards every rank
    holds all rows.  Equal shards (n divisible by the world size): one all_gather.  Host-staged here; a GPU-direct version would
    hand RCCL the device pointers."""
    import torch
    world = dist.get_world_size()
    assert n % world == 0 and hi - lo == n // world
    rows = hi - lo
    mine = torch.from_numpy(_b.memcpy_d2h(buf.ptr + 32 * lo, 32 * rows)).to(device)
    recv = torch.empty(world * mine.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, mine)
    full = recv.cpu().numpy().view(np.uint64).reshape(n, 4)
    for r in range(world):
        if r * rows != lo:
            _b.memcpy_h2d(buf.ptr + 32 * r * rows, np.ascontiguousarray(full[r * rows:(r + 1) * rows]))


class DeviceBytes:
    """a library-owned device buffer seen through __cuda_array_interface__, so that torch (and through it RCCL) can work on it in
    place: torch.as_tensor(DeviceBytes(ptr, n), device=...) aliases the memory, it does not copy"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


_direct_gather_ok = [True]


def allgather_device_rows(ptr, total, off, nbytes, dist, device):
    """In-place all_gather on a device buffer of `total` bytes of which this rank wrote [off, off + nbytes) (equal slices, rank
    order).  On the GPU box RCCL works on the device pointers directly (in-place all_gather: the input is the rank's slice of the
    output); with gloo -- or if torch refuses the pointer -- the slices are staged through the host."""
    import os
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    assert total == world * nbytes and off == rank * nbytes
    if getattr(device, "type", "cpu") == "cuda" and _direct_gather_ok[0] and not os.environ.get("EZKL_GATHER_HOST"):
        try:
            full = torch.as_tensor(DeviceBytes(ptr, total), device=device)
        except Exception:                                    # decided before any collective is entered: the same on every rank
            _direct_gather_ok[0] = False
        else:
            dist.all_gather_into_tensor(full, full[off:off + nbytes])
            torch.cuda.synchronize(device)
            return
    mine = torch.from_numpy(_b.memcpy_d2h(ptr + off, nbytes)).to(device)
    recv = torch.empty(total, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, mine)
    full = recv.cpu().numpy()
    for r in range(world):
        if r != rank:
            _b.memcpy_h2d(ptr + r * nbytes, np.ascontiguousarray(full[r * nbytes:(r + 1) * nbytes]))


def probe_direct_gather(dist, device):
    """Called once by every rank before the first sharded sweep: an in-place all_gather on a small library-owned buffer through
    DeviceBytes, checked on every rank, and the ranks agree (all_reduce MIN) on whether the direct path is usable; otherwise the
    slices are staged through the host.  gloo / CPU: nothing to probe."""
    import torch
    if getattr(device, "type", "cpu") != "cuda":
        _direct_gather_ok[0] = False
        return False
    world, rank = dist.get_world_size(), dist.get_rank()
    def agree(ok):                                           # every rank takes part in the same collectives whatever happened locally
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag[0]))
    buf, full = None, None
    try:
        buf = _b.DeviceBuffer(world * 256)
        _b.memcpy_h2d(buf.ptr + 256 * rank, np.full(256, rank + 1, np.uint8))
        full = torch.as_tensor(DeviceBytes(buf.ptr, world * 256), device=device)
    except Exception:
        full = None
    if not agree(full is not None):                          # a rank that cannot alias the pointer must not leave the others in the all_gather
        _direct_gather_ok[0] = False
        return False
    ok = True
    try:
        dist.all_gather_into_tensor(full, full[256 * rank:256 * (rank + 1)])
        torch.cuda.synchronize(device)
        got = _b.memcpy_d2h(buf.ptr, world * 256).reshape(world, 256)
        ok = all((got[r] == r + 1).all() for r in range(world))
    except Exception:
        ok = False
    _direct_gather_ok[0] = agree(ok)
    return _direct_gather_ok[0]


# ---- the callbacks of the column-sharded C++ prover (include/ezkl_prover.h ezkl_prover_cs_set_shard_exchange), torch.distributed
#      versions: what a prover uses when the library communicator is not available (gloo; two ranks sharing one GPU in the tests) ----
def allgather_host_bytes(ptr, per, dist, device):
    """in place on a HOST buffer of world * per bytes: rank r's slice is valid going in, every slice coming out"""
    import ctypes
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    buf = (ctypes.c_uint8 * (world * per)).from_address(ptr)
    arr = np.frombuffer(buf, np.uint8)
    mine = torch.from_numpy(arr[rank * per:(rank + 1) * per].copy()).to(device)
    recv = torch.empty(world * per, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, mine)
    arr[:] = recv.cpu().numpy()


def exchange_segments(sends, recvs, dist, device):
    """sends / recvs: lists of (peer, device pointer, bytes) in the matching order of ezkl_hip_comm_alltoallv_dev.  Host-staged: the
    segments for one peer are concatenated into one buffer, the ranks all_gather how much each sends to each, the buffers travel as
    point-to-point sends / receives (one per ordered pair of ranks, all posted at once), and the received bytes are copied back to
    the device segment by segment."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out_by_peer = [[] for _ in range(world)]
    for peer, ptr, nbytes in sends:
        out_by_peer[peer].append(_b.memcpy_d2h(ptr, nbytes) if nbytes else np.zeros(0, np.uint8))
    flat = [np.concatenate(x) if x else np.zeros(0, np.uint8) for x in out_by_peer]
    lens = torch.tensor([len(f) for f in flat], dtype=torch.int64, device=device)
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    all_lens = torch.stack(all_lens).cpu().numpy()                  # [src][dst]
    send_t = {d: torch.from_numpy(flat[d]).to(device) for d in range(world) if d != rank and len(flat[d])}
    recv_t = {s_: torch.empty(int(all_lens[s_][rank]), dtype=torch.uint8, device=device) for s_ in range(world) if s_ != rank and all_lens[s_][rank]}
    ops = [dist.P2POp(dist.isend, t, d) for d, t in send_t.items()] + [dist.P2POp(dist.irecv, t, s_) for s_, t in recv_t.items()]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    got = [recv_t[s_].cpu().numpy() if s_ in recv_t else (flat[rank] if s_ == rank else np.zeros(0, np.uint8)) for s_ in range(world)]
    cursor = [0] * world
    for peer, ptr, nbytes in recvs:
        if nbytes:
            _b.memcpy_h2d(ptr, np.ascontiguousarray(got[peer][cursor[peer]:cursor[peer] + nbytes]))
        cursor[peer] += nbytes
    for peer in range(world):
        assert cursor[peer] == len(got[peer]), "exchange: segment lists of ranks %d and %d do not match" % (peer, rank)


def selftest(dist, device):
    """The torch.distributed twin of ezkl_hip_comm_selftest (csrc/comm.hip), for provers whose exchanges go through the callbacks above
    (gloo; ranks sharing one GPU): every rank calls it once after init_process_group, before anything depends on the group --
      1. all_gather of the rank ids through allgather_host_bytes;
      2. ONE all-to-all of odd-sized segments (17 + 13 r + 7 p bytes from r to p, two segments per peer, cut differently on the two sides)
         through exchange_segments;
      3. the fold of partial points: rank r contributes (r + 1) G, the fold must be (world (world + 1) / 2) G (fold_partials).
    Raises RuntimeError naming the step on the rank that saw wrong data.  A missing peer is the process group's own timeout
    (init_process_group(timeout=...)): torch raises instead of hanging."""
    import ctypes
    world, rank = dist.get_world_size(), dist.get_rank()

    def fail(what):
        raise RuntimeError("rank %d of %d: process-group self-test FAILED at: %s" % (rank, world, what))
    per = 16
    buf = (ctypes.c_uint8 * (world * per))(*([0xff] * (world * per)))
    for i in range(per):
        buf[rank * per + i] = rank
    allgather_host_bytes(ctypes.addressof(buf), per, dist, device)
    if any(buf[r * per + i] != r for r in range(world) for i in range(per)):
        fail("all_gather of rank ids")
    seg_len = lambda src, dst: 17 + 13 * src + 7 * dst
    pat = lambda src, dst, n: ((31 * src + 17 * dst + 3 + 7 * np.arange(n)) % 256).astype(np.uint8)
    keep, sends, recvs = [], [], []
    for p in range(world):
        ls, lr = seg_len(rank, p), seg_len(p, rank)
        s_buf, r_buf = _b.DeviceBuffer(ls), _b.DeviceBuffer(lr)
        keep += [s_buf, r_buf]
        _b.memcpy_h2d(s_buf.ptr, pat(rank, p, ls))
        _b.memcpy_h2d(r_buf.ptr, np.zeros(lr, np.uint8))
        sends += [(p, s_buf.ptr, 5), (p, s_buf.ptr + 5, ls - 5)]
        recvs += [(p, r_buf.ptr, lr - 9), (p, r_buf.ptr + (lr - 9), 9)]
    exchange_segments(sends, recvs, dist, device)
    for p in range(world):
        lr = seg_len(p, rank)
        if not (_b.memcpy_d2h(keep[2 * p + 1].ptr, lr) == pat(p, rank, lr)).all():
            fail("all-to-all of odd-sized segments")
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    mont = lambda v: np.frombuffer((v * (1 << 256) % q).to_bytes(32, "little"), np.uint64)
    G = np.concatenate([mont(1), mont(2)])
    def times(m):
        acc = np.zeros(8, np.uint64)
        for _ in range(m):
            acc = _b.g1_add_affine(acc, G)
        return acc
    if not (fold_partials(times(rank + 1), dist, device) == times(world * (world + 1) // 2)).all():
        fail("fold of partial points")
    return True
