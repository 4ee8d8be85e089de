"""Tile-parallel sharding across GPUs and the exchange of the final detection sets.

The reference's only parallel axis is tiles (``multiprocessing.Pool().map`` over the 256-px
patches of an image, ``models/mpp/mpp_model.py:250-262``).  Here one process drives one GPU,
every rank owns a contiguous block of tiles (so that the part of the image it needs score maps for
is one compact region), chains run with no data-path collective, and the results (a few hundred
rectangles per tile) are combined with ONE all-gather of a fixed-capacity record buffer that is
filled on the device (``mpp_pack_detections``) -- RCCL over xGMI when the backend is "nccl", gloo on
CPU for the tests.  Scores of the gathered points are computed where their score maps live and
combined with one all-reduce of a vector that is zero everywhere but at a rank's own points.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

RECORD = 7           # tile id, x, y, size, ratio, angle, score  (include/mpp_hip.h: mpp_pack_detections)
PER_TILE_CAPACITY = 1024


def shard_tiles(n_tiles: int, rank: int, world_size: int) -> List[int]:
    """Tiles owned by ``rank``: the contiguous block [rank*n/world, (rank+1)*n/world) of the row-major tile list.
    Block sizes differ by at most one; with fewer tiles than ranks some ranks own nothing."""
    return list(range(rank * n_tiles // world_size, (rank + 1) * n_tiles // world_size))


def tile_owner(n_tiles: int, world_size: int) -> np.ndarray:
    """owner[t] = the rank whose ``shard_tiles`` block holds tile t"""
    owner = np.zeros(n_tiles, dtype=np.int64)
    for r in range(world_size):
        owner[shard_tiles(n_tiles, r, world_size)] = r
    return owner


def gather_capacity(n_tiles: int, world_size: int, per_tile: int = PER_TILE_CAPACITY) -> int:
    """Record capacity of one rank's gather buffer -- the same on every rank (``all_gather_into_tensor`` needs
    equal inputs) although ranks may own different numbers of tiles: the largest block times the per-tile capacity."""
    return per_tile * max(1, -(-n_tiles // world_size))


def init_process_group(backend: str = None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = os.environ.get("MPP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    # the group this module made is torn down when the interpreter exits: a process that ends with the backend's worker
    # threads still joinable aborts now and then ("terminate called without an active exception", exit code -6)
    import atexit
    atexit.register(_shutdown)
    return rank, world


def _shutdown():
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:              # noqa: BLE001 -- the run itself is over; nothing to report to
        pass


def pack_detections(tile_ids: Sequence[int], points: Sequence[Tuple[np.ndarray, np.ndarray]],
                    scores: Sequence[np.ndarray], capacity: int) -> np.ndarray:
    """Host twin of ``mpp_pack_detections``: [capacity+1, RECORD] float64, row 0 = (count, 0...), rows 1.. = records."""
    buf = np.zeros((capacity + 1, RECORD), dtype=np.float64)
    k = 0
    for tid, (xy, marks), sc in zip(tile_ids, points, scores):
        n = len(xy)
        if k + n > capacity:
            raise ValueError(f"detection buffer capacity {capacity} exceeded")
        buf[1 + k:1 + k + n, 0] = tid
        buf[1 + k:1 + k + n, 1:3] = xy
        buf[1 + k:1 + k + n, 3:6] = marks
        buf[1 + k:1 + k + n, 6] = sc if sc is not None else 0.0
        k += n
    buf[0, 0] = k
    return buf


class RankFailure(RuntimeError):
    """a rank reported a failure of its local phase in the gather (row 0, column 1 of its record buffer)"""


def unpack_detections(gathered: np.ndarray) -> np.ndarray:
    """[world, capacity+1, RECORD] -> [total, RECORD], ordered by (tile id, original order).
    Row 0 of a rank's buffer is (count, status, 0...): a non-zero status -- the rank's sampling or packing failed -- raises
    ``RankFailure`` on every rank alike, after the collective."""
    failed = [r for r, g in enumerate(gathered) if g[0, 1] != 0]
    if failed:
        raise RankFailure(f"the local phase failed on rank(s) {failed}")
    rows = [g[1:1 + int(g[0, 0])] for g in gathered]
    allr = np.concatenate(rows, axis=0) if rows else np.zeros((0, RECORD))
    return allr[np.argsort(allr[:, 0], kind="stable")]


def _backend_is_nccl() -> bool:
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "nccl"


def all_gather_detections(local, device=None) -> np.ndarray:
    """One all-gather of the fixed-capacity buffer; every rank gets every rank's detections.
    ``local``: the [capacity+1, RECORD] buffer, a device tensor (filled by ``MppContext.pack_detections``; sent as it
    is over RCCL, copied to the host only for gloo) or a NumPy array."""
    import torch
    import torch.distributed as dist
    is_tensor = hasattr(local, "data_ptr")
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_detections((local.cpu().numpy() if is_tensor else local)[None])
    t = local if is_tensor else torch.from_numpy(np.ascontiguousarray(local))
    if _backend_is_nccl():
        if not t.is_cuda:
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    elif t.is_cuda:
        t = t.cpu()
    t = t.contiguous()
    world = dist.get_world_size()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenated form
    dist.all_gather_into_tensor(out, t)
    return unpack_detections(out.reshape((world,) + tuple(t.shape)).cpu().numpy())


def all_reduce_owned(values: np.ndarray, device=None) -> np.ndarray:
    """Combine per-rank vectors that are zero everywhere except at the entries the rank owns (its points' scores):
    one all-reduce (sum); exact, since every entry has at most one non-zero contribution."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(values, dtype=np.float64)
    t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64))
    if _backend_is_nccl():
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
