"""Tile-parallel sharding across GPUs and the gather of the final detection sets.

The reference's only parallel axis is tiles (``multiprocessing.Pool().map`` over the 256-px
patches of an image, ``models/mpp/mpp_model.py:250-262``).  Here one process drives one GPU,
tiles are dealt round-robin to ranks, chains run with no data-path collective, and the results
(a few hundred rectangles per tile) are combined with ONE all-gather of a fixed-capacity record
buffer -- RCCL over xGMI when the backend is "nccl", gloo on CPU for the tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

RECORD = 7   # tile id, x, y, size, ratio, angle, score


def shard_tiles(n_tiles: int, rank: int, world_size: int) -> List[int]:
    """Tiles owned by ``rank``: t with t % world_size == rank."""
    return [t for t in range(n_tiles) if t % world_size == rank]


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
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def pack_detections(tile_ids: Sequence[int], points: Sequence[Tuple[np.ndarray, np.ndarray]],
                    scores: Sequence[np.ndarray], capacity: int) -> np.ndarray:
    """[capacity+1, RECORD] float64: row 0 = (count, 0...), rows 1.. = records."""
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


def unpack_detections(gathered: np.ndarray) -> np.ndarray:
    """[world, capacity+1, RECORD] -> [total, RECORD], ordered by (tile id, original order)."""
    rows = [g[1:1 + int(g[0, 0])] for g in gathered]
    allr = np.concatenate(rows, axis=0) if rows else np.zeros((0, RECORD))
    return allr[np.argsort(allr[:, 0], kind="stable")]


def all_gather_detections(local: np.ndarray, device=None) -> np.ndarray:
    """One all-gather of the fixed-capacity buffer; every rank gets every rank's detections."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_detections(local[None])
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenated form
    dist.all_gather_into_tensor(out, t)
    return unpack_detections(out.reshape((world,) + tuple(t.shape)).cpu().numpy())
