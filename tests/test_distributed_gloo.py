"""The N>1 path on CPU: two processes over gloo (127.0.0.1) shard tiles and all-gather their detections."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from helpers import REPO

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, os.environ["MPP_REPO"])
    from mpp_cnn_rs_object_detection_amd import distributed as mdist
    rank, world = mdist.init_process_group(backend="gloo")
    n_tiles = 5
    mine = mdist.shard_tiles(n_tiles, rank, world)
    rng = np.random.default_rng(100)
    all_pts = []
    for t in range(n_tiles):                      # every rank can rebuild every tile's (fake) detections
        n = 3 + t
        all_pts.append((rng.integers(0, 256, size=(n, 2)).astype(float), rng.random((n, 3)), rng.random(n)))
    buf = mdist.pack_detections(mine, [all_pts[t][:2] for t in mine], [all_pts[t][2] for t in mine], capacity=64)
    rec = mdist.all_gather_detections(buf)
    expect = np.concatenate([np.concatenate([np.full((len(p[0]), 1), t), p[0], p[1], p[2][:, None]], axis=1)
                             for t, p in enumerate(all_pts)])
    ok = rec.shape == expect.shape and np.array_equal(rec, expect)
    # blocks are contiguous and of different sizes (5 tiles on 2 ranks: 2 + 3), the buffer capacity is not
    ok = ok and mine == ([0, 1] if rank == 0 else [2, 3, 4]) and mdist.gather_capacity(n_tiles, world, per_tile=16) == 48
    # scores: every rank contributes the entries of its own tiles' points, one all-reduce combines them exactly
    owner = mdist.tile_owner(n_tiles, world)[rec[:, 0].astype(int)]
    scores = np.where(owner == rank, rec[:, 6], 0.0)
    ok = ok and np.array_equal(mdist.all_reduce_owned(scores), rec[:, 6])
    # a device-style tensor buffer goes through the same gather
    import torch
    rec2 = mdist.all_gather_detections(torch.from_numpy(buf))
    ok = ok and np.array_equal(rec2, expect)
    # a rank whose local phase failed says so in row 0 of its buffer: EVERY rank raises after the collective, none waits
    bad = buf.copy()
    if rank == 1:
        bad[:] = 0.0
        bad[0, 1] = 1.0
    try:
        mdist.all_gather_detections(bad)
        ok = False
    except mdist.RankFailure as e:
        ok = ok and "[1]" in str(e)
    print(json.dumps({"rank": rank, "world": world, "mine": mine, "ok": bool(ok), "n": int(len(rec))}))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_shard_tiles_and_gather_detections(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MPP_REPO=REPO)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    import json
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    res = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    assert all(r["ok"] and r["world"] == 2 and r["n"] == sum(3 + t for t in range(5)) for r in res)
    assert sorted(res[0]["mine"] + res[1]["mine"]) == [0, 1, 2, 3, 4]
