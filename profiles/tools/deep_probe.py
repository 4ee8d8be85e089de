#!/usr/bin/env python3
"""Deep rounds against one-wave-per-step rounds on the bench workload (one 512x512 tile, 100 001 steps) and on many
256-px tiles: kernel time, rounds, committed steps per round; the final configurations must be identical."""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth


def run(tile, objects, iters, spec, deep, fixed, reps, ntiles=1, cap=1024, replicas=1, seed=0):
    setup, model = bench.load_model()
    maps = mappings.default_mappings()
    tiles = [synth.make_tile(tile, objects, tile_id=i) for i in range(ntiles)]
    ctx = hip_api.MppContext(0, point_capacity=cap, spec_waves=spec, replicas=replicas)
    ctx.set_option("deep", deep)
    ctx.set_option("deep_fixed", fixed)
    if os.environ.get("DEEP_GAIN"):
        ctx.set_option("deep_gain", int(os.environ["DEEP_GAIN"]))
    ctx.set_maps(np.stack([t.det for t in tiles]), [np.stack([t.marks[k] for t in tiles]) for k in range(3)])
    ctx.set_model(model, maps)
    ctx.naive_init(setup.detection_threshold, 6.0)
    n_chains = ctx.get_option("n_chains")
    inten = np.maximum(1, ctx.counts()[:n_chains]).astype(np.float64)
    ctx.set_kernels(kernels.make_kernels(maps, 1.0), intensity=inten)
    ms = []
    for r in range(reps):
        ctx.naive_init(setup.detection_threshold, 6.0)
        ctx.set_schedule(1.0, 0.999, 0.0)
        ctx.run(iters, seed=seed)
        ms.append(ctx.last_kernel_ms())
    st = ctx.deep_stats() if deep else {}
    if deep and os.environ.get("MPP_LIB_PATH", "").endswith("dprof.so"):
        names = ["A:types", "A:bar1", "A:sort+bar2", "B:draw", "B:pre", "E:delta(total)", "B:post", "bar3", "C:decide+trace+ring", "D:apply-delta",
                 "D:bar4", "D:mutate", "e:tasks", "e:append", "e:load", "e:overlap", "e:pairs+rescan", "e:combine"]
        R = max(1, st["rounds"])
        st["phase_cycles_per_round"] = {}
        for w in range(spec):
            ph = [ctx.get_option(f"deep_stat{16 + 24 * w + i}") for i in range(len(names))]
            st["phase_cycles_per_round"][f"wave{w}"] = {k: round(v / R) for k, v in zip(names, ph)}
    pts = ctx.get_points(0)
    return {"spec": spec, "deep": deep, "fixed": fixed, "kernel_ms": min(ms), "proposals_per_s": n_chains * iters / (min(ms) * 1e-3),
            "stats": st, "committed_per_round": (st["committed"] / st["rounds"]) if st.get("rounds") else None,
            "n_final": int(len(pts[0]))}, pts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100001)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--objects", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--configs", default="8:0:0,8:512:0,8:256:0,8:128:0,8:64:0,8:512:128,8:512:256")
    ap.add_argument("--ntiles", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=1)
    ap.add_argument("--cap", type=int, default=1024)
    a = ap.parse_args()
    ref = None
    for cfg in a.configs.split(","):
        spec, deep, fixed = (int(v) for v in cfg.split(":"))
        res, pts = run(a.tile, a.objects, a.iters, spec, deep, fixed, a.reps, a.ntiles, a.cap, a.replicas)
        if ref is None:
            ref = pts
        res["same_as_first"] = bool(ref[0].tobytes() == pts[0].tobytes() and ref[1].tobytes() == pts[1].tobytes())
        print(json.dumps(res), flush=True)
