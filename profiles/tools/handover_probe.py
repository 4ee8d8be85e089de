import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth
setup, model = bench.load_model()
maps = mappings.default_mappings()
t = synth.make_tile(512, 200, tile_id=0)
ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps)
ctx.naive_init(setup.detection_threshold, 6.0)
inten = np.maximum(1, ctx.counts()[:1]).astype(np.float64)
ctx.set_kernels(kernels.make_kernels(maps, 1.0), intensity=inten)
ref = None
for K in (0, 1000, 2000, 3500, 5000, 8000, 100001):
    best = 1e9
    for rep in range(3):
        ctx.naive_init(setup.detection_threshold, 6.0)
        ctx.set_schedule(1.0, 0.999, 0.0)
        ms = 0.0
        if K > 0:
            ctx.set_option("deep", 0); ctx.run(min(K, 100001), seed=0); ms += ctx.last_kernel_ms()
        if K < 100001:
            ctx.set_option("deep", 128); ctx.run(100001 - K, seed=0); ms += ctx.last_kernel_ms()
        best = min(best, ms)
    pts = ctx.get_points(0)
    if ref is None: ref = pts
    print(json.dumps({"wave_kernel_steps": K, "kernel_ms": round(best, 2), "same": bool(ref[0].tobytes() == pts[0].tobytes() and ref[1].tobytes() == pts[1].tobytes())}), flush=True)
