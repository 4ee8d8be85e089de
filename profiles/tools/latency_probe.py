"""How much of a step is memory latency?  Same object density, map footprint inside / outside the 4 MB L2."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth
setup, model = bench.load_model(); maps = mappings.default_mappings()
for T, nobj in ((96, 7), (128, 12), (256, 50), (512, 200)):
    for spec in (1, 8):
        t = synth.make_tile(T, nobj, 0)
        ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=spec)
        ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps); ctx.naive_init(setup.detection_threshold, 6.0)
        xy, mk = ctx.get_points(); ctx.set_kernels(kernels.make_kernels(maps, float(max(1, len(xy)))))
        ctx.set_schedule(1.0, 0.999, 0.0)
        ctx.run(20001, seed=0); ms0 = ctx.last_kernel_ms()
        ctx.run(100001, seed=1); ms = ctx.last_kernel_ms()
        print(f"tile {T} objects {nobj} spec {spec}: {100001 / ms:.1f} k proposals/s  ({ms * 1e3 / 100001:.2f} us/step), n={ctx.count(0) if hasattr(ctx,'count') else -1}", flush=True)
