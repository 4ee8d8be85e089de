"""debug: tile 0 of config 3 (seed 20261004) GPU vs oracle -- which instantiation differs, and where"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle
from helpers import hrc_model
from mpp_cnn_rs_object_detection_amd import energies as E, hip_api, kernels, mappings, synth

setup, comb = hrc_model()
unit, pair = setup.make_energies()
model = E.build_model_desc(unit, pair, comb)
maps = mappings.default_mappings()
t = synth.make_tile(512, 200, tile_id=0)
steps, seed = 100001, 20261004
o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
xy0, mk0 = o.naive_detection(setup.detection_threshold, 6.0)
kd = kernels.make_kernels(maps, float(max(1, len(xy0))))
o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kd)
o.set_points(xy0, mk0); o.set_temperature(1.0, 0.999, 0.0)
oout, oprops = o.run(steps, seed, chain=0, trace=True)
oxy, om = o.get_points()

def gpu(spec, trace):
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=spec)
    ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps)
    ctx.set_points(0, xy0, mk0); ctx.set_kernels(kd); ctx.set_schedule(1.0, 0.999, 0.0)
    res = ctx.run(steps, seed=seed, chain0=0, trace_tile=0 if trace else -1)
    xy, m = ctx.get_points(0)
    ctx.close()
    return res, xy, m

for spec in (8, 1):
    for trace in (False, True):
        res, xy, m = gpu(spec, trace)
        same = xy.shape == oxy.shape and np.array_equal(xy, oxy) and np.allclose(m, om, rtol=1e-9, atol=1e-9)
        print(f"spec {spec} trace {trace}: final == oracle: {same}  n={len(xy)} vs {len(oxy)}")
        if not same and xy.shape == oxy.shape:
            bad = np.argwhere(~np.isclose(m, om, rtol=1e-9, atol=1e-9))
            print("   differing entries", bad.tolist(), m[bad[:, 0]].tolist(), om[bad[:, 0]].tolist(), xy[bad[:, 0]].tolist())
        if trace:
            gout, gprops = res
            d = np.nonzero((gout["accepted"] != oout["accepted"]) | ~np.isclose(gout["dE"], oout["dE"], rtol=1e-9, atol=1e-9))[0]
            pd = [f for f in gprops.dtype.names if not np.array_equal(gprops[f], oprops[f])]
            print("   first differing step (accept/dE):", d[:5].tolist(), " differing proposal fields:", pd)
            for f in pd:
                k = np.nonzero(gprops[f] != oprops[f])[0]
                print("     field", f, "first diffs at", k[:5].tolist(), gprops[f][k[:3]].tolist(), oprops[f][k[:3]].tolist())
            if len(d):
                s = int(d[0])
                print("   gpu  ", gout[s], gprops[s]); print("   orac ", oout[s], oprops[s])
