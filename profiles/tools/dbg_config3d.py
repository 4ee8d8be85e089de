"""debug: is the traced chain of tile 0 / config 3 independent of how it is cut into launches, and of spec_waves?"""
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
seed = 20261004
o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
xy0, mk0 = o.naive_detection(setup.detection_threshold, 6.0)
kd = kernels.make_kernels(maps, float(max(1, len(xy0))))

def trace(chunks, spec, init="points"):
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=spec)
    ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps)
    if init == "points":
        ctx.set_points(0, xy0, mk0)
    else:
        ctx.naive_init(setup.detection_threshold, 6.0)
    ctx.set_kernels(kd); ctx.set_schedule(1.0, 0.999, 0.0)
    outs, props = [], []
    for n in chunks:
        g, p = ctx.run(n, seed=seed, chain0=0, trace_tile=0)
        outs.append(g); props.append(p)
    return np.concatenate(outs), np.concatenate(props)

A = trace([20000, 20000], 8)
for name, B in (("8 waves, chunks 20000+17810+2190", trace([20000, 17810, 2190], 8)), ("8 waves again, 20000+20000", trace([20000, 20000], 8)),
                ("1 wave, one launch", trace([40000], 1)), ("8 waves, naive_init on the GPU", trace([20000, 20000], 8, "naive")),
                ("8 waves, 40 x 1000", trace([1000] * 40, 8))):
    same_o = all(np.array_equal(A[0][f], B[0][f]) for f in A[0].dtype.names)
    same_p = all(np.array_equal(A[1][f], B[1][f]) for f in A[1].dtype.names)
    print(name, ": step records equal", same_o, " proposals equal", same_p)
    if not same_o:
        for f in A[0].dtype.names:
            k = np.nonzero(A[0][f] != B[0][f])[0]
            if len(k):
                print("    ", f, "first diffs at", k[:5].tolist(), A[0][f][k[:3]].tolist(), B[0][f][k[:3]].tolist())
print("A at 37803:", A[0][37803], A[1][37803])
