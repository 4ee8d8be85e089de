"""Throughput of the two callers'-side kernels added after the chain: mpp_quad_iou (DOTA task-1 evaluation) and
mpp_delta_vectors (weight learning).  Run on the GPU box: python profiles/tools/bench_eval.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import torch

import bench
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

out = {}
ctx = hip_api.MppContext(0)
rng = np.random.default_rng(0)


def quads(n, extent):
    c = rng.uniform(0, extent, (n, 2)); L = rng.uniform(6, 14, n); W = rng.uniform(3, 6, n); a = rng.uniform(0, np.pi, n)
    q = np.zeros((n, 8))
    for k, (sx, sy) in enumerate(((1, 1), (-1, 1), (-1, -1), (1, -1))):
        vx, vy = sx * L / 2, sy * W / 2
        q[:, 2 * k] = c[:, 0] + np.cos(a) * vx - np.sin(a) * vy
        q[:, 2 * k + 1] = c[:, 1] + np.sin(a) * vx + np.cos(a) * vy
    return q


for n, extent in ((4096, 300.0), (8192, 4096.0)):
    a = torch.tensor(quads(n, extent), device="cuda:0"); b = torch.tensor(quads(n, extent), device="cuda:0")
    o = torch.zeros((n, n), dtype=torch.float64, device="cuda:0")
    L = hip_api.load_library()
    call = lambda: L.mpp_quad_iou(ctx._h, n, a.data_ptr(), n, b.data_ptr(), o.data_ptr(), 1)
    call(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        call()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
    frac_clipped = float((o > 0).double().mean())
    out[f"quad_iou_{n}x{n}_extent{int(extent)}"] = {"pairs_per_s": n * n / dt, "ms": dt * 1e3, "pairs_with_overlap": frac_clipped,
                                               "out_GBps": n * n * 8 / dt / 1e9}

# delta_vectors: 256 aggregated perturbations of a 200-object tile
setup, model = bench.load_model(); maps = mappings.default_mappings()
from mpp_cnn_rs_object_detection_amd.perturbation_sampler import sample_multiple_kernel_perturbations
from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
t = synth.make_tile(512, 200, 0)
gt = [Rectangle(int(x), int(y), size=float(m[0]), ratio=float(m[1]), angle=float(m[2])) for (x, y), m in zip(t.gt_xy, t.gt_marks)]
data = ImageWMaps(name="0", shape=t.shape, image=None, detection_map=t.det, param_dist_maps=t.marks, mappings=maps,
                  param_names=Rectangle.PARAMETERS, gt_config=gt)
unit, pair = setup.make_energies(data)
base = EPointsSet(gt, data.shape, unit, pair, image_data=data); data.gt_config_set = base
t0 = time.perf_counter()
perts = sample_multiple_kernel_perturbations(data, n_samples=256, rng=np.random.default_rng(0), energy_setup=setup,
                                             iter_per_point=1.0, return_perturbations=True, aggregate_pert=True)
t_walk = time.perf_counter() - t0
base.energy_delta_vectors(perts[:4])
t0 = time.perf_counter(); before, after, mask = base.energy_delta_vectors(perts); t_vec = time.perf_counter() - t0
t0 = time.perf_counter(); d = base.energy_delta_batch(perts); t_d = time.perf_counter() - t0
out["learning_512_200obj_256perts"] = {"kernel_walks_s": t_walk, "delta_vectors_s": t_vec, "delta_batch_s": t_d,
                                       "rows_touched": int((mask > 0).sum()), "mean_dE": float(np.mean(d))}
print(json.dumps(out, indent=1))
