"""A whole scene on one GPU (BASELINE configs 4/5 in spirit, without the multi-GPU part): a 2048 x 2048 and a
4096 x 4096 synthetic scene with one object per 14-px lattice cell picked at random (1250 / 5000 objects), score maps
rendered on the host and handed to MPPModel.infer_image as device tensors: 256-px tiles, ALL tiles in one launch of the
chain kernel, merge, Papangelou scores.  Reports wall-clock per stage and the recovered fraction."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

out = {}
for cfg_name in ("mpp_hrcM.json", "config_mpp_log.json"):
    cfg = json.load(open(os.path.join("model_configs", "mpp", cfg_name)))
    model = MPPModel(cfg, phase="val", load=True)
    for size, n_obj in ((2048, 1250), (4096, 5000)):
        gt_xy, gt_marks = synth.make_gt(size, n_obj, tile_id=500)
        t0 = time.perf_counter()
        det, marks = synth.render_maps((size, size), gt_xy, gt_marks)
        t_render = time.perf_counter() - t0
        data = ImageWMaps(name="0001", shape=(size, size), image=None, detection_map=torch.from_numpy(det).cuda(),
                          param_dist_maps=[torch.from_numpy(m).cuda() for m in marks], mappings=mappings.default_mappings(),
                          param_names=Rectangle.PARAMETERS, gt_config=[])
        torch.cuda.synchronize()
        model.rng = np.random.default_rng(0)
        t0 = time.perf_counter()
        pts, scores = model.infer_image(data)
        torch.cuda.synchronize()
        t_infer = time.perf_counter() - t0
        xy = np.array([[p.x, p.y] for p in pts], dtype=float).reshape(-1, 2)
        d = np.sqrt(((xy[:, None, :] - gt_xy[None]) ** 2).sum(-1))
        n_tiles = (int(np.ceil(size / 256))) ** 2
        steps = cfg["inference"]["rjmcmc_params"]["burn_in"] + 2 * cfg["inference"]["rjmcmc_params"]["samples_interval"] + 1
        out[f"{cfg['model_name']}_{size}"] = {
            "objects": int(len(gt_xy)), "detections": int(len(pts)), "matched_within_2px": int((d.min(axis=0) <= 2).sum()),
            "tiles": n_tiles, "steps_per_tile": steps, "infer_image_s": t_infer,
            "proposals_per_s_end_to_end": n_tiles * steps / t_infer, "host_render_s": t_render,
        }
        print(json.dumps({f"{cfg['model_name']}_{size}": out[f"{cfg['model_name']}_{size}"]}), flush=True)
json.dump(out, open(os.path.join("gpurun_out", "bench_scene.json"), "w"), indent=1)
