"""BASELINE config 5 on ONE MI355X, end to end with the nets: a 4096x4096 scene of the reference's image recipe
(data/make_synth_data.py:16-47, ~5 000 rectangles) -> PosNet + ShapeNet + epilogues -> 256 chains in one launch ->
merge -> Papangelou scores.  Seeded random weights (no trained model.pt in the container), the posnet's 1x1 div_clf
calibrated so that the detection map fires.  Prints one JSON line with the wall-clock of each stage; `world` > 1 times the
stages of ONE rank of a multi-GPU run (its region only) without the collectives."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_configs import calibrate_div_clf, make_model, random_nets
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {"image": size}
img, gt_xy, gt_marks = synth.make_scene_image((size, size), int(5250 * (size / 4096) ** 2), noise=0.02, seed=5)
out["objects_in_image"] = int(len(gt_xy))
for dtype in (torch.float32, torch.bfloat16):
    nets = random_nets(dtype=dtype)
    calibrate_div_clf(nets, img[:1024, :1024])
    for cfg in ("mpp_hrcM.json", "config_mpp_log.json"):
        mpp = make_model(cfg, nets=nets)
        data = ImageWMaps(name="0005", shape=(size, size), image=img, detection_map=None, param_dist_maps=None,
                          mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
        for world in (1, 8):
            best = None
            for rep in range(3):
                mpp.rng = np.random.default_rng(0)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                region = mpp.region_maps(data, 0, world)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                if world == 1:
                    pts, scores = mpp.infer_image(data, region_data=region)
                    torch.cuda.synchronize(); t2 = time.perf_counter()
                    r = {"nets_s": t1 - t0, "sample_merge_score_s": t2 - t1, "total_s": t2 - t0, "kernel_ms": mpp.last_run["kernel_ms"],
                         "detections": len(pts), "tiles": len(mpp.last_run["anchors"]), "steps_per_chain": mpp.last_run["total_steps"]}
                else:
                    r = {"nets_s_rank0_region": t1 - t0, "region": [int(v) for v in region.shape]}
                key = "total_s" if world == 1 else "nets_s_rank0_region"
                if best is None or r[key] < best[key]:
                    best = r
            out[f"{cfg.split('.')[0]}_{str(dtype).split('.')[-1]}_world{world}"] = best
print(json.dumps(out))
