"""Does building the remap tables pay in a many-tiles launch?  config 5 (256 tiles of 256 px, 30 257 steps) end to end with the
tables on (auto) and off, every repetition printed: python profiles/tools/probe_remap_cost.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_configs import calibrate_div_clf, make_model, random_nets
from mpp_cnn_rs_object_detection_amd import hip_api, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

size = 4096
img, gt_xy, gt_marks = synth.make_scene_image((size, size), 5250, noise=0.02, seed=5)
nets = random_nets(dtype=torch.float32)
calibrate_div_clf(nets, img[:1024, :1024])
mpp = make_model("mpp_hrcM.json", nets=nets)
data = ImageWMaps(name="0005", shape=(size, size), image=img, detection_map=None, param_dist_maps=None,
                  mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
region = mpp.region_maps(data, 0, 1)
orig = hip_api.MppContext.set_maps
out = {}
for mode in (-1, 0, -1, 0):
    def set_maps(self, det, marks, _m=mode):
        self.set_option("remap_table", _m)
        return orig(self, det, marks)
    hip_api.MppContext.set_maps = set_maps
    ts = []
    for rep in range(4):
        mpp.rng = np.random.default_rng(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pts, scores = mpp.infer_image(data, region_data=region)
        torch.cuda.synchronize(); ts.append(round(time.perf_counter() - t0, 4))
    out.setdefault(f"remap_table={mode}", []).append({"sample_merge_score_s": ts, "kernel_ms": round(mpp.last_run["kernel_ms"], 2), "detections": len(pts)})
print(json.dumps(out, indent=1))
