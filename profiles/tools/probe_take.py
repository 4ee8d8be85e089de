import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import mappings, synth, sampler as S
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
cfg = json.load(open("model_configs/mpp/config_mpp_log.json"))
model = MPPModel(cfg, phase="val", load=True)
size, n_obj = 4096, 5000
gt_xy, gt_marks = synth.make_gt(size, n_obj, tile_id=500)
det, marks = synth.render_maps((size, size), gt_xy, gt_marks)
data = ImageWMaps(name="0001", shape=(size, size), image=None, detection_map=torch.from_numpy(det).cuda(),
                  param_dist_maps=[torch.from_numpy(m).cuda() for m in marks], mappings=mappings.default_mappings(),
                  param_names=Rectangle.PARAMETERS, gt_config=[])
orig_to = S._to_rectangles
acc = {"to_rect": 0.0, "calls": 0}
def timed(xy, mk):
    t0 = time.perf_counter(); r = orig_to(xy, mk); acc["to_rect"] += time.perf_counter() - t0; acc["calls"] += 1; return r
S._to_rectangles = timed
from mpp_cnn_rs_object_detection_amd import hip_api
og = hip_api.MppContext.get_points_all
def gpa(self):
    t0 = time.perf_counter(); r = og(self); acc["get_points_all"] = acc.get("get_points_all", 0) + time.perf_counter() - t0; return r
hip_api.MppContext.get_points_all = gpa
for rep in range(3):
    acc.update({"to_rect": 0.0, "calls": 0, "get_points_all": 0.0})
    model.rng = np.random.default_rng(0)
    t0 = time.perf_counter(); model.infer_image(data); torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(rep, "total", round(tt, 4), {k: round(v, 4) for k, v in acc.items()}, "kernel_ms", round(model.last_run["kernel_ms"], 1))
