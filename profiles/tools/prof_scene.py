import json, os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import mappings, synth
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
model.rng = np.random.default_rng(0); model.infer_image(data)
model.rng = np.random.default_rng(0)
pr = cProfile.Profile(); pr.enable(); model.infer_image(data); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
