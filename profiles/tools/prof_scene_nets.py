#!/usr/bin/env python3
"""Host profile of `infer_image` on the BASELINE config 5 scene (4096 x 4096, score maps from the random-init nets, left on
the device): where the 16 ms around the 73 ms of chains go.  python profiles/tools/prof_scene_nets.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
from test_gpu_configs import make_model

img, gt_xy, gt_marks = synth.make_scene_image((4096, 4096), 5250, noise=0.02, seed=5)
nets = synth.random_score_nets(0, 0, None)
synth.calibrate_div_clf(nets, img[:1024, :1024], 0.0015)
mpp = make_model("mpp_hrcM.json", nets=nets)
data = ImageWMaps(name="0005", shape=(4096, 4096), image=img, detection_map=None, param_dist_maps=None,
                  mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
region = mpp.region_maps(data)
for _ in range(2):
    mpp.rng = np.random.default_rng(0); mpp.infer_image(data, region_data=region); torch.cuda.synchronize()
mpp.rng = np.random.default_rng(0)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable(); mpp.infer_image(data, region_data=region); torch.cuda.synchronize(); pr.disable()
print("infer_image: %.1f ms, chain kernel %.1f ms" % ((time.perf_counter() - t0) * 1e3, mpp.last_run["kernel_ms"]))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
