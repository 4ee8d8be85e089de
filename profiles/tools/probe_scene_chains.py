#!/usr/bin/env python3
"""The chains of BASELINE config 5 (4096 x 4096 scene, random-init nets, 256 tiles): kernel time, deep-round counters and
points per tile, with deep rounds of 128 / 64 / 32 steps and without: python profiles/tools/probe_scene_chains.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from mpp_cnn_rs_object_detection_amd import hip_api, mappings, synth
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
orig_run = hip_api.MppContext.run
stats = {}
def run(self, *a, **k):
    n0 = self.counts()
    r = orig_run(self, *a, **k)
    if self.last_kernel_ms() < stats.get("kernel_ms", 0.0):
        return r                                             # (keep the long launch, not the few steps after the snapshot)
    stats.update(kernel_ms=self.last_kernel_ms(), spec=self.get_option("spec_waves"), deep=self.get_option("deep"),
                 lds=self.get_option("lds_bytes"), cap=self.get_option("point_capacity"), deep_stats=self.deep_stats(),
                 n0_mean=float(np.mean(n0)), n0_max=int(np.max(n0)), n_end_mean=float(np.mean(self.counts())))
    return r
hip_api.MppContext.run = run
orig_init = hip_api.MppContext.__init__
for deep, gain in [(128, int(g)) for g in os.environ.get('PROBE_GAINS', '12,16').split(',')]:
    def init(self, *a, _d=deep, _g=gain, **k):
        orig_init(self, *a, **k)
        self.set_option("deep", _d)
        self.set_option("deep_gain", _g)
    hip_api.MppContext.__init__ = init
    mpp.rng = np.random.default_rng(0)
    stats.clear()
    pts, scores = mpp.infer_image(data, region_data=region)
    st = dict(stats)
    ds = st.get("deep_stats") or {}
    st["committed_per_round"] = ds.get("committed", 0) / max(1, ds.get("rounds", 1))
    print(json.dumps({"deep_option": deep, "gain": gain, **st}), flush=True)
