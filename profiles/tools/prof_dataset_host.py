"""cProfile of the dataset path (MPPModel.infer_images, 56 images of 600x600): where the host time goes."""
import cProfile, pstats, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_configs import make_model
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
n_img = 56
images = []
for k in range(n_img):
    gt_xy, gt_marks = synth.make_gt(600, 260, tile_id=700 + k)
    det, marks = synth.render_maps((600, 600), gt_xy, gt_marks)
    images.append(ImageWMaps(name=f"{k:04}", shape=(600, 600), image=None, detection_map=det, param_dist_maps=marks,
                             mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[]))
mpp = make_model("mpp_hrcM.json")
mpp.rng = np.random.default_rng(0)
mpp.infer_images(images[:28])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for b in range(0, n_img, 28):
    mpp.infer_images(images[b:b + 28])
pr.disable()
print("total", time.perf_counter() - t0)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:9000])
