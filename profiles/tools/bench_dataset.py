"""Dataset throughput on one GPU: N images of 600x600 (9 tiles each, score maps as host arrays) through
`MPPModel.infer_image` one by one (the reference's order of work: a launch per image) and through
`MPPModel.infer_images` (tiles of all images of a batch in ONE launch, `TILES_PER_LAUNCH` = 256).  mpp_hrcM, 30 257 steps."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_configs import make_model
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 28
images = []
for k in range(n_img):
    gt_xy, gt_marks = synth.make_gt(600, 260, tile_id=700 + k)
    det, marks = synth.render_maps((600, 600), gt_xy, gt_marks)
    images.append(ImageWMaps(name=f"{k:04}", shape=(600, 600), image=None, detection_map=det, param_dist_maps=marks,
                             mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[]))
mpp = make_model("mpp_hrcM.json")
mpp.rng = np.random.default_rng(0)
mpp.infer_image(images[0])                                   # warm-up
out = {"images": n_img, "tiles_per_image": 9, "steps_per_chain": 30257}
mpp.rng = np.random.default_rng(0)
t0 = time.perf_counter()
one = [mpp.infer_image(d) for d in images]
out["image_by_image_s"] = time.perf_counter() - t0
mpp.rng = np.random.default_rng(0)
t0 = time.perf_counter()
many = []
for b in range(0, n_img, 28):                                # 28 images x 9 tiles = 252 tiles per launch
    many += mpp.infer_images(images[b:b + 28])
out["batched_s"] = time.perf_counter() - t0
same = all(sorted(p.as_row() for p in a[0]) == sorted(p.as_row() for p in b[0]) for a, b in zip(one, many))
out.update(images_per_s_image_by_image=n_img / out["image_by_image_s"], images_per_s_batched=n_img / out["batched_s"],
           speedup=out["image_by_image_s"] / out["batched_s"], identical_detections=bool(same),
           detections=int(sum(len(a[0]) for a in many)))
print(json.dumps(out))
