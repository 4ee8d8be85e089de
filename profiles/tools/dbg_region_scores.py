"""debug: Papangelou scores of gathered points, whole image vs per-rank regions"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_configs import make_model
from mpp_cnn_rs_object_detection_amd import distributed as mdist, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.data_loaders import crop_region
from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

H = W = 600
mpp = make_model("mpp_hrcM.json")
mpp.config["inference"]["rjmcmc_params"]["burn_in"] = 20000
gt_xy, gt_marks = synth.make_gt(600, 260, tile_id=300)
det, marks = synth.render_maps((H, W), gt_xy, gt_marks, noise=0.05, noise_seed=0)
data = ImageWMaps(name="0", shape=(H, W), image=None, detection_map=det, param_dist_maps=marks,
                  mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
mpp.infer_image(data)
run = mpp.last_run
pts, tile_of = [], []
for t, (a, res) in enumerate(zip(run["anchors"], run["tile_results"])):
    for r in res:
        pts.append(Rectangle(int(r.x + a[0]), int(r.y + a[1]), size=r.size, ratio=r.ratio, angle=r.angle)); tile_of.append(t)
tile_of = np.array(tile_of)
xy = np.array([[p.x, p.y] for p in pts])
unit, pair = mpp.energy_setup.make_energies(data)
full = EPointsSet(pts, (H, W), unit, pair, image_data=data, point_capacity=2048)
s_full = full.papangelou_all(energy_combinator=mpp.energy_model)
print("points", len(pts))
owner = mdist.tile_owner(9, 2)[tile_of]
for rank in (0, 1):
    region = mpp.own_region((H, W), rank, 2)
    rd = crop_region(data, region)
    x0, x1, y0, y1 = region
    inside = (xy[:, 0] >= x0) & (xy[:, 0] < x1) & (xy[:, 1] >= y0) & (xy[:, 1] < y1)
    idx = np.nonzero(inside)[0]
    local = [Rectangle(int(xy[k, 0] - x0), int(xy[k, 1] - y0), size=pts[k].size, ratio=pts[k].ratio, angle=pts[k].angle) for k in idx]
    u2, p2 = mpp.energy_setup.make_energies(rd)
    sub = EPointsSet(local, rd.shape, u2, p2, image_data=rd, point_capacity=2048)
    s = sub.papangelou_all(energy_combinator=mpp.energy_model)
    own = owner[idx] == rank
    d = np.abs(s[own] - s_full[idx[own]]) / np.abs(s_full[idx[own]])
    print(f"rank {rank} region {region}: {len(idx)} points inside, {own.sum()} owned, max rel diff of owned scores {d.max():.3e}")
    for j in np.argsort(-d)[:5]:
        k = idx[own][j]
        print("   ", pts[k].as_row(), "tile", tile_of[k], "full", s_full[k], "region", s[own][j])
    for gm in (0, 100000):
        sub._ctx.set_option("scratch_grid_min_points", gm)
        sub._dirty = True
        s2 = sub.papangelou_all(energy_combinator=mpp.energy_model)
        print(f"    scratch_grid_min_points={gm}: max |diff| vs default {np.abs(s2 - s).max():.3e}")
for gm in (0, 100000):
    full._ctx.set_option("scratch_grid_min_points", gm); full._dirty = True
    s2 = full.papangelou_all(energy_combinator=mpp.energy_model)
    print(f"full set, scratch_grid_min_points={gm}: max |diff| vs default {np.abs(s2 - s_full).max():.3e}")
k = [i for i, p in enumerate(pts) if (p.x, p.y) == (191, 501)]
print("the duplicate pair:", [(pts[i].as_row(), tile_of[i], s_full[i]) for i in k])
