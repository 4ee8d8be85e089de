"""BASELINE.json configs 3, 4 and 5 at their full sizes on one MI355X (the multi-GPU half of each -- tiles dealt to
ranks, one all-gather of the device-packed detections -- is covered by tests/test_gpu_multirank.py with two ranks):

* config 3: eight independent 512x512 / 200-object tiles in ONE context, 100 001 steps each, against eight chains of
  the CPU oracle (final configurations equal), then the gather buffer packed on the device;
* config 4: the 2048x2048 mosaic (4 x 4 of the 512-px generator) through ``MPPModel.infer_image`` with ``mpp_log``:
  64 overlapping-free 256-px tiles in one launch, two of them against the oracle, merge and scores as properties;
* config 5: a 4096x4096 scene of the reference's *image* recipe (data/make_synth_data.py:16-47, ~5 000 rectangles)
  through PosNet + ShapeNet + epilogues + 256 chains + merge + scores; two tiles against the oracle on the maps the
  nets produced.
"""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import REPO, hrc_model
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

pytestmark = pytest.mark.gpu


def make_model(config_name, **kwargs):
    from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
    cfg = json.load(open(os.path.join(REPO, "model_configs", "mpp", config_name)))
    cwd = os.getcwd()
    os.chdir(REPO)                       # paths_config.json is resolved from the working directory, as upstream
    try:
        return MPPModel(cfg, phase="val", load=True, **kwargs)
    finally:
        os.chdir(cwd)


def matched_fraction(centers, gt_xy, tol=2.0):
    from scipy.spatial import cKDTree
    if len(centers) == 0:
        return 0.0
    d, _ = cKDTree(np.asarray(centers, dtype=float)).query(np.asarray(gt_xy, dtype=float))
    return float((d <= tol).mean())


def check_tile_against_oracle(det, marks, setup, model, steps, seed, chain, T0, alpha, expect_xy, expect_marks):
    """One tile's chain, as ``sample_rjmcmc`` runs it (naive init, intensity = max(1, n0)), step by step against the CPU
    oracle (``helpers.lockstep_vs_oracle``: proposals, dE to 1e-9, accept decisions; ties within the dE tolerance at
    frozen temperatures are counted), on a one-tile context with the chain id ``chain``.  The final configuration must
    be ``expect_xy`` / ``expect_marks`` EXACTLY -- i.e. what the production launch over all tiles returned for this
    tile -- and equal the oracle's.  Returns the number of ties."""
    from helpers import lockstep_vs_oracle
    maps = mappings.default_mappings()
    o = oracle.Oracle(det.shape, det, marks, model, kernels.make_kernels(maps, 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    kd = kernels.make_kernels(maps, float(max(1, len(xy))))
    o = oracle.Oracle(det.shape, det, marks, model, kd)
    o.set_points(xy, mk)
    o.set_temperature(T0, alpha, 0.0)
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
    ctx.set_maps(det, marks)
    ctx.set_model(model, maps)
    ctx.naive_init(setup.detection_threshold, 6.0)
    gxy0, gm0 = ctx.get_points(0)
    np.testing.assert_array_equal(gxy0, xy)                      # naive_detection: GPU == oracle
    np.testing.assert_array_equal(gm0, mk)
    ctx.set_kernels(kd)
    ctx.set_schedule(T0, alpha, 0.0)
    ties = lockstep_vs_oracle(ctx, o, steps, seed, chain, alpha)
    gxy, gm = ctx.get_points(0)
    oxy, om = o.get_points()
    np.testing.assert_array_equal(gxy, oxy)
    np.testing.assert_allclose(gm, om, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(gxy, np.asarray(expect_xy).reshape(-1, 2))        # traced one-tile chain == production
    np.testing.assert_array_equal(gm, np.asarray(expect_marks).reshape(-1, 3))
    ctx.close()
    return ties


def test_config3_eight_512_tiles_in_one_context_equal_eight_oracle_chains():
    setup, comb = hrc_model()
    unit, pair = setup.make_energies()
    model = E.build_model_desc(unit, pair, comb)
    maps = mappings.default_mappings()
    tiles = [synth.make_tile(512, 200, tile_id=i) for i in range(8)]
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
    ctx.set_maps(np.stack([t.det for t in tiles]), [np.stack([t.marks[k] for t in tiles]) for k in range(3)])
    ctx.set_model(model, maps)
    ctx.naive_init(setup.detection_threshold, 6.0)
    n0 = ctx.counts()[:8]
    ctx.set_kernels(kernels.make_kernels(maps, 1.0), intensity=np.maximum(1, n0).astype(np.float64))
    ctx.set_schedule(1.0, 0.999, 0.0)
    steps, seed = 100001, 20261004
    ctx.run(steps, seed=seed, chain0=0)
    got = ctx.get_points_all()[:8]
    ties = []
    for i, t in enumerate(tiles):
        gxy, gm = got[i]
        ties.append(check_tile_against_oracle(t.det, t.marks, setup, model, steps, seed, i, 1.0, 0.999, gxy, gm))
        assert matched_fraction(gxy, t.gt_xy) >= 0.97 and len(gxy) <= 1.03 * len(t.gt_xy)
    print("config 3: ties within the dE tolerance per 100 001-step chain:", ties)
    assert max(ties) <= 5             # 70 000 of the steps run at T < 1e-13; a handful (0 - 2) of zero-dE proposals tie
    # the all-gather's send buffer, packed on the device: records in tile order, image coordinates = tile + anchor
    import torch
    cap = 8 * 1024
    buf = torch.zeros((cap + 1, 7), dtype=torch.float64, device="cuda:0")
    anchors = np.array([[512 * (i // 4), 512 * (i % 4)] for i in range(8)])
    n = ctx.pack_detections(np.arange(8) + 100, anchors, cap, buf)
    rec = buf.cpu().numpy()
    assert n == sum(len(g[0]) for g in got) == int(rec[0, 0]) and not rec[1 + n:].any()
    k = 1
    for i, (gxy, gm) in enumerate(got):
        r = rec[k:k + len(gxy)]
        assert np.all(r[:, 0] == 100 + i)
        np.testing.assert_array_equal(r[:, 1:3], gxy + anchors[i])
        np.testing.assert_array_equal(r[:, 3:6], gm)
        k += len(gxy)
    with pytest.raises(hip_api.MppError):
        ctx.pack_detections(np.arange(8), anchors, 16, torch.zeros((17, 7), dtype=torch.float64, device="cuda:0"))


def test_config4_2048_mosaic_mpp_log_through_infer_image():
    from helpers import log_model
    det, marks, gt_xy, gt_marks = synth.make_mosaic(4, 512, 200)
    data = ImageWMaps(name="0004", shape=(2048, 2048), image=None, detection_map=det, param_dist_maps=marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
    mpp = make_model("config_mpp_log.json")
    pts, scores = mpp.infer_image(data)
    run = mpp.last_run
    assert len(run["anchors"]) == 64 and run["patch"] == 256 and run["total_steps"] == 30003
    centers = np.array([[p.x, p.y] for p in pts], dtype=float)
    # full-size properties: the objects are found, duplicates of neighbouring tiles are merged, every survivor is scored
    assert matched_fraction(centers, gt_xy) >= 0.97 and len(centers) <= 1.03 * len(gt_xy)
    from scipy.spatial import cKDTree
    assert len(cKDTree(centers).query_pairs(3.0)) == 0
    assert len(scores) == len(centers) and np.all(np.isfinite(scores)) and np.all(scores > 0)
    # two tiles against the oracle: the same Philox chain (seed drawn from default_rng(0), chain id = tile index)
    setup, comb = log_model()
    unit, pair = setup.make_energies()
    model = E.build_model_desc(unit, pair, comb)
    p = mpp.config["inference"]["rjmcmc_params"]
    assert run["seed"] == int(np.random.default_rng(0).integers(0, 2 ** 63 - 1))
    for t in (9, 62):
        ax, ay = run["anchors"][t]
        sl = (slice(ax, ax + 256), slice(ay, ay + 256))
        res = run["tile_results"][t]
        check_tile_against_oracle(np.ascontiguousarray(det[sl]), [np.ascontiguousarray(m[sl]) for m in marks], setup, model,
                                  run["snapshot_step"] + 1, run["seed"], t, p["init_temperature"], p["alpha_t"],
                                  [(q.x, q.y) for q in res], [(q.size, q.ratio, q.angle) for q in res])
    # the scores are the Papangelou intensities of the merged configuration on the FULL image (mpp_model.py:296-304)
    o = oracle.Oracle((2048, 2048), det, marks, model)
    o.set_points(centers.astype(np.int32), np.array([[q.size, q.ratio, q.angle] for q in pts]))
    np.testing.assert_allclose(scores, np.exp(-o.papangelou()), rtol=1e-8)


def random_nets(seed=0, device=0, dtype=None):
    return synth.random_score_nets(seed, device, dtype)


def calibrate_div_clf(nets, img_crop, frac=0.0015):
    return synth.calibrate_div_clf(nets, img_crop, frac)


def test_config5_4096_scene_with_the_nets():
    import torch
    img, gt_xy, gt_marks = synth.make_scene_image((4096, 4096), 5250, noise=0.02, seed=5)
    assert 4800 <= len(gt_xy) <= 5250
    nets = random_nets()
    calibrate_div_clf(nets, img[:1024, :1024])
    mpp = make_model("mpp_hrcM.json", nets=nets)
    data = ImageWMaps(name="0005", shape=(4096, 4096), image=img, detection_map=None, param_dist_maps=None,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
    region = mpp.region_maps(data)                                   # PosNet + ShapeNet + epilogues, maps stay in HBM
    assert region.detection_map.is_cuda and tuple(region.detection_map.shape) == (4096, 4096)
    assert tuple(region.param_dist_maps[2].shape) == (4096, 4096, 32)
    pts, scores = mpp.infer_image(data, region_data=region)
    run = mpp.last_run
    assert len(run["anchors"]) == 256 and run["total_steps"] == 30257
    centers = np.array([[p.x, p.y] for p in pts], dtype=float).reshape(-1, 2)
    assert len(centers) > 200, "the calibrated random posnet should fire on some thousand pixels"
    from scipy.spatial import cKDTree
    assert len(cKDTree(centers).query_pairs(3.0)) == 0
    assert len(scores) == len(centers) and np.all(np.isfinite(scores))
    # two tiles against the oracle, on the maps the nets left on the device
    setup, comb = hrc_model()
    unit, pair = setup.make_energies()
    model = E.build_model_desc(unit, pair, comb)
    busiest = int(np.argmax([len(r) for r in run["tile_results"]]))
    for t in (busiest, 255):
        ax, ay = (int(v) for v in run["anchors"][t])
        sl = (slice(ax, ax + 256), slice(ay, ay + 256))
        tdet = region.detection_map[sl].contiguous().cpu().numpy()
        tmarks = [m[sl].contiguous().cpu().numpy() for m in region.param_dist_maps]
        res = run["tile_results"][t]
        check_tile_against_oracle(tdet, tmarks, setup, model, run["snapshot_step"] + 1, run["seed"], t, 1.0, 0.999,
                                  [(q.x, q.y) for q in res], [(q.size, q.ratio, q.angle) for q in res])
