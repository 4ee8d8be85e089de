"""The drop-in boundary end to end on the GPU: ``sample_rjmcmc`` with the reference's signature,
``EPointsSet`` facade, ``perturbation_sampler`` kernel walks, and ``main.py -p infer -m mpp`` on a
synthetic dataset laid out like the reference's (images / annotations / inference pickles)."""
import json
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest

import oracle
from helpers import REPO, hrc_model, log_model
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps, Perturbation
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

pytestmark = pytest.mark.gpu


def image_data(tile, name="0000"):
    gt = [Rectangle(int(x), int(y), size=float(m[0]), ratio=float(m[1]), angle=float(m[2]))
          for (x, y), m in zip(tile.gt_xy, tile.gt_marks)]
    return ImageWMaps(name=name, shape=tile.shape, image=None, detection_map=tile.det, param_dist_maps=tile.marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=gt)


def matched(points, gt_xy, tol=2.0):
    xy = np.array([[p.x, p.y] for p in points], dtype=float).reshape(-1, 2)
    if len(xy) == 0:
        return 0
    d = np.sqrt(((xy[:, None, :] - gt_xy[None]) ** 2).sum(-1))
    return int((d.min(axis=0) <= tol).sum())


def test_sample_rjmcmc_has_the_reference_signature_and_finds_the_objects():
    from mpp_cnn_rs_object_detection_amd.sampler import sample_rjmcmc
    tile = synth.make_tile(256, 50, tile_id=0)
    setup, comb = hrc_model()
    res = sample_rjmcmc(image_data(tile), rng=np.random.default_rng(0), num_samples=1, energy_combinator=comb,
                        init_config="naive", init_temperature=1, alpha_t=0.999, burn_in=30000, energy_setup=setup,
                        samples_interval=128, target_temperature=0.0)
    assert isinstance(res, list) and len(res) == 1 and all(isinstance(p, Rectangle) for p in res[-1])
    # the reference recovers 50/50 on this tile with this schedule
    assert matched(res[-1], tile.gt_xy) >= 49 and len(res[-1]) <= 53


def test_sample_rjmcmc_batch_equals_the_oracle_chain_including_the_returned_snapshot():
    from mpp_cnn_rs_object_detection_amd import kernels
    from mpp_cnn_rs_object_detection_amd.sampler import TileBatchSampler, resolve_schedule
    tile = synth.make_tile(96, 16, tile_id=3, noise=0.1)
    setup, comb = log_model()
    alpha, Tt, total, snaps = resolve_schedule(1, 1.0, 0.995, 2000, 128, 0.0)
    s = TileBatchSampler([image_data(tile)], setup, comb, spec_waves=4, point_capacity=256)
    s.init("naive")
    out = s.run(total, snaps, 1, 1.0, alpha, Tt, seed=9)[0][-1]
    unit, pair = setup.make_energies()
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, E.build_model_desc(unit, pair, comb),
                      kernels.make_kernels(mappings.default_mappings(), float(s.intensity[0])))
    oxy, om = o.naive_detection(setup.detection_threshold, 6.0)
    o.set_points(oxy, om)
    o.set_temperature(1.0, alpha, Tt)
    o.run(snaps[-1] + 1, 9, chain=0)           # the returned configuration is the LAST SAMPLED one, not the final one
    sxy, sm = o.get_points()
    assert [(p.x, p.y) for p in out] == [tuple(r) for r in sxy.tolist()]
    np.testing.assert_allclose([[p.size, p.ratio, p.angle] for p in out], sm, rtol=1e-9, atol=1e-9)


def test_epointsset_facade():
    from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
    tile = synth.make_tile(96, 16, tile_id=4, noise=0.1)
    data = image_data(tile)
    setup, comb = hrc_model()
    unit, pair = setup.make_energies(data)
    pts = EPointsSet(data.gt_config, data.shape, unit, pair, image_data=data)
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, E.build_model_desc(unit, pair, comb))
    o.set_points(tile.gt_xy, tile.gt_marks)
    e0 = pts.total_energy(comb)
    assert e0 == pytest.approx(o.total_energy(), rel=1e-10, abs=1e-9)
    u, v = data.gt_config[3], Rectangle(40, 41, size=6.0, ratio=0.5, angle=0.4)
    assert u in pts and v not in pts and len(pts) == 16
    d = pts.energy_delta(Perturbation(type=None, removal=u, addition=v), comb)
    assert d == pytest.approx(o.delta([3], [[40, 41]], [[6.0, 0.5, 0.4]]), rel=1e-10, abs=1e-9)
    new = pts.apply_perturbation(Perturbation(type=None, removal=u, addition=v), inplace=False)
    assert u in pts and u not in new and v in new and len(new) == 16
    assert new.total_energy(comb) - e0 == pytest.approx(d, abs=1e-8)       # test_perturbation_sampler.py:99
    assert pts.total_energy(comb) == pytest.approx(e0, abs=1e-12)          # the copy shares the GPU context safely
    np.testing.assert_allclose(pts.papangelou_all(comb, return_energy_delta=True), o.papangelou(), rtol=1e-9, atol=1e-9)
    assert pts.papangelou(u, comb, remove_u_from_point_set=True) == pytest.approx(np.exp(-o.papangelou()[3]))
    with pytest.raises(ValueError):
        pts.papangelou(u, comb)                                            # energy_point_set.py:104-107
    with pytest.raises(KeyError):
        pts.energy_delta(Perturbation(type=None, removal=v), comb)         # energy_point_set.py:88-100
    with pytest.raises(AssertionError):
        pts.add(Rectangle(96, 0, 1, 1, 0))                                 # point_set.py:99


def test_kernel_perturbation_walks_keep_delta_consistent():
    """The reference's own property (test/test_perturbation_sampler.py:87-99) for aggregated kernel walks."""
    from mpp_cnn_rs_object_detection_amd.perturbation_sampler import sample_multiple_kernel_perturbations
    from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
    tile = synth.make_tile(128, 30, tile_id=8, noise=0.2)
    data = image_data(tile)
    setup, comb = hrc_model()
    unit, pair = setup.make_energies(data)
    base = EPointsSet(data.gt_config, data.shape, unit, pair, image_data=data)
    data.gt_config_set = base
    perts = sample_multiple_kernel_perturbations(data, n_samples=12, rng=np.random.default_rng(0), energy_setup=setup,
                                                 iter_per_point=2, return_perturbations=True, aggregate_pert=True)
    e0 = base.total_energy()
    deltas = base.energy_delta_batch(perts)
    assert len({len(p.addition) + len(p.removal) for p in perts}) > 1        # different walks
    for p, d in zip(perts, deltas):
        assert all(a not in base for a in p.addition) and all(r in base for r in p.removal)
        e1 = base.apply_perturbation(p, inplace=False).total_energy()
        assert abs(d - (e1 - e0)) < 1e-8


@pytest.fixture
def synthetic_dataset(tmp_path):
    """A dataset directory in the reference's layout with score maps handed off as pickles."""
    root = tmp_path
    for d in ("model_configs", "models_storage"):
        shutil.copytree(os.path.join(REPO, d), root / d)
    with open(root / "paths_config.json", "w") as f:
        json.dump({"dataset_path": ["data/"], "model_path": ["models_storage/"]}, f)
    gt_xy = write_image(root, "val", 7, 21)
    return root, gt_xy


def write_image(root, subset, image_id, seed):
    """one 300 x 420 image (not a multiple of 256: overlapping tiles + merge) with its score-map pickles"""
    H, W = 300, 420
    gt_xy, gt_marks = synth.make_gt(300, 70, tile_id=seed)
    extra_xy, extra_marks = synth.make_gt(300, 20, tile_id=seed + 1)
    sel = extra_xy[:, 1] < 110
    gt_xy = np.concatenate([gt_xy, extra_xy[sel] + np.array([0, 300])])
    gt_marks = np.concatenate([gt_marks, extra_marks[sel]])
    det, marks = synth.render_maps((H, W), gt_xy, gt_marks)
    from matplotlib import pyplot as plt
    base = root / "data" / "SYNTH" / subset
    for sub in ("images", "annotations", "metadata"):
        os.makedirs(base / sub, exist_ok=True)
    plt.imsave(base / "images" / f"{image_id:04}.png", np.stack([det] * 3, axis=-1))
    b = 2 * gt_marks[:, 0] / (1 + gt_marks[:, 1])
    params = np.stack([b * gt_marks[:, 1], b, gt_marks[:, 2]], axis=1)      # (a, b, angle)
    with open(base / "annotations" / f"{image_id:04}.pkl", "wb") as f:
        pickle.dump({"centers": gt_xy.astype(np.int64), "parameters": params,
                     "categories": np.array(["small-vehicle"] * len(gt_xy), dtype=object),
                     "difficult": np.zeros(len(gt_xy), dtype=np.int64)}, f)
    with open(base / "metadata" / f"{image_id:04}.json", "w") as f:
        json.dump({"shape": [H, W], "n_objects": int(len(gt_xy))}, f)
    for model, payload in (("posvec_dota", {"detection_map": det}),
                           ("shape_dota", {"output": [np.moveaxis(m, -1, 0)[None] for m in marks],
                                           "mappings": mappings.default_mappings()})):
        d = root / "data" / "inference" / "SYNTH" / subset / model
        os.makedirs(d, exist_ok=True)
        with open(d / f"{image_id:04}_results.pkl", "wb") as f:
            pickle.dump(payload, f)
    return gt_xy
    return root, gt_xy


@pytest.mark.parametrize("config", ["mpp_hrcM", "config_mpp_log.json"])
def test_main_infer_mpp_end_to_end(synthetic_dataset, config):
    root, gt_xy = synthetic_dataset
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-p", "infereval", "-m", "mpp", "-c", config,
                        "-d", "SYNTH", "-o"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    name = "mpp_hrcM" if config == "mpp_hrcM" else "mpp_log"
    out = root / "data" / "inference" / "SYNTH" / "val" / name
    with open(out / "0007_results.pkl", "rb") as f:
        res = pickle.load(f)
    centers = np.asarray(res["detection_center"], dtype=float)
    d = np.sqrt(((centers[:, None, :] - gt_xy[None]) ** 2).sum(-1))
    assert (d.min(axis=0) <= 2).mean() >= 0.95, "fewer than 95% of the objects recovered"
    assert len(centers) <= 1.1 * len(gt_xy)
    # duplicates of the overlapping tiles were merged: no two detections closer than 3 px
    dd = np.sqrt(((centers[:, None, :] - centers[None]) ** 2).sum(-1)) + 1e9 * np.eye(len(centers))
    assert dd.min() > 3
    lines = open(out / "dota" / "det" / "vehicle.txt").read().splitlines()
    assert len(lines) == len(centers) and all(len(ln.split()) == 10 and ln.startswith("0007 ") for ln in lines)
    assert len(open(out / "dota" / "gt" / "0007.txt").read().splitlines()) == len(gt_xy)
    assert len(res["detection_score"]) == len(centers) and min(res["detection_score"]) > 0
    # -p infereval also ran the DOTA task-1 evaluation (metrics/dota_eval.py:16-87): one json per IoU threshold
    for postfix in ("", "-SV"):
        with open(out / ("dota" + postfix) / "metrics0.25.json") as f:
            m = json.load(f)["vehicle"]
        assert m["ap"] > 0.9 and len(m["precision"]) == len(m["recall"]) == len(centers)
        assert os.path.exists(out / ("dota" + postfix) / "metrics0.75.json")


def test_main_train_then_infereval_with_the_learned_weights(synthetic_dataset):
    """``main.py -p train -m mpp -c config_mpp_log.json``: calibration + ordering-criterion weight learning on random
    training patches (mpp_model.py:106-200), then inference + evaluation with what was learned."""
    root, gt_xy = synthetic_dataset
    for k, seed in enumerate((31, 41, 51)):
        write_image(root, "train", k, seed)
    store = root / "models_storage" / "mpp" / "mpp_log"
    for f in ("calibration.json", "energy_combination_model.json"):
        os.remove(store / f)                                   # they must come from this run
    cfg = json.load(open(root / "model_configs" / "mpp" / "config_mpp_log.json"))
    cfg["ordering_criterion"].update(n_epochs=6, samples_per_image=8)
    cfg["data_loader"]["batch_size"] = 3
    cfg["inference"]["rjmcmc_params"]["burn_in"] = 12000
    with open(root / "cfg_train.json", "w") as f:
        json.dump(cfg, f)
    env = dict(os.environ, PYTHONPATH=REPO)
    run = lambda proc: subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-p", proc, "-m", "mpp", "-c",
                                       str(root / "cfg_train.json"), "-d", "SYNTH", "-o"], cwd=root, env=env,
                                      capture_output=True, text=True, timeout=900)
    r = run("train")
    assert r.returncode == 0, r.stderr[-3000:]
    cal = json.load(open(store / "calibration.json"))
    assert 20 < cal["min_area"] < cal["max_area"] < 80          # objects are ~4.5 x 9 px
    comb = json.load(open(store / "energy_combination_model.json"))
    assert comb["type"] == "LogisticEnergyCombinator" and len(comb["weights"]) == 8
    log = json.load(open(store / "log.json"))
    assert len(log["loss"]) == 6 and log["loss"][-1] < log["loss"][0] and "PositionEnergy_weight" in log
    r = run("infereval")
    assert r.returncode == 0, r.stderr[-3000:]
    out = root / "data" / "inference" / "SYNTH" / "val" / "mpp_log"
    with open(out / "dota" / "metrics0.25.json") as f:
        assert json.load(f)["vehicle"]["ap"] > 0.85


def test_borrowed_device_maps_give_the_same_chain_as_uploaded_host_maps():
    """The U-Net epilogues leave the score maps in HBM and the sampler borrows them (``mpp_set_maps(on_device=1)``,
    SURVEY 8(b) "score-map hand-off stays on device"): same detections as with host arrays, tile by tile."""
    import torch
    from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
    cfg = json.load(open(os.path.join(REPO, "model_configs", "mpp", "mpp_hrcM.json")))
    cfg["inference"]["rjmcmc_params"]["burn_in"] = 4000
    cwd = os.getcwd()
    os.chdir(REPO)
    try:
        model = MPPModel(cfg, phase="val", load=True)
    finally:
        os.chdir(cwd)
    gt_xy, gt_marks = synth.make_gt(300, 60, tile_id=61)
    det, marks = synth.render_maps((300, 400), gt_xy, gt_marks)
    host = ImageWMaps(name="0001", shape=(300, 400), image=None, detection_map=det, param_dist_maps=marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
    dev = ImageWMaps(name="0001", shape=(300, 400), image=None, detection_map=torch.from_numpy(det).cuda(),
                     param_dist_maps=[torch.from_numpy(m).cuda() for m in marks],
                     mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
    model.rng = np.random.default_rng(0)
    a, sa = model.infer_image(host)
    model.rng = np.random.default_rng(0)
    b, sb = model.infer_image(dev)
    pa = sorted((p.x, p.y, p.size, p.ratio, p.angle) for p in a)
    pb = sorted((p.x, p.y, p.size, p.ratio, p.angle) for p in b)
    assert pa == pb and len(pa) > 5            # (a short, still hot chain: only the equality matters here)
    np.testing.assert_allclose(sorted(sa), sorted(sb), rtol=1e-12)


def test_split_merge_walks_and_sampling():
    """use_split_merge through the host interfaces (sample_rjmcmc.py:38-44, perturbation_sampler.py:125-149): the
    host replay of a kernel walk (split/merge rectangles recomputed in Python) lands on the device's final state, the
    aggregated perturbations keep delta = E1 - E0, and a full sampling run still finds the objects."""
    from mpp_cnn_rs_object_detection_amd.perturbation_sampler import sample_multiple_kernel_perturbations
    from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
    from mpp_cnn_rs_object_detection_amd.sampler import sample_rjmcmc
    tile = synth.make_tile(128, 30, tile_id=9, noise=0.2)
    data = image_data(tile)
    setup, comb = hrc_model()
    unit, pair = setup.make_energies(data)
    base = EPointsSet(data.gt_config, data.shape, unit, pair, image_data=data)
    data.gt_config_set = base
    walks = sample_multiple_kernel_perturbations(data, n_samples=6, rng=np.random.default_rng(5), energy_setup=setup,
                                                 iter_per_point=3, return_perturbations=True, aggregate_pert=False,
                                                 use_split_merge=True)
    kinds = {p.type for w in walks for p in w}
    assert {"Split", "Merge"} <= kinds
    assert any(isinstance(p.addition, list) and len(p.addition) == 2 for w in walks for p in w)
    finals = sample_multiple_kernel_perturbations(data, n_samples=6, rng=np.random.default_rng(5), energy_setup=setup,
                                                  iter_per_point=3, use_split_merge=True)
    perts = sample_multiple_kernel_perturbations(data, n_samples=6, rng=np.random.default_rng(5), energy_setup=setup,
                                                 iter_per_point=3, return_perturbations=True, aggregate_pert=True,
                                                 use_split_merge=True)
    e0 = base.total_energy()
    for p, d, f in zip(perts, base.energy_delta_batch(perts), finals):
        new = base.apply_perturbation(p, inplace=False)
        assert abs(d - (new.total_energy() - e0)) < 1e-8
        assert sorted((q.x, q.y, q.size, q.ratio, q.angle) for q in new) == sorted((q.x, q.y, q.size, q.ratio, q.angle) for q in f)
    res = sample_rjmcmc(image_data(synth.make_tile(256, 50, tile_id=0)), rng=np.random.default_rng(0), num_samples=1,
                        energy_combinator=comb, init_config="naive", init_temperature=1, alpha_t=0.999, burn_in=30000,
                        energy_setup=setup, samples_interval=128, target_temperature=0.0, use_split_merge=True)
    gt = synth.make_tile(256, 50, tile_id=0).gt_xy
    assert matched(res[-1], gt) >= 48 and len(res[-1]) <= 53


def test_real_parking_lot_layout_from_the_reference_data_sample():
    """The layout of data_sample/DOTA_gsd50/val/2781 (272 vehicles in 469 x 753, rows of cars 4.5-7 px apart, up to 8
    per 32-px cell) with score maps rendered from it: (i) a crowded 256-px crop, kernel against the oracle step by
    step; (ii) the whole image through MPPModel.infer_image (overlapping tiles, merge, scores)."""
    import oracle
    from mpp_cnn_rs_object_detection_amd import hip_api, kernels
    from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dota_2781.npz"))
    H, W = (int(v) for v in z["shape"])
    r = z["ref_rects"]
    gt_xy, gt_marks = r[:, :2].astype(np.int32), r[:, 2:5]
    det, marks = synth.render_maps((H, W), gt_xy, gt_marks, noise=0.1, noise_seed=3)
    # (i) the densest 256 x 256 window
    best = max(((x0, y0) for x0 in range(0, H - 255, 16) for y0 in range(0, W - 255, 16)),
               key=lambda a: int(((gt_xy[:, 0] >= a[0]) & (gt_xy[:, 0] < a[0] + 256) & (gt_xy[:, 1] >= a[1]) & (gt_xy[:, 1] < a[1] + 256)).sum()))
    sl = (slice(best[0], best[0] + 256), slice(best[1], best[1] + 256))
    cdet, cmarks = np.ascontiguousarray(det[sl]), [np.ascontiguousarray(m[sl]) for m in marks]
    setup, comb = hrc_model()
    unit, pair = setup.make_energies()
    model = E.build_model_desc(unit, pair, comb)
    o = oracle.Oracle((256, 256), cdet, cmarks, model, kernels.make_kernels(mappings.default_mappings(), 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    assert len(xy) > 80
    kd = kernels.make_kernels(mappings.default_mappings(), float(len(xy)))
    o = oracle.Oracle((256, 256), cdet, cmarks, model, kd)
    o.set_points(xy, mk)
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
    ctx.set_maps(cdet, cmarks); ctx.set_model(model, mappings.default_mappings()); ctx.set_kernels(kd); ctx.set_points(0, xy, mk)
    o.set_temperature(1.0, 0.999, 0.0); ctx.set_schedule(1.0, 0.999, 0.0)
    oout, oprops = o.run(6000, 5, chain=0, trace=True)
    gout, gprops = ctx.run(6000, 5, chain0=0, trace_tile=0)
    np.testing.assert_array_equal(gprops["kernel"], oprops["kernel"])
    np.testing.assert_array_equal(gprops["target"], oprops["target"])
    np.testing.assert_array_equal(gout["accepted"], oout["accepted"])
    np.testing.assert_allclose(gout["dE"], oout["dE"], rtol=1e-9, atol=1e-9)
    gxy, gm = ctx.get_points(); oxy, om = o.get_points()
    np.testing.assert_array_equal(gxy, oxy)
    np.testing.assert_allclose(gm, om, rtol=1e-9, atol=1e-9)
    # (ii) the whole image
    cfg = json.load(open(os.path.join(REPO, "model_configs", "mpp", "mpp_hrcM.json")))
    cwd = os.getcwd()
    os.chdir(REPO)
    try:
        mpp = MPPModel(cfg, phase="val", load=True)
    finally:
        os.chdir(cwd)
    data = ImageWMaps(name="2781", shape=(H, W), image=None, detection_map=det, param_dist_maps=marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
    pts, scores = mpp.infer_image(data)
    found = matched(list(pts), gt_xy.astype(float))
    # (tightly parked cars: annotated boxes 4.5 px apart overlap, which the overlap prior of mpp_hrcM penalises -- the
    # chain keeps ~86 % of them and invents none)
    assert found >= 0.8 * len(gt_xy) and len(pts) <= 1.05 * len(gt_xy), (found, len(pts))
    assert len(scores) == len(pts)


def test_main_infer_with_the_unets_on_the_gpu(synthetic_dataset):
    """``main.py -p infer -m mpp --unet``: PosNet + ShapeNet + epilogues + sampler on the GPU, the next image's score
    maps computed on a side stream while the current one is sampled.  The container has no trained ``model.pt``; the
    files are written with seeded random weights, so the detections mean nothing -- the path must run, keep the score
    maps on the device and write well-formed outputs for every image."""
    import torch
    from mpp_cnn_rs_object_detection_amd import unet
    root, _ = synthetic_dataset
    for k, seed in ((8, 23), (9, 24)):
        write_image(root, "val", k, seed)
    torch.manual_seed(0)
    for kind, name, net in (("posnet", "posvec_dota", unet.PosNet()), ("shapenet", "shape_dota", unet.ShapeNet())):
        d = root / "models_storage" / kind / name
        os.makedirs(d, exist_ok=True)
        torch.save(net.state_dict(), d / "model.pt")
    cfg = json.load(open(root / "model_configs" / "mpp" / "mpp_hrcM.json"))
    cfg["inference"]["rjmcmc_params"]["burn_in"] = 2000
    with open(root / "cfg_unet.json", "w") as f:
        json.dump(cfg, f)
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-p", "infer", "-m", "mpp", "-c", str(root / "cfg_unet.json"),
                        "-d", "SYNTH", "-o", "--unet"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = root / "data" / "inference" / "SYNTH" / "val" / "mpp_hrcM"
    for k in (7, 8, 9):
        with open(out / f"{k:04}_results.pkl", "rb") as f:
            res = pickle.load(f)
        assert len(res["detection_score"]) == len(res["detection_center"]) == len(res["detection_params"])
    assert len(open(out / "dota" / "imageSet.txt").read().split()) == 3


def test_dataset_inference_batches_tiles_of_several_images_without_changing_any_result(synthetic_dataset):
    """``MPPModel.infer`` on one GPU samples the tiles of consecutive images in ONE launch (``infer_images``); every tile
    keeps the seed of its image and its chain id, so each image's detections and scores are what image-by-image
    launches give."""
    root, _ = synthetic_dataset
    for k, seed in ((8, 23), (9, 24)):
        write_image(root, "val", k, seed)
    outs = {}
    for name, limit in (("batched", 256), ("single", 1)):
        cfg = json.load(open(root / "model_configs" / "mpp" / "mpp_hrcM.json"))
        cfg["inference"]["rjmcmc_params"]["burn_in"] = 6000
        cfg["inference"]["tiles_per_launch"] = limit
        with open(root / f"cfg_{name}.json", "w") as f:
            json.dump(cfg, f)
        env = dict(os.environ, PYTHONPATH=REPO)
        r = subprocess.run([sys.executable, os.path.join(REPO, "main.py"), "-p", "infer", "-m", "mpp", "-c", str(root / f"cfg_{name}.json"),
                            "-d", "SYNTH", "-o"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        launches = [ln for ln in r.stderr.splitlines() if "rjmcmc chains" in ln]
        assert len(launches) == (1 if name == "batched" else 3), launches            # 3 images x 4 tiles: one launch or three
        res = {}
        for k in (7, 8, 9):
            with open(root / "data" / "inference" / "SYNTH" / "val" / "mpp_hrcM" / f"{k:04}_results.pkl", "rb") as f:
                res[k] = pickle.load(f)
        outs[name] = res
    for k in (7, 8, 9):
        a, b = outs["batched"][k], outs["single"][k]
        assert len(a["detection_score"]) > 20
        np.testing.assert_array_equal(np.asarray(a["detection_center"]), np.asarray(b["detection_center"]))
        np.testing.assert_array_equal(np.asarray(a["detection_points"]), np.asarray(b["detection_points"]))
        np.testing.assert_array_equal(np.asarray(a["detection_score"]), np.asarray(b["detection_score"]))
