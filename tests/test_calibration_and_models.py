"""Host logic of the training side: calibration (calibration/energy_calibration.py:19-185), the torch weight models
against their NumPy combinators (combination/{hierarchical,logistic}.py), the combinator JSON round trip."""
import json
import os

import numpy as np
import pytest
import torch

from mpp_cnn_rs_object_detection_amd import calibration as C
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle


def image_data(tile):
    gt = [Rectangle(int(x), int(y), size=float(m[0]), ratio=float(m[1]), angle=float(m[2]))
          for (x, y), m in zip(tile.gt_xy, tile.gt_marks)]
    return ImageWMaps(name="0000", shape=tile.shape, image=None, detection_map=tile.det, param_dist_maps=tile.marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=gt,
                      labels={"centers": tile.gt_xy.astype(int)})


def test_min_area_quantiles():
    rects = [[Rectangle(5, 5, size=s, ratio=0.5, angle=0.0) for s in np.linspace(4, 10, 101)]]
    areas = np.array([r.length * r.width for r in rects[0]])
    lo, hi = C.calibrate_min_area(rects)
    assert lo == pytest.approx(np.quantile(areas, 0.01)) and hi == pytest.approx(np.quantile(areas, 0.99))
    assert areas.min() <= lo < hi <= areas.max()


def test_detection_threshold_and_pr_curve():
    tiles = [image_data(synth.make_tile(96, 12, tile_id=k)) for k in (1, 2)]
    thr, metrics = C.precision_recall_curve_on_detection_map([t.detection_map for t in tiles], [t.labels for t in tiles],
                                                            num_thresholds=100, dilation=2)
    assert len(thr) == 100 and metrics["recall"][0] == 1.0 and np.all(np.diff(metrics["recall"]) <= 1e-12)
    t = C.calibrate_detection_threshold([t.detection_map for t in tiles], [t.labels for t in tiles], target="f1")
    # synthetic blobs: exp(-d^2 / (2 * 1.2^2)) over a 0.02 floor; the 5x5 dilated discs hold values down to ~0.06
    assert 0.02 < t < 0.9
    assert C.f_beta(0.5, 0.5, 2.0) == pytest.approx(0.5) and C.f_beta(0.0, 0.0, 1.0) == 0


def test_wrong_value_keeps_its_distance():
    rng = np.random.default_rng(0)
    size_map, _, angle_map = mappings.default_mappings()
    for _ in range(200):
        assert abs(C.generate_wrong_value(10, size_map, 2, rng) - 10) >= 2
        w = C.generate_wrong_value(0, angle_map, 2, rng)
        assert w not in (0, 1, 31)                       # cyclic neighbours excluded


def test_param_dist_calibration_separates_true_from_wrong_classes():
    tiles = [image_data(synth.make_tile(128, 30, tile_id=k)) for k in (3, 4)]
    coefs, intercepts = C.calibrate_param_dists([t.param_dist_maps for t in tiles], [t.gt_config for t in tiles],
                                                tiles[0].mappings, Rectangle.PARAMETERS, np.random.default_rng(0))
    assert len(coefs) == len(intercepts) == 3 and all(c > 0 for c in coefs)
    for c, i in zip(coefs, intercepts):                  # p = 0.9 (true class) is classified valid, p = 0.1/31 is not
        assert c * 0.9 + i > 0 > c * (0.1 / 31) + i


def test_setups_write_the_calibration_file(tmp_path):
    tiles = [image_data(synth.make_tile(128, 30, tile_id=k)) for k in (5, 6)]
    legacy = E.LegacyEnergySetup(calibration_params={"threshold_target": "f1"})
    legacy.calibrate(tiles, np.random.default_rng(0), str(tmp_path))
    d = json.load(open(tmp_path / "calibration.json"))
    assert set(d) == {"detection_threshold", "param_dist_remap_coefs", "param_dist_remap_intercepts", "min_area", "max_area"}
    again = E.LegacyEnergySetup()
    again.load_calibration(str(tmp_path))
    assert again.detection_threshold == d["detection_threshold"] and len(again.make_energies()[0]) == 3
    nocal = E.NoCalibrationEnergySetup(ratio_prior=True)
    nocal.calibrate(tiles, np.random.default_rng(0), str(tmp_path))
    d2 = json.load(open(tmp_path / "calibration.json"))
    assert d2["min_area"] == d["min_area"] and d2["param_dist_remap_coefs"] is None
    nocal2 = E.NoCalibrationEnergySetup(ratio_prior=True)
    nocal2.load_calibration(str(tmp_path))
    assert len(nocal2.make_energies()[0]) == 6


@pytest.mark.parametrize("kind", ["hierarchical", "logistic"])
def test_torch_weight_models_agree_with_their_numpy_combinators(kind, tmp_path):
    from mpp_cnn_rs_object_detection_amd.mpp_model import load_energy_combinator, save_energy_combinator
    from mpp_cnn_rs_object_detection_amd.weight_models import init_model
    setup = E.LegacyEnergySetup() if kind == "hierarchical" else E.NoCalibrationEnergySetup(ratio_prior=True)
    torch.manual_seed(1)
    wm = init_model(kind, setup)
    with torch.no_grad():
        for p in wm.parameters():
            p.add_(0.5 * torch.randn_like(p))
    names = setup.energy_names
    x = np.random.default_rng(0).normal(0, 1, (40, len(names)))
    comb = wm.get_energy_combination_function()
    vectors = {n: x[:, i].tolist() for i, n in enumerate(names)}
    assert comb.compute(vectors) == pytest.approx(float(wm.forward(torch.tensor(x, dtype=torch.float32)).detach()), rel=1e-5, abs=1e-4)
    np.testing.assert_allclose(wm.get_decision_function()(x), wm.point_energies(torch.tensor(x, dtype=torch.float32)).detach().numpy(),
                               rtol=1e-5, atol=1e-5)
    assert set(wm.as_dict()) >= ({"bias"} | ({n + "_weight" for n in names} if kind == "logistic" else {"data_weight", "prior_weight"}))
    save_energy_combinator(comb, str(tmp_path))
    back = load_energy_combinator(str(tmp_path))
    assert back.compute(vectors) == pytest.approx(comb.compute(vectors), rel=1e-6)
    assert back.coefficients(names)[1] == pytest.approx(comb.coefficients(names)[1], rel=1e-6)
