"""Host halves of the path against values the REFERENCE produced (tests/golden/host_golden.npz, written by
tests/golden/make_golden.py `host`): ``crop_image_w_maps`` (models/mpp/data_loaders.py:74-119) and the calibration
functions (models/mpp/calibration/energy_calibration.py:19-185).  The GPU halves (merge, ordering-criterion loss) are in
tests/test_gpu_host_golden.py."""
import numpy as np
import pytest

from helpers import GOLDEN
from mpp_cnn_rs_object_detection_amd import mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.data_loaders import crop_image_w_maps, labels_to_rectangles, tile_anchors
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

Z = np.load(f"{GOLDEN}/host_golden.npz")


def toy_image():
    """the 256 x 300 image of make_host_golden (rebuilt from the stored ground truth; the maps are seeded)"""
    gt_xy, gt_marks = Z["merge_gt_xy"], Z["merge_gt_marks"]
    det, marks = synth.render_maps((256, 300), gt_xy, gt_marks, noise=0.1, noise_seed=9)
    np.testing.assert_array_equal(det, Z["merge_det"])
    b = 2 * gt_marks[:, 0] / (1 + gt_marks[:, 1])
    labels = {"centers": gt_xy.astype(np.int64), "parameters": np.stack([b * gt_marks[:, 1], b, gt_marks[:, 2]], axis=1),
              "categories": np.array(["small-vehicle"] * len(gt_xy)), "difficult": np.zeros(len(gt_xy), dtype=np.int64)}
    return ImageWMaps(name="0000", shape=(256, 300), image=np.zeros((256, 300, 3), np.float32), detection_map=det,
                      param_dist_maps=marks, mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS,
                      labels=labels, gt_config=labels_to_rectangles(labels))


def test_tiles_and_crops_equal_the_reference():
    image = toy_image()
    anchors = tile_anchors(image.shape, 256)
    assert [a.tolist() for a in anchors] == [[0, 0], [0, 44]]            # mpp_model.py:233-240
    for k, a in enumerate(anchors):
        crop = crop_image_w_maps(image, a, 256)
        assert tuple(crop.shape) == tuple(Z[f"crop{k}_shape"]) and crop.crop_data["tl_anchor"].tolist() == a.tolist()
        np.testing.assert_array_equal(np.asarray(crop.labels["centers"]).reshape(-1, 2), Z[f"crop{k}_centers"])
        np.testing.assert_array_equal(np.asarray(crop.labels["parameters"]).reshape(-1, 3), Z[f"crop{k}_parameters"])
        got = np.array([r.as_row() for r in crop.gt_config]).reshape(-1, 5)
        np.testing.assert_allclose(got, Z[f"crop{k}_gt"], rtol=1e-15, atol=0)
        assert float(np.sum(crop.detection_map, dtype=np.float64)) == float(Z[f"crop{k}_det_sum"])
        assert all(m.shape == (256, 256, 32) for m in crop.param_dist_maps)


def test_calibration_equals_the_reference():
    from mpp_cnn_rs_object_detection_amd import calibration as cal
    tiles = [synth.make_tile(128, 30, tile_id=int(t), noise=0.25) for t in Z["cal_tile_ids"]]
    dets, labels, gts = [], [], []
    for t in tiles:
        nrng = np.random.default_rng(3)
        dets.append(np.clip(t.det + 0.25 * nrng.random(t.det.shape, dtype=np.float32), 0, 1).astype(np.float32))
        labels.append({"centers": t.gt_xy, "parameters": t.gt_marks})
        gts.append([Rectangle(int(x), int(y), size=float(m[0]), ratio=float(m[1]), angle=float(m[2]))
                    for (x, y), m in zip(t.gt_xy, t.gt_marks)])
    assert cal.calibrate_detection_threshold(dets, labels) == pytest.approx(float(Z["cal_threshold"]), abs=1e-12)
    mn, mx = cal.calibrate_min_area(gts)
    assert mn == pytest.approx(float(Z["cal_min_area"]), rel=1e-12) and mx == pytest.approx(float(Z["cal_max_area"]), rel=1e-12)
    coefs, icpts = cal.calibrate_param_dists([t.marks for t in tiles], gts, mappings.default_mappings(), Rectangle.PARAMETERS,
                                             np.random.default_rng(4))
    # same draws of the wrong classes (generator consumed in the same order), same scikit-learn solver
    np.testing.assert_allclose(coefs, Z["cal_coefs"], rtol=1e-6)
    np.testing.assert_allclose(icpts, Z["cal_intercepts"], rtol=1e-6)
