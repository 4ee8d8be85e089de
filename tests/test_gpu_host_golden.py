"""``merge_patches`` (models/mpp/data_loaders.py:122-161) and the ordering-criterion loss
(train_energy_combination/train_ordering_criterion.py:101-118) against values the REFERENCE produced
(tests/golden/host_golden.npz, tests/golden/make_golden.py `host`)."""
import numpy as np
import pytest

from helpers import GOLDEN, hrc_model, log_model, sorted_rows
from mpp_cnn_rs_object_detection_amd import hip_api, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps, Perturbation
from mpp_cnn_rs_object_detection_amd.data_loaders import crop_image_w_maps, merge_patches, tile_anchors
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
from test_host_golden import Z, toy_image

pytestmark = pytest.mark.gpu


def rects(rows):
    return [Rectangle(int(r[0]), int(r[1]), size=float(r[2]), ratio=float(r[3]), angle=float(r[4])) for r in rows]


def test_merge_patches_equals_the_reference():
    image = toy_image()
    patches = [crop_image_w_maps(image, a, 256) for a in tile_anchors(image.shape, 256)]
    results = [rects(Z["merge_in0"]), rects(Z["merge_in1"])]
    setup, comb = hrc_model()
    merged = merge_patches(patches=patches, results=results, original_image=image, energy_model=comb, method="distance",
                           energy_setup=setup, distance=3)
    pts = list(merged)
    got = sorted_rows([p.as_row() for p in pts])
    np.testing.assert_array_equal(got, Z["merge_out"])                   # the same survivors, bit for bit
    assert len(Z["merge_in0"]) + len(Z["merge_in1"]) > len(got)          # (29 duplicates across the seam were removed)
    scores = dict(zip([tuple(p.as_row()) for p in pts], merged.papangelou_all(energy_combinator=comb)))
    np.testing.assert_allclose([scores[tuple(r)] for r in Z["merge_out"]], Z["merge_scores"], rtol=2e-6)


def test_device_merge_equals_the_reference_and_the_host_merge():
    """``mpp_merge_score`` (scores, dedupe walk, removals, scores of the survivors -- all on the device, several images at
    once): the reference's survivors bit for bit, and survivor for survivor, order and scores included, what
    ``merge_patches`` + ``papangelou_all`` give."""
    from mpp_cnn_rs_object_detection_amd.data_loaders import merge_score_images
    image = toy_image()
    patches = [crop_image_w_maps(image, a, 256) for a in tile_anchors(image.shape, 256)]
    rows = [Z["merge_in0"], Z["merge_in1"]]
    agg_xy = np.concatenate([r[:, :2].astype(np.int64) + np.asarray(p.crop_data["tl_anchor"]) for r, p in zip(rows, patches)])
    agg_mk = np.concatenate([r[:, 2:5] for r in rows])
    for setup, comb in (hrc_model(), log_model()):
        merged = merge_patches(patches=patches, results=[rects(r) for r in rows], original_image=image, energy_model=comb,
                               method="distance", energy_setup=setup, distance=3)
        host_rows = np.array([p.as_row() for p in merged])
        host_scores = merged.papangelou_all(energy_combinator=comb)
        # two images in one batch: the toy image and a copy whose detections come in reverse tile order
        agg2_xy = np.concatenate([agg_xy[len(rows[0]):], agg_xy[:len(rows[0])]])
        agg2_mk = np.concatenate([agg_mk[len(rows[0]):], agg_mk[:len(rows[0])]])
        res = merge_score_images([image, image], [(agg_xy, agg_mk), (agg2_xy, agg2_mk)], comb, setup, 3)
        det, scores = res[0]
        got = np.array([p.as_row() for p in det])
        np.testing.assert_array_equal(got, host_rows)                      # same survivors in the same order
        np.testing.assert_array_equal(scores, host_scores)                 # and the same scores, bit for bit
        if setup.__class__.__name__ == "LegacyEnergySetup":
            np.testing.assert_array_equal(sorted_rows(got), Z["merge_out"])  # = the reference's survivors
        det2, scores2 = res[1]
        assert len(det2) == len(det)
        np.testing.assert_allclose(np.sort(scores2), np.sort(scores), rtol=1e-12)


def test_device_merge_declines_what_its_lds_does_not_hold():
    """The dedupe walk keeps flags, positions and scores of a tile's points in LDS: a context whose point capacity exceeds
    what 160 KB hold is refused with code -4 -- ``MPPModel.infer_image`` then merges on the host (``merge_patches``) --
    and a capacity just below the limit works."""
    from mpp_cnn_rs_object_detection_amd import energies as E
    image = toy_image()
    setup, comb = hrc_model()
    unit, pair = setup.make_energies(image)
    rows = Z["merge_in0"]
    for cap, ok in ((9000, True), (12000, False)):
        ctx = hip_api.MppContext(0, point_capacity=cap)
        ctx.set_maps(np.asarray(image.detection_map, dtype=np.float32), [np.asarray(m, dtype=np.float32) for m in image.param_dist_maps])
        ctx.set_model(E.build_model_desc(unit, pair, comb), image.mappings)
        ctx.set_points(0, rows[:, :2].astype(np.int32), rows[:, 2:5].astype(np.float64))
        if ok:
            res, removed = ctx.merge_score(3)
            assert len(res) == 1 and len(res[0][0]) == len(rows) - int(removed[0])
        else:
            with pytest.raises(hip_api.MppError) as ei:
                ctx.merge_score(3)
            assert ei.value.code == -4
        ctx.close()


@pytest.mark.parametrize("tag", ["log", "hrc"])
def test_ordering_criterion_loss_and_gradients_equal_the_reference(tag):
    import torch
    from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet
    from mpp_cnn_rs_object_detection_amd.train_ordering_criterion import criterion_loss, perturbation_rows
    from mpp_cnn_rs_object_detection_amd.weight_models import HierarchicalEnergyModel, LogisticEnergyModel
    tile = synth.make_tile(96, 18, tile_id=42, noise=0.2)
    gt = rects(np.concatenate([tile.gt_xy.astype(float), tile.gt_marks], axis=1))
    data = ImageWMaps(name="0", shape=tile.shape, image=None, detection_map=tile.det, param_dist_maps=tile.marks,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=gt)
    setup, _ = log_model() if tag == "log" else hrc_model()
    unit, pair = setup.make_energies(data)
    base = EPointsSet(gt, data.shape, unit, pair, image_data=data)
    by_row = {tuple(p.as_row()): p for p in gt}
    perts, ia, ir = [], 0, 0
    for na, nr in zip(Z[f"oc_{tag}_add_len"], Z[f"oc_{tag}_rem_len"]):
        add = rects(Z[f"oc_{tag}_add_flat"][ia:ia + na])
        rem = [by_row[tuple(r)] for r in Z[f"oc_{tag}_rem_flat"][ir:ir + nr]]
        perts.append(Perturbation(type=None, removal=rem, addition=add))
        ia, ir = ia + na, ir + nr
    if tag == "log":
        wm = LogisticEnergyModel(energy_names=setup.energy_names, use_bias=True)
        with torch.no_grad():
            wm.weights.copy_(torch.tensor(Z["oc_log_param_weights"], dtype=torch.float32))
            wm.bias.copy_(torch.tensor(float(Z["oc_log_param_bias"])))
    else:
        wm = HierarchicalEnergyModel(threshold=0.0)
        with torch.no_grad():
            for k in ("data_prior_weight", "data_weight", "prior_weight"):
                getattr(wm, k).copy_(torch.tensor(Z[f"oc_hrc_param_{k}"], dtype=torch.float32))
    rows, sign, case = perturbation_rows(base, perts, names=setup.energy_names)
    loss = criterion_loss(wm, [(rows, sign, case, len(perts))])
    assert float(loss.detach()) == pytest.approx(float(Z[f"oc_{tag}_loss"]), rel=2e-5)
    loss.backward()
    for name, prm in wm.named_parameters():
        key = f"oc_{tag}_grad_{name}"
        if key in Z.files:
            np.testing.assert_allclose(prm.grad.detach().numpy(), Z[key], rtol=5e-4, atol=1e-6)
