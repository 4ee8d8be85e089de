"""Host-side logic that needs no GPU: schedule arithmetic, perturbation sampling (against vectors
recorded from the reference), aggregation, kernel mixture, energy-setup tables, tile sharding."""
import numpy as np
import pytest

from helpers import GOLDEN, hrc_model, log_model
from mpp_cnn_rs_object_detection_amd import distributed as mdist
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import kernels, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps, Perturbation
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle, rect_to_poly, sra_to_wla, wla_to_sra


def test_schedule_matches_reference_arithmetic():
    from mpp_cnn_rs_object_detection_amd.sampler import resolve_schedule
    # mpp_hrcM: burn_in 30000, interval 128, one sample -> max_iter 30256, 30257 steps; the returned state is
    # the one after step 30208 = 236*128, NOT the last one
    alpha, Tt, total, snaps = resolve_schedule(1, 1.0, 0.999, 30000, 128, 0.0)
    assert (alpha, Tt, total) == (0.999, 0.0, 30257)
    assert snaps == [30080, 30208]
    # mpp_log: interval 1 -> samples at 30000, 30001, 30002; 30003 steps
    alpha, Tt, total, snaps = resolve_schedule(1, 1.0, 0.999, 30000, 1, 0.0)
    assert total == 30003 and snaps[-1] == 30002
    alpha, Tt, total, snaps = resolve_schedule(1, 1.0, "auto", 1000, 10, 1e-3)
    assert Tt == 0 and alpha == pytest.approx((1e-3) ** (1 / 1000))
    alpha, Tt, total, snaps = resolve_schedule(1, 1.0, 0.99, 100, 10, 0.0, iter_multiplier=2)
    assert alpha == pytest.approx(0.99 ** 0.5) and total == 200 + 2 * 20 + 1


def test_sample_perturbations_matches_reference_vectors():
    from mpp_cnn_rs_object_detection_amd import perturbation_sampler as ps
    z = np.load(f"{GOLDEN}/perturbations_golden.npz")
    gt = [Rectangle(int(x), int(y), size=float(m[0]), ratio=float(m[1]), angle=float(m[2]))
          for (x, y), m in zip(z["gt_xy"], z["gt_marks"])]
    data = ImageWMaps(name="0", shape=(96, 96), image=None, detection_map=None, param_dist_maps=None,
                      mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=gt)
    for name in ("PERTURBATION_LIGHT", "PERTURBATION_MEDIUM", "PERTURBATION_MEDIUM_OVERLAP", "PERTURBATION_STRONG"):
        rng = np.random.default_rng(123)
        res = ps.sample_perturbations(image_data=data, rng=rng, n_samples=3, **getattr(ps, name))
        assert [len(r) for r in res] == list(z[name + "_len"])
        flat = np.array([p.as_row() for r in res for p in r], dtype=float).reshape(-1, 5)
        np.testing.assert_array_equal(flat, z[name + "_flat"])
        assert rng.random() == float(z[name + "_next"])


def test_aggregate_perturbations():
    from mpp_cnn_rs_object_detection_amd.perturbation_sampler import aggregate_perturbations
    a, b, c = (Rectangle(1, 1, 1, 1, 0), Rectangle(2, 2, 1, 1, 0), Rectangle(3, 3, 1, 1, 0))
    agg = aggregate_perturbations([Perturbation(None, addition=a), Perturbation(None, removal=b),
                                   Perturbation(None, removal=a), Perturbation(None, removal=c, addition=b)])
    assert agg.addition == [] and agg.removal == [c]


def test_kernel_mixture_is_the_reference_one():
    k = kernels.make_kernels(mappings.default_mappings(), 7)
    np.testing.assert_allclose(k.p_kernel, [1 / 18, 1 / 18, 1 / 9, 1 / 9, 1 / 9, 2 / 9, 1 / 9, 2 / 9, 0, 0], rtol=1e-15)
    assert k.intensity == 7 and k.max_delta == 8 and k.sigma_trans == 2.0 and k.sigma_transform == 0.1
    # use_split_merge: four top-level branches of weight 1 (make_kernels.py:75-77), split and merge half of theirs each
    ks = kernels.make_kernels(mappings.default_mappings(), 1, use_split_merge=True)
    np.testing.assert_allclose(ks.p_kernel, [1 / 24, 1 / 24, 1 / 12, 1 / 12, 1 / 12, 1 / 6, 1 / 12, 1 / 6, 1 / 8, 1 / 8],
                               rtol=1e-15)
    assert ks.split_radius == 16.0 and ks.split_sigma == 0.1


def test_energy_tables():
    setup, comb = hrc_model()
    unit, pair = setup.make_energies()
    d = E.build_model_desc(unit, pair, comb)
    assert d.names == ["PositionEnergy", "ShapeEnergy", "AreaPriorEnergy", "RectangleOverlapEnergy",
                       "ShapeAlignmentEnergy"]
    assert d.gate_term == 0 and d.combinator == E.C_LINEAR
    coef = [u[2] for u in d.unit] + [p[3] for p in d.pair]
    np.testing.assert_allclose(coef, [0.5 * 0.8, 0.5 * 0.2, 0.5 * 0.2 / 0.85, 0.5 * 0.6 / 0.85, 0.5 * 0.05 / 0.85])
    assert setup.detection_threshold == pytest.approx(0.6464646464646465)
    setup, comb = log_model()
    unit, pair = setup.make_energies()
    d = E.build_model_desc(unit, pair, comb)
    assert d.combinator == E.C_LOGISTIC and d.gate_term == -1
    assert d.lin0 == pytest.approx(8 * 0.7927545309066772)       # the bias enters once per term
    assert setup.detection_threshold == 0.5
    # the host-side combinators compute what the flattened tables describe
    vec = {n: [0.3 * (i + 1), -0.2 * i] for i, n in enumerate(setup.energy_names)}
    lin = d.lin0 + sum(c * np.array(vec[n]) for n, c in zip(d.names, [u[2] for u in d.unit] + [p[3] for p in d.pair]))
    assert comb.compute(vec) == pytest.approx(float(np.sum(2 / (1 + np.exp(-lin)) - 1)))


def test_shapes_and_mappings():
    r = Rectangle(10, 20, size=6.0, ratio=0.5, angle=0.3)
    assert r.length == pytest.approx(8.0) and r.width == pytest.approx(4.0)
    poly = r.poly_coord
    x, y = poly[:, 0], poly[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    assert area == pytest.approx(32.0)
    assert wla_to_sra(*sra_to_wla(6.0, 0.5, 0.3)) == pytest.approx((6.0, 0.5, 0.3))
    assert rect_to_poly((0, 0), 2, 4, 0.0).tolist() == [[1, 2], [1, -2], [-1, -2], [-1, 2]]
    m = mappings.default_mappings()[2]
    assert m.value_to_class(m.class_to_value(17)) == 17
    assert m.clip(np.pi + 0.1) == pytest.approx(0.1)
    assert mappings.default_mappings()[0].clip(40.0) == 32.0


def test_tile_sharding_and_detection_packing():
    # contiguous blocks (a rank's tiles form one compact image region), sizes differing by at most one
    assert mdist.shard_tiles(10, 1, 4) == [2, 3, 4] and mdist.shard_tiles(9, 0, 2) == [0, 1, 2, 3]
    assert sum([mdist.shard_tiles(16, r, 8) for r in range(8)], []) == list(range(16))
    assert mdist.shard_tiles(3, 0, 8) == [] and mdist.shard_tiles(3, 5, 8) == [1] and sum([mdist.shard_tiles(3, r, 8) for r in range(8)], []) == [0, 1, 2]
    assert mdist.tile_owner(9, 2).tolist() == [0, 0, 0, 0, 1, 1, 1, 1, 1]
    # the gather buffer has the same capacity on every rank, whatever the rank owns (9 tiles on 2 / 8 ranks)
    assert mdist.gather_capacity(9, 2) == 5 * 1024 and mdist.gather_capacity(9, 8) == 2 * 1024
    pts = [(np.array([[1, 2], [3, 4]]), np.array([[5., .5, 1.], [6., .6, 2.]])), (np.zeros((0, 2)), np.zeros((0, 3)))]
    buf = mdist.pack_detections([3, 7], pts, [np.array([.9, .8]), np.zeros(0)], capacity=8)
    rec = mdist.unpack_detections(buf[None])
    assert rec.shape == (2, 7) and rec[0].tolist() == [3, 1, 2, 5, .5, 1, .9]
    with pytest.raises(ValueError):
        mdist.pack_detections([0], [pts[0]], [None], capacity=1)


def test_get_neighbors_like_the_reference_tests():
    """test/test_points_set.py:187-245 (exact Euclidean neighbours, the point itself excluded)"""
    from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
    pts = [Rectangle(10, 20, 5, 0.5, 0), Rectangle(10, 24, 5, 0.5, 0), Rectangle(14, 21, 5, 0.5, 0),
           Rectangle(20, 20, 5, 0.5, 0), Rectangle(20, 200, 5, 0.5, 0)]

    class _Set:                                  # EPointsSet.get_neighbors without a GPU context
        from mpp_cnn_rs_object_detection_amd.point_set import EPointsSet as _E
        get_neighbors = _E.get_neighbors
        _points = pts
    s = _Set()
    near5 = s.get_neighbors(pts[0], radius=5)
    assert pts[0] not in near5 and pts[1] in near5 and pts[2] in near5 and pts[3] not in near5 and pts[4] not in near5
    near12 = s.get_neighbors(pts[0], radius=12)
    assert pts[1] in near12 and pts[2] in near12 and pts[3] in near12 and pts[4] not in near12
    rng = np.random.default_rng(0)
    cloud = [Rectangle(int(rng.integers(0, 127)), int(rng.integers(0, 127)), 5, 0.5, 0) for _ in range(200)]
    s._points = cloud
    for r in (8, 64):
        for p1 in cloud[:40]:
            neigh = s.get_neighbors(p1, radius=r)
            for p2 in cloud:
                if p1 is not p2:
                    assert (np.hypot(p1.x - p2.x, p1.y - p2.y) <= r) == (p2 in neigh)


def test_torch_divergence_equals_the_numpy_divergence():
    """test/test_torch_div.py:9-46: the 'ij' divergence with torch.gradient against np.gradient, mean error < 1e-8"""
    import torch
    from mpp_cnn_rs_object_detection_amd import unet
    rng = np.random.default_rng(0)
    a = (rng.random((20, 20, 2)) - 0.5) * 2
    a[10:, :, :] = 0
    a[:, 5:7, 0] = 0; a[:, 5:7, 1] = 1
    a[:, 7:9, 0] = 0; a[:, 7:9, 1] = -1
    f = np.moveaxis(a, 2, 0)
    div_np = np.gradient(f[0], 1.0, axis=0) + np.gradient(f[1], 1.0, axis=1)        # utils/math_utils.py:24-25
    div_t = unet.torch_divergence_ij(torch.tensor(f))
    assert tuple(div_t.shape) == (20, 20)
    assert np.mean(np.abs(div_np - div_t.numpy())) < 1e-8


def test_labels_to_rectangles_on_the_reference_data_sample():
    """data_sample/DOTA_gsd50/val/2781 (the reference's own fixture): (a, b, angle) annotations -> Rectangle, against the
    rectangles recorded from the reference's labels_to_rectangles (models/mpp/data_loaders.py:254-262)"""
    import os
    from mpp_cnn_rs_object_detection_amd.data_loaders import labels_to_rectangles
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dota_2781.npz"))
    labels = {"centers": z["centers"], "parameters": z["parameters"], "difficult": z["difficult"], "categories": z["categories"]}
    rects = labels_to_rectangles(labels)
    got = np.array([[r.x, r.y, r.size, r.ratio, r.angle] for r in rects])
    assert got.shape == (272, 5)
    np.testing.assert_allclose(got, z["ref_rects"], rtol=1e-13, atol=1e-13)
    assert (got[:, 3] > 0).all() and (got[:, 3] <= 1).all() and (got[:, 4] >= 0).all() and (got[:, 4] < np.pi).all()


def test_choose_spec_waves_follows_the_measured_optimum():
    """The launch policy (speculative waves per chain from the number of tiles and a chain's LDS footprint) against the
    sweep in profiles/r01_batched_sweep.json: with 1024 slots one chain fills a CU's LDS -> always 8 waves; with 128 slots
    8 waves up to 256 tiles, 4 up to 512, 2 up to 1024, 1 beyond.  Split / merge kernels exist for 1 and 8 waves only."""
    from mpp_cnn_rs_object_detection_amd.sampler import choose_spec_waves

    class FakeCtx:
        def __init__(self, lds_of):
            self.lds_of, self.spec = lds_of, 8

        def set_option(self, name, v):
            assert name == "spec_waves"
            self.spec = v

        def get_option(self, name):
            assert name == "lds_bytes"
            return self.lds_of(self.spec)

    big = FakeCtx(lambda spec: 100 * 1024 + 1024 * spec)          # point_capacity 1024
    small = FakeCtx(lambda spec: 16 * 1024 + 1024 * spec)         # point_capacity 128
    for tiles in (1, 64, 256, 512, 1024, 4096):
        assert choose_spec_waves(big, tiles) == 8
    assert [choose_spec_waves(small, t) for t in (1, 256, 257, 512, 513, 1024, 1025, 2048, 4096, 16384)] == \
        [8, 8, 4, 4, 2, 2, 1, 1, 1, 1]
    assert [choose_spec_waves(small, t, use_split_merge=True) for t in (256, 512, 2048)] == [8, 8, 1]


def test_distance_merge_tolerates_non_finite_scores():
    """merge_patches' dedupe rule with NaN / inf Papangelou intensities (contrast setups can produce them): no
    exception, and the reference's np.argmax choice (data_loaders.py:151)."""
    from mpp_cnn_rs_object_detection_amd.data_loaders import distance_merge
    xy = np.array([[10, 10], [11, 10], [10, 12], [100, 100]], dtype=float)
    for sc, keep in (([np.nan, 1.0, 2.0, 0.5], 0), ([np.inf, 1.0, 2.0, 0.5], 0), ([1.0, np.inf, np.inf, 0.5], 1),
                     ([1.0, 2.0, np.nan, 0.5], 2), ([-np.inf, -np.inf, -np.inf, 0.5], 0), ([1.0, 3.0, 2.0, 0.5], 1)):
        removed = distance_merge(xy, np.array(sc), 3.0)
        assert not removed[3]
        assert list(np.nonzero(~removed[:3])[0]) == [keep], (sc, removed)
    # ties within 1e-9: the first point wins
    removed = distance_merge(xy, np.array([1.0, 1.0 + 1e-12, 1.0 - 1e-12, 0.0]), 3.0)
    assert list(np.nonzero(~removed)[0]) == [0, 3]


def test_stack_tiles_equals_tile_by_tile_crops():
    """``data_loaders.stack_tiles``: the maps of all tiles of an image by one strided copy per map must be what
    ``crop_image_w_maps`` cuts tile by tile (reference data_loaders.py:74-119 with the anchors of mpp_model.py:233-240) --
    overlapping tiles included; anchors that are not on a regular grid, labelled images and numpy maps decline (None)."""
    import torch
    from mpp_cnn_rs_object_detection_amd import mappings
    from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
    from mpp_cnn_rs_object_detection_amd.data_loaders import crop_image_w_maps, stack_tiles, tile_anchors
    from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
    g = torch.Generator().manual_seed(0)
    for H, W, p in ((96, 160, 32), (100, 100, 48), (64, 64, 64)):
        det = torch.rand((H, W), generator=g)
        marks = [torch.rand((H, W, 32), generator=g) for _ in range(3)]
        data = ImageWMaps(name="0", shape=(H, W), image=None, detection_map=det, param_dist_maps=marks,
                          mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
        anchors = tile_anchors((H, W), p)
        got = stack_tiles(data, anchors, p, require_cuda=False)
        assert got is not None and tuple(got[0].shape) == (len(anchors), p, p)
        for k, a in enumerate(anchors):
            t = crop_image_w_maps(data, a, p)
            assert torch.equal(got[0][k], t.detection_map)
            for m in range(3):
                assert torch.equal(got[1][m][k], t.param_dist_maps[m])
        assert stack_tiles(data, anchors, p) is None                                     # host tensors: the caller crops
    irregular = [np.array([0, 0]), np.array([0, 33]), np.array([0, 64]), np.array([31, 0]), np.array([31, 33]), np.array([31, 64])]
    assert stack_tiles(data, irregular, 16, require_cuda=False) is None
    assert stack_tiles(data, [np.array([0, 0]), np.array([0, 60])], 32, require_cuda=False) is None     # beyond the image
    data.labels = {"centers": np.zeros((0, 2))}
    assert stack_tiles(data, tile_anchors((64, 64), 32), 32, require_cuda=False) is None
    data.labels = None
    data.detection_map = det.numpy()
    assert stack_tiles(data, tile_anchors((64, 64), 32), 32, require_cuda=False) is None
