"""DOTA task-1 evaluation on the GPU: ``mpp_quad_iou`` against the CPU oracle (independent triangle-fan algorithm),
``dota_eval.voc_eval`` against the oracle's voc_eval on the same files."""
import os

import numpy as np
import pytest

from oracle import voc_eval as V
from test_oracle_voc_eval import rect_quad, write_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mpp_cnn_rs_object_detection_amd import hip_api
    c = hip_api.MppContext(0)
    yield c
    c.close()


def random_quads(rng, n, lo=0, hi=60, length=(3, 14), width=(2, 7)):
    return np.array([rect_quad(*rng.uniform(lo, hi, 2), rng.uniform(*length), rng.uniform(*width), rng.uniform(0, np.pi))
                     for _ in range(n)])


def test_quad_iou_known_answers(ctx):
    sq = [0, 0, 2, 0, 2, 2, 0, 2]
    others = [sq, [0, 2, 2, 2, 2, 0, 0, 0], [1, 0, 3, 0, 3, 2, 1, 2], [50, 50, 60, 50, 60, 60, 50, 60], [1, 0, 2, 1, 1, 2, 0, 1],
              [2, 0, 4, 0, 4, 2, 2, 2]]
    out = ctx.quad_iou([sq], others)[0]
    np.testing.assert_allclose(out, [1.0, 1.0, 1 / 3, -1.0, 0.5, 0.0], atol=1e-12)
    z = [1, 1, 1, 1, 1, 1, 1, 1]
    assert ctx.quad_iou([z], [z])[0, 0] == 1.0                         # the devkit's degenerate-union rule
    assert ctx.quad_iou(np.zeros((0, 8)), [sq]).shape == (0, 1)


def test_quad_iou_matches_the_oracle(ctx):
    rng = np.random.default_rng(5)
    a, b = random_quads(rng, 70), random_quads(rng, 90)
    b[::7] = b[::7].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)          # some clockwise ground truths
    out = ctx.quad_iou(a, b)
    assert out.shape == (70, 90)
    n_pos = 0
    for i in range(70):
        keep = V.hbb_overlaps(a[i], b) > 0
        np.testing.assert_array_equal(out[i] >= 0, keep)               # the axis-aligned pre-filter
        for j in np.where(keep)[0]:
            assert out[i, j] == pytest.approx(V.iou_poly(b[j], a[i]), abs=1e-9)
            n_pos += out[i, j] > 0.05
    assert n_pos > 100


def test_quad_iou_large_is_symmetric_and_bounded(ctx):
    rng = np.random.default_rng(6)
    a = random_quads(rng, 1500, 0, 400)
    out = ctx.quad_iou(a, a)
    assert out.shape == (1500, 1500)
    np.testing.assert_allclose(np.diag(out), 1.0, atol=1e-9)
    np.testing.assert_allclose(out, out.T, atol=1e-9)
    assert out.max() <= 1.0 + 1e-9 and out[out >= 0].min() >= 0.0


def test_voc_eval_equals_the_oracle(ctx, tmp_path):
    from mpp_cnn_rs_object_detection_amd import dota_eval
    rng = np.random.default_rng(11)
    gts, dets = {}, []
    for k in range(4):
        name = f"{k:04}"
        g = np.round(random_quads(rng, 40, 0, 300, (10, 24), (6, 12)))
        gts[name] = [(q, int(rng.random() < 0.15)) for q in g]
        for q in g[rng.random(len(g)) < 0.85]:                        # jittered copies of most objects
            jit = q + np.tile(rng.normal(0, 0.7, 2), 4) + rng.normal(0, 0.2, 8)
            dets.append((name, float(rng.uniform(0.3, 1.0)), jit))
        for q in random_quads(rng, 12, 0, 300):                       # clutter
            dets.append((name, float(rng.uniform(0.0, 0.6)), q))
    det, ann, iset = write_case(tmp_path, gts, dets)
    for thr in (0.05, 0.25, 0.5, 0.75):
        r0, p0, a0 = V.voc_eval(det, ann, iset, "vehicle", ovthresh=thr)
        r1, p1, a1 = dota_eval.voc_eval(det, ann, iset, "vehicle", ovthresh=thr, ctx=ctx)
        np.testing.assert_array_equal(r0, r1)
        np.testing.assert_array_equal(p0, p1)
        assert a0 == a1 and 0.2 < a1 <= 1.0


def test_hand_computed_table(ctx, tmp_path):
    from mpp_cnn_rs_object_detection_amd import dota_eval
    g1, g2, g3 = [0, 0, 10, 0, 10, 4, 0, 4], [20, 20, 30, 20, 30, 24, 20, 24], [50, 50, 60, 50, 60, 54, 50, 54]
    gts = {"0001": [(g1, 0), (g2, 0)], "0002": [(g3, 1)]}
    dets = [("0001", 0.9, g1), ("0001", 0.8, [1, 0, 11, 0, 11, 4, 1, 4]), ("0002", 0.7, g3),
            ("0001", 0.6, [100, 100, 110, 100, 110, 104, 100, 104]), ("0001", 0.5, g2)]
    det, ann, iset = write_case(tmp_path, gts, dets)
    rec, prec, ap = dota_eval.voc_eval(det, ann, iset, "vehicle", ovthresh=0.5, ctx=ctx)
    np.testing.assert_allclose(rec, [0.5, 0.5, 0.5, 0.5, 1.0])
    np.testing.assert_allclose(prec, [1.0, 0.5, 0.5, 1 / 3, 0.5])
    assert ap == pytest.approx(0.75)
