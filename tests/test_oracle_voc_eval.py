"""Known answers for the CPU oracle of the DOTA task-1 evaluation (oracle/voc_eval.py).

The devkit the reference calls (metrics/dota_eval.py:37-47) is not in the container, so the oracle is anchored by
geometry whose answer is known in closed form and by hand-computed precision/recall tables."""
import os

import numpy as np
import pytest

from oracle import voc_eval as V


def rect_quad(cx, cy, length, width, angle):
    c, s = np.cos(angle), np.sin(angle)
    pts = []
    for sx, sy in ((1, 1), (-1, 1), (-1, -1), (1, -1)):
        vx, vy = sx * length / 2, sy * width / 2
        pts += [cx + c * vx - s * vy, cy + s * vx + c * vy]
    return pts


def test_iou_known_answers():
    sq = [0, 0, 2, 0, 2, 2, 0, 2]
    assert V.iou_poly(sq, sq) == pytest.approx(1.0, abs=1e-12)
    assert V.iou_poly(sq, [0, 2, 2, 2, 2, 0, 0, 0]) == pytest.approx(1.0, abs=1e-12)              # clockwise copy
    assert V.iou_poly(sq, [1, 0, 3, 0, 3, 2, 1, 2]) == pytest.approx(1.0 / 3.0, abs=1e-12)        # half shift
    assert V.iou_poly(sq, [5, 5, 6, 5, 6, 6, 5, 6]) == 0.0                                        # disjoint
    assert V.iou_poly(sq, [1, 0, 2, 1, 1, 2, 0, 1]) == pytest.approx(0.5, abs=1e-12)              # inscribed diamond
    assert V.iou_poly(sq, [2, 0, 4, 0, 4, 2, 2, 2]) == pytest.approx(0.0, abs=1e-12)              # shared edge
    # two equal 4 x 2 rectangles crossed at 90 degrees: intersection 2 x 2
    a, b = rect_quad(10, 10, 4, 2, 0.0), rect_quad(10, 10, 4, 2, np.pi / 2)
    assert V.iou_poly(a, b) == pytest.approx(4.0 / (8 + 8 - 4), abs=1e-9)
    # contained: 2 x 1 inside 6 x 4, same angle
    a, b = rect_quad(3, 4, 6, 4, 0.3), rect_quad(3, 4, 2, 1, 0.3)
    assert V.iou_poly(a, b) == pytest.approx(2.0 / 24.0, abs=1e-9)
    # degenerate union (both zero-area at the same place): the devkit's (0+1)/(0+1)
    z = [1, 1, 1, 1, 1, 1, 1, 1]
    assert V.iou_poly(z, z) == 1.0


def test_iou_is_symmetric_and_rotation_invariant():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rect_quad(*rng.uniform(20, 30, 2), rng.uniform(4, 12), rng.uniform(2, 6), rng.uniform(0, np.pi))
        b = rect_quad(*rng.uniform(20, 30, 2), rng.uniform(4, 12), rng.uniform(2, 6), rng.uniform(0, np.pi))
        v = V.iou_poly(a, b)
        assert -1e-9 <= v <= 1.0 + 1e-9          # the triangle-fan sum leaves rounding residue on disjoint pairs
        assert V.iou_poly(b, a) == pytest.approx(v, abs=1e-9)
        th = rng.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        ra = (np.array(a).reshape(4, 2) @ R.T).ravel()
        rb = (np.array(b).reshape(4, 2) @ R.T).ravel()
        assert V.iou_poly(ra, rb) == pytest.approx(v, abs=1e-9)


def test_voc_ap_area_and_11_point():
    rec = np.array([0.25, 0.5, 0.5, 0.75, 1.0])
    prec = np.array([1.0, 1.0, 2 / 3, 0.75, 0.8])
    # monotone envelope: [1, 1, .8, .8, .8] -> 0.25*1 + 0.25*1 + 0.25*0.8 + 0.25*0.8
    assert V.voc_ap(rec, prec, False) == pytest.approx(0.9)
    assert V.voc_ap(rec, prec, True) == pytest.approx((6 * 1.0 + 5 * 0.8) / 11.0)
    assert V.voc_ap(np.array([]), np.array([]), False) == 0.0


def write_case(tmp_path, gts, dets):
    os.makedirs(tmp_path / "gt"), os.makedirs(tmp_path / "det")
    names = sorted(gts)
    (tmp_path / "imageSet.txt").write_text("\n".join(names))
    for n in names:
        (tmp_path / "gt" / f"{n}.txt").write_text(
            "\n".join(" ".join(str(int(v)) for v in q) + f" vehicle {d}" for q, d in gts[n]))
    (tmp_path / "det" / "vehicle.txt").write_text(
        "\n".join(f"{n} {s} " + " ".join(f"{v:.1f}" for v in q) for n, s, q in dets))
    return str(tmp_path / "det" / "{:s}.txt"), str(tmp_path / "gt" / "{:s}.txt"), str(tmp_path / "imageSet.txt")


def test_voc_eval_hand_computed(tmp_path):
    g1, g2, g3 = [0, 0, 10, 0, 10, 4, 0, 4], [20, 20, 30, 20, 30, 24, 20, 24], [50, 50, 60, 50, 60, 54, 50, 54]
    gts = {"0001": [(g1, 0), (g2, 0)], "0002": [(g3, 1)]}          # g3 is "difficult"
    dets = [("0001", 0.9, g1),                                       # tp
            ("0001", 0.8, [1, 0, 11, 0, 11, 4, 1, 4]),               # duplicate of g1 (IoU 9/11) -> fp
            ("0002", 0.7, g3),                                       # difficult -> ignored
            ("0001", 0.6, [100, 100, 110, 100, 110, 104, 100, 104]),  # nothing there -> fp
            ("0001", 0.5, g2)]                                       # tp
    det, ann, iset = write_case(tmp_path, gts, dets)
    rec, prec, ap = V.voc_eval(det, ann, iset, "vehicle", ovthresh=0.5)
    np.testing.assert_allclose(rec, [0.5, 0.5, 0.5, 0.5, 1.0])
    np.testing.assert_allclose(prec, [1.0, 0.5, 0.5, 1 / 3, 0.5])
    assert ap == pytest.approx(0.5 * 1.0 + 0.5 * 0.5)
    # a stricter threshold turns the shifted duplicate into an unmatched detection all the same (still fp)
    assert V.voc_eval(det, ann, iset, "vehicle", ovthresh=0.9)[2] == pytest.approx(ap)
