"""Calibration of the energy terms from ground truth (host side, CPU in the reference as well): writes the
``calibration.json`` the energy setups read.  Mirrors ``models/mpp/calibration/energy_calibration.py:19-185`` and
``metrics/detection.py:8-62`` without the figures."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def f_beta(p, r, beta):
    div = (beta ** 2 * p) + r
    return (1 + beta ** 2) * p * r / div if div > 0 else 0


def _dilate(mask: np.ndarray, iterations: int) -> np.ndarray:
    from scipy.ndimage import binary_dilation
    return binary_dilation(mask, iterations=iterations)


def precision_recall_curve_on_detection_map(detection_map, labels, num_thresholds: int = None, dilation: int = 1,
                                            thresholds=None):
    """``metrics/detection.py:8-62``: pixel-wise precision / recall of ``det > t`` against dilated centres"""
    if thresholds is None:
        assert num_thresholds is not None
        thresholds = np.linspace(0, 1, num_thresholds)
    if not isinstance(detection_map, list):
        detection_map, labels = [detection_map], [labels]
    x, y = [], []
    for d, l in zip(detection_map, labels):
        m = np.zeros(d.shape[:2], dtype=bool)
        centers = np.asarray(l["centers"])
        if len(centers) > 0:
            m[centers[:, 0], centers[:, 1]] = True
            m = _dilate(m, dilation)
        x.append(np.asarray(d).ravel())
        y.append(m.ravel())
    x, y = np.concatenate(x), np.concatenate(y)
    precision, recall = [], []
    n_pos = np.sum(y)
    with np.errstate(divide="ignore", invalid="ignore"):
        for t in thresholds:
            pos = x > t
            tp = np.sum(pos & y)
            fp = np.sum(pos & ~y)
            precision.append(tp / (tp + fp))
            recall.append(tp / n_pos)
        precision, recall = np.array(precision), np.array(recall)
        metrics = {"precision": precision, "recall": recall, "f1": (precision * recall) / (precision + recall)}
    return thresholds, metrics


def calibrate_detection_threshold(detection_maps: List[np.ndarray], labels: List[Dict], target: str = "f1") -> float:
    """``energy_calibration.py:19-77``: threshold of the best F-score (NaN precision where nothing is detected
    compares as in the reference: ``p + r > 0`` is False for NaN, the score is 0)"""
    thresh, metrics = precision_recall_curve_on_detection_map(detection_maps, labels, num_thresholds=100, dilation=2)
    pr = list(zip(metrics["precision"], metrics["recall"]))
    scores = {"f1": [2 * p * r / (p + r) if (p + r) > 0 else 0 for p, r in pr],
              "f2": [f_beta(p, r, 2.0) for p, r in pr], "f0.5": [f_beta(p, r, 0.5) for p, r in pr]}
    return float(thresh[int(np.argmax(scores[target or "f1"]))])


def generate_wrong_value(gt_class_value: int, mapping, min_offset: int, rng: np.random.Generator) -> int:
    """``energy_calibration.py:142-156`` (a class at least ``min_offset`` bins away from the true one)"""
    possible = set(range(mapping.n_classes)) - {gt_class_value}
    for v in range(1, min_offset):
        for o in (v, -v):
            c = gt_class_value + o
            if mapping.is_cyclic:
                c = c % mapping.n_classes
            possible -= {c}
    return int(rng.choice(list(possible)))


def calibrate_param_dists(param_dist_maps: List[List[np.ndarray]], gt_rectangles, mappings, param_names: Sequence[str],
                          rng: np.random.Generator):
    """``energy_calibration.py:80-131``: per mark, a 1-D logistic regression separating the softmax value of the true
    class from the one of a wrong class -> (coefs, intercepts) of the remap -2 sigmoid(c p + i) + 1"""
    from sklearn.linear_model import LogisticRegression
    coefs, intercepts = [], []
    for i_p, (mapping, p_name) in enumerate(zip(mappings, param_names)):
        values, labels = [], []
        for k in range(len(param_dist_maps)):
            for gt in gt_rectangles[k]:
                local = param_dist_maps[k][i_p][gt.x, gt.y]
                cls = mapping.value_to_class(getattr(gt, p_name))
                values.append(local[cls]); labels.append(1)
                values.append(local[generate_wrong_value(cls, mapping, 2, rng)]); labels.append(0)
        clf = LogisticRegression(penalty=None, class_weight="balanced").fit(np.array(values).reshape(-1, 1), np.array(labels))
        coefs.append(float(clf.coef_[0, 0])); intercepts.append(float(clf.intercept_[0]))
    return coefs, intercepts


def calibrate_min_area(gt_configs, quantile: float = 0.01):
    """``energy_calibration.py:159-185``: the 1 % and 99 % quantiles of the ground-truth areas"""
    areas = np.array([p.length * p.width for conf in gt_configs for p in conf], dtype=float)
    return float(np.quantile(areas, quantile)), float(np.quantile(areas, 1 - quantile))


def calibrate_contrast_threshold(contrast_energy_maker, image_configs, rng: np.random.Generator, target: str = "f1",
                                 device: int = 0) -> float:
    """``energy_setup_contrast.py:165-205``: the classic image energy of every ground-truth rectangle against that of
    4x as many random rectangles; the threshold of the best F-score on 100 thresholds between the extreme values, negated.
    The energies are evaluated on the GPU (``energies.classic_values``); the random rectangles consume ``rng`` exactly as
    the reference does (position, size, ratio, angle per rectangle)."""
    from .energies import classic_values
    from .shapes import Rectangle
    x, y = [], []
    for image_data in image_configs:
        term = contrast_energy_maker(image_data, detection_thresh=0.0)
        gt = list(image_data.gt_config)
        H, W = np.asarray(image_data.image).shape[:2]
        rd = []
        for _ in range(4 * len(gt)):
            rd.append(Rectangle(x=int(rng.integers(0, H)), y=int(rng.integers(0, W)), size=float(rng.normal(8, 1.0)),
                                ratio=float(np.clip(rng.normal(0.5, 0.1), 0.1, 1)), angle=float(rng.uniform(0, np.pi))))
        vals = -classic_values(term, gt + rd, getattr(image_data, "mappings", None), device=device)
        x.append(vals)
        y.append(np.array([True] * len(gt) + [False] * len(rd)))
    x, y = np.concatenate(x), np.concatenate(y).astype(bool)
    thresholds = np.linspace(np.min(x), np.max(x), 100)
    precision, recall = [], []
    with np.errstate(divide="ignore", invalid="ignore"):
        for t in thresholds:
            pos = x > t
            tp, fp = np.sum(pos & y), np.sum(pos & ~y)
            precision.append(tp / (tp + fp))
            recall.append(tp / np.sum(y))
    pr = list(zip(precision, recall))
    scores = {"f1": [2 * p * r / (p + r) if (p + r) > 0 else 0 for p, r in pr],
              "f2": [f_beta(p, r, 2.0) for p, r in pr], "f0.5": [f_beta(p, r, 0.5) for p, r in pr]}
    return float(-thresholds[int(np.argmax(scores[target]))])
