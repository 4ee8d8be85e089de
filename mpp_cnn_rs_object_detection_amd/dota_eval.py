"""DOTA task-1 (oriented boxes) evaluation of the detection files written by ``dota_results.DOTAResultsTranslator``.

Mirror of the reference's ``metrics/dota_eval.py:16-87`` (same arguments, same ``metricsX.XX.json`` /
``prec_rec_curve_X.XX.png`` outputs, IoU thresholds 0.05, 0.1, 0.25, 0.5, 0.75, class ``vehicle``,
``use_07_metric=False``).  Upstream delegates to ``data/DOTA_devkit/dota_evaluation_task1.voc_eval`` and the SWIG
module ``polyiou`` of an un-vendored clone (``README.md:22-30``); here ``voc_eval`` keeps the devkit's signature and
matching rule, and the rotated IoUs of one image come from ONE launch of ``mpp_quad_iou`` (``csrc/mpp_metrics.hip``)
-- there is no CPU fallback: without the HIP library this module raises.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np

from . import hip_api
from .paths import get_inference_path

IOU_THRESHOLDS = [0.05, 0.1, 0.25, 0.5, 0.75]


def parse_gt(filename: str) -> List[dict]:
    """``x1 y1 x2 y2 x3 y3 x4 y4 class difficult`` per line (dota_evaluation_task1.parse_gt)"""
    objects = []
    with open(filename) as f:
        for line in f:
            parts = line.strip().split(" ")
            if len(parts) < 9:
                continue
            objects.append({"name": parts[8], "difficult": int(parts[9]) if len(parts) > 9 else 0,
                            "bbox": [float(v) for v in parts[:8]]})
    return objects


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.0
        return float(ap)
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def _convex(q: np.ndarray) -> np.ndarray:
    """per quad: are the four turns of one sign (or degenerate)?  The device clipper needs a convex clip polygon."""
    p = q.reshape(-1, 4, 2)
    e = np.roll(p, -1, axis=1) - p
    cr = e[:, :, 0] * np.roll(e, -1, axis=1)[:, :, 1] - e[:, :, 1] * np.roll(e, -1, axis=1)[:, :, 0]
    return np.all(cr >= -1e-9, axis=1) | np.all(cr <= 1e-9, axis=1)


def voc_eval(detpath: str, annopath: str, imagesetfile: str, classname: str, ovthresh: float = 0.5,
             use_07_metric: bool = False, ctx: Optional["hip_api.MppContext"] = None, device: int = 0):
    """rec, prec, ap = voc_eval(...) -- arguments and matching rule of the devkit's task-1 ``voc_eval``."""
    own = ctx is None
    if own:
        ctx = hip_api.MppContext(device)
    with open(imagesetfile) as f:
        imagenames = [x.strip() for x in f.readlines() if x.strip()]
    class_recs: Dict[str, dict] = {}
    npos = 0
    for name in imagenames:
        R = [o for o in parse_gt(annopath.format(name)) if o["name"] == classname]
        bbox = np.array([o["bbox"] for o in R], dtype=np.float64).reshape(-1, 8)
        difficult = np.array([o["difficult"] for o in R]).astype(bool)
        if not _convex(bbox).all():
            raise ValueError(f"{name}: non-convex ground-truth quadrilateral (not supported by the device clipper)")
        npos += int(np.sum(~difficult))
        class_recs[name] = {"bbox": bbox, "difficult": difficult, "det": np.zeros(len(R), bool)}
    with open(detpath.format(classname)) as f:
        splitlines = [x.strip().split(" ") for x in f.readlines() if x.strip()]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines], dtype=np.float64).reshape(-1, 8)
    if not _convex(BB).all():
        raise ValueError("non-convex detection quadrilateral")
    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :]
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)
    # one IoU matrix per image: rows = this image's detections in score order
    rows_of: Dict[str, List[int]] = {}
    for d, name in enumerate(image_ids):
        rows_of.setdefault(name, []).append(d)
    iou_row: Dict[int, np.ndarray] = {}
    for name, rows in rows_of.items():
        gt = class_recs[name]["bbox"]
        if len(gt) == 0:
            continue
        mat = ctx.quad_iou(BB[rows], gt)
        for r, d in enumerate(rows):
            iou_row[d] = mat[r]
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        ovmax, jmax = -np.inf, -1
        row = iou_row.get(d)
        if row is not None and np.any(row >= 0.0):        # -1 marks pairs dropped by the axis-aligned pre-filter
            jmax = int(np.argmax(row))
            ovmax = float(row[jmax])
        if ovmax > ovthresh:
            if not R["difficult"][jmax]:
                if not R["det"][jmax]:
                    tp[d] = 1.0
                    R["det"][jmax] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else tp * 0.0
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = voc_ap(rec, prec, use_07_metric)
    if own:
        ctx.close()
    return rec, prec, ap


def dota_eval(model_dir: str, dataset: str, subset: str, det_type: str, postfix: str = "", device: int = 0) -> dict:
    assert det_type in ["obb", "hbb"]
    if det_type != "obb":
        raise NotImplementedError("only the oriented-box task is evaluated (the reference's hbb branch is marked broken, "
                                  "metrics/dota_eval.py:49)")
    model_name = os.path.split(model_dir.rstrip("/"))[1]
    dota_files_path = os.path.join(get_inference_path(model_name=model_name, dataset=dataset, subset=subset), "dota" + postfix)
    det_path = os.path.join(dota_files_path, "det", r"{:s}.txt")
    annot_path = os.path.join(dota_files_path, "gt", r"{:s}.txt")
    image_set_file = os.path.join(dota_files_path, "imageSet.txt")
    classnames = ["vehicle"]
    ctx = hip_api.MppContext(device)
    summary = {}
    for iou_t in IOU_THRESHOLDS:
        print(f"IOU thresh = {iou_t}")
        results, classaps = {}, []
        for classname in classnames:
            rec, prec, ap = voc_eval(det_path, annot_path, image_set_file, classname, ovthresh=iou_t,
                                     use_07_metric=False, ctx=ctx)
            classaps.append(ap)
            print(f"ap : {ap}")
            results[classname] = {"ap": ap, "precision": prec.tolist(), "recall": rec.tolist()}
            try:
                from matplotlib import pyplot as plt
                plt.figure(figsize=(8, 4))
                plt.xlabel("recall")
                plt.ylabel("precision")
                plt.plot(rec, prec)
                plt.savefig(os.path.join(dota_files_path, f"prec_rec_curve_{iou_t:.2f}.png"))
                plt.close("all")
            except Exception as e:      # figures are a convenience, as upstream
                print("error occurred while saving the figure", e)
        print("map:", float(np.mean(classaps)))
        with open(os.path.join(dota_files_path, f"metrics{iou_t:.2f}.json"), "w") as f:
            json.dump(results, f, indent=1)
        summary[iou_t] = float(np.mean(classaps))
    ctx.close()
    return summary
