"""CPU oracle of the DOTA task-1 (oriented boxes) evaluation -- TEST INFRASTRUCTURE ONLY.

The reference evaluates through ``data/DOTA_devkit/dota_evaluation_task1.voc_eval`` (call site
``metrics/dota_eval.py:37-47``, ``use_07_metric=False``, IoU thresholds 0.05..0.75) and the devkit's SWIG module
``polyiou``.  The devkit is a clone the README asks the user to make (``README.md:22-30``, CAPTAIN-WHU/DOTA_devkit,
no pinned revision) and is NOT in the container: **parity unpinned** -- this file restates the published algorithm
(voc_eval / voc_ap / parse_gt of ``dota_evaluation_task1.py`` and ``polyiou.cpp``'s triangle-fan intersection) in plain
Python/NumPy; it is anchored by known answers (tests/test_oracle_voc_eval.py), not by an execution of the devkit.

Only tests/, ``__graft_entry__.smoke()`` and bench's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-8


# ---- polyiou.cpp -----------------------------------------------------------------------------------------
def _sig(d):
    return int(d > EPS) - int(d < -EPS)


def _cross(o, a, b):
    return (a[0] - o[0]) * (b[1] - o[1]) - (b[0] - o[0]) * (a[1] - o[1])


def _area(ps):
    res = 0.0
    n = len(ps)
    for i in range(n):
        res += ps[i][0] * ps[(i + 1) % n][1] - ps[i][1] * ps[(i + 1) % n][0]
    return res / 2.0


def _line_cross(a, b, c, d):
    s1 = _cross(a, b, c)
    s2 = _cross(a, b, d)
    if _sig(s1) == 0 and _sig(s2) == 0:
        return 2, None
    if _sig(s2 - s1) == 0:
        return 0, None
    return 1, ((c[0] * s2 - d[0] * s1) / (s2 - s1), (c[1] * s2 - d[1] * s1) / (s2 - s1))


def _polygon_cut(p, a, b):
    """keep the part of polygon p on the left of a -> b"""
    out = []
    n = len(p)
    for i in range(n):
        cur, nxt, prv = p[i], p[(i + 1) % n], p[(i - 1) % n]
        if _sig(_cross(a, b, cur)) > 0:
            out.append(cur)
        if _sig(_cross(a, b, cur)) != _sig(_cross(a, b, nxt)):
            code, pt = _line_cross(a, b, cur, nxt)
            if code == 1:
                out.append(pt)
    # drop consecutive duplicates, as the original does
    res = []
    for q in out:
        if not res or not (_sig(q[0] - res[-1][0]) == 0 and _sig(q[1] - res[-1][1]) == 0):
            res.append(q)
    while len(res) > 1 and _sig(res[0][0] - res[-1][0]) == 0 and _sig(res[0][1] - res[-1][1]) == 0:
        res.pop()
    return res


def _tri_intersect(a, b, c, d):
    o = (0.0, 0.0)
    s1, s2 = _sig(_cross(o, a, b)), _sig(_cross(o, c, d))
    if s1 == 0 or s2 == 0:
        return 0.0
    if s1 == -1:
        a, b = b, a
    if s2 == -1:
        c, d = d, c
    p = [o, a, b]
    p = _polygon_cut(p, o, c)
    p = _polygon_cut(p, c, d)
    p = _polygon_cut(p, d, o)
    res = abs(_area(p)) if len(p) >= 3 else 0.0
    return -res if s1 * s2 == -1 else res


def intersect_area(ps1, ps2):
    ps1, ps2 = [tuple(p) for p in ps1], [tuple(p) for p in ps2]
    if _area(ps1) < 0:
        ps1 = ps1[::-1]
    if _area(ps2) < 0:
        ps2 = ps2[::-1]
    res = 0.0
    for i in range(len(ps1)):
        for j in range(len(ps2)):
            res += _tri_intersect(ps1[i], ps1[(i + 1) % len(ps1)], ps2[j], ps2[(j + 1) % len(ps2)])
    return res


def iou_poly(p, q) -> float:
    """polyiou.iou_poly on two 8-vectors x1 y1 .. x4 y4"""
    ps = [(p[2 * i], p[2 * i + 1]) for i in range(4)]
    qs = [(q[2 * i], q[2 * i + 1]) for i in range(4)]
    inter = intersect_area(ps, qs)
    union = abs(_area(ps)) + abs(_area(qs)) - inter
    if union == 0:
        return (inter + 1.0) / (union + 1.0)
    return inter / union


def hbb_overlaps(bb, BBGT):
    """the axis-aligned pre-filter of voc_eval (inclusive-pixel extents)"""
    BBGT = np.asarray(BBGT, dtype=float).reshape(-1, 8)
    bx0, by0, bx1, by1 = np.min(bb[0::2]), np.min(bb[1::2]), np.max(bb[0::2]), np.max(bb[1::2])
    gx0, gy0 = np.min(BBGT[:, 0::2], axis=1), np.min(BBGT[:, 1::2], axis=1)
    gx1, gy1 = np.max(BBGT[:, 0::2], axis=1), np.max(BBGT[:, 1::2], axis=1)
    iw = np.maximum(np.minimum(gx1, bx1) - np.maximum(gx0, bx0) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gy1, by1) - np.maximum(gy0, by0) + 1.0, 0.0)
    inters = iw * ih
    uni = (bx1 - bx0 + 1.0) * (by1 - by0 + 1.0) + (gx1 - gx0 + 1.0) * (gy1 - gy0 + 1.0) - inters
    return inters / uni


# ---- dota_evaluation_task1.py ------------------------------------------------------------------------------
def parse_gt(filename):
    objects = []
    with open(filename) as f:
        for line in f:
            parts = line.strip().split(" ")
            if len(parts) < 9:
                continue
            objects.append({"name": parts[8], "difficult": int(parts[9]) if len(parts) > 9 else 0,
                            "bbox": [float(v) for v in parts[:8]]})
    return objects


def voc_ap(rec, prec, use_07_metric=False) -> float:
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=0.5, use_07_metric=False):
    with open(imagesetfile) as f:
        imagenames = [x.strip() for x in f.readlines() if x.strip()]
    class_recs, npos = {}, 0
    for name in imagenames:
        R = [o for o in parse_gt(annopath.format(name)) if o["name"] == classname]
        bbox = np.array([o["bbox"] for o in R], dtype=float).reshape(-1, 8)
        difficult = np.array([o["difficult"] for o in R]).astype(bool)
        npos += int(np.sum(~difficult))
        class_recs[name] = {"bbox": bbox, "difficult": difficult, "det": [False] * len(R)}
    with open(detpath.format(classname)) as f:
        splitlines = [x.strip().split(" ") for x in f.readlines() if x.strip()]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines]).reshape(-1, 8)
    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :]
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        bb = BB[d, :]
        ovmax, jmax = -np.inf, -1
        BBGT = R["bbox"]
        if BBGT.size > 0:
            keep = np.where(hbb_overlaps(bb, BBGT) > 0)[0]
            if len(keep) > 0:
                overlaps = np.array([iou_poly(BBGT[k], bb) for k in keep])
                ovmax = float(np.max(overlaps))
                jmax = int(keep[int(np.argmax(overlaps))])
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
    return rec, prec, voc_ap(rec, prec, use_07_metric)
