"""DOTA task-1 text files for detections and ground truth.

Same files and line format as the reference's ``metrics/dota_results_translator.py:10-80``
(``<results>/dota<postfix>/{det/<class>.txt, gt/NNNN.txt, imageSet.txt}``), without the
un-vendored DOTA_devkit: the only thing taken from it there is the bounding box of a polygon.
"""
from __future__ import annotations

import os
from typing import List

import numpy as np


class DOTAResultsTranslator:
    def __init__(self, dataset: str, subset: str, results_dir: str, det_type: str, all_classes: List[str],
                 postfix: str = ""):
        assert det_type in ["obb", "hbb"]
        self.det_type = det_type
        self.det_dir = os.path.join(results_dir, "dota" + postfix, "det")
        self.annot_dir = os.path.join(results_dir, "dota" + postfix, "gt")
        self.image_set: List[str] = []
        self.image_set_file = os.path.join(results_dir, "dota" + postfix, "imageSet.txt")
        self.det_lines_per_cat = {k: [] for k in all_classes}
        os.makedirs(self.det_dir, exist_ok=True)
        os.makedirs(self.annot_dir, exist_ok=True)

    def add_gt(self, image_id: int, difficulty, polygons: np.ndarray, categories, flip_coor: bool = True):
        self.image_set.append(f"{image_id:04}")
        lines = []
        for p, cat, dif in zip(polygons, categories, difficulty):
            p = np.asarray(p)
            if flip_coor:
                p = np.flip(p, axis=-1)
            if self.det_type == "hbb":
                (x0, y0), (x1, y1) = p.min(axis=0), p.max(axis=0)
                p = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
            lines.append(" ".join([" ".join(str(a) for a in p.astype(int).ravel()), cat, str(int(dif))]))
        with open(os.path.join(self.annot_dir, f"{image_id:04}.txt"), "w") as f:
            f.write("\n".join(lines))

    def add_detections(self, image_id: int, scores, class_names, polygons: np.ndarray = None, bbox=None,
                       flip_coor: bool = True):
        n = len(polygons) if polygons is not None else len(bbox)
        for i in range(n):
            if polygons is not None:
                p = np.flip(polygons[i], axis=-1) if flip_coor else np.asarray(polygons[i])
                coords = p.ravel()
            else:
                b = bbox[i]
                coords = [b[1], b[0], b[3], b[2]] if flip_coor else list(b[:4])
            line = " ".join([f"{image_id:04}", str(scores[i]), " ".join(f"{a:.1f}" for a in coords)])
            self.det_lines_per_cat[class_names[i]].append(line)

    def save(self):
        for name, lines in self.det_lines_per_cat.items():
            with open(os.path.join(self.det_dir, f"{name}.txt"), "w") as f:
                f.write("\n".join(lines))
        with open(self.image_set_file, "w") as f:
            f.write("\n".join(self.image_set))
