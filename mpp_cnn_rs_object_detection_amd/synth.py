"""Synthetic tiles for parity tests and the bench (SURVEY.md section 8(d) recipe).

There is no network and no trained ``model.pt`` in the build container, so the
score maps a tile is sampled on are synthesised: a detection map with a narrow
Gaussian bump on every ground-truth centre, and three 32-bin mark
distributions that put 0.9 on the ground-truth class in a 7x7 window around
each centre.  Array layouts match what the reference hands to its sampler
(``models/mpp/data_loaders.py:48-55``): ``det`` is ``[H, W] float32`` and each
mark map is ``[H, W, 32] float32``.

Only ``numpy.random.Generator.random/integers/permutation/normal`` are used so
the scenes are reproducible wherever this file runs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

N_CLASSES = 32
#: (v_min, v_max, cyclic) of the three marks; reference
#: ``models/shape_net/shape_net_model.py:80-85``.
MARK_RANGES = ((0.0, 32.0, False), (0.0, 1.0, False), (0.0, float(np.pi), True))


def mark_edges() -> np.ndarray:
    """``[3, 32] float64`` lower bin edges = ``linspace(vmin, vmax, 33)[:-1]``
    (reference ``models/shape_net/mappings.py:16-17``)."""
    return np.stack([np.linspace(lo, hi, N_CLASSES + 1)[:-1] for lo, hi, _ in MARK_RANGES])


def wla_to_sra(a, b, angle):
    """(width a, length b, angle) -> (size, ratio, angle); reference
    ``base/shapes/rectangle.py:103-104``."""
    return (a + b) / 2, a / b, angle


def value_to_class(values: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """max{i : v >= edge_i}  (reference ``mappings.py:54-55``)."""
    return np.maximum(np.searchsorted(edges, values, side="right") - 1, 0)


@dataclass
class SynthTile:
    shape: Tuple[int, int]
    det: np.ndarray            # [H, W] float32
    marks: List[np.ndarray]    # 3 x [H, W, 32] float32
    gt_xy: np.ndarray          # [N, 2] int32 (row, col)
    gt_marks: np.ndarray       # [N, 3] float64 (size, ratio, angle)


def make_gt(tile: int, n_objects: int, tile_id: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(1234 + tile_id)
    lattice = np.arange(10, tile - 10, 14)
    n_cells = len(lattice) ** 2
    n_objects = min(n_objects, n_cells)
    pick = rng.permutation(n_cells)[:n_objects]
    cx = lattice[pick // len(lattice)] + rng.integers(-2, 3, size=n_objects)
    cy = lattice[pick % len(lattice)] + rng.integers(-2, 3, size=n_objects)
    a = rng.normal(4.5, 0.4, size=n_objects)
    b = rng.normal(9.0, 0.8, size=n_objects)
    theta = rng.random(n_objects) * np.pi
    size, ratio, angle = wla_to_sra(a, b, theta)
    xy = np.stack([cx, cy], axis=1).astype(np.int32)
    marks = np.stack([size, np.clip(ratio, 0.05, 1.0), angle], axis=1)
    return xy, marks


def render_maps(shape: Tuple[int, int], gt_xy: np.ndarray, gt_marks: np.ndarray,
                noise: float = 0.0, noise_seed: int = 0) -> Tuple[np.ndarray, List[np.ndarray]]:
    """Score maps for a set of ground-truth rectangles.

    ``noise`` > 0 mixes seeded uniform noise into the mark distributions (then
    renormalises each 32-bin row) so that data-driven proposals see non-trivial
    probabilities; 0 gives the plain SURVEY recipe."""
    H, W = shape
    det = np.full((H, W), 0.02, dtype=np.float64)
    R = 5
    for (x, y) in gt_xy:
        x0, x1 = max(0, x - R), min(H, x + R + 1)
        y0, y1 = max(0, y - R), min(W, y + R + 1)
        xx, yy = np.mgrid[x0:x1, y0:y1]
        bump = np.exp(-((xx - x) ** 2 + (yy - y) ** 2) / (2 * 1.2 ** 2))
        det[x0:x1, y0:y1] = np.maximum(det[x0:x1, y0:y1], bump)
    det = det.astype(np.float32)

    edges = mark_edges()
    marks = []
    nrng = np.random.default_rng(noise_seed)
    for k in range(3):
        m = np.full((H, W, N_CLASSES), 1.0 / N_CLASSES, dtype=np.float32)
        cls = value_to_class(gt_marks[:, k], edges[k])
        for (x, y), c in zip(gt_xy, cls):
            x0, x1 = max(0, x - 3), min(H, x + 4)
            y0, y1 = max(0, y - 3), min(W, y + 4)
            m[x0:x1, y0:y1, :] = np.float32(0.1 / (N_CLASSES - 1))
            m[x0:x1, y0:y1, c] = np.float32(0.9)
        if noise > 0:
            m = m + np.float32(noise) * nrng.random(m.shape, dtype=np.float32)
            m = m / m.sum(axis=-1, keepdims=True, dtype=np.float32)
        marks.append(np.ascontiguousarray(m, dtype=np.float32))
    return det, marks


def make_tile(tile: int = 512, n_objects: int = 200, tile_id: int = 0, noise: float = 0.0) -> SynthTile:
    """The SURVEY 8(d) synthetic tile: config 1 is (256, 50), config 2 is (512, 200)."""
    gt_xy, gt_marks = make_gt(tile, n_objects, tile_id)
    det, marks = render_maps((tile, tile), gt_xy, gt_marks, noise=noise, noise_seed=77 + tile_id)
    return SynthTile(shape=(tile, tile), det=det, marks=marks, gt_xy=gt_xy, gt_marks=gt_marks)


# ---- the IMAGE recipe of the reference's synthetic dataset (BASELINE config 5) ------------------------------------
def _rect_corners(x, y, size, ratio, angle) -> np.ndarray:
    """(4, 2) corners of Rectangle(x, y, size, ratio, angle).poly_coord (``base/shapes/rectangle.py:20-31,69-100``)"""
    length = 2.0 * size / (1.0 + ratio)
    width = ratio * length
    hs, hl = length / 2.0, width / 2.0
    local = np.array([[hs, hl], [hs, -hl], [-hs, -hl], [-hs, hl]])
    a = angle + np.pi / 2
    c, s = np.cos(a), np.sin(a)
    return local @ np.array([[c, -s], [s, c]]).T + np.array([x, y], dtype=float)


def _quads_overlap(a: np.ndarray, b: np.ndarray) -> bool:
    """Two convex quads share area  <=>  none of the 8 edge normals separates them (separating-axis theorem)."""
    for q in (a, b):
        e = np.roll(q, -1, axis=0) - q
        normals = np.stack([-e[:, 1], e[:, 0]], axis=1)
        pa, pb = a @ normals.T, b @ normals.T
        if np.any((pa.max(axis=0) <= pb.min(axis=0)) | (pb.max(axis=0) <= pa.min(axis=0))):
            return False
    return True


def make_scene_image(shape: Tuple[int, int] = (256, 256), n_rect: int = 230, noise: float = 0.02, seed: int = 0):
    """Restates ``make_synth`` of the reference's ``data/make_synth_data.py:16-47``: ``n_rect`` rectangles with
    centre ~ U(image), size ~ N(8, 1), ratio ~ clip(N(0.5, 0.1), 0.1, 1), angle ~ U(0, pi), drawn in that order from
    one generator; a rectangle is kept if it overlaps none of those kept before it; the image is 0.5 everywhere,
    each kept rectangle filled with choice{0, 1} + N(0, 0.1), clipped, plus pixel noise N(0, ``noise``), clipped.
    (The reference tests overlap with shapely and rasterises with ``skimage.draw.polygon``, both absent here: the
    separating-axis test and a pixel-centre-inside test stand in; overlap candidates come from a coarse grid so that
    a 4096 x 4096 scene with ~5 000 rectangles -- BASELINE config 5 -- takes seconds.)
    Returns (image [H, W, 3] float32, gt_xy [N, 2] int32, gt_marks [N, 3] float64)."""
    rng = np.random.default_rng(seed)
    H, W = shape
    rects = []
    for _ in range(n_rect):
        x = int(rng.integers(0, H)); y = int(rng.integers(0, W))
        size = float(rng.normal(8, 1.0))
        ratio = float(np.clip(rng.normal(0.5, 0.1), 0.1, 1))
        angle = float(rng.uniform(0, np.pi))
        rects.append((x, y, size, ratio, angle))
    cell = 32
    grid: dict = {}
    kept, kept_poly = [], []
    for r in rects:
        poly = _rect_corners(*r)
        ci, cj = r[0] // cell, r[1] // cell
        ok = True
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for k in grid.get((ci + di, cj + dj), ()):
                    if _quads_overlap(poly, kept_poly[k]):
                        ok = False
                        break
                if not ok:
                    break
            if not ok:
                break
        if ok:
            grid.setdefault((ci, cj), []).append(len(kept))
            kept.append(r)
            kept_poly.append(poly)
    img = np.ones((H, W, 3)) * 0.5
    for r, poly in zip(kept, kept_poly):
        value = float(rng.choice([0, 1.0])) + float(rng.normal(0, 0.1))
        x0, x1 = max(0, int(np.floor(poly[:, 0].min()))), min(H, int(np.ceil(poly[:, 0].max())) + 1)
        y0, y1 = max(0, int(np.floor(poly[:, 1].min()))), min(W, int(np.ceil(poly[:, 1].max())) + 1)
        if x1 <= x0 or y1 <= y0:
            continue
        xx, yy = np.mgrid[x0:x1, y0:y1]
        e = np.roll(poly, -1, axis=0) - poly
        side = [(e[k, 0] * (yy - poly[k, 1]) - e[k, 1] * (xx - poly[k, 0])) for k in range(4)]
        inside = np.all(np.stack(side) >= 0, axis=0) | np.all(np.stack(side) <= 0, axis=0)
        img[x0:x1, y0:y1][inside] = value
    img = np.clip(img, 0, 1)
    img = img + rng.normal(0, noise, size=img.shape)
    img = np.clip(img, 0, 1).astype(np.float32)
    gt_xy = np.array([[r[0], r[1]] for r in kept], dtype=np.int32).reshape(-1, 2)
    gt_marks = np.array([[r[2], r[3], r[4]] for r in kept], dtype=np.float64).reshape(-1, 3)
    return img, gt_xy, gt_marks


def make_mosaic(n_side: int = 4, tile: int = 512, n_objects: int = 200, first_tile_id: int = 0, noise: float = 0.0):
    """BASELINE config 4: an ``n_side`` x ``n_side`` mosaic of the synthetic tile generator:
    -> det [S, S] f32, marks 3 x [S, S, 32] f32, gt_xy [N, 2], gt_marks [N, 3] with S = n_side * tile."""
    S = n_side * tile
    det = np.zeros((S, S), np.float32)
    marks = [np.zeros((S, S, N_CLASSES), np.float32) for _ in range(3)]
    xy, mk = [], []
    for i in range(n_side):
        for j in range(n_side):
            t = make_tile(tile, n_objects, tile_id=first_tile_id + i * n_side + j, noise=noise)
            sl = (slice(i * tile, (i + 1) * tile), slice(j * tile, (j + 1) * tile))
            det[sl] = t.det
            for k in range(3):
                marks[k][sl] = t.marks[k]
            xy.append(t.gt_xy + np.array([i * tile, j * tile], dtype=np.int32))
            mk.append(t.gt_marks)
    return det, marks, np.concatenate(xy), np.concatenate(mk)


# ---- score-map nets without trained weights (there is no model.pt in the build container) ---------------------------
def random_score_nets(seed: int = 0, device: int = 0, dtype=None):
    """Seeded random-init PosNet + ShapeNet as a ``ScoreMapNets``.  What such nets 'detect' is meaningless; the path
    from an image to scored detections -- and its cost -- is what runs."""
    import torch
    from . import unet
    torch.manual_seed(seed)
    pos, shp = unet.PosNet(), unet.ShapeNet()
    return unet.ScoreMapNets(pos, shp, device=device, dtype=dtype or torch.float32)


def calibrate_div_clf(nets, img_crop, frac: float = 0.0015):
    """Re-scale the 1x1 "div_clf" of a random posnet so that the detection map fires: the most convergent ``frac`` of the
    pixels of ``img_crop`` reach det = 0.9 (bias as shipped, pos_net_model.py:338-346)."""
    import torch
    nets.div_w, nets.div_b = -1.0, 0.0
    det0, _ = nets.infer(img_crop)
    z = torch.logit(det0.flatten().double().clamp(1e-9, 1 - 1e-9))          # = -(divergence * mask)
    q = float(torch.quantile(z[::7].float(), 1.0 - frac))
    nets.div_b = -2.128434
    nets.div_w = -(2.2 - nets.div_b) / max(q, 1e-9)
