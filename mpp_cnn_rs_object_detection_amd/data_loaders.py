"""Image + score-map loading, tiling and merging for MPP inference.

Mirrors the reference's ``models/mpp/data_loaders.py:30-161, :305-314`` (``load_image_w_maps``,
``crop_image_w_maps``, ``merge_patches``, ``labels_to_rectangles``) and the tiling rule of
``models/mpp/mpp_model.py:231-248``.  Score maps come either from the reference's on-disk hand-off
(``NNNN_results.pkl`` of posnet / shapenet) or straight from the two U-Nets on the GPU
(``unet.ScoreMapNets``), in which case they never leave the device.
"""
from __future__ import annotations

import io
import logging
import os
import pickle
from copy import copy
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import mappings as _mappings
from .custom_types import ImageWMaps
from .paths import fetch_data_paths, get_dataset_base_path, get_inference_path
from .point_set import EPointsSet
from .energies import build_model_desc as E_build
from .shapes import Rectangle, wla_to_sra

PARAM_NAMES = ["size", "ratio", "angle"]
PATCH_SIZE = 256          # hard-coded in the reference, mpp_model.py:231


class _CompatUnpickler(pickle.Unpickler):
    """The reference pickles its own ``ValueMapping`` class inside shapenet results; map it onto ours."""

    def find_class(self, module, name):
        if name == "ValueMapping":
            return _mappings.ValueMapping
        return super().find_class(module, name)


def _load_pickle(path: str):
    with open(path, "rb") as f:
        return _CompatUnpickler(io.BytesIO(f.read())).load()


def labels_to_rectangles(labels: Dict, param_names: Sequence[str] = PARAM_NAMES) -> List[Rectangle]:
    """(a, b, angle) annotations -> Rectangle(size, ratio, angle mod pi) (reference ``data_loaders.py:305-314``)."""
    out = []
    for c, p in zip(labels["centers"], labels["parameters"]):
        s, r, a = wla_to_sra(p[0], p[1], p[2])
        out.append(Rectangle(int(c[0]), int(c[1]), size=float(s), ratio=float(r), angle=float(a % np.pi)))
    return out


def load_image_w_maps(patch_id, dataset: str, subset: str, position_model: str, shape_model: str,
                      nets=None, defer_maps: bool = False) -> ImageWMaps:
    """Reference ``data_loaders.py:30-71``.  With ``nets`` (a ``ScoreMapNets``) the maps are computed on the GPU
    instead of being read from the inference pickles; ``defer_maps``: leave them ``None`` (the caller runs the nets on
    the part of the image its rank owns, ``MPPModel.region_maps``)."""
    from matplotlib import pyplot as plt
    patch_id = int(patch_id)
    base = os.path.join(get_dataset_base_path(), dataset, subset)
    image = plt.imread(os.path.join(base, "images", f"{patch_id:04}.png"))[:, :, :3]
    labels = _load_pickle(os.path.join(base, "annotations", f"{patch_id:04}.pkl"))
    if nets is not None:
        det, marks = (None, None) if defer_maps else nets.infer(image)
        maps = _mappings.default_mappings()
    else:
        pos = _load_pickle(os.path.join(get_inference_path(position_model, dataset, subset), f"{patch_id:04}_results.pkl"))
        shp = _load_pickle(os.path.join(get_inference_path(shape_model, dataset, subset), f"{patch_id:04}_results.pkl"))
        det = np.asarray(pos["detection_map"], dtype=np.float32)
        marks = [np.ascontiguousarray(np.moveaxis(np.asarray(p)[0], 0, -1), dtype=np.float32) for p in shp["output"]]
        maps = shp.get("mappings") or _mappings.default_mappings()
    return ImageWMaps(image=image, name=f"{patch_id:04}", shape=tuple(image.shape[:2]), detection_map=det,
                      param_dist_maps=marks, mappings=maps, param_names=PARAM_NAMES, labels=labels,
                      gt_config=labels_to_rectangles(labels))


def tile_anchors(shape, patch_size: int = PATCH_SIZE) -> List[np.ndarray]:
    """Overlapping tiles covering the image: ``linspace(0, H - patch, ceil(H / patch))`` per axis
    (reference ``mpp_model.py:233-240``)."""
    H, W = shape[:2]
    ax = np.linspace(0, H - patch_size, max(1, int(np.ceil(H / patch_size))), dtype=int)
    ay = np.linspace(0, W - patch_size, max(1, int(np.ceil(W / patch_size))), dtype=int)
    return [np.array([x, y]) for x in ax for y in ay]


def crop_image_w_maps(image_data: ImageWMaps, tl_anchor: np.ndarray, patch_size: int) -> ImageWMaps:
    """Reference ``data_loaders.py:74-119``; works on numpy arrays and on GPU tensors alike."""
    x, y = int(tl_anchor[0]), int(tl_anchor[1])
    sl = (slice(x, x + patch_size), slice(y, y + patch_size))
    det = image_data.detection_map[sl]
    marks = [m[sl] for m in image_data.param_dist_maps]
    image = image_data.image[sl] if image_data.image is not None else None
    shape = tuple(int(v) for v in det.shape[:2])
    labels = image_data.labels
    new_labels = None
    if labels is not None and len(labels.get("centers", [])) > 0:
        c = np.asarray(labels["centers"]) - np.array([x, y])
        keep = np.all(c >= 0, axis=1) & np.all(c < np.array(shape), axis=1)
        new_labels = {k: np.asarray(labels[k])[keep] for k in ("parameters", "categories", "difficult") if k in labels}
        new_labels["centers"] = c[keep]
    elif labels is not None:
        new_labels = {k: np.asarray(v) for k, v in labels.items()}
    gt = labels_to_rectangles(new_labels) if new_labels is not None and "parameters" in new_labels else []
    return ImageWMaps(image=image, name=image_data.name, shape=shape, detection_map=det, param_dist_maps=marks,
                      mappings=image_data.mappings, param_names=PARAM_NAMES, labels=new_labels, gt_config=gt,
                      crop_data={"tl_anchor": np.array([x, y])})


def stack_tiles(image_data: ImageWMaps, anchors: Sequence[np.ndarray], patch_size: int, require_cuda: bool = True):
    """The score maps of ALL tiles of an image as one tensor per map -- (det [T,p,p], marks 3 x [T,p,p,32]) -- in one strided
    copy each, for maps that are contiguous GPU tensors and anchors on a regular grid (``tile_anchors`` gives one whenever
    ``linspace`` lands on integers); None otherwise (the caller crops tile by tile).  256 tiles of a 4096 x 4096 image:
    four launches instead of a thousand slicing calls and a 256-way concatenation."""
    det = image_data.detection_map
    marks = image_data.param_dist_maps
    if not hasattr(det, "as_strided") or image_data.labels is not None:
        return None
    if not (det.is_contiguous() and all(hasattr(m, "as_strided") and m.is_contiguous() for m in marks)):
        return None
    if require_cuda and not (det.is_cuda and all(m.is_cuda for m in marks)):
        return None
    a = np.asarray(anchors, dtype=np.int64).reshape(-1, 2)
    xs, ys = np.unique(a[:, 0]), np.unique(a[:, 1])
    if len(xs) * len(ys) != len(a) or not np.array_equal(a, np.array([[x, y] for x in xs for y in ys])):
        return None
    sx = int(xs[1] - xs[0]) if len(xs) > 1 else patch_size
    sy = int(ys[1] - ys[0]) if len(ys) > 1 else patch_size
    if (len(xs) > 2 and np.any(np.diff(xs) != sx)) or (len(ys) > 2 and np.any(np.diff(ys) != sy)):
        return None
    H, W = (int(v) for v in det.shape[:2])
    x0, y0, p = int(xs[0]), int(ys[0]), int(patch_size)
    if x0 < 0 or y0 < 0 or int(xs[-1]) + p > H or int(ys[-1]) + p > W:
        return None
    nx, ny = len(xs), len(ys)
    d = det.as_strided((nx, ny, p, p), (sx * W, sy, W, 1), x0 * W + y0).reshape(nx * ny, p, p).contiguous()
    ms = []
    for m in marks:
        C = int(m.shape[2])
        ms.append(m.as_strided((nx, ny, p, p, C), (sx * W * C, sy * C, W * C, C, 1), (x0 * W + y0) * C)
                  .reshape(nx * ny, p, p, C).contiguous())
    return d, ms


def distance_merge(xy: np.ndarray, scores: np.ndarray, distance: float) -> np.ndarray:
    """The dedupe rule of the reference's ``merge_patches(method='distance')`` (``data_loaders.py:140-159``) as a
    function of the aggregated points alone: walking the points in order, every not-yet-removed point keeps, among
    the not-yet-removed points within ``distance`` of it (itself included), only the one with the best Papangelou
    intensity (ties to 1e-9: the first).  Returns the mask of removed points.  Deterministic in its inputs, so ranks
    that hold the same gathered points and scores take the same decision."""
    n = len(xy)
    removed = np.zeros(n, dtype=bool)
    if n == 0:
        return removed
    from scipy.spatial import cKDTree                  # neighbour lists once (integer coordinates: exact comparisons)
    xy = np.asarray(xy, dtype=float).reshape(-1, 2)
    balls = cKDTree(xy).query_ball_point(xy, r=float(distance))
    for i in range(n):
        if removed[i] or len(balls[i]) == 1:           # (alone within `distance`: it is its own best, nothing to remove)
            continue
        near = np.array(sorted(j for j in balls[i] if not removed[j]), dtype=np.int64)
        if len(near) == 0:
            continue
        # the best one; scores equal to 1e-9 count as a tie and the first point (tile order) wins.  (Two tiles that
        # overlap report the same object twice, often with identical energies; the reference breaks such ties by the
        # iteration order of a Python set.  The tolerance keeps the decision independent of the last-place differences
        # between scores computed on different ranks' regions.)
        sc = scores[near]
        top = sc.max()
        if np.isfinite(top):
            best = near[np.nonzero(sc >= top - 1e-9 * abs(top))[0][0]]
        else:
            # a non-finite Papangelou intensity (the `craciun` contrast measure on a one-pixel mask gives inf / NaN
            # energies): the reference's np.argmax (data_loaders.py:151) -- a NaN first, else the first infinity
            best = near[int(np.argmax(sc))]
        removed[near] = True
        removed[best] = False
    return removed


def merge_patches(patches: List[ImageWMaps], results: List[List[Rectangle]], original_image: ImageWMaps, energy_model,
                  method: str, energy_setup, device: int = 0, **kwargs):
    """Reference ``data_loaders.py:122-161``: shift every tile's detections by its anchor, then (method
    'distance') keep, among points closer than ``distance``, the one with the best Papangelou intensity
    in the FULL aggregated configuration.  The intensities of all points come from one GPU launch on ``device``."""
    assert method in ["distance", "rjmcmc"]
    unit, pair = energy_setup.make_energies(original_image)
    merged: List[Rectangle] = []
    for patch, result in zip(patches, results):
        ax, ay = patch.crop_data["tl_anchor"]
        ax, ay = int(ax), int(ay)
        for r in result:
            if type(r) is Rectangle:                  # (copy.copy costs four times the constructor)
                q = Rectangle(int(r.x) + ax, int(r.y) + ay, size=r.size, ratio=r.ratio, angle=r.angle)
            else:
                q = copy(r)
                q.x, q.y = int(q.x + ax), int(q.y + ay)
            merged.append(q)
    agg = EPointsSet(merged, original_image.shape, unit, pair, image_data=original_image, device=device,
                     point_capacity=max(1024, len(merged) + 64))
    if method == "distance" and len(merged) > 0:
        scores = agg.papangelou_all(energy_combinator=energy_model)
        removed = distance_merge(np.array([[p.x, p.y] for p in merged], dtype=float), scores, kwargs["distance"])
        logging.info(f"merge removing {int(removed.sum())} point(s)")
        for i in np.nonzero(removed)[0]:
            agg.remove(merged[i])
    return agg


class Detections:
    """The merged detections of an image as arrays -- ``xy`` [n, 2] int, ``marks`` [n, 3] (size, ratio, angle) -- that
    reads like the list of ``Rectangle`` the reference's ``merge_patches`` hands on (iteration, ``len``, indexing build the
    objects on demand): objects only at the API edge."""

    def __init__(self, xy: np.ndarray, marks: np.ndarray):
        self.xy = np.asarray(xy, dtype=np.int64).reshape(-1, 2)
        self.marks = np.asarray(marks, dtype=np.float64).reshape(-1, 3)
        self._objs = None

    def _list(self) -> List[Rectangle]:
        if self._objs is None:
            self._objs = [Rectangle(x, y, size=s, ratio=r, angle=a)
                          for (x, y), (s, r, a) in zip(self.xy.tolist(), self.marks.tolist())]
        return self._objs

    def __len__(self):
        return len(self.xy)

    def __iter__(self):
        return iter(self._list())

    def __getitem__(self, i):
        return self._list()[i]


def merge_score_images(regions: List[ImageWMaps], aggregated, energy_model, energy_setup, distance: float, device: int = 0):
    """``merge_patches(method='distance')`` + the final Papangelou scores (mpp_model.py:296-304) for SEVERAL images of one
    shape in four kernel launches (``mpp_merge_score``): ``aggregated[k]`` = (xy, marks) of image k's detections in image
    coordinates, tile order; ``regions[k]`` its score maps (device tensors, or arrays).  -> [(Detections, scores)].
    The survivors, their order and their scores are those of ``merge_patches`` followed by ``papangelou_all``."""
    import torch
    from .hip_api import MppContext
    dev = torch.device("cuda", device)

    def stack(arrs):
        if all(hasattr(a, "data_ptr") for a in arrs):
            if len(arrs) == 1 and arrs[0].is_contiguous():
                return arrs[0].unsqueeze(0)              # (one image: its maps as they are -- 6.5 GB at 4096 x 4096)
            base = arrs[0]._base if hasattr(arrs[0], "_base") else None
            # (views of one batch tensor, in order: borrow it as it is instead of copying 100 B per pixel)
            if base is not None and len(base) == len(arrs) and all(a._base is base and a.data_ptr() == base[k].data_ptr()
                                                                   for k, a in enumerate(arrs)) and base.is_contiguous():
                return base
            return torch.stack(list(arrs)).contiguous()
        return torch.from_numpy(np.stack([np.asarray(a, dtype=np.float32) for a in arrs])).to(dev)

    det = stack([r.detection_map for r in regions])
    marks = [stack([r.param_dist_maps[k] for r in regions]) for k in range(3)]
    unit, pair = energy_setup.make_energies(regions[0])
    n_max = max((len(a[0]) for a in aggregated), default=0)
    ctx = MppContext(device, point_capacity=max(256, n_max + 64))
    try:
        ctx.set_maps(det, marks)
        ctx.set_model(E_build(unit, pair, energy_model), regions[0].mappings)
        for k, (xy, mk) in enumerate(aggregated):
            ctx.set_points(k, np.asarray(xy, dtype=np.int32).reshape(-1, 2), np.asarray(mk, dtype=np.float64).reshape(-1, 3))
        res, removed = ctx.merge_score(distance)
    finally:
        ctx.close()
    logging.info(f"merge removing {removed.tolist()} point(s) of {len(regions)} image(s)")
    return [(Detections(xy, mk), np.exp(-dE)) for xy, mk, dE in res]


def crop_region(image_data: ImageWMaps, region) -> ImageWMaps:
    """The score maps of the image region ``(x0, x1, y0, y1)`` as an ``ImageWMaps`` of its own (views, no copy);
    ``crop_data['tl_anchor']`` holds the region's origin in image coordinates."""
    x0, x1, y0, y1 = (int(v) for v in region)
    sl = (slice(x0, x1), slice(y0, y1))
    det = image_data.detection_map[sl]
    return ImageWMaps(image=image_data.image[sl] if image_data.image is not None else None, name=image_data.name,
                      shape=(x1 - x0, y1 - y0), detection_map=det, param_dist_maps=[m[sl] for m in image_data.param_dist_maps],
                      mappings=image_data.mappings, param_names=PARAM_NAMES, labels=None, gt_config=[],
                      crop_data={"tl_anchor": np.array([x0, y0]), "full_shape": tuple(image_data.shape[:2])})


class MPPDataset:
    """Random 256-px training patches with their ground truth (reference ``data_loaders.py:163-251`` with the
    samplers of ``data/patch_samplers.py:38-108``): a uniformly drawn image, the patch centred on a uniformly drawn
    pixel (weight 1/10) or on a jittered object centre (9/10, sigma 10)."""

    def __init__(self, dataset: str, subset: str, position_model: str, shape_model: str, patch_size: int,
                 patch_ids: List[int] = None, nets=None):
        import re
        self.dataset, self.subset, self.patch_size = dataset, subset, patch_size
        self.position_model, self.shape_model, self.nets = position_model, shape_model, nets
        files = fetch_data_paths(dataset, subset)
        assert len(files["images"]) > 0
        pat = re.compile(r"([0-9]+)\.[a-zA-z]+")
        ids = [pat.match(os.path.split(p)[1]).group(1) for p in files["images"]]
        if patch_ids is not None:
            keep = {f"{i:04}" for i in patch_ids}
            sel = [k for k, i in enumerate(ids) if i in keep]
        else:
            sel = list(range(len(ids)))
        self.patches_index = [ids[k] for k in sel]
        self.rng = np.random.default_rng(0)
        # Upstream builds both samplers with n_patches == n_images, for which the per-image densities
        # (count / sum) * (n_patches - n_images) + 1 are uniform (patch_samplers.py:56-58, 92-93).

    def __len__(self):
        return len(self.patches_index)

    def __getitem__(self, index) -> ImageWMaps:
        sampler = 0 if self.rng.random() < 0.1 else 1                          # MixedSampler weights [1/10, 9/10]
        k = int(self.rng.integers(0, len(self)))
        data = load_image_w_maps(self.patches_index[k], self.dataset, self.subset, self.position_model,
                                 self.shape_model, nets=self.nets)
        centers = np.asarray(data.labels["centers"])
        if sampler == 1 and len(centers) > 0:
            c = self.rng.normal(self.rng.choice(centers, axis=0).astype(int), 10).astype(int)
            c = np.clip(c, (0, 0), data.shape[:2])
        else:
            c = self.rng.integers((0, 0), data.shape[:2])
        ps = min(self.patch_size, data.shape[0], data.shape[1])
        tl = np.clip((c - ps // 2).astype(int), (0, 0), (data.shape[0] - ps, data.shape[1] - ps))
        return crop_image_w_maps(image_data=data, tl_anchor=tl, patch_size=ps)

    def batches(self, batch_size: int):
        """one epoch of lists of patches (the reference's DataLoader with ``collate_fn`` = identity)"""
        n = len(self)
        return [[self[i] for i in range(b, min(b + batch_size, n))] for b in range(0, n, batch_size)]
