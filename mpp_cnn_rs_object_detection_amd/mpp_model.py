"""``MPPModel``: config -> energy setup -> tiled RJMCMC inference -> merged, scored detections.

Mirrors the reference's ``models/mpp/mpp_model.py:43-387`` for the inference path
(``main.py -p infer -m mpp``): same config keys, same stored artefacts
(``<model_path>/mpp/<name>/{config.json, calibration.json, energy_combination_model.*}``), same
tiling / merge / scoring steps, same output files.  What changes is where the work runs: all
tiles of an image -- of many images, when a dataset is inferred -- are sampled in ONE kernel launch (one
workgroup per tile) instead of a process pool; under ``torchrun --nproc-per-node N`` a dataset is sharded
by image over the N GPUs (one gather of the results at the end) and a single image larger than one
GPU's worth of chains has its tiles dealt to the ranks and the detections all-gathered.

``train`` covers the ``manual`` mode (what ``mpp_hrcM`` uses) and the weight learning of ``mpp_log``
(``train_ordering_criterion.py`` / ``train_integral_criterion.py``); ``calibrate`` the three
calibrations of ``calibration/energy_calibration.py``.
"""
from __future__ import annotations

import functools
import gc
import json
import logging
import os
import pickle
import re
import time
from typing import Dict, List, Optional

import numpy as np

from . import distributed as mdist
from . import energies as E
from .custom_types import ImageWMaps
from .data_loaders import (PATCH_SIZE, Detections, crop_image_w_maps, crop_region, distance_merge, load_image_w_maps, stack_tiles,
                           merge_patches, merge_score_images, tile_anchors)
from .hip_api import MppError
from .point_set import EPointsSet
from .dota_results import DOTAResultsTranslator
from .paths import fetch_data_paths, get_inference_path, get_model_base_path
from .sampler import TileBatchSampler, resolve_schedule
from .shapes import Rectangle, rect_to_poly, sra_to_wla

TRAIN_MODES = ["manual", "grad_descent", "integral_criterion", "ordering_criterion"]


def save_energy_combinator(model, save_path: str):
    """the JSON twin of the reference's ``energy_combination_model.pkl`` (mpp_model.py:199-200)"""
    if isinstance(model, E.LogisticEnergyCombinator):
        d = {"type": "LogisticEnergyCombinator", "weights": np.asarray(model.weights, dtype=float).tolist(),
             "bias": float(model.bias), "energy_names": list(model.energy_names)}
    elif isinstance(model, E.HierarchicalEnergyCombinator):
        d = {"type": "HierarchicalEnergyCombinator", "weights_data": np.asarray(model.weights_data, float).tolist(),
             "weights_prior": np.asarray(model.weights_prior, float).tolist(),
             "data_prior_weights": np.asarray(model.data_prior_weights, float).tolist(),
             "detection_threshold": float(model.detection_threshold), "bias": float(model.bias)}
    elif isinstance(model, E.ManualHierarchicalEnergyCombinator):
        d = {"type": "ManualHierarchicalEnergyCombinator", "weights_dict": model.weights_dict,
             "indicator_energy": model.indicator_energy, "detection_threshold": model.detection_threshold}
    else:
        raise TypeError(type(model))
    with open(os.path.join(save_path, "energy_combination_model.json"), "w") as f:
        json.dump(d, f, indent=1)


def load_energy_combinator(save_path: str):
    """``energy_combination_model.json`` (this build) or, for models stored by the reference, its pickle
    -- read with a restricted unpickler that only rebuilds the two combinator classes and numpy arrays."""
    js = os.path.join(save_path, "energy_combination_model.json")
    if os.path.exists(js):
        with open(js) as f:
            d = json.load(f)
        if d["type"] == "LogisticEnergyCombinator":
            return E.LogisticEnergyCombinator(weights=np.array(d["weights"], dtype=np.float32), bias=d["bias"],
                                              energy_names=d["energy_names"])
        if d["type"] == "HierarchicalEnergyCombinator":
            return E.HierarchicalEnergyCombinator(np.array(d["weights_data"]), np.array(d["weights_prior"]),
                                                  np.array(d["data_prior_weights"]), d["detection_threshold"],
                                                  d.get("bias", 0.0))
        if d["type"] == "ManualHierarchicalEnergyCombinator":
            return E.ManualHierarchicalEnergyCombinator(d["weights_dict"], d["indicator_energy"],
                                                        d.get("detection_threshold", 0.0))
        raise ValueError(f"unknown combinator type {d['type']}")
    pk = os.path.join(save_path, "energy_combination_model.pkl")
    if os.path.exists(pk):
        allowed = {"HierarchicalEnergyCombinator": E.HierarchicalEnergyCombinator,
                   "LogisticEnergyCombinator": E.LogisticEnergyCombinator,
                   "ManualHierarchicalEnergyCombinator": E.ManualHierarchicalEnergyCombinator}

        class _U(pickle.Unpickler):
            def find_class(self, module, name):
                if name in allowed:
                    return allowed[name]
                if module.startswith("numpy"):
                    return super().find_class(module, name)
                raise pickle.UnpicklingError(f"refusing to load {module}.{name}")

        with open(pk, "rb") as f:
            return _U(f).load()
    raise FileNotFoundError(js)


class _EpochLoader:
    """What the reference's DataLoader(collate_fn=identity) is to the trainers: iterating it yields one epoch of
    batches (lists) of freshly drawn random patches."""

    def __init__(self, data, batch_size: int):
        self.data, self.batch_size = data, batch_size

    def __len__(self):
        return (len(self.data) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        return iter(self.data.batches(self.batch_size))


def _gc_paused(fn):
    """Run ``fn`` with Python's cyclic garbage collector paused.  An image makes thousands of small acyclic objects
    (rectangles, tuples); every few images that trips a full collection, which walks every long-lived object of the process
    (the torch modules of the U-Nets among them) and takes 45-60 ms -- half of what a 4096 x 4096 image costs after its
    chains (profiles/tools/probe_take.py).  Reference counting still frees everything an image makes."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        was = gc.isenabled()
        gc.disable()
        try:
            return fn(*args, **kwargs)
        finally:
            if was:
                gc.enable()
    return wrapped


class MPPModel:
    def __init__(self, config: Dict, phase: str = "val", overwrite: bool = False, load: bool = False,
                 dataset: str = None, device: int = 0, nets=None, spec_waves: Optional[int] = None):
        assert phase in ["val", "train"]
        self.config = config
        self.save_path = os.path.join(get_model_base_path(), "mpp", config["model_name"])
        os.makedirs(self.save_path, exist_ok=True)
        if dataset is not None:
            self.config["dataset"]["dataset"] = dataset
        self.rng = np.random.default_rng(0)                       # reference mpp_model.py:52
        self.dataset = self.config["dataset"]["dataset"]
        self.position_model = self.config["dataset"]["position_model"]
        self.shape_model = self.config["dataset"]["shape_model"]
        self.device, self.nets, self.spec_waves = device, nets, spec_waves
        logging.basicConfig(format="%(levelname)-8s [%(filename)s:%(lineno)d] %(message)s", level=logging.INFO)

        kind = self.config.get("energy_setup") or "legacy"
        params = self.config.get("energy_setup_params") or {}
        if kind == "legacy":
            self.energy_setup = E.LegacyEnergySetup(calibration_params=self.config.get("calibration", {}).get("params", {}))
        elif kind == "no-calibration":
            self.energy_setup = E.NoCalibrationEnergySetup(**params)
        elif kind == "contrast":
            self.energy_setup = E.ContrastMeasureEnergySetup(**params)
        else:
            print("energy_setup must be one of : 'legacy', 'no-calibration', 'contrast'")
            raise ValueError(kind)
        logging.info(f"using {kind} energies {self.energy_setup.energy_names}")
        self.energy_model = None
        if load:
            self.energy_setup.load_calibration(self.save_path)
            try:
                self.energy_model = load_energy_combinator(self.save_path)
            except FileNotFoundError:
                if self._find_train_mode() == "manual":
                    self.train()
                else:
                    raise
        else:
            assert phase == "train"
            self.__init_data__("train")
            self.calibrate()

    def __init_data__(self, subset: str):
        """``mpp_model.py:96-104``: random training patches, ``batch_size`` of them per step"""
        from .data_loaders import MPPDataset
        ds = self.config["dataset"]
        self.data = MPPDataset(dataset=ds["dataset"], subset=subset, position_model=ds["position_model"],
                               shape_model=ds["shape_model"], patch_size=ds.get("patch_size", PATCH_SIZE), nets=self.nets)
        self.batch_size = int(self.config.get("data_loader", {}).get("batch_size", 8))

    def calibrate(self):
        """``mpp_model.py:106-122``"""
        n = min(int(self.config["calibration"]["n_images"]), len(self.data))
        idx = self.rng.choice(range(len(self.data)), size=n, replace=False)
        self.energy_setup.calibrate(image_configs=[self.data[i] for i in idx], rng=self.rng, save_path=self.save_path)

    def _find_train_mode(self):
        modes = [t for t in TRAIN_MODES if t in self.config]
        if len(modes) > 1:
            logging.error(f"found {modes} in model config : can only have one train mode")
            raise ValueError
        return modes[0] if modes else None

    def train(self):
        mode = self._find_train_mode()
        if mode == "ordering_criterion":
            from .train_ordering_criterion import Logger, train_ordering_criterion

            self.energy_model = train_ordering_criterion(
                train_loader=_EpochLoader(self.data, self.batch_size), rng=self.rng, save_dir=self.save_path,
                logger=Logger(self.save_path), energy_setup=self.energy_setup, device=self.device,
                **self.config["ordering_criterion"])
            save_energy_combinator(self.energy_model, self.save_path)
            return
        if mode in ("grad_descent", "integral_criterion"):         # mpp_model.py:142-154
            from .train_integral_criterion import train_integral_criterion
            from .train_ordering_criterion import Logger

            self.energy_model = train_integral_criterion(
                train_loader=_EpochLoader(self.data, self.batch_size), rng=self.rng, save_dir=self.save_path,
                logger=Logger(self.save_path), energy_setup=self.energy_setup, device=self.device, **self.config[mode])
            save_energy_combinator(self.energy_model, self.save_path)
            return
        if mode != "manual":
            raise NotImplementedError(f"train mode {mode!r} (reference mpp_model.py:137-197)")
        m = self.config["manual"]
        if isinstance(self.energy_setup, E.LegacyEnergySetup):
            self.energy_model = E.hierarchical_from_manual(m)
            d = {"type": "HierarchicalEnergyCombinator", "weights_data": self.energy_model.weights_data.tolist(),
                 "weights_prior": self.energy_model.weights_prior.tolist(),
                 "data_prior_weights": self.energy_model.data_prior_weights.tolist(),
                 "detection_threshold": self.energy_model.detection_threshold, "bias": 0.0}
        else:
            self.energy_model = E.ManualHierarchicalEnergyCombinator(m.get("weights"), m.get("indicator_energy"),
                                                                     m.get("threshold"))
            d = {"type": "ManualHierarchicalEnergyCombinator", "weights_dict": m.get("weights"),
                 "indicator_energy": m.get("indicator_energy"), "detection_threshold": m.get("threshold")}
        with open(os.path.join(self.save_path, "energy_combination_model.json"), "w") as f:
            json.dump(d, f, indent=1)

    # ------------------------------------------------------------------------------------------------
    #: a rank keeps its tiles plus this margin.  The Papangelou intensity of a point u changes the energy of every
    #: neighbour v within the interaction radius (32 px, prior_energies.py / energy_graph.py:26-29), and v's energy holds
    #: max / min reductions over ITS neighbours w: everything within two radii of u must be part of the configuration
    #: the rank scores u in, and the unit energies of every v need the score maps there
    SCORE_MARGIN = 64

    def tile_layout(self, shape):
        """(patch size, anchors) of an image: the reference's overlapping 256-px tiles (``mpp_model.py:231-248``)"""
        patch = min(PATCH_SIZE, shape[0], shape[1])
        return patch, tile_anchors(shape, patch)

    def own_region(self, shape, rank: int, world_size: int):
        """The image region ``(x0, x1, y0, y1)`` rank needs score maps for: the bounding box of its block of tiles
        plus ``SCORE_MARGIN``, clipped to the image; ``None`` for a rank without tiles; the whole image for one rank."""
        if world_size == 1:
            return (0, int(shape[0]), 0, int(shape[1]))
        patch, anchors = self.tile_layout(shape)
        mine = mdist.shard_tiles(len(anchors), rank, world_size)
        if not mine:
            return None
        xs, ys, m = [anchors[i][0] for i in mine], [anchors[i][1] for i in mine], self.SCORE_MARGIN
        return (max(0, int(min(xs)) - m), min(int(shape[0]), int(max(xs)) + patch + m),
                max(0, int(min(ys)) - m), min(int(shape[1]), int(max(ys)) + patch + m))

    def region_maps(self, image_data: ImageWMaps, rank: int = 0, world_size: int = 1) -> Optional[ImageWMaps]:
        """Score maps of this rank's region as an ``ImageWMaps`` in region coordinates: a view of the image's maps when
        it has them (pickle hand-off), otherwise the two U-Nets run on the region plus their halo
        (``ScoreMapNets.infer_region``) -- each rank's forward covers its own tiles only."""
        region = self.own_region(image_data.shape[:2], rank, world_size)
        if region is None:
            return None
        if image_data.detection_map is not None:
            return self._resident(crop_region(image_data, region))
        if self.nets is None:
            raise ValueError("the image carries no score maps and no nets were given")
        det, marks = self.nets.infer_region(image_data.image, region)
        # the chains read these maps from a stream of their own: the forward (two side streams joined on torch's current
        # stream) must be complete first -- said here, not left to the default stream's implicit ordering
        import torch
        torch.cuda.current_stream(self.device).synchronize()
        x0, x1, y0, y1 = region
        img = image_data.image[x0:x1, y0:y1] if image_data.image is not None else None     # (the classic image energies read it)
        return ImageWMaps(image=img, name=image_data.name, shape=(x1 - x0, y1 - y0), detection_map=det,
                          param_dist_maps=marks, mappings=image_data.mappings, param_names=image_data.param_names,
                          gt_config=[], crop_data={"tl_anchor": np.array([x0, y0]), "full_shape": tuple(image_data.shape[:2])})

    def _resident(self, region: ImageWMaps) -> ImageWMaps:
        """Host score maps (the pickle hand-off) go to the GPU ONCE per image: the tiles are then device views that the
        sampler stacks on the device, and merge / scoring borrow the same tensors -- instead of one upload of every tile
        plus one of the whole image for the merge (2 x 100 B per pixel over PCIe)."""
        if hasattr(region.detection_map, "data_ptr"):
            return region
        import torch
        dev = torch.device("cuda", self.device)
        region.detection_map = torch.from_numpy(np.ascontiguousarray(region.detection_map, dtype=np.float32)).to(dev)
        region.param_dist_maps = [torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(dev)
                                  for m in region.param_dist_maps]
        return region

    @_gc_paused
    def infer_image(self, image_data: ImageWMaps, rank: int = 0, world_size: int = 1, region_data: ImageWMaps = None,
                    seed: int = None):
        """Tile, sample, merge and score one image.  Returns (detections, scores): an ``EPointsSet`` for one rank,
        the list of merged ``Rectangle``s on every rank of a multi-GPU run (all ranks return the same).

        Multi-GPU (reference: the process pool of ``mpp_model.py:250-262``): rank r samples its block of tiles on the
        score maps of its own region, the sampled configurations are packed on the device and all-gathered once,
        every rank scores the gathered points of ITS tiles on its own maps (their neighbours lie inside the region by
        construction), one all-reduce shares the scores, every rank takes the same ``distance_merge`` decision, and a
        second scoring + all-reduce gives the final Papangelou scores of the survivors."""
        shape = tuple(int(v) for v in image_data.shape[:2])
        patch, anchors = self.tile_layout(shape)
        n_tiles = len(anchors)
        mine = mdist.shard_tiles(n_tiles, rank, world_size)
        p = self.config["inference"]["rjmcmc_params"]
        alpha, T_target, total, snaps = resolve_schedule(1, p["init_temperature"], p["alpha_t"], p["burn_in"],
                                                          p["samples_interval"], p["target_temperature"],
                                                          p.get("iter_multiplier"))
        # drawn by every rank for every image, with or without tiles of its own: the generators of all ranks stay in
        # step, so the result does not depend on the number of ranks
        if seed is None:
            seed = int(self.rng.integers(0, 2 ** 63 - 1))
        if region_data is None:
            region_data = self.region_maps(image_data, rank, world_size)
        origin = region_data.crop_data["tl_anchor"] if (region_data is not None and region_data.crop_data) else np.zeros(2, int)
        tiles, stacked = [], None
        if world_size == 1 and len(mine) > 16 and not isinstance(self.energy_setup, E.ContrastMeasureEnergySetup):
            # all tiles of the image at once (maps on the GPU, anchors on a regular grid): one strided copy per map
            stacked = stack_tiles(region_data, [anchors[i] - origin for i in mine], patch)
        for k, i in enumerate(mine):
            if stacked is not None and k > 0:              # (tile 0 keeps real views: the energy setup looks at one tile)
                t = ImageWMaps(image=None, name=region_data.name, shape=(patch, patch), detection_map=None, param_dist_maps=None,
                               mappings=region_data.mappings, param_names=region_data.param_names, labels=None, gt_config=[])
            else:
                t = crop_image_w_maps(region_data, anchors[i] - origin, patch)
            t.crop_data = {"tl_anchor": np.array(anchors[i])}              # image coordinates, as merge_patches expects
            tiles.append(t)
        self._stacked_maps = stacked
        start = time.perf_counter()
        sampler, buf, capacity = None, None, mdist.gather_capacity(n_tiles, world_size)
        if world_size > 1:
            import torch
            buf = torch.zeros((capacity + 1, mdist.RECORD), dtype=torch.float64, device=torch.device("cuda", self.device))
        results: List[List[Rectangle]] = [[] for _ in mine]
        local_error = None
        try:
            if mine:
                sampler = self._sample_tiles(tiles, mine, anchors, p, total, snaps, alpha, T_target, seed, world_size, capacity, buf)
                results = sampler.tile_results
        except Exception as e:                      # noqa: BLE001 -- several ranks: carried through the gather, raised by all
            if world_size == 1:
                raise
            local_error = e
            sampler = None
        # what a caller may want to look at afterwards (tests compare single tiles with the CPU oracle)
        arrays = sampler is not None and getattr(sampler, "arrays", False)
        raw = results
        if arrays:
            results = [Detections(*r) for r in raw]            # (arrays that read like lists of Rectangle)
        self.last_run = {"seed": seed, "anchors": anchors, "patch": patch, "mine": mine, "tile_results": results,
                         "total_steps": total, "snapshot_step": snaps[-1] if snaps else total - 1,
                         "kernel_ms": sampler.kernel_ms if sampler else 0.0}
        if world_size == 1 and arrays:
            # merge + scores on the device (``mpp_merge_score``); an image with more detections than its walk takes: the host
            xy = [r[0] + np.asarray(t.crop_data["tl_anchor"], dtype=np.int32) for r, t in zip(raw, tiles)]
            agg = (np.concatenate(xy) if xy else np.zeros((0, 2), np.int32),
                   np.concatenate([r[1] for r in raw]) if raw else np.zeros((0, 3)))
            try:
                return merge_score_images([region_data], [agg], self.energy_model, self.energy_setup, 3, device=self.device)[0]
            except MppError as e:
                if e.code != -4:
                    raise
            results = [list(r) for r in results]
        if world_size == 1:
            logging.info(f"merging {n_tiles} patches ...")
            merged = merge_patches(patches=tiles, results=results, original_image=region_data, method="distance",
                                   energy_model=self.energy_model, distance=3, energy_setup=self.energy_setup,
                                   device=self.device)
            scores = merged.papangelou_all(energy_combinator=self.energy_model) if len(merged) else np.zeros(0)
            return merged, scores

        # ---- several ranks: ONE all-gather of the device-packed records (tile id, x, y, size, ratio, angle; image coords)
        # (a rank whose local phase failed -- a chain at a hard capacity limit, a full record buffer -- says so in row 0 of its
        # buffer; every rank then raises the same error after the collective instead of waiting for the failed one in it)
        if local_error is not None:
            buf.zero_()
            buf[0, 1] = 1.0
        try:
            rec = mdist.all_gather_detections(buf, device=f"cuda:{self.device}")
        except mdist.RankFailure as e:
            raise RuntimeError(f"image {image_data.name}: {e}" + (f" (here: {local_error})" if local_error is not None else "")) from local_error
        points = [Rectangle(int(r[1]), int(r[2]), size=float(r[3]), ratio=float(r[4]), angle=float(r[5])) for r in rec]
        owned = mdist.tile_owner(n_tiles, world_size)[rec[:, 0].astype(np.int64)] == rank if len(rec) else np.zeros(0, bool)
        xy = rec[:, 1:3]
        alive = np.ones(len(points), dtype=bool)

        def score_owned() -> np.ndarray:
            """Papangelou intensity of this rank's alive points within the configuration of all alive points.  The reduced
            vector carries one more entry, the number of ranks whose local scoring failed: all ranks raise together."""
            out = np.zeros(len(points) + 1)
            try:
                if region_data is not None and np.any(owned & alive):
                    (h, w), (ox, oy) = region_data.shape[:2], origin
                    inside = alive & (xy[:, 0] >= ox) & (xy[:, 0] < ox + h) & (xy[:, 1] >= oy) & (xy[:, 1] < oy + w)
                    idx = np.nonzero(inside)[0]
                    local = [Rectangle(int(xy[k, 0] - ox), int(xy[k, 1] - oy), size=points[k].size, ratio=points[k].ratio,
                                       angle=points[k].angle) for k in idx]
                    unit, pair = self.energy_setup.make_energies(region_data)
                    pts = EPointsSet(local, (h, w), unit, pair, image_data=region_data, device=self.device,
                                     point_capacity=max(1024, len(local) + 64))
                    sc = pts.papangelou_all(energy_combinator=self.energy_model)
                    sel = owned[idx]
                    out[idx[sel]] = sc[sel]
            except Exception:                       # noqa: BLE001
                logging.exception("scoring failed on this rank")
                out[:] = 0.0
                out[-1] = 1.0
            out = mdist.all_reduce_owned(out, device=f"cuda:{self.device}")
            if out[-1] != 0:
                raise RuntimeError(f"image {image_data.name}: scoring the gathered detections failed on {int(out[-1])} rank(s)")
            return out[:-1]

        logging.info(f"merging {n_tiles} patches of {world_size} ranks ...")
        scores = score_owned()
        removed = distance_merge(xy, scores, 3)
        logging.info(f"merge removing {int(removed.sum())} point(s)")
        alive &= ~removed
        scores = score_owned() if removed.any() else scores
        # the order one rank's EPointsSet ends up in: removals swap the last point into the hole (point_set.remove)
        keep = list(range(len(points)))
        slot = {k: k for k in keep}
        for k in np.nonzero(removed)[0]:
            i, last = slot.pop(int(k)), keep.pop()
            if last != k:
                keep[i] = last
                slot[last] = i
        keep = np.array(keep, dtype=np.int64)
        return [points[k] for k in keep], scores[keep]

    def _sample_tiles(self, tiles, mine, anchors, p, total, snaps, alpha, T_target, seed, world_size, capacity, buf):
        """the local phase of ``infer_image``: this rank's tiles in one launch, their configurations packed for the gather"""
        start = time.perf_counter()
        sampler = TileBatchSampler(tiles, self.energy_setup, self.energy_model, device=self.device,
                                   spec_waves=self.spec_waves, use_split_merge=bool(p.get("use_split_merge", False)),
                                   stacked_maps=getattr(self, "_stacked_maps", None))
        self._stacked_maps = None
        sampler.init("naive")
        pack = None
        if world_size > 1:
            def pack(ctx):
                ctx.pack_detections(mine, np.array([anchors[i] for i in mine]), capacity, buf)
        # one rank: the configurations stay arrays when merge + scores run on the device (no picture-reading energy)
        sampler.arrays = (world_size == 1 and E.classic_image(sampler.model_units) is None
                          and not self.config["inference"].get("host_merge", False))
        out = sampler.run(total, snaps, 1, p["init_temperature"], alpha, T_target, seed, chain0=mine[0], on_device=pack,
                          as_arrays=sampler.arrays)
        sampler.tile_results = [res[-1] if res else [] for res in out]
        self.last_intensity = sampler.intensity
        logging.info(f"ran {len(mine)} rjmcmc chains of {total} steps in one launch in "
                     f"{time.perf_counter() - start:.2f}s (kernel {sampler.kernel_ms:.1f} ms)")
        return sampler

    #: tiles sampled per launch when a dataset is inferred on one GPU: tiles of consecutive images are sampled together
    #: (one workgroup per tile: a launch wants at least the 256 CUs' worth), each with the seed and chain id of its image
    TILES_PER_LAUNCH = 256

    @_gc_paused
    def infer_images(self, images: List[ImageWMaps], regions: List[ImageWMaps] = None, image_seeds: List[int] = None):
        """``infer_image`` for several images at once on one GPU: ALL their tiles in ONE launch (the reference samples
        image after image, `mpp_model.py:220-262`; a DOTA image has 4 - 40 tiles, a launch per image leaves most of the
        256 CUs idle).  Every tile keeps the seed of its image and its tile index as chain id, so each image's result is
        exactly what ``infer_image`` returns for it, in the same order of seed draws (``image_seeds``: the images' seeds when the
        caller drew them already -- several ranks draw the seeds of ALL images of a dataset, each samples its own).
        Returns [(detections, scores)]."""
        regions = regions or self._regions_of_batch(images)
        p = self.config["inference"]["rjmcmc_params"]
        alpha, T_target, total, snaps = resolve_schedule(1, p["init_temperature"], p["alpha_t"], p["burn_in"],
                                                          p["samples_interval"], p["target_temperature"],
                                                          p.get("iter_multiplier"))
        layout, tiles, seeds, chains = [], [], [], []
        for k, (data, region) in enumerate(zip(images, regions)):
            patch, anchors = self.tile_layout(tuple(int(v) for v in data.shape[:2]))
            seed = int(self.rng.integers(0, 2 ** 63 - 1)) if image_seeds is None else int(image_seeds[k])
            mine = []
            for i, a in enumerate(anchors):
                t = crop_image_w_maps(region, a, patch)
                t.crop_data = {"tl_anchor": np.array(a)}
                mine.append(t)
            layout.append((len(tiles), len(mine)))
            tiles += mine
            seeds += [seed] * len(mine)
            chains += list(range(len(mine)))
        if len({tuple(t.shape[:2]) for t in tiles}) > 1:
            raise ValueError("images whose tiles differ in size cannot share a launch")
        start = time.perf_counter()
        sampler = TileBatchSampler(tiles, self.energy_setup, self.energy_model, device=self.device, spec_waves=self.spec_waves,
                                   use_split_merge=bool(p.get("use_split_merge", False)), keys=(seeds, chains))
        sampler.init("naive")
        # merge + scores on the device for the whole batch when the images share a shape and a picture-free energy setup
        # (the classic image energies read a per-image picture the batch context does not hold)
        on_device = (len({tuple(int(v) for v in r.shape[:2]) for r in regions}) == 1 and E.classic_image(sampler.model_units) is None
                     and not self.config["inference"].get("host_merge", False))
        out = sampler.run(total, snaps, 1, p["init_temperature"], alpha, T_target, seed=0, chain0=0, as_arrays=on_device)
        logging.info(f"ran {len(tiles)} rjmcmc chains ({len(images)} images) of {total} steps in one launch in "
                     f"{time.perf_counter() - start:.2f}s (kernel {sampler.kernel_ms:.1f} ms)")
        if on_device:
            aggregated = []
            for first, n in layout:
                xy = [out[t][-1][0] + np.asarray(tiles[t].crop_data["tl_anchor"], dtype=np.int32) for t in range(first, first + n)]
                mk = [out[t][-1][1] for t in range(first, first + n)]
                aggregated.append((np.concatenate(xy) if xy else np.zeros((0, 2), np.int32),
                                   np.concatenate(mk) if mk else np.zeros((0, 3))))
            try:
                return merge_score_images(regions, aggregated, self.energy_model, self.energy_setup, 3, device=self.device)
            except MppError as e:
                if e.code != -4:                      # (-4: an image with more points than the device walk takes)
                    raise
                from .sampler import _to_rectangles
                out = [[_to_rectangles(*o[-1])] for o in out]
        results = []
        for (first, n), data, region in zip(layout, images, regions):
            res = [r[-1] if r else [] for r in out[first:first + n]]
            merged = merge_patches(patches=tiles[first:first + n], results=res, original_image=region, method="distance",
                                   energy_model=self.energy_model, distance=3, energy_setup=self.energy_setup,
                                   device=self.device)
            scores = merged.papangelou_all(energy_combinator=self.energy_model) if len(merged) else np.zeros(0)
            results.append((merged, scores))
        return results

    def _regions_of_batch(self, images: List[ImageWMaps]) -> List[ImageWMaps]:
        """Score maps of a batch of images on the GPU.  Host maps of one shape go into ONE tensor per map kind (the images
        are views of it): the batch's merge / scoring context borrows those tensors as they are."""
        if (self.nets is not None or any(d.detection_map is None or hasattr(d.detection_map, "data_ptr") for d in images)
                or len({tuple(np.shape(d.detection_map)) for d in images}) != 1):
            return [self.region_maps(d) for d in images]
        import torch
        dev = torch.device("cuda", self.device)
        B, (H, W) = len(images), np.shape(images[0].detection_map)
        det = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        marks = [torch.empty((B, H, W) + tuple(np.shape(images[0].param_dist_maps[k])[2:]), dtype=torch.float32, device=dev) for k in range(3)]
        regions = []
        for i, d in enumerate(images):
            det[i].copy_(torch.from_numpy(np.ascontiguousarray(d.detection_map, dtype=np.float32)))
            for k in range(3):
                marks[k][i].copy_(torch.from_numpy(np.ascontiguousarray(d.param_dist_maps[k], dtype=np.float32)))
            r = crop_region(d, (0, H, 0, W))
            r.detection_map, r.param_dist_maps = det[i], [marks[k][i] for k in range(3)]
            regions.append(r)
        return regions

    def _prefetch_images(self, patch_ids, dataset, subset, rank: int = 0, world_size: int = 1):
        """Images with the score maps of this rank's region, one image ahead of the consumer: while the chain kernel of
        image i runs, a worker thread reads image i+1 and (with ``nets``) runs the two U-Nets and their epilogues on a
        side stream.  Yields (image_data, region_data)."""
        from concurrent.futures import ThreadPoolExecutor
        side = None
        if self.nets is not None:
            import torch
            side = torch.cuda.Stream(device=self.device)

        def load(pid):
            data = load_image_w_maps(pid, dataset=dataset, subset=subset, position_model=self.position_model,
                                     shape_model=self.shape_model, nets=self.nets, defer_maps=True)
            if side is None:
                return data, self.region_maps(data, rank, world_size)
            import torch
            with torch.cuda.stream(side):
                region = self.region_maps(data, rank, world_size)
            side.synchronize()
            return data, region

        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(load, patch_ids[0]) if patch_ids else None
            for k in range(len(patch_ids)):
                data = fut.result()
                fut = pool.submit(load, patch_ids[k + 1]) if k + 1 < len(patch_ids) else None
                yield data

    #: an image is sampled by ALL ranks together (its tiles dealt to them, ``infer_image``) only when it has more tiles than
    #: one GPU runs chains at a time; every other image of a dataset belongs to one rank (``infer``)
    TILE_SHARD_MIN = 2048

    def _result_record(self, patch_id: int, out_file: str, image_data: ImageWMaps, merged, scores) -> dict:
        """Write ``NNNN_results.pkl`` of one image (mpp_model.py:333-366) and return what the two DOTA translators need
        of it (plain arrays: the records of all ranks travel to rank 0 in one gather)."""
        pts = list(merged)
        pred_params = [sra_to_wla(p.size, p.ratio, p.angle) for p in pts]
        pred_centers = np.array([[p.x, p.y] for p in pts]).reshape(-1, 2)
        labels = image_data.labels
        gt_poly = np.array([rect_to_poly(c, short=q[0], long=q[1], angle=q[2])
                            for c, q in zip(labels["centers"], labels["parameters"])]).reshape(-1, 4, 2)
        det_poly = np.array([rect_to_poly(c, q[0], q[1], q[2]) for c, q in zip(pred_centers, pred_params)]).reshape(-1, 4, 2)
        difficult = np.asarray(labels.get("difficult", np.zeros(len(gt_poly), int)))
        cats = list(labels.get("categories", ["vehicle"] * len(gt_poly)))
        max_score = self.config["inference"].get("max_score") or 4.0
        score01 = np.asarray(scores) / max_score
        if len(score01) > 0 and np.max(score01) > 1.0:
            logging.warning(f"pred score higher than max, effective score is {np.max(scores)} while param says {max_score}")
        with open(out_file, "wb") as f:
            pickle.dump({"detection": det_poly, "detection_points": [p.as_row() for p in pts],
                         "detection_type": "poly", "detection_center": pred_centers,
                         "detection_score": list(map(float, scores)), "detection_params": pred_params}, f)
        return {"patch_id": patch_id, "gt_poly": gt_poly, "difficult": difficult, "cats": cats, "score01": score01,
                "det_poly": det_poly}

    def infer(self, subset: str, min_confidence: float = 0.1, display_min_confidence: float = 0.5,
              overwrite: bool = True):
        """Reference ``mpp_model.py:202-370`` (figures are not drawn).

        Several ranks (one per GPU): the DATASET is sharded by image -- the reference's own loop is serial over images
        (``mpp_model.py:220``) with a process pool inside each; here rank r takes the r-th block of the images and runs them
        through the batched one-GPU path (tiles of many images in one launch), writes their ``_results.pkl`` files, and ONE
        gather at the end brings the DOTA lines of all ranks to rank 0.  The seeds of ALL images are drawn by every rank, so
        every image is sampled with the seed -- and gives the files -- of a one-rank run.  Only an image with more tiles than
        one GPU runs chains at a time (``TILE_SHARD_MIN``) is sampled by all ranks together."""
        rank, world = mdist.init_process_group()
        dataset = self.config["dataset"]["dataset"]
        results_dir = get_inference_path(os.path.split(self.save_path)[1], dataset, subset)
        os.makedirs(results_dir, exist_ok=True)
        id_re = re.compile(r"([0-9]+).*.png")
        todo, sizes = [], []
        for pf in fetch_data_paths(dataset, subset)["images"]:
            patch_id = int(id_re.match(os.path.split(pf)[1]).group(1))
            out_file = os.path.join(results_dir, f"{patch_id:04}_results.pkl")
            if os.path.exists(out_file) and not overwrite:
                print(f"{patch_id:04}_results.pkl exists, skipping")
                continue
            todo.append((patch_id, out_file))
            sizes.append(pf)
        seeds = [int(self.rng.integers(0, 2 ** 63 - 1)) for _ in todo]        # one per image, dataset order, on every rank
        shared, own = [], list(range(len(todo)))
        if world > 1:
            from PIL import Image
            n_tiles = []
            for pf in sizes:                                                    # (the header only: no pixel is decoded)
                with Image.open(pf) as im:
                    n_tiles.append(len(self.tile_layout((im.size[1], im.size[0]))[1]))
            shared = [k for k in range(len(todo)) if n_tiles[k] > self.TILE_SHARD_MIN]
            rest = [k for k in range(len(todo)) if n_tiles[k] <= self.TILE_SHARD_MIN]
            own = [rest[j] for j in mdist.shard_tiles(len(rest), rank, world)]
        records, failure = [], None

        # ---- images too large for one GPU: all ranks, one image at a time, tiles dealt to the ranks
        for k, (image_data, region_data) in zip(shared, self._prefetch_images([todo[k][0] for k in shared], dataset, subset,
                                                                                rank, world)):
            merged, scores = self.infer_image(image_data, rank, world, region_data=region_data, seed=seeds[k])
            if rank == 0:
                records.append(self._result_record(todo[k][0], todo[k][1], image_data, merged, scores))

        # ---- this rank's images: the tiles of consecutive images share a launch (``infer_images``)
        def inferred():
            stream = zip(own, self._prefetch_images([todo[k][0] for k in own], dataset, subset))
            limit = int(self.config["inference"].get("tiles_per_launch", self.TILES_PER_LAUNCH))
            batch, n_batch, patch0 = [], 0, None

            def flush():
                res = self.infer_images([b[1] for b in batch], [b[2] for b in batch], image_seeds=[seeds[b[0]] for b in batch])
                for (k, image_data, _), (merged, scores) in zip(batch, res):
                    yield k, image_data, merged, scores

            for k, (image_data, region_data) in stream:
                patch, anchors = self.tile_layout(tuple(int(v) for v in image_data.shape[:2]))
                if batch and (patch != patch0 or n_batch + len(anchors) > limit):
                    yield from flush()
                    batch, n_batch = [], 0
                batch.append((k, image_data, region_data))
                n_batch, patch0 = n_batch + len(anchors), patch
            if batch:
                yield from flush()

        try:
            for k, image_data, merged, scores in inferred():
                records.append(self._result_record(todo[k][0], todo[k][1], image_data, merged, scores))
        except Exception as e:                      # noqa: BLE001 -- several ranks: reported through the gather, raised by all
            if world == 1:
                raise
            failure = f"rank {rank}: {type(e).__name__}: {e}"
            logging.exception("inference failed on this rank")

        # ---- ONE gather of the ranks' records (their failures included: nobody waits in a collective for a rank that raised)
        if world > 1:
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, (records, failure))
            failures = [f for _, f in gathered if f]
            if failures:
                raise RuntimeError("; ".join(failures))
            records = [r for recs, _ in gathered for r in recs]
        if rank == 0:
            tr = DOTAResultsTranslator(dataset, subset, results_dir, det_type="obb", all_classes=["vehicle"])
            tr_sv = DOTAResultsTranslator(dataset, subset, results_dir, det_type="obb", all_classes=["vehicle"], postfix="-SV")
            order = {pid: i for i, (pid, _) in enumerate(todo)}
            for rec in sorted(records, key=lambda r: order[r["patch_id"]]):          # dataset order, as one rank writes them
                n_gt = len(rec["gt_poly"])
                tr.add_gt(image_id=rec["patch_id"], polygons=rec["gt_poly"], difficulty=rec["difficult"],
                          categories=["vehicle"] * n_gt)
                tr_sv.add_gt(image_id=rec["patch_id"], polygons=rec["gt_poly"],
                             difficulty=[bool(d) or c == "large-vehicle" for d, c in zip(rec["difficult"], rec["cats"])],
                             categories=["vehicle"] * n_gt)
                for t in (tr, tr_sv):
                    t.add_detections(image_id=rec["patch_id"], scores=rec["score01"], polygons=rec["det_poly"], flip_coor=True,
                                     class_names=["vehicle"] * len(rec["score01"]))
            tr.save()
            tr_sv.save()
            print("saved dota translation")

    def eval(self):
        """mpp_model.py:372-387: the "vehicle" and the small-vehicle ("-SV") DOTA files written by infer()"""
        from .dota_eval import dota_eval
        out = {}
        if mdist.init_process_group()[0] != 0:       # several ranks: rank 0 wrote the DOTA files and evaluates them
            return out
        for postfix in ("", "-SV"):
            out[postfix] = dota_eval(model_dir=self.save_path, dataset=self.dataset, subset="val", det_type="obb",
                                     postfix=postfix, device=self.device)
        return out

    def data_preview(self):
        raise NotImplementedError("figures are outside this build")
