"""Perturbed configurations for energy-weight learning.

Mirrors the reference's ``models/mpp/perturbation_sampler.py:15-211``:

* ``sample_perturbations``: jitter / drop / add points of a ground-truth configuration with plain
  NumPy draws -- the draws are made in the reference's order, so with the same ``Generator`` the
  result is the same configuration;
* ``sample_multiple_kernel_perturbations`` / ``sample_kernel_perturbations``: random walks of the
  RJMCMC proposal kernels *without* accept/reject.  The walk itself runs on the GPU (chain kernel
  with the ``force_accept`` option); the host only rebuilds the list of ``Perturbation`` objects from
  the recorded proposals;
* ``aggregate_perturbations``: net additions / removals of a list of perturbations.
"""
from __future__ import annotations

from copy import copy
from typing import List, Sequence, Tuple

import math

import numpy as np

from . import energies as E
from .custom_types import ImageWMaps, Perturbation
from .hip_api import MppContext
from .kernels import KERNEL_NAMES, make_kernels
from .mappings import ValueMapping
from .point_set import EPointsSet
from .shapes import Rectangle

PERTURBATION_LIGHT = {"move_proba": 0.1, "param_shift_proba": [0.1, 0.1, 0.1], "position_sigma": 1,
                      "param_sigmas": [0.02, 0.02, 0.02], "point_number_sigma": 0.1, "no_addition": True}
PERTURBATION_MEDIUM = {"move_proba": 0.5, "param_shift_proba": [0.5, 0.5, 0.5], "position_sigma": 5,
                       "param_sigmas": [0.1, 0.1, 0.1], "point_number_sigma": 1.0}
PERTURBATION_HP_MEDIUM = {"move_proba": 0.8, "param_shift_proba": [0.9, 0.9, 0.9], "position_sigma": 5,
                          "param_sigmas": [0.1, 0.1, 0.1], "point_number_sigma": 1.0}
PERTURBATION_MEDIUM_OVERLAP = {"move_proba": 0.8, "param_shift_proba": [0.9, 0.9, 0.9], "position_sigma": 5,
                               "param_sigmas": [0.1, 0.1, 0.1], "point_number_sigma": 5.0, "make_overlap": 0.9}
PERTURBATION_STRONG = {"move_proba": 0.9, "param_shift_proba": [0.9, 0.9, 0.9], "position_sigma": 20,
                       "param_sigmas": [0.5, 0.5, 0.5], "point_number_sigma": 10.0}


def sample_perturbations(image_data: ImageWMaps = None, gt_rectangles: List[Rectangle] = None,
                         rng: np.random.Generator = None, image_shape: Tuple[int, int] = None,
                         mappings: List[ValueMapping] = None, move_proba: float = None,
                         param_shift_proba: List[float] = None, position_sigma: float = None,
                         param_sigmas: List[float] = None, make_overlap: float = None, no_addition: bool = False,
                         point_number_sigma: float = None, n_samples: int = 1):
    """Reference ``perturbation_sampler.py:58-122``; same draws in the same order."""
    if image_data is not None:
        gt_rectangles, image_shape, mappings = image_data.gt_config, image_data.shape, image_data.mappings
    assert gt_rectangles is not None and image_shape is not None and mappings is not None
    names = Rectangle.PARAMETERS
    n0 = len(gt_rectangles)
    out = []
    for _ in range(n_samples):
        pts = [copy(p) for p in gt_rectangles]
        target = int(np.clip(rng.normal(n0, point_number_sigma), a_min=0, a_max=1e4))
        if no_addition:
            target = int(np.clip(target, 0, n0))
        if target < n0:
            keep = rng.choice(range(n0), size=target, replace=False)
            pts = [pts[i] for i in keep]
        else:
            for _k in range(target - n0):
                if make_overlap is not None and rng.random() <= make_overlap:
                    pts.append(copy(pts[int(rng.choice(len(pts)))]))
                else:
                    pos = rng.integers((0, 0), image_shape)
                    vals = {nm: rng.uniform(m.v_min, m.v_max) for nm, m in zip(names, mappings)}
                    pts.append(Rectangle(x=pos[0], y=pos[1], **vals))
        hi = (image_shape[0] - 1, image_shape[1] - 1)
        for p in pts:
            if rng.random() < move_proba:
                shift = rng.normal(0, position_sigma, size=2)
                p.x, p.y = np.clip((p.x + shift[0], p.y + shift[1]), (0, 0), hi).astype(int)
            for i, (m, nm) in enumerate(zip(mappings, names)):
                if rng.random() < param_shift_proba[i]:
                    span = m.v_max - m.v_min
                    v = getattr(p, nm) + rng.normal(0, param_sigmas[i] * span)
                    if m.is_cyclic:
                        v = ((v - m.v_min) % span) + m.v_min
                    setattr(p, nm, np.clip(v, m.v_min, m.v_max))
        out.append(pts)
    return out


class DummyKernel:
    pass


def aggregate_perturbations(perturbations: Sequence[Perturbation]) -> Perturbation:
    """Net effect of a sequence (reference ``perturbation_sampler.py:176-211``): an addition cancels an
    earlier removal of the same object and vice versa."""
    added, removed = {}, {}          # dicts keep insertion order; keys hash by identity
    for p in perturbations:
        adds = p.addition if isinstance(p.addition, list) else ([] if p.addition is None else [p.addition])
        rems = p.removal if isinstance(p.removal, list) else ([] if p.removal is None else [p.removal])
        for q in adds:
            if q in removed:
                del removed[q]
            else:
                added[q] = True
        for q in rems:
            if q in added:
                del added[q]
            else:
                removed[q] = True
    return Perturbation(type=DummyKernel, removal=list(removed), addition=list(added))


def _clip_mark(mapping, v: float) -> float:
    """ValueMapping.clip (shape_net/mappings.py:52-58) in the arithmetic of the device code"""
    lo, hi = float(mapping.v_min), float(mapping.v_max)
    if mapping.is_cyclic:
        return math.fmod(v - lo, hi - lo) + (hi - lo if math.fmod(v - lo, hi - lo) < 0 else 0.0) + lo
    return lo if v < lo else (hi if v > hi else v)


def _clip_int(v: float, hi: int) -> int:
    return int(0.0 if v < 0.0 else (float(hi) if v > hi else v))


def split_rectangles(p: Rectangle, pos_delta, shape_delta, shape, mappings):
    """the two rectangles a split makes of ``p`` (split_and_merge_kernels.py:56-73)"""
    H, W = shape[:2]
    marks = (p.size, p.ratio, p.angle)
    out = []
    for sgn in (-1.0, 1.0):
        m = [_clip_mark(mp, mk + sgn * d) for mp, mk, d in zip(mappings, marks, shape_delta)]
        out.append(Rectangle(_clip_int(p.x + sgn * pos_delta[0], H - 1), _clip_int(p.y + sgn * pos_delta[1], W - 1),
                             size=m[0], ratio=m[1], angle=m[2]))
    return out


def merged_rectangle(p0: Rectangle, p1: Rectangle, shape, mappings) -> Rectangle:
    """split_and_merge_kernels.py:128-135 (the column is clipped with shape[0] upstream; reproduced)"""
    H = shape[0]
    m = [_clip_mark(mp, (a + b) / 2.0) for mp, a, b in zip(mappings, (p0.size, p0.ratio, p0.angle),
                                                          (p1.size, p1.ratio, p1.angle))]
    return Rectangle(_clip_int((p0.x + p1.x) / 2.0, H - 1), _clip_int((p0.y + p1.y) / 2.0, H - 1),
                     size=m[0], ratio=m[1], angle=m[2])


def _walk(ctx: MppContext, start: Sequence[Rectangle], n_iter: int, seed: int, chain: int, shape=None, mappings=None):
    """n_iter always-applied kernel proposals from `start`; returns (perturbations, final points)."""
    xy = np.array([[p.x, p.y] for p in start], dtype=np.int32).reshape(-1, 2)
    mk = np.array([[p.size, p.ratio, p.angle] for p in start], dtype=np.float64).reshape(-1, 3)
    ctx.set_points(0, xy, mk)
    ctx.set_schedule(1.0, 1.0, 0.0)
    state = list(start)
    perts = []
    if n_iter <= 0:
        return perts, state
    out, props = ctx.run(n_iter, seed, chain0=chain, trace_tile=0)
    for o, pr in zip(out, props):
        k, t = int(pr["kernel"]), int(pr["target"])
        kind = KERNEL_NAMES[k]
        is_birth, is_death = k in (0, 2), k in (1, 3)
        if is_birth:
            new = Rectangle(int(pr["ax"]), int(pr["ay"]), size=float(pr["as"]), ratio=float(pr["ar"]), angle=float(pr["aa"]))
            perts.append(Perturbation(type=kind, addition=new))
            state.append(new)
        elif t < 0:
            perts.append(Perturbation(type=kind))                  # empty configuration: nothing to do
        elif k == 8:                                               # split: state[t] -> a0 (same slot), a1 appended
            pd, sd = (float(pr["aux0"]), float(pr["aux1"])), (float(pr["as"]), float(pr["ar"]), float(pr["aa"]))
            a0, a1 = split_rectangles(state[t], pd, sd, shape, mappings)
            perts.append(Perturbation(type=kind, removal=state[t], addition=[a0, a1],
                                      data={"pos_delta": pd, "shape_delta": sd}))
            state[t] = a0
            state.append(a1)
        elif k == 9:                                               # merge: q takes p0's slot, p1 is swap-removed
            j = int(pr["param_id"])
            if j < 0:
                perts.append(Perturbation(type=kind))
                continue
            q = merged_rectangle(state[t], state[j], shape, mappings)
            perts.append(Perturbation(type=kind, removal=[state[t], state[j]], addition=q))
            state[t] = q
            state[j] = state[-1]
            state.pop()
        elif is_death:
            perts.append(Perturbation(type=kind, removal=state[t]))
            state[t] = state[-1]
            state.pop()
        else:
            new = Rectangle(int(pr["ax"]), int(pr["ay"]), size=float(pr["as"]), ratio=float(pr["ar"]), angle=float(pr["aa"]))
            if k == 4:
                data = {"delta": (float(pr["aux0"]), float(pr["aux1"]))}
            elif k == 6:
                data = {"param_id": int(pr["param_id"]), "delta": float(pr["aux0"])}
            elif k == 7:
                data = {"param_id": int(pr["param_id"]), "new_param_class_value": int(pr["new_class"])}
            else:
                data = None
            perts.append(Perturbation(type=kind, removal=state[t], addition=new, data=data))
            state[t] = new
    return perts, state


def sample_multiple_kernel_perturbations(image_data: ImageWMaps, n_samples: int, rng: np.random.Generator,
                                         energy_setup, iter_per_point: float, return_perturbations: bool = False,
                                         aggregate_pert: bool = False, use_split_merge: bool = False, device: int = 0):
    """Reference ``perturbation_sampler.py:125-149``.  Every sample starts again from the ground truth."""
    start = list(image_data.gt_config_set) if image_data.gt_config_set is not None else list(image_data.gt_config)
    unit, pair = energy_setup.make_energies(image_data)
    ctx = MppContext(device, point_capacity=max(256, 4 * len(start) + 64))
    ctx.set_maps(image_data.detection_map, image_data.param_dist_maps)
    if E.classic_image(unit) is not None:
        ctx.set_image(E.classic_image(unit))
    ctx.set_model(E.build_model_desc(unit, pair, None), image_data.mappings)
    ctx.set_kernels(make_kernels(image_data.mappings, intensity=1.0, use_split_merge=use_split_merge))
    ctx.set_option("force_accept", 1)
    n_iter = int(iter_per_point * len(start))
    seed = int(rng.integers(0, 2 ** 63 - 1))
    results, perts_out = [], []
    if (aggregate_pert or not return_perturbations) and n_samples > 1 and n_iter > 0:
        # Only the net effect of each walk is wanted: all walks run as chains s = 0..n_samples-1 of ONE launch over
        # replicas of the tile (the chains the loop below runs one launch at a time), and the net removals /
        # additions are read off the final configurations.
        finals = _walks_batched(image_data, start, n_samples, n_iter, seed, unit, pair, use_split_merge, device)
        for final, (removed, added) in finals:
            results.append(EPointsSet(final, image_data.shape, unit, pair, image_data=image_data, _ctx=ctx))
            perts_out.append(Perturbation(type=DummyKernel, removal=removed, addition=added))
        return perts_out if return_perturbations else results
    for s in range(n_samples):
        perts, final = _walk(ctx, start, n_iter, seed, chain=s, shape=image_data.shape, mappings=image_data.mappings)
        results.append(EPointsSet(final, image_data.shape, unit, pair, image_data=image_data, _ctx=ctx))
        perts_out.append(aggregate_perturbations(perts) if aggregate_pert else perts)
    return perts_out if return_perturbations else results


def _walks_batched(image_data, start, n_samples, n_iter, seed, unit, pair, use_split_merge, device):
    """-> per chain: (final configuration, (removed start points, added points)); points that end where they started
    (same five values) count as kept, which is the aggregate's energy-relevant content"""
    wctx = MppContext(device, point_capacity=max(256, 4 * len(start) + 64), replicas=n_samples)
    wctx.set_maps(image_data.detection_map, image_data.param_dist_maps)
    if E.classic_image(unit) is not None:
        wctx.set_image(E.classic_image(unit))
    wctx.set_model(E.build_model_desc(unit, pair, None), image_data.mappings)
    xy = np.array([[p.x, p.y] for p in start], dtype=np.int32).reshape(-1, 2)
    mk = np.array([[p.size, p.ratio, p.angle] for p in start], dtype=np.float64).reshape(-1, 3)
    for i in range(n_samples):
        wctx.set_points(i, xy, mk)
    wctx.set_kernels(make_kernels(image_data.mappings, intensity=1.0, use_split_merge=use_split_merge),
                     intensity=np.ones(n_samples))
    wctx.set_option("force_accept", 1)
    wctx.set_schedule(1.0, 1.0, 0.0)
    wctx.run(n_iter, seed, chain0=0)
    out = []
    for i in range(n_samples):
        fxy, fmk = wctx.get_points(i)
        pool = {}
        for p in start:
            pool.setdefault((p.x, p.y, p.size, p.ratio, p.angle), []).append(p)
        final, added = [], []
        for (x, y), (sz, ra, an) in zip(fxy.tolist(), fmk.tolist()):
            same = pool.get((x, y, sz, ra, an))
            if same:
                final.append(same.pop(0))
            else:
                q = Rectangle(int(x), int(y), size=float(sz), ratio=float(ra), angle=float(an))
                final.append(q); added.append(q)
        removed = [p for p in start if any(p is r for r in pool.get((p.x, p.y, p.size, p.ratio, p.angle), []))]
        out.append((final, (removed, added)))
    wctx.close()
    return out


def sample_kernel_perturbations(image_data: ImageWMaps, energy_setup, iter_per_point: float, points: EPointsSet,
                                rng: np.random.Generator, aggregate_pert: bool = False):
    """Reference ``perturbation_sampler.py:152-169`` for one sample starting from ``points``."""
    data = ImageWMaps(name=image_data.name, shape=image_data.shape, image=image_data.image,
                      detection_map=image_data.detection_map, param_dist_maps=image_data.param_dist_maps,
                      mappings=image_data.mappings, param_names=image_data.param_names, gt_config=list(points))
    perts = sample_multiple_kernel_perturbations(data, 1, rng, energy_setup, iter_per_point, return_perturbations=True,
                                                 aggregate_pert=aggregate_pert)[0]
    new_points = points.apply_perturbation(aggregate_perturbations(perts) if not aggregate_pert else perts)
    return new_points, perts
