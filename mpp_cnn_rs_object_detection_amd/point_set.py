"""``EPointsSet``: the interacting point set, backed by the GPU.

Mirrors the reference's ``models/mpp/point_set/energy_point_set.py:18-166`` (same method names,
argument meaning and error behaviour): a configuration of marked points together with the
bookkeeping needed to evaluate energies and energy deltas.  The reference keeps Python sets of
point objects plus a graph of pair-energy objects; here the host keeps only the list of
``Rectangle`` objects in *slot order* (identity <-> slot), and every energy question goes
through ``libmppgpu.so`` (``mpp_total_energy`` / ``mpp_delta_batch`` / ``mpp_papangelou``).

The unit/pair "constructors" are the term descriptions of ``energies.py``; the score maps they
refer to are given once (``image_data``) instead of being captured inside each constructor.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import energies as E
from .custom_types import ImageWMaps, Perturbation
from .hip_api import MppContext, MppError
from .mappings import default_mappings
from .shapes import Point, Rectangle


def _as_list(x):
    if x is None:
        return []
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _rows(points: Sequence[Point]):
    xy = np.array([[p.x, p.y] for p in points], dtype=np.int32).reshape(-1, 2)
    marks = np.array([[getattr(p, "size", 0.0), getattr(p, "ratio", 0.0), getattr(p, "angle", 0.0)] for p in points],
                     dtype=np.float64).reshape(-1, 3)
    return xy, marks


class EPointsSet:
    def __init__(self, points: Iterable[Point], support_shape: Tuple[int, int],
                 unit_energies_constructors: List[E.UnitTerm], pair_energies_constructors: List[E.PairTerm],
                 image_data: Optional[ImageWMaps] = None, debug: bool = False, device: int = 0,
                 point_capacity: Optional[int] = None, _ctx: Optional[MppContext] = None):
        names = [t.name for t in unit_energies_constructors] + [t.name for t in pair_energies_constructors]
        assert len(names) == len(set(names))          # does not support same-name energies (reference :26-30)
        self.debug = debug
        self.support_shape = (int(support_shape[0]), int(support_shape[1]))
        self.ue_constructors = list(unit_energies_constructors)
        self.pe_constructors = list(pair_energies_constructors)
        self.image_data = image_data
        self.maximum_interaction_radius = max([t.max_dist for t in self.pe_constructors], default=0)
        self.mappings = image_data.mappings if image_data is not None else default_mappings()
        self._points: List[Point] = []
        self._index = {}
        pts = list(points)
        if _ctx is not None:
            self._ctx, self._owns_ctx = _ctx, False
        else:
            cap = point_capacity or max(256, 2 * len(pts) + 64)
            self._ctx, self._owns_ctx = MppContext(device, point_capacity=cap), True
            H, W = self.support_shape
            if image_data is not None:
                self._ctx.set_maps(image_data.detection_map, image_data.param_dist_maps)
            else:
                self._ctx.set_maps(np.zeros((H, W), np.float32), [np.zeros((H, W, 32), np.float32)] * 3)
            pic = E.classic_image(self.ue_constructors)        # the picture a classic image energy reads (classics.py)
            if pic is not None:
                self._ctx.set_image(pic)
        self._dirty = True
        for u in pts:
            self._check_bounds(u)
            self._index[u] = len(self._points)
            self._points.append(u)

    # -- plumbing ------------------------------------------------------------------------------
    def _check_bounds(self, u: Point):
        H, W = self.support_shape
        assert 0 <= u.x < H and 0 <= u.y < W          # point out of bounds (reference point_set.py:99)

    def _use(self, combinator):
        """Make this set (and this combinator) the one resident on the device.  Copies of a set share
        one GPU context (and its score maps); whichever copy asks a question uploads its points."""
        ctx = self._ctx
        key = (id(combinator) if combinator is not None else None, id(self.ue_constructors[0]) if self.ue_constructors else 0)
        if getattr(ctx, "_resident_model", None) != key:
            desc = E.build_model_desc(self.ue_constructors, self.pe_constructors, combinator)
            ctx.set_model(desc, self.mappings)
            ctx._resident_model = key
            ctx._keep_combinator = combinator
        self._last_combinator = combinator
        if self._dirty or getattr(ctx, "_resident_points", None) is not self:
            xy, marks = _rows(self._points)
            try:
                ctx.set_points(0, xy, marks)
            except MppError as e:
                raise AssertionError(str(e))
            ctx._resident_points = self
            self._dirty = False

    # -- container behaviour -----------------------------------------------------------------------
    def __len__(self):
        return len(self._points)

    def __contains__(self, u: Point):
        return u in self._index

    def __iter__(self):
        return iter(list(self._points))

    @property
    def points(self):
        """The reference exposes the spatial hash here; iterating / len() / ``in`` are what callers use."""
        return self

    def add(self, u: Point):
        self._check_bounds(u)
        if u in self._index:
            return
        self._index[u] = len(self._points)
        self._points.append(u)
        self._dirty = True

    def remove(self, u: Point):
        if u not in self._index:
            raise KeyError(u)
        slot = self._index.pop(u)
        last = self._points.pop()
        if last is not u:
            self._points[slot] = last
            self._index[last] = slot
        self._dirty = True

    def copy(self) -> "EPointsSet":
        new = EPointsSet(self._points, self.support_shape, self.ue_constructors, self.pe_constructors,
                         image_data=self.image_data, debug=self.debug, _ctx=self._ctx)
        if len(new) > self._ctx.get_option("point_capacity"):
            raise MemoryError("point capacity of the shared context exceeded")
        return new

    __copy__ = copy

    def get_neighbors(self, u: Point, radius: float, exclude_itself: bool = True):
        """Exact Euclidean neighbours (reference ``point_set.py:147-149``)."""
        out = set()
        for p in self._points:
            if exclude_itself and p is u:
                continue
            if np.hypot(p.x - u.x, p.y - u.y) <= radius:
                out.add(p)
        return out

    # -- energies ------------------------------------------------------------------------------------
    def total_energy(self, energy_combinator=None, return_vectors: bool = False):
        """Reference ``energy_point_set.py:80`` (plain sum when no combinator is given)."""
        self._use(energy_combinator)
        return self._ctx.total_energy(0, return_vectors=return_vectors)

    def energy_vectors(self):
        """Per-point energy vectors as a dict name -> list (reference ``compute_subset(return_vector=True)``)."""
        self._use(getattr(self, "_last_combinator", None))
        _, vec = self._ctx.total_energy(0, return_vectors=True)
        return {n: vec[:, i].tolist() for i, n in enumerate(self._ctx.names)}

    def _encode(self, p: Perturbation):
        removed, added = _as_list(p.removal), _as_list(p.addition)
        slots = []
        for r in removed:
            if r not in self._index:
                raise KeyError(r)                       # reference energy_point_set.py:88-100
            slots.append(self._index[r])
        for a in added:
            self._check_bounds(a)
        axy, am = _rows(added)
        return slots, axy, am

    def energy_delta(self, p: Perturbation, energy_combinator=None) -> float:
        return float(self.energy_delta_batch([p], energy_combinator)[0])

    def energy_delta_batch(self, perturbations: Sequence[Perturbation], energy_combinator=None) -> np.ndarray:
        """dE of many perturbations of the SAME configuration in one launch (one workgroup each)."""
        self._use(energy_combinator)
        enc = [self._encode(p) for p in perturbations]
        return self._ctx.delta_batch(0, [e[0] for e in enc], [e[1] for e in enc], [e[2] for e in enc])

    def energy_delta_vectors(self, perturbations: Sequence[Perturbation], names: Optional[Sequence[str]] = None):
        """Energy vectors of the points each perturbation touches, before and after it (``mpp_delta_vectors``):
        before, after [n][stride][n_terms], mask [n][stride] -- the input of a differentiable combinator.
        ``names`` orders the columns (``energy_setup.energy_names``); default: unit terms, then pair terms."""
        self._use(None)
        enc = [self._encode(p) for p in perturbations]
        before, after, mask = self._ctx.delta_vectors(0, [e[0] for e in enc], [e[1] for e in enc], [e[2] for e in enc],
                                                      len(self._ctx.names))
        if names is not None:
            cols = [self._ctx.names.index(n) for n in names]
            before, after = before[..., cols], after[..., cols]
        return before, after, mask

    def papangelou(self, u: Point, energy_combinator=None, remove_u_from_point_set: bool = False,
                   return_energy_delta: bool = False):
        """Reference ``energy_point_set.py:102-116``."""
        if u in self._index:
            if not remove_u_from_point_set:
                print(f"point {u} is already in current set, cannot compute papangelou conditional intensity")
                raise ValueError
            delta = -self.energy_delta(Perturbation(type=None, removal=u), energy_combinator)
        else:
            delta = self.energy_delta(Perturbation(type=None, addition=u), energy_combinator)
        return delta if return_energy_delta else float(np.exp(-delta))

    def papangelou_all(self, energy_combinator=None, return_energy_delta: bool = False) -> np.ndarray:
        """Papangelou intensity of every point of the set, removed from it (one launch)."""
        self._use(energy_combinator)
        d = self._ctx.papangelou(0)
        return d if return_energy_delta else np.exp(-d)

    # -- mutation ------------------------------------------------------------------------------------
    def apply_perturbation(self, p: Perturbation, inplace: bool = False) -> "EPointsSet":
        """Reference ``energy_point_set.py:118-154``."""
        new_x = self if inplace else self.copy()
        for r in _as_list(p.removal):
            if self.debug:
                assert r in new_x
            new_x.remove(r)
        for a in _as_list(p.addition):
            new_x.add(a)
        return new_x

    def unapply_perturbation(self, p: Perturbation) -> "EPointsSet":
        new_x = self.copy()
        for a in _as_list(p.addition):
            new_x.remove(a)
        for r in _as_list(p.removal):
            new_x.add(r)
        return new_x
