"""Energy setups and combinators of the MPP model, as flat term tables.

The reference wires Python objects: ``EnergySetup.make_energies`` returns lists
of unit / pair energy constructors (``models/mpp/energies/energy_setups/*.py``)
and an ``EnergyCombinationModel`` folds the per-point energy vectors into a
scalar (``models/mpp/energies/combination/*.py``).  The HIP sampler cannot call
Python per point, so the same classes here *describe* their terms; the
description is flattened into :class:`ModelDesc`, which is what crosses the
C ABI (``include/mpp_hip.h``: ``mpp_model``).

Every combinator the two shipped configs use has the form

    E(x) = sum_u F( lin0 + sum_k coef_k * g_k(u) * v_k(u) ),
    g_k = [v_gate(u) <= thr] if term k is gated else 1,  F = id  or  2*sigmoid-1

(hierarchical.py:21-32, :41-48; logistic.py:20-26), so ``ModelDesc`` stores one
coefficient and one gate flag per term.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# unit term kinds (same numbering as include/mpp_hip.h)
U_POSITION, U_SHAPE_REMAP, U_MARK_NEG, U_MARK_REMAP, U_AREA, U_RATIO_PRIOR, U_CONST, U_CONTRAST, U_GRADIENT = range(9)
# pair term kinds
P_OVERLAP, P_ALIGN, P_DIST_LE, P_DIST_LT = range(4)
REDUCE_MAX, REDUCE_MIN = 0, 1
C_LINEAR, C_LOGISTIC = 0, 1
MAX_UNIT, MAX_PAIR = 8, 2


@dataclass
class UnitTerm:
    """One unit energy (reference ``UnitEnergyConstructor``, base_energies.py:8-27)."""
    name: str
    kind: int
    params: Sequence[float] = ()
    #: the prepared picture a classic image energy reads ([H, W, C] float32; ``contrast_term`` / ``gradient_term``)
    image: Optional[np.ndarray] = None
    #: ``ContrastEnergy.q_fun`` (classics.py:109,162-163): a Python callable applied to the value.  A chain cannot call back
    #: into Python: ``classic_values`` applies it, ``build_model_desc`` refuses a term that carries one
    q_fun: Optional[object] = None


@dataclass
class PairTerm:
    """One pair energy (reference ``PairEnergyConstructor``, base_energies.py:42-69)."""
    name: str
    kind: int
    max_dist: float
    reduce: int = REDUCE_MAX
    params: Sequence[float] = ()


@dataclass
class ModelDesc:
    unit: List[Tuple[int, int, float, Tuple[float, ...]]]         # (kind, gated, coef, params)
    pair: List[Tuple[int, int, int, float, float, Tuple[float, ...]]]  # (kind, gated, reduce, coef, max_dist, params)
    combinator: int = C_LINEAR
    gate_term: int = -1
    gate_thr: float = 0.0
    lin0: float = 0.0
    names: List[str] = field(default_factory=list)

    @property
    def n_terms(self) -> int:
        return len(self.unit) + len(self.pair)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------- combinators
class EnergyCombinationModel:
    """Reference ``custom_types/energy.py:8-11``."""

    def compute(self, vectors: Dict[str, Sequence[float]]) -> float:
        raise NotImplementedError

    def coefficients(self, names: Sequence[str]):
        """-> (combinator, coef per name, gated per name, gate_name or None, gate_thr, lin0)."""
        raise NotImplementedError


@dataclass
class HierarchicalEnergyCombinator(EnergyCombinationModel):
    """Reference ``combination/hierarchical.py:13-32``."""
    weights_data: np.ndarray
    weights_prior: np.ndarray
    data_prior_weights: np.ndarray
    detection_threshold: float
    bias: float = 0.0

    def compute(self, vectors):
        pos = np.asarray(vectors["PositionEnergy"], dtype=float)
        ind = pos <= self.detection_threshold
        data = self.weights_data[0] * pos + ind * (self.weights_data[1] * np.asarray(vectors["ShapeEnergy"], float))
        prior = ind * (self.weights_prior[0] * np.asarray(vectors["RectangleOverlapEnergy"], float)
                       + self.weights_prior[1] * np.asarray(vectors["ShapeAlignmentEnergy"], float)
                       + self.weights_prior[2] * np.asarray(vectors["AreaPriorEnergy"], float))
        return float(np.sum(self.data_prior_weights[0] * data + self.data_prior_weights[1] * prior + self.bias))

    def coefficients(self, names):
        wd, wp, dp = (np.asarray(a, dtype=float) for a in (self.weights_data, self.weights_prior,
                                                             self.data_prior_weights))
        table = {
            "PositionEnergy": (dp[0] * wd[0], 0),
            "ShapeEnergy": (dp[0] * wd[1], 1),
            "RectangleOverlapEnergy": (dp[1] * wp[0], 1),
            "ShapeAlignmentEnergy": (dp[1] * wp[1], 1),
            "AreaPriorEnergy": (dp[1] * wp[2], 1),
        }
        missing = [n for n in names if n not in table]
        if missing:
            raise KeyError(f"HierarchicalEnergyCombinator has no weight for {missing}")
        return (C_LINEAR, [table[n][0] for n in names], [table[n][1] for n in names], "PositionEnergy",
                float(self.detection_threshold), float(self.bias))


@dataclass
class ManualHierarchicalEnergyCombinator(EnergyCombinationModel):
    """Reference ``combination/hierarchical.py:35-48``."""
    weights_dict: Dict[str, float]
    indicator_energy: str
    detection_threshold: float = 0.0

    def compute(self, vectors):
        ind_v = np.asarray(vectors[self.indicator_energy], dtype=float)
        ind = ind_v <= self.detection_threshold
        rest = sum(w * np.asarray(vectors[k], float) for k, w in self.weights_dict.items()
                   if k != self.indicator_energy)
        return float(np.sum(self.weights_dict[self.indicator_energy] * ind_v + ind * rest))

    def coefficients(self, names):
        coef = [float(self.weights_dict.get(n, 0.0)) for n in names]
        gated = [0 if n == self.indicator_energy else 1 for n in names]
        return C_LINEAR, coef, gated, self.indicator_energy, float(self.detection_threshold), 0.0


@dataclass
class LogisticEnergyCombinator(EnergyCombinationModel):
    """Reference ``combination/logistic.py:14-26`` (the bias enters once per term)."""
    weights: np.ndarray
    bias: float
    energy_names: List[str]

    def compute(self, vectors):
        v = np.array([vectors[k] for k in self.energy_names], dtype=float).T
        if len(v) == 0:
            return 0.0
        return float(np.sum(2 * sigmoid(np.sum(self.bias + np.asarray(self.weights, float) * v, axis=-1)) - 1))

    def coefficients(self, names):
        w = {n: float(x) for n, x in zip(self.energy_names, np.asarray(self.weights, dtype=float))}
        return (C_LOGISTIC, [w[n] for n in names], [0] * len(names), None, 0.0,
                float(self.bias) * len(self.energy_names))


def build_model_desc(unit_terms: Sequence[UnitTerm], pair_terms: Sequence[PairTerm],
                     combinator: Optional[EnergyCombinationModel]) -> ModelDesc:
    """Flatten (unit terms, pair terms, combinator) for the C ABI.  ``combinator=None`` is the
    plain sum the reference uses when no combinator is passed (energy_graph.py:130-131)."""
    if len(unit_terms) > MAX_UNIT or len(pair_terms) > MAX_PAIR:
        raise ValueError(f"at most {MAX_UNIT} unit and {MAX_PAIR} pair terms")
    names = [t.name for t in unit_terms] + [t.name for t in pair_terms]
    for t in unit_terms:
        if getattr(t, "q_fun", None) is not None and not getattr(t, "_q_fun_on_host", False):
            raise ValueError(f"{t.name}: q_fun is a Python callable -- available through classic_values() only, not inside a chain "
                             "(no shipped energy setup sets it, energy_setup_contrast.py:29-205)")
    if len(set(names)) != len(names):
        raise AssertionError(f"duplicate energy names in {names}")      # energy_graph.py:37-42
    if combinator is None:
        kind, coef, gated, gate_name, thr, lin0 = C_LINEAR, [1.0] * len(names), [0] * len(names), None, 0.0, 0.0
    else:
        kind, coef, gated, gate_name, thr, lin0 = combinator.coefficients(names)
    nu = len(unit_terms)
    gate_term = -1
    if gate_name is not None:
        gate_term = names.index(gate_name)
        if gate_term >= nu:
            raise ValueError("the gating energy must be a unit term")
    unit = [(t.kind, int(gated[i]), float(coef[i]), tuple(float(p) for p in t.params))
            for i, t in enumerate(unit_terms)]
    pair = [(t.kind, int(gated[nu + i]), int(t.reduce), float(coef[nu + i]), float(t.max_dist),
             tuple(float(p) for p in t.params)) for i, t in enumerate(pair_terms)]
    return ModelDesc(unit=unit, pair=pair, combinator=kind, gate_term=gate_term, gate_thr=thr, lin0=lin0,
                     names=names)


# --------------------------------------------------------------------------- setups
class EnergySetup:
    """Reference ``energies/energy_utils.py:14-37``."""

    @property
    def energy_names(self) -> List[str]:
        raise NotImplementedError

    def make_energies(self, image_data=None) -> Tuple[List[UnitTerm], List[PairTerm]]:
        raise NotImplementedError

    def load_calibration(self, save_dir: str):
        raise NotImplementedError

    @property
    def detection_threshold(self) -> float:
        raise NotImplementedError


class LegacyEnergySetup(EnergySetup):
    """Reference ``energy_setups/energy_setup_legacy.py:34-139``: 3 unit + 2 pair terms."""
    NAMES = ["PositionEnergy", "ShapeEnergy", "RectangleOverlapEnergy", "ShapeAlignmentEnergy", "AreaPriorEnergy"]

    def __init__(self, calibration_params=None, rewarding_priors: bool = True, energy_calibration: dict = None):
        self.calibration_params = calibration_params or {}
        self.rewarding_priors = rewarding_priors
        self.energy_calibration = energy_calibration

    @property
    def energy_names(self):
        return list(self.NAMES)

    def load_calibration(self, save_dir: str):
        with open(os.path.join(save_dir, "calibration.json")) as f:
            d = json.load(f)
        self.energy_calibration = {k: d[k] for k in ("detection_threshold", "param_dist_remap_coefs",
                                                     "param_dist_remap_intercepts", "min_area", "max_area")}

    def calibrate(self, image_configs, rng, save_path: str = None):
        """``energy_setup_legacy.py:88-123`` (the alignment histogram there is a figure only)"""
        from . import calibration as C
        from .shapes import Rectangle
        thr = C.calibrate_detection_threshold([c.detection_map for c in image_configs], [c.labels for c in image_configs],
                                              target=self.calibration_params.get("threshold_target"))
        coefs, intercepts = C.calibrate_param_dists([c.param_dist_maps for c in image_configs],
                                                    [c.gt_config for c in image_configs], image_configs[0].mappings,
                                                    Rectangle.PARAMETERS, rng)
        min_area, max_area = C.calibrate_min_area([c.gt_config for c in image_configs])
        self.energy_calibration = {"detection_threshold": thr, "param_dist_remap_coefs": coefs,
                                   "param_dist_remap_intercepts": intercepts, "min_area": min_area, "max_area": max_area}
        if save_path:
            with open(os.path.join(save_path, "calibration.json"), "w") as f:
                json.dump(self.energy_calibration, f, indent=1)

    @property
    def detection_threshold(self):
        return float(self.energy_calibration["detection_threshold"])

    def make_energies(self, image_data=None):
        c = self.energy_calibration
        unit = [
            UnitTerm(self.NAMES[0], U_POSITION, [c["detection_threshold"]]),
            UnitTerm(self.NAMES[1], U_SHAPE_REMAP, list(c["param_dist_remap_coefs"]) +
                     list(c["param_dist_remap_intercepts"])),
            UnitTerm(self.NAMES[4], U_AREA, [c["min_area"], c["max_area"]]),
        ]
        pair = [
            PairTerm(self.NAMES[2], P_OVERLAP, max_dist=32.0, reduce=REDUCE_MAX),
            PairTerm(self.NAMES[3], P_ALIGN, max_dist=16.0,
                     reduce=REDUCE_MIN if self.rewarding_priors else REDUCE_MAX,
                     params=[1.0 if self.rewarding_priors else 0.0]),
        ]
        return unit, pair


# ---- classic image energies (reference ``energies/classics.py``) -----------------------------------------------------
#: contrast_measure_type -> (device index, fac, default_value)   (classics.py:117-143)
CONTRAST_MEASURES = {"lafarge": (0, 1.0, 1e1), "craciun": (1, -1.0, 0.0), "craciun2": (2, -1.0, 0.0),
                     "mean": (3, -1.0, 0.0), "t-test": (4, -1.0, 0.0), "debug": (5, 1.0, -1.0)}


def contrast_term(name: str, image, dilation: int, contrast_measure_type: str, gap: int = 0, rgb: bool = False,
                  thresh: float = 0.0, erode: int = 0, normalize: bool = False, q_fun=None) -> UnitTerm:
    """``ContrastEnergy(...)`` (classics.py:100-149) as a term of the flat model: the measure's index, sign and default
    value, and the picture ``compute`` reads -- the channel mean for ``rgb=False`` (taken BEFORE any normalisation, as the
    reference does), the (optionally normalised) picture itself for ``rgb=True``."""
    if contrast_measure_type not in CONTRAST_MEASURES:
        raise ValueError(contrast_measure_type)
    idx, fac, default = CONTRAST_MEASURES[contrast_measure_type]
    image = np.asarray(image)
    if not rgb:
        pic = np.mean(image, axis=-1)[..., None]
    else:
        pic = image
        if normalize:
            pic = pic - np.mean(pic, axis=(0, 1))
            pic = pic / np.mean(np.abs(pic), axis=(0, 1))
    return UnitTerm(name, U_CONTRAST, [idx, int(dilation), int(gap), int(erode), float(thresh), fac, default],
                    image=np.ascontiguousarray(pic, dtype=np.float32), q_fun=q_fun)


def gradient_term(name: str, image, dilation: int = 1, eps: float = 1e-8, thresh: float = 0.0,
                  rgb: bool = False) -> UnitTerm:
    """``GradientEnergy(...)`` (classics.py:199-216): the picture handed to the device is ``np.gradient`` of the image,
    laid out [H, W, C, 2]."""
    image = np.asarray(image)
    img = image if rgb else np.mean(image, axis=-1)
    grad = np.moveaxis(np.array(np.gradient(img, axis=(0, 1))), 0, -1)
    H, W = image.shape[:2]
    return UnitTerm(name, U_GRADIENT, [float(thresh), float(eps)],
                    image=np.ascontiguousarray(grad.reshape(H, W, -1), dtype=np.float32))


def classic_image(unit_terms: Sequence[UnitTerm]):
    """The picture of the classic image energy among ``unit_terms`` (at most one), or None."""
    pics = [t.image for t in unit_terms if t.kind in (U_CONTRAST, U_GRADIENT)]
    if len(pics) > 1:
        raise ValueError("at most one classic image energy per model (they share the context's picture)")
    if pics and pics[0] is None:
        raise ValueError("a classic image energy without its picture: build it with contrast_term / gradient_term")
    return pics[0] if pics else None


def classic_values(term: UnitTerm, rects, mappings=None, device: int = 0) -> np.ndarray:
    """``[energy.compute(u) for u in rects]`` for a classic image energy, on the GPU (one context, one call)."""
    from .hip_api import MppContext
    from .mappings import default_mappings
    rects = list(rects)
    if not rects:
        return np.zeros(0)
    H, W = term.image.shape[:2]
    ctx = MppContext(device, point_capacity=max(256, len(rects)))
    ctx.set_maps(np.zeros((H, W), np.float32), [np.zeros((H, W, 32), np.float32)] * 3)
    ctx.set_image(term.image)
    plain = UnitTerm(term.name, term.kind, term.params, image=term.image)        # the value before q_fun
    ctx.set_model(build_model_desc([plain], [], None), mappings or default_mappings())
    ctx.set_points(0, np.array([[u.x, u.y] for u in rects], np.int32),
                   np.array([[u.size, u.ratio, u.angle] for u in rects], np.float64))
    _, vec = ctx.total_energy(0, return_vectors=True)
    vals = np.asarray(vec)[:, 0].copy()
    if term.q_fun is not None:
        # classics.py:154-163: q_fun is applied to the measured value only -- a rectangle without fill pixels returns the
        # measure's default value as it is
        default = float(term.params[6]) if term.kind == U_CONTRAST else None
        vals = np.array([v if (default is not None and v == default) else term.q_fun(v) for v in vals], dtype=np.float64)
    return vals


class ContrastMeasureEnergySetup(EnergySetup):
    """Reference ``energy_setups/energy_setup_contrast.py:29-161``: one classic image energy + two priors, two pair terms."""
    NAMES = ["ContrastEnergy", "OverlapPriorEnergy", "AlignmentPriorEnergy", "AreaPriorEnergy", "RatioPriorEnergy"]

    def __init__(self, contrast_type: str, learn_threshold: bool = False, rewarding_priors: bool = True,
                 manual_threshold=None):
        if contrast_type != "gradient" and contrast_type not in CONTRAST_MEASURES:
            raise ValueError(contrast_type)
        self.energy_cal = None
        self.contrast_type = contrast_type
        self.rewarding_priors = rewarding_priors
        self.learn_threshold = learn_threshold
        self.manual_threshold = manual_threshold

    @property
    def energy_names(self):
        return list(self.NAMES)

    def _make_contrast_energy(self, image_data, detection_thresh) -> UnitTerm:
        """``energy_setup_contrast.py:50-79``"""
        thresh = detection_thresh if detection_thresh is not None else 0.0
        image = np.asarray(image_data.image)
        if self.contrast_type == "gradient":
            return gradient_term(self.NAMES[0], image, dilation=1, rgb=True, thresh=thresh)
        # (the reference draws this noise for every contrast type and uses it for 't-test' only: same calls on numpy's global stream)
        noisy = np.clip(image + np.random.normal(0, 0.05, size=image.shape), 0, 1)
        t = self.contrast_type
        return contrast_term(self.NAMES[0], image if t != "t-test" else noisy, dilation=2,
                             gap=1 if t != "craciun" else 0, erode=1 if t != "craciun" else 0,
                             contrast_measure_type=t, rgb=t != "t-test", thresh=thresh, normalize=t == "t-test")

    def make_energies(self, image_data=None):
        c = self.energy_cal
        unit = [
            self._make_contrast_energy(image_data, c["detection_thresh"]),
            UnitTerm(self.NAMES[3], U_AREA, [c["min_area"], c["max_area"]]),
            UnitTerm(self.NAMES[4], U_RATIO_PRIOR, [0.5]),
        ]
        pair = [
            PairTerm(self.NAMES[1], P_OVERLAP, max_dist=32.0, reduce=REDUCE_MAX),
            PairTerm(self.NAMES[2], P_ALIGN, max_dist=16.0,
                     reduce=REDUCE_MIN if self.rewarding_priors else REDUCE_MAX,
                     params=[1.0 if self.rewarding_priors else 0.0]),
        ]
        return unit, pair

    def calibrate(self, image_configs, rng, save_path: str = None):
        """``energy_setup_contrast.py:107-141`` (the alignment histogram there is a figure only)"""
        from . import calibration as C
        thr = None
        if self.learn_threshold:
            thr = C.calibrate_contrast_threshold(self._make_contrast_energy, image_configs, rng)
        elif self.manual_threshold is not None:
            thr = self.manual_threshold
        min_area, max_area = C.calibrate_min_area([c.gt_config for c in image_configs])
        self.energy_cal = {"detection_thresh": thr, "min_area": min_area, "max_area": max_area}
        if save_path:
            with open(os.path.join(save_path, "calibration.json"), "w") as f:
                json.dump({"detection_thresh": thr, "min_area": min_area, "max_area": max_area}, f, indent=1)

    def load_calibration(self, save_dir: str):
        with open(os.path.join(save_dir, "calibration.json")) as f:
            d = json.load(f)
        self.energy_cal = {k: d[k] for k in ("detection_thresh", "min_area", "max_area")}

    @property
    def detection_threshold(self):
        return 0.5


class NoCalibrationEnergySetup(EnergySetup):
    """Reference ``energy_setups/energy_setup_no_calibration.py:31-159``."""

    def __init__(self, rewarding_priors: bool = True, ratio_prior: bool = False, calib_marks: bool = False):
        self.energy_calibration = None
        self.rewarding_priors = rewarding_priors
        self.ratio_prior = ratio_prior
        self.calib_marks = calib_marks
        self.NAMES = ["PositionEnergy", "SizeEnergy", "RatioEnergy", "AngleEnergy", "OverlapPriorEnergy",
                      "AlignmentPriorEnergy", "AreaPriorEnergy"]
        if ratio_prior:
            self.NAMES.append("RatioPriorEnergy")

    @property
    def energy_names(self):
        return list(self.NAMES)

    def load_calibration(self, save_dir: str):
        with open(os.path.join(save_dir, "calibration.json")) as f:
            d = json.load(f)
        self.energy_calibration = {"min_area": d["min_area"], "max_area": d["max_area"],
                                   "param_dist_remap_coefs": d.get("param_dist_remap_coefs"),
                                   "param_dist_remap_intercepts": d.get("param_dist_remap_intercepts")}

    def calibrate(self, image_configs, rng, save_path: str = None):
        """``energy_setup_no_calibration.py:112-144``"""
        from . import calibration as C
        from .shapes import Rectangle
        min_area, max_area = C.calibrate_min_area([c.gt_config for c in image_configs])
        coefs, intercepts = None, None
        if self.calib_marks:
            coefs, intercepts = C.calibrate_param_dists([c.param_dist_maps for c in image_configs],
                                                        [c.gt_config for c in image_configs], image_configs[0].mappings,
                                                        Rectangle.PARAMETERS, rng)
        self.energy_calibration = {"min_area": min_area, "max_area": max_area, "param_dist_remap_coefs": coefs,
                                   "param_dist_remap_intercepts": intercepts}
        if save_path:
            with open(os.path.join(save_path, "calibration.json"), "w") as f:
                json.dump({"detection_threshold": None, **self.energy_calibration}, f, indent=1)

    @property
    def detection_threshold(self):
        return 0.5

    def make_energies(self, image_data=None):
        c = self.energy_calibration
        unit = [UnitTerm(self.NAMES[0], U_POSITION, [0.0])]
        for k in range(3):
            if self.calib_marks:
                unit.append(UnitTerm(self.NAMES[1 + k], U_MARK_REMAP,
                                     [k, c["param_dist_remap_coefs"][k], c["param_dist_remap_intercepts"][k]]))
            else:
                unit.append(UnitTerm(self.NAMES[1 + k], U_MARK_NEG, [k]))
        unit.append(UnitTerm(self.NAMES[6], U_AREA, [c["min_area"], c["max_area"]]))
        if self.ratio_prior:
            unit.append(UnitTerm(self.NAMES[7], U_RATIO_PRIOR, [0.5]))
        pair = [
            PairTerm(self.NAMES[4], P_OVERLAP, max_dist=32.0, reduce=REDUCE_MAX),
            PairTerm(self.NAMES[5], P_ALIGN, max_dist=16.0,
                     reduce=REDUCE_MIN if self.rewarding_priors else REDUCE_MAX,
                     params=[1.0 if self.rewarding_priors else 0.0]),
        ]
        return unit, pair

    def ordered_terms(self):
        """Terms in ``energy_names`` order (unit and pair terms interleave in the reference's name list)."""
        unit, pair = self.make_energies()
        by_name = {t.name: t for t in unit + pair}
        return [by_name[n] for n in self.NAMES]


def normalize_l1(values: Sequence[float]) -> np.ndarray:
    """``utils/math_utils.py:45-49`` with l=1."""
    a = np.asarray(values, dtype=float)
    return a / np.sum(np.abs(a))


def hierarchical_from_manual(manual: Dict[str, float]) -> HierarchicalEnergyCombinator:
    """The ``manual`` train mode for the legacy setup (reference ``mpp_model.py:155-175``):
    each weight group is L1-normalised."""
    return HierarchicalEnergyCombinator(
        weights_data=normalize_l1([manual["PositionEnergy"], manual["ShapeEnergy"]]),
        weights_prior=normalize_l1([manual["RectangleOverlapEnergy"], manual["ShapeAlignmentEnergy"],
                                    manual["AreaPriorEnergy"]]),
        data_prior_weights=normalize_l1([manual["Data"], manual["Prior"]]),
        detection_threshold=float(manual.get("threshold", 0.0)))
