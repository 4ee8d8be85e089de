"""CPU oracle for the MPP / RJMCMC sampling path.  TEST INFRASTRUCTURE ONLY.

``oracle/mpp_oracle.c`` restates the reference's algorithm in plain C; this
module builds it with gcc and exposes it through ctypes.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only to check or to time it -- the product never does.

Parity status: pinned (reference known answers + tapes recorded from the
reference), except the shapely/GEOS polygon-intersection boundary which is
pinned by analytic known answers (see ``mpp_oracle.h``), and the scikit-image
0.18.1 rasterisation behind the classic image energies (energies/classics.py),
restated from its published algorithm: fixtures recorded from the reference
over that restatement pin everything above the primitive.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB_PATH = os.path.join(_BUILD, "liboracle.so")
_SRC = [os.path.join(_HERE, "mpp_oracle.c"), os.path.join(_HERE, "mpp_oracle.h")]

MAX_UNIT, MAX_PAIR, NCLASS, NKERNEL = 8, 2, 32, 10


class UnitTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("gated", C.c_int32), ("coef", C.c_double), ("p", C.c_double * 8)]


class PairTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("gated", C.c_int32), ("reduce", C.c_int32), ("_pad", C.c_int32),
                ("coef", C.c_double), ("max_dist", C.c_double), ("p", C.c_double * 2)]


class Model(C.Structure):
    _fields_ = [("n_unit", C.c_int32), ("n_pair", C.c_int32), ("combinator", C.c_int32), ("gate_term", C.c_int32),
                ("gate_thr", C.c_double), ("lin0", C.c_double),
                ("unit", UnitTerm * MAX_UNIT), ("pair", PairTerm * MAX_PAIR)]


class Kernels(C.Structure):
    _fields_ = [("p_kernel", C.c_double * NKERNEL), ("intensity", C.c_double), ("sigma_trans", C.c_double),
                ("sigma_transform", C.c_double), ("max_delta", C.c_int32), ("cyclic", C.c_int32 * 3),
                ("vmin", C.c_double * 3), ("vmax", C.c_double * 3), ("edges", (C.c_double * NCLASS) * 3),
                ("split_radius", C.c_double), ("split_sigma", C.c_double)]


class Proposal(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("target", C.c_int32), ("ax", C.c_int32), ("ay", C.c_int32),
                ("as_", C.c_double), ("ar", C.c_double), ("aa", C.c_double),
                ("aux0", C.c_double), ("aux1", C.c_double),
                ("param_id", C.c_int32), ("new_class", C.c_int32), ("u_accept", C.c_double)]


class StepOut(C.Structure):
    _fields_ = [("dE", C.c_double), ("fwd", C.c_double), ("bwd", C.c_double), ("log_alpha", C.c_double),
                ("T", C.c_double), ("accepted", C.c_int32), ("n_after", C.c_int32)]


PROPOSAL_DTYPE = np.dtype([("kernel", "<i4"), ("target", "<i4"), ("ax", "<i4"), ("ay", "<i4"),
                           ("as", "<f8"), ("ar", "<f8"), ("aa", "<f8"), ("aux0", "<f8"), ("aux1", "<f8"),
                           ("param_id", "<i4"), ("new_class", "<i4"), ("u_accept", "<f8")], align=True)
STEPOUT_DTYPE = np.dtype([("dE", "<f8"), ("fwd", "<f8"), ("bwd", "<f8"), ("log_alpha", "<f8"), ("T", "<f8"),
                          ("accepted", "<i4"), ("n_after", "<i4")], align=True)
assert PROPOSAL_DTYPE.itemsize == C.sizeof(Proposal) and STEPOUT_DTYPE.itemsize == C.sizeof(StepOut)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, -ffp-contract=off so no FMA is formed behind our back)."""
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in _SRC):
        return _LIB_PATH
    os.makedirs(_BUILD, exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", _LIB_PATH, _SRC[0], "-lm"]
    subprocess.check_call(cmd)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(Model), C.POINTER(Kernels)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_count.argtypes = [C.c_void_p]
        L.orc_total_energy.restype = C.c_double
        L.orc_total_energy.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_delta.restype = C.c_double
        L.orc_delta.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_papangelou.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_temperature.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_replay.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_step_index.restype = C.c_int64
        L.orc_step_index.argtypes = [C.c_void_p]
        L.orc_follow.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_step_index.argtypes = [C.c_void_p, C.c_int64]
        L.orc_temperature.restype = C.c_double
        L.orc_temperature.argtypes = [C.c_void_p]
        L.orc_replay_forced.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_overlap.restype = C.c_double
        L.orc_overlap.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_naive_detection.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_set_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_contrast_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_outline.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_unit_value.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_unit_value.restype = C.c_double
        _lib = L
    return _lib


def _model_struct(desc) -> Model:
    """``desc``: anything shaped like mpp_cnn_rs_object_detection_amd.energies.ModelDesc (duck-typed)."""
    m = Model()
    m.n_unit, m.n_pair = len(desc.unit), len(desc.pair)
    m.combinator, m.gate_term, m.gate_thr, m.lin0 = desc.combinator, desc.gate_term, desc.gate_thr, desc.lin0
    for i, (kind, gated, coef, params) in enumerate(desc.unit):
        m.unit[i].kind, m.unit[i].gated, m.unit[i].coef = kind, gated, coef
        for j, p in enumerate(params):
            m.unit[i].p[j] = p
    for i, (kind, gated, red, coef, max_dist, params) in enumerate(desc.pair):
        t = m.pair[i]
        t.kind, t.gated, t.reduce, t.coef, t.max_dist = kind, gated, red, coef, max_dist
        for j, p in enumerate(params):
            t.p[j] = p
    return m


def _kernel_struct(kd) -> Kernels:
    k = Kernels()
    if kd is None:
        # the mark mappings travel with the kernel description; energies need them too, so a model
        # without kernels still gets the default size/ratio/angle bins (shape_net_model.py:80-85)
        for j, (lo, hi, cyc) in enumerate(((0.0, 32.0, 0), (0.0, 1.0, 0), (0.0, float(np.pi), 1))):
            k.cyclic[j], k.vmin[j], k.vmax[j] = cyc, lo, hi
            e = np.linspace(lo, hi, NCLASS + 1)[:-1]
            for i in range(NCLASS):
                k.edges[j][i] = float(e[i])
        return k
    for i in range(NKERNEL):
        k.p_kernel[i] = float(kd.p_kernel[i]) if i < len(kd.p_kernel) else 0.0
    k.split_radius, k.split_sigma = float(getattr(kd, "split_radius", 16.0)), float(getattr(kd, "split_sigma", 0.1))
    k.intensity, k.sigma_trans, k.sigma_transform, k.max_delta = kd.intensity, kd.sigma_trans, kd.sigma_transform, \
        kd.max_delta
    for j in range(3):
        k.cyclic[j], k.vmin[j], k.vmax[j] = int(kd.cyclic[j]), float(kd.vmin[j]), float(kd.vmax[j])
        for i in range(NCLASS):
            k.edges[j][i] = float(kd.edges[j][i])
    return k


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One tile: score maps + energy model + kernel mixture + a point configuration."""

    def __init__(self, shape, det, marks, model_desc, kernel_desc=None):
        H, W = int(shape[0]), int(shape[1])
        self.shape = (H, W)
        self._det = _f32(det)
        self._marks = [None, None, None] if marks is None else [_f32(m) for m in marks]
        self.n_terms = len(model_desc.unit) + len(model_desc.pair)
        self.names = list(getattr(model_desc, "names", []))
        m, k = _model_struct(model_desc), _kernel_struct(kernel_desc)
        self._h = lib().orc_create(H, W, _ptr(self._det), _ptr(self._marks[0]), _ptr(self._marks[1]),
                                   _ptr(self._marks[2]), C.byref(m), C.byref(k))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    # ---- classic image energies (energies/classics.py) -----------------------------------------------------------
    def set_image(self, img):
        """float32 [H][W][C] image behind ORC_U_CONTRAST (C = 1 or 3) / ORC_U_GRADIENT (np.gradient, [H][W][C/2][2])."""
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(self.shape[0], self.shape[1], -1)
        lib().orc_set_image(self._h, img.shape[2], _ptr(img))

    def contrast_masks(self, rect, dilation, gap, erode, cap=65536):
        """(fill, rim) pixel lists [(row, col)] of ContrastEnergy.compute_masks, row-major."""
        r = np.ascontiguousarray(rect, dtype=np.float64)
        fill, rim = np.zeros((cap, 2), np.int32), np.zeros((cap, 2), np.int32)
        nf, nr = C.c_int32(0), C.c_int32(0)
        lib().orc_contrast_masks(self._h, _ptr(r), int(dilation), int(gap), int(erode), cap, _ptr(fill), C.byref(nf),
                                 _ptr(rim), C.byref(nr))
        return fill[:nf.value].copy(), rim[:nr.value].copy()

    def outline(self, rect, eps=1e-8, cap=4096):
        """(pixels [(row, col)], normals) of GradientEnergy.compute_outline_and_normal, in outline order."""
        r = np.ascontiguousarray(rect, dtype=np.float64)
        rc, nrm = np.zeros((cap, 2), np.int32), np.zeros((cap, 2), np.float64)
        n = lib().orc_outline(self._h, _ptr(r), float(eps), cap, _ptr(rc), _ptr(nrm))
        return rc[:n].copy(), nrm[:n].copy()

    def unit_value(self, term, rect):
        """one unit term (kind, gated, coef, params) of a rectangle (x, y, size, ratio, angle)."""
        t = UnitTerm()
        t.kind, t.gated, t.coef = int(term[0]), int(term[1]), float(term[2])
        for j, p in enumerate(term[3]):
            t.p[j] = float(p)
        r = np.ascontiguousarray(rect, dtype=np.float64)
        return float(lib().orc_unit_value(self._h, C.byref(t), _ptr(r)))

    def set_points(self, xy, marks):
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        marks = np.ascontiguousarray(marks, dtype=np.float64).reshape(-1, 3)
        if lib().orc_set_points(self._h, len(xy), _ptr(xy), _ptr(marks)) != 0:
            raise AssertionError("point out of bounds")

    def get_points(self):
        n = lib().orc_count(self._h)
        xy, marks = np.zeros((n, 2), np.int32), np.zeros((n, 3), np.float64)
        lib().orc_get_points(self._h, n, _ptr(xy), _ptr(marks))
        return xy, marks

    def __len__(self):
        return lib().orc_count(self._h)

    def total_energy(self, return_vectors=False):
        n = len(self)
        vec = np.zeros((n, self.n_terms), np.float64)
        e = lib().orc_total_energy(self._h, _ptr(vec))
        return (e, vec) if return_vectors else e

    def delta(self, removal_slots: Sequence[int] = (), add_xy=None, add_marks=None) -> float:
        rem = np.ascontiguousarray(removal_slots, dtype=np.int32).reshape(-1)
        axy = np.zeros((0, 2), np.int32) if add_xy is None else np.ascontiguousarray(add_xy, np.int32).reshape(-1, 2)
        am = np.zeros((0, 3)) if add_marks is None else np.ascontiguousarray(add_marks, np.float64).reshape(-1, 3)
        return lib().orc_delta(self._h, len(rem), _ptr(rem), len(axy), _ptr(axy), _ptr(am))

    def papangelou(self):
        out = np.zeros(len(self), np.float64)
        lib().orc_papangelou(self._h, _ptr(out))
        return out

    def set_temperature(self, T, alpha, T_target=0.0):
        lib().orc_set_temperature(self._h, float(T), float(alpha), float(T_target))

    def replay(self, tape: np.ndarray) -> np.ndarray:
        tape = np.ascontiguousarray(tape, dtype=PROPOSAL_DTYPE)
        out = np.zeros(len(tape), STEPOUT_DTYPE)
        rc = lib().orc_replay(self._h, len(tape), _ptr(tape), _ptr(out))
        if rc != 0:
            raise RuntimeError(f"replay failed at step {-rc - 1}: target slot out of range")
        return out

    def replay_forced(self, tape: np.ndarray, accept: np.ndarray):
        """replay with the accept decision of every step imposed (tests: resync after a tie within the dE tolerance)"""
        tape = np.ascontiguousarray(tape, dtype=PROPOSAL_DTYPE)
        acc = np.ascontiguousarray(accept, dtype=np.int32)
        if lib().orc_replay_forced(self._h, len(tape), _ptr(tape), _ptr(acc), None) != 0:
            raise RuntimeError("forced replay: target slot out of range")

    def follow(self, tape: np.ndarray, seed: int, chain: int = 0):
        """Perform the tape's proposals with the oracle's own decisions; returns (step records, the proposals the oracle
        would have drawn itself at each step) -- see ``orc_follow``."""
        tape = np.ascontiguousarray(tape, dtype=PROPOSAL_DTYPE)
        out, native = np.zeros(len(tape), STEPOUT_DTYPE), np.zeros(len(tape), PROPOSAL_DTYPE)
        rc = lib().orc_follow(self._h, len(tape), int(seed), int(chain), _ptr(tape), _ptr(out), _ptr(native))
        if rc != 0:
            raise RuntimeError(f"follow failed at step {-rc - 1}: target slot out of range")
        return out, native

    def step_index(self) -> int:
        return int(lib().orc_step_index(self._h))

    def save(self):
        """(points, step index, temperature) -- enough to come back to this moment of the chain"""
        return self.get_points(), self.step_index(), float(lib().orc_temperature(self._h))

    def restore(self, saved, alpha: float, T_target: float = 0.0):
        (xy, marks), step, T = saved
        self.set_points(xy, marks)
        lib().orc_set_step_index(self._h, int(step))
        self.set_temperature(T, alpha, T_target)

    def run(self, n_steps: int, seed: int, chain: int = 0, trace: bool = False):
        out = np.zeros(n_steps if trace else 0, STEPOUT_DTYPE)
        props = np.zeros(n_steps if trace else 0, PROPOSAL_DTYPE)
        lib().orc_run(self._h, int(n_steps), int(seed), int(chain), _ptr(out) if trace else None,
                      _ptr(props) if trace else None)
        return (out, props) if trace else None

    def naive_detection(self, threshold: float, nms_dist: float = 6.0, cap: int = 1 << 16):
        xy, marks = np.zeros((cap, 2), np.int32), np.zeros((cap, 3), np.float64)
        n = lib().orc_naive_detection(self._h, float(threshold), float(nms_dist), cap, _ptr(xy), _ptr(marks))
        return xy[:n].copy(), marks[:n].copy()


def philox(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.zeros(4, np.uint32)
    lib().orc_philox(_ptr(c), _ptr(k), _ptr(o))
    return o


def overlap(r1, r2) -> float:
    a = np.ascontiguousarray(r1, dtype=np.float64)
    b = np.ascontiguousarray(r2, dtype=np.float64)
    return lib().orc_overlap(_ptr(a), _ptr(b))
