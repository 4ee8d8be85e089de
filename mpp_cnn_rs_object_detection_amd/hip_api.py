"""ctypes binding of ``libmppgpu.so`` (C ABI: ``include/mpp_hip.h``).

This is the only door between the Python host code and the HIP kernels.  There
is no CPU fallback: if the library is missing or no GPU is visible, importing
the symbols or creating a context raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from .energies import ModelDesc
from .kernels import KernelDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPP_LIB_PATH") or os.path.join(_HERE, "libmppgpu.so")   # override: diagnostic builds only

MAX_UNIT, MAX_PAIR, NCLASS, NKERNEL = 8, 2, 32, 10

#: every symbol include/mpp_hip.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "mpp_create", "mpp_destroy", "mpp_last_error", "mpp_set_stream", "mpp_synchronize", "mpp_set_option",
    "mpp_get_option", "mpp_set_maps", "mpp_set_image", "mpp_set_model", "mpp_set_kernels", "mpp_set_points", "mpp_get_points",
    "mpp_count", "mpp_get_points_all", "mpp_pack_detections", "mpp_total_energy", "mpp_delta_batch", "mpp_delta_vectors", "mpp_papangelou", "mpp_merge_score", "mpp_naive_init", "mpp_set_schedule",
    "mpp_replay", "mpp_run", "mpp_set_chain_keys", "mpp_step_index", "mpp_last_kernel_ms", "mpp_posnet_epilogue",
    "mpp_shapenet_epilogue", "mpp_posnet_epilogue_nhwc", "mpp_shapenet_epilogue_nhwc", "mpp_affine_relu", "mpp_nhwc_glue", "mpp_conv3x3_c32", "mpp_conv3x3_stem", "mpp_shapenet_heads", "mpp_quad_iou", "mpp_philox4x32", "mpp_abi_version",
]


class UnitTermC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("gated", C.c_int32), ("coef", C.c_double), ("p", C.c_double * 8)]


class PairTermC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("gated", C.c_int32), ("reduce", C.c_int32), ("_pad", C.c_int32),
                ("coef", C.c_double), ("max_dist", C.c_double), ("p", C.c_double * 2)]


class ModelC(C.Structure):
    _fields_ = [("n_unit", C.c_int32), ("n_pair", C.c_int32), ("combinator", C.c_int32), ("gate_term", C.c_int32),
                ("gate_thr", C.c_double), ("lin0", C.c_double),
                ("unit", UnitTermC * MAX_UNIT), ("pair", PairTermC * MAX_PAIR)]


class MappingsC(C.Structure):
    _fields_ = [("cyclic", C.c_int32 * 3), ("_pad", C.c_int32), ("vmin", C.c_double * 3), ("vmax", C.c_double * 3),
                ("edges", (C.c_double * NCLASS) * 3)]


class KernelsC(C.Structure):
    _fields_ = [("p_kernel", C.c_double * NKERNEL), ("sigma_trans", C.c_double), ("sigma_transform", C.c_double),
                ("max_delta", C.c_int32), ("_pad", C.c_int32), ("split_radius", C.c_double), ("split_sigma", C.c_double)]


PROPOSAL_DTYPE = np.dtype([("kernel", "<i4"), ("target", "<i4"), ("ax", "<i4"), ("ay", "<i4"),
                           ("as", "<f8"), ("ar", "<f8"), ("aa", "<f8"), ("aux0", "<f8"), ("aux1", "<f8"),
                           ("param_id", "<i4"), ("new_class", "<i4"), ("u_accept", "<f8")], align=True)
STEPOUT_DTYPE = np.dtype([("dE", "<f8"), ("fwd", "<f8"), ("bwd", "<f8"), ("log_alpha", "<f8"), ("T", "<f8"),
                          ("accepted", "<i4"), ("n_after", "<i4")], align=True)


class MppError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmppgpu error {code}: {message}")
        self.code = code


_lib = None


def _torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the SAME sonames (libamdhip64.so.7,
    libhsa-runtime64.so.1) as the system one libmppgpu.so is linked against.  Whichever is loaded first serves both:
    with torch first the process has ONE runtime (and device pointers are shared freely); with /opt/rocm's first, torch
    later stacks its own HIP on the foreign HSA and reports "No HIP GPUs are available".  So torch, when installed, is
    imported before the library is opened."""
    try:
        import torch  # noqa: F401
    except Exception:                     # torch missing: the sampler itself does not need it
        pass


def load_library(path: Optional[str] = None):
    """Load libmppgpu.so and declare the prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback for the sampler")
    _torch_runtime_first()
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    protos = {
        "mpp_create": (i32, [i32, C.POINTER(vp)]),
        "mpp_destroy": (i32, [vp]),
        "mpp_last_error": (C.c_char_p, [vp]),
        "mpp_set_stream": (i32, [vp, vp]),
        "mpp_synchronize": (i32, [vp]),
        "mpp_set_option": (i32, [vp, C.c_char_p, i64]),
        "mpp_get_option": (i64, [vp, C.c_char_p]),
        "mpp_set_maps": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, i32]),
        "mpp_set_image": (i32, [vp, i32, i32, vp, i32]),
        "mpp_set_model": (i32, [vp, C.POINTER(ModelC), C.POINTER(MappingsC)]),
        "mpp_set_kernels": (i32, [vp, C.POINTER(KernelsC), vp]),
        "mpp_set_points": (i32, [vp, i32, i32, vp, vp]),
        "mpp_get_points": (i32, [vp, i32, i32, C.POINTER(C.c_int32), vp, vp]),
        "mpp_count": (i32, [vp, i32, C.POINTER(C.c_int32)]),
        "mpp_get_points_all": (i32, [vp, i32, vp, vp, vp]),
        "mpp_pack_detections": (i32, [vp, i32, vp, vp, i32, vp, C.POINTER(C.c_int32)]),
        "mpp_total_energy": (i32, [vp, i32, C.POINTER(dbl), vp]),
        "mpp_delta_batch": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp]),
        "mpp_delta_vectors": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
        "mpp_papangelou": (i32, [vp, i32, vp]),
        "mpp_conv3x3_c32": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
        "mpp_shapenet_heads": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "mpp_conv3x3_stem": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
        "mpp_merge_score": (i32, [vp, C.c_double, i32, vp, vp, vp, vp, vp]),
        "mpp_naive_init": (i32, [vp, dbl, dbl]),
        "mpp_set_schedule": (i32, [vp, dbl, dbl, dbl]),
        "mpp_replay": (i32, [vp, i32, i32, vp, vp]),
        "mpp_run": (i32, [vp, i64, C.c_uint64, C.c_uint32, i32, vp, vp]),
        "mpp_set_chain_keys": (i32, [vp, i32, vp, vp]),
        "mpp_step_index": (i32, [vp, i32, C.POINTER(C.c_int64)]),
        "mpp_last_kernel_ms": (i32, [vp, C.POINTER(dbl)]),
        "mpp_posnet_epilogue": (i32, [vp, i32, i32, i32, i32, vp, dbl, dbl, vp]),
        "mpp_shapenet_epilogue": (i32, [vp, i32, i32, i32, i32, vp, vp]),
        "mpp_affine_relu": (i32, [vp, vp, i32, i32, i64, i32, vp, vp]),
        "mpp_posnet_epilogue_nhwc": (i32, [vp, i32, i32, i32, i32, vp, i32, dbl, dbl, vp]),
        "mpp_shapenet_epilogue_nhwc": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
        "mpp_nhwc_glue": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
        "mpp_quad_iou": (i32, [vp, i32, vp, i32, vp, vp, i32]),
        "mpp_philox4x32": (None, [vp, vp, vp]),
        "mpp_abi_version": (i32, []),
    }
    for name, (res, args) in protos.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if path == LIB_PATH:
        _lib = L
    return L


def model_struct(desc: ModelDesc) -> ModelC:
    m = ModelC()
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


def mappings_struct(mappings) -> MappingsC:
    s = MappingsC()
    for j, mp in enumerate(mappings):
        if mp.n_classes != NCLASS:
            raise ValueError("marks must have 32 classes")
        s.cyclic[j], s.vmin[j], s.vmax[j] = int(mp.is_cyclic), float(mp.v_min), float(mp.v_max)
        for i in range(NCLASS):
            s.edges[j][i] = float(mp.feature_mapping[i])
    return s


def kernels_struct(kd: KernelDesc) -> KernelsC:
    k = KernelsC()
    for i in range(NKERNEL):
        k.p_kernel[i] = float(kd.p_kernel[i]) if i < len(kd.p_kernel) else 0.0
    k.sigma_trans, k.sigma_transform, k.max_delta = float(kd.sigma_trans), float(kd.sigma_transform), int(kd.max_delta)
    k.split_radius, k.split_sigma = float(getattr(kd, "split_radius", 16.0)), float(getattr(kd, "split_sigma", 0.1))
    return k


def _is_torch(a) -> bool:
    return hasattr(a, "data_ptr") and hasattr(a, "device")


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


class MppContext:
    """One GPU context holding ``n_tiles`` tiles of equal shape (thin, 1:1 over the C ABI)."""

    def __init__(self, device: int = 0, point_capacity: Optional[int] = None, cell_capacity: Optional[int] = None,
                 spec_waves: Optional[int] = None, spec_lanes: Optional[int] = None, replicas: Optional[int] = None,
                 deep: Optional[int] = None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.mpp_create(int(device), C.byref(h))
        if rc != 0:
            raise MppError(rc, "mpp_create failed: no usable MI355X/HIP device" if rc == -3 else "mpp_create failed")
        self._h = h
        self.device = int(device)
        self._keep = []          # keeps borrowed device tensors / host arrays alive
        self.n_tiles = 0
        self.shape = (0, 0)
        self.n_terms = 0
        self.names: List[str] = []
        if point_capacity is not None:
            self.set_option("point_capacity", point_capacity)
        if cell_capacity is not None:
            self.set_option("cell_capacity", cell_capacity)
        if spec_waves is not None:
            self.set_option("spec_waves", spec_waves)
        if spec_lanes is not None:
            self.set_option("spec_lanes", spec_lanes)
        if replicas is not None:
            self.set_option("replicas", replicas)
        if deep is not None:
            self.set_option("deep", deep)
        if os.environ.get("MPP_HANDOVER_TILES"):           # (experiments: the largest launch that starts hot)
            self.set_option("handover_tiles", int(os.environ["MPP_HANDOVER_TILES"]))
        if os.environ.get("MPP_HANDOVER_AT"):
            self.set_option("handover_at", int(os.environ["MPP_HANDOVER_AT"]))

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise MppError(rc, self._L.mpp_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.mpp_destroy(self._h)
            self._h = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._check(self._L.mpp_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        return int(self._L.mpp_get_option(self._h, name.encode()))

    def set_stream(self, stream_handle: Optional[int]):
        self._check(self._L.mpp_set_stream(self._h, C.c_void_p(stream_handle) if stream_handle else None))

    def synchronize(self):
        self._check(self._L.mpp_synchronize(self._h))

    # -- inputs --------------------------------------------------------------------------------
    def set_maps(self, det, marks: Sequence):
        """det: [T,H,W] or [H,W]; marks: three [T,H,W,32] or [H,W,32]; numpy (copied) or torch
        tensors already on this GPU (borrowed, zero-copy)."""
        on_device = _is_torch(det)
        if on_device:
            if any(not _is_torch(m) for m in marks):
                raise TypeError("det and marks must all be device tensors or all be numpy arrays")
            import torch
            arrs = [a.contiguous() if a.dtype == torch.float32 else a.float().contiguous() for a in [det] + list(marks)]
            if not all(a.is_cuda for a in arrs):
                raise TypeError("torch maps must live on the GPU")
            if any(a.device.index != self.device for a in arrs):
                # borrowed pointers are dereferenced by kernels launched on this ctx's GPU; peer access is never enabled
                raise ValueError(f"borrowed maps live on {arrs[0].device} but this context is bound to GPU {self.device}")
            shape = tuple(arrs[0].shape)
        else:
            arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in [det] + list(marks)]
            shape = arrs[0].shape
        if len(shape) == 2:
            T, (H, W) = 1, shape
        else:
            T, H, W = shape
        for a in arrs[1:]:
            if tuple(a.shape[-3:]) != (H, W, NCLASS):
                raise ValueError(f"mark map of shape {tuple(a.shape)} does not match tile {H}x{W}x32")
        self._check(self._L.mpp_set_maps(self._h, T, H, W, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]),
                                         1 if on_device else 0))
        self._keep = arrs if on_device else []
        self.n_tiles, self.shape = self.get_option("n_chains"), (H, W)     # = T * replicas

    def set_image(self, img):
        """The picture behind the classic image energies (``energies/classics.py``): [T,H,W,C] or [H,W,C] float32, numpy
        (copied) or a torch tensor on this GPU (borrowed); after :meth:`set_maps`."""
        on_device = _is_torch(img)
        if on_device:
            import torch
            a = img.contiguous() if img.dtype == torch.float32 else img.float().contiguous()
            if not a.is_cuda or a.device.index != self.device:
                raise ValueError(f"a borrowed image must live on GPU {self.device}")
        else:
            a = np.ascontiguousarray(img, dtype=np.float32)
        if a.ndim == 3:
            a = a.reshape((1,) + tuple(a.shape))
        if a.ndim != 4 or tuple(a.shape[1:3]) != tuple(self.shape):
            raise ValueError(f"image of shape {tuple(a.shape)} does not match tiles {self.shape}")
        self._check(self._L.mpp_set_image(self._h, int(a.shape[0]), int(a.shape[3]), _ptr(a), 1 if on_device else 0))
        self._keep_img = a if on_device else None

    def set_model(self, desc: ModelDesc, mappings):
        m, mp = model_struct(desc), mappings_struct(mappings)
        self._check(self._L.mpp_set_model(self._h, C.byref(m), C.byref(mp)))
        self.n_terms = len(desc.unit) + len(desc.pair)
        self.names = list(desc.names)

    def set_kernels(self, kd: KernelDesc, intensity=None):
        k = kernels_struct(kd)
        inten = np.ascontiguousarray(np.broadcast_to(np.asarray(kd.intensity if intensity is None else intensity,
                                                                dtype=np.float64), (max(self.n_tiles, 1),)))
        self._check(self._L.mpp_set_kernels(self._h, C.byref(k), _ptr(inten)))

    def set_schedule(self, T0: float, alpha: float, T_target: float = 0.0):
        self._check(self._L.mpp_set_schedule(self._h, float(T0), float(alpha), float(T_target)))

    # -- configuration -------------------------------------------------------------------------
    def set_points(self, tile: int, xy, marks):
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        marks = np.ascontiguousarray(marks, dtype=np.float64).reshape(-1, 3)
        self._check(self._L.mpp_set_points(self._h, tile, len(xy), _ptr(xy), _ptr(marks)))

    def count(self, tile: int = 0) -> int:
        n = C.c_int32()
        self._check(self._L.mpp_count(self._h, tile, C.byref(n)))
        return n.value

    def get_points(self, tile: int = 0):
        n = self.count(tile)
        xy, marks = np.zeros((n, 2), np.int32), np.zeros((n, 3), np.float64)
        nn = C.c_int32()
        self._check(self._L.mpp_get_points(self._h, tile, n, C.byref(nn), _ptr(xy), _ptr(marks)))
        return xy, marks

    def counts(self) -> np.ndarray:
        """number of points of every tile (one device read)"""
        n = np.zeros(self.get_option("n_chains"), np.int32)
        self._check(self._L.mpp_get_points_all(self._h, 0, _ptr(n), None, None))
        return n

    def get_points_all(self):
        """[(xy, marks)] of every tile with five strided device reads in total"""
        n = self.counts()
        cap = int(n.max()) if len(n) else 0
        xy, marks = np.zeros((len(n), cap, 2), np.int32), np.zeros((len(n), cap, 3), np.float64)
        if cap:
            self._check(self._L.mpp_get_points_all(self._h, cap, _ptr(n), _ptr(xy), _ptr(marks)))
        return [(xy[t, :n[t]], marks[t, :n[t]]) for t in range(len(n))]

    def pack_detections(self, tile_ids, anchors, capacity: int, out) -> int:
        """Every tile's configuration as records (tile id, x + anchor_x, y + anchor_y, size, ratio, angle, 0) in the
        device tensor ``out`` [capacity+1, 7] float64 (row 0 = count): the all-gather's send buffer, filled without a
        host round trip.  Returns the number of records."""
        tile_ids = np.ascontiguousarray(tile_ids, dtype=np.int32).reshape(-1)
        anchors = np.ascontiguousarray(anchors, dtype=np.int32).reshape(-1, 2)
        if len(tile_ids) != len(anchors) or len(tile_ids) > self.n_tiles:
            raise ValueError("pack_detections: one id and one anchor per tile of the context")
        if not (_is_torch(out) and out.is_cuda and out.device.index == self.device and out.is_contiguous()
                and tuple(out.shape) == (capacity + 1, 7) and out.element_size() == 8):
            raise ValueError(f"pack_detections: out must be a contiguous [{capacity + 1}, 7] float64 tensor on GPU {self.device}")
        n = C.c_int32()
        self._check(self._L.mpp_pack_detections(self._h, len(tile_ids), _ptr(tile_ids), _ptr(anchors), int(capacity),
                                                _ptr(out), C.byref(n)))
        return n.value

    def naive_init(self, threshold: float, nms_distance: float = 6.0):
        self._check(self._L.mpp_naive_init(self._h, float(threshold), float(nms_distance)))

    # -- energies ------------------------------------------------------------------------------
    def total_energy(self, tile: int = 0, return_vectors: bool = False):
        e = C.c_double()
        vec = np.zeros((self.count(tile), self.n_terms), np.float64) if return_vectors else None
        self._check(self._L.mpp_total_energy(self._h, tile, C.byref(e), _ptr(vec)))
        return (e.value, vec) if return_vectors else e.value

    @staticmethod
    def _pack_cases(removals, add_xy, add_marks):
        n = len(removals)
        rem_off = np.zeros(n + 1, np.int32)
        add_off = np.zeros(n + 1, np.int32)
        rem_off[1:] = np.cumsum([len(r) for r in removals])
        add_off[1:] = np.cumsum([len(a) for a in add_xy])
        rem = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int32).reshape(-1) for r in removals] +
                                                  [np.zeros(0, np.int32)]), dtype=np.int32)
        axy = np.ascontiguousarray(np.concatenate([np.asarray(a, np.int32).reshape(-1, 2) for a in add_xy] +
                                                  [np.zeros((0, 2), np.int32)]), dtype=np.int32)
        am = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).reshape(-1, 3) for a in add_marks] +
                                                 [np.zeros((0, 3))]), dtype=np.float64)
        if len(rem) == 0:
            rem = np.zeros(1, np.int32)
        if len(axy) == 0:
            axy, am = np.zeros((1, 2), np.int32), np.zeros((1, 3))
        return n, rem_off, rem, add_off, axy, am

    def delta_batch(self, tile: int, removals: Sequence[Sequence[int]], add_xy: Sequence, add_marks: Sequence):
        n, rem_off, rem, add_off, axy, am = self._pack_cases(removals, add_xy, add_marks)
        out = np.zeros(n, np.float64)
        self._check(self._L.mpp_delta_batch(self._h, tile, n, _ptr(rem_off), _ptr(rem), _ptr(add_off), _ptr(axy),
                                            _ptr(am), _ptr(out)))
        return out

    def delta_vectors(self, tile: int, removals: Sequence[Sequence[int]], add_xy: Sequence, add_marks: Sequence,
                      n_terms: int):
        """before, after [n_cases][stride][n_terms] and mask [n_cases][stride] (see mpp_delta_vectors)"""
        n, rem_off, rem, add_off, axy, am = self._pack_cases(removals, add_xy, add_marks)
        stride = self.count(tile) + (int(np.max(np.diff(add_off))) if n else 0)
        stride = max(stride, 1)
        before = np.zeros((n, stride, n_terms), np.float64)
        after = np.zeros((n, stride, n_terms), np.float64)
        mask = np.zeros((n, stride), np.uint8)
        self._check(self._L.mpp_delta_vectors(self._h, tile, n, _ptr(rem_off), _ptr(rem), _ptr(add_off), _ptr(axy),
                                              _ptr(am), stride, _ptr(before), _ptr(after), _ptr(mask)))
        return before, after, mask

    def papangelou(self, tile: int = 0) -> np.ndarray:
        out = np.zeros(max(self.count(tile), 1), np.float64)
        self._check(self._L.mpp_papangelou(self._h, tile, _ptr(out)))
        return out[:self.count(tile)]

    # -- the chain -----------------------------------------------------------------------------
    def merge_score(self, distance: float):
        """``merge_patches(method='distance')`` and the two scorings around it for every tile of the ctx on the device
        (``mpp_merge_score``): -> [(xy, marks, dE)] of the survivors per tile (score = exp(-dE)), and the removed counts."""
        T = self.get_option("n_chains")
        cap = int(max(1, self.counts()[:T].max(initial=0)))
        n = np.zeros(T, np.int32)
        xy, marks = np.zeros((T, cap, 2), np.int32), np.zeros((T, cap, 3), np.float64)
        dE, removed = np.zeros((T, cap), np.float64), np.zeros(T, np.int32)
        self._check(self._L.mpp_merge_score(self._h, float(distance), cap, _ptr(n), _ptr(xy), _ptr(marks), _ptr(dE), _ptr(removed)))
        return [(xy[t, :n[t]].copy(), marks[t, :n[t]].copy(), dE[t, :n[t]].copy()) for t in range(T)], removed

    def replay(self, tile: int, tape: np.ndarray) -> np.ndarray:
        tape = np.ascontiguousarray(tape, dtype=PROPOSAL_DTYPE)
        out = np.zeros(len(tape), STEPOUT_DTYPE)
        self._check(self._L.mpp_replay(self._h, tile, len(tape), _ptr(tape), _ptr(out)))
        return out

    def run(self, n_steps: int, seed: int, chain0: int = 0, trace_tile: int = -1):
        if trace_tile >= 0:
            out = np.zeros(n_steps, STEPOUT_DTYPE)
            props = np.zeros(n_steps, PROPOSAL_DTYPE)
            self._check(self._L.mpp_run(self._h, int(n_steps), int(seed), int(chain0), trace_tile, _ptr(out),
                                        _ptr(props)))
            return out, props
        self._check(self._L.mpp_run(self._h, int(n_steps), int(seed), int(chain0), -1, None, None))
        return None

    def set_chain_keys(self, seeds=None, chains=None):
        """Per-chain Philox key and chain id (``mpp_set_chain_keys``); ``None``: back to ``run``'s seed and chain0 + tile."""
        if seeds is None or chains is None:
            self._check(self._L.mpp_set_chain_keys(self._h, 0, None, None))
            return
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(-1)
        chains = np.ascontiguousarray(chains, dtype=np.uint32).reshape(-1)
        if len(seeds) != len(chains):
            raise ValueError("one seed and one chain id per chain")
        self._check(self._L.mpp_set_chain_keys(self._h, len(seeds), _ptr(seeds), _ptr(chains)))

    def deep_stats(self) -> dict:
        """counters of the last run's deep rounds (all tiles): rounds, steps evaluated, rounds with a change, steps committed"""
        v = [self.get_option(f"deep_stat{i}") for i in range(4)]
        return {"rounds": v[0], "evaluated": v[1], "rounds_with_change": v[2], "committed": v[3]}

    def step_index(self, tile: int = 0) -> int:
        s = C.c_int64()
        self._check(self._L.mpp_step_index(self._h, tile, C.byref(s)))
        return s.value

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        self._check(self._L.mpp_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    # -- U-Net epilogues (device tensors) ---------------------------------------------------------
    def posnet_epilogue(self, pos_out, H: int, W: int, div_w: float, div_b: float, det_out):
        ldh, ldw = int(pos_out.shape[-2]), int(pos_out.shape[-1])
        self._check(self._L.mpp_posnet_epilogue(self._h, H, W, ldh, ldw, _ptr(pos_out), float(div_w), float(div_b),
                                                _ptr(det_out)))

    def shapenet_epilogue(self, logits, H: int, W: int, marks_out):
        ldh, ldw = int(logits.shape[-2]), int(logits.shape[-1])
        self._check(self._L.mpp_shapenet_epilogue(self._h, H, W, ldh, ldw, _ptr(logits), _ptr(marks_out)))

    def posnet_epilogue_nhwc(self, pos_out, H: int, W: int, div_w: float, div_b: float, det_out):
        """``posnet_epilogue`` on a [1,3,Hp,Wp] channels-last (NHWC memory) float32 / bfloat16 network output"""
        ldh, ldw, ch = nhwc_shape(pos_out)
        if ch != 3:
            raise ValueError("posnet output must have 3 channels")
        self._check(self._L.mpp_posnet_epilogue_nhwc(self._h, H, W, ldh, ldw, _ptr(pos_out), int(pos_out.element_size()),
                                                     float(div_w), float(div_b), _ptr(det_out)))

    def shapenet_epilogue_nhwc(self, logits, H: int, W: int, marks_out):
        """``shapenet_epilogue`` on a [1,32,Hp,Wp] channels-last (NHWC memory) float32 / bfloat16 head output"""
        ldh, ldw, ch = nhwc_shape(logits)
        if ch != NCLASS:
            raise ValueError(f"a shapenet head must have {NCLASS} channels")
        self._check(self._L.mpp_shapenet_epilogue_nhwc(self._h, H, W, ldh, ldw, _ptr(logits), int(logits.element_size()),
                                                       _ptr(marks_out)))

    def affine_relu(self, x, scale, shift):
        """x <- max(0, x * scale[c] + shift[c]) in place; x: contiguous NCHW float32 / bfloat16 CUDA tensor"""
        n, ch = int(x.shape[0]), int(x.shape[1])
        hw = int(x.shape[2]) * int(x.shape[3])
        self._check(self._L.mpp_affine_relu(self._h, _ptr(x), n * ch, ch, hw, int(x.element_size()), _ptr(scale), _ptr(shift)))

    def nhwc_glue(self, x0, x1=None, pad: int = 1, pool: bool = False, scale=None, shift=None, out_dtype=None, out=None):
        """One pass between two convolutions on channels-last activations (``mpp_nhwc_glue``):
        reflect_pad(relu(affine(maxpool2(cat(x0, x1))))).  x0 / x1: [1,C,H,W] CUDA tensors with channels_last
        strides;
        returns a [1,C0+C1,H',W'] channels_last tensor (``out``: write there, e.g. ``x0`` itself for pad = 0)."""
        import torch
        h0, w0, c0 = nhwc_shape(x0)
        c1 = 0
        if x1 is not None:
            h1, w1, c1 = nhwc_shape(x1)
            if (h1, w1) != (h0, w0) or x1.dtype != x0.dtype:
                raise ValueError("nhwc_glue: the two sources differ in size or element type")
        H, W = (h0 // 2, w0 // 2) if pool else (h0, w0)
        if out is None:
            out = torch.empty((1, H + 2 * pad, W + 2 * pad, c0 + c1), dtype=out_dtype or x0.dtype, device=x0.device)
            out = out.permute(0, 3, 1, 2)
        self._check(self._L.mpp_nhwc_glue(self._h, _ptr(x0), _ptr(x1), _ptr(out), H, W, c0, c1, pad, 1 if pool else 0,
                                          int(x0.element_size()), int(out.element_size()), _ptr(scale), _ptr(shift)))
        return out

    def conv3x3_c32(self, x0, wp, x1=None, in_scale=None, in_shift=None, out_scale=None, out_shift=None, relu: bool = True, out=None):
        """3x3 reflect-padded convolution to 32 channels on channels-last float32 activations (``mpp_conv3x3_c32``):
        x0 (and x1, the second half of a concatenation): [1,32,H,W] CUDA tensors with channels_last strides; wp: the repacked
        weights [C_in/32, 9, 32, 32]; returns relu(conv * out_scale + out_shift) as a [1,32,H,W] channels_last tensor."""
        import torch
        h, w, c = nhwc_shape(x0)
        if c != 32 or x0.dtype != torch.float32 or (x1 is not None and nhwc_shape(x1) != (h, w, 32)):
            raise ValueError("conv3x3_c32: float32 channels-last sources of 32 channels")
        if out is None:
            out = torch.empty((1, h, w, 32), dtype=torch.float32, device=x0.device).permute(0, 3, 1, 2)
        self._check(self._L.mpp_conv3x3_c32(self._h, _ptr(x0), _ptr(x1), h, w, _ptr(wp), _ptr(in_scale), _ptr(in_shift),
                                            _ptr(out_scale), _ptr(out_shift), 1 if relu else 0, _ptr(out)))
        return out

    def conv3x3_stem(self, x, wp, scale, shift):
        """Conv2d(3 -> 32, 3x3, reflect) + folded BatchNorm + ReLU of a U-Net's stem (``mpp_conv3x3_stem``): x [1,3,H,W] float32
        with channels_last strides, wp [9,3,32]; returns a [1,32,H,W] channels_last tensor."""
        import torch
        h, w, c = nhwc_shape(x)
        if c != 3 or x.dtype != torch.float32 or tuple(wp.shape) != (9, 3, 32):
            raise ValueError("conv3x3_stem: a float32 channels-last picture of 3 channels, weights [9,3,32]")
        out = torch.empty((1, h, w, 32), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        self._check(self._L.mpp_conv3x3_stem(self._h, _ptr(x), h, w, _ptr(wp), _ptr(scale), _ptr(shift), _ptr(out)))
        return out

    def shapenet_heads(self, h, w, b, H: int, W: int, marks):
        """ShapeNet's three 1x1 heads + biases + softmax in one pass (``mpp_shapenet_heads``): h [1,32,ldh,ldw] float32
        channels_last, w [3,32,32] (head, class, input channel), b [3,32]; marks: three [H,W,32] float32 CUDA tensors."""
        import torch
        ldh, ldw, c = nhwc_shape(h)
        if c != 32 or h.dtype != torch.float32 or tuple(w.shape) != (3, 32, 32) or tuple(b.shape) != (3, 32) or len(marks) != 3:
            raise ValueError("shapenet_heads: float32 channels-last activations of 32 channels, w [3,32,32], b [3,32]")
        if not (w.is_contiguous() and b.is_contiguous() and all(m.is_contiguous() and tuple(m.shape) == (H, W, 32) for m in marks)):
            raise ValueError("shapenet_heads: contiguous weights and [H,W,32] mark maps")
        self._check(self._L.mpp_shapenet_heads(self._h, H, W, ldh, ldw, _ptr(h), _ptr(w), _ptr(b), _ptr(marks[0]), _ptr(marks[1]),
                                               _ptr(marks[2])))

    # -- evaluation --------------------------------------------------------------------------------
    def quad_iou(self, a, b) -> np.ndarray:
        """IoU matrix [n][m] of convex quads a [n][8] and b [m][8] (-1: axis-aligned extents apart)."""
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 8))
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1, 8))
        out = np.zeros((len(a), len(b)), np.float64)
        self._check(self._L.mpp_quad_iou(self._h, len(a), _ptr(a), len(b), _ptr(b), _ptr(out), 0))
        return out


def nhwc_shape(x):
    """(H, W, C) of a [1,C,H,W] tensor whose memory is NHWC (channels_last strides); raises otherwise."""
    import torch
    if x.dim() != 4 or x.shape[0] != 1 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("expected a [1,C,H,W] tensor in channels_last memory format")
    return int(x.shape[2]), int(x.shape[3]), int(x.shape[1])


def philox(ctr, key) -> np.ndarray:
    L = load_library()
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.zeros(4, np.uint32)
    L.mpp_philox4x32(_ptr(c), _ptr(k), _ptr(o))
    return o
