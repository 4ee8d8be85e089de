"""Proposal-kernel mixture of the RJMCMC sampler.

Restates the reference's ``models/mpp/rjmcmc_sampler/kernels/make_kernels.py:13-177``:
eight kernels (uniform / data-driven birth and death, Gaussian / data-driven
translation, Gaussian / data-driven mark transform) whose selection
probabilities come from a weight tree.  The kernels themselves run on the GPU
(``csrc/mpp_sampler.hip``); this module only produces their parameters.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

from .mappings import ValueMapping

K_UBIRTH, K_UDEATH, K_DBIRTH, K_DDEATH, K_GTRANS, K_DTRANS, K_GTRANSF, K_DTRANSF, K_SPLIT, K_MERGE = range(10)
KERNEL_NAMES = ["UniformBirth", "UniformDeath", "DataBirth", "DataDeath", "GaussianTranslation",
                "DataDrivenTranslation", "GaussianShapeTransform", "DataDrivenShapeTransform", "Split", "Merge"]

BASE_KERNEL_WEIGHTS = {
    "bd_weight": 1, "uniform_bd_weight": 1, "data_bd_weight": 2, "ms_weight": 1,
    "translation_weight": 1, "gaussian_translation_weight": 1, "data_translation_weight": 2,
    "transformation_weight": 1, "gaussian_transformation_weight": 1, "data_transformation_weight": 2,
}


@dataclass
class KernelDesc:
    p_kernel: np.ndarray                  # [10]; the last two (split, merge) are 0 unless use_split_merge
    intensity: float
    vmin: np.ndarray                      # [3]
    vmax: np.ndarray
    cyclic: np.ndarray                    # [3] int
    edges: np.ndarray                     # [3, 32]
    sigma_trans: float = 2.0              # make_kernels.py:118
    sigma_transform: float = 0.1          # make_kernels.py:130
    max_delta: int = 8                    # make_kernels.py:124
    split_radius: float = 16.0            # make_kernels.py:148
    split_sigma: float = 0.1              # make_kernels.py:150


def _l1(v: Sequence[float]) -> np.ndarray:
    a = np.asarray(v, dtype=float)
    return a / np.sum(np.abs(a))


def make_kernels(mappings: List[ValueMapping], intensity: float, use_split_merge: bool = False,
                 kernel_weights: Dict[str, float] = None) -> KernelDesc:
    w = kernel_weights or BASE_KERNEL_WEIGHTS
    if use_split_merge:                   # make_kernels.py:75-77
        p_bd, p_ms, p_trl, p_trf = _l1([w["bd_weight"], w["ms_weight"], w["translation_weight"], w["transformation_weight"]])
    else:
        p_bd, p_trl, p_trf = _l1([w["bd_weight"], w["translation_weight"], w["transformation_weight"]])
        p_ms = 0.0
    p_bd_unif, p_bd_data = _l1([w["uniform_bd_weight"], w["data_bd_weight"]])
    p_trl_g, p_trl_d = _l1([w["gaussian_translation_weight"], w["data_translation_weight"]])
    p_trf_g, p_trf_d = _l1([w["gaussian_transformation_weight"], w["data_transformation_weight"]])
    p = np.array([0.5 * p_bd_unif * p_bd, 0.5 * p_bd_unif * p_bd, 0.5 * p_bd_data * p_bd, 0.5 * p_bd_data * p_bd,
                  p_trl * p_trl_g, p_trl * p_trl_d, p_trf * p_trf_g, p_trf * p_trf_d, p_ms * 0.5, p_ms * 0.5])
    if abs(1 - p.sum()) < 1e-8:
        p = p / p.sum()
    if len(mappings) != 3 or any(m.n_classes != 32 for m in mappings):
        raise ValueError("the sampler is built for three marks of 32 classes each")
    return KernelDesc(
        p_kernel=p, intensity=float(intensity),
        vmin=np.array([m.v_min for m in mappings], dtype=float),
        vmax=np.array([m.v_max for m in mappings], dtype=float),
        cyclic=np.array([int(m.is_cyclic) for m in mappings], dtype=np.int32),
        edges=np.stack([np.asarray(m.feature_mapping, dtype=float) for m in mappings]))
