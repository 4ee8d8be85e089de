"""Plain data carriers shared by the sampler's host code.

Mirror the reference's ``models/mpp/custom_types/{image_w_maps,perturbation,rjmcmc}.py``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type, Union

import numpy as np

from .mappings import ValueMapping
from .shapes import Point, Rectangle


@dataclass
class ImageWMaps:
    name: str
    shape: Tuple[int, int]
    image: Optional[np.ndarray]
    detection_map: Any              # [H, W] float32: numpy array or a torch tensor already on the GPU
    param_dist_maps: List[Any]      # 3 x [H, W, 32] float32
    mappings: List[ValueMapping]
    param_names: List[str]
    labels: Dict[str, Any] = None
    gt_config: List[Rectangle] = None
    gt_config_set: Any = None
    crop_data: Dict = None


@dataclass
class Perturbation:
    type: Any
    removal: Union[None, Point, List[Point]] = None
    addition: Union[None, Point, List[Point]] = None
    data: Optional[Dict[str, Any]] = None


@dataclass
class RJMCMCStateSummary:
    iter: int
    n_points: int
    temperature: float = None
    energy: Union[None, float] = None
    kernel: Union[None, int] = None
    move_accepted: Union[None, bool] = None
    alpha: Union[None, float] = None
    initial_energy: Union[None, float] = None
    proposed_energy: Union[None, float] = None
