"""``ValueMapping``: value <-> class-bin conversion of the three marks.

Mirrors the reference's ``models/shape_net/mappings.py:9-74`` (same attribute
and method names).  Bins are ``linspace(v_min, v_max, n+1)[:-1]``; a value maps
to the last bin whose lower edge it reaches.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


@dataclass
class ValueMapping:
    n_classes: int
    v_min: float
    v_max: float
    is_cyclic: bool = False

    def __post_init__(self):
        self.feature_mapping = np.linspace(self.v_min, self.v_max, num=self.n_classes + 1)[:-1]

    @property
    def range(self) -> float:
        return self.v_max - self.v_min

    def get_range(self) -> float:
        return self.v_max - self.v_min

    def get_step(self) -> float:
        return float(np.mean(np.diff(self.feature_mapping)))

    def clip(self, value: float) -> float:
        if not self.is_cyclic:
            return float(np.clip(value, self.v_min, self.v_max))
        return ((value - self.v_min) % self.range) + self.v_min

    def value_to_class(self, value):
        idx = np.searchsorted(self.feature_mapping, value, side="right") - 1
        if np.any(np.asarray(idx) < 0):
            raise ValueError(f"feature value {value} below v_min={self.v_min}")
        return idx

    def class_to_value(self, class_id):
        if hasattr(class_id, "detach"):
            class_id = class_id.cpu().detach().numpy()
        return self.feature_mapping[class_id]


def default_mappings() -> List[ValueMapping]:
    """size in [0,32), ratio in [0,1), angle in [0,pi) cyclic (reference ``shape_net_model.py:80-85``)."""
    return [ValueMapping(32, 0, 32), ValueMapping(32, 0, 1), ValueMapping(32, 0, np.pi, is_cyclic=True)]
