"""Marked points of the process: ``Point`` and ``Rectangle``.

Mirrors the reference's ``base/shapes/base_shapes.py:10-31`` and
``base/shapes/rectangle.py:11-37, :69-109`` (same field names and derived
properties) so that code written against the reference's objects keeps working.
Points hash by identity, as in the reference (``base_shapes.py:16-17``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple, Union

import numpy as np


@dataclass(eq=False)
class Point:
    x: int
    y: int

    def __hash__(self):
        return id(self)

    def get_coord(self) -> np.ndarray:
        return np.array([self.x, self.y])


@dataclass(eq=False)
class Rectangle(Point):
    size: float
    ratio: float
    angle: float
    PARAMETERS = ["size", "ratio", "angle"]

    def __hash__(self):
        return id(self)

    @property
    def length(self) -> float:
        return (2 * self.size) / (1 + self.ratio)

    @property
    def width(self) -> float:
        return self.ratio * self.length

    @property
    def poly_coord(self) -> np.ndarray:
        return rect_to_poly((self.x, self.y), short=self.length, long=self.width, angle=self.angle + np.pi / 2)

    @property
    def area(self) -> float:
        return self.length * self.width

    def as_row(self):
        return [float(self.x), float(self.y), float(self.size), float(self.ratio), float(self.angle)]


def rect_to_poly(center: Union[Tuple[int, int], np.ndarray], short: float, long: float, angle: float,
                 dilation: int = 0) -> np.ndarray:
    """(4, 2) corner coordinates; the local corners (+-short/2, +-long/2) are rotated by ``angle``
    and shifted to ``center`` (reference ``rectangle.py:69-100``)."""
    hs, hl = short / 2 + dilation, long / 2 + dilation
    local = np.array([[hs, hl], [hs, -hl], [-hs, -hl], [-hs, hl]])
    c, s = np.cos(angle), np.sin(angle)
    rot = np.array([[c, -s], [s, c]])
    return local @ rot.T + np.asarray(center)


def wla_to_sra(a, b, angle):
    return (a + b) / 2, a / b, angle


def sra_to_wla(s, r, angle):
    b = (2 * s) / (1 + r)
    return b * r, b, angle
