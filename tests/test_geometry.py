"""Analytic known answers for the rectangle-overlap energy (reference prior_energies.py:11-24).

The reference computes the intersection with shapely/GEOS, which is not available in the build
container; this is the boundary where the oracle is pinned by geometry instead of recorded values.
A rectangle (x, y, size, ratio, angle) has length 2*size/(1+ratio), width ratio*length."""
import numpy as np
import pytest

import oracle
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import mappings


def rect(x, y, length, width, angle):
    ratio = width / length
    return [x, y, length * (1 + ratio) / 2, ratio, angle]


L, W_ = 8.0, 4.0
A = L * W_
CASES = [
    # (r1, r2, expected intersection area)
    ("identical", rect(50, 50, L, W_, 0.3), rect(50, 50, L, W_, 0.3), A),
    ("disjoint", rect(50, 50, L, W_, 0.3), rect(50, 70, L, W_, 1.1), 0.0),
    # corners = R(angle+pi/2)(+-l/2, +-w/2) + centre with x = row: at angle pi/2 the long side runs along the rows
    ("half shift along the width", rect(50, 50, L, W_, np.pi / 2), rect(50, 52, L, W_, np.pi / 2), A / 2),
    ("half shift along the length", rect(50, 50, L, W_, np.pi / 2), rect(54, 50, L, W_, np.pi / 2), A / 2),
    ("three quarters", rect(50, 50, L, W_, np.pi / 2), rect(52, 50, L, W_, np.pi / 2), 0.75 * A),
    ("crossed at 90 degrees", rect(50, 50, L, W_, 0.0), rect(50, 50, L, W_, np.pi / 2), W_ * W_),
    ("contained", rect(50, 50, L, W_, 0.7), rect(50, 50, L / 2, W_ / 2, 0.7), A / 4),
    ("shared edge", rect(50, 50, L, W_, np.pi / 2), rect(50, 54, L, W_, np.pi / 2), 0.0),
    ("zero width", rect(50, 50, L, W_, 0.2), [50, 50, 4.0, 0.0, 0.2], 0.0),
    ("45 degree square in square", rect(50, 50, 4.0, 4.0, 0.0), rect(50, 50, 4.0, 4.0, np.pi / 4),
     2 * 16.0 * (np.sqrt(2) - 1)),            # regular octagon
]


def expected_energy(r1, r2, inter):
    a1 = (2 * r1[2] / (1 + r1[3])) ** 2 * r1[3]
    a2 = (2 * r2[2] / (1 + r2[3])) ** 2 * r2[3]
    return inter / (min(a1, a2) + 1e-6)


@pytest.mark.parametrize("name,r1,r2,inter", CASES, ids=[c[0] for c in CASES])
def test_oracle_overlap(name, r1, r2, inter):
    e = expected_energy(r1, r2, inter)
    assert oracle.overlap(r1, r2) == pytest.approx(e, abs=1e-12)
    assert oracle.overlap(r2, r1) == oracle.overlap(r1, r2)           # a function of the unordered pair


def overlap_model():
    unit = [E.UnitTerm("U", E.U_CONST, [0.0])]
    pair = [E.PairTerm("O", E.P_OVERLAP, max_dist=32.0, reduce=E.REDUCE_MAX)]
    return E.build_model_desc(unit, pair, None)


@pytest.mark.gpu
@pytest.mark.parametrize("name,r1,r2,inter", CASES, ids=[c[0] for c in CASES])
def test_gpu_overlap(name, r1, r2, inter):
    from mpp_cnn_rs_object_detection_amd import hip_api, kernels
    ctx = hip_api.MppContext(0, point_capacity=16)
    ctx.set_maps(np.zeros((128, 128), np.float32), [np.zeros((128, 128, 32), np.float32)] * 3)
    ctx.set_model(overlap_model(), mappings.default_mappings())
    pts = np.array([r1, r2], dtype=float)
    ctx.set_points(0, pts[:, :2].astype(np.int32), pts[:, 2:])
    e = expected_energy(r1, r2, inter)
    assert ctx.total_energy() == pytest.approx(2 * e, abs=1e-11)     # each endpoint carries the max over its edges
    # and through the chain kernel: birth of r2 next to r1 (always accepted)
    ctx.set_points(0, pts[:1, :2].astype(np.int32), pts[:1, 2:])
    ctx.set_kernels(kernels.make_kernels(mappings.default_mappings(), 1.0))
    ctx.set_schedule(1e12, 1.0, 0.0)
    tape = np.zeros(1, hip_api.PROPOSAL_DTYPE)
    tape[0]["kernel"], tape[0]["target"], tape[0]["ax"], tape[0]["ay"] = 0, -1, int(r2[0]), int(r2[1])
    tape[0]["as"], tape[0]["ar"], tape[0]["aa"], tape[0]["u_accept"] = r2[2], r2[3], r2[4], 1e-300
    out = ctx.replay(0, tape)
    assert out["dE"][0] == pytest.approx(2 * e, abs=1e-11) and out["accepted"][0] == 1


def test_rectangle_area_is_length_times_width():
    # shoelace of rect_to_poly == length*width
    from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = Rectangle(int(rng.integers(0, 100)), int(rng.integers(0, 100)), size=float(rng.uniform(1, 30)),
                      ratio=float(rng.uniform(0.05, 1)), angle=float(rng.uniform(0, np.pi)))
        p = r.poly_coord
        x, y = p[:, 0], p[:, 1]
        area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
        assert area == pytest.approx(r.length * r.width, rel=1e-12)
        assert oracle.overlap(r.as_row(), r.as_row()) == pytest.approx(area / (area + 1e-6), abs=1e-12)
