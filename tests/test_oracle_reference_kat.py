"""The reference's own known answers, run against the C oracle.

Scenarios and expected numbers are those of the reference's tests
(``test/test_energy_graph.py:94-244``, ``test/test_interacting_points_set.py:149-272``):
a constant unit energy and an indicator pair energy reduced with max.
"""
import numpy as np
import pytest

import oracle
from mpp_cnn_rs_object_detection_amd import energies as E


def graph_model(unit_value, pair_kind, max_dist):
    unit = [E.UnitTerm("U", E.U_CONST, [unit_value])]
    pair = [E.PairTerm("P", pair_kind, max_dist=max_dist, reduce=E.REDUCE_MAX)]
    return E.build_model_desc(unit, pair, None)


def make(shape, unit_value, pair_kind, max_dist, pts=()):
    o = oracle.Oracle(shape, None, None, graph_model(unit_value, pair_kind, max_dist))
    o.set_points(np.array(pts, dtype=np.int32).reshape(-1, 2), np.zeros((len(pts), 3)))
    return o


Z3 = [[0.0, 0.0, 0.0]]


def test_total_energy_energy_graph():
    # test_energy_graph.py:94-130: -10 per point, +1 per endpoint of a pair at distance <= 1
    pts = []
    o = make((64, 64), -10.0, E.P_DIST_LE, 1.0, pts)
    assert o.total_energy() == 0.0
    for p, expect in (((10, 10), -10.0), ((10, 11), -18.0), ((20, 20), -28.0)):
        pts.append(p)
        o.set_points(pts, np.zeros((len(pts), 3)))
        assert o.total_energy() == expect
    pts.remove((10, 11))
    o.set_points(pts, np.zeros((len(pts), 3)))
    assert o.total_energy() == -20.0


def test_compute_delta_energy_graph():
    # test_energy_graph.py:177-244
    pts = []
    o = make((64, 64), -10.0, E.P_DIST_LE, 1.0, pts)

    def reset():
        o.set_points(pts, np.zeros((len(pts), 3)))

    assert o.delta(add_xy=[[10, 10]], add_marks=Z3) == -10.0
    pts.append((10, 10)); reset()
    assert o.delta(add_xy=[[10, 11]], add_marks=Z3) == -10.0 + 2 * 1.0
    pts.append((10, 11)); reset()
    assert o.delta(add_xy=[[20, 20]], add_marks=Z3) == -10.0
    pts.append((20, 20)); reset()
    # move p3 (20,20) -> (10,12): p2 already interacts with p1, only the moved point gains an interaction
    assert o.delta(removal_slots=[2], add_xy=[[10, 12]], add_marks=Z3) == 1.0
    pts[2] = (10, 12); reset()
    assert o.delta(add_xy=[[5, 5]], add_marks=Z3) == -10.0
    pts.append((5, 5)); reset()
    assert o.delta(add_xy=[[5, 6]], add_marks=Z3) == -10.0 + 2 * 1.0
    pts.append((5, 7)); reset()
    assert o.delta(removal_slots=[4], add_xy=[[5, 8]], add_marks=Z3) == 0.0
    pts[4] = (5, 8); reset()
    assert o.delta(removal_slots=[1]) == +10.0 - 3 * 1.0


def test_total_energy_interacting_points_set():
    # test_interacting_points_set.py:149-208: unit 1, pair "distance < 3" (edges exist for d <= 3), max-reduced
    o = make((10, 10), 1.0, E.P_DIST_LT, 3.0, [(0, 0), (0, 1), (0, 4)])
    assert o.total_energy() == 5.0
    o.set_points([(0, 0), (0, 1), (0, 4), (0, 5)], np.zeros((4, 3)))
    assert o.total_energy() == 8.0
    o.set_points([(0, 0), (0, 1), (1, 0)], np.zeros((3, 3)))
    assert o.total_energy() == 6.0


def test_energy_delta_interacting_points_set():
    # test_interacting_points_set.py:211-272
    pts = [(0, 0), (0, 1), (0, 5)]
    o = make((10, 10), 1.0, E.P_DIST_LT, 3.0, pts)
    e0 = o.total_energy()
    assert e0 == 5.0
    d = o.delta(removal_slots=[2])
    assert d == -1.0
    o.set_points(pts[:2], np.zeros((2, 3)))
    assert o.total_energy() == 4.0 == e0 + d
    o.set_points(pts, np.zeros((3, 3)))
    d = o.delta(removal_slots=[2], add_xy=[[1, 0]], add_marks=Z3)
    assert d == 1.0
    o.set_points([(0, 0), (0, 1), (1, 0)], np.zeros((3, 3)))
    assert o.total_energy() == 6.0 == e0 + d


def test_out_of_bounds_point_is_rejected():
    # point_set.py:99 asserts the cell index is inside the grid
    o = make((10, 10), 1.0, E.P_DIST_LT, 3.0)
    with pytest.raises(AssertionError):
        o.set_points([(10, 3)], np.zeros((1, 3)))


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, out in kat:
        assert [int(v) for v in oracle.philox(ctr, key)] == out
