"""Capacity overflow -> grow and continue (the reference's point set has no capacity, point_set.py:45-188).

A chain stops BEFORE the step that would exceed ``point_capacity`` / ``cell_capacity``; ``mpp_run`` doubles the capacity
and issues the same launch again, finished tiles return at once, the stopped one continues with the very next step.
The chain must be the one an ample capacity produces, step for step (also on a traced tile, whose record continues
across the re-launch), and equal the CPU oracle's."""
import numpy as np
import pytest

import oracle
from helpers import lockstep_vs_oracle, model_for, soak_case
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth

pytestmark = pytest.mark.gpu


def make(tile, model, kd, xy, mk, T0, alpha, spec=8, **caps):
    ctx = hip_api.MppContext(0, spec_waves=spec, **caps)
    ctx.set_maps(tile.det, tile.marks)
    ctx.set_model(model, mappings.default_mappings())
    ctx.set_kernels(kd)
    ctx.set_points(0, xy, mk)
    ctx.set_schedule(T0, alpha, 0.0)
    return ctx


@pytest.mark.parametrize("spec", [1, 8])
def test_tiny_capacities_grow_and_give_the_same_chain(spec):
    tile = synth.make_tile(96, 14, tile_id=71, noise=0.1)
    setup, comb, model = model_for("legacy")
    maps = mappings.default_mappings()
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, model, kernels.make_kernels(maps, 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    # hot, and with a birth intensity of 150 (the equilibrium population of a hot chain is ~intensity * exp(-dE / T)):
    # uniform births pile up before the chain cools
    kd = kernels.make_kernels(maps, 150.0)
    T0, alpha, steps, seed = 5.0, 0.9995, 12000, 31
    big = make(tile, model, kd, xy, mk, T0, alpha, spec, point_capacity=1024, cell_capacity=32)
    bout, bprops = big.run(steps, seed, chain0=3, trace_tile=0)
    assert big.get_option("grow_events") == 0 and int(bout["n_after"].max()) > 40
    small = make(tile, model, kd, xy, mk, T0, alpha, spec, point_capacity=16, cell_capacity=2)
    sout, sprops = small.run(steps, seed, chain0=3, trace_tile=0)
    assert small.get_option("grow_events") >= 3                           # 16 -> 32 -> 64 slots, 2 -> 4 -> ... per cell
    assert small.get_option("point_capacity") >= 64 and small.get_option("cell_capacity") >= 4
    for f in sout.dtype.names:
        np.testing.assert_array_equal(sout[f], bout[f], err_msg=f)        # the traced record continues across re-launches
    for f in sprops.dtype.names:
        np.testing.assert_array_equal(sprops[f], bprops[f], err_msg=f)
    sxy, sm = small.get_points()
    bxy, bm = big.get_points()
    np.testing.assert_array_equal(sxy, bxy)
    np.testing.assert_array_equal(sm, bm)
    assert small.step_index() == steps
    # untraced production kernel, several tiles at once: only the tile that overflows is continued
    multi = hip_api.MppContext(0, spec_waves=spec, point_capacity=16, cell_capacity=2)
    tiles = [tile, synth.make_tile(96, 3, tile_id=72, noise=0.1)]
    multi.set_maps(np.stack([t.det for t in tiles]), [np.stack([t.marks[k] for t in tiles]) for k in range(3)])
    multi.set_model(model, maps)
    multi.set_kernels(kd, intensity=np.array([kd.intensity, kd.intensity]))
    multi.set_points(0, xy, mk)
    multi.set_points(1, xy[:2], mk[:2])
    multi.set_schedule(T0, alpha, 0.0)
    multi.run(steps, seed, chain0=3)
    mxy, mm = multi.get_points(0)
    np.testing.assert_array_equal(mxy, bxy)
    np.testing.assert_array_equal(mm, bm)
    assert multi.step_index(0) == steps and multi.step_index(1) == steps
    # and the oracle agrees, step by step, with the growing context
    small2 = make(tile, model, kd, xy, mk, T0, alpha, spec, point_capacity=16, cell_capacity=2)
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, model, kd)
    o.set_points(xy, mk)
    o.set_temperature(T0, alpha, 0.0)
    lockstep_vs_oracle(small2, o, steps, seed, 3, alpha, chunk=5000)
    assert small2.get_option("grow_events") >= 3


def test_without_auto_grow_the_chain_stops_before_the_step_and_can_be_continued_by_hand():
    tile = synth.make_tile(96, 14, tile_id=71, noise=0.1)
    setup, comb, model = model_for("legacy")
    maps = mappings.default_mappings()
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, model, kernels.make_kernels(maps, 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    kd = kernels.make_kernels(maps, 150.0)
    ref = make(tile, model, kd, xy, mk, 5.0, 0.9995, 8, point_capacity=1024)
    ref.run(6000, 31, chain0=3)
    ctx = make(tile, model, kd, xy, mk, 5.0, 0.9995, 8, point_capacity=1024, cell_capacity=3)
    ctx.set_option("auto_grow", 0)
    with pytest.raises(hip_api.MppError) as e:
        ctx.run(6000, 31, chain0=3)
    assert e.value.code == -11 and "cell" in str(e.value)
    stopped = ctx.step_index()
    assert 0 < stopped < 6000
    # the state of that moment was written back: raise the limit by hand, clear the sticky error by re-uploading the
    # points, restore the schedule position and run the rest
    pxy, pm = ctx.get_points()
    ctx.set_option("cell_capacity", 32)
    ctx.set_points(0, pxy, pm)
    ctx.run(6000 - stopped, 31, chain0=3)
    a, b = ctx.get_points(), ref.get_points()
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("k", [1, 17, 21, 32, 66, 84, 100])
def test_hot_soak_cases_complete(k):
    """T0 = 5 cases of the chain soak (profiles/tools/soak.py): 25 of 400 such cases stopped with 'cell overflowed' in
    round 1 (profiles/r01_soak.txt).  They now run to the end and equal the oracle.  (Chains with the split / merge
    kernels grow their cells the same way -- their two-point steps check the room in the target cells before they change
    anything -- but a hot, crowded one can still stop at the merge kernel's 32-candidate neighbour list.)"""
    c = soak_case(k)
    t = c["tile"]
    o = oracle.Oracle(t.shape, t.det, t.marks, c["model"], c["kd"])
    o.set_points(c["xy"], c["marks"])
    o.set_temperature(c["T0"], c["alpha"], 0.0)
    ctx = make(t, c["model"], c["kd"], c["xy"], c["marks"], c["T0"], c["alpha"], 8, point_capacity=256, cell_capacity=8)
    lockstep_vs_oracle(ctx, o, c["steps"], c["seed"], c["chain"], c["alpha"], chunk=5000)
    gxy, gm = ctx.get_points()
    oxy, om = o.get_points()
    np.testing.assert_array_equal(gxy, oxy)
    np.testing.assert_allclose(gm, om, rtol=1e-9, atol=1e-9)
