"""Deep rounds (csrc/mpp_deep.hip): every lane evaluates one speculative step, up to 64 x waves steps per round.

The chain must be the sequential one bit for bit -- traces and final configuration equal to the one-wave kernel's --
for every round depth, and equal the CPU oracle's at the same seed."""
import numpy as np
import pytest

from test_gpu_chain import compare_traces, setup_case

pytestmark = pytest.mark.gpu


def deep_case(tile, n_obj, setup_name, spec, deep, fixed=0, **kw):
    t, o, ctx = setup_case(tile, n_obj, setup_name, spec=spec, **kw)
    ctx.set_option("deep", deep)
    ctx.set_option("deep_fixed", fixed)
    return t, o, ctx


@pytest.mark.parametrize("spec,deep,fixed", [(1, 64, 0), (1, 64, 64), (1, 8, 8), (2, 128, 0), (4, 256, 32), (8, 256, 0),
                                             (8, 256, 256), (8, 64, 8), (8, 128, 96)])
@pytest.mark.parametrize("setup_name", ["legacy", "no-calibration"])
def test_deep_rounds_reproduce_the_sequential_chain(spec, deep, fixed, setup_name):
    n_steps, seed = 8000, 7
    t, o, c1 = setup_case(128, 40, setup_name, spec=1)
    _, _, cd = deep_case(128, 40, setup_name, spec, deep, fixed)
    for c in (c1, cd):
        c.set_schedule(1.0, 0.9985, 0.0)
    out1, props1 = c1.run(n_steps, seed, trace_tile=0)
    outd, propsd = cd.run(n_steps, seed, trace_tile=0)
    st = cd.deep_stats()
    assert st["committed"] == n_steps and st["rounds"] > 0
    for f in out1.dtype.names:
        np.testing.assert_array_equal(out1[f], outd[f], err_msg=f)
    assert props1.tobytes() == propsd.tobytes()
    xy1, m1 = c1.get_points()
    xyd, md = cd.get_points()
    assert xy1.tobytes() == xyd.tobytes() and m1.tobytes() == md.tobytes()
    assert cd.step_index() == n_steps
    # untraced: the production instantiation
    _, _, cu = deep_case(128, 40, setup_name, spec, deep, fixed)
    cu.set_schedule(1.0, 0.9985, 0.0)
    cu.run(n_steps, seed)
    xyu, mu = cu.get_points()
    assert xy1.tobytes() == xyu.tobytes() and m1.tobytes() == mu.tobytes()


def test_deep_chain_matches_oracle():
    n_steps, seed = 12000, 99
    t, o, ctx = deep_case(160, 60, "legacy", 8, 256, tile_id=3)
    o.set_temperature(1.0, 0.999, 0.0)
    ctx.set_schedule(1.0, 0.999, 0.0)
    oout, oprops = o.run(n_steps, seed, chain=0, trace=True)
    gout, gprops = ctx.run(n_steps, seed, chain0=0, trace_tile=0)
    compare_traces(gout, gprops, oout, oprops)
    gxy, gm = ctx.get_points()
    oxy, om = o.get_points()
    np.testing.assert_array_equal(gxy, oxy)
    np.testing.assert_allclose(gm, om, rtol=1e-9, atol=1e-9)
    assert ctx.total_energy() == pytest.approx(o.total_energy(), rel=1e-10, abs=1e-9)
    # continued in a second call: the ring of temperatures and the step counter carry over
    o.run(3000, seed, chain=0)
    ctx.run(3000, seed, chain0=0)
    gxy, gm = ctx.get_points()
    oxy, om = o.get_points()
    np.testing.assert_array_equal(gxy, oxy)
    np.testing.assert_allclose(gm, om, rtol=1e-9, atol=1e-9)


def test_hot_start_hands_the_chain_over_without_changing_it():
    """A chain of 8 waves starts with one wave per step and is handed to the deep rounds once ~6 of 8 steps commit per
    round (``handover``, csrc/mpp_sampler.hip: ERR_HANDOVER; the launch ends after a round's commits like a capacity
    stop and ``mpp_run`` continues with the next step): the final configuration, the step counter and the temperature
    are those of the chain run in deep rounds alone and with one wave per step alone; the deep rounds did the rest."""
    n_steps, seed = 30000, 11
    finals = []
    for handover, deep in ((1, 128), (0, 128), (0, 0)):
        t, o, ctx = setup_case(256, 80, "legacy", spec=8)
        ctx.set_option("deep", deep)
        ctx.set_option("handover", handover)
        ctx.set_schedule(1.0, 0.999, 0.0)
        ctx.run(n_steps, seed)
        st = ctx.deep_stats()
        if handover:
            assert 0 < st["committed"] < n_steps, st          # both kernels took part
        elif deep:
            assert st["committed"] == n_steps
        assert ctx.step_index() == n_steps
        finals.append(ctx.get_points())
        ctx.close()
    for xy, m in finals[1:]:
        assert xy.tobytes() == finals[0][0].tobytes() and m.tobytes() == finals[0][1].tobytes()
