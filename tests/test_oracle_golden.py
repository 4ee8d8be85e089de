"""The C oracle against vectors recorded from the reference itself (tests/golden/*.npz)."""
import numpy as np
import pytest

import oracle
from helpers import GOLDEN, Tape, model_for, sorted_rows
from mpp_cnn_rs_object_detection_amd import energies as E
from mpp_cnn_rs_object_detection_amd import synth

TAPES = ["tape_hrc_64.npz", "tape_log_96.npz", "tape_hrc_128_gt.npz", "tape_log_64_empty.npz",
         "tape_hrc_96_sm.npz", "tape_log_64_sm.npz",      # *_sm: recorded with use_split_merge=True
         "tape_hrc_256.npz", "tape_log_256.npz", "tape_hrc_256_warm.npz",     # BASELINE config 1 (256x256 tile)
         "tape_contrast_96.npz"]       # the contrast energy setup (energy_setup_contrast.py), picture handed over as float64
# energies are float64 sums of float32 map reads; the reference does part of the arithmetic in float32
DE_ATOL, DE_RTOL, P_RTOL = 2e-6, 2e-6, 2e-5
TIE = 1e-5   # |log u - log alpha| below which an accept decision may legitimately differ


def make_oracle(t: Tape):
    o = oracle.Oracle(t.shape, t.det, t.marks, t.model, t.kernels)
    if t.image is not None:
        o.set_image(t.image)
    o.set_points(t.init_xy, t.init_marks)
    return o


@pytest.mark.parametrize("name", TAPES)
def test_tape_replay(name):
    t = Tape(name)
    assert np.allclose(t.kernels.p_kernel[:len(t.p_kernels)], t.p_kernels, rtol=0, atol=1e-15)
    assert np.all(t.kernels.p_kernel[len(t.p_kernels):] == 0)
    o = make_oracle(t)
    assert o.total_energy() == pytest.approx(t.E0, rel=1e-6, abs=1e-6)
    p = t.params
    o.set_temperature(p["init_temperature"], p["alpha_t"], p["target_temperature"])
    out = o.replay(t.proposals)
    ref_acc = t.col("accepted").astype(int)
    margin = np.abs(np.log(t.col("u_accept") + 1e-16) - out["log_alpha"])
    disagree = np.nonzero(out["accepted"] != ref_acc)[0]
    assert all(margin[i] < TIE for i in disagree), f"accept decisions differ at {disagree[:5]}"
    assert len(disagree) == 0, "a near-tie flipped; regenerate the tape or raise the tolerance consciously"
    np.testing.assert_allclose(out["T"], t.col("T"), rtol=1e-12)
    np.testing.assert_array_equal(out["n_after"], t.col("n_after").astype(int))
    np.testing.assert_allclose(out["dE"], t.col("dE"), rtol=DE_RTOL, atol=DE_ATOL)
    np.testing.assert_allclose(out["fwd"], t.col("fwd"), rtol=P_RTOL, atol=1e-300)
    np.testing.assert_allclose(out["bwd"], t.col("bwd"), rtol=P_RTOL, atol=1e-300)
    xy, marks = o.get_points()
    got = np.concatenate([xy.astype(float), marks], axis=1)
    np.testing.assert_allclose(sorted_rows(got), sorted_rows(t.final_by_slots), rtol=0, atol=1e-12)
    if p["samples_interval"] == 1:   # the returned sample is then the last state of the chain
        np.testing.assert_array_equal(sorted_rows(got), sorted_rows(t.final))


@pytest.mark.parametrize("name", ["tape_hrc_64.npz", "tape_log_96.npz"])
def test_naive_detection_matches_reference_init(name):
    t = Tape(name)
    o = make_oracle(t)
    xy, marks = o.naive_detection(t.setup.detection_threshold, 6.0)
    got = np.concatenate([xy.astype(float), marks], axis=1)
    np.testing.assert_allclose(sorted_rows(got), sorted_rows(t.init), rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag,setup_name", [("hrc", "legacy"), ("log", "no-calibration")])
def test_delta_cases(tag, setup_name):
    z = np.load(f"{GOLDEN}/delta_cases.npz", allow_pickle=False)
    g = lambda k: z[f"{tag}_{k}"]
    det = g("det")
    _, marks = synth.render_maps(det.shape, g("gt_xy"), g("gt_marks"), noise=float(z["noise"]),
                                 noise_seed=int(z["noise_seed"]))
    setup, comb, model = model_for(setup_name)
    base = g("base")
    o = oracle.Oracle(det.shape, det, marks, model)
    o.set_points(base[:, :2].astype(np.int32), base[:, 2:])
    e0, vec = o.total_energy(return_vectors=True)
    assert e0 == pytest.approx(float(g("E0")), rel=1e-6, abs=1e-6)
    ref_names = [str(n) for n in g("names")]
    order = [o.names.index(n) for n in ref_names]
    np.testing.assert_allclose(vec[:, order], g("vec"), rtol=2e-6, atol=2e-6)
    # plain sum (no combinator): energy_graph.py:130-131
    unit, pair = setup.make_energies()
    o_sum = oracle.Oracle(det.shape, det, marks, E.build_model_desc(unit, pair, None))
    o_sum.set_points(base[:, :2].astype(np.int32), base[:, 2:])
    assert o_sum.total_energy() == pytest.approx(float(g("E0_sum")), rel=1e-6, abs=1e-5)
    np.testing.assert_allclose(o.papangelou(), g("papangelou_dE"), rtol=2e-6, atol=2e-6)

    rows = [tuple(r) for r in base]
    add_off = np.concatenate([[0], np.cumsum(g("add_len"))])
    rem_off = np.concatenate([[0], np.cumsum(g("rem_len"))])
    for i, (d_ref, e1_ref) in enumerate(zip(g("dE"), g("E1"))):
        add = g("add_flat")[add_off[i]:add_off[i + 1]]
        rem = g("rem_flat")[rem_off[i]:rem_off[i + 1]]
        slots = [rows.index(tuple(r)) for r in rem]
        d = o.delta(removal_slots=slots, add_xy=add[:, :2].astype(np.int32), add_marks=add[:, 2:])
        assert d == pytest.approx(float(d_ref), rel=2e-6, abs=5e-6), f"case {i}"
        keep = [r for j, r in enumerate(rows) if j not in slots] + [tuple(r) for r in add]
        new = np.array(keep, dtype=float).reshape(-1, 5)
        o2 = oracle.Oracle(det.shape, det, marks, model)
        o2.set_points(new[:, :2].astype(np.int32), new[:, 2:])
        e1 = o2.total_energy()
        assert e1 == pytest.approx(float(e1_ref), rel=1e-6, abs=5e-6)
        assert abs(d - (e1 - e0)) < 1e-8    # the reference's own criterion, test_perturbation_sampler.py:99


def test_follow_and_forced_replay_reproduce_a_native_chain():
    """The test-side resync tools of the oracle: following its own tape gives the same records and draws the same
    proposals; a forced replay with the recorded decisions lands on the same state; a forced replay with the opposite
    decision for one step does not."""
    import oracle
    from helpers import model_for
    from mpp_cnn_rs_object_detection_amd import kernels, mappings, synth
    t = synth.make_tile(96, 18, tile_id=21, noise=0.1)
    setup, comb, model = model_for("legacy")
    maps = mappings.default_mappings()
    o = oracle.Oracle(t.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    kd = kernels.make_kernels(maps, float(max(1, len(xy))))
    o = oracle.Oracle(t.shape, t.det, t.marks, model, kd)
    o.set_points(xy, mk)
    o.set_temperature(1.0, 0.998, 0.0)
    start = o.save()
    out, props = o.run(1500, 7, chain=3, trace=True)
    end_xy, end_m = o.get_points()
    o.restore(start, 0.998)
    out2, native = o.follow(props, 7, chain=3)
    assert np.array_equal(out2, out) and np.array_equal(native, props) and o.step_index() == 1500
    o.restore(start, 0.998)
    o.replay_forced(props, out["accepted"])
    xy2, m2 = o.get_points()
    assert np.array_equal(xy2, end_xy) and np.array_equal(m2, end_m)
    k = int(np.nonzero((out["accepted"] > 0) & (props["kernel"] == 4))[0][0])       # an accepted translation
    flipped = out["accepted"].copy()
    flipped[k] = 0
    o.restore(start, 0.998)
    o.replay_forced(props[:k + 1], flipped[:k + 1])
    xy3, _ = o.get_points()
    o.restore(start, 0.998)
    o.replay_forced(props[:k + 1], out["accepted"][:k + 1])
    xy4, _ = o.get_points()
    assert not np.array_equal(xy3, xy4)
