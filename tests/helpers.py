"""Shared helpers for the test-suite: fixture loading and tape -> proposal conversion."""
from __future__ import annotations

import json
import os

import numpy as np

from mpp_cnn_rs_object_detection_amd import energies, kernels, mappings, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROPOSAL_DTYPE = np.dtype([("kernel", "<i4"), ("target", "<i4"), ("ax", "<i4"), ("ay", "<i4"),
                           ("as", "<f8"), ("ar", "<f8"), ("aa", "<f8"), ("aux0", "<f8"), ("aux1", "<f8"),
                           ("param_id", "<i4"), ("new_class", "<i4"), ("u_accept", "<f8")], align=True)


def load_json(*parts):
    with open(os.path.join(REPO, *parts)) as f:
        return json.load(f)


def hrc_model():
    setup = energies.LegacyEnergySetup()
    setup.load_calibration(os.path.join(REPO, "models_storage", "mpp", "mpp_hrcM"))
    comb = energies.hierarchical_from_manual(load_json("model_configs", "mpp", "mpp_hrcM.json")["manual"])
    return setup, comb


def log_model():
    setup = energies.NoCalibrationEnergySetup(ratio_prior=True)
    setup.load_calibration(os.path.join(REPO, "models_storage", "mpp", "mpp_log"))
    j = load_json("models_storage", "mpp", "mpp_log", "energy_combination_model.json")
    comb = energies.LogisticEnergyCombinator(weights=np.array(j["weights"], dtype=np.float32), bias=j["bias"],
                                             energy_names=j["energy_names"])
    return setup, comb


def model_for(setup_name: str):
    setup, comb = hrc_model() if setup_name == "legacy" else log_model()
    unit, pair = setup.make_energies()
    return setup, comb, energies.build_model_desc(unit, pair, comb)


def contrast_model(z):
    """The contrast energy setup a tape was recorded with (tape_contrast_*.npz: picture, measure, calibration, weights)."""
    from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
    cal, weights = json.loads(str(z["calibration"])), json.loads(str(z["weights"]))
    setup = energies.ContrastMeasureEnergySetup(contrast_type=str(z["contrast_type"]), manual_threshold=cal["detection_thresh"])
    setup.energy_cal = cal
    shape = tuple(int(v) for v in z["shape"])
    data = ImageWMaps(name="0", shape=shape, image=z["image"], detection_map=None, param_dist_maps=None,
                      mappings=mappings.default_mappings(), param_names=["size", "ratio", "angle"], labels=None, gt_config=[])
    unit, pair = setup.make_energies(data)
    comb = energies.ManualHierarchicalEnergyCombinator(weights, "ContrastEnergy", 0.0)
    return setup, comb, energies.build_model_desc(unit, pair, comb), energies.classic_image(unit)


class Tape:
    """A recorded reference chain (tests/golden/tape_*.npz) turned into replayable proposals."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.shape = tuple(int(v) for v in z["shape"])
        self.det = z["det"]
        _, self.marks = synth.render_maps(self.shape, z["gt_xy"], z["gt_marks"], noise=float(z["noise"]),
                                          noise_seed=int(z["noise_seed"]))
        self.setup_name = str(z["setup"])
        self.params = json.loads(str(z["params"]))
        self.cols = {c: i for i, c in enumerate(z["columns"])}
        self.raw = z["tape"]
        self.init = z["init"]
        self.final = z["final"]
        self.p_kernels = z["p_kernels"]
        self.intensity = float(z["intensity"])
        self.E0 = float(z["E0"])
        self.image = None                  # the prepared picture of a classic image energy (contrast setup)
        if self.setup_name == "contrast":
            self.setup, self.comb, self.model, self.image = contrast_model(z)
        else:
            self.setup, self.comb, self.model = model_for(self.setup_name)
        self.kernels = kernels.make_kernels(mappings.default_mappings(), self.intensity,
                                            use_split_merge=len(self.p_kernels) == 10)
        self.proposals = self._proposals()

    def col(self, name):
        return self.raw[:, self.cols[name]]

    @property
    def init_xy(self):
        return self.init[:, :2].astype(np.int32)

    @property
    def init_marks(self):
        return np.ascontiguousarray(self.init[:, 2:5], dtype=np.float64)

    def _proposals(self):
        """Translate point identities into slots under the canonical slot discipline
        (birth appends, death swap-removes, move rewrites in place), following the REFERENCE's
        accept decisions."""
        state = [tuple(r) for r in self.init]
        out = np.zeros(len(self.raw), PROPOSAL_DTYPE)
        c = self.cols
        for i, row in enumerate(self.raw):
            rem = tuple(row[c["rx"]:c["rx"] + 5])
            add = tuple(row[c["ax"]:c["ax"] + 5])
            has_rem, has_add = not np.isnan(rem[0]), not np.isnan(add[0])
            p = out[i]
            p["kernel"] = int(row[c["kernel"]])
            p["target"] = state.index(rem) if has_rem else -1
            if p["kernel"] in (8, 9):              # split / merge (tapes recorded with use_split_merge)
                rem2 = tuple(row[c["r2x"]:c["r2x"] + 5])
                add2 = tuple(row[c["a2x"]:c["a2x"] + 5])
                p["param_id"], p["new_class"], p["u_accept"] = -1, -1, row[c["u_accept"]]
                if p["kernel"] == 8 and has_rem:   # target, position delta, mark deltas
                    p["aux0"], p["aux1"] = row[c["delta0"]], row[c["delta1"]]
                    p["as"], p["ar"], p["aa"] = row[c["sd0"]], row[c["sd1"]], row[c["sd2"]]
                elif p["kernel"] == 9 and has_rem:
                    p["param_id"] = state.index(rem2)
                if row[c["accepted"]] > 0 and has_rem:
                    state[p["target"]] = add
                    if p["kernel"] == 8:
                        state.append(add2)
                    else:
                        j = p["param_id"]
                        state[j] = state[-1]
                        state.pop()
                continue
            if has_add:
                p["ax"], p["ay"], p["as"], p["ar"], p["aa"] = int(add[0]), int(add[1]), add[2], add[3], add[4]
            p["aux0"] = 0.0 if np.isnan(row[c["delta0"]]) else row[c["delta0"]]
            p["aux1"] = 0.0 if np.isnan(row[c["delta1"]]) else row[c["delta1"]]
            p["param_id"] = int(row[c["param_id"]])
            p["new_class"] = int(row[c["new_class"]])
            p["u_accept"] = row[c["u_accept"]]
            if row[c["accepted"]] > 0:
                if has_rem and has_add:
                    state[p["target"]] = add
                elif has_rem:
                    t = p["target"]
                    state[t] = state[-1]
                    state.pop()
                elif has_add:
                    state.append(add)
        self.final_by_slots = np.array(state, dtype=float).reshape(-1, 5)
        return out


def sorted_rows(a):
    a = np.asarray(a, dtype=float).reshape(-1, 5)
    return a[np.lexsort(a.T[::-1])] if len(a) else a


def soak_case(k: int):
    """Case k of the chain soak (profiles/tools/soak.py and the regression tests): a random tile, density, crowding,
    temperature, energy setup and kernel mixture, each case from its own generator so that it can be replayed alone.
    -> dict(tile, setup, model, kd (kernels), xy, marks, T0, alpha, steps, seed, chain, text)"""
    import oracle
    rng = np.random.default_rng([2026, k])
    size = int(rng.choice([64, 96, 128, 160, 200, 256]))
    n_obj = int(rng.integers(5, max(6, (size // 14) ** 2 // 2)))
    setup_name = str(rng.choice(["legacy", "no-calibration"]))
    sm = bool(rng.random() < 0.3)
    T0, alpha = float(rng.choice([0.3, 1.0, 2.0, 5.0])), float(rng.choice([0.999, 0.9995, 0.9999]))
    steps, seed, chain = int(rng.integers(3000, 20000)), int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 1000))
    crowd = bool(rng.random() < 0.3)
    t = synth.make_tile(size, n_obj, tile_id=5000 + k, noise=float(rng.choice([0.0, 0.1, 0.3])))
    setup, comb, model = model_for(setup_name)
    maps = mappings.default_mappings()
    o = oracle.Oracle(t.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    if crowd and len(xy):
        xy = np.concatenate([xy, np.clip(xy + rng.integers(-4, 5, size=xy.shape), 0, size - 1)]).astype(np.int32)
        mk = np.concatenate([mk, mk])
    kd = kernels.make_kernels(maps, float(max(1, len(xy))), use_split_merge=sm)
    text = (f"{size}px {setup_name} split/merge={int(sm)} crowd={int(crowd)} T0={T0} alpha={alpha} steps={steps} "
            f"seed={seed} chain={chain} n0={len(xy)}")
    return dict(tile=t, setup=setup, model=model, kd=kd, xy=xy, marks=mk, T0=T0, alpha=alpha, steps=steps, seed=seed,
                chain=chain, text=text)


DE_TOL = 1e-9       # |dE_gpu - dE_oracle| allowed (absolute and relative), DESIGN.md section 2
# ... except for proposals whose rectangle is smaller than TINY_AREA px^2 (a uniform birth with size ~0.01): the overlap
# term divides a shoelace area, formed from absolute pixel coordinates of ~500 (rounding ~1e-10 px^2), by that
# rectangle's area + 1e-6, so a last-place difference between the device's and libm's sincos shows up at ~1e-8.
TINY_AREA, DE_TOL_TINY = 1e-2, 1e-5


def de_tolerance(props: np.ndarray, dE: np.ndarray) -> np.ndarray:
    """per-step tolerance on dE (see DE_TOL / TINY_AREA); props: the tape records of the steps"""
    length = 2.0 * props["as"] / (1.0 + props["ar"])
    area = length * length * props["ar"]
    has_add = ~np.isin(props["kernel"], (1, 3))
    tiny = has_add & (area < TINY_AREA)
    return np.where(tiny, DE_TOL_TINY, DE_TOL) * np.maximum(1.0, np.abs(dE))


def lockstep_vs_oracle(ctx, o, total_steps: int, seed: int, chain: int, alpha: float, T_target: float = 0.0,
                       chunk: int = 20000, tile: int = 0):
    """Run the GPU chain (traced tile ``tile`` of ``ctx``) and the CPU oracle side by side for ``total_steps`` steps and
    check them step by step.  Two questions are kept apart (``oracle.follow``):

    * proposals: at every step the oracle draws its own proposal from the same Philox counter and the same state;
      kernel, target, pixel, class and the uniform must be equal, Gaussian marks agree to 1e-9 (the device's log /
      sincos differ from libm's in the last place);
    * energies and decisions: the oracle then performs the KERNEL's proposal (so both states stay bit-identical --
      intersection areas of nearly collinear rectangle edges amplify a last-place difference of a mark to ~1e-8) and
      takes its own Metropolis decision: dE must agree to ``DE_TOL`` (``DE_TOL_TINY`` for proposals of a rectangle
      smaller than 0.01 px^2, see above), the decision and the population must be equal.

    The kernel sums only the terms a proposal changes, the oracle (like the reference, energy_graph.py:139-225) subtracts
    two sums over the whole neighbourhood, so dE agrees to ~1e-13, not bit for bit.  Once the chain is frozen (T below
    ~1e-13, which the 100 001-step schedule of BASELINE configs 2/3 reaches after 30 000 steps) a proposal whose true dE
    is 0 is decided by that rounding noise.  Such a step is a *tie within the stated tolerance*: it is let through only
    if dE agrees to ``DE_TOL`` AND the uniform of the accept test lies between the two log-acceptance values; the oracle
    is then put on the kernel's decision (forced replay) and the comparison goes on.  Anything else fails.
    Returns the number of ties."""
    ties, done = 0, 0
    exact = ("kernel", "target", "ax", "ay", "param_id", "new_class", "u_accept")
    while done < total_steps:
        n = min(chunk, total_steps - done)
        gout, gprops = ctx.run(n, seed=seed, chain0=chain - tile, trace_tile=tile)
        start = 0
        while start < n:
            saved = o.save()
            oout, native = o.follow(gprops[start:], seed, chain)
            g = gout[start:]
            tol = de_tolerance(gprops[start:], oout["dE"])
            with np.errstate(invalid="ignore"):       # (NaN on both sides -- a non-finite neighbourhood energy -- is agreement)
                bad = np.nonzero((g["accepted"] != oout["accepted"]) | (np.abs(g["dE"] - oout["dE"]) > tol) |
                                 (np.isnan(g["dE"]) != np.isnan(oout["dE"])))[0]
            k = int(bad[0]) if len(bad) else len(g)             # steps start .. start+k-1 agree; step start+k is the suspect
            upto = min(k + 1, len(g))
            for f in exact:
                assert np.array_equal(gprops[f][start:start + upto], native[f][:upto]), \
                    f"proposal field {f} differs in steps {done + start}..{done + start + upto}"
            for f in ("as", "ar", "aa", "aux0", "aux1"):
                np.testing.assert_allclose(gprops[f][start:start + upto], native[f][:upto], rtol=1e-9, atol=1e-9)
            assert np.array_equal(g["n_after"][:k], oout["n_after"][:k])
            np.testing.assert_allclose(g["fwd"][:k], oout["fwd"][:k], rtol=1e-9, atol=0)
            np.testing.assert_allclose(g["bwd"][:k], oout["bwd"][:k], rtol=1e-9, atol=0)
            if k == len(g):
                break
            s = start + k
            dg, dq = float(gout["dE"][s]), float(oout["dE"][k])
            assert abs(dg - dq) <= tol[k], f"step {done + s}: dE {dg!r} vs oracle {dq!r}"
            lu = np.log(float(gprops["u_accept"][s]) + 1e-16)
            la = sorted([float(gout["log_alpha"][s]), float(oout["log_alpha"][k])])
            assert la[0] - 1e-9 <= lu <= la[1] + 1e-9, \
                f"step {done + s}: accept {gout['accepted'][s]} vs {oout['accepted'][k]} is no tie (log alpha {la}, log u {lu})"
            ties += 1
            o.restore(saved, alpha, T_target)                   # back to step `start`, then the kernel's decisions up to s
            o.replay_forced(gprops[start:s + 1], gout["accepted"][start:s + 1])
            start = s + 1
        done += n
    return ties
