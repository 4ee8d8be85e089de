"""What a chain of the shipped setup does, measured with the oracle (CPU): the facts the deep-round kernel is built on
("The deep-round kernel"; profiles/r03_chain_profile.md has the full-size numbers).

After the hot start most ACCEPTED steps re-write a point with the values it already has -- a data-driven transform that
draws the class the mark sits in (the mark map peaks on it), a translation onto the same pixel -- and only a few percent of
all steps change the configuration.  A kernel that ends a speculative round at every accepted step throws that away."""
import numpy as np

import oracle
from helpers import model_for
from mpp_cnn_rs_object_detection_amd import kernels, mappings, synth


def test_most_accepted_steps_of_a_cold_chain_change_nothing():
    tile = synth.make_tile(256, 50, tile_id=0)
    setup, comb, model = model_for("legacy")
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, model, None)
    xy, mk = o.naive_detection(setup.detection_threshold, 6.0)
    kd = kernels.make_kernels(mappings.default_mappings(), float(max(1, len(xy))))
    o = oracle.Oracle(tile.shape, tile.det, tile.marks, model, kd)
    o.set_points(xy, mk)
    o.set_temperature(1.0, 0.999, 0.0)
    n = 30257                                    # the mpp_hrcM schedule
    out, props = o.run(n, 0, chain=0, trace=True)
    pts = [(int(x), int(y), float(s), float(r), float(a)) for (x, y), (s, r, a) in zip(xy, mk)]
    changed = np.zeros(n, bool)
    for i in range(n):
        p, acc, t = props[i], out["accepted"][i], props[i]["target"]
        k = p["kernel"]
        new = (int(p["ax"]), int(p["ay"]), float(p["as"]), float(p["ar"]), float(p["aa"]))
        if k in (0, 2):                          # births
            if acc:
                pts.append(new)
                changed[i] = True
        elif t < 0 or not pts:
            continue
        elif k in (1, 3):                        # deaths: the last point takes the hole
            if acc:
                pts[t] = pts[-1]
                pts.pop()
                changed[i] = True
        elif acc:
            changed[i] = new != pts[t]
            pts[t] = new
    assert len(pts) == out["n_after"][-1]
    acc = out["accepted"].astype(bool)
    cold = slice(8000, n)
    assert acc[cold].mean() > 0.15                               # a fifth of the cold steps is accepted ...
    assert changed[cold].mean() < 0.08                           # ... but only a few percent change the state
    assert (acc[cold] & ~changed[cold]).sum() > 2 * changed[cold].sum()
    assert changed[:3000].mean() > 0.2                           # the hot start is another regime
