#!/usr/bin/env python3
"""Proposals/s of a chain under the contrast energy setup (classic image energies, one-thread slow path) on the 96x96
fixture scene and on a 512x512 scene of the reference's image recipe: python profiles/tools/bench_contrast.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpp_cnn_rs_object_detection_amd import energies as E  # noqa: E402
from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth  # noqa: E402
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps  # noqa: E402

out = {}
for size, n_rect, steps in ((96, 40, 20000), (512, 900, 30000)):
    img, gt_xy, gt_marks = synth.make_scene_image((size, size), n_rect=n_rect, seed=5)
    rng = np.random.default_rng(3)
    det = np.clip(0.05 + 0.9 * (np.abs(img.mean(-1) - 0.5) > 0.3) + rng.normal(0, 0.02, (size, size)), 0.01, 1).astype(np.float32)
    marks = [np.full((size, size, 32), 1 / 32, np.float32) for _ in range(3)]
    data = ImageWMaps(name="0", shape=(size, size), image=img, detection_map=det, param_dist_maps=marks,
                      mappings=mappings.default_mappings(), param_names=["size", "ratio", "angle"], labels=None, gt_config=[])
    for ctype in ("craciun2", "gradient"):
        setup = E.ContrastMeasureEnergySetup(contrast_type=ctype, manual_threshold=-0.3)
        setup.energy_cal = {"detection_thresh": -0.3, "min_area": 20.0, "max_area": 90.0}
        unit, pair = setup.make_energies(data)
        comb = E.ManualHierarchicalEnergyCombinator(dict(zip(setup.NAMES, [1.0, 2.0, 0.5, 0.25, 0.75])), "ContrastEnergy", 0.0)
        for spec in (8, 1):
            ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=spec)
            ctx.set_maps(det, marks)
            ctx.set_image(E.classic_image(unit))
            ctx.set_model(E.build_model_desc(unit, pair, comb), data.mappings)
            ctx.set_points(0, gt_xy, gt_marks)
            ctx.set_kernels(kernels.make_kernels(data.mappings, float(len(gt_xy))))
            ctx.set_schedule(0.1, 0.9999, 0.0)
            ctx.run(2000, seed=1)
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.run(steps, seed=2)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            rec = {"proposals_per_s": steps / dt, "kernel_ms": ctx.last_kernel_ms(), "n_points": int(ctx.count(0))}
            st = ctx.deep_stats()
            if st.get("rounds"):
                rec["deep_rounds"] = st["rounds"]
                rec["steps_committed_per_round"] = st["committed"] / st["rounds"]
                rec["steps_evaluated_per_round"] = st["evaluated"] / st["rounds"]
                if os.environ.get("MPP_LIB_PATH", "").endswith("dprof.so"):      # per-phase clocks of the diagnostic build
                    names = ["A:types", "A:bar1", "A:sort+bar2", "B:draw", "B:pre", "E:delta(total)", "B:post", "bar3", "C:decide+trace+ring",
                             "D:apply-delta", "D:bar4", "D:mutate"]
                    rec["phase_cycles_per_round"] = {f"wave{w}": {k: round(ctx.get_option(f"deep_stat{16 + 24 * w + i}") / st["rounds"])
                                                                  for i, k in enumerate(names)} for w in range(spec)}
            out[f"{size}px_{ctype}_spec{spec}"] = rec
            ctx.close()
print(json.dumps(out, indent=1))
