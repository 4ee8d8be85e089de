"""The binding INTEGRATION.md section B shows a reference maintainer -- executed VERBATIM against libmppgpu.so, so that the
documented struct layouts, prototypes and call order cannot drift from the library.  The stub imports the REFERENCE's
``Rectangle`` (``base.shapes.rectangle``); here that module name is served by this repository's twin of the class, and the
arguments are stand-ins with the reference's attribute names (``energy_setup.energy_calibration.detection_threshold`` ...)."""
import ctypes
import os
import re
import sys
import types

import numpy as np
import pytest

from helpers import REPO, hrc_model
from mpp_cnn_rs_object_detection_amd import hip_api, mappings, synth
from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
from mpp_cnn_rs_object_detection_amd.shapes import Rectangle

pytestmark = pytest.mark.gpu


def stub_source():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# models/mpp/gpu_backend.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md lost its section B stub"
    return m.group(1)


def test_the_documented_binding_runs_and_equals_the_host_code(monkeypatch):
    import torch  # noqa: F401  (before the library is opened: INTEGRATION.md "Loading order")
    from mpp_cnn_rs_object_detection_amd.sampler import sample_rjmcmc_batch
    base, shapes, rect = types.ModuleType("base"), types.ModuleType("base.shapes"), types.ModuleType("base.shapes.rectangle")
    rect.Rectangle = Rectangle
    base.shapes, shapes.rectangle = shapes, rect
    for name, mod in (("base", base), ("base.shapes", shapes), ("base.shapes.rectangle", rect)):
        monkeypatch.setitem(sys.modules, name, mod)
    real_cdll = ctypes.CDLL
    monkeypatch.setattr(ctypes, "CDLL", lambda name, *a, **k: real_cdll(hip_api.LIB_PATH if name == "libmppgpu.so" else name, *a, **k))
    ns = {}
    exec(compile(stub_source(), "INTEGRATION.md#B", "exec"), ns)

    tiles = [synth.make_tile(96, 14, tile_id=80 + i, noise=0.1) for i in range(3)]
    patches = [ImageWMaps(name=str(i), shape=t.shape, image=None, detection_map=t.det, param_dist_maps=t.marks,
                          mappings=mappings.default_mappings(), param_names=Rectangle.PARAMETERS, gt_config=[])
               for i, t in enumerate(tiles)]
    setup, comb = hrc_model()
    ref_setup = types.SimpleNamespace(energy_calibration=types.SimpleNamespace(**setup.energy_calibration),
                                      detection_threshold=setup.detection_threshold)
    sched = dict(init_temperature=1.0, alpha_t=0.998, burn_in=3000, samples_interval=64, target_temperature=0.0)
    got = ns["sample_tiles_gpu"](patches, ref_setup, comb, np.random.default_rng(5), **sched)
    want = sample_rjmcmc_batch(patches, np.random.default_rng(5), 1, comb, "naive", energy_setup=setup, spec_waves=1, **sched)
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert [p.as_row() for p in g[-1]] == [p.as_row() for p in w[-1]] and len(g[-1]) > 5
