"""The C-ABI library builds for gfx950, loads on a CPU-only box and exports every symbol that
include/mpp_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from helpers import REPO
from mpp_cnn_rs_object_detection_amd import build as hip_build
from mpp_cnn_rs_object_detection_amd import hip_api


@pytest.fixture(scope="module")
def lib_path():
    return hip_build.build()


def test_header_and_binding_agree():
    text = open(os.path.join(REPO, "include", "mpp_hip.h")).read()
    declared = set(re.findall(r"\b(mpp_[a-z0-9_]+)\s*\(", text))
    assert declared == set(hip_api.ABI_SYMBOLS)


def test_library_exports_every_symbol(lib_path):
    L = ctypes.CDLL(lib_path)
    for name in hip_api.ABI_SYMBOLS:
        assert hasattr(L, name), name
    assert L.mpp_abi_version() == 9


def test_struct_sizes_match_the_header(lib_path):
    # sizes implied by include/mpp_hip.h with natural alignment
    assert ctypes.sizeof(hip_api.UnitTermC) == 4 + 4 + 8 + 64
    assert ctypes.sizeof(hip_api.PairTermC) == 16 + 16 + 16
    assert ctypes.sizeof(hip_api.ModelC) == 16 + 16 + 8 * 80 + 2 * 48
    assert ctypes.sizeof(hip_api.MappingsC) == 16 + 24 + 24 + 3 * 32 * 8
    assert ctypes.sizeof(hip_api.KernelsC) == 80 + 16 + 8 + 16
    assert hip_api.PROPOSAL_DTYPE.itemsize == 72 and hip_api.STEPOUT_DTYPE.itemsize == 48


def test_philox_host_entry_point(lib_path):
    assert [int(v) for v in hip_api.philox([0, 0, 0, 0], [0, 0])] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_no_silent_fallback_without_gpu(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(hip_api.MppError):
        hip_api.MppContext(0)
