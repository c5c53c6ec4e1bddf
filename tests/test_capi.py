"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/vampire_hip.h declares, rejects bad descriptors without touching a GPU, and the
Python host refuses CPU tensors (there is no fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT
from vampire_amd import _capi
from vampire_amd.build import build_library


@pytest.fixture(scope="module")
def lib():
    build_library(verbose=False)
    return _capi.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vampire_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vamp_[a-z_0-9]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    names = header_symbols()
    assert len(names) >= 15
    raw = C.CDLL(_capi.lib_path())
    for n in names:
        assert hasattr(raw, n), f"{n} declared in vampire_hip.h but not exported"
    assert set(names) == set(_capi.SIGNATURES), "ctypes binding out of sync with the header"
    assert lib.vamp_abi_version() == _capi.ABI_VERSION


def test_struct_layout_matches_header(lib):
    # 9+... ints and floats, no padding surprises: sizes are part of the ABI
    assert C.sizeof(_capi.VampLiftDesc) == 18 * 4
    assert C.sizeof(_capi.VampRenderDesc) == 29 * 4
    assert C.sizeof(_capi.VampSampleDesc) == 22 * 4
    assert C.sizeof(_capi.VampConvDesc) == 6 * 4


def test_bad_descriptor_is_rejected_without_gpu(lib):
    d = _capi.VampLiftDesc()          # all zeros
    rc = lib.vamp_lift_indices(C.byref(d), None, None, None, None, None, None, None, None, None)
    assert rc == -1
    assert b"requirement failed" in lib.vamp_last_error()
    rd = _capi.VampRenderDesc()
    assert lib.vamp_frustum_geometry(C.byref(rd), None, None, None, None, None, None) == -1
    assert lib.vamp_lift_workspace_bytes(None) == 0


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setenv("VAMPIRE_HIP_LIB", "/nonexistent/libvampire_hip.so")
    with pytest.raises(_capi.VampireHipError):
        _capi.load()
    monkeypatch.delenv("VAMPIRE_HIP_LIB")
    monkeypatch.setattr(_capi, "_lib", None)
    _capi.load()


def test_cpu_tensors_are_refused():
    """The product path must fail loudly rather than fall back to a CPU implementation."""
    from vampire_amd.config import CFG_TINY
    from vampire_amd.ops import _chk
    with pytest.raises(_capi.VampireHipError):
        _chk(torch.zeros(2, 2), (2, 2), "x")
    from vampire_amd import ops
    from vampire_amd.ops import HotPath
    with pytest.raises(_capi.VampireHipError):
        ops.upsample_trilinear(torch.zeros(1, 1, 2, 2, 2), (4, 4, 4))
    with pytest.raises(_capi.VampireHipError):
        ops.conv3d_3x3x3(torch.zeros(1, 16, 2, 2, 2), torch.zeros(16, 16, 3, 3, 3))


def test_new_entry_points_reject_bad_arguments_without_gpu(lib):
    """The widening rows' entry points validate before touching the device."""
    cd = _capi.VampConvDesc()
    cd.B, cd.cin, cd.cout, cd.Z, cd.Y, cd.X = 1, 8, 16, 4, 4, 4           # 8 input channels: unsupported
    assert lib.vamp_conv3d_forward(C.byref(cd), None, None, None, None) == -1
    assert b"cin, cout must be 16 or 32" in lib.vamp_last_error()
    assert lib.vamp_conv3d_workspace_bytes(C.byref(cd)) == 0
    assert lib.vamp_conv3d_supported(C.byref(cd)) == 0
    cd.cin, cd.X = 32, 256                                                  # cfg-A's conv6 row: supported
    assert lib.vamp_conv3d_supported(C.byref(cd)) == 1
    cd.X = 400                                                              # cfg-D: row too long
    assert lib.vamp_conv3d_supported(C.byref(cd)) == 0
    assert lib.vamp_upsample_trilinear_forward(0, 1, 1, 1, 2, 2, 2, None, None, None) == -1
    assert lib.vamp_depth_softmax_forward(1, 0, 4, None, 0, None, None) == -1
    assert lib.vamp_density_gate_forward(1, 4, 8, 7, None, None, None, None) == -1
