"""CPU-side checks of the C-ABI library: it loads, exports every symbol of
include/stereo_recon_hip.h, its host-only entry points agree with the oracle, and the product
path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "stereo_recon_hip.h")).read()
    declared = set(re.findall(r"\b(srh_[a-z_0-9]+)\s*\(", hdr)) - {"srh_progress_fn"}
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.srh_abi_version() == 5


def _bytes(s):
    return bytes(memoryview(s))


def test_struct_layouts_match_the_oracle():
    assert C.sizeof(capi.Camera) == C.sizeof(O.Camera)
    assert C.sizeof(capi.Params) == C.sizeof(O.Params)
    assert [f[0] for f in capi.Params._fields_] == [f[0] for f in O.Params._fields_]
    assert _bytes(capi.params_twoview()) == _bytes(O.params_twoview())
    assert _bytes(capi.params_mvs()) == _bytes(O.params_mvs())
    p = capi.params_twoview()
    # the reference's hard-coded constants (twoviewstereo.cpp:64-80, geodesicweight.cpp:33-41)
    assert (p.window_radius, p.bad_ret, p.max_color_diff, p.second_best_factor, p.inconsistency_thresh) == (5, 1000, 120, 0.95, 1)
    assert (p.geodesic_sigma, p.geodesic_iters, p.geodesic_init, p.adaptive_color_sigma) == (50.0, 3, 1e6, 10.0)
    m = capi.params_mvs()
    assert (m.window_radius, m.top_k, m.num_neighbours, m.peak_threshold) == (2, 9, 3, 0.95)


@pytest.mark.parametrize("name", ["geodesic_rect", "geodesic_distorted", "adaptive_verged", "adaptive_refractive"])
def test_camera_snapshot_matches_oracle(name):
    case = cases.get_twoview(name)
    _, ocams, _ = cases.oracle_inputs(case)
    cams, _ = cases.hip_inputs(case)
    for a, b in zip(cams, ocams):
        assert _bytes(a) == _bytes(b)
    assert cams[0].is_distorted == (1 if "distorted" in name else 0)
    assert cams[0].is_refractive == (1 if "refractive" in name else 0)
    if name == "geodesic_rect":
        assert list(cams[0].pdir) == [0.0, 0.0, 1.0] and list(cams[1].C) == [1.0, 0.0, 0.0]


def test_mvs_neighbour_selection_matches_oracle():
    case = cases.get_mvs("mvs_five_views")
    _, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    assert capi.mvs_neighbours(cams, p) == [[int(v) for v in n] for n in O.mvs_neighbours(ocams, op)]
    # fewer candidates than NUM_NEIGHBOURING_VIEWS: all kept, in index order (multiviewstereo.cpp:351-358)
    assert capi.mvs_neighbours(cams[:3], p) == [[1, 2], [0, 2], [0, 1]]


def test_argument_errors_are_reported():
    L = capi.lib()
    cam = capi.Camera()
    assert L.srh_camera_from_krt(None, None, None, None, None, 0.0, 1.0, C.byref(cam)) == capi.SRH_E_INVALID
    assert b"null" in L.srh_last_error()
    assert L.srh_set_option(None, b"force_generic", 1) == capi.SRH_E_INVALID


def test_no_gpu_means_no_context():
    """Without a HIP device the product refuses to run instead of falling back to the CPU."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.StereoHipError) as ei:
        capi.Context(0)
    assert ei.value.code == capi.SRH_E_NO_DEVICE
    assert "no CPU path" in str(ei.value)
