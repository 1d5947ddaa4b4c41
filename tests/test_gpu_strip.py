"""The persistent strip form of the dense TwoView cost kernel (srh_strip.hip) against the per-tile kernel
(srh_dense.hip) and the oracle: TwoViewStereo::cost_ncc, stereo/twoviewstereo.cpp:909-977.

Both forms of the strip kernel (4 waves / 8 block lanes, 8 waves / 16 block lanes) and the per-tile kernel must give
the same 64-bit patterns and the same evaluation counts; the default path is also compared with the oracle.
"""
import numpy as np
import pytest

import cases
import oracle_ffi as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9

# (case, overrides): rectified pairs of several shapes -- image widths that are not a tile multiple, ranges that
# touch both image borders, masks (general-form candidates), radius 2, strips shorter and longer than an item
STRIP_CASES = [
    ("geodesic_rect", dict()),
    ("adaptive_rect", dict()),
    ("geodesic_masks", dict()),
    ("adaptive_masks", dict(w=97, h=53, D=24)),
    ("geodesic_r2", dict()),
    ("geodesic_rect", dict(w=200, h=70, D=48)),
    ("adaptive_rect", dict(w=161, h=37, D=130)),        # 17+ blocks per pixel: the 8-wave form by default
    ("geodesic_scaled", dict()),
]


def _run(ctx, p, strip):
    ctx.set_option("strip", strip)
    out = []
    for a, b in ((0, 1), (1, 0)):
        ctx.twoview_wta(a, b, p)
        st = ctx.stats()
        out.append((ctx.download_depth(a), st))
    ctx.set_option("strip", 1)
    return out


@pytest.mark.parametrize("name,over", STRIP_CASES)
def test_strip_forms_match_the_per_tile_kernel_and_the_oracle(hip_ctx, name, over):
    case = cases.get_twoview(name, **over)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    ref = _run(hip_ctx, p, 0)
    assert not ref[0][1]["used_strip_kernel"] and ref[0][1]["used_dense_path"]
    for strip in (4, 8):          # (strip = 1, the default, takes the strip kernel only for images of thousands of tiles)
        got = _run(hip_ctx, p, strip)
        for d in range(2):
            assert got[d][1]["used_strip_kernel"], "strip=%d: the strip kernel did not run" % strip
            assert np.array_equal(got[d][0].view(np.uint64), ref[d][0].view(np.uint64)), \
                "strip=%d direction %d: depth bits differ from the per-tile kernel" % (strip, d)
            assert got[d][1]["n_eval"] == ref[d][1]["n_eval"]
            if strip == 4:      # same block geometry as the per-tile kernel (16 block lanes leave other columns to the scan)
                # (but for the exact redo of flagged pixels, a cost row each: the strip kernel settles uncovered candidates and
                # flat-window pixels itself, so the two kernels' scans flag different pixels)
                slack = (got[d][1]["n_flagged"] + ref[d][1]["n_flagged"]) * (2 * p.num_depth_levels + 8)
                assert abs(got[d][1]["n_eval_device"] - ref[d][1]["n_eval_device"]) <= slack
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    ok, msg, _ = cases.compare_depth(_run(hip_ctx, p, 8)[0][0], want, RTOL)
    assert ok, msg


def test_strip_row_bands_and_small_budget(hip_ctx):
    """Row bands (y0, y1) and a band budget that splits the image into several launches give the same map."""
    case = cases.get_twoview("geodesic_masks", w=96, h=64, D=20)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.set_option("strip", 0)
    hip_ctx.twoview_wta(0, 1, p)
    full = hip_ctx.download_depth(0)
    hip_ctx.set_option("strip", 4)
    try:
        hip_ctx.upload_depth(0, np.full(full.shape, np.nan))
        for y0, y1 in ((0, 7), (7, 30), (30, 64)):
            hip_ctx.twoview_wta(0, 1, p, y0, y1)
            assert hip_ctx.stats()["used_strip_kernel"]
        assert np.array_equal(hip_ctx.download_depth(0).view(np.uint64), full.view(np.uint64))
        hip_ctx.set_option("band_budget_mb", 1)
        hip_ctx.twoview_wta(0, 1, p)
        assert hip_ctx.stats()["used_strip_kernel"]
        assert np.array_equal(hip_ctx.download_depth(0).view(np.uint64), full.view(np.uint64))
    finally:
        hip_ctx.set_option("band_budget_mb", 32768)
        hip_ctx.set_option("strip", 1)


def test_strip_falls_back_when_a_range_is_wider_than_a_chunk(hip_ctx):
    """More candidate columns than one LDS chunk holds (320): the host takes the per-tile kernel, same result as generic."""
    case = cases.get_twoview("adaptive_rect", w=400, h=24, D=330)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.set_option("strip", 8)
    try:
        hip_ctx.twoview_wta(0, 1, p)
    finally:
        hip_ctx.set_option("strip", 1)
    st = hip_ctx.stats()
    got = hip_ctx.download_depth(0)
    assert st["used_dense_path"] and not st["used_strip_kernel"]
    hip_ctx.set_option("force_generic", 2)
    hip_ctx.twoview_wta(0, 1, p)
    hip_ctx.set_option("force_generic", 0)
    assert np.array_equal(got.view(np.uint64), hip_ctx.download_depth(0).view(np.uint64))
