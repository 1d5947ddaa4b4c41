"""The template scan of the dense TwoView path (twoview_tscan_kernel, srh_dense.hip): the candidate sequence made once per
pass by the reference's operations at one pixel, every pixel verifying label by label (certified projections)
that its own curve is that one, the look-ups running over the shared sequence.  Against the per-pixel curve walk
(twoview_scan_kernel, option tscan = 0): the same depth bits, the same reference-evaluation counts, in both arithmetics;
the statistics say which of the two settled the tiles; a rig the template cannot serve is walked entirely."""
import numpy as np
import pytest

import cases
from stereoreconstruction_amd import capi, synthetic

pytestmark = pytest.mark.gpu

CASES = [("geodesic_rect", dict()), ("adaptive_rect", dict()), ("geodesic_masks", dict()), ("adaptive_masks", dict(w=97, h=53, D=24)),
         ("geodesic_r2", dict()), ("geodesic_rect", dict(w=200, h=70, D=48)), ("adaptive_rect", dict(w=161, h=37, D=130)),
         ("geodesic_scaled", dict())]


def _both(ctx, p, arith):
    out = {}
    for ts in (1, 0):
        ctx.set_option("tscan", ts)
        ctx.set_option("arith", arith)
        try:
            res = []
            for a, b in ((0, 1), (1, 0)):
                ctx.twoview_wta(a, b, p)
                res.append((ctx.download_depth(a), ctx.stats()))
            out[ts] = res
        finally:
            ctx.set_option("tscan", 1)
            ctx.set_option("arith", capi.ARITH_DEFAULT)
    return out


@pytest.mark.parametrize("name,over", CASES)
@pytest.mark.parametrize("arith", [capi.ARITH_CERTIFIED, capi.ARITH_EXACT], ids=["certified", "exact"])
def test_template_scan_equals_the_curve_walk(hip_ctx, name, over, arith):
    case = cases.get_twoview(name, **over)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    out = _both(hip_ctx, p, arith)
    for d in range(2):
        (m1, s1), (m0, s0) = out[1][d], out[0][d]
        assert s1["used_dense_path"] and s0["used_dense_path"]
        assert np.array_equal(m1.view(np.uint64), m0.view(np.uint64)), (name, over, d)
        assert s1["n_eval"] == s0["n_eval"] and s1["n_pixels"] == s0["n_pixels"], (name, d, s1["n_eval"], s0["n_eval"])
        assert s0["scan_tiles_template"] == 0 and s0["scan_tiles_walked"] == 0
        assert s1["scan_tiles_template"] > 0, (name, d, s1)
        # (pixels whose range is cut by the image border verify like any other: the template's columns are clipped per pixel)
        assert s1["scan_tiles_walked"] <= s1["scan_tiles_template"] // 8, (name, d, s1)


def test_a_rig_the_template_cannot_serve_is_walked(hip_ctx):
    """force_dense proposes the dense plan for a verged pair: the template's segments leave the row, every tile goes to the
    curve walk, which refutes the plan (candidates off their row) -- the pass is redone on the general kernels as before."""
    case = cases.get_twoview("adaptive_verged", w=72, h=44, D=20, radius=5)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.twoview_wta(0, 1, p)
    want = hip_ctx.download_depth(0)
    hip_ctx.set_option("force_dense", 1)
    try:
        hip_ctx.twoview_wta(0, 1, p)
        st = hip_ctx.stats()
        got = hip_ctx.download_depth(0)
    finally:
        hip_ctx.set_option("force_dense", 0)
    assert not st["used_dense_path"]
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_template_scan_at_c3_size_with_a_narrow_mask(hip_ctx):
    """C3 size, left -> right and back, a mask that keeps a band of columns next to the left border (where the reference's
    truncation towards zero moves off-image end points) and a diagonal stripe: template scan == curve walk, bit for bit."""
    W, H, D = 1920, 1080, 256
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0003)
    yy, xx = np.mgrid[0:H, 0:W]
    keep = (xx < 300) | (np.abs(xx - 2 * yy) < 150)
    ml = np.where(keep, ml, 0).astype(np.uint8)
    mr = np.where(keep | (xx > W - 280), mr, 0).astype(np.uint8)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    hip_ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    hip_ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
    out = _both(hip_ctx, p, capi.ARITH_CERTIFIED)
    for d in range(2):
        (m1, s1), (m0, s0) = out[1][d], out[0][d]
        assert np.array_equal(m1.view(np.uint64), m0.view(np.uint64)), d
        assert s1["n_eval"] == s0["n_eval"] and s1["scan_tiles_template"] > 0 and s1["scan_tiles_walked"] == 0, (d, s1)


def test_a_template_range_that_ends_on_column_minus_one_at_a_tile_edge(hip_ctx):
    """d0 = 64 = the scan's tile width: the template's largest offset is -64, so for the LAST pixel of the first tile
    x + smax = -1.  The reference truncates that end point towards zero -- the segment ends ON column 0 -- and the look-ups
    move the entry there: its mask byte lies one past the bytes the tile's offsets reach and its cost entry outside the
    clipped template range (the byte was never loaded, the range never checked).  The other view's column 0 is
    masked out on every second row so that the byte matters.  Template scan == curve walk == oracle, both directions."""
    import oracle_ffi as O
    W, H, D, d0 = 200, 36, 24, 64
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0A64, d0=d0)
    mr = mr.copy(); mr[::2, 0] = 0
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D, d0=d0)
    case = dict(name="d0_64", kind="twoview", gt_disparity=None,
                views=[(L, ml, (Kl, Rl, tl), None, None), (R, mr, (Kr, Rr, tr), None, None)],
                params=dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, window_radius=5, weight_kind=1, image_scale=1.0))
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    # the situation the test is about exists: the curve of pixel x = 63 ends on column 0 of the other view
    pts = O.epipolar_curve(ocams[0], ocams[1], imgs[1], op, False, 63, 11)      # (row 11: column 0 of the other view is white there)
    assert len(pts) and int(pts[:, 0].max()) == 0 and int(pts[:, 0].min()) == 0, pts[-4:]
    assert int(O.epipolar_curve(ocams[0], ocams[1], imgs[1], op, False, 64, 11)[:, 0].max()) == 0
    want = [O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op), O.twoview_wta(imgs[1], imgs[0], ocams[1], ocams[0], op)]
    for arith in (capi.ARITH_CERTIFIED, capi.ARITH_EXACT):
        out = _both(hip_ctx, p, arith)
        for d in range(2):
            (m1, s1), (m0, s0) = out[1][d], out[0][d]
            assert s1["used_dense_path"] and s1["scan_tiles_template"] > 0, (d, s1)
            assert np.array_equal(m1.view(np.uint64), m0.view(np.uint64)), (arith, d)
            assert s1["n_eval"] == s0["n_eval"], (arith, d, s1["n_eval"], s0["n_eval"])
            ok, msg, _ = cases.compare_depth(m1, want[d], 1e-9)
            assert ok, (arith, d, msg)


def test_negative_margin_walks_every_visit(hip_ctx):
    """With wta_margin < 0 a revisited winner beats itself (cost + margin < cost) and moves secondBest: the first-visit
    sequence would be wrong there, so the template scan walks every visit -- against the oracle and the curve walk."""
    import oracle_ffi as O
    case = cases.get_twoview("geodesic_rect", w=120, h=40, D=40)
    case["params"] = dict(case["params"], wta_margin=-0.5)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    assert p.wta_margin == -0.5 and op.wta_margin == -0.5
    cases.upload_case(hip_ctx, case, cams)
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    out = _both(hip_ctx, p, capi.ARITH_CERTIFIED)                 # (a negative margin switches the certified arithmetic off: mode 0)
    (m1, s1), (m0, s0) = out[1][0], out[0][0]
    assert s1["scan_tiles_template"] > 0 and s1["n_certified"] == 0
    assert np.array_equal(m1.view(np.uint64), m0.view(np.uint64)) and s1["n_eval"] == s0["n_eval"]
    ok, msg, _ = cases.compare_depth(m1, want, 1e-9)
    assert ok, msg
    # the margin matters on this input: the same pair with the default margin gives another map
    case2 = cases.get_twoview("geodesic_rect", w=120, h=40, D=40)
    cams2, p2 = cases.hip_inputs(case2)
    hip_ctx.twoview_wta(0, 1, p2)
    assert not np.array_equal(hip_ctx.download_depth(0).view(np.uint64), m1.view(np.uint64))


def test_negative_margin_on_general_geometry_takes_the_walk_kernel(hip_ctx):
    """The candidate lists drop joint duplicates, which a negative margin would make matter: such a run goes through the
    one-thread-per-pixel walk kernel -- against the oracle."""
    import oracle_ffi as O
    case = cases.get_twoview("adaptive_verged", w=72, h=44, D=20, radius=5)
    case["params"] = dict(case["params"], wta_margin=-0.25)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    hip_ctx.twoview_wta(0, 1, p)
    assert not hip_ctx.stats()["used_dense_path"]
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), want, 1e-9)
    assert ok, msg
