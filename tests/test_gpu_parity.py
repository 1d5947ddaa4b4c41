"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Tolerance: finite / +INF / NaN / -1 classification
identical, finite depths within 1e-9*max(1,|z|).  All arithmetic is double on both
sides; integer work (candidate pixels, winners) is bit-exact by construction.
"""
import numpy as np
import pytest

import cases
import oracle_ffi as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _twoview_oracle(case):
    imgs, cams, p = cases.oracle_inputs(case)
    dl = O.twoview_wta(imgs[0], imgs[1], cams[0], cams[1], p)
    dr = O.twoview_wta(imgs[1], imgs[0], cams[1], cams[0], p)
    return (imgs, cams, p), dl, dr


@pytest.mark.parametrize("name", sorted(cases.TWOVIEW_CASES))
def test_twoview_wta_and_cross_check(hip_ctx, name):
    case = cases.get_twoview(name)
    (imgs, ocams, op), dl, dr = _twoview_oracle(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)

    hip_ctx.twoview_wta(0, 1, p)
    gl = hip_ctx.download_depth(0)
    st = hip_ctx.stats()
    hip_ctx.twoview_wta(1, 0, p)
    gr = hip_ctx.download_depth(1)
    ok, msg, _ = cases.compare_depth(gl, dl, RTOL)
    assert ok, "left WTA: " + msg
    ok, msg, _ = cases.compare_depth(gr, dr, RTOL)
    assert ok, "right WTA: " + msg
    # the one-thread-per-pixel curve-walk kernel (last resort) must give the same bits
    hip_ctx.set_option("force_generic", 2)
    hip_ctx.twoview_wta(0, 1, p)
    walk = hip_ctx.download_depth(0)
    hip_ctx.set_option("force_generic", 0)
    assert np.array_equal(gl.view(np.uint64), walk.view(np.uint64)), "default path and curve-walk kernel differ"
    assert st["n_pixels"] == int((case["views"][0][1] == 1).sum())
    assert np.isfinite(dl).sum() > 0.2 * dl.size, "degenerate case: too few finite depths"

    # cross-check, starting from the oracle's WTA maps on both sides
    cl, cr = O.twoview_cross_check(ocams[0], ocams[1], op, dl, dr)
    hip_ctx.upload_depth(0, dl)
    hip_ctx.upload_depth(1, dr)
    hip_ctx.twoview_cross_check(0, 1, p)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), cl, RTOL)
    assert ok, "left cross-check: " + msg
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(1), cr, RTOL)
    assert ok, "right cross-check: " + msg


def test_twoview_compute_end_to_end(hip_ctx):
    case = cases.get_twoview("geodesic_masks")
    (imgs, ocams, op), dl, dr = _twoview_oracle(case)
    cl, cr = O.twoview_cross_check(ocams[0], ocams[1], op, dl, dr)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    steps = []
    hip_ctx.set_hooks(None, lambda step, stage: steps.append(step))
    gl, gr = hip_ctx.twoview_compute(0, 1, p)
    hip_ctx.set_hooks(None, None)
    assert steps == [1, 3, 5, 8]            # TwoViewStereo progress steps (twoviewstereo.cpp:234,405,597,225)
    ok, msg, _ = cases.compare_depth(gl, cl, RTOL)
    assert ok, msg
    ok, msg, _ = cases.compare_depth(gr, cr, RTOL)
    assert ok, msg


def test_twoview_row_band_and_empty(hip_ctx):
    case = cases.get_twoview("adaptive_rect")
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op, 10, 17)
    hip_ctx.upload_depth(0, np.full(want.shape, np.nan))
    hip_ctx.twoview_wta(0, 1, p, 10, 17)
    got = hip_ctx.download_depth(0)
    ok, msg, _ = cases.compare_depth(got[10:17], want[10:17], RTOL)
    assert ok, msg
    assert np.isnan(got[:10]).all() and np.isnan(got[17:]).all()   # rows outside the band untouched
    # all-masked reference view: every depth is NaN, no evaluations
    rgba, mask, cam, dist, plane = case["views"][0]
    hip_ctx.upload_view(0, rgba, np.zeros_like(mask), cams[0])
    hip_ctx.twoview_wta(0, 1, p)
    assert np.isnan(hip_ctx.download_depth(0)).all()
    assert hip_ctx.stats()["n_eval"] == 0


def _mvs_oracle(case):
    imgs, cams, p = cases.oracle_inputs(case)
    neigh = O.mvs_neighbours(cams, p)
    depths = []
    for v in range(len(cams)):
        d, _ = O.mvs_initial_estimate(imgs, cams, v, neigh[v], p)
        depths.append(d)
    return (imgs, cams, p), neigh, depths


@pytest.mark.parametrize("name", sorted(cases.MVS_CASES))
def test_mvs_initial_estimate_and_cross_check(hip_ctx, name):
    from stereoreconstruction_amd import capi
    case = cases.get_mvs(name)
    (imgs, ocams, op), neigh, want = _mvs_oracle(case)
    cams, p = cases.hip_inputs(case)
    assert capi.mvs_neighbours(cams, p) == [[int(x) for x in n] for n in neigh]
    cases.upload_case(hip_ctx, case, cams)
    nv = len(cams)
    some_peak = False
    for v in range(nv):
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        got = hip_ctx.download_depth(v)
        ok, msg, _ = cases.compare_depth(got, want[v], RTOL)
        assert ok, "view %d initial estimate: %s" % (v, msg)
        n_eval = hip_ctx.stats()["n_eval"]
        # the one-thread-per-pixel kernel (walk, costs and top-K bookkeeping inline) gives the same bits
        hip_ctx.set_option("force_generic", 1)
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        hip_ctx.set_option("force_generic", 0)
        assert np.array_equal(got.view(np.uint64), hip_ctx.download_depth(v).view(np.uint64)), "view %d: two-stage vs inline kernel" % v
        assert hip_ctx.stats()["n_eval"] == n_eval
        # the gathering list cost kernel alone (no LDS copies of the other view's window boxes) gives the same bits
        hip_ctx.set_option("mvs_staged", 0)
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        hip_ctx.set_option("mvs_staged", 1)
        assert np.array_equal(got.view(np.uint64), hip_ctx.download_depth(v).view(np.uint64)), "view %d: staged vs gathering cost kernel" % v
        assert hip_ctx.stats()["n_eval"] == n_eval
        some_peak |= bool((want[v][np.isfinite(want[v])] > 0).any())
    assert some_peak, "degenerate case: no NCC peak above threshold anywhere"
    # cross-check in view order, each view reading the already filtered earlier views
    ref = [w.copy() for w in want]
    for v in range(nv):
        O.mvs_cross_check(imgs, ocams, v, op, ref)
    for v in range(nv):
        hip_ctx.upload_depth(v, want[v])
    for v in range(nv):
        hip_ctx.mvs_cross_check(list(range(nv)), v, p)
    for v in range(nv):
        ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(v), ref[v], RTOL)
        assert ok, "view %d cross-check: %s" % (v, msg)


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("name,over", [
    ("geodesic_rect", dict(w=100, h=37, D=24)),                 # width not a multiple of the 32-pixel tile
    ("adaptive_rect", dict(w=75, h=20, D=40)),
    ("geodesic_masks", dict(w=90, h=33, D=20)),                 # masked taps: general form inside the dense kernel
    ("geodesic_r2", dict(w=70, h=30, D=18)),
])
def test_dense_path_matches_oracle_and_generic(hip_ctx, name, over):
    """Row-aligned geometry takes the LDS-tiled dense kernels; they must reproduce the
    oracle and be bit-identical to both general-geometry paths (candidate lists, curve walk)."""
    case = cases.get_twoview(name, **over)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    for ref, oth in ((0, 1), (1, 0)):
        want, diag = O.twoview_wta(imgs[ref], imgs[oth], ocams[ref], ocams[oth], op, want_diag=True)
        hip_ctx.set_option("force_generic", 0)
        hip_ctx.set_option("band_budget_mb", 1)                 # several row bands
        hip_ctx.twoview_wta(ref, oth, p)
        dense = hip_ctx.download_depth(ref)
        st = hip_ctx.stats()
        assert st["used_dense_path"], "rectified case did not take the dense path"
        assert st["n_eval"] == diag["n_eval"]                   # candidates the reference evaluates
        assert st["n_eval_device"] <= st["n_eval"]              # joint duplicates are evaluated once
        others = {}
        # the general-geometry paths: candidate lists evaluated in row runs / in list order, curve walk
        for mode, rows, tag in ((1, 1, "row-run list"), (1, 0, "ordered list"), (2, 1, "curve-walk")):
            hip_ctx.set_option("force_generic", mode)
            hip_ctx.set_option("list_rows", rows)
            hip_ctx.twoview_wta(ref, oth, p)
            others[tag] = hip_ctx.download_depth(ref)
            st2 = hip_ctx.stats()
            assert not st2["used_dense_path"] and st2["n_eval"] == diag["n_eval"]
        hip_ctx.set_option("force_generic", 0)
        hip_ctx.set_option("list_rows", 1)
        hip_ctx.set_option("band_budget_mb", 32768)
        ok, msg, _ = cases.compare_depth(dense, want, RTOL)
        assert ok, "dense vs oracle (ref %d): %s" % (ref, msg)
        for tag, other in others.items():
            assert _same_bits(dense, other), "dense and %s kernels differ (ref %d)" % (tag, ref)


def test_non_aligned_geometry_falls_back(hip_ctx):
    case = cases.get_twoview("adaptive_verged", radius=5)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.twoview_wta(0, 1, p)
    assert not hip_ctx.stats()["used_dense_path"]
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), want, RTOL)
    assert ok, msg


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_rigs(hip_ctx, seed):
    """Randomised two-view rigs (rotation, baseline direction, focal lengths, distortion, masks,
    refraction, scale, radius, weight kind): the default path against the oracle."""
    rng = np.random.default_rng(seed)
    w, h, D = int(rng.integers(36, 72)), int(rng.integers(24, 44)), int(rng.integers(6, 20))
    L, R, ml, mr, _ = cases.S.rectified_pair(w, h, D, 0x5EED0D00 + seed)
    scale = float(rng.choice([1.0, 1.0, 0.5]))
    f = w * rng.uniform(0.8, 1.3) / scale
    def K():
        return np.array([[f * rng.uniform(0.97, 1.03), 0, (w / 2 + rng.uniform(-3, 3)) / scale],
                         [0, f * rng.uniform(0.97, 1.03), (h / 2 + rng.uniform(-3, 3)) / scale], [0, 0, 1.0]])
    def rot(sc):
        a = rng.uniform(-sc, sc, 3)
        return cases._rot_z(a[2]) @ cases._rot_x(a[0]) @ cases._rot_y(a[1])
    Rl, Rr = rot(0.03), rot(0.06)
    Cl = np.zeros(3)
    Cr = np.array([1.0, rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)])
    dist = (lambda: np.array([rng.uniform(-0.2, 0.1), rng.uniform(-0.3, 0.3), rng.uniform(-5e-3, 5e-3),
                              rng.uniform(-5e-3, 5e-3), rng.uniform(-0.5, 0.5)])) if rng.random() < 0.6 else (lambda: None)
    plane = (np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 1.0]), 0.1, 1.333) if rng.random() < 0.35 else None
    if rng.random() < 0.5:
        yy, xx = np.mgrid[0:h, 0:w]
        ml = ((xx + yy * 0.3) % 11 != 0).astype(np.uint8)
        mr = ((xx * 0.5 + yy) % 13 != 0).astype(np.uint8)
    zmid = (w / scale) * scale / (8 + D / 2.0)          # f*B/d with f ~ w/scale pixels of the unscaled image ...
    zmin, zmax = cases.S.rectified_depth_range(w, D)
    zmin, zmax = zmin * (f * scale / w), zmax * (f * scale / w)
    params = dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, window_radius=int(rng.integers(1, 6)),
                  weight_kind=int(rng.integers(0, 2)), image_scale=scale)
    case = dict(name="random%d" % seed, kind="twoview",
                views=[(L, ml, (K(), Rl, -Rl @ Cl), dist(), plane), (R, mr, (K(), Rr, -Rr @ Cr), dist(), plane)],
                params=params)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    total_finite = 0
    for ref, oth in ((0, 1), (1, 0)):
        want = O.twoview_wta(imgs[ref], imgs[oth], ocams[ref], ocams[oth], op)
        hip_ctx.twoview_wta(ref, oth, p)
        got = hip_ctx.download_depth(ref)
        ok, msg, _ = cases.compare_depth(got, want, RTOL)
        assert ok, "seed %d ref %d: %s" % (seed, ref, msg)
        hip_ctx.set_option("list_rows", 0)                      # list-order evaluation: same bits
        hip_ctx.twoview_wta(ref, oth, p)
        hip_ctx.set_option("list_rows", 1)
        assert _same_bits(got, hip_ctx.download_depth(ref)), "seed %d ref %d: row-run and ordered lists differ" % (seed, ref)
        total_finite += int(np.isfinite(want).sum())
    assert total_finite > 0, "degenerate random rig"


def test_steep_curves_use_ordered_lists(hip_ctx):
    """A vertical baseline makes every epipolar curve cross more image rows than the row-run
    evaluation holds (32): the pair is evaluated in list order, same result as the curve walk."""
    w, h, D = 48, 96, 60
    L, R, ml, mr, _ = cases.S.rectified_pair(h, w, D, 0x57EE9)     # build wide, then transpose to tall
    L = np.ascontiguousarray(np.transpose(L, (1, 0, 2))); R = np.ascontiguousarray(np.transpose(R, (1, 0, 2)))
    f = float(h)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    I = np.eye(3)
    zmin, zmax = cases.S.rectified_depth_range(h, D)
    case = dict(name="vertical", kind="twoview",
                views=[(L, None, (K, I, np.zeros(3)), None, None), (R, None, (K, I, -np.array([0.0, 1.0, 0.0])), None, None)],
                params=dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, window_radius=2, weight_kind=0, image_scale=1.0))
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    hip_ctx.twoview_wta(0, 1, p)
    got = hip_ctx.download_depth(0)
    assert not hip_ctx.stats()["used_dense_path"]
    ok, msg, _ = cases.compare_depth(got, want, RTOL)
    assert ok, msg
    assert np.isfinite(want).sum() > 0
    hip_ctx.set_option("force_generic", 2)
    hip_ctx.twoview_wta(0, 1, p)
    hip_ctx.set_option("force_generic", 0)
    assert _same_bits(got, hip_ctx.download_depth(0))


@pytest.mark.parametrize("name,mvs", [
    ("adaptive_rect", False), ("geodesic_verged_dist_masks", False), ("adaptive_refractive", False),
    ("geodesic_scaled", False), ("mvs_geodesic", True), ("mvs_distorted", True),
])
def test_epipolar_curves_are_the_oracles_point_lists(hip_ctx, name, mvs):
    """SURVEY 8(a) #6/#7/#14, integer work: the candidate list of a pixel -- points, order, joint
    duplicates (TwoView) / consecutive-duplicate removal and clipping (MVS) -- is bit-exact."""
    case = cases.get_mvs(name) if mvs else cases.get_twoview(name)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    h, w = case["views"][0][0].shape[:2]
    rng = np.random.default_rng(7)
    xy = np.stack([rng.integers(0, w, 96), rng.integers(0, h, 96)], axis=1)
    xy = np.concatenate([xy, [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]]]).astype(np.int32)
    nonempty = 0
    for ref, oth in ((0, 1), (1, 0)):
        got = hip_ctx.epipolar_curves(ref, oth, p, xy, mvs=mvs, max_pts=8)       # small cap: exercises the regrow
        for (x, y), g in zip(xy, got):
            want = O.epipolar_curve(ocams[ref], ocams[oth], imgs[oth], op, mvs, int(x), int(y))
            assert g.shape == want.shape and np.array_equal(g, want), "pixel (%d,%d) ref %d" % (x, y, ref)
            nonempty += len(want) > 0
    assert nonempty > 50
