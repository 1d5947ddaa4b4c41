"""The oracle (oracle/sr_oracle.c) against tests/second_reading.py, an independently written numpy /
python reading of the same reference sources (twoviewstereo.cpp:268-305, 596-672, 909-1054;
multiviewstereo.cpp:113-189, 335-360, 557-604, 666-810; camera.cpp:95-138, 380-459; ray.cpp:53-106;
lineiter.hpp/.cpp).  Rows 4-6, 8-11, 13-17 of SURVEY 8(a) cannot be pinned against a compiled reference
here (Eigen / GSL absent); this gives them a second witness that shares no text with the oracle or the
kernels.  Integer results (candidate lists, winners, classes, neighbour lists) must be identical; real
values agree within 1e-11 relative (numpy's `@` may associate a 3-term sum differently)."""
import ctypes as C
import math

import numpy as np
import pytest

import cases
import oracle_ffi as O
import second_reading as S2

RT = 1e-11


def _close(a, b, rt=RT):
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= rt * max(1.0, abs(a), abs(b))


def _scene(case):
    imgs, masks, cams = [], [], []
    for (rgba, mask, (K, R, t), dist, plane) in case["views"]:
        h, w = rgba.shape[:2]
        imgs.append(S2.VImage(rgba))
        masks.append(S2.mask_image(mask, w, h))
        cams.append(S2.Cam(K, R, t, dist, plane))
    p = case["params"]
    P = S2.Params(min_depth=p["min_depth"], max_depth=p["max_depth"], levels=p["num_depth_levels"],
                  scale=p["image_scale"], radius=p["window_radius"],
                  cross_check=p.get("cross_check_threshold", 1.0))
    return imgs, masks, cams, P


def _pixels(w, h, n, seed):
    rng = np.random.default_rng(seed)
    pts = [(0, 0), (w - 1, h - 1), (w // 2, h // 2), (1, h // 2), (w - 2, 2)]
    while len(pts) < n:
        pts.append((int(rng.integers(0, w)), int(rng.integers(0, h))))
    return pts[:n]


def test_cameras_agree():
    """Camera::set / updatePrincipleRay / unproject / project (distorted and refractive)."""
    for name in ("geodesic_verged_dist_masks", "adaptive_refractive"):
        case = cases.get_twoview(name)
        _, ocams, _ = cases.oracle_inputs(case)
        _, _, cams, _ = _scene(case)
        L = O.lib()
        for oc, sc in zip(ocams, cams):
            for k in range(9):
                assert _close(oc.R[k], sc.R.reshape(9)[k]) and _close(oc.Kinv[k], sc.Kinv.reshape(9)[k])
            for k in range(3):
                assert _close(oc.C[k], sc.C[k]) and _close(oc.pdir[k], sc.pdir[k])
            assert bool(oc.is_distorted) == sc.distorted and bool(oc.is_refractive) == sc.refractive
            src, dr = np.zeros(3), np.zeros(3)
            for (x, y) in ((3.5, 7.5), (40.25, 11.0), (63.5, 39.5)):
                L.sro_unproject(C.byref(oc), x, y, O.dptr(src), O.dptr(dr))
                ray = sc.unproject(x, y)
                assert all(_close(src[k], ray.s[k]) and _close(dr[k], ray.d[k]) for k in range(3))
                for depth in (30.0, 70.0, 200.0):
                    X = ray.s + depth * ray.d
                    pt = np.array(X, dtype=np.float64)
                    ok = L.sro_project(C.byref(oc), O.dptr(pt))
                    amb = S2.AMBIGUOUS_PROJECTIONS
                    xy = sc.project(X)
                    if S2.AMBIGUOUS_PROJECTIONS != amb:
                        continue
                    assert bool(ok) == (xy is not None)
                    if xy is not None:
                        assert _close(pt[0], xy[0], 1e-9) and _close(pt[1], xy[1], 1e-9)


TWOVIEW = [("geodesic_rect", dict(w=48, h=30, D=12), 10),
           ("adaptive_masks", dict(w=48, h=30, D=12, radius=3), 10),
           ("geodesic_verged_dist_masks", dict(w=48, h=30, D=12), 12),
           ("adaptive_refractive", dict(w=40, h=28, D=10), 10),
           ("geodesic_scaled", dict(w=40, h=28, D=10), 10)]


@pytest.mark.parametrize("name,over,npix", TWOVIEW, ids=[t[0] for t in TWOVIEW])
def test_twoview_pixels_agree(name, over, npix):
    case = cases.get_twoview(name, **over)
    oimgs, ocams, op = cases.oracle_inputs(case)
    imgs, masks, cams, P = _scene(case)
    w, h = oimgs[0].w, oimgs[0].h
    maps = {}
    for ref, oth in ((0, 1), (1, 0)):
        depth, diag = O.twoview_wta(oimgs[ref], oimgs[oth], ocams[ref], ocams[oth], op, want_diag=True)
        maps[ref] = depth
        n_finite = n_skipped = 0
        for (x, y) in _pixels(w, h, npix, 17 + ref):
            wt = O.weights(oimgs[ref], x, y, op)            # rows 2/3: pinned by the compiled reference
            amb = S2.AMBIGUOUS_PROJECTIONS
            d2, win, mc, sc, ncand = S2.twoview_pixel(P, imgs[ref], imgs[oth], masks[ref], masks[oth],
                                                      cams[ref], cams[oth], wt, x, y)
            if S2.AMBIGUOUS_PROJECTIONS != amb:             # GSL root order decides in the reference
                n_skipped += 1
                continue
            if S2.is_white(masks[ref].pixel(x, y)):
                cur = O.epipolar_curve(ocams[ref], ocams[oth], oimgs[oth], op, False, x, y)
                ray = cams[ref].unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
                cur2 = S2.epipolar_curve(P, ray, cams[ref].C, cams[ref].pdir, masks[oth], cams[oth], False)
                assert [tuple(c) for c in cur.tolist()] == cur2, (name, ref, x, y)
                for (cx, cy) in cur2[:: max(1, len(cur2) // 4)]:
                    c1 = O.lib().sro_twoview_cost_ncc(C.byref(oimgs[ref].c), C.byref(oimgs[oth].c), O.dptr(wt),
                                                      C.byref(op), x, y, cx, cy)
                    c2 = S2.twoview_cost_ncc(P, imgs[ref], imgs[oth], masks[ref], masks[oth], wt, x, y, cx, cy)
                    assert _close(c1, c2), (name, ref, x, y, cx, cy, c1, c2)
            assert _close(float(depth[y, x]), d2), (name, ref, x, y, depth[y, x], d2)
            ow = tuple(diag["win_xy"][y, x])
            assert (ow == (-1, -1) and win is None) or ow == win, (name, ref, x, y, ow, win)
            assert _close(float(diag["min_cost"][y, x]), mc) and _close(float(diag["second_cost"][y, x]), sc)
            n_finite += math.isfinite(d2)
        assert n_finite >= 1 and n_skipped <= npix // 2
    # crossCheck: left pass on the raw maps, right pass on the filtered left map (twoviewstereo.cpp:604-670)
    cl, cr = O.twoview_cross_check(ocams[0], ocams[1], op, maps[0], maps[1])
    n_checked = 0
    for (x, y) in _pixels(w, h, 40, 5):
        amb = S2.AMBIGUOUS_PROJECTIONS
        got_l = S2.twoview_cross_check_pixel(P, cams[0], cams[1], float(maps[0][y, x]), maps[1], x, y)
        got_r = S2.twoview_cross_check_pixel(P, cams[1], cams[0], float(maps[1][y, x]), cl, x, y)
        if S2.AMBIGUOUS_PROJECTIONS != amb:
            continue
        assert _close(float(cl[y, x]), got_l), ("left", x, y)
        assert _close(float(cr[y, x]), got_r), ("right", x, y)
        n_checked += 1
    assert n_checked >= 20


MVS = [("mvs_geodesic", dict(w=40, h=28, D=12, nviews=4), 8),
       ("mvs_distorted", dict(w=40, h=28, D=12), 8),
       ("mvs_refractive", dict(w=40, h=28, D=10), 8),
       ("mvs_scaled", dict(w=40, h=28, D=10), 8)]


@pytest.mark.parametrize("name,over,npix", MVS, ids=[t[0] for t in MVS])
def test_mvs_pixels_agree(name, over, npix):
    case = cases.get_mvs(name, **over)
    oimgs, ocams, op = cases.oracle_inputs(case)
    imgs, masks, cams, P = _scene(case)
    neigh = O.mvs_neighbours(ocams, op)
    assert [list(map(int, n)) for n in neigh] == S2.mvs_neighbours(P, cams)
    depths = []
    n_skipped = 0
    for v in range(len(cams)):
        d, peaks, _ = O.mvs_initial_estimate(oimgs, ocams, v, neigh[v], op, want_peaks=True)
        depths.append(d)
        if v > 1:
            continue
        w, h = oimgs[v].w, oimgs[v].h
        rng = np.random.default_rng(3 + v)
        ys, xs = np.nonzero(case["views"][v][1] == 1)
        pick = rng.choice(len(xs), size=min(npix, len(xs)), replace=False)
        pts = [(int(xs[k]), int(ys[k])) for k in pick] + [(0, 0), (w - 1, h // 2)]
        for (x, y) in pts:
            wt = O.weights(oimgs[v], x, y, op)
            amb = S2.AMBIGUOUS_PROJECTIONS
            d2, pk2, _ = S2.mvs_pixel(P, imgs, masks, cams, v, neigh[v], wt, x, y)
            if S2.AMBIGUOUS_PROJECTIONS != amb:             # GSL root order decides in the reference
                n_skipped += 1
                continue
            if S2.is_white(masks[v].pixel(x, y)):
                for v2 in neigh[v]:
                    cur = O.epipolar_curve(ocams[v], ocams[v2], oimgs[v2], op, True, x, y)
                    ray = cams[v].unproject((x + 0.5) / P.scale, (y + 0.5) / P.scale)
                    cur2 = S2.epipolar_curve(P, ray, cams[v].C, cams[v].pdir, masks[v2], cams[v2], True)
                    assert [tuple(c) for c in cur.tolist()] == cur2, (name, v, v2, x, y)
                for k in range(P.K):
                    assert _close(float(peaks[y, x, k, 0]), pk2[k][0]) and _close(float(peaks[y, x, k, 1]), pk2[k][1]), \
                        (name, v, x, y, k, peaks[y, x, k], pk2[k])
            assert _close(float(d[y, x]), d2), (name, v, x, y, d[y, x], d2)
    # ordered cross-check chain (multiviewstereo.cpp:427-431, 666-729): view 0 on raw maps, view 1 on the result
    assert n_skipped <= npix
    work = [d.copy() for d in depths]
    n_checked = 0
    for v in range(2):
        before = [m.copy() for m in work]
        O.mvs_cross_check(oimgs, ocams, v, op, work)
        h, w = work[v].shape
        fin = np.argwhere(np.isfinite(before[v]))
        rng = np.random.default_rng(9 + v)
        for k in rng.choice(len(fin), size=min(30, len(fin)), replace=False):
            y, x = map(int, fin[k])
            amb = S2.AMBIGUOUS_PROJECTIONS
            got = S2.mvs_cross_check_pixel(P, cams, before, v, x, y)
            if S2.AMBIGUOUS_PROJECTIONS != amb:
                continue
            assert _close(float(work[v][y, x]), got), (name, v, x, y, work[v][y, x], got)
            n_checked += 1
    assert n_checked >= 10


def test_gui_curve_preview_and_refraction_error_agree():
    """StereoWidget::epipolarLineItem (stereowidget.cpp:621-672) and RefractiveCalibrationFunction::diff
    (refractioncalibration.cpp:175-199): same vertices / same errors from both readings."""
    for name in ("geodesic_verged_dist_masks", "adaptive_refractive", "geodesic_rect"):
        case = cases.get_twoview(name)
        _, ocams, _ = cases.oracle_inputs(case)
        _, _, cams, P = _scene(case)
        h, w = case["views"][0][0].shape[:2]
        before = S2.AMBIGUOUS_PROJECTIONS
        for (x, y) in _pixels(w, h, 12, 77):
            for nd in (2, 37, 500):
                S2.AMBIGUOUS_PROJECTIONS = 0
                want = S2.epipolar_preview(cams[0], cams[1], x, y, P.min_depth, P.max_depth, nd)
                if S2.AMBIGUOUS_PROJECTIONS:
                    continue
                got = O.epipolar_preview(ocams[0], ocams[1], x, y, P.min_depth, P.max_depth, nd)
                assert len(got) == len(want), (name, x, y, nd)
                for g, s in zip(got, want):
                    assert _close(g[0], s[0], 1e-9) and _close(g[1], s[1], 1e-9)
        rng = np.random.default_rng(5)
        for _ in range(40):
            p1 = rng.uniform(0, [w, h]); p2 = rng.uniform(0, [w, h])
            assert _close(O.refraction_pair_error(ocams[0], ocams[1], p1, p2),
                          S2.refraction_pair_error(cams[0], cams[1], p1, p2), 1e-9)
        S2.AMBIGUOUS_PROJECTIONS = before
