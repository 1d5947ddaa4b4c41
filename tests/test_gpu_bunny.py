"""BASELINE config C1: the example project's `bunny` pair (cameras 7310085 / 7310087) as the
reference ingests it (tests/golden/bunny_pair.npz: Qt smooth scaling 0.25, alpha mask, lens
distortion, 100 levels) -- TwoView WTA + cross-check, HIP vs oracle.

Depth range: the README suggests 300-800, but the projection matrices of example/project.xml
put these two cameras 19 units apart, converging at z ~ 48, and the object at z ~ 41-44 (the
README range projects outside the other image: empty curves).  The test sweeps 30-80."""
import os

import numpy as np
import pytest

import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bunny_pair.npz")


def _load():
    g = np.load(GOLD)
    views = []
    for tag in ("left", "right"):
        views.append((g[tag + "_rgba"], g[tag + "_mask"], (g[tag + "_K"], g[tag + "_R"], g[tag + "_t"]),
                      g[tag + "_dist"], None))
    params = dict(min_depth=30.0, max_depth=80.0, num_depth_levels=100, image_scale=float(g["scale"][0]),
                  window_radius=5, weight_kind=1)
    return dict(name="bunny", kind="twoview", views=views, params=params)


def test_bunny_pair_twoview(hip_ctx):
    case = _load()
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    assert cams[0].is_distorted and cams[1].is_distorted
    cases.upload_case(hip_ctx, case, cams)
    y0, y1 = 84, 108                       # a band through the object: keeps the CPU oracle to seconds
    want_l = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op, y0, y1)
    want_r = O.twoview_wta(imgs[1], imgs[0], ocams[1], ocams[0], op, y0, y1)
    nan = np.full(want_l.shape, np.nan)
    hip_ctx.upload_depth(0, nan)
    hip_ctx.upload_depth(1, nan)
    hip_ctx.twoview_wta(0, 1, p, y0, y1)
    assert not hip_ctx.stats()["used_dense_path"]          # distorted, verged cameras: general kernels
    hip_ctx.twoview_wta(1, 0, p, y0, y1)
    got_l, got_r = hip_ctx.download_depth(0), hip_ctx.download_depth(1)
    ok, msg, _ = cases.compare_depth(got_l[y0:y1], want_l[y0:y1], 1e-9)
    assert ok, "left: " + msg
    ok, msg, _ = cases.compare_depth(got_r[y0:y1], want_r[y0:y1], 1e-9)
    assert ok, "right: " + msg
    fin = np.isfinite(want_l[y0:y1])
    assert fin.sum() > 200, "band misses the object"
    z = want_l[y0:y1][fin]
    assert 38 < np.median(z) < 48                           # the bunny sits at z ~ 41-44
    # cross-check on the band (rows outside are NaN on both sides)
    cl, cr = O.twoview_cross_check(ocams[0], ocams[1], op, want_l, want_r)
    hip_ctx.upload_depth(0, want_l)
    hip_ctx.upload_depth(1, want_r)
    hip_ctx.twoview_cross_check(0, 1, p)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), cl, 1e-9)
    assert ok, "left cross-check: " + msg
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(1), cr, 1e-9)
    assert ok, "right cross-check: " + msg


def _oracle_map(ri, oi, rc, oc, op, H, workers=16):
    """The oracle's whole WTA map, its rows spread over a thread pool (the C oracle runs outside the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    step = max(1, (H + 4 * workers - 1) // (4 * workers))
    bands = [(y, min(H, y + step)) for y in range(0, H, step)]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        parts = list(ex.map(lambda b: O.twoview_wta(ri, oi, rc, oc, op, b[0], b[1]), bands))
    out = np.full(parts[0].shape, np.nan)
    for (a, b), m in zip(bands, parts):
        out[a:b] = m[a:b]
    return out


def test_bunny_pair_whole_maps_and_cross_check(hip_ctx):
    """C1 -- a BASELINE configuration -- in full: the whole 256x192x100 pair, both directions, then the
    ordered cross-check, through srh_twoview_compute (TwoViewStereo::computeDepthMaps) against the oracle's two whole WTA
    maps and its cross-check of them."""
    case = _load()
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    H = imgs[0].h
    want_l = _oracle_map(imgs[0], imgs[1], ocams[0], ocams[1], op, H)
    want_r = _oracle_map(imgs[1], imgs[0], ocams[1], ocams[0], op, H)
    hip_ctx.twoview_wta(0, 1, p)
    got_l = hip_ctx.download_depth(0)
    hip_ctx.twoview_wta(1, 0, p)
    got_r = hip_ctx.download_depth(1)
    for got, want, tag in ((got_l, want_l, "left"), (got_r, want_r, "right")):
        ok, msg, _ = cases.compare_depth(got, want, 1e-9)
        assert ok, tag + ": " + msg
    assert np.isfinite(want_l).sum() > 3000 and np.isfinite(want_r).sum() > 3000       # the object is there
    cl, cr = O.twoview_cross_check(ocams[0], ocams[1], op, want_l, want_r)
    dl, dr = hip_ctx.twoview_compute(0, 1, p)                                        # both passes + cross-check, as the class runs them
    for got, want, tag in ((dl, cl, "left"), (dr, cr, "right")):
        ok, msg, _ = cases.compare_depth(got, want, 1e-9)
        assert ok, "after the cross-check, " + tag + ": " + msg
    assert (np.isfinite(want_l) & ~np.isfinite(cl)).any()                              # the cross-check rejects something


def test_bunny_pair_from_the_project_file(hip_ctx):
    """The same pair with its cameras taken from the project XML fixture through Camera::setP
    (srh_camera_from_p on the product side, sro_camera_set_p on the oracle side): SURVEY 8(f) rank 1."""
    import xml.etree.ElementTree as ET
    g = np.load(GOLD)
    root = ET.parse(os.path.join(os.path.dirname(GOLD), "project_fixture.xml")).getroot()
    ocams, cams = [], []
    for cid in ("7310085", "7310087"):
        cam = [c for c in root.find("cameras") if c.get("id") == cid][0]
        pm, ld = cam.find("projectionMatrix"), cam.find("lensDistortion")
        P = np.array([[float(pm.get("m%d%d" % (i, j))) for j in (1, 2, 3, 4)] for i in (1, 2, 3)])
        dist = np.array([float(ld.get(k, "0")) for k in ("k1", "k2", "p1", "p2", "k3")])
        ocams.append(O.camera_set_p(P, dist))
        cams.append(capi.camera_from_p(P, dist))
        for f in ("K", "R", "t", "C", "pdir", "dist"):
            assert np.array_equal(np.array(getattr(ocams[-1], f)), np.array(getattr(cams[-1], f))), (cid, f)
    params = dict(min_depth=30.0, max_depth=80.0, num_depth_levels=100, image_scale=float(g["scale"][0]),
                  window_radius=5, weight_kind=1)
    op, p = O.params_twoview(**params), capi.params_twoview(**params)
    imgs = [O.OImage(g["left_rgba"], g["left_mask"]), O.OImage(g["right_rgba"], g["right_mask"])]
    hip_ctx.upload_view(0, g["left_rgba"], g["left_mask"], cams[0])
    hip_ctx.upload_view(1, g["right_rgba"], g["right_mask"], cams[1])
    y0, y1 = 90, 102
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op, y0, y1)
    hip_ctx.upload_depth(0, np.full(want.shape, np.nan))
    hip_ctx.twoview_wta(0, 1, p, y0, y1)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0)[y0:y1], want[y0:y1], 1e-9)
    assert ok, msg
    assert np.isfinite(want[y0:y1]).sum() > 100
