"""Full-size runs of the BASELINE configurations, checked through size-independent properties
(the oracle needs hours at these sizes):
  * the tuned dense kernels and the general curve-walk kernels agree bit for bit,
  * the result does not depend on how the image is cut into row bands,
  * one full-width row agrees with the oracle,
  * reference-evaluation counts (n_eval) are identical on both paths.
"""
import numpy as np
import pytest

import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def _setup(ctx, W, H, D, seed, wkind):
    L, R, ml, mr, disp = synthetic.rectified_pair(W, H, D, seed)
    cams3 = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    (Kl, Rl, tl), (Kr, Rr, tr) = cams3
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind)
    op = O.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind)
    return (L, R, ml, mr), cams3, p, op


def test_c2_whole_maps_and_cross_check_against_the_oracle(hip_ctx):
    """C2 -- a BASELINE configuration (640x480, 64 levels, AdaptiveWeight r = 5) -- IN FULL against the oracle: both WTA maps,
    every pixel (the oracle's rows on 16 host threads, ~10 s), then srh_twoview_compute (both passes + the ordered
    cross-check, as TwoViewStereo::computeDepthMaps runs them) against the oracle's cross-check of its own two maps."""
    from concurrent.futures import ThreadPoolExecutor
    W, H, D = 640, 480, 64
    (L, R, ml, mr), cams3, p, op = _setup(hip_ctx, W, H, D, 0x5EED0002, capi.WEIGHT_ADAPTIVE)
    (Kl, Rl, tl), (Kr, Rr, tr) = cams3
    oi = [O.OImage(L, ml), O.OImage(R, mr)]
    oc = [O.camera_set(Kl, Rl, tl), O.camera_set(Kr, Rr, tr)]
    bands = [(y, min(H, y + 8)) for y in range(0, H, 8)]

    def whole(ref, oth):
        with ThreadPoolExecutor(max_workers=16) as ex:
            parts = list(ex.map(lambda b: O.twoview_wta(oi[ref], oi[oth], oc[ref], oc[oth], op, b[0], b[1]), bands))
        out = np.full((H, W), np.nan)
        for (a, b), m in zip(bands, parts):
            out[a:b] = m[a:b]
        return out
    want = [whole(0, 1), whole(1, 0)]
    for ref, oth in ((0, 1), (1, 0)):
        hip_ctx.twoview_wta(ref, oth, p)
        assert hip_ctx.stats()["used_dense_path"]
        ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(ref), want[ref], 1e-9)
        assert ok, (ref, msg)
    cl, cr = O.twoview_cross_check(oc[0], oc[1], op, want[0], want[1])
    dl, dr = hip_ctx.twoview_compute(0, 1, p)
    for got, w_, tag in ((dl, cl, "left"), (dr, cr, "right")):
        ok, msg, _ = cases.compare_depth(got, w_, 1e-9)
        assert ok, ("after the cross-check", tag, msg)
    assert np.isfinite(cl).mean() > 0.2 and (np.isfinite(want[0]) & ~np.isfinite(cl)).any()


def test_c2_full_size_dense_equals_general(hip_ctx):
    """C2: 640x480, 64 levels, AdaptiveWeight r=5, both directions + cross-check."""
    W, H, D = 640, 480, 64
    (L, R, ml, mr), cams3, p, op = _setup(hip_ctx, W, H, D, 0x5EED0002, capi.WEIGHT_ADAPTIVE)
    out = {}
    # dense row-aligned kernels; candidate lists in row runs; lists in list order; one thread per pixel
    for mode, generic, rows in (("dense", 0, 1), ("rows", 1, 1), ("ordered", 1, 0), ("general", 2, 1)):
        hip_ctx.set_option("force_generic", generic)
        hip_ctx.set_option("list_rows", rows)
        dl, dr = hip_ctx.twoview_compute(0, 1, p)
        st = hip_ctx.stats()
        assert st["used_dense_path"] == (mode == "dense")
        out[mode] = (dl, dr, st["n_eval"], st["n_pixels"])
    hip_ctx.set_option("force_generic", 0)
    hip_ctx.set_option("list_rows", 1)
    for mode in ("rows", "ordered", "general"):
        assert _same_bits(out["dense"][0], out[mode][0]) and _same_bits(out["dense"][1], out[mode][1]), mode
        assert out["dense"][2] == out[mode][2] and out["dense"][3] == out[mode][3] == W * H   # counters of the last pass
    # sanity of the content: a good share of pixels survive the ratio test and the cross-check,
    # and surviving left depths reproduce the ground-truth disparity f*B/z for most of them
    dl = out["dense"][0]
    fin = np.isfinite(dl)
    assert fin.mean() > 0.2
    # one full-width row against the oracle (WTA stage)
    (Kl, Rl, tl), (Kr, Rr, tr) = cams3
    li, ri = O.OImage(L, ml), O.OImage(R, mr)
    y = H // 2
    want = O.twoview_wta(li, ri, O.camera_set(Kl, Rl, tl), O.camera_set(Kr, Rr, tr), op, y, y + 1)
    hip_ctx.twoview_wta(0, 1, p, y, y + 1)
    got = hip_ctx.download_depth(0)
    assert _same_bits(got[y], want[y]) or np.allclose(got[y], want[y], rtol=1e-9, equal_nan=True)


def test_c5_refractive_rows_equal_ordered_lists(hip_ctx):
    """C5 geometry (refractive interface, curved epipolar lines) at 960x540x128: the row-run evaluation and
    the list-order evaluation give the same bits, at two band budgets; one row agrees with the oracle."""
    W, H, D = 960, 540, 128
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0050)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333)
    hip_ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl, None, *plane))
    hip_ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr, None, *plane))
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
    res = {}
    for tag, rows, budget in (("rows", 1, 32768), ("rows_bands", 1, 96), ("ordered", 0, 32768)):
        hip_ctx.set_option("list_rows", rows)
        hip_ctx.set_option("band_budget_mb", budget)
        hip_ctx.twoview_wta(1, 0, p)
        res[tag] = hip_ctx.download_depth(1)
        assert not hip_ctx.stats()["used_dense_path"]
    hip_ctx.set_option("list_rows", 1)
    hip_ctx.set_option("band_budget_mb", 32768)
    assert _same_bits(res["rows"], res["rows_bands"]) and _same_bits(res["rows"], res["ordered"])
    assert np.isfinite(res["rows"]).mean() > 0.2
    op = O.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
    y = H // 2
    want = O.twoview_wta(O.OImage(R, mr), O.OImage(L, ml), O.camera_set(Kr, Rr, tr, None, *plane),
                         O.camera_set(Kl, Rl, tl, None, *plane), op, y, y + 1)
    assert np.allclose(res["rows"][y], want[y], rtol=1e-9, equal_nan=True)


def _oracle_rows(li, ri, cl, cr, op, rows):
    """One-row oracle WTA passes on a thread pool (the C oracle runs outside the GIL): {y: row of the map}."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, len(rows))) as ex:       # (the GPU box gives a one-GPU job 16 host threads)
        maps = list(ex.map(lambda y: O.twoview_wta(li, ri, cl, cr, op, y, y + 1)[y], rows))
    return dict(zip(rows, maps))


def _stratified_rows(H, R=5):
    """16 rows of a full-size image: the first and last row, the rows either side of where the window stops
    crossing the top / bottom border (R - 1 | R, H - R - 1 | H - R), the rows either side of the strip kernel's work-item
    boundaries (its 16- / 8- / 4-row segments: launch_twoview_strip_cost's n1, n2), the rest spread over the interior."""
    n1 = ((H * 3 // 4) // 16) * 16
    n2 = n1 + (((H - n1) * 2 // 3) // 8) * 8
    rows = [0, R - 1, R, H - R - 1, H - R, H - 1, 15, 16, n1 - 1, n1, n2 - 1, n2, H // 5, H // 3, H // 2 + 7, 2 * H // 3 + 1]
    assert len(set(rows)) == 16 and all(0 <= y < H for y in rows)
    return rows


def _assert_rows(got, want_rows, tag):
    for y, want in want_rows.items():
        ok, msg, _ = cases.compare_depth(got[y], want, 1e-9)
        assert ok, (tag, y, msg)


def test_c3_full_size_band_invariance(hip_ctx):
    """C3: 1920x1080, 256 levels, GeodesicWeight r=5: the depth map must not depend on the band
    split (one band at the default 32 GB budget vs ~50 bands at 128 MB), left->right pass.  Then, through the STRIP
    kernel: 16 stratified rows of both directions against the oracle (_stratified_rows: the border rows -- rows within R
    of the top / bottom take the select form and the phase-2 work lists --, the rows either side of the strip kernel's
    work-item boundaries, interior rows), and the whole left / right cross-check
    (twoviewstereo.cpp:596-672) against the oracle's pass over the same two maps."""
    W, H, D = 1920, 1080, 256
    (L, R, ml, mr), cams3, p, op = _setup(hip_ctx, W, H, D, 0x5EED0003, capi.WEIGHT_GEODESIC)
    hip_ctx.set_option("band_budget_mb", 32768)
    hip_ctx.twoview_wta(0, 1, p)
    a = hip_ctx.download_depth(0)
    st_a = hip_ctx.stats()
    assert st_a["used_strip_kernel"]                      # one band: the persistent strip kernel ...
    hip_ctx.set_option("band_budget_mb", 128)
    hip_ctx.twoview_wta(0, 1, p)
    b = hip_ctx.download_depth(0)
    st_b = hip_ctx.stats()
    assert not st_b["used_strip_kernel"]                  # ... ~50 thin bands: one workgroup per tile
    hip_ctx.set_option("band_budget_mb", 32768)
    assert st_a["used_dense_path"] and st_b["used_dense_path"]
    assert _same_bits(a, b)
    hip_ctx.twoview_wta(1, 0, p)
    ar = hip_ctx.download_depth(1)
    assert hip_ctx.stats()["used_strip_kernel"]
    (Kl, Rl, tl), (Kr, Rr, tr) = cams3
    li, ri, cl, cr = O.OImage(L, ml), O.OImage(R, mr), O.camera_set(Kl, Rl, tl), O.camera_set(Kr, Rr, tr)
    rows = _stratified_rows(H)
    _assert_rows(a, _oracle_rows(li, ri, cl, cr, op, rows), "left->right")
    _assert_rows(ar, _oracle_rows(ri, li, cr, cl, op, rows), "right->left")
    assert st_a["n_eval"] == st_b["n_eval"] and st_a["n_pixels"] == W * H
    # every reference pixel got a verdict: finite depth or +INF (ratio test); NaN only without candidates
    assert np.isnan(a).mean() < 0.01
    assert np.isfinite(a).mean() > 0.3
    # the order-dependent cross-check at full size: the oracle's two passes over the device's WTA maps
    want_l, want_r = O.twoview_cross_check(cl, cr, op, a, ar)
    hip_ctx.twoview_wta(0, 1, p)                          # (view 0 holds the 50-band map: the same bits, but be plain)
    hip_ctx.twoview_cross_check(0, 1, p)
    for got, want, tag in ((hip_ctx.download_depth(0), want_l, "left"), (hip_ctx.download_depth(1), want_r, "right")):
        ok, msg, _ = cases.compare_depth(got, want, 1e-9)
        assert ok, ("cross-check", tag, msg)
    assert (~np.isfinite(want_l[np.isfinite(a)])).any()   # the cross-check rejects something


def test_c4_like_mvs_two_stage_equals_inline_kernel(hip_ctx):
    """C4 geometry at 640x480x64, 4 views: walk -> list cost -> combine equals the one-thread-per-pixel kernel bit
    for bit, independently of the band split; one row of view 1 agrees with the oracle; cross-check chain runs."""
    W, H, D, NV = 640, 480, 64, 4
    cams3 = synthetic.semicircle_rig(NV, W, H, radius=10.0, step_deg=22.5, focal=float(W))
    # view 2 rolled by 35 degrees about its optical axis: the curves of its links run obliquely through the other image,
    # their window boxes do not fit the LDS share and those waves stay on the gathering cost kernel
    roll = np.deg2rad(35.0)
    Rz = np.array([[np.cos(roll), -np.sin(roll), 0.0], [np.sin(roll), np.cos(roll), 0.0], [0.0, 0.0, 1.0]])
    K2, R2, t2 = cams3[2]
    cams3[2] = (K2, Rz @ R2, Rz @ t2)
    rgba, masks, _ = synthetic.render_sphere_views(cams3, W, H, 0x5EED0004, sphere_radius=2.0, tex_size=512)
    cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
    kw = dict(min_depth=8.0, max_depth=12.0, num_depth_levels=D, cross_check_threshold=2 * 4.0 / (D - 1))
    p = capi.params_mvs(**kw)
    neigh = capi.mvs_neighbours(cams, p)
    for v in range(NV):
        hip_ctx.upload_view(v, rgba[v], masks[v], cams[v])
    res = {}
    for tag, generic, budget, staged, in_flight in (("two_stage", 0, 32768, 1, 0), ("two_stage_bands", 0, 64, 1, 0),
                                                   ("gathering", 0, 32768, 0, 0), ("inline", 1, 32768, 1, 0)):
        hip_ctx.set_option("force_generic", generic)
        hip_ctx.set_option("band_budget_mb", budget)
        hip_ctx.set_option("mvs_staged", staged)
        hip_ctx.set_option("mvs_async", in_flight)
        maps, evals = [], []
        for v in range(NV):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)
            maps.append(hip_ctx.download_depth(v))
            evals.append(hip_ctx.stats()["n_eval"])
        res[tag] = (maps, evals)
    hip_ctx.set_option("force_generic", 0)
    hip_ctx.set_option("band_budget_mb", 32768)
    hip_ctx.set_option("mvs_staged", 1)
    # the default: the calls only queue the views' kernels (two views in flight on two streams); the maps are complete
    # whenever they are looked at.  All four queued back to back, then read; the last call's counters are its own.
    hip_ctx.set_option("mvs_async", 1)
    for v in range(NV):
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
    assert hip_ctx.stats()["n_eval"] == res["two_stage"][1][NV - 1]
    for v in range(NV):
        assert _same_bits(res["two_stage"][0][v], hip_ctx.download_depth(v)), ("in flight", v)
    for v in range(NV):
        for tag in ("two_stage_bands", "gathering", "inline"):
            assert _same_bits(res["two_stage"][0][v], res[tag][0][v]), (tag, v)
            assert res["two_stage"][1][v] == res[tag][1][v], (tag, v)
        m = masks[v] == 1
        assert np.isposinf(res["two_stage"][0][v][~m]).all()          # masked-out pixels keep +INF
        assert (res["two_stage"][0][v][m] > 0).mean() > 0.5           # most sphere pixels have a >0.95 peak
    ocams = [O.camera_set(K, R, t) for (K, R, t) in cams3]
    op = O.params_mvs(**kw)
    imgs = [O.OImage(rgba[v], masks[v]) for v in range(NV)]
    y = H // 2
    want, n_eval = O.mvs_initial_estimate(imgs, ocams, 1, neigh[1], op, y, y + 1)
    assert np.allclose(res["two_stage"][0][1][y], want[y], rtol=1e-9, equal_nan=True)
    for v in range(NV):
        hip_ctx.mvs_cross_check(list(range(NV)), v, p)
    assert np.isnan(hip_ctx.download_depth(0)[masks[0] == 1]).any()   # the cross-check rejects something


def test_c4_full_size(hip_ctx):
    """C4 at BASELINE size: 8 views 1280x960, 128 uniform levels, r=2, 3 neighbours.  Two-stage kernels ==
    inline kernel bit for bit (also across a band split); 12 stratified full-width rows PER VIEW against the oracle; the
    ordered cross-check chain of all 8 views against the oracle's chain run on the same initial maps."""
    W, H, D, NV = 1280, 960, 128, 8
    cams3 = synthetic.semicircle_rig(NV, W, H, radius=10.0, step_deg=22.5, focal=float(W))
    rgba, masks, _ = synthetic.render_sphere_views(cams3, W, H, 0x5EED0004, sphere_radius=2.0, tex_size=1024)
    cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
    kw = dict(min_depth=8.0, max_depth=12.0, num_depth_levels=D, cross_check_threshold=2 * 4.0 / (D - 1))
    p = capi.params_mvs(**kw)
    neigh = capi.mvs_neighbours(cams, p)
    assert all(len(n) == 3 for n in neigh)
    for v in range(NV):
        hip_ctx.upload_view(v, rgba[v], masks[v], cams[v])
    res = {}
    for tag, generic, budget in (("two_stage", 0, 32768), ("two_stage_bands", 0, 256), ("inline", 1, 32768)):
        hip_ctx.set_option("force_generic", generic)
        hip_ctx.set_option("band_budget_mb", budget)
        maps, evals = [], []
        for v in range(NV):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)
            maps.append(hip_ctx.download_depth(v))
            evals.append(hip_ctx.stats()["n_eval"])
        res[tag] = (maps, evals)
    hip_ctx.set_option("force_generic", 0)
    hip_ctx.set_option("band_budget_mb", 32768)
    for v in range(NV):
        for tag in ("two_stage_bands", "inline"):
            assert _same_bits(res["two_stage"][0][v], res[tag][0][v]), (tag, v)
            assert res["two_stage"][1][v] == res[tag][1][v], (tag, v)
    ocams = [O.camera_set(K, R, t) for (K, R, t) in cams3]
    op = O.params_mvs(**kw)
    imgs = [O.OImage(rgba[v], masks[v]) for v in range(NV)]
    # 12 stratified full-width rows PER VIEW against the oracle (round 6: one row before): the image's first and last rows
    # and the rows where the r = 2 window stops crossing them, rows through the sphere's silhouette and its centre
    from concurrent.futures import ThreadPoolExecutor
    rows = [0, 1, 2, 3, H // 4 - 3, H // 3 + 1, H // 2 - 40, H // 2 + 7, 2 * H // 3, 3 * H // 4 + 5, H - 3, H - 1]
    jobs = [(v, y) for v in range(NV) for y in rows]
    with ThreadPoolExecutor(max_workers=16) as ex:
        wants = list(ex.map(lambda j: O.mvs_initial_estimate(imgs, ocams, j[0], neigh[j[0]], op, j[1], j[1] + 1)[0][j[1]], jobs))
    for (v, y), want in zip(jobs, wants):
        ok, msg, _ = _cmp(res["two_stage"][0][v][y], want)
        assert ok, (v, y, msg)
    # cross-check chain (multiviewstereo.cpp:427-431): the oracle starts from the device's initial maps
    work = [m.copy() for m in res["two_stage"][0]]
    for v in range(NV):
        O.mvs_cross_check(imgs, ocams, v, op, work)
    for v in range(NV):                                              # the maps of the last mode ("inline") are resident
        hip_ctx.mvs_cross_check(list(range(NV)), v, p)
    for v in range(NV):
        ok, msg, _ = _cmp(hip_ctx.download_depth(v), work[v])
        assert ok, (v, msg)
    assert np.isnan(work[0][masks[0] == 1]).any()


def _cmp(got, want):
    import cases
    return cases.compare_depth(got, want, 1e-9)


@pytest.mark.parametrize("seed", [0x5EED0050, 0x5EED0057])
def test_c5_full_size(hip_ctx, seed):
    """C5 at BASELINE size (1920x1080, 256 levels, refractive interface), first and last pair seed of SURVEY 8(d):
    row-run evaluation == list-order evaluation bit for bit (both directions), n_eval equal; 16 stratified full-width
    rows per direction against the oracle."""
    W, H, D = 1920, 1080, 256
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, seed)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333)
    hip_ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl, None, *plane))
    hip_ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr, None, *plane))
    kw = dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
    p = capi.params_twoview(**kw)
    res = {}
    for tag, rows in (("rows", 1), ("ordered", 0)):
        hip_ctx.set_option("list_rows", rows)
        out = []
        for ref, oth in ((0, 1), (1, 0)):
            hip_ctx.twoview_wta(ref, oth, p)
            st = hip_ctx.stats()
            assert not st["used_dense_path"]
            out.append((hip_ctx.download_depth(ref), st["n_eval"]))
        res[tag] = out
    hip_ctx.set_option("list_rows", 1)
    for k in range(2):
        assert _same_bits(res["rows"][k][0], res["ordered"][k][0]), k
        assert res["rows"][k][1] == res["ordered"][k][1]
        assert np.isfinite(res["rows"][k][0]).mean() > 0.2
    op = O.params_twoview(**kw)
    oc = [O.camera_set(Kl, Rl, tl, None, *plane), O.camera_set(Kr, Rr, tr, None, *plane)]
    oi = [O.OImage(L, ml), O.OImage(R, mr)]
    # 16 stratified rows per direction: border rows (windows cut by the image's top / bottom: the blocked select form), the
    # rows where that stops, interior rows
    for ref, oth in ((0, 1), (1, 0)):
        _assert_rows(res["rows"][ref][0], _oracle_rows(oi[ref], oi[oth], oc[ref], oc[oth], op, _stratified_rows(H)), (hex(seed), ref))


def test_c5_full_size_tilted_interface(hip_ctx):
    """C5 geometry at BASELINE size with the refractive interface TILTED against the optical axis (normal not (0,0,1):
    the axis on which the reference's y-only side test of projectRefraction, camera.cpp:119-135, is harmless): one
    full-width row per direction against the oracle, row-run evaluation == list-order evaluation for the left view."""
    W, H, D = 1920, 1080, 256
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0053)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    n = np.array([0.12, -0.07, 1.0]); n /= np.linalg.norm(n)
    plane = (n, 0.1, 1.333)
    hip_ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl, None, *plane))
    hip_ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr, None, *plane))
    kw = dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
    p = capi.params_twoview(**kw)
    maps = []
    for ref, oth in ((0, 1), (1, 0)):
        hip_ctx.twoview_wta(ref, oth, p)
        assert not hip_ctx.stats()["used_dense_path"]
        maps.append(hip_ctx.download_depth(ref))
        assert np.isfinite(maps[-1]).mean() > 0.2
    hip_ctx.set_option("list_rows", 0)
    hip_ctx.twoview_wta(0, 1, p)
    hip_ctx.set_option("list_rows", 1)
    assert _same_bits(maps[0], hip_ctx.download_depth(0))
    op = O.params_twoview(**kw)
    oc = [O.camera_set(Kl, Rl, tl, None, *plane), O.camera_set(Kr, Rr, tr, None, *plane)]
    oi = [O.OImage(L, ml), O.OImage(R, mr)]
    for ref, oth, y in ((0, 1, 2 * H // 3), (1, 0, H // 4)):
        want = O.twoview_wta(oi[ref], oi[oth], oc[ref], oc[oth], op, y, y + 1)
        ok, msg, _ = _cmp(maps[ref][y], want[y])
        assert ok, (ref, y, msg)


@pytest.mark.parametrize("name", ["mvs_distorted", "mvs_refractive", "mvs_scaled"])
def test_mvs_list_path_medium_size_other_camera_models(hip_ctx, name):
    """The list path's machinery -- serpentine units, aligned lists, window boxes, staged cost kernel, views in flight --
    with lens distortion, a refractive interface and an image scale at 320x240 (the small parity cases are mostly
    image border): default == gathering kernel == inline kernel bit for bit, two rows of view 1 against the oracle."""
    case = cases.get_mvs(name, w=320, h=240, D=40, nviews=3)
    cams, p = cases.hip_inputs(case)
    neigh = [list(map(int, n)) for n in capi.mvs_neighbours(cams, p)]
    cases.upload_case(hip_ctx, case, cams)
    nv = len(cams)
    res = {}
    for tag, opts in (("default", {}), ("gathering", {"mvs_staged": 0, "mvs_async": 0}), ("inline", {"force_generic": 1})):
        for k, val in opts.items():
            hip_ctx.set_option(k, val)
        try:
            for v in range(nv):
                hip_ctx.mvs_initial_estimate(v, neigh[v], p)
            res[tag] = [hip_ctx.download_depth(v) for v in range(nv)]
        finally:
            hip_ctx.set_option("mvs_staged", 1)
            hip_ctx.set_option("mvs_async", 1)
            hip_ctx.set_option("force_generic", 0)
    for v in range(nv):
        for tag in ("gathering", "inline"):
            assert _same_bits(res["default"][v], res[tag][v]), (name, tag, v)
    imgs, ocams, op = cases.oracle_inputs(case)
    y = case["views"][1][1].shape[0] // 2
    want, _ = O.mvs_initial_estimate(imgs, ocams, 1, neigh[1], op, y, y + 2)
    ok, msg, _ = cases.compare_depth(res["default"][1][y:y + 2], want[y:y + 2], 1e-9)
    assert ok, (name, msg)
    assert (res["default"][1][case["views"][1][1] == 1] > 0).any()


def _rot(axis, ang):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


@pytest.mark.parametrize("seed", range(8))
def test_mvs_list_path_random_rigs_and_ragged_masks(hip_ctx, seed):
    """Random rigs around the sphere -- cameras rolled about their axes by up to 90 degrees and tilted, so that epipolar
    curves run in every direction through the other image --, masks with random holes and ragged rows, odd image sizes:
    the list path (serpentine units, aligned lists, window boxes, LDS-staged and gathering cost kernels, two views in
    flight) gives the inline one-thread-per-pixel kernel's maps bit for bit, with the same evaluation counts."""
    rng = np.random.default_rng(0xC0FFEE + seed)
    W, H, D, NV = int(rng.integers(70, 200)), int(rng.integers(50, 140)), int(rng.integers(12, 40)), 3
    cams3 = synthetic.semicircle_rig(NV, W, H, radius=10.0, step_deg=float(rng.uniform(6.0, 25.0)), focal=float(rng.uniform(0.9, 1.6)) * W)
    for v in range(NV):
        K, R, t = cams3[v]
        Q = _rot([0, 0, 1], rng.uniform(-np.pi / 2, np.pi / 2)) @ _rot(rng.normal(size=3), rng.uniform(0, 0.08))
        cams3[v] = (K, Q @ R, Q @ t)                         # the camera turned about its own centre
    rgba, masks, _ = synthetic.render_sphere_views(cams3, W, H, 0xABCD00 + seed, sphere_radius=2.0, tex_size=256)
    for v in range(NV):
        holes = rng.random((H, W)) < 0.03                    # isolated masked-out pixels
        masks[v] = np.where(holes, 0, masks[v]).astype(np.uint8)
        y = int(rng.integers(0, H))
        masks[v][y, : W // 2] = 0                            # a ragged row
        if seed % 3 == 0:
            masks[v][:] = 1                                  # every pixel a unit (also those that see no sphere)
    cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
    # (peak thresholds: the default, zero, negative -- scores of exactly 0 and negative scores then count -- and a high one)
    thr = (None, 0.0, -0.5, 0.9)[seed % 4]
    kw = dict(min_depth=7.5, max_depth=12.5, num_depth_levels=D, cross_check_threshold=0.3)
    if thr is not None:
        kw["peak_threshold"] = thr
    p = capi.params_mvs(**kw)
    neigh = [list(map(int, n)) for n in capi.mvs_neighbours(cams, p)]
    for v in range(NV):
        hip_ctx.upload_view(v, rgba[v], masks[v], cams[v])
    out = {}
    for tag, opts in (("default", {}), ("gathering", {"mvs_staged": 0}), ("inline", {"force_generic": 1})):
        for k, val in opts.items():
            hip_ctx.set_option(k, val)
        try:
            maps, evals = [], []
            for v in range(NV):
                hip_ctx.mvs_initial_estimate(v, neigh[v], p)
            for v in range(NV):
                maps.append(hip_ctx.download_depth(v))
            for v in range(NV):                              # (counters: one view at a time)
                hip_ctx.mvs_initial_estimate(v, neigh[v], p)
                evals.append(hip_ctx.stats()["n_eval"])
            out[tag] = (maps, evals)
        finally:
            hip_ctx.set_option("mvs_staged", 1)
            hip_ctx.set_option("force_generic", 0)
    for v in range(NV):
        for tag in ("gathering", "inline"):
            assert _same_bits(out["default"][0][v], out[tag][0][v]), (seed, tag, v, W, H, D)
            assert out["default"][1][v] == out[tag][1][v], (seed, tag, v)
        assert np.isposinf(out["default"][0][v][masks[v] != 1]).all()
    assert sum(out["default"][1]) > 20000, "degenerate rig: hardly any candidate"
    # the top-K request (what the MRF branch consumes): staged + gathering list kernels against the inline kernel, view 0
    import torch
    pks = []
    for generic in (0, 1):
        pk = torch.full((H, W, p.top_k, 2), 7.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        hip_ctx.set_option("force_generic", generic)
        try:
            hip_ctx.mvs_initial_estimate(0, neigh[0], p, peaks_dev=pk.data_ptr())
            hip_ctx.synchronize()
        finally:
            hip_ctx.set_option("force_generic", 0)
        pks.append(pk.cpu().numpy())
    assert np.array_equal(pks[0].view(np.uint64), pks[1].view(np.uint64)), (seed, "top-K lists")
    assert thr == 0.9 or any((m[np.isfinite(m)] > 0).any() for m in out["default"][0]), "no NCC peak anywhere"
