"""Small deterministic scenes shared by the parity tests, smoke() and the golden-fixture script.

Each case is a dict: views = [(rgba, mask, (K, R, t), dist, plane)] and the
srh/sro parameter overrides.  Only numpy here: the same arrays are handed to the
oracle (tests/oracle_ffi.py) and to the HIP library (stereoreconstruction_amd.capi).
"""
import numpy as np

from stereoreconstruction_amd import synthetic as S


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def twoview_case(name, w=64, h=40, D=16, seed=0x5EED0A00, weight_kind=1, radius=5,
                 masks=False, distortion=False, verged=False, refractive=False, scale=1.0):
    """A two-view scene on the synthetic rectified pair, optionally perturbed."""
    L, R, ml, mr, disp = S.rectified_pair(w, h, D, seed)
    (Kl, Rl, tl), (Kr, Rr, tr) = S.rectified_cameras(w, h)
    zmin, zmax = S.rectified_depth_range(w, D)
    if scale != 1.0:
        # the images handed over are ALREADY scaled: cameras describe the full-size image
        Kl = Kl.copy(); Kr = Kr.copy()
        Kl[:2] /= scale; Kr[:2] /= scale
    if verged:
        # toe the right camera in and roll it a little: sloped, non-parallel epipolar lines
        Rr = _rot_z(0.05) @ _rot_x(0.02) @ _rot_y(-0.04)
        C = np.array([1.0, 0.03, 0.02])
        tr = -Rr @ C
        Rl = _rot_y(0.03)
        tl = -Rl @ np.zeros(3)
    dist_l = dist_r = None
    if distortion:
        dist_l = np.array([-0.131, 0.4, 0.004, 0.003, -0.6])
        dist_r = np.array([-0.058, -0.2, 0.0, 0.006, 0.3])
    plane = None
    if refractive:
        plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333)
    if masks:
        yy, xx = np.mgrid[0:h, 0:w]
        ml = (((xx - w * 0.45) ** 2 / (w * 0.42) ** 2 + (yy - h * 0.5) ** 2 / (h * 0.45) ** 2) < 1).astype(np.uint8)
        mr = (((xx - w * 0.40) ** 2 / (w * 0.45) ** 2 + (yy - h * 0.5) ** 2 / (h * 0.47) ** 2) < 1).astype(np.uint8)
        ml[h // 3, w // 3:w // 3 + 5] = 0          # a hole inside the object
    params = dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, window_radius=radius,
                  weight_kind=weight_kind, image_scale=scale)
    views = [(L, ml, (Kl, Rl, tl), dist_l, plane), (R, mr, (Kr, Rr, tr), dist_r, plane)]
    return dict(name=name, kind="twoview", views=views, params=params, gt_disparity=disp)


TWOVIEW_CASES = {
    "adaptive_rect": dict(weight_kind=0),
    "geodesic_rect": dict(weight_kind=1),
    "geodesic_r2": dict(weight_kind=1, radius=2, w=48, h=36),
    "adaptive_masks": dict(weight_kind=0, masks=True),
    "geodesic_masks": dict(weight_kind=1, masks=True),
    "geodesic_distorted": dict(weight_kind=1, distortion=True, radius=3),
    "adaptive_verged": dict(weight_kind=0, verged=True, radius=3),
    "geodesic_verged_dist_masks": dict(weight_kind=1, verged=True, distortion=True, masks=True, radius=2),
    "adaptive_refractive": dict(weight_kind=0, refractive=True, radius=2),
    "geodesic_scaled": dict(weight_kind=1, scale=0.5, radius=2),
}


def get_twoview(name, **over):
    kw = dict(TWOVIEW_CASES[name])
    kw.update(over)
    return twoview_case(name, **kw)


def mvs_case(name="mvs_sphere", nviews=4, w=56, h=40, D=24, seed=0x5EED0B00, weight_kind=1, radius=2,
             distortion=False, step_deg=12.0, scale=1.0, refractive=False, mixed_sizes=False):
    sizes = [(w, h)] * nviews
    if mixed_sizes:                                      # every view its own raster (and calibration)
        sizes = [(w - 8 * (v % 3), h - 4 * (v % 2)) for v in range(nviews)]
    views, depth = [], []
    for v in range(nviews):
        wv, hv = sizes[v]
        cams = S.semicircle_rig(nviews, wv, hv, radius=10.0, step_deg=step_deg, focal=1.4 * wv)
        rgba, masks, dep = S.render_sphere_views([cams[v]], wv, hv, seed, sphere_radius=2.0, tex_size=256)
        K, R, t = cams[v]
        if scale != 1.0:
            # the images handed over are ALREADY scaled: cameras describe the full-size image
            K = K.copy()
            K[:2] /= scale
        dist = None
        if distortion:
            dist = np.array([-0.1 + 0.01 * v, 0.2, 0.002, -0.001 * v, 0.0])
        plane = (np.array([0.01 * v, -0.02, 1.0]), 0.1, 1.333) if refractive else None
        views.append((rgba[0], masks[0], (K, R, t), dist, plane))
        depth.append(dep[0])
    zmin, zmax = 7.5, 10.5
    params = dict(min_depth=zmin, max_depth=zmax, num_depth_levels=D, window_radius=radius,
                  weight_kind=weight_kind, image_scale=scale,
                  cross_check_threshold=2.0 * (zmax - zmin) / (D - 1))
    return dict(name=name, kind="mvs", views=views, params=params, gt_depth=depth)


MVS_CASES = {
    "mvs_geodesic": dict(weight_kind=1),
    "mvs_adaptive": dict(weight_kind=0),
    "mvs_distorted": dict(weight_kind=1, distortion=True, nviews=3),
    "mvs_five_views": dict(weight_kind=1, nviews=5, w=48, h=32, D=16),
    "mvs_mixed_sizes": dict(weight_kind=1, nviews=3, mixed_sizes=True, D=16),
    "mvs_scaled": dict(weight_kind=0, nviews=3, scale=0.5, w=48, h=36, D=16),
    "mvs_refractive": dict(weight_kind=1, nviews=3, refractive=True, w=48, h=32, D=14),
}


def get_mvs(name, **over):
    kw = dict(MVS_CASES[name])
    kw.update(over)
    return mvs_case(name, **kw)


# ---------------------------------------------------------------- adapters

def oracle_inputs(case):
    """-> (list[OImage], list[sro_camera], sro_params)"""
    import oracle_ffi as O
    imgs, cams = [], []
    for (rgba, mask, (K, R, t), dist, plane) in case["views"]:
        imgs.append(O.OImage(rgba, mask))
        if plane is None:
            cams.append(O.camera_set(K, R, t, dist))
        else:
            cams.append(O.camera_set(K, R, t, dist, plane[0], plane[1], plane[2]))
    mk = O.params_twoview if case["kind"] == "twoview" else O.params_mvs
    return imgs, cams, mk(**case["params"])


def hip_inputs(case):
    """-> (list[srh_camera], srh_params); images are uploaded by the caller."""
    from stereoreconstruction_amd import capi
    cams = []
    for (rgba, mask, (K, R, t), dist, plane) in case["views"]:
        if plane is None:
            cams.append(capi.camera_from_krt(K, R, t, dist))
        else:
            cams.append(capi.camera_from_krt(K, R, t, dist, plane[0], plane[1], plane[2]))
    mk = capi.params_twoview if case["kind"] == "twoview" else capi.params_mvs
    return cams, mk(**case["params"])


def upload_case(ctx, case, cams):
    for slot, (rgba, mask, _, _, _) in enumerate(case["views"]):
        ctx.upload_view(slot, rgba, mask, cams[slot])


def compare_depth(got, want, rtol=1e-9):
    """Parity rule of SURVEY 8(d): identical finite/+INF/NaN/-1 classes; finite depths within
    rtol*max(1,|z|).  Returns (ok, message, n_bad)."""
    got = np.asarray(got); want = np.asarray(want)
    assert got.shape == want.shape
    cls_g = np.where(np.isnan(got), 0, np.where(np.isposinf(got), 1, np.where(np.isneginf(got), 2, 3)))
    cls_w = np.where(np.isnan(want), 0, np.where(np.isposinf(want), 1, np.where(np.isneginf(want), 2, 3)))
    bad_cls = cls_g != cls_w
    fin = (cls_g == 3) & (cls_w == 3)
    tol = rtol * np.maximum(1.0, np.abs(want[fin]))
    bad_val = np.zeros(got.shape, dtype=bool)
    bad_val[fin] = np.abs(got[fin] - want[fin]) > tol
    n_bad = int(bad_cls.sum() + bad_val.sum())
    msg = "class mismatches %d, value mismatches %d of %d pixels (finite %d)" % (
        bad_cls.sum(), bad_val.sum(), got.size, fin.sum())
    if n_bad:
        idx = np.argwhere(bad_cls | bad_val)[:5]
        msg += "; first: " + ", ".join("(%d,%d): got %r want %r" % (y, x, got[y, x], want[y, x]) for y, x in idx)
    return n_bad == 0, msg, n_bad
