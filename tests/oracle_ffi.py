"""ctypes bindings for oracle/liboracle.so and oracle/_ref/libref_pieces.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under stereoreconstruction_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_pieces.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Camera(C.Structure):
    _fields_ = [
        ("K", C.c_double * 9), ("Kinv", C.c_double * 9), ("R", C.c_double * 9), ("Rinv", C.c_double * 9),
        ("t", C.c_double * 3), ("C", C.c_double * 3),
        ("dist", C.c_double * 5),
        ("is_distorted", C.c_int32), ("is_refractive", C.c_int32),
        ("plane_normal", C.c_double * 3), ("plane_dist", C.c_double), ("refr_index", C.c_double),
        ("pdir", C.c_double * 3),
    ]


class Image(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("rgba", c_uint8_p), ("mask", c_uint8_p)]


class Params(C.Structure):
    _fields_ = [
        ("min_depth", C.c_double), ("max_depth", C.c_double),
        ("num_depth_levels", C.c_int32), ("window_radius", C.c_int32),
        ("image_scale", C.c_double),
        ("weight_kind", C.c_int32), ("geodesic_iters", C.c_int32),
        ("geodesic_sigma", C.c_double), ("geodesic_init", C.c_double),
        ("adaptive_color_sigma", C.c_double), ("weight_cutoff", C.c_double),
        ("bad_ret", C.c_double), ("max_color_diff", C.c_double),
        ("second_best_factor", C.c_double), ("wta_margin", C.c_double),
        ("inconsistency_thresh", C.c_double),
        ("peak_threshold", C.c_double), ("cross_check_threshold", C.c_double),
        ("neighbour_min_dot", C.c_double),
        ("top_k", C.c_int32), ("num_neighbours", C.c_int32),
    ]


class Diag(C.Structure):
    _fields_ = [("win_xy", c_int32_p), ("min_cost", c_double_p), ("second_cost", c_double_p),
                ("n_eval", C.c_int64)]


WEIGHT_ADAPTIVE, WEIGHT_GEODESIC = 0, 1

_lib = None
_ref = None


def build_oracle(force=False):
    """Compile oracle/liboracle.so (gcc, a second or two)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                                               for f in ("sr_oracle.c", "sr_oracle.h")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(ORACLE_SO)
    L.sro_params_twoview_defaults.argtypes = [C.POINTER(Params)]
    L.sro_params_mvs_defaults.argtypes = [C.POINTER(Params)]
    L.sro_camera_set.argtypes = [C.POINTER(Camera), c_double_p, c_double_p, c_double_p, c_double_p,
                                 c_double_p, C.c_double, C.c_double]
    L.sro_image_sample.argtypes = [C.POINTER(Image), C.c_double, C.c_double, c_double_p]
    L.sro_image_sample.restype = C.c_int
    L.sro_to_gray.argtypes = [C.c_double] * 3
    L.sro_to_gray.restype = C.c_double
    L.sro_line_points.argtypes = [C.c_double] * 4 + [C.c_int] * 3 + [c_int32_p, C.c_int]
    L.sro_line_points.restype = C.c_int
    L.sro_weights.argtypes = [C.POINTER(Image), C.c_int, C.c_int, C.POINTER(Params), c_double_p]
    L.sro_unproject.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, c_double_p, c_double_p]
    L.sro_project.argtypes = [C.POINTER(Camera), c_double_p]
    L.sro_project.restype = C.c_int
    L.sro_closest_points.argtypes = [c_double_p] * 6
    L.sro_epipolar_preview.argtypes = [C.POINTER(Camera), C.POINTER(Camera), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p, C.c_int]
    L.sro_epipolar_preview.restype = C.c_int
    L.sro_refraction_pair_error.argtypes = [C.POINTER(Camera), C.POINTER(Camera), c_double_p, c_double_p]
    L.sro_refraction_pair_error.restype = C.c_double
    L.sro_mrf_params_defaults.argtypes = [C.POINTER(MrfParams)]
    L.sro_mrf_data_cost.argtypes = [C.POINTER(MrfParams), C.c_int, c_double_p, C.c_int]
    L.sro_mrf_data_cost.restype = C.c_double
    L.sro_mrf_smooth_cost.argtypes = [C.POINTER(MrfParams), C.c_int, c_double_p, c_double_p, C.c_int, C.c_int]
    L.sro_mrf_smooth_cost.restype = C.c_double
    L.sro_mvs_mrf.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, C.POINTER(C.c_uint8), C.POINTER(MrfParams),
                              c_double_p, c_int32_p, c_double_p, c_double_p, C.POINTER(MrfInfo)]
    L.sro_back_project.argtypes = [C.POINTER(Camera), C.POINTER(Params), C.c_int, C.c_int, C.c_double, c_double_p]
    L.sro_back_project.restype = C.c_int
    L.sro_epipolar_curve.argtypes = [C.POINTER(Camera), C.POINTER(Camera), C.POINTER(Image),
                                     C.POINTER(Params), C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int]
    L.sro_epipolar_curve.restype = C.c_int
    for f in (L.sro_twoview_cost_ncc, L.sro_mvs_cost_ncc):
        f.argtypes = [C.POINTER(Image), C.POINTER(Image), c_double_p, C.POINTER(Params)] + [C.c_int] * 4
        f.restype = C.c_double
    L.sro_twoview_wta.argtypes = [C.POINTER(Image), C.POINTER(Image), C.POINTER(Camera), C.POINTER(Camera),
                                  C.POINTER(Params), C.c_int, C.c_int, c_double_p, C.POINTER(Diag)]
    L.sro_twoview_cross_check.argtypes = [C.c_int, C.c_int, C.POINTER(Camera), C.POINTER(Camera),
                                          C.POINTER(Params), c_double_p, c_double_p]
    L.sro_mvs_neighbours.argtypes = [C.c_int, C.POINTER(Camera), C.POINTER(Params), c_int32_p, c_int32_p]
    L.sro_mvs_initial_estimate.argtypes = [C.c_int, C.POINTER(Image), C.POINTER(Camera), C.c_int, c_int32_p,
                                           C.c_int, C.POINTER(Params), C.c_int, C.c_int, c_double_p, c_double_p,
                                           C.POINTER(C.c_int64)]
    L.sro_mvs_cross_check.argtypes = [C.c_int, C.POINTER(Image), C.POINTER(Camera), C.c_int,
                                      C.POINTER(Params), C.POINTER(c_double_p)]
    _lib = L
    return L


def ref_available():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own compiled pieces (container + GPU box: travels as a built .so)."""
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(REF_SO)
    L.refp_image_create.argtypes = [c_uint8_p, C.c_int, C.c_int]
    L.refp_image_create.restype = C.c_void_p
    L.refp_image_free.argtypes = [C.c_void_p]
    L.refp_image_pixel.argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p]
    L.refp_image_pixel.restype = C.c_int
    L.refp_image_sample.argtypes = [C.c_void_p, C.c_double, C.c_double, c_double_p]
    L.refp_image_sample.restype = C.c_int
    L.refp_to_gray.argtypes = [C.c_double] * 3
    L.refp_to_gray.restype = C.c_double
    L.refp_pixel_is_white.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.refp_pixel_is_white.restype = C.c_int
    L.refp_line_points.argtypes = [C.c_double] * 4 + [C.c_int] * 3 + [c_int32_p, C.c_int]
    L.refp_line_points.restype = C.c_int
    L.refp_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p]
    _ref = L
    return L


# ---------------------------------------------------------------- helpers

def dptr(a):
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    return a.ctypes.data_as(c_int32_p)


def u8ptr(a):
    return a.ctypes.data_as(c_uint8_p)


class OImage:
    """Keeps the numpy buffers alive next to the C struct."""

    def __init__(self, rgba, mask=None):
        self.rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert self.rgba.ndim == 3 and self.rgba.shape[2] == 4
        h, w = self.rgba.shape[:2]
        self.mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.c = Image(w, h, u8ptr(self.rgba), u8ptr(self.mask) if self.mask is not None else None)
        self.w, self.h = w, h


def params_twoview(**kw):
    p = Params()
    lib().sro_params_twoview_defaults(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def params_mvs(**kw):
    p = Params()
    lib().sro_params_mvs_defaults(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def camera_set(K, R, t, dist=None, plane_normal=None, plane_dist=0.0, refr_index=1.0):
    cam = Camera()
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    d = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
    n = None if plane_normal is None else np.ascontiguousarray(plane_normal, dtype=np.float64).reshape(3)
    lib().sro_camera_set(C.byref(cam), dptr(K), dptr(R), dptr(t),
                         dptr(d) if d is not None else None,
                         dptr(n) if n is not None else None, plane_dist, refr_index)
    return cam


def camera_set_p(P, dist=None, plane_normal=None, plane_dist=0.0, refr_index=1.0):
    cam = Camera()
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(12)
    d = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
    n = None if plane_normal is None else np.ascontiguousarray(plane_normal, dtype=np.float64).reshape(3)
    lib().sro_camera_set_p(C.byref(cam), dptr(P), dptr(d) if d is not None else None,
                           dptr(n) if n is not None else None, C.c_double(plane_dist), C.c_double(refr_index))
    return cam


def weights(img, cx, cy, p):
    ws = 2 * p.window_radius + 1
    out = np.empty((ws, ws), dtype=np.float64)
    lib().sro_weights(C.byref(img.c), cx, cy, C.byref(p), dptr(out))
    return out


def line_points(x0, y0, x1, y1, clip=False, w=0, h=0, cap=1 << 16, bounded=False):
    """clip: 6-arg LineIterator; bounded: the 4-arg walk in the jump-ahead form restricted to [0,w)x[0,h)
    that sro_epipolar_curve uses for TwoView curves (sr_oracle.c line_walk with bounds)."""
    out = np.empty((cap, 2), dtype=np.int32)
    n = lib().sro_line_points(x0, y0, x1, y1, 2 if bounded else int(clip), w, h, iptr(out), cap)
    return out[:min(n, cap)].copy()


class MrfParams(C.Structure):
    _fields_ = [("beta", C.c_double), ("lambda_", C.c_double), ("phi_u", C.c_double), ("psi_u", C.c_double),
                ("max_iters", C.c_int32), ("min_energy_drop", C.c_double)]


class MrfInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("energy_initial", C.c_double), ("energy_final", C.c_double),
                ("lower_bound", C.c_double)]


def mrf_params(**over):
    m = MrfParams()
    lib().sro_mrf_params_defaults(C.byref(m))
    for k, v in over.items():
        setattr(m, "lambda_" if k == "lambda" else k, v)
    return m


def mvs_mrf(peaks, mask, m, depth=None, data_costs=None, want_messages=False):
    """peaks (h,w,K,2) -> dict(depth, labels, info, messages)"""
    peaks = np.ascontiguousarray(peaks, dtype=np.float64)
    h, w, K, _ = peaks.shape
    depth = np.full((h, w), np.inf) if depth is None else np.ascontiguousarray(depth, dtype=np.float64).copy()
    labels = np.zeros((h, w), dtype=np.int32)
    msgs = np.zeros((h, w, 2, K + 1)) if want_messages else None
    info = MrfInfo()
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    dc = None if data_costs is None else np.ascontiguousarray(data_costs, dtype=np.float64)
    lib().sro_mvs_mrf(w, h, K, dptr(peaks), None if mk is None else mk.ctypes.data_as(C.POINTER(C.c_uint8)),
                      C.byref(m), dptr(depth), iptr(labels), None if dc is None else dptr(dc),
                      None if msgs is None else dptr(msgs), C.byref(info))
    return dict(depth=depth, labels=labels, messages=msgs, iterations=info.iterations,
                energy_initial=info.energy_initial, energy_final=info.energy_final, lower_bound=info.lower_bound)


def mrf_energy(peaks, labels, m, data_costs=None):
    """totalEnergy of a labelling, term by term through sro_mrf_data_cost / sro_mrf_smooth_cost (python loops: small grids)."""
    L = lib()
    peaks = np.ascontiguousarray(peaks, dtype=np.float64)
    h, w, K, _ = peaks.shape
    e = 0.0
    for y in range(h):
        for x in range(w):
            pk = peaks[y, x]
            e += (data_costs[y, x, labels[y, x]] if data_costs is not None
                  else L.sro_mrf_data_cost(C.byref(m), K, dptr(pk), int(labels[y, x])))
            if x + 1 < w:
                e += L.sro_mrf_smooth_cost(C.byref(m), K, dptr(pk), dptr(peaks[y, x + 1]), int(labels[y, x]), int(labels[y, x + 1]))
            if y + 1 < h:
                e += L.sro_mrf_smooth_cost(C.byref(m), K, dptr(pk), dptr(peaks[y + 1, x]), int(labels[y, x]), int(labels[y + 1, x]))
    return e


def epipolar_preview(refcam, othcam, px, py, min_depth, max_depth, num_depths):
    """StereoWidget::epipolarLineItem: the vertices (k,2) of the previewed path (k == 0: nothing drawn)."""
    out = np.empty((num_depths, 2), dtype=np.float64)
    n = lib().sro_epipolar_preview(C.byref(refcam), C.byref(othcam), px, py, min_depth, max_depth, num_depths,
                                   dptr(out), num_depths)
    return out[:n].copy()


def refraction_pair_error(cam1, cam2, p1, p2):
    a = np.ascontiguousarray(p1, dtype=np.float64)
    b = np.ascontiguousarray(p2, dtype=np.float64)
    return lib().sro_refraction_pair_error(C.byref(cam1), C.byref(cam2), dptr(a), dptr(b))


def epipolar_curve(refcam, othcam, oth, p, mvs, x, y, cap=1 << 16):
    out = np.empty((cap, 2), dtype=np.int32)
    n = lib().sro_epipolar_curve(C.byref(refcam), C.byref(othcam), C.byref(oth.c), C.byref(p),
                                 int(mvs), x, y, iptr(out), cap)
    assert n <= cap
    return out[:n].copy()


def twoview_wta(ref_img, oth_img, refcam, othcam, p, y0=0, y1=None, want_diag=False):
    w, h = ref_img.w, ref_img.h
    y1 = h if y1 is None else y1
    depth = np.full((h, w), np.nan, dtype=np.float64)
    if want_diag:
        win = np.full((h, w, 2), -1, dtype=np.int32)
        mc = np.full((h, w), np.inf)
        sc = np.full((h, w), np.inf)
        d = Diag(iptr(win), dptr(mc), dptr(sc), 0)
        lib().sro_twoview_wta(C.byref(ref_img.c), C.byref(oth_img.c), C.byref(refcam), C.byref(othcam),
                              C.byref(p), y0, y1, dptr(depth), C.byref(d))
        return depth, dict(win_xy=win, min_cost=mc, second_cost=sc, n_eval=d.n_eval)
    lib().sro_twoview_wta(C.byref(ref_img.c), C.byref(oth_img.c), C.byref(refcam), C.byref(othcam),
                          C.byref(p), y0, y1, dptr(depth), None)
    return depth


def twoview_cross_check(lcam, rcam, p, dl, dr):
    dl = np.ascontiguousarray(dl, dtype=np.float64).copy()
    dr = np.ascontiguousarray(dr, dtype=np.float64).copy()
    h, w = dl.shape
    lib().sro_twoview_cross_check(w, h, C.byref(lcam), C.byref(rcam), C.byref(p), dptr(dl), dptr(dr))
    return dl, dr


def mvs_neighbours(cams, p):
    n = len(cams)
    arr = (Camera * n)(*cams)
    neigh = np.full((n, p.num_neighbours), -1, dtype=np.int32)
    cnt = np.zeros(n, dtype=np.int32)
    lib().sro_mvs_neighbours(n, arr, C.byref(p), iptr(neigh), iptr(cnt))
    return [list(neigh[v, :cnt[v]]) for v in range(n)]


def mvs_initial_estimate(imgs, cams, view, neigh, p, y0=0, y1=None, want_peaks=False):
    n = len(cams)
    carr = (Camera * n)(*cams)
    iarr = (Image * n)(*[im.c for im in imgs])
    w, h = imgs[view].w, imgs[view].h
    y1 = h if y1 is None else y1
    depth = np.full((h, w), np.inf, dtype=np.float64)
    peaks = np.zeros((h, w, p.top_k, 2), dtype=np.float64) if want_peaks else None
    ng = np.ascontiguousarray(neigh, dtype=np.int32)
    ne = C.c_int64(0)
    lib().sro_mvs_initial_estimate(n, iarr, carr, view, iptr(ng), len(ng), C.byref(p), y0, y1,
                                   dptr(depth), dptr(peaks) if want_peaks else None, C.byref(ne))
    if want_peaks:
        return depth, peaks, ne.value
    return depth, ne.value


def mvs_cross_check(imgs, cams, view, p, depths):
    """In place on depths[view] (list of 2-D float64 arrays)."""
    n = len(cams)
    carr = (Camera * n)(*cams)
    iarr = (Image * n)(*[im.c for im in imgs])
    ptrs = (c_double_p * n)(*[dptr(d) for d in depths])
    lib().sro_mvs_cross_check(n, iarr, carr, view, C.byref(p), ptrs)
