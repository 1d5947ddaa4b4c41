"""ctypes binding of include/stereo_recon_hip.h (libstereo_recon_hip.so).

This is the ONLY compute path of the package: there is no CPU / numpy fallback.
If the shared library is missing the import of :func:`lib` raises, and if no HIP
device is present :class:`Context` raises ``StereoHipError`` (SRH_E_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRH_LIBRARY names another build of the same library (the -DSRH_PROFILE_PHASES diagnostic build)
LIB_PATH = os.environ.get("SRH_LIBRARY") or os.path.join(_HERE, "libstereo_recon_hip.so")

SRH_OK = 0
SRH_E_INVALID, SRH_E_DEVICE, SRH_E_NO_DEVICE, SRH_E_CANCELLED, SRH_E_UNSUPPORTED = -1, -2, -3, -4, -5
WEIGHT_ADAPTIVE, WEIGHT_GEODESIC = 0, 1
MAX_VIEWS = 64

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Camera(C.Structure):
    """srh_camera"""
    _fields_ = [
        ("K", C.c_double * 9), ("Kinv", C.c_double * 9), ("R", C.c_double * 9), ("Rinv", C.c_double * 9),
        ("t", C.c_double * 3), ("C", C.c_double * 3),
        ("dist", C.c_double * 5),
        ("is_distorted", C.c_int32), ("is_refractive", C.c_int32),
        ("plane_normal", C.c_double * 3), ("plane_dist", C.c_double), ("refr_index", C.c_double),
        ("pdir", C.c_double * 3),
    ]


class Params(C.Structure):
    """srh_params"""
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


class SrhMrfParams(C.Structure):
    _fields_ = [("beta", C.c_double), ("lambda_", C.c_double), ("phi_u", C.c_double), ("psi_u", C.c_double),
                ("max_iters", C.c_int32), ("min_energy_drop", C.c_double)]


class SrhMrfInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("energy_initial", C.c_double), ("energy_final", C.c_double)]


# option "arith" (include/stereo_recon_hip.h)
ARITH_EXACT, ARITH_FMA, ARITH_F32, ARITH_CERTIFIED = 0, 1, 2, 3
ARITH_DEFAULT = ARITH_CERTIFIED


class Stats(C.Structure):
    """srh_stats"""
    _fields_ = [("n_pixels", C.c_int64), ("n_eval", C.c_int64), ("n_eval_device", C.c_int64),
                ("used_dense_path", C.c_int32), ("used_fused_kernel", C.c_int32),
                ("used_strip_kernel", C.c_int32), ("band_retries", C.c_int32),
                ("n_certified", C.c_int64), ("n_flagged", C.c_int64), ("band_budget_bytes", C.c_int64),
                ("mvs_waves_staged", C.c_int64), ("mvs_waves_listed", C.c_int64),
                ("scan_tiles_template", C.c_int64), ("scan_tiles_walked", C.c_int64)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)

# every symbol include/stereo_recon_hip.h declares
EXPORTS = [
    "srh_abi_version", "srh_build_id", "srh_last_error", "srh_device_count", "srh_hw_queues_requested",
    "srh_params_twoview_defaults", "srh_params_mvs_defaults", "srh_camera_from_krt", "srh_camera_from_p",
    "srh_mvs_neighbours", "srh_cert_bound", "srh_cert_sigma3",
    "srh_create", "srh_destroy", "srh_set_stream", "srh_set_hooks", "srh_synchronize", "srh_set_option",
    "srh_view_upload", "srh_view_size", "srh_view_depth_download", "srh_view_depth_upload",
    "srh_view_depth_device_ptr", "srh_view_depth_copy_to_device", "srh_view_depth_copy_from_device",
    "srh_twoview_wta", "srh_twoview_cross_check", "srh_twoview_compute", "srh_twoview_cost_rows", "srh_debug_exp",
    "srh_mvs_initial_estimate", "srh_mvs_cross_check", "srh_view_point_cloud", "srh_epipolar_curves",
    "srh_epipolar_preview", "srh_refraction_error",
    "srh_mrf_params_defaults", "srh_mvs_mrf_estimate", "srh_mvs_mrf_state", "srh_mvs_mrf_dims", "srh_mvs_initial_estimate_mrf",
    "srh_mvs_initial_estimate_peaks", "srh_mvs_mrf_estimate_views",
    "srh_comm_unique_id", "srh_comm_init", "srh_comm_gather_depth", "srh_comm_allgather_depth", "srh_comm_allgather_host",
    "srh_comm_allgather_views",
    "srh_comm_destroy", "srh_comm_set_timeout_ms", "srh_comm_version", "srh_comm_info",
    "srh_get_stats", "srh_profile_enable", "srh_profile_reset", "srh_profile_get", "srh_profile_dump",
]


class CertInfo(C.Structure):
    _fields_ = [("e0", C.c_double), ("k1", C.c_double), ("k2", C.c_double), ("k3", C.c_double), ("zmax2", C.c_double),
                ("m_hi", C.c_double), ("ok", C.c_int32), ("taps", C.c_int32)]


class StereoHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("srh error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libstereo_recon_hip.so; raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `make -C stereoreconstruction_amd/csrc` "
                      "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.srh_abi_version.restype = C.c_int
    L.srh_build_id.restype = C.c_char_p
    L.srh_last_error.restype = C.c_char_p
    L.srh_device_count.argtypes = [C.POINTER(C.c_int)]
    L.srh_params_twoview_defaults.argtypes = [C.POINTER(Params)]
    L.srh_params_twoview_defaults.restype = None
    L.srh_params_mvs_defaults.argtypes = [C.POINTER(Params)]
    L.srh_params_mvs_defaults.restype = None
    L.srh_camera_from_krt.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                      C.c_double, C.c_double, C.POINTER(Camera)]
    L.srh_camera_from_p.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.POINTER(Camera)]
    L.srh_mvs_neighbours.argtypes = [C.c_int, C.POINTER(Camera), C.POINTER(Params), c_int32_p, c_int32_p]
    L.srh_cert_bound.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(CertInfo)]
    L.srh_cert_sigma3.argtypes = [C.POINTER(Params), C.c_int, C.c_double]
    L.srh_cert_sigma3.restype = C.c_double
    L.srh_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.srh_destroy.argtypes = [vp]
    L.srh_destroy.restype = None
    L.srh_set_stream.argtypes = [vp, vp]
    L.srh_set_hooks.argtypes = [vp, C.POINTER(C.c_int), PROGRESS_FN, vp]
    L.srh_synchronize.argtypes = [vp]
    L.srh_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.srh_view_upload.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_uint8_p, c_uint8_p, C.POINTER(Camera)]
    L.srh_view_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.srh_view_depth_download.argtypes = [vp, C.c_int, c_double_p]
    L.srh_view_depth_upload.argtypes = [vp, C.c_int, c_double_p]
    L.srh_view_depth_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.srh_mrf_params_defaults.argtypes = [C.POINTER(SrhMrfParams)]
    L.srh_mrf_params_defaults.restype = None
    L.srh_mvs_mrf_estimate.argtypes = [vp, C.c_int, C.c_int, C.c_void_p, C.POINTER(SrhMrfParams), C.POINTER(SrhMrfInfo)]
    L.srh_mvs_initial_estimate_mrf.argtypes = [vp, C.c_int, c_int32_p, C.c_int, C.POINTER(Params), C.POINTER(SrhMrfParams), C.POINTER(SrhMrfInfo)]
    L.srh_mvs_initial_estimate_peaks.argtypes = [vp, C.c_int, c_int32_p, C.c_int, C.POINTER(Params)]
    L.srh_mvs_mrf_estimate_views.argtypes = [vp, c_int32_p, C.c_int, C.POINTER(SrhMrfParams), C.POINTER(SrhMrfInfo)]
    L.srh_mvs_mrf_state.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_int32_p, c_double_p, c_double_p]
    L.srh_mvs_mrf_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.srh_comm_allgather_views.argtypes = [vp, c_int32_p, C.c_int]
    L.srh_hw_queues_requested.restype = C.c_int
    L.srh_epipolar_preview.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, c_double_p, c_double_p, c_int32_p]
    L.srh_refraction_error.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.srh_view_point_cloud.argtypes = [vp, C.c_int, C.POINTER(Params), c_double_p, c_uint8_p, c_uint8_p, vp, vp, vp]
    L.srh_view_depth_copy_to_device.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.srh_view_depth_copy_from_device.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.srh_twoview_wta.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int]
    L.srh_twoview_cross_check.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Params)]
    L.srh_twoview_cost_rows.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, C.c_size_t,
                                        c_int32_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.srh_twoview_compute.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Params), c_double_p, c_double_p]
    L.srh_debug_exp.argtypes = [vp, c_double_p, C.c_int, c_double_p, c_double_p]
    L.srh_mvs_initial_estimate.argtypes = [vp, C.c_int, c_int32_p, C.c_int, C.POINTER(Params), C.c_int, C.c_int, vp]
    L.srh_mvs_cross_check.argtypes = [vp, c_int32_p, C.c_int, C.c_int, C.POINTER(Params)]
    L.srh_epipolar_curves.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int, c_int32_p,
                                      c_int32_p, C.c_int, c_int32_p]
    L.srh_comm_unique_id.argtypes = [vp]
    L.srh_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.srh_comm_gather_depth.argtypes = [vp, C.c_int, C.c_int, vp]
    L.srh_comm_allgather_depth.argtypes = [vp, C.c_int, vp]
    L.srh_comm_allgather_host.argtypes = [vp, c_double_p, C.c_size_t, c_double_p]
    L.srh_comm_destroy.argtypes = [vp]
    L.srh_comm_set_timeout_ms.argtypes = [C.c_int]
    L.srh_comm_version.restype = C.c_int
    L.srh_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.srh_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.srh_profile_enable.argtypes = [vp, C.c_int]
    L.srh_profile_reset.argtypes = [vp]
    L.srh_profile_get.argtypes = [vp, C.c_char_p, c_double_p, C.POINTER(C.c_int64)]
    L.srh_profile_dump.argtypes = [vp, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _check(rc):
    if rc != SRH_OK:
        raise StereoHipError(rc, lib().srh_last_error().decode("utf-8", "replace"))


def _dptr(a):
    return a.ctypes.data_as(c_double_p)


def params_twoview(**kw):
    p = Params()
    lib().srh_params_twoview_defaults(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def mrf_params(**kw):
    """srh_mrf_params with the reference's constants (multiviewstereo.cpp:98-101, 631, 641)."""
    m = SrhMrfParams()
    lib().srh_mrf_params_defaults(C.byref(m))
    for k, v in kw.items():
        setattr(m, "lambda_" if k == "lambda" else k, v)
    return m


def params_mvs(**kw):
    p = Params()
    lib().srh_params_mvs_defaults(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def camera_from_krt(K, R, t, dist=None, plane_normal=None, plane_dist=0.0, refr_index=1.0):
    """Camera::set(K,R,t) + distortion + refractive interface -> srh_camera snapshot."""
    cam = Camera()
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    d = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
    n = None if plane_normal is None else np.ascontiguousarray(plane_normal, dtype=np.float64).reshape(3)
    _check(lib().srh_camera_from_krt(_dptr(K), _dptr(R), _dptr(t),
                                     _dptr(d) if d is not None else None,
                                     _dptr(n) if n is not None else None,
                                     plane_dist, refr_index, C.byref(cam)))
    return cam


def camera_from_p(P, dist=None, plane_normal=None, plane_dist=0.0, refr_index=1.0):
    """Camera::setP (3x4 projection matrix, the project-file path) -> srh_camera snapshot."""
    cam = Camera()
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(12)
    d = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
    n = None if plane_normal is None else np.ascontiguousarray(plane_normal, dtype=np.float64).reshape(3)
    _check(lib().srh_camera_from_p(_dptr(P), _dptr(d) if d is not None else None,
                                   _dptr(n) if n is not None else None, plane_dist, refr_index, C.byref(cam)))
    return cam


def mvs_neighbours(cams, p):
    n = len(cams)
    arr = (Camera * n)(*cams)
    neigh = np.full((n, max(1, p.num_neighbours)), -1, dtype=np.int32)
    cnt = np.zeros(n, dtype=np.int32)
    _check(lib().srh_mvs_neighbours(n, arr, C.byref(p), neigh.ctypes.data_as(c_int32_p),
                                    cnt.ctypes.data_as(c_int32_p)))
    return [[int(v) for v in neigh[i, :cnt[i]]] for i in range(n)]


def comm_version():
    """NCCL_VERSION_CODE of the librccl the library loads (ncclGetVersion), 0 when there is none: srh_comm_version."""
    return int(lib().srh_comm_version())


def comm_set_timeout_ms(ms):
    _check(lib().srh_comm_set_timeout_ms(int(ms)))


def cert_bound(p, mvs=False):
    """srh_cert_bound: the constants of the certified arithmetic's error bound for these parameters, as a dict."""
    ci = CertInfo()
    _check(lib().srh_cert_bound(C.byref(p), 1 if mvs else 0, C.byref(ci)))
    return {k: getattr(ci, k) for k, _ in CertInfo._fields_}


def cert_sigma3(p, sum2, mvs=False):
    """srh_cert_sigma3: the smallest sum3 for which a candidate of a pixel with this sum2 is certified (+inf: none)."""
    return float(lib().srh_cert_sigma3(C.byref(p), 1 if mvs else 0, float(sum2)))


def hw_queues_requested():
    """Hardware queues the HIP runtime was asked for (GPU_MAX_HW_QUEUES, else the runtime's default 4): srh_hw_queues_requested."""
    return int(lib().srh_hw_queues_requested())


def device_count():
    n = C.c_int(0)
    _check(lib().srh_device_count(C.byref(n)))
    return n.value


def _torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; this library links the
    system ROCm.  Both can live in one process, but the bundled runtime must attach to the GPU first
    (the other order leaves torch with "No HIP GPUs are available").  So when the process has torch
    loaded, its runtime is initialised before the first srh_create; torch itself is never required."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and not torch.cuda.is_initialized():
        torch.cuda.init()


def build_id():
    """Hash of the sources the loaded library was built from (srh_build_id)."""
    return lib().srh_build_id().decode()


class Context:
    """One srh_context bound to one GPU."""

    def __init__(self, device=0):
        _torch_runtime_first()
        self._h = C.c_void_p()
        _check(lib().srh_create(device, C.byref(self._h)))
        self._keep = []
        self._comm_ranks = 0

    def close(self):
        if self._h:
            lib().srh_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing
    def set_stream(self, stream_ptr):
        _check(lib().srh_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_hooks(self, cancel_flag=None, progress=None):
        """cancel_flag: ctypes.c_int; progress: callable(step, stage_str)."""
        cb = PROGRESS_FN(lambda step, stage, user: progress(step, stage.decode())) if progress else PROGRESS_FN()
        self._keep = [cb, cancel_flag]
        _check(lib().srh_set_hooks(self._h, C.byref(cancel_flag) if cancel_flag is not None else None, cb, None))

    def set_option(self, name, value):
        _check(lib().srh_set_option(self._h, name.encode(), int(value)))

    def synchronize(self):
        _check(lib().srh_synchronize(self._h))

    # -- views
    def upload_view(self, slot, rgba, mask, cam):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("rgba must be HxWx4 uint8")
        h, w = rgba.shape[:2]
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (h, w):
                raise ValueError("mask must be HxW uint8")
        _check(lib().srh_view_upload(self._h, slot, w, h, rgba.ctypes.data_as(c_uint8_p),
                                     m.ctypes.data_as(c_uint8_p) if m is not None else None, C.byref(cam)))

    def view_size(self, slot):
        w, h = C.c_int(0), C.c_int(0)
        _check(lib().srh_view_size(self._h, slot, C.byref(w), C.byref(h)))
        return w.value, h.value

    def download_depth(self, slot):
        w, h = self.view_size(slot)
        out = np.empty((h, w), dtype=np.float64)
        _check(lib().srh_view_depth_download(self._h, slot, _dptr(out)))
        return out

    def upload_depth(self, slot, depth):
        w, h = self.view_size(slot)
        d = np.ascontiguousarray(depth, dtype=np.float64)
        if d.shape != (h, w):
            raise ValueError("depth must be %dx%d" % (h, w))
        _check(lib().srh_view_depth_upload(self._h, slot, _dptr(d)))

    def depth_device_ptr(self, slot):
        p = C.c_void_p()
        _check(lib().srh_view_depth_device_ptr(self._h, slot, C.byref(p)))
        return p.value

    def copy_depth_to_device(self, slot, dst_dev_ptr, dst_bytes):
        _check(lib().srh_view_depth_copy_to_device(self._h, slot, C.c_void_p(dst_dev_ptr), C.c_size_t(dst_bytes)))

    def copy_depth_from_device(self, slot, src_dev_ptr, src_bytes):
        _check(lib().srh_view_depth_copy_from_device(self._h, slot, C.c_void_p(src_dev_ptr), C.c_size_t(src_bytes)))

    # -- TwoViewStereo
    def twoview_wta(self, ref_slot, oth_slot, p, y0=0, y1=0):
        _check(lib().srh_twoview_wta(self._h, ref_slot, oth_slot, C.byref(p), y0, y1))

    def twoview_cross_check(self, left_slot, right_slot, p):
        _check(lib().srh_twoview_cross_check(self._h, left_slot, right_slot, C.byref(p)))

    def twoview_compute(self, left_slot, right_slot, p):
        w, h = self.view_size(left_slot)
        dl = np.empty((h, w), dtype=np.float64)
        dr = np.empty((h, w), dtype=np.float64)
        _check(lib().srh_twoview_compute(self._h, left_slot, right_slot, C.byref(p), _dptr(dl), _dptr(dr)))
        return dl, dr

    def debug_exp(self, x):
        """srh_debug_exp -> (the geodesic kernels' exp sequence, the device library's exp) of the arguments x."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        a, b = np.empty_like(x), np.empty_like(x)
        _check(lib().srh_debug_exp(self._h, _dptr(x), x.size, _dptr(a), _dptr(b)))
        return a, b

    def twoview_compute_device(self, left_slot, right_slot, p):
        """srh_twoview_compute without host outputs: both passes + cross-check, the maps stay in the slots' device memory."""
        _check(lib().srh_twoview_compute(self._h, left_slot, right_slot, C.byref(p), None, None))

    # -- MultiViewStereo
    def twoview_cost_rows(self, ref, oth, p, y0, y1, form, raw=False):
        """srh_twoview_cost_rows (diagnostic) -> (cost (rows, w, cstride) float64 with cost[r, x, k] the cost of column
        lo + k, range (rows, w, 2) int32, used_strip_kernel)."""
        w, h = self.view_size(ref)
        cs, us = C.c_int(0), C.c_int(0)
        _check(lib().srh_twoview_cost_rows(self._h, ref, oth, C.byref(p), y0, y1, form, 1 if raw else 0, None, 0, None, C.byref(cs), C.byref(us)))
        rows, tiles = y1 - y0, (w + 31) // 32
        buf = np.empty((rows, tiles, cs.value, 32), dtype=np.float64)
        rng = np.empty((rows, w, 2), dtype=np.int32)
        _check(lib().srh_twoview_cost_rows(self._h, ref, oth, C.byref(p), y0, y1, form, 1 if raw else 0, _dptr(buf), buf.size,
                                           rng.ctypes.data_as(c_int32_p), C.byref(cs), C.byref(us)))
        cost = np.ascontiguousarray(buf.transpose(0, 1, 3, 2)).reshape(rows, tiles * 32, cs.value)[:, :w]
        return cost, rng, bool(us.value)

    def mvs_initial_estimate(self, view_slot, neigh_slots, p, y0=0, y1=0, peaks_dev=None):
        ng = np.ascontiguousarray(neigh_slots, dtype=np.int32)
        _check(lib().srh_mvs_initial_estimate(self._h, view_slot, ng.ctypes.data_as(c_int32_p), len(ng),
                                              C.byref(p), y0, y1, C.c_void_p(peaks_dev) if peaks_dev else None))

    def mvs_cross_check(self, slots, view_index, p):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        _check(lib().srh_mvs_cross_check(self._h, s.ctypes.data_as(c_int32_p), len(s), view_index, C.byref(p)))

    def point_cloud(self, slot, p):
        """-> dict(xyz (h,w,3) float64 with NaN where there is no point, rgb (h,w,3) uint8, valid (h,w) uint8,
        n_points, n_masked, n_finite): srh_view_point_cloud."""
        w, h = self.view_size(slot)
        xyz = np.empty((h, w, 3), dtype=np.float64)
        rgb = np.empty((h, w, 3), dtype=np.uint8)
        valid = np.empty((h, w), dtype=np.uint8)
        n = (C.c_int64 * 3)()
        _check(lib().srh_view_point_cloud(self._h, slot, C.byref(p), _dptr(xyz), rgb.ctypes.data_as(c_uint8_p),
                                          valid.ctypes.data_as(c_uint8_p), C.cast(C.byref(n, 0), C.c_void_p),
                                          C.cast(C.byref(n, 8), C.c_void_p), C.cast(C.byref(n, 16), C.c_void_p)))
        return dict(xyz=xyz, rgb=rgb, valid=valid, n_points=int(n[0]), n_masked=int(n[1]), n_finite=int(n[2]))

    def mvs_mrf_estimate(self, view_slot, top_k, peaks_dev, m=None):
        """MRF branch of computeInitialEstimate on the device peaks buffer -> dict(iterations, energy_initial, energy_final)."""
        m = m if m is not None else mrf_params()
        info = SrhMrfInfo()
        _check(lib().srh_mvs_mrf_estimate(self._h, view_slot, top_k, C.c_void_p(peaks_dev), C.byref(m), C.byref(info)))
        return dict(iterations=info.iterations, energy_initial=info.energy_initial, energy_final=info.energy_final)

    def mvs_initial_estimate_mrf(self, view_slot, neigh_slots, p, m=None):
        """computeInitialEstimate of a CONFIG+=mrf build: peaks, then TRW-S on them."""
        m = m if m is not None else mrf_params()
        info = SrhMrfInfo()
        nb = np.ascontiguousarray(neigh_slots, dtype=np.int32)
        _check(lib().srh_mvs_initial_estimate_mrf(self._h, view_slot, nb.ctypes.data_as(c_int32_p), len(nb), C.byref(p),
                                                  C.byref(m), C.byref(info)))
        return dict(iterations=info.iterations, energy_initial=info.energy_initial, energy_final=info.energy_final)

    def mvs_initial_estimate_peaks(self, view_slot, neigh_slots, p):
        """Initial estimate of one view, its top-K peaks kept in the context for mvs_mrf_estimate_views."""
        nb = np.ascontiguousarray(neigh_slots, dtype=np.int32)
        _check(lib().srh_mvs_initial_estimate_peaks(self._h, view_slot, nb.ctypes.data_as(c_int32_p), len(nb), C.byref(p)))

    def mvs_mrf_estimate_views(self, view_slots, m=None):
        """MRF stage of several views side by side -> list of dict(iterations, energy_initial, energy_final)."""
        m = m if m is not None else mrf_params()
        sl = np.ascontiguousarray(view_slots, dtype=np.int32)
        infos = (SrhMrfInfo * len(sl))()
        _check(lib().srh_mvs_mrf_estimate_views(self._h, sl.ctypes.data_as(c_int32_p), len(sl), C.byref(m), infos))
        return [dict(iterations=i.iterations, energy_initial=i.energy_initial, energy_final=i.energy_final) for i in infos]

    def mvs_mrf_state(self, w, h, top_k):
        """(labels (h,w) int32, data_costs (h,w,K+1), messages (h,w,2,K+1)) of the last MRF run."""
        labels = np.zeros((h, w), dtype=np.int32)
        D = np.zeros((h, w, top_k + 1), dtype=np.float64)
        M = np.zeros((h, w, 2, top_k + 1), dtype=np.float64)
        _check(lib().srh_mvs_mrf_state(self._h, w, h, top_k, labels.ctypes.data_as(c_int32_p), _dptr(D), _dptr(M)))
        return labels, D, M

    def mvs_mrf_dims(self):
        """(w, h, top_k) of the single-view MRF run whose state mvs_mrf_state reports."""
        w, h, k = C.c_int(), C.c_int(), C.c_int()
        _check(lib().srh_mvs_mrf_dims(self._h, C.byref(w), C.byref(h), C.byref(k)))
        return w.value, h.value, k.value

    def epipolar_preview(self, ref_slot, oth_slot, min_depth, max_depth, num_depths, xy):
        """The GUI's curve preview (StereoWidget::epipolarLineItem) for the pixels `xy` (n,2) -> list of (k,2) float64."""
        q = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        n = q.shape[0]
        out = np.zeros((max(n, 1), num_depths, 2), dtype=np.float64)
        counts = np.zeros(max(n, 1), dtype=np.int32)
        _check(lib().srh_epipolar_preview(self._h, ref_slot, oth_slot, C.c_double(min_depth), C.c_double(max_depth),
                                          num_depths, n, _dptr(q), _dptr(out), counts.ctypes.data_as(c_int32_p)))
        return [out[i, :counts[i]].copy() for i in range(n)]

    def refraction_error(self, slot1, slot2, p1, p2):
        """RefractionCalibration::error per pair and totalError -> (errors (n,), total, average)."""
        a = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 2)
        b = np.ascontiguousarray(p2, dtype=np.float64).reshape(-1, 2)
        assert a.shape == b.shape
        err = np.zeros(max(len(a), 1), dtype=np.float64)
        tot, avg = C.c_double(0), C.c_double(0)
        _check(lib().srh_refraction_error(self._h, slot1, slot2, len(a), _dptr(a), _dptr(b), _dptr(err), C.byref(tot), C.byref(avg)))
        return err[:len(a)].copy(), tot.value, avg.value

    def epipolar_curves(self, ref_slot, oth_slot, p, xy, mvs=False, max_pts=4096):
        """Candidate pixels of each reference pixel in `xy` (n,2), in the reference's visiting order
        -> list of (len,2) int32 arrays (TwoViewStereo/MultiViewStereo::epipolarCurve)."""
        q = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        n = q.shape[0]
        counts = np.zeros(max(n, 1), dtype=np.int32)
        while True:
            out = np.zeros((max(n, 1), max_pts, 2), dtype=np.int32)
            _check(lib().srh_epipolar_curves(self._h, ref_slot, oth_slot, C.byref(p), int(bool(mvs)), n,
                                             q.ctypes.data_as(c_int32_p), out.ctypes.data_as(c_int32_p), max_pts,
                                             counts.ctypes.data_as(c_int32_p)))
            if n == 0 or counts[:n].max() <= max_pts:
                return [out[i, :counts[i]].copy() for i in range(n)]
            max_pts = int(counts[:n].max())

    # -- multi-GPU exchange (RCCL)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().srh_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        _check(lib().srh_comm_init(self._h, nranks, rank, C.c_char_p(unique_id)))
        self._comm_ranks = nranks

    def comm_gather_depth(self, slot, root, recv_dev_ptr):
        _check(lib().srh_comm_gather_depth(self._h, slot, root, C.c_void_p(recv_dev_ptr)))

    def comm_allgather_depth(self, slot, recv_dev_ptr):
        _check(lib().srh_comm_allgather_depth(self._h, slot, C.c_void_p(recv_dev_ptr)))

    def comm_allgather_host(self, send):
        """Every rank contributes `send` (float64 host array), every rank receives ranks*len(send) values (host)."""
        a = np.ascontiguousarray(send, dtype=np.float64).reshape(-1)
        out = np.empty(a.size * max(1, self._comm_ranks), dtype=np.float64)
        _check(lib().srh_comm_allgather_host(self._h, a.ctypes.data_as(c_double_p), a.size, out.ctypes.data_as(c_double_p)))
        return out

    def comm_allgather_views(self, view_slots):
        """Sharded MultiViewStereo: every rank contributes the depth maps of its shard of `view_slots`, device to device."""
        sl = np.ascontiguousarray(view_slots, dtype=np.int32)
        _check(lib().srh_comm_allgather_views(self._h, sl.ctypes.data_as(c_int32_p), len(sl)))

    def comm_destroy(self):
        _check(lib().srh_comm_destroy(self._h))

    # -- measurement
    def stats(self):
        s = Stats()
        _check(lib().srh_get_stats(self._h, C.byref(s)))
        return dict(n_pixels=s.n_pixels, n_eval=s.n_eval, n_eval_device=s.n_eval_device,
                    used_dense_path=bool(s.used_dense_path), used_fused_kernel=bool(s.used_fused_kernel),
                    used_strip_kernel=bool(s.used_strip_kernel), band_retries=s.band_retries,
                    n_certified=s.n_certified, n_flagged=s.n_flagged, band_budget_bytes=s.band_budget_bytes,
                    mvs_waves_staged=s.mvs_waves_staged, mvs_waves_listed=s.mvs_waves_listed,
                    scan_tiles_template=s.scan_tiles_template, scan_tiles_walked=s.scan_tiles_walked)

    def profile_enable(self, on=True):
        _check(lib().srh_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _check(lib().srh_profile_reset(self._h))

    def profile(self):
        buf = C.create_string_buffer(1 << 16)
        _check(lib().srh_profile_dump(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, n = line.split()
            out[name] = (float(ms), int(n))
        return out
