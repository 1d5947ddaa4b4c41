"""The rows of the oracle that cannot be pinned against a compiled reference here (Eigen / GSL are absent:
camera model, rays, refraction, costs) are at least pinned against what they must mean physically and against
closed-form answers -- independent of how the restatement was written."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle_ffi as O


def _unproject(cam, x, y):
    s, d = np.zeros(3), np.zeros(3)
    O.lib().sro_unproject(C.byref(cam), x, y, O.dptr(s), O.dptr(d))
    return s, d


def _project(cam, X):
    p = np.array(X, dtype=np.float64)
    ok = O.lib().sro_project(C.byref(cam), O.dptr(p))
    return bool(ok), p


def _dist_point_ray(X, s, d):
    v = X - s
    return np.linalg.norm(v - (v @ d) * d / (d @ d))


K = np.array([[900.0, 0.0, 320.5], [0.0, 880.0, 241.25], [0.0, 0.0, 1.0]])
R = cases._rot_z(0.1) @ cases._rot_x(-0.2) @ cases._rot_y(0.3)
Cc = np.array([0.4, -0.2, 1.5])


def test_pinhole_projection_is_K_R_t():
    cam = O.camera_set(K, R, -R @ Cc)
    rng = np.random.default_rng(1)
    for _ in range(50):
        X = Cc + R.T @ np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(3, 20)])
        ok, p = _project(cam, X)
        q = K @ (R @ X - R @ Cc)
        assert ok and np.allclose(p[:2], q[:2] / q[2], rtol=1e-12, atol=1e-9)
        s, d = _unproject(cam, p[0], p[1])
        assert np.allclose(s, Cc, atol=1e-12) and _dist_point_ray(X, s, d) < 1e-9
        assert abs(np.linalg.norm(d) - 1) < 1e-12


def test_distorted_camera_round_trip():
    """project applies the OpenCV forward model; unproject inverts it with 5 fixed-point iterations
    (camera.cpp:439-446), so the round trip closes to the convergence of that iteration."""
    dist = np.array([-0.12, 0.25, 0.002, -0.001, -0.3])
    cam = O.camera_set(K, R, -R @ Cc, dist)
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(50):
        X = Cc + R.T @ np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(6, 20)])
        ok, p = _project(cam, X)
        assert ok
        # forward model checked in closed form
        xc = R @ (X - Cc)
        x, y = xc[0] / xc[2], xc[1] / xc[2]
        r2 = x * x + y * y
        cd = 1 + ((dist[4] * r2 + dist[1]) * r2 + dist[0]) * r2
        xd = x * cd + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
        yd = y * cd + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * xd * y          # the reference reuses the updated x
        assert np.allclose(p[:2], [K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], rtol=1e-12, atol=1e-9)
        s, d = _unproject(cam, p[0], p[1])
        worst = max(worst, _dist_point_ray(X, s, d) / np.linalg.norm(X - Cc))
    assert worst < 1e-5


def test_refraction_obeys_snell_and_projection_inverts_it():
    n = np.array([0.03, -0.02, 1.0])
    dist_if, ratio = 0.1, 1.333
    cam = O.camera_set(K, R, -R @ Cc, None, n, dist_if, ratio)
    assert cam.is_refractive
    nn = n / np.linalg.norm(n)
    rng = np.random.default_rng(3)
    for _ in range(40):
        px, py = rng.uniform(40, 600), rng.uniform(40, 440)
        s, d = _unproject(cam, px, py)
        # in camera coordinates: the ray leaves the interface point s_c with direction d_c
        s_c, d_c = R @ (s - Cc), R @ d
        assert abs(s_c @ nn - dist_if) < 1e-9                                   # it starts on the interface plane
        a = np.linalg.inv(K) @ np.array([px, py, 1.0])
        a /= np.linalg.norm(a)                                                   # the ray inside the housing
        sin_in = np.linalg.norm(np.cross(a, nn)); sin_out = np.linalg.norm(np.cross(d_c, nn)) / np.linalg.norm(d_c)
        assert abs(sin_in - ratio * sin_out) < 1e-9                             # Snell: sin(in) = n * sin(out)
        assert abs(np.dot(np.cross(a, nn), d_c)) < 1e-9                         # the refracted ray stays in the plane of incidence
        # a point on the refracted ray projects back to the pixel
        X = s + rng.uniform(2, 15) * d
        ok, p = _project(cam, X)
        assert ok and np.allclose(p[:2], [px, py], atol=1e-6)


def test_closest_points_and_depth():
    P = np.array([0.3, -0.4, 7.0])
    s1, s2 = np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.2, -0.1])
    d1, d2 = (P - s1) / np.linalg.norm(P - s1), (P - s2) / np.linalg.norm(P - s2)
    p1, p2 = np.zeros(3), np.zeros(3)
    O.lib().sro_closest_points(O.dptr(s1), O.dptr(d1), O.dptr(s2), O.dptr(d2), O.dptr(p1), O.dptr(p2))
    assert np.allclose(p1, P, atol=1e-9) and np.allclose(p2, P, atol=1e-9)
    # skew rays: the segment between the closest points is perpendicular to both
    d2b = d2 + np.array([0.0, 0.05, 0.0]); d2b /= np.linalg.norm(d2b)
    O.lib().sro_closest_points(O.dptr(s1), O.dptr(d1), O.dptr(s2), O.dptr(d2b), O.dptr(p1), O.dptr(p2))
    assert abs((p1 - p2) @ d1) < 1e-9 and abs((p1 - p2) @ d2b) < 1e-9


@pytest.mark.parametrize("mvs", [False, True])
def test_costs_of_identical_and_inverted_windows(mvs):
    """NCC of a window with itself is perfect (TwoView: cost 0, MVS: 1); with its photographic negative the
    TwoView cost is also 0 (abs) and the MVS score -1; a constant window has no variance (TwoView: the NaN of
    0/0 falls to MAX_COLOR_DIFF through std::min, MVS: 0)."""
    rng = np.random.default_rng(5)
    h, w, r = 16, 16, 2
    a = np.zeros((h, w, 4), np.uint8); a[..., :3] = rng.integers(0, 256, (h, w, 1)); a[..., 3] = 255
    neg = a.copy(); neg[..., :3] = 255 - a[..., :3]
    const = a.copy(); const[..., :3] = 77
    mk = O.params_mvs if mvs else O.params_twoview
    p = mk(min_depth=1, max_depth=2, num_depth_levels=4, window_radius=r, weight_kind=0)
    f = O.lib().sro_mvs_cost_ncc if mvs else O.lib().sro_twoview_cost_ncc
    f.restype = C.c_double
    mask = np.ones((h, w), np.uint8)
    ia = O.OImage(a, mask)
    wts = O.weights(ia, 8, 8, p)
    same = f(C.byref(ia.c), C.byref(O.OImage(a, mask).c), O.dptr(wts), C.byref(p), 8, 8, 8, 8)
    # (the weight multiplies the gray value *before* the mean is subtracted -- twoviewstereo.cpp:962-966 -- so
    # the negative is a perfect anti-correlation only under uniform weights)
    ones = np.ones_like(wts)
    inv = f(C.byref(ia.c), C.byref(O.OImage(neg, mask).c), O.dptr(ones), C.byref(p), 8, 8, 8, 8)
    flat = f(C.byref(ia.c), C.byref(O.OImage(const, mask).c), O.dptr(ones), C.byref(p), 8, 8, 8, 8)
    if mvs:
        assert abs(same - 1.0) < 1e-12 and abs(inv + 1.0) < 1e-12 and flat == 0.0
    else:
        assert abs(same) < 1e-9 and abs(inv) < 1e-9 and flat == 120.0


def test_depth_sampling_of_the_labels():
    """TwoView: t = l/(D-1); t /= 5-4t (denser near minDepth); MVS: uniform.  Seen through the curve: the first
    and the last label of a rectified pair land on the columns of max / min disparity."""
    w, h, D = 64, 24, 16
    case = cases.get_twoview("adaptive_rect", w=w, h=h, D=D)
    imgs, ocams, op = cases.oracle_inputs(case)
    pts = O.epipolar_curve(ocams[0], ocams[1], imgs[1], op, False, 40, 12)
    assert pts[:, 1].min() == pts[:, 1].max() == 12                            # rectified: the curve is the row
    assert pts[:, 0].max() == 40 - 8 and pts[:, 0].min() in (40 - 8 - D + 1, 40 - 8 - D + 2)   # d0 = 8; last fragment dropped


def test_refractive_root_is_the_companion_matrix_root_on_0_r():
    """camera.cpp:95-138 solves (n^2-1)x^4 - 2r(n^2-1)x^3 + (r^2(n^2-1) + d^2 n^2 - (z-d)^2)x^2 - 2 d^2 n^2 r x
    + d^2 n^2 r^2 = 0 with GSL's companion-matrix solver and keeps the root with 0 <= x <= r.  numpy.roots is the
    same method (eigenvalues of the companion matrix): exactly one real root lies in [0, r], and the oracle's
    Newton iteration lands on it."""
    nrm, d, n = np.array([0.0, 0.0, 1.0]), 0.1, 1.333
    cam = O.camera_set(np.eye(3), np.eye(3), np.zeros(3), None, nrm, d, n)
    rng = np.random.default_rng(11)
    for _ in range(60):
        X = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.5, 12)])
        z, r = X[2], np.hypot(X[0], X[1])
        co = [n * n - 1, -2 * r * (n * n - 1), r * r * (n * n - 1) + d * d * n * n - (z - d) ** 2,
              -2 * d * d * n * n * r, d * d * n * n * r * r]
        roots = np.roots(co)
        real = roots[np.abs(roots.imag) <= 1e-10].real
        inside = real[(real >= -1e-12) & (real <= r + 1e-12)]
        assert len(inside) == 1, (X, roots)
        ok, p = _project(cam, X)
        assert ok
        # with K = R = I the "pixel" is the crossing point of the interface divided by its depth d
        cross = np.array([p[0], p[1]]) * d
        want = inside[0] * np.array([X[0], X[1]]) / r
        assert np.allclose(cross, want, rtol=1e-9, atol=1e-12), (X, cross, want)
