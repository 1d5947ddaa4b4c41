"""SURVEY 8(f) rank 4 on the device: the GUI's uniform-depth curve preview (StereoWidget::epipolarLineItem,
gui/widgets/stereowidget.cpp:621-672) and the refractive-calibration error (RefractiveCalibrationFunction::diff and
RefractionCalibration::totalError, stereo/refractioncalibration.cpp:175-199, 408-469) through the C-ABI against the
oracle.  Vertex counts are integer work: identical; coordinates / errors within 1e-9 relative."""
import numpy as np
import pytest

import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

pytestmark = pytest.mark.gpu

RT = 1e-9


def _close(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= RT * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))
    return bool(np.all(ok | both_nan | (a == b)))


@pytest.mark.parametrize("name", ["geodesic_rect", "geodesic_verged_dist_masks", "adaptive_refractive"])
def test_curve_preview(hip_ctx, name):
    case = cases.get_twoview(name)
    _, ocams, _ = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    h, w = case["views"][0][0].shape[:2]
    rng = np.random.default_rng(11)
    q = np.concatenate([[[0, 0], [w - 1, h - 1], [w / 2, h / 2]], rng.uniform(-2, [w + 2, h + 2], size=(150, 2))])
    zmin, zmax = case["params"]["min_depth"], case["params"]["max_depth"]
    for ref, oth in ((0, 1), (1, 0)):
        for nd in (2, 64, 1000):
            got = hip_ctx.epipolar_preview(ref, oth, zmin, zmax, nd, q)
            drawn = 0
            for i, (x, y) in enumerate(q):
                want = O.epipolar_preview(ocams[ref], ocams[oth], x, y, zmin, zmax, nd)
                assert len(got[i]) == len(want), (name, ref, nd, x, y, len(got[i]), len(want))
                assert _close(got[i], want), (name, ref, nd, x, y)
                drawn += len(want) > 0
            assert nd == 2 or drawn > len(q) // 2
    assert hip_ctx.epipolar_preview(0, 1, zmin, zmax, 16, np.zeros((0, 2))) == []
    with pytest.raises(capi.StereoHipError):
        hip_ctx.epipolar_preview(0, 1, zmin, zmax, 1, q)            # (numDepthLevels - 1.0) would divide by zero
    with pytest.raises(capi.StereoHipError):
        hip_ctx.epipolar_preview(0, 40, zmin, zmax, 16, q)          # empty slot


@pytest.mark.parametrize("name", ["adaptive_refractive", "geodesic_verged_dist_masks"])
def test_refraction_error(hip_ctx, name):
    case = cases.get_twoview(name)
    _, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    h, w = case["views"][0][0].shape[:2]
    rng = np.random.default_rng(3)
    # correspondences the way the calibration sees them: a scene point in both images, plus detection noise
    p1, p2 = [], []
    zmid = 0.5 * (case["params"]["min_depth"] + case["params"]["max_depth"])
    while len(p1) < 300:
        x, y = rng.uniform(0, w), rng.uniform(0, h)
        pts = O.epipolar_preview(ocams[0], ocams[1], x, y, zmid, zmid * 1.0001, 2)
        a = np.array([x, y])
        b = pts[0] if len(pts) else rng.uniform(0, [w, h])
        p1.append(a); p2.append(b + rng.normal(0, 0.7, 2))
    p1 = np.array(p1); p2 = np.array(p2)
    err, total, avg = hip_ctx.refraction_error(0, 1, p1, p2)
    want = np.array([O.refraction_pair_error(ocams[0], ocams[1], a, b) for a, b in zip(p1, p2)])
    assert _close(err, want)
    wtot = 0.0
    for e in want:
        wtot += e * e
    assert _close(total, wtot) and _close(avg, wtot / len(want))
    assert np.isfinite(want).sum() > 250
    err0, tot0, avg0 = hip_ctx.refraction_error(0, 1, np.zeros((0, 2)), np.zeros((0, 2)))
    assert len(err0) == 0 and tot0 == 0.0 and np.isnan(avg0)        # 0 / 0, as totalError does with no pairs
