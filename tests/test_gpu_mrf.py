"""SURVEY 8(f) rank 2 on the device: the MRF branch of MultiViewStereo::computeInitialEstimate (multiviewstereo.cpp:
481-516, 610-652) -- TRW-S over the top-K peaks -- through the C-ABI against the oracle's restatement.
PARITY UNPINNED (the reference's -lMRF library is not in its tree): "the oracle" is the published algorithm as
oracle/sr_oracle.c writes it down, itself checked in tests/test_oracle_mrf.py.

What must hold:
  * data costs: LAMBDA * exp(-BETA * cost) -- the device's exp() and libm's may differ in the last place: <= 4 ulp;
  * the optimiser, fed the SAME data costs: labels, every stored message, the sweep count and the depth map identical
    (the sign of a zero message aside); energies within 1e-10 relative (the device sums them as a tree, the oracle one after the other);
  * end to end (each side with its own exp): energies within 1e-9, at most 0.1 % of the labels differ."""
import numpy as np
import pytest

import cases
import mrf_cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

pytestmark = pytest.mark.gpu


def _upload_blank_view(ctx, slot, mask):
    h, w = mask.shape
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    rgba[..., 3] = 255
    K = np.array([[100.0, 0, w / 2], [0, 100.0, h / 2], [0, 0, 1]])
    ctx.upload_view(slot, rgba, mask, capi.camera_from_krt(K, np.eye(3), np.zeros(3), None))


def _run_device(ctx, peaks, mask, m, before=-7.0):
    import torch
    h, w, K, _ = peaks.shape
    _upload_blank_view(ctx, 0, mask)
    ctx.upload_depth(0, np.full((h, w), before))
    pk = torch.from_numpy(peaks).to("cuda:0")
    torch.cuda.synchronize()
    info = ctx.mvs_mrf_estimate(0, K, pk.data_ptr(), m)
    labels, D, M = ctx.mvs_mrf_state(w, h, K)
    return info, labels, D, M, ctx.download_depth(0)


def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def _check_against_oracle(ctx, peaks, mask, over, tag):
    m = capi.mrf_params(**over)
    om = O.mrf_params(**over)
    info, labels, D, M, depth = _run_device(ctx, peaks, mask, m)
    ref_own = O.mvs_mrf(peaks, mask, om)                                  # libm exp
    h, w, K, _ = peaks.shape
    import ctypes as C
    # data costs
    pk = np.ascontiguousarray(peaks)
    wantD = np.where(pk[..., 1] < 0, om.lambda_, om.lambda_ * np.exp(-om.beta * pk[..., 0]))
    assert _ulps(D[..., :K], wantD).max() <= 4, tag
    assert (D[..., K] == om.phi_u).all()
    # the optimiser on identical data costs
    ref = O.mvs_mrf(peaks, mask, om, depth=np.full((h, w), -7.0), data_costs=D, want_messages=True)
    assert info["iterations"] == ref["iterations"], (tag, info, ref["iterations"])
    bad = np.argwhere(labels != ref["labels"])
    assert len(bad) == 0, "%s: %d labels differ, first at (y, x) = %s" % (tag, len(bad), bad[:5].tolist())
    neq = np.argwhere(M != ref["messages"])
    assert len(neq) == 0, "%s: %d message entries differ, first %s: %r vs %r" % (
        tag, len(neq), neq[:3].tolist(), M[tuple(neq[0])], ref["messages"][tuple(neq[0])])
    assert np.array_equal(depth.view(np.uint64), ref["depth"].view(np.uint64)), tag
    for k in ("energy_initial", "energy_final"):
        assert abs(info[k] - ref[k]) <= 1e-10 * max(1.0, abs(ref[k])), (tag, k, info[k], ref[k])
    # end to end, each side with its own exp()
    assert abs(info["energy_final"] - ref_own["energy_final"]) <= 1e-9 * max(1.0, abs(ref_own["energy_final"]))
    assert (labels != ref_own["labels"]).mean() <= 1e-3
    return info, ref


GRIDS = [
    (40, 28, 9), (1, 1, 9), (1, 37, 9), (53, 1, 9), (7, 16, 9), (3, 17, 9), (100, 70, 9),
    (33, 35, 4), (20, 33, 15), (21, 19, 1),
]


@pytest.mark.parametrize("w,h,K", GRIDS, ids=["%dx%d-K%d" % g for g in GRIDS])
def test_trws_on_synthetic_peaks(hip_ctx, w, h, K):
    peaks, mask = mrf_cases.peaks_case(w, h, K=K, seed=w * 131 + h, fill=0.6)
    _check_against_oracle(hip_ctx, peaks, mask, dict(), "default %dx%d" % (w, h))
    # a fixed number of sweeps (the stopping test never says stop): messages after 3 sweeps
    info, _ = _check_against_oracle(hip_ctx, peaks, mask, dict(min_energy_drop=-1.0, max_iters=2), "3 sweeps %dx%d" % (w, h))
    assert info["iterations"] == 3


def test_trws_many_bands_under_load(hip_ctx):
    """30 bands in flight: every hand-off between workgroups is exercised with all of them running."""
    peaks, mask = mrf_cases.peaks_case_fast(640, 480, K=9, seed=11)
    info, ref = _check_against_oracle(hip_ctx, peaks, mask, dict(), "640x480")
    assert 1 <= info["iterations"] <= 51 and info["energy_final"] < info["energy_initial"]
    peaks, mask = mrf_cases.peaks_case_fast(500, 333, K=9, seed=12, fill=0.4)
    _check_against_oracle(hip_ctx, peaks, mask, dict(min_energy_drop=-1.0, max_iters=1), "500x333")


def test_mrf_after_the_initial_estimate(hip_ctx):
    """computeInitialEstimate with USE_MRF: peaks of a real view, then the optimiser, against the oracle's."""
    import torch
    case = cases.get_mvs("mvs_geodesic", w=48, h=36, D=20, nviews=3)
    imgs, ocams, op = cases.oracle_inputs(case)
    neigh = O.mvs_neighbours(ocams, op)
    _, want_pk, _ = O.mvs_initial_estimate(imgs, ocams, 0, neigh[0], op, want_peaks=True)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    pk = torch.zeros((36, 48, p.top_k, 2), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.mvs_initial_estimate(0, neigh[0], p, peaks_dev=pk.data_ptr())
    hip_ctx.synchronize()
    got_pk = pk.cpu().numpy()
    info = hip_ctx.mvs_mrf_estimate(0, p.top_k, pk.data_ptr())
    labels, D, M = hip_ctx.mvs_mrf_state(48, 36, p.top_k)
    depth = hip_ctx.download_depth(0)
    mask = case["views"][0][1]
    ref = O.mvs_mrf(got_pk, mask, O.mrf_params(), data_costs=D)
    assert info["iterations"] == ref["iterations"] and np.array_equal(labels, ref["labels"])
    sel = mask == 1
    assert np.array_equal(depth[sel].view(np.uint64), ref["depth"][sel].view(np.uint64))
    assert np.isinf(depth[~sel]).all()                      # computeInitialEstimate left INF there (multiviewstereo.cpp:539)
    # and from the oracle's own peaks: the same picture up to the 1e-9 the peaks agree to
    ref2 = O.mvs_mrf(want_pk, mask, O.mrf_params())
    assert (labels != ref2["labels"]).mean() <= 0.01
    assert np.isfinite(depth[sel]).any()


def test_mrf_argument_errors(hip_ctx):
    import torch
    peaks, mask = mrf_cases.peaks_case(12, 9, K=3, seed=2)
    _upload_blank_view(hip_ctx, 0, mask)
    pk = torch.from_numpy(peaks).to("cuda:0")
    with pytest.raises(capi.StereoHipError) as e:
        hip_ctx.mvs_mrf_estimate(0, 16, pk.data_ptr())
    assert e.value.code == capi.SRH_E_UNSUPPORTED
    with pytest.raises(capi.StereoHipError):
        hip_ctx.mvs_mrf_estimate(0, 0, pk.data_ptr())
    with pytest.raises(capi.StereoHipError):
        hip_ctx.mvs_mrf_estimate(41, 3, pk.data_ptr())


def test_mrf_of_several_views_side_by_side(hip_ctx):
    """srh_mvs_initial_estimate_peaks + srh_mvs_mrf_estimate_views: the MRF stage of all views in flight together
    (one stream and one scratch buffer per view) gives each view exactly what the one-view entry point gives it."""
    case = cases.get_mvs("mvs_geodesic", w=48, h=36, D=20, nviews=3)
    imgs, ocams, op = cases.oracle_inputs(case)
    neigh = O.mvs_neighbours(ocams, op)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    one = []
    for v in range(3):
        info = hip_ctx.mvs_initial_estimate_mrf(v, neigh[v], p)
        one.append((info, hip_ctx.download_depth(v)))
    for v in range(3):
        hip_ctx.upload_depth(v, np.full((36, 48), -9.0))
        hip_ctx.mvs_initial_estimate_peaks(v, neigh[v], p)
    infos = hip_ctx.mvs_mrf_estimate_views([0, 1, 2])
    for v in range(3):
        assert infos[v] == one[v][0], (v, infos[v], one[v][0])
        assert np.array_equal(hip_ctx.download_depth(v).view(np.uint64), one[v][1].view(np.uint64))
    # a subset, in another order, with its own stopping parameters
    infos = hip_ctx.mvs_mrf_estimate_views([2, 0], capi.mrf_params(min_energy_drop=-1.0, max_iters=1))
    assert [i["iterations"] for i in infos] == [2, 2]
    with pytest.raises(capi.StereoHipError):
        hip_ctx.mvs_mrf_estimate_views([0, 0])
    with pytest.raises(capi.StereoHipError):
        hip_ctx.mvs_mrf_estimate_views([0, 7])                   # a slot without an image / without peaks


def test_trws_at_the_size_of_a_c4_view(hip_ctx):
    """1280x960, K = 9 (60 bands): two sweeps, every label, message and depth against the oracle (about 10 s of it)."""
    peaks, mask = mrf_cases.peaks_case_fast(1280, 960, K=9, seed=21)
    info, ref = _check_against_oracle(hip_ctx, peaks, mask, dict(min_energy_drop=-1.0, max_iters=1), "1280x960")
    assert info["iterations"] == 2 and info["energy_final"] < info["energy_initial"]
