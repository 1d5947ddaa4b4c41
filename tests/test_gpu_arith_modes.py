"""Opt-in arithmetic modes of the dense TwoView path (option "arith").  The default (0) is the reference's arithmetic,
bit for bit.  Mode 1 ("fma") fuses the multiply-adds of the cost loops: costs move in their last bits, so a depth
can change only where two candidates were (nearly) tied.  The winner-mismatch rate against the exact mode is measured
here on C2 at full size and must stay tiny; where the winner is the same the depth is identical (the depth of a
winner comes from geometry, not from the cost)."""
import numpy as np
import pytest

from stereoreconstruction_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _pair(ctx, W, H, D, seed, wkind):
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, seed)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    return capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind)


@pytest.mark.parametrize("W,H,D,wkind,seed", [(640, 480, 64, capi.WEIGHT_ADAPTIVE, 0x5EED0002), (320, 240, 64, capi.WEIGHT_GEODESIC, 0x5EED0009)],
                         ids=["C2", "small-geodesic"])
def test_fma_mode_mismatch_rate(hip_ctx, W, H, D, wkind, seed):
    p = _pair(hip_ctx, W, H, D, seed, wkind)
    hip_ctx.set_option("arith", 0)
    hip_ctx.twoview_wta(0, 1, p)
    exact = hip_ctx.download_depth(0)
    assert hip_ctx.stats()["used_dense_path"]
    hip_ctx.set_option("arith", 1)
    try:
        hip_ctx.twoview_wta(0, 1, p)
        fma = hip_ctx.download_depth(0)
        assert hip_ctx.stats()["used_dense_path"]
    finally:
        hip_ctx.set_option("arith", 0)
    hip_ctx.twoview_wta(0, 1, p)
    again = hip_ctx.download_depth(0)
    assert np.array_equal(exact.view(np.uint64), again.view(np.uint64))          # the default is untouched
    differ = exact.view(np.uint64) != fma.view(np.uint64)
    rate = differ.mean()
    assert rate < 2e-3, "winner-mismatch rate of the fma mode: %.3g" % rate
    # a pixel either keeps its depth bit for bit or moves to another candidate / class: no "small" differences
    both = np.isfinite(exact) & np.isfinite(fma) & differ
    if both.any():
        rel = np.abs(exact[both] - fma[both]) / np.maximum(1.0, np.abs(exact[both]))
        assert (rel > 1e-9).all()


def test_arith_option_validation(hip_ctx):
    with pytest.raises(capi.StereoHipError):
        hip_ctx.set_option("arith", 2)
