"""Device evidence for the error bound of the certified arithmetic: the cost rows the
scan looks up, fetched through the diagnostic C-ABI call srh_twoview_cost_rows in the reference's arithmetic (form 0) and in
the fused forms the default path runs (5: one pass, 3: two fused sweeps), compared entry by entry:

  * a stored fused value (certified by the kernel's own test) differs from the reference's by at most e0 -- the maximum
    observed |fused - exact| / e0 is printed and asserted <= 1;
  * a fused value stored as the clamp stands for a reference value that IS the clamp;
  * with the in-kernel redo (the default) no uncertified candidate is left behind: every entry the raw form marks NaN
    holds the reference's own bits, and a pixel the kernel evaluates in the reference's arithmetic throughout
    (cert_pixel_exact: flat window, unusable taps) has the reference's bits in every entry.

On every parity and adversarial case of tests/test_gpu_arith_modes.py, in all three kernel forms (per tile, 4- and
8-wave strip), and on C3 at full size band by band (the 8-wave strip kernel the headline runs)."""
import numpy as np
import pytest

from stereoreconstruction_amd import capi, synthetic
import test_gpu_arith_modes as am

pytestmark = pytest.mark.gpu

UNWRITTEN = np.uint64(0xFFFFFFFFFFFFFFFF)


def _valid(rng, cstride):
    """mask (rows, w, cstride): entry k of a pixel is a column of its range"""
    lo, hi = rng[..., 0], rng[..., 1]
    k = np.arange(cstride, dtype=np.int32)[None, None, :]
    return k <= (hi - lo)[..., None]


def _compare(exact, fused, valid, p, e0, tag):
    """-> (max |fused - exact| / e0 over the certified entries, number of certified / clamp / uncertified entries)"""
    eb, fb = exact.view(np.uint64), fused.view(np.uint64)
    # both forms write the same set of entries (a column of the range that neither writes: outside dense_cover_hi and not lazily filled)
    assert np.array_equal(eb[valid] == UNWRITTEN, fb[valid] == UNWRITTEN), tag
    live = valid & (eb != UNWRITTEN)
    unc = live & np.isnan(fused)
    clamp = live & (fused == p.max_color_diff)
    cert = live & ~unc & ~clamp
    assert np.all(exact[clamp] == p.max_color_diff), "%s: a fused clamp stands for a reference value below the clamp" % tag
    # values above the clamp by more than e0 come from the exact select forms (bad_ret): the same number
    big = cert & (fused > p.max_color_diff + e0)
    assert np.array_equal(eb[big], fb[big]), tag
    d = np.abs(fused[cert] - exact[cert])
    assert not np.isnan(d).any(), "%s: a certified fused value stands for an undefined reference value" % tag
    ratio = float(d.max() / e0) if d.size else 0.0
    return ratio, int(cert.sum()), int(clamp.sum()), int(unc.sum()), unc


def _rows_check(ctx, p, y0, y1, strip, tag, forms=(5, 3)):
    cb = capi.cert_bound(p)
    assert cb["ok"]
    e0 = cb["e0"]
    ctx.set_option("strip", strip)
    try:
        exact, rng, us = ctx.twoview_cost_rows(0, 1, p, y0, y1, 0)
        assert us == (strip != 0), tag
        valid = _valid(rng, exact.shape[2])
        worst = 0.0
        for form in forms:
            raw, rng2, _ = ctx.twoview_cost_rows(0, 1, p, y0, y1, form, raw=True)
            assert np.array_equal(rng, rng2)
            ratio, n_cert, n_clamp, n_unc, unc = _compare(exact, raw, valid, p, e0, "%s form %d raw" % (tag, form))
            worst = max(worst, ratio)
            assert ratio <= 1.0, "%s form %d: |fused - exact| = %.3g e0" % (tag, form, ratio)
            if strip:
                # the default: uncovered candidates re-evaluated in place -- nothing uncertified is left, what the raw form
                # could not certify holds the reference's bits
                redo, _, _ = ctx.twoview_cost_rows(0, 1, p, y0, y1, form)
                assert np.array_equal(redo.view(np.uint64)[unc], exact.view(np.uint64)[unc]), "%s form %d: a redone candidate is not the reference's" % (tag, form)
                r2, _, _, n_unc2, _ = _compare(exact, redo, valid, p, e0, "%s form %d" % (tag, form))
                assert n_unc2 == 0 and r2 <= 1.0, (tag, form, n_unc2, r2)
        return worst, n_cert, n_clamp, n_unc
    finally:
        ctx.set_option("strip", 1)


@pytest.mark.parametrize("name,over", am.CERT_CASES)
@pytest.mark.parametrize("strip", [0, 4, 8])
def test_fused_rows_within_the_bound_on_the_parity_cases(hip_ctx, name, over, strip):
    import cases
    case = cases.get_twoview(name, **over)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    h = case["views"][0][0].shape[0]
    worst, n_cert, n_clamp, n_unc = _rows_check(hip_ctx, p, 0, h, strip, "%s %s strip=%d" % (name, over, strip))
    print("cost rows %s strip=%d: max |fused - exact| = %.3g e0 over %d certified entries (%d clamps, %d uncertified)" % (name, strip, worst, n_cert, n_clamp, n_unc))
    assert n_cert > 0


@pytest.mark.parametrize("strip", [8, 0], ids=["strip", "per-tile"])
@pytest.mark.parametrize("wkind", [capi.WEIGHT_GEODESIC, capi.WEIGHT_ADAPTIVE], ids=["geodesic", "adaptive"])
@pytest.mark.parametrize("kind", ["periodic", "flat", "near_flat", "saturated_half", "two_level", "two_matches", "ramp"])
def test_fused_rows_within_the_bound_on_adversarial_images(hip_ctx, kind, wkind, strip):
    W, H, D = 192, 96, 40
    L, R, ml, mr = am._adversarial_pair(kind, W, H, D)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    hip_ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    hip_ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind)
    worst, n_cert, n_clamp, n_unc = _rows_check(hip_ctx, p, 0, H, strip, "%s strip=%d" % (kind, strip))
    print("cost rows %s / %s strip=%d: max |fused - exact| = %.3g e0 over %d certified entries (%d clamps, %d uncertified in the raw form)"
          % (kind, "geodesic" if wkind else "adaptive", strip, worst, n_cert, n_clamp, n_unc))


def test_fused_rows_within_the_bound_at_c3_size(hip_ctx):
    """C3 (1920 x 1080 x 256, geodesic r = 5), left -> right, the whole image band by band through the 8-wave strip kernel
    in its one-pass form (what the headline runs), raw against the reference's arithmetic."""
    W, H, D = 1920, 1080, 256
    p = am._pair(hip_ctx, W, H, D, 0x5EED0003, capi.WEIGHT_GEODESIC)
    cb = capi.cert_bound(p)
    e0 = cb["e0"]
    hip_ctx.set_option("strip", 8)
    worst, n_cert, n_unc, n_clamp = 0.0, 0, 0, 0
    try:
        for y0 in range(0, H, 90):
            exact, rng, us = hip_ctx.twoview_cost_rows(0, 1, p, y0, y0 + 90, 0)
            raw, _, _ = hip_ctx.twoview_cost_rows(0, 1, p, y0, y0 + 90, 5, raw=True)
            assert us
            valid = _valid(rng, exact.shape[2])
            ratio, c, k, u, _ = _compare(exact, raw, valid, p, e0, "C3 rows %d.." % y0)
            worst = max(worst, ratio)
            n_cert += c
            n_clamp += k
            n_unc += u
            del exact, raw, valid
    finally:
        hip_ctx.set_option("strip", 1)
    print("C3 cost rows, one-pass strip kernel: max |fused - exact| = %.4g e0 (%.3g absolute) over %d certified entries; %d clamps, %d uncertified"
          % (worst, worst * e0, n_cert, n_clamp, n_unc))
    assert worst <= 1.0 and n_cert > 4e8
