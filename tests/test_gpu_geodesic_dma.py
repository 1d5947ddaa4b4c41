"""The dense TwoView path's geodesic windows by the persistent LDS-DMA kernel (geodesic_dma_kernel, srh_dense.hip; option
`geodma`, default 1): four-row tiles fetched from planes with their borders written out, the next tile arriving while this
one is worked on.  Against the register-staged kernel (geodesic_reg_kernel, geodma = 0): the same depth bits, the same cost
rows (the reference's arithmetic on each kernel's windows: every weight bit counts), the same evaluation counts -- on
unmasked and masked pairs (holes, fully masked rows and tile columns, a mask that leaves whole tiles empty), widths that are
no multiple of 64, bands that do not start at row 0 (a small band budget), and after a view is uploaded again."""
import numpy as np
import pytest

from stereoreconstruction_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _pair(ctx, W, H, D, seed, mask_kind):
    L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, seed)
    rng = np.random.default_rng(seed)
    if mask_kind == "holes":
        for m in (ml, mr):
            for _ in range(40):
                x, y = int(rng.integers(0, W - 40)), int(rng.integers(0, H - 30))
                m[y:y + int(rng.integers(3, 30)), x:x + int(rng.integers(3, 40))] = 0
            m[H // 3:H // 3 + 9, :] = 0                          # whole rows (more than one tile's four)
            m[:, 640:640 + 128] = 0                              # two whole tile columns
    elif mask_kind == "sparse":
        for m in (ml, mr):
            keep = np.zeros_like(m)
            keep[40:H - 60, 100:W - 300] = 1
            m *= keep
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    return capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)


def _run(ctx, p, geodma, rows):
    ctx.set_option("geodma", geodma)
    try:
        out = []
        ctx.profile_reset(); ctx.profile_enable(True)
        for a, b in ((0, 1), (1, 0)):
            ctx.twoview_wta(a, b, p)
            out.append((ctx.download_depth(a), ctx.stats()))
        ctx.synchronize(); ctx.profile_enable(False)
        kernels = set(ctx.profile().keys())
        costs = [ctx.twoview_cost_rows(0, 1, p, y0, y1, 0)[0] for y0, y1 in rows]
        return out, kernels, costs
    finally:
        ctx.set_option("geodma", 1)


@pytest.mark.parametrize("W,H,mask_kind", [(1280, 704, "none"), (1250, 702, "holes"), (1344, 641, "sparse")])
def test_dma_windows_equal_the_register_staged_windows(hip_ctx, W, H, mask_kind):
    D = 24
    p = _pair(hip_ctx, W, H, D, 0x5EED0500 + W, mask_kind)
    rows = [(0, 8), (H // 3 - 4, H // 3 + 14), (H - 9, H)]
    new, kn, cn = _run(hip_ctx, p, 1, rows)
    old, ko, co = _run(hip_ctx, p, 0, rows)
    assert "geodesic_dma_kernel" in kn and "geodesic_reg_kernel" not in kn, kn
    assert "geodesic_reg_kernel" in ko and "geodesic_dma_kernel" not in ko, ko
    for d in range(2):
        (m1, s1), (m0, s0) = new[d], old[d]
        assert s1["used_dense_path"] and s0["used_dense_path"]
        assert np.array_equal(m1.view(np.uint64), m0.view(np.uint64)), (mask_kind, d)
        assert s1["n_eval"] == s0["n_eval"] and s1["n_pixels"] == s0["n_pixels"]
    for a, b in zip(cn, co):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), mask_kind


def test_dma_windows_in_row_bands_and_after_a_new_upload(hip_ctx):
    """bands that start anywhere (a band budget of half the image: the second band starts at a row that is no multiple of
    four) read the padded planes at their own offset; a new upload into the slot invalidates them"""
    W, H, D = 1920, 1080, 12
    p = _pair(hip_ctx, W, H, D, 0x5EED0510, "holes")
    whole, _, _ = _run(hip_ctx, p, 1, [])
    hip_ctx.set_option("band_budget_mb", 1100)
    try:
        hip_ctx.profile_reset()
        banded, _, _ = _run(hip_ctx, p, 1, [])
        calls = hip_ctx.profile()["geodesic_dma_kernel"][1]
    finally:
        hip_ctx.set_option("band_budget_mb", 32768)
    assert calls >= 4, calls                                      # two bands per direction
    for d in range(2):
        assert banded[d][1]["used_dense_path"]
        assert np.array_equal(whole[d][0].view(np.uint64), banded[d][0].view(np.uint64)), d
    # another pair into the same slots: the planes of the old images must not survive
    p2 = _pair(hip_ctx, W, H, D, 0x5EED0511, "none")
    new, _, _ = _run(hip_ctx, p2, 1, [])
    old, _, _ = _run(hip_ctx, p2, 0, [])
    for d in range(2):
        assert np.array_equal(new[d][0].view(np.uint64), old[d][0].view(np.uint64)), d
        assert not np.array_equal(new[d][0].view(np.uint64), whole[d][0].view(np.uint64))
