"""The fused row-aligned TwoView kernel (srh_fused.hip: geometry + cost + WTA per 16-pixel tile, nothing
staged in device memory) against the three-kernel form (srh_dense.hip), the general kernels and the
oracle: identical bits and identical reference-evaluation counts on every rectified scene -- masks,
odd widths, candidate ranges running off either image border, scaled images, both weight kinds and
both radii, tiny and wide label counts."""
import os

import numpy as np
import pytest

import cases
import oracle_ffi as O

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fused_debug")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _run(ctx, ref, oth, p, fused, generic=0):
    ctx.set_option("fused", fused)
    ctx.set_option("force_generic", generic)
    try:
        ctx.twoview_wta(ref, oth, p)
        st = ctx.stats()
        return ctx.download_depth(ref), st
    finally:
        ctx.set_option("fused", 0)                   # the library's default
        ctx.set_option("force_generic", 0)


def _report(tag, got, want):
    """Write what differs where to gpurun_out/ (merged back to the build container) and return a summary."""
    os.makedirs(OUT, exist_ok=True)
    bad = _bits(got) != _bits(want)
    idx = np.argwhere(bad)
    lines = ["%s: %d of %d pixels differ" % (tag, bad.sum(), bad.size)]
    for (y, x) in idx[:40]:
        lines.append("  (x=%d, y=%d): got %r want %r" % (x, y, got[y, x], want[y, x]))
    cols = np.bincount(idx[:, 1], minlength=got.shape[1]) if len(idx) else np.zeros(1, int)
    lines.append("  columns with mismatches: " + " ".join("%d:%d" % (c, n) for c, n in enumerate(cols) if n))
    with open(os.path.join(OUT, tag.replace("/", "_") + ".txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return "\n".join(lines[:12])


SCENES = [
    ("geodesic_rect", dict()),
    ("geodesic_rect", dict(w=50, h=21, D=9)),                 # width not a multiple of the tile
    ("geodesic_rect", dict(w=33, h=17, D=24)),                # candidate ranges run off both image borders
    ("adaptive_rect", dict(w=96, h=40, D=40)),
    ("geodesic_r2", dict()),
    ("adaptive_masks", dict()),
    ("geodesic_masks", dict(w=70, h=44, D=20)),
    ("geodesic_scaled", dict()),
    ("geodesic_rect", dict(w=80, h=12, D=3)),                 # fewer labels than lanes per pixel
    ("geodesic_rect", dict(w=160, h=10, D=120)),              # > 64 columns: every lane has a block, several chunks of joints
]


@pytest.mark.parametrize("name,over", SCENES, ids=["%s-%s" % (n, "-".join("%s%s" % kv for kv in o.items())) for n, o in SCENES])
def test_fused_equals_three_kernel_form_general_kernels_and_oracle(hip_ctx, name, over):
    case = cases.get_twoview(name, **over)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    for ref, oth in ((0, 1), (1, 0)):
        tag = "%s_%s_%d" % (name, "_".join("%s%s" % kv for kv in over.items()), ref)
        want, diag = O.twoview_wta(imgs[ref], imgs[oth], ocams[ref], ocams[oth], op, want_diag=True)
        fused, st_f = _run(hip_ctx, ref, oth, p, 1)
        three, st_3 = _run(hip_ctx, ref, oth, p, 0)
        gen, st_g = _run(hip_ctx, ref, oth, p, 1, generic=2)
        assert st_3["used_dense_path"] and not st_3["used_fused_kernel"]
        assert not st_g["used_dense_path"]
        assert st_f["used_fused_kernel"], "the fused kernel gave up on a rectified scene (%s)" % tag
        assert np.array_equal(_bits(three), _bits(gen)), "three-kernel form vs general: " + _report(tag + "_3g", three, gen)
        assert np.array_equal(_bits(fused), _bits(gen)), "fused vs general: " + _report(tag + "_fg", fused, gen)
        ok, msg, _ = cases.compare_depth(fused, want, 1e-9)
        assert ok, msg
        assert st_f["n_pixels"] == st_g["n_pixels"] == int((case["views"][ref][1] == 1).sum())
        assert st_f["n_eval"] == st_g["n_eval"] == diag["n_eval"], (st_f["n_eval"], st_3["n_eval"], st_g["n_eval"], diag["n_eval"])


def test_fused_band_split_and_row_range(hip_ctx):
    """Rows [y0,y1) only, and a band budget that cuts the image into many bands: same bits."""
    case = cases.get_twoview("geodesic_masks", w=64, h=40, D=16)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    full, st = _run(hip_ctx, 0, 1, p, 1)
    assert st["used_fused_kernel"]
    hip_ctx.set_option("band_budget_mb", 1)
    try:
        banded, st2 = _run(hip_ctx, 0, 1, p, 1)
    finally:
        hip_ctx.set_option("band_budget_mb", 32768)
    assert st2["used_fused_kernel"] and np.array_equal(_bits(full), _bits(banded))
    hip_ctx.upload_depth(0, np.full_like(full, -7.0))
    hip_ctx.twoview_wta(0, 1, p, 11, 23)
    part = hip_ctx.download_depth(0)
    assert np.array_equal(_bits(part[11:23]), _bits(full[11:23]))
    assert (part[:11] == -7.0).all() and (part[23:] == -7.0).all()


def test_fused_gives_way_when_the_range_does_not_fit(hip_ctx):
    """More labels or a wider candidate range than an LDS cost row holds: the other kernels run, same result as the
    oracle (the host plan refuses, or the device flags the overflow and the pass is repeated)."""
    case = cases.get_twoview("geodesic_rect", w=400, h=6, D=300, radius=2)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    got, st = _run(hip_ctx, 0, 1, p, 1)
    assert st["used_dense_path"] and not st["used_fused_kernel"]
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    ok, msg, _ = cases.compare_depth(got, want, 1e-9)
    assert ok, msg
