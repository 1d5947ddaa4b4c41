"""The CPU oracle against (a) the known-answer vectors of SURVEY.md 8(c), produced by the
reference's own unmodified sources, (b) the committed fixtures tests/golden/ref_pieces.npz
generated from those sources (tests/golden/make_fixtures.py), and (c) -- when
oracle/_ref/libref_pieces.so is present -- the compiled reference pieces, live, on random input.
All comparisons are bit-exact (np.array_equal on float64)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ffi as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _T():
    w, h = 6, 5
    T = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        for x in range(w):
            T[y, x] = ((17 * x + 3 * y) % 256, (5 * x + 29 * y) % 256, (7 * x * y) % 256, 255)
    return O.OImage(T)


def test_survey_line_iterator_vectors():
    lp = lambda *a, **k: O.line_points(*a, **k).tolist()
    assert lp(0, 0, 5, 2) == [[0, 0], [1, 0], [2, 1], [3, 1], [4, 2], [5, 2]]
    assert lp(5, 2, 0, 0) == [[0, 0], [1, 0], [2, 1], [3, 1], [4, 2], [5, 2]]      # endpoints swapped internally
    assert lp(0, 0, 2, 5) == [[0, 0], [0, 1], [1, 2], [1, 3], [2, 4], [2, 5]]      # steep
    assert lp(3, 3, 3, 3) == [[3, 3]]
    assert lp(2.9, 1.2, -0.7, 1.9) == [[0, 1], [1, 1], [2, 1]]                     # truncation toward zero
    assert lp(10.5, 4.5, 7.2, 4.5) == [[7, 4], [8, 4], [9, 4], [10, 4]]
    assert lp(-3, 1, 4, 6, True, 5, 5) == [[0, 3], [1, 4]]                         # Cohen-Sutherland, int division
    assert lp(-5, -5, -1, -2, True, 5, 5) == []
    assert lp(1, 1, 9, 3, True, 5, 5) == [[1, 1], [2, 1], [3, 1], [4, 1]]
    assert lp(6, 2, -2, 2, True, 5, 5) == [[0, 2], [1, 2], [2, 2], [3, 2], [4, 2]]


def test_survey_weight_vectors():
    img = _T()
    pa = O.params_twoview(window_radius=1, weight_kind=O.WEIGHT_ADAPTIVE)
    pg = O.params_twoview(window_radius=1, weight_kind=O.WEIGHT_GEODESIC)
    assert np.array_equal(O.weights(img, 3, 2, pa), np.array([
        [0.001072937946292734, 0.010121951025286229, 0.0060284229453273514],
        [0.038453023246379295, 1, 0.038453023246379295],
        [0.008408696610783202, 0.010121951025286229, 0.00042562830486553418]]))
    assert np.array_equal(O.weights(img, 3, 2, pg), np.array([
        [0.38004297908737988, 0.48742942467157674, 0.53673329379766721],
        [0.63656828060929127, 1, 0.63656828060929127],
        [0.5736718364940584, 0.48742942467157674, 0.31588128281709921]]))
    assert np.array_equal(O.weights(img, 0, 0, pa), np.array([
        [0, 0, 0], [0, 1, 0.062536523640957922], [0, 0.019931060251083248, 0.0024633202971322057]]))
    assert np.array_equal(O.weights(img, 0, 0, pg), np.array([
        [0, 0, 0], [0, 1, 0.70159364001835567], [0, 0.55816805423246652, 0.44876809137055268]]))


def test_survey_sample_and_gray_vectors():
    img = _T()
    L = O.lib()
    out = np.zeros(3)
    assert L.sro_image_sample(C.byref(img.c), 2.25, 1.5, O.dptr(out)) == 1 and out[0] == 42.75
    assert L.sro_image_sample(C.byref(img.c), 4.0, 1.0, O.dptr(out)) == 1 and out[0] == 71.0
    assert L.sro_image_sample(C.byref(img.c), 5.0, 1.0, O.dptr(out)) == 0          # x + 1 < w fails
    assert L.sro_to_gray(10, 20, 30) == 21.899999999999999


def test_golden_ref_pieces():
    g = np.load(os.path.join(GOLD, "ref_pieces.npz"))
    imgs = [O.OImage(im) for im in g["images"]]
    for kind, kname in ((O.WEIGHT_ADAPTIVE, "adaptive"), (O.WEIGHT_GEODESIC, "geodesic")):
        for r in (1, 2, 5):
            want = g["weights_%s_r%d" % (kname, r)]
            p = O.params_twoview(window_radius=r, weight_kind=kind)
            for ii, im in enumerate(imgs):
                for ci, (cx, cy) in enumerate(g["centres"]):
                    got = O.weights(im, int(cx), int(cy), p)
                    assert np.array_equal(got, want[ii, ci]), (kname, r, ii, cx, cy)
    h, w = g["images"].shape[1:3]
    for clip in (0, 1):
        pts, offs = g["line_points_clip%d" % clip], g["line_offsets_clip%d" % clip]
        for k, s in enumerate(g["line_segments"]):
            got = O.line_points(s[0], s[1], s[2], s[3], bool(clip), w, h)
            assert np.array_equal(got, pts[offs[k]:offs[k + 1]]), (clip, k, s)
    L = O.lib()
    out = np.zeros(3)
    for (x, y), want in zip(g["sample_xy"], g["sample_rgbv"]):
        v = L.sro_image_sample(C.byref(imgs[0].c), x, y, O.dptr(out))
        assert v == int(want[3])
        if v:
            assert np.array_equal(out, want[:3])
    assert L.sro_to_gray(10, 20, 30) == g["gray_of_10_20_30"][0]


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_pieces_random():
    R = O.ref()
    rng = np.random.default_rng(7)
    w, h = 31, 19
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    oi = O.OImage(img)
    rh = R.refp_image_create(O.u8ptr(oi.rgba), w, h)
    try:
        for kind in (0, 1):
            for r in (1, 3, 5):
                p = O.params_twoview(window_radius=r, weight_kind=kind)
                for _ in range(12):
                    cx, cy = int(rng.integers(-1, w + 1)), int(rng.integers(-1, h + 1))
                    b = np.empty((2 * r + 1, 2 * r + 1))
                    R.refp_weights(rh, cx, cy, r, kind, O.dptr(b))
                    assert np.array_equal(O.weights(oi, cx, cy, p), b), (kind, r, cx, cy)
        buf = np.empty((4096, 2), np.int32)
        for i in range(3000):
            c = rng.uniform(-40, 70, 4)
            if i % 3 == 0:
                c = np.round(c)
            clip = i % 2
            n = R.refp_line_points(c[0], c[1], c[2], c[3], clip, w, h, O.iptr(buf), 4096)
            assert np.array_equal(O.line_points(c[0], c[1], c[2], c[3], bool(clip), w, h), buf[:n])
    finally:
        R.refp_image_free(rh)


def _in_image(pts, w, h):
    pts = np.asarray(pts, dtype=np.int32).reshape(-1, 2)
    keep = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < w) & (pts[:, 1] < h)
    return pts[keep]


def test_bounded_walk_against_the_reference_fixture():
    """The TwoView curve rasterises with the 4-arg LineIterator and then drops every point outside the
    other image (mask.pixel() is INVALID there, twoviewstereo.cpp:1028-1040).  The oracle (and the
    kernels) walk a *bounded* form that jumps over the off-image prefix through the closed form of the
    Bresenham state (sr_oracle.c line_walk, bound_w > 0).  Pinned here: its in-image points equal the
    in-image points of the reference's own unbounded walk (fixture = output of util/lineiter.cpp)."""
    g = np.load(os.path.join(GOLD, "ref_pieces.npz"))
    h, w = g["images"].shape[1:3]
    pts, offs = g["line_points_clip0"], g["line_offsets_clip0"]
    nonempty = 0
    for k, s in enumerate(g["line_segments"]):
        want = _in_image(pts[offs[k]:offs[k + 1]], w, h)
        got = _in_image(O.line_points(s[0], s[1], s[2], s[3], w=w, h=h, bounded=True), w, h)
        assert np.array_equal(got, want), (k, s)
        nonempty += len(want) > 0
    assert nonempty > 50


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_bounded_walk_against_the_live_reference():
    """Same property against the compiled reference on random segments that start (far) off-image, at
    both slopes and in both directions -- the jump-ahead formula sr_oracle.c:337-346 is exercised with
    k from 1 to thousands of steps."""
    R = O.ref()
    rng = np.random.default_rng(11)
    buf = np.empty((1 << 16, 2), np.int32)
    checked = jumped = 0
    for i in range(6000):
        w, h = int(rng.integers(3, 90)), int(rng.integers(3, 70))
        far = (40, 400, 6000)[i % 3]
        a = rng.uniform(-far, far, 2)
        b = rng.uniform(-20, max(w, h) + 20, 2) if i % 2 else rng.uniform(-far, far, 2)
        if i % 5 == 0:
            a, b = np.round(a), np.round(b)
        if i % 7 == 0:
            a, b = b, a
        n = R.refp_line_points(a[0], a[1], b[0], b[1], 0, w, h, O.iptr(buf), buf.shape[0])
        assert n <= buf.shape[0]
        want = _in_image(buf[:n], w, h)
        got_all = O.line_points(a[0], a[1], b[0], b[1], w=w, h=h, bounded=True)
        got = _in_image(got_all, w, h)
        assert np.array_equal(got, want), (i, w, h, a, b)
        checked += len(want) > 0
        jumped += len(got_all) < n and len(want) > 0
    assert checked > 1500 and jumped > 500


def test_oracle_twoview_rectified_candidates():
    """Rectified geometry: every candidate of pixel (x,y) lies on row y and the distinct columns
    are x-d for the label disparities (SURVEY.md 8(a) pixel-centre note, 8(d))."""
    import cases
    case = cases.get_twoview("adaptive_rect", w=96, h=24, D=32)
    imgs, cams, p = cases.oracle_inputs(case)
    cur = O.epipolar_curve(cams[0], cams[1], imgs[1], p, False, 80, 12)
    assert len(cur) > 0 and (cur[:, 1] == 12).all()
    cols = np.unique(cur[:, 0])
    assert cols.max() == 80 - 8 - 1 or cols.max() == 80 - 8     # d0 = 8 (the last <1px fragment may drop)
    assert cols.min() == 80 - (8 + 32 - 1)
    assert np.array_equal(cols, np.arange(cols.min(), cols.max() + 1))
    # joints are evaluated twice by the reference: more curve points than distinct columns
    assert len(cur) > len(cols)
    # right->left pass: ascending inside a segment, segments descending (scan-order note)
    cur_r = O.epipolar_curve(cams[1], cams[0], imgs[0], p, False, 20, 12)
    assert (cur_r[:, 1] == 12).all() and cur_r[0, 0] > cur_r[-1, 0]
