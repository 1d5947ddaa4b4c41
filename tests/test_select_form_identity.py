"""The arithmetic identity behind the blocked select forms of the strip and row-run cost kernels (srh_strip.hip
strip_select_block, srh_rows.hip general_block): "a skipped tap adds +0.0 to every sum" is evaluated
without a select per tap and candidate -- an unusable gray value becomes 0.0 with a flag 0.0 / 1.0 beside it, a tap unusable
on the reference side gets weight 0.0, what the reference skips is multiplied by that 0.0 -- and the window rows that lie
outside rows 0 .. H-2 of the reference image are left out altogether.  Replayed here in numpy float64, operation by operation
in the kernels' order (no contraction), against the guarded form they replace: the same bits of every sum and of the cost, on
random windows with NaN patterns (masks, image borders), weights at and below the cut-off, zero weights and zero grays."""
import numpy as np
import pytest

R, WS, NCB = 5, 11, 8
NR = NCB + 2*R


def _guarded(w, gl, rr, cutoff):
    """the form with a guard per tap and candidate (twoviewstereo.cpp:917-976 with every tap guarded; the kernels' form
    before round 5): rows x cols taps, NCB candidates sharing the row segment rr[row][col + j]"""
    rows = w.shape[0]
    mL = np.zeros(NCB); mR = np.zeros(NCB); tw = np.zeros(NCB)
    for row in range(rows):
        for col in range(WS):
            okl = (gl[row, col] == gl[row, col]) and (w[row, col] > cutoff)
            pl = w[row, col]*gl[row, col]
            for j in range(NCB):
                gr = rr[row, col + j]
                ok = okl and gr == gr
                pr = w[row, col]*gr
                mL[j] = mL[j] + (pl if ok else 0.0)
                mR[j] = mR[j] + (pr if ok else 0.0)
                tw[j] = tw[j] + (w[row, col] if ok else 0.0)
    with np.errstate(all="ignore"):
        mLd, mRd = mL/tw, mR/tw
    s1 = np.zeros(NCB); s2 = np.zeros(NCB); s3 = np.zeros(NCB)
    for row in range(rows):
        for col in range(WS):
            okl = (gl[row, col] == gl[row, col]) and (w[row, col] > cutoff)
            pl = w[row, col]*gl[row, col]
            for j in range(NCB):
                gr = rr[row, col + j]
                ok = okl and gr == gr
                with np.errstate(all="ignore"):
                    a = pl - mLd[j]
                    b = w[row, col]*gr - mRd[j]
                    ab, aa, bb = a*b, a*a, b*b
                s1[j] = s1[j] + (ab if ok else 0.0)
                s2[j] = s2[j] + (aa if ok else 0.0)
                s3[j] = s3[j] + (bb if ok else 0.0)
    return mL, mR, tw, s1, s2, s3


def _flags(w, gl, rr, cutoff, ra, rb):
    """the kernels' form since round 5: per-value flags, multiplications, window rows [ra, rb) only"""
    mL = np.zeros(NCB); mR = np.zeros(NCB); tw = np.zeros(NCB)
    for row in range(ra, rb):
        okr = rr[row] == rr[row]
        rv = np.where(okr, 1.0, 0.0)
        r0 = np.where(okr, rr[row], 0.0)
        for col in range(WS):
            okl = (gl[row, col] == gl[row, col]) and (w[row, col] > cutoff)
            w0 = w[row, col] if okl else 0.0
            pl0 = w[row, col]*gl[row, col] if okl else 0.0
            for j in range(NCB):
                mL[j] = mL[j] + pl0*rv[col + j]
                mR[j] = mR[j] + w0*r0[col + j]
                tw[j] = tw[j] + w0*rv[col + j]
    with np.errstate(all="ignore"):
        mLd, mRd = mL/tw, mR/tw
    s1 = np.zeros(NCB); s2 = np.zeros(NCB); s3 = np.zeros(NCB)
    for row in range(ra, rb):
        okr = rr[row] == rr[row]
        rv = np.where(okr, 1.0, 0.0)
        r0 = np.where(okr, rr[row], 0.0)
        for col in range(WS):
            g = gl[row, col]
            okl = (g == g) and (w[row, col] > cutoff)
            pl = w[row, col]*(g if g == g else 0.0)
            kl = 1.0 if okl else 0.0
            for j in range(NCB):
                k = kl*rv[col + j]
                with np.errstate(all="ignore"):
                    a = (pl - mLd[j])*k
                    b = (w[row, col]*r0[col + j] - mRd[j])*k
                s1[j] = s1[j] + a*b
                s2[j] = s2[j] + a*a
                s3[j] = s3[j] + b*b
    return mL, mR, tw, s1, s2, s3


def _window(rng, kind):
    w = np.exp(-rng.uniform(0, 12, (WS, WS)))                   # weights in (6e-6, 1]
    gl = rng.integers(0, 256, (WS, WS)).astype(np.float64)*0.59 + rng.integers(0, 256, (WS, WS))*0.11
    rr = rng.integers(0, 256, (WS, NR)).astype(np.float64)*0.3 + rng.integers(0, 256, (WS, NR))*0.59
    cutoff = 1e-4
    ra, rb = 0, WS
    if kind == "masks":
        gl[rng.random((WS, WS)) < 0.2] = np.nan
        rr[rng.random((WS, NR)) < 0.2] = np.nan
    elif kind == "cutoff":
        w[rng.random((WS, WS)) < 0.3] = cutoff                  # at the cut-off: not above it
        w[rng.random((WS, WS)) < 0.1] = 0.0
        gl[rng.random((WS, WS)) < 0.1] = 0.0
    elif kind == "top":                                          # a pixel on one of the image's first rows: window rows above the image
        ra = int(rng.integers(1, R + 1))
        gl[:ra] = np.nan; rr[:ra] = np.nan
    elif kind == "bottom":                                       # ... last rows (the last image row's tap values are NaN too)
        rb = int(rng.integers(R, WS))
        gl[rb:] = np.nan; rr[rb:] = np.nan
        rr[rng.random((WS, NR)) < 0.05] = np.nan
    elif kind == "left":                                         # candidates next to the left border: the segment's first columns outside
        k = int(rng.integers(1, 6))
        rr[:, :k] = np.nan
    elif kind == "nothing":                                      # no usable tap at all
        gl[:] = np.nan
    return w, gl, rr, cutoff, ra, rb


@pytest.mark.parametrize("kind", ["plain", "masks", "cutoff", "top", "bottom", "left", "nothing"])
def test_flag_form_gives_the_guarded_forms_bits(kind):
    rng = np.random.default_rng(sum(kind.encode()) + 0x5E1EC7)
    for _ in range(6):
        w, gl, rr, cutoff, ra, rb = _window(rng, kind)
        want = _guarded(w, gl, rr, cutoff)
        got = _flags(w, gl, rr, cutoff, ra, rb)
        live = ~(want[2] < 1e-10)          # (a candidate without weight gets bad_ret: its means are 0/0 and its second sweep is never looked at)
        for name, a, b in zip(("meanL", "meanR", "totalWeight", "sum1", "sum2", "sum3"), want, got):
            sel = live if name.startswith("sum") else slice(None)
            assert np.array_equal(a[sel].view(np.uint64), b[sel].view(np.uint64)), (kind, name, a, b)
        # and the cost the kernels store from them
        with np.errstate(all="ignore"):
            ca = np.where(want[2] < 1e-10, -1.0, 255*(1.0 - np.abs(want[3])/np.sqrt(want[4]*want[5])))
            cb = np.where(got[2] < 1e-10, -1.0, 255*(1.0 - np.abs(got[3])/np.sqrt(got[4]*got[5])))
        assert np.array_equal(ca.view(np.uint64), cb.view(np.uint64)), kind
