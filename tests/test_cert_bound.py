"""The error bound the default (certified) arithmetic stands on, CHECKED instead of argued.

One candidate's cost is replayed on the CPU in the three arithmetics the kernels run -- the reference's operation order
(no contraction; stereo/twoviewstereo.cpp:909-977, stereo/multiviewstereo.cpp:113-189), the two fused sweeps (AR = 3,
MultiViewStereo staged kernel) and the ONE-PASS form (AR = 5) -- with every fused multiply-add emulated EXACTLY
(`fractions.Fraction`: the product and the sum are formed as rationals and rounded once, which is what v_fma_f64 does),
and compared with the real-number value of the formula (rationals, the square root in 60-digit decimals).  Windows are
random and adversarial and PLACED AT THE CERTIFICATION THRESHOLDS: sum3 within a factor 1 ... 4 of sigma3(sum2) (the
smallest sum3 the bound covers), Q3/sum3 within 1 ... 4 of zmax^2 (one-pass form), weights down at the cut-off, grays 0 and
255.  Asserted, with the constants srh_cert_bound() hands out (the very numbers the kernels use):

  * each arithmetic is within half of k1/B + k2/A + k3 of the real-number value, the fused within the whole of it of the
    reference's; a candidate the kernel would certify (sum3 >= sigma3) differs by at most e0;
  * the one-pass value -- its finish without IEEE division or square root: m = P*fl(1/tw), 1/sqrt by a seed of ANY accuracy
    that passes the finish's own residual test |1 - x*y0^2| <= 2^-20 and one third-order Newton step -- is within
    255*1.01*gamma_(T+6)*(3*z + 2.002*z^2) + 2600u of the real-number value, and a candidate it certifies (sum3 >= sigma3,
    Q3 <= zmax2*sum3, residual test) differs from the reference's by at most e0;
  * the same for MultiViewStereo's score (25 taps, scale 1, e0 = 2^-36, two partial sums per sum).

No GPU: srh_cert_bound / srh_cert_sigma3 are host arithmetic of the shipped library."""
import decimal
import math
from fractions import Fraction as F

import numpy as np
import pytest

from stereoreconstruction_amd import capi

U = 2.0 ** -53
decimal.getcontext().prec = 60


def gamma(k):
    return k * U / (1.0 - k * U)


def fma(a, b, c):
    """fl(a*b + c) with ONE rounding (Fraction -> float is correctly rounded)."""
    return float(F(a) * F(b) + F(c))


def dsqrt(fr):
    return decimal.Decimal(fr.numerator).sqrt() / decimal.Decimal(fr.denominator).sqrt()


def shared_constants(w, l):
    """meanL, totalWeight, sum2 and a_t as the weights kernels leave them (the same bits in every arithmetic)."""
    acc = 0.0
    tw = 0.0
    for wt, lt in zip(w, l):
        acc += wt * lt
        tw += wt
    mL = acc / tw
    a = [wt * lt - mL for wt, lt in zip(w, l)]
    s2 = 0.0
    for at in a:
        s2 += at * at
    return mL, tw, s2, a


def real_value(w, l, r, mL, tw, s2, scale):
    """The formula over the reals, with the shared float constants as given numbers."""
    fw, fl_, fr = [F(x) for x in w], [F(x) for x in l], [F(x) for x in r]
    mu = sum(a * b for a, b in zip(fw, fr)) / F(tw)
    beta = [a * b - mu for a, b in zip(fw, fr)]
    alpha = [a * b - F(mL) for a, b in zip(fw, fl_)]
    S1 = sum(a * b for a, b in zip(alpha, beta))
    S3 = sum(b * b for b in beta)
    if S3 == 0:
        return None, S1, S3
    rho = decimal.Decimal(abs(S1).numerator) / decimal.Decimal(abs(S1).denominator) / dsqrt(F(s2) * S3)
    if scale == 255:
        return float(decimal.Decimal(255) * (1 - rho)), S1, S3
    return float(rho if S1 >= 0 else -rho), S1, S3


def cost_reference(w, r, a, tw, s2, scale):
    acc = 0.0
    for wt, rt in zip(w, r):
        acc += wt * rt
    mR = acc / tw
    s1 = s3 = 0.0
    for wt, rt, at in zip(w, r, a):
        b = wt * rt - mR
        s1 += at * b
        s3 += b * b
    if scale == 255:
        return 255 * (1.0 - abs(s1) / math.sqrt(s2 * s3)), s3
    return s1 / math.sqrt(s2 * s3), s3


def cost_two_fused_sweeps(w, l, r, mL, tw, s2, a_shared, scale, partial=1):
    """AR = 3 of the TwoView kernels (a_t fused as well); MultiViewStereo: a_t shared, `partial` partial sums per sum."""
    accs = [0.0] * partial
    for t, (wt, rt) in enumerate(zip(w, r)):
        accs[t % partial] = fma(wt, rt, accs[t % partial])
    acc = accs[0]
    for k in range(1, partial):
        acc += accs[k]
    # (MultiViewStereo's certified kernel multiplies by the refined reciprocal of tw -- within an ulp of 1/tw -- instead of dividing)
    mR = acc / tw if scale == 255 else acc * (1.0 / tw)
    s1 = [0.0] * partial
    s3 = [0.0] * partial
    for t, (wt, lt, rt) in enumerate(zip(w, l, r)):
        a = fma(wt, lt, -mL) if scale == 255 else a_shared[t]
        b = fma(wt, rt, -mR)
        s1[t % partial] = fma(a, b, s1[t % partial])
        s3[t % partial] = fma(b, b, s3[t % partial])
    S1, S3 = s1[0], s3[0]
    for k in range(1, partial):
        S1 += s1[k]
        S3 += s3[k]
    if scale == 255:
        return 255 * (1.0 - abs(S1) / math.sqrt(s2 * S3)), S3
    return S1 / math.sqrt(s2 * S3), S3


def cost_one_pass(w, l, r, mL, tw, s2, seed_err=0.0):
    """AR = 5 (srh_strip.hip / srh_dense.hip / srh_rows.hip): P, Q, U in one sweep, the sums recovered afterwards."""
    T = float(len(w))
    P = Q = Us = SA = 0.0
    for wt, lt, rt in zip(w, l, r):
        q = rt * rt
        a = fma(wt, lt, -mL)
        c = a * wt
        d = wt * wt
        SA += a
        P = fma(wt, rt, P)
        Q = fma(c, rt, Q)
        Us = fma(d, q, Us)
    # the finish (srh_internal.hpp::onepass_finish, round 6): no IEEE division, no IEEE square root
    itw = 1.0 / tw                                   # pconst slot 3, correctly rounded by the weights kernels
    m = P * itw
    p2 = P + P
    s3 = fma(-m, fma(-T, m, p2), Us)
    s1 = fma(-m, SA, Q)
    q3 = fma(m, fma(T, m, p2), Us)
    if not s3 > 0:
        return float("nan"), s3, q3, False
    x = s2 * s3
    # v_rsq_f64 is a seed the certificate trusts for nothing: ANY y0 whose residual passes |e| <= 2^-20 must do.  The replay
    # takes the seed at `seed_err` relative distance from 1/sqrt(x) (the caller sweeps it up to the edge of the test).
    y0 = float(1 / dsqrt(F(x))) * (1.0 + seed_err)
    e = fma(-(x * y0), y0, 1.0)
    y1 = fma(y0 * e, fma(0.375, e, 0.5), y0)
    v = fma(-255.0, abs(s1) * y1, 255.0)
    return v, s3, q3, abs(e) <= 2.0 ** -20


def windows(rng, T, n):
    """(w, l, base, noise): weight patterns from uniform to geodesic-like with taps at the cut-off; grays smooth, binary, extreme."""
    for k in range(n):
        kind = k % 6
        if kind == 0:
            w = np.ones(T)
        elif kind == 1:
            w = np.exp(-rng.random(T) * 12.0)                     # geodesic-like: down to 6e-6
        elif kind == 2:
            w = np.where(rng.random(T) < 0.5, 1.0, 1.0000001e-10)  # half the taps just above the cut-off (weight_cutoff = 1e-10)
        elif kind == 3:
            w = 1.0 - rng.random(T) * 1e-3                        # nearly uniform
        elif kind == 4:
            w = np.exp(-np.abs(np.arange(T) - T // 2) / 7.0)      # smooth fall-off
        else:
            w = rng.random(T) ** 4 + 1e-9
        lk = k % 4
        if lk == 0:
            l = rng.random(T) * 255.0
        elif lk == 1:
            l = np.where(rng.random(T) < 0.5, 0.0, 255.0)
        elif lk == 2:
            l = 200.0 + rng.random(T) * 3.0
        else:
            l = np.round(rng.random(T) * 255.0)
        base = [255.0 * rng.random(), 0.0, 255.0, 128.0][k % 4 if k % 3 else 0]
        noise = rng.standard_normal(T)
        yield [float(x) for x in w], [float(x) for x in l], float(base), noise


def right_window(base, noise, s):
    return [float(min(255.0, max(0.0, base + s * n))) for n in noise]


def place_sum3(w, base, noise, tw, target):
    """noise amplitude s for which sum3 (float estimate) of r = clip(base + s*noise) is `target`; None when out of reach."""
    def s3_of(s):
        r = right_window(base, noise, s)
        mu = sum(a * b for a, b in zip(w, r)) / tw
        return sum((a * b - mu) ** 2 for a, b in zip(w, r))
    lo, hi = 0.0, 300.0
    if not (s3_of(hi) > target > s3_of(lo)):
        return None
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if s3_of(mid) < target:
            lo = mid
        else:
            hi = mid
    return hi


@pytest.mark.parametrize("radius", [5, 2])
def test_twoview_bound_at_the_certification_thresholds(radius):
    p = capi.params_twoview(window_radius=radius)
    cb = capi.cert_bound(p)
    assert cb["ok"] == 1 and cb["taps"] == (2 * radius + 1) ** 2
    T = cb["taps"]
    e0, k1, k2, k3, zmax2 = cb["e0"], cb["k1"], cb["k2"], cb["k3"], cb["zmax2"]
    one_pass_bound = lambda z2: 255 * 1.01 * (3 * gamma(T + 6) * math.sqrt(z2) + 2.002 * gamma(T + 6) * z2) + 2600 * U
    seed_errs = (0.0, 2.0 ** -29, -2.0 ** -26, 2.0 ** -22, -(2.0 ** -21) * 0.999, 2.0 ** -21 * 0.999, 2.0 ** -19)   # the last fails the residual test
    rng = np.random.default_rng(20260 + radius)
    worst = {"two_sweeps_vs_bound": 0.0, "one_pass_vs_bound": 0.0, "certified_two_sweeps_vs_e0": 0.0, "certified_one_pass_vs_e0": 0.0}
    n_cert2 = n_cert1 = n_unc = n = 0
    for w, l, base, noise in windows(rng, T, 36 if radius == 5 else 60):
        mL, tw, s2, a = shared_constants(w, l)
        if not s2 > 0:
            continue
        sig3 = capi.cert_sigma3(p, s2)
        targets = []
        if math.isfinite(sig3):
            targets += [sig3 * f for f in (0.5, 1.0000001, 1.3, 2.0, 4.0)]        # sum3 at 1 ... 4 x sigma3 (and one below)
        # Q3/sum3 at zmax2 / (1 ... 4) and beyond it: Q3 ~ 4*sum((w r)^2), so sum3 = Q3 / (zmax2 / f)
        q3_est = 4.0 * sum((a_ * max(base, 1.0)) ** 2 for a_ in w)
        targets += [q3_est * f / zmax2 for f in (0.25, 0.9, 1.0, 1.1, 2.0, 4.0)]
        targets += [10.0 ** rng.uniform(-3, 6)]
        for target in targets:
            s = place_sum3(w, base, noise, tw, target)
            if s is None:
                continue
            r = right_window(base, noise, s)
            V, S1, S3 = real_value(w, l, r, mL, tw, s2, 255)
            if V is None:
                continue
            ref, s3_ref = cost_reference(w, r, a, tw, s2, 255)
            f2, s3_f2 = cost_two_fused_sweeps(w, l, r, mL, tw, s2, a, 255)
            seed = seed_errs[n % len(seed_errs)]
            f1, s3_f1, q3, res_ok = cost_one_pass(w, l, r, mL, tw, s2, seed)
            if not (s3_ref > 0 and s3_f2 > 0):
                continue
            n += 1
            A, B = math.sqrt(s2), math.sqrt(min(s3_ref, s3_f2))
            # the hypotheses under which 1.01 covers the second-order terms: the bound is only ever USED for certified candidates
            # (sum3 >= sigma3(sum2), hence B >= 4.1 and A >= 0.022 at r = 5); below that it is checked where it still holds
            bound = k1 / B + k2 / A + k3
            if B >= 1.0 and A >= 0.01:
                assert abs(ref - V) <= bound / 2, (radius, "reference vs real", abs(ref - V), bound / 2, B, A)
                assert abs(f2 - V) <= bound / 2, (radius, "two sweeps vs real", abs(f2 - V), bound / 2, B, A)
                assert abs(f2 - ref) <= bound
                worst["two_sweeps_vs_bound"] = max(worst["two_sweeps_vs_bound"], abs(f2 - ref) / bound)
            if s3_f2 >= sig3:                                       # the kernel's own test, on its own sum3
                n_cert2 += 1
                assert bound <= e0 * (1 + 1e-9), (bound, e0)         # that is what sigma3 promises
                assert abs(f2 - ref) <= e0
                worst["certified_two_sweeps_vs_e0"] = max(worst["certified_two_sweeps_vs_e0"], abs(f2 - ref) / e0)
            if s3_f1 > 0 and not res_ok:
                assert seed == 2.0 ** -19                                   # only the seed beyond the residual test is refused
                n_unc += 1
            elif s3_f1 > 0:
                z2 = q3 / s3_f1
                if z2 < 1e9:
                    b1 = one_pass_bound(z2 * (1 + 1e-6))
                    assert abs(f1 - V) <= b1, (radius, "one pass vs real", abs(f1 - V), b1, z2)
                    worst["one_pass_vs_bound"] = max(worst["one_pass_vs_bound"], abs(f1 - V) / b1)
                if s3_f1 >= sig3 and s3_f1 * zmax2 >= q3:           # certified by the kernel's own test
                    n_cert1 += 1
                    assert one_pass_bound(z2) <= e0 / 2 * (1 + 1e-6)
                    assert abs(f1 - ref) <= e0, (radius, abs(f1 - ref), e0, z2, s3_f1, sig3)
                    worst["certified_one_pass_vs_e0"] = max(worst["certified_one_pass_vs_e0"], abs(f1 - ref) / e0)
                else:
                    n_unc += 1
    print("TwoView r=%d: %d candidates replayed (certified: two sweeps %d, one pass %d; left uncertified by the one-pass test %d); "
          "worst observed / bound: %s" % (radius, n, n_cert2, n_cert1, n_unc, {k: float("%.3g" % v) for k, v in worst.items()}))
    assert n_cert2 >= 20 and n_cert1 >= 20 and n_unc >= 10         # both sides of the thresholds were visited
    assert max(worst.values()) <= 1.0


def test_multiview_score_bound_at_the_certification_threshold():
    p = capi.params_mvs()
    cb = capi.cert_bound(p, mvs=True)
    assert cb["ok"] == 1 and cb["taps"] == 25 and cb["e0"] == 2.0 ** -36
    T, e0, k1, k2, k3 = cb["taps"], cb["e0"], cb["k1"], cb["k2"], cb["k3"]
    rng = np.random.default_rng(77)
    worst_bound = worst_e0 = 0.0
    n = n_cert = 0
    for w, l, base, noise in windows(rng, T, 120):
        mL, tw, s2, a = shared_constants(w, l)
        if not s2 > 0:
            continue
        sig3 = capi.cert_sigma3(p, s2, mvs=True)
        targets = [10.0 ** rng.uniform(-2, 6)]
        if math.isfinite(sig3):
            targets += [sig3 * f for f in (0.7, 1.0000001, 1.5, 2.5, 4.0)]
        for target in targets:
            s = place_sum3(w, base, noise, tw, target)
            if s is None:
                continue
            r = right_window(base, noise, s)
            V, S1, S3 = real_value(w, l, r, mL, tw, s2, 1)
            if V is None:
                continue
            ref, s3_ref = cost_reference(w, r, a, tw, s2, 1)
            fz, s3_f = cost_two_fused_sweeps(w, l, r, mL, tw, s2, a, 1, partial=2)   # MS_CA = 2 partial sums per sum
            if not (s3_ref > 0 and s3_f > 0):
                continue
            n += 1
            A, B = math.sqrt(s2), math.sqrt(min(s3_ref, s3_f))
            bound = k1 / B + k2 / A + k3
            if B >= 1.0 and A >= 0.01:
                assert abs(ref - V) <= bound / 2 and abs(fz - V) <= bound / 2, (abs(ref - V), abs(fz - V), bound / 2)
                worst_bound = max(worst_bound, abs(fz - ref) / bound)
            if s3_f >= sig3:
                n_cert += 1
                assert bound <= e0 * (1 + 1e-9)
                assert abs(fz - ref) <= e0
                worst_e0 = max(worst_e0, abs(fz - ref) / e0)
    print("MultiViewStereo: %d candidates replayed, %d certified; worst |fused - reference| / bound %.3g, / e0 %.3g" % (n, n_cert, worst_bound, worst_e0))
    assert n_cert >= 50 and worst_bound <= 1.0 and worst_e0 <= 1.0


def test_bound_constants_are_the_documented_ones():
    """r = 5: k1 = 6.04e-8, k2 = 3.2e-10, k3 = 1.4e-11, zmax^2 = 978, e0 = 2^-26; a candidate needs
    sqrt(sum3) >= 4.1 at large sum2; parameters outside the host checks switch the bound off."""
    p = capi.params_twoview()
    cb = capi.cert_bound(p)
    assert cb["e0"] == 2.0 ** -26 and abs(cb["k1"] - 6.04e-8) < 1e-10 and abs(cb["k2"] - 3.22e-10) < 1e-12
    assert abs(cb["k3"] - 1.41e-11) < 1e-13 and 975 < cb["zmax2"] < 982 and cb["m_hi"] == p.max_color_diff + cb["e0"]
    assert 4.0 < math.sqrt(capi.cert_sigma3(p, 1e6)) < 4.2
    assert capi.cert_sigma3(p, 0.0) == math.inf and capi.cert_sigma3(p, 1e-6) == math.inf
    for kw in (dict(wta_margin=-1e-3), dict(max_color_diff=1e7), dict(geodesic_sigma=0.0), dict(bad_ret=1e9)):
        assert capi.cert_bound(capi.params_twoview(**kw))["ok"] == 0, kw
