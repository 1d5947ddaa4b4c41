"""geo_exp_n (srh_dense.hip): the geodesic windows kernel evaluates its 121 exponentials per pixel with the device library's
own exp sequence written step-major (n = rint(x / ln 2), r = x - n ln 2 in two pieces, a degree-11 Horner polynomial in fused
multiply-adds, ldexp) -- the same bits as exp() of the ROCm it was taken from, which the GPU tests check through depth maps
and cost rows.  Here, without a GPU: the sequence is replayed from the constants IN THE SHIPPED SOURCE with every fused
multiply-add emulated exactly (rationals, one rounding), and must stay within one unit in the last place of the correctly
rounded exponential over the kernel's argument range, be exactly 1 at 0 and underflow to 0 where the fast path relies on it
(no range selects: DESIGN.md section 4)."""
import math
import os
import re
import struct
from fractions import Fraction

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stereoreconstruction_amd", "csrc", "srh_dense.hip")


def _constants():
    txt = open(SRC).read()
    body = txt[txt.index("void geo_exp_n("):txt.index("// w[0..WS) <- exp(-w / sigma)")]
    bits = [int(h, 16) for h in re.findall(r"GEO_D\((0x[0-9a-f]{16})\)", body)]
    assert len(bits) == 13, bits                                   # 1/ln2, -ln2 hi, -ln2 lo, c11, c10, c9 .. c2
    return [struct.unpack("<d", struct.pack("<Q", b))[0] for b in bits]


def _fma(a, b, c):
    return float(Fraction(a)*Fraction(b) + Fraction(c))            # one rounding (float(Fraction) rounds to nearest even)


def _replay(x, K):
    inv_ln2, nl2h, nl2l, c11, c10, *rest = K
    n = float(round(x*inv_ln2))                                    # v_rndne_f64 (Python's round: half to even)
    r = _fma(nl2h, n, x)
    r = _fma(nl2l, n, r)
    p = _fma(c11, r, c10)
    for c in rest:
        p = _fma(r, p, c)
    p = _fma(r, p, 1.0)
    p = _fma(r, p, 1.0)
    try:
        return math.ldexp(p, int(n))
    except OverflowError:
        return math.inf


def _ulp(v):
    return math.ulp(v) if v > 0 else 5e-324


def test_constants_are_the_librarys():
    K = _constants()
    assert K[0] == 1/math.log(2) or abs(K[0] - 1/math.log(2)) <= math.ulp(K[0])
    assert abs(-K[1] - math.log(2)) < 1e-10 and abs(K[2]) < 1e-16
    # the polynomial's coefficients are 1/k! to within a relative 5e-3 (a minimax fit, not the Taylor series: the top two terms
    # deviate by 1e-3 and 3e-3, the others by less than 1e-5)
    for k, c in zip(range(11, 1, -1), K[3:]):
        assert abs(c*math.factorial(k) - 1) < (5e-3 if k >= 10 else 1e-5), (k, c)


def test_replay_is_within_one_ulp_of_exp():
    K = _constants()
    worst = 0.0
    xs = [0.0, -0.0, -1e-300, -1e-17, -0.5*math.log(2), -math.log(2), -1.0, -50.0, -700.0, -744.0, -745.13]
    xs += [-(i*0.37 + (i*i % 17)*1e-3) for i in range(1, 400)]      # the kernel's range: -w / sigma, w up to the initial value
    xs += [-20000.0, -1075.0, -1076.0, -1e9]
    for x in xs:
        got = _replay(x, K)
        want = math.exp(x) if x > -745.2 else 0.0
        if want == 0.0 or got == 0.0:
            assert abs(got - want) <= 5e-324*2, (x, got, want)
            continue
        err = abs(got - want)/_ulp(want)
        worst = max(worst, err)
        assert err <= 1.0, (x, got, want, err)
    assert _replay(0.0, K) == 1.0 and _replay(-0.0, K) == 1.0
    # below -1075 the library selects 0; the kernel's fast path has no select and relies on ldexp: the same 0
    assert _replay(-1076.0, K) == 0.0 and _replay(-20000.0, K) == 0.0 and _replay(-2.0**30, K) == 0.0
    print("geo_exp_n replay: worst error %.3f ulp over %d arguments" % (worst, len(xs)))
