"""exp(-w / sigma) of GeodesicWeight (geodesicweight.cpp:128-130) on the device: the geodesic kernels carry the device
library's exp sequence in their own source (srh_dense.hip, geo_exp_n -- the constants of this ROCm's ocml, written step-major
for six taps at once), and both of them share it.  ADVICE r5: nothing compared it bit for bit with exp(), which the adaptive
and generic paths use.  srh_debug_exp evaluates both, argument by argument: identical bits over the windows' range (and, for
the form without the library's range selects, over the range on which the kernels use it) -- so that a ROCm upgrade that
moves exp() is caught here.  The library's exp is within 1 ulp of the correctly rounded one: tests/test_geodesic_exp_replay.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_kernel_exp_sequence_is_the_device_librarys_exp(hip_ctx):
    rng = np.random.default_rng(606)
    xs = np.concatenate([
        -rng.uniform(0.0, 40.0, 400000),                          # the windows' everyday range: geodesic distances / sigma
        -rng.uniform(0.0, 760.0, 300000),                         # down to the underflow
        -np.exp(rng.uniform(np.log(1e-300), np.log(2.0 ** 30), 200000)),   # log-spaced: denormal arguments ... -2^30
        -np.arange(0.0, 1100.0, 0.25),                            # exact quarter steps across -708 (denormal results) and -745
        np.array([0.0, -0.0, -1e-320, -2.0 ** -1074, -708.3964185322641, -745.1332191019411, -745.1332191019412, -1075.0,
                  -1075.5, -2.0 ** 30, -2e4 / 50.0, -1e6 / 50.0]),   # 1e6: the windows' initial cost (never reached, never a NaN)
    ])
    got, lib = hip_ctx.debug_exp(xs)
    assert not np.isnan(got).any() and not np.isnan(lib).any()
    same = got.view(np.uint64) == lib.view(np.uint64)
    assert same.all(), (int((~same).sum()), xs[~same][:5], got[~same][:5], lib[~same][:5])
    # and it is exp: against the host's libm within 1 ulp (the replay test pins the sequence's accuracy; this one pins identity)
    ref = np.exp(xs)
    ulp = np.spacing(np.maximum(ref, 5e-324))
    assert (np.abs(got - ref) <= ulp).all()
    # positive arguments take the form with the range selects only (the kernels never see them): still the library's bits
    xp = rng.uniform(0.0, 710.0, 20000)
    g2, l2 = hip_ctx.debug_exp(xp)
    assert (g2.view(np.uint64) == l2.view(np.uint64)).all()
