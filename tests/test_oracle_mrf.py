"""The oracle's TRW-S stage (MRF branch of computeInitialEstimate, multiviewstereo.cpp:481-516, 610-652).
PARITY UNPINNED: the reference's -lMRF library is not in its tree nor in this image (oracle/sr_oracle.h), so these
are the checks the published algorithm itself offers: exact on chains (TRW-S = min-sum dynamic programming on a
tree), lower bound <= optimum <= energy found on grids small enough to enumerate, energies recomputed term by term
from the reference's two cost functions, the reference's stopping rule and label -> depth rule."""
import itertools

import numpy as np
import pytest

import mrf_cases
import oracle_ffi as O


def _chain_optimum(peaks, m):
    """Viterbi over a 1-pixel-high or 1-pixel-wide strip."""
    L = O.lib()
    import ctypes as C
    h, w, K, _ = peaks.shape
    px = [peaks[y, x] for y in range(h) for x in range(w)]
    nl = K + 1
    cost = np.array([L.sro_mrf_data_cost(C.byref(m), K, O.dptr(np.ascontiguousarray(px[0])), l) for l in range(nl)])
    for i in range(1, len(px)):
        a, b = np.ascontiguousarray(px[i - 1]), np.ascontiguousarray(px[i])
        new = np.empty(nl)
        for lb in range(nl):
            new[lb] = min(cost[la] + L.sro_mrf_smooth_cost(C.byref(m), K, O.dptr(a), O.dptr(b), la, lb) for la in range(nl)) \
                + L.sro_mrf_data_cost(C.byref(m), K, O.dptr(b), lb)
        cost = new
    return cost.min()


@pytest.mark.parametrize("shape", [(17, 1), (1, 23), (2, 1), (1, 1)])
def test_exact_on_chains(shape):
    w, h = shape
    for seed in range(4):
        peaks, mask = mrf_cases.peaks_case(w, h, K=4, seed=seed, mask_frac=1.0)
        m = O.mrf_params(min_energy_drop=-1.0, max_iters=2)
        r = O.mvs_mrf(peaks, mask, m)
        want = _chain_optimum(peaks, m)
        assert abs(r["energy_final"] - want) <= 1e-12 * max(1, abs(want)), (shape, seed, r["energy_final"], want)
        assert abs(r["lower_bound"] - want) <= 1e-9
        assert abs(O.mrf_energy(peaks, r["labels"], m) - r["energy_final"]) <= 1e-12


@pytest.mark.parametrize("shape,K", [((3, 2), 2), ((2, 3), 2), ((3, 3), 1), ((4, 2), 1)])
def test_bound_optimum_and_found_energy_on_enumerable_grids(shape, K):
    w, h = shape
    hits = 0
    for seed in range(6):
        peaks, mask = mrf_cases.peaks_case(w, h, K=K, seed=100 + seed, mask_frac=1.0, fill=0.6)
        m = O.mrf_params(min_energy_drop=-1.0, max_iters=20)
        r = O.mvs_mrf(peaks, mask, m)
        best = min(O.mrf_energy(peaks, np.array(lab).reshape(h, w), m)
                   for lab in itertools.product(range(K + 1), repeat=w * h))
        assert r["lower_bound"] <= best + 1e-9
        assert best <= r["energy_final"] + 1e-12
        assert abs(O.mrf_energy(peaks, r["labels"], m) - r["energy_final"]) <= 1e-12
        hits += abs(r["energy_final"] - best) <= 1e-9
    assert hits >= 4                                  # TRW-S finds the optimum on most of these


def test_stopping_rule_and_initial_energy():
    """do { ... } while(prevEnergy - energy > 5 && numIters-- > 0)  (multiviewstereo.cpp:632-641)"""
    peaks, mask = mrf_cases.peaks_case(24, 18, K=5, seed=7)
    zero = np.zeros(peaks.shape[:2], dtype=np.int32)
    m = O.mrf_params()
    r = O.mvs_mrf(peaks, mask, m)
    assert abs(r["energy_initial"] - O.mrf_energy(peaks, zero, m)) <= 1e-9        # clearAnswer(): label 0 everywhere
    assert O.mvs_mrf(peaks, mask, O.mrf_params(max_iters=0))["iterations"] == 1
    assert O.mvs_mrf(peaks, mask, O.mrf_params(min_energy_drop=1e300))["iterations"] == 1
    assert O.mvs_mrf(peaks, mask, O.mrf_params(min_energy_drop=-1.0, max_iters=3))["iterations"] == 4
    assert 1 <= r["iterations"] <= 51 and r["lower_bound"] <= r["energy_final"] + 1e-9
    assert r["energy_final"] < r["energy_initial"]


def test_label_to_depth_rule():
    """label K -> INF; a placeholder peak (depth -1) -> INF; pixels outside the mask keep what they had (:645-652)."""
    peaks, mask = mrf_cases.peaks_case(20, 14, K=4, seed=3, mask_frac=0.7, fill=0.3)
    before = np.full(mask.shape, -7.0)
    r = O.mvs_mrf(peaks, mask, O.mrf_params(), depth=before)
    lab, d = r["labels"], r["depth"]
    K = peaks.shape[2]
    assert (d[mask == 0] == -7.0).all()
    for y, x in np.argwhere(mask == 1):
        z = np.inf if lab[y, x] == K else peaks[y, x, lab[y, x], 1]
        assert d[y, x] == (z if z > 0 else np.inf)
    assert (lab[mask == 0] == K).all()               # all-placeholder pixels cost LAMBDA = 1 per peak label, PHIU = 0.5 unknown
    assert np.isfinite(d[mask == 1]).any() and np.isinf(d[mask == 1]).any()


def test_data_cost_override_is_what_the_engine_uses():
    peaks, mask = mrf_cases.peaks_case(12, 9, K=3, seed=5)
    import ctypes as C
    m = O.mrf_params()
    h, w, K, _ = peaks.shape
    D = np.array([[[O.lib().sro_mrf_data_cost(C.byref(m), K, O.dptr(np.ascontiguousarray(peaks[y, x])), l)
                    for l in range(K + 1)] for x in range(w)] for y in range(h)])
    a = O.mvs_mrf(peaks, mask, m, want_messages=True)
    b = O.mvs_mrf(peaks, mask, m, data_costs=D, want_messages=True)
    assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["messages"].view(np.uint64), b["messages"].view(np.uint64))
    c = O.mvs_mrf(peaks, mask, m, data_costs=D * 3.0)
    assert not np.array_equal(a["labels"], c["labels"])


@pytest.mark.parametrize("w,h,K,seed", [(7, 5, 3, 1), (6, 9, 4, 2), (1, 8, 2, 3), (9, 1, 5, 4), (5, 5, 9, 5)])
def test_oracle_against_the_second_reading(w, h, K, seed):
    """tests/second_reading.py::Trws is a second implementation of the same published algorithm, typed from the
    paper with numpy label vectors (it shares no text with oracle/sr_oracle.c or the kernels): sweep counts, labels and
    depths identical, every stored message, the energies and the lower bound within 1e-12."""
    import second_reading as S2
    peaks, mask = mrf_cases.peaks_case(w, h, K=K, seed=40 + seed, fill=0.6, mask_frac=0.8)
    for over in (dict(), dict(min_energy_drop=-1.0, max_iters=2)):
        m = O.mrf_params(**over)
        ref = O.mvs_mrf(peaks, mask, m, depth=np.full((h, w), -5.0), want_messages=True)
        s2 = S2.Trws(peaks, S2.MrfParams(max_iters=m.max_iters, min_drop=m.min_energy_drop))
        iters, e0, e1, bound = s2.run()
        assert iters == ref["iterations"]
        assert np.array_equal(s2.labels, ref["labels"])
        assert np.allclose(s2.right, ref["messages"][:, :, 0], rtol=0, atol=1e-12)
        assert np.allclose(s2.down, ref["messages"][:, :, 1], rtol=0, atol=1e-12)
        for a, b in ((e0, ref["energy_initial"]), (e1, ref["energy_final"]), (bound, ref["lower_bound"])):
            assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
        assert np.array_equal(s2.depths(mask, np.full((h, w), -5.0)).view(np.uint64), ref["depth"].view(np.uint64))
