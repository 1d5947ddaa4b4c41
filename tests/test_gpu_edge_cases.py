"""Edge cases of the C-ABI on the device: argument errors, degenerate inputs, cancellation."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

pytestmark = pytest.mark.gpu


def test_argument_validation(hip_ctx):
    case = cases.get_twoview("adaptive_rect", w=32, h=20, D=8)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    with pytest.raises(capi.StereoHipError) as e:
        hip_ctx.twoview_wta(0, 0, p)
    assert e.value.code == capi.SRH_E_INVALID
    with pytest.raises(capi.StereoHipError):
        hip_ctx.twoview_wta(0, 63, p)                       # empty slot
    with pytest.raises(capi.StereoHipError):
        hip_ctx.twoview_wta(0, 64, p)                       # slot out of range
    bad = capi.params_twoview(num_depth_levels=1)
    with pytest.raises(capi.StereoHipError):
        hip_ctx.twoview_wta(0, 1, bad)
    bad = capi.params_twoview(window_radius=0)
    with pytest.raises(capi.StereoHipError):
        hip_ctx.twoview_wta(0, 1, bad)
    # unequal sizes: the reference sizes the right map from the left image (twoviewstereo.cpp:119)
    rgba, mask, cam, dist, plane = case["views"][1]
    hip_ctx.upload_view(2, rgba[:, :-1].copy(), mask[:, :-1].copy(), cams[1])
    with pytest.raises(capi.StereoHipError) as e:
        hip_ctx.twoview_wta(0, 2, p)
    assert "equal-sized" in str(e.value)
    with pytest.raises(capi.StereoHipError):
        hip_ctx.mvs_initial_estimate(0, [0], capi.params_mvs())      # own neighbour
    with pytest.raises(capi.StereoHipError) as e:
        hip_ctx.mvs_initial_estimate(0, [1, 2] * 5, capi.params_mvs())   # more than SRH_MAX_NEIGH = 8
    assert e.value.code == capi.SRH_E_UNSUPPORTED


def test_other_view_fully_masked_and_out_of_range_depths(hip_ctx):
    case = cases.get_twoview("geodesic_rect", w=40, h=24, D=8)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    # no candidate survives an all-zero mask of the other view: empty curve => NaN everywhere
    rgba, mask, cam, dist, plane = case["views"][1]
    hip_ctx.upload_view(1, rgba, np.zeros_like(mask), cams[1])
    hip_ctx.twoview_wta(0, 1, p)
    assert np.isnan(hip_ctx.download_depth(0)).all()
    assert hip_ctx.stats()["n_eval"] == 0
    hip_ctx.upload_view(1, rgba, mask, cams[1])
    # a depth range that projects outside the other image: same answer as the oracle (all NaN)
    far = dict(case["params"], min_depth=1e-3, max_depth=2e-3)
    p2 = capi.params_twoview(**far)
    op2 = O.params_twoview(**far)
    hip_ctx.twoview_wta(0, 1, p2)
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op2)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), want, 1e-9)
    assert ok, msg


@pytest.mark.parametrize("radius", [1, 3, 4])
def test_other_window_radii_use_the_general_kernels(hip_ctx, radius):
    case = cases.get_twoview("geodesic_rect", w=44, h=26, D=10, radius=radius)
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.twoview_wta(0, 1, p)
    assert not hip_ctx.stats()["used_dense_path"]
    want = O.twoview_wta(imgs[0], imgs[1], ocams[0], ocams[1], op)
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), want, 1e-9)
    assert ok, msg


def test_cancel_flag_is_honoured(hip_ctx):
    case = cases.get_twoview("adaptive_rect", w=32, h=20, D=8)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    flag = C.c_int(1)
    hip_ctx.set_hooks(flag, None)
    try:
        with pytest.raises(capi.StereoHipError) as e:
            hip_ctx.twoview_wta(0, 1, p)
        assert e.value.code == capi.SRH_E_CANCELLED
    finally:
        hip_ctx.set_hooks(None, None)
    hip_ctx.twoview_wta(0, 1, p)                             # and works again afterwards


def test_mvs_topk_peaks_match_oracle(hip_ctx):
    """The sorted top-K (cost, depth) list the MRF branch would consume (multiviewstereo.cpp:600-602)."""
    import torch
    case = cases.get_mvs("mvs_geodesic", w=40, h=28, D=16, nviews=3)
    imgs, ocams, op = cases.oracle_inputs(case)
    neigh = O.mvs_neighbours(ocams, op)
    want_d, want_pk, _ = O.mvs_initial_estimate(imgs, ocams, 0, neigh[0], op, want_peaks=True)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    pk = torch.zeros((28, 40, p.top_k, 2), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.mvs_initial_estimate(0, neigh[0], p, peaks_dev=pk.data_ptr())
    hip_ctx.synchronize()
    got = pk.cpu().numpy()
    assert np.allclose(got[..., 0], want_pk[..., 0], rtol=0, atol=1e-12)
    assert np.allclose(got[..., 1], want_pk[..., 1], rtol=1e-9, atol=0)
    assert (want_pk[..., 0] > 0.95).any()
    # several real peaks per pixel somewhere, i.e. the lists are not just the best pair
    assert ((want_pk[..., 0] > 0.95).sum(axis=-1) >= 3).any()
    ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(0), want_d, 1e-9)
    assert ok, msg
    # the one-thread-per-pixel kernel (lists kept inline) gives the same bits as walk -> list cost -> merge
    hip_ctx.set_option("force_generic", 1)
    pk2 = torch.full((28, 40, p.top_k, 2), 7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    hip_ctx.mvs_initial_estimate(0, neigh[0], p, peaks_dev=pk2.data_ptr())
    hip_ctx.synchronize()
    hip_ctx.set_option("force_generic", 0)
    assert np.array_equal(got.view(np.uint64), pk2.cpu().numpy().view(np.uint64))


def test_rccl_exchange_single_rank(hip_ctx):
    """The C-ABI's RCCL gather / all-gather on a one-rank communicator (all a 1-GPU box allows):
    exercises the lazy librccl load, communicator set-up and the device-to-device paths."""
    import torch
    case = cases.get_twoview("adaptive_rect", w=32, h=20, D=8)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    hip_ctx.twoview_wta(0, 1, p)
    want = hip_ctx.download_depth(0)
    uid = capi.Context.comm_unique_id()
    assert len(uid) == 128
    hip_ctx.comm_init(1, 0, uid)
    try:
        a = torch.zeros((1, 20, 32), dtype=torch.float64, device="cuda:0")
        b = torch.zeros((1, 20, 32), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        hip_ctx.comm_gather_depth(0, 0, a.data_ptr())
        hip_ctx.comm_allgather_depth(0, b.data_ptr())
        hip_ctx.synchronize()
        for t in (a, b):
            got = t.cpu().numpy()[0]
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        with pytest.raises(capi.StereoHipError):
            hip_ctx.comm_gather_depth(0, 3, a.data_ptr())          # root outside the communicator
    finally:
        hip_ctx.comm_destroy()
    with pytest.raises(capi.StereoHipError):
        hip_ctx.comm_allgather_depth(0, 1)                          # no communicator any more


def _twoview_vs_oracle(ctx, case, expect_dense):
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(ctx, case, cams)
    for ref, oth in ((0, 1), (1, 0)):
        want = O.twoview_wta(imgs[ref], imgs[oth], ocams[ref], ocams[oth], op)
        ctx.twoview_wta(ref, oth, p)
        assert bool(ctx.stats()["used_dense_path"]) == expect_dense
        ok, msg, _ = cases.compare_depth(ctx.download_depth(ref), want, 1e-9)
        assert ok, (ref, msg)


def test_dense_plan_refuted_on_device_is_redone_on_the_general_kernels(hip_ctx):
    """srh_twoview_wta proposes the dense row-aligned plan from a host-side rig check and lets the scan kernel
    verify every candidate; a candidate off its row makes the host redo the pass on the general kernels
    (srh_api.hip, `not_row_aligned`).  The host check is strict enough that no real rig reaches the redo, so the
    plan is forced ("force_dense") on a verged pinhole pair: the result must still equal the oracle."""
    case = cases.get_twoview("adaptive_verged", w=72, h=44, D=20, radius=5)
    hip_ctx.set_option("force_dense", 1)
    try:
        _twoview_vs_oracle(hip_ctx, case, expect_dense=False)
    finally:
        hip_ctx.set_option("force_dense", 0)
    # and the same rig with the option off never tries the dense plan
    _twoview_vs_oracle(hip_ctx, case, expect_dense=False)


@pytest.mark.parametrize("zmin,dense", [(64.0 / 300.0, True), (64.0 / 5000.0, False)],
                         ids=["span_wider_than_the_image", "span_4096_or_more"])
def test_dense_plan_span_limits(hip_ctx, zmin, dense):
    """Candidate ranges wider than the image clamp the cost-row stride to W+8 (still the dense kernels);
    a nominal span of 4096 columns or more is left to the general kernels (srh_api.hip plan)."""
    case = cases.get_twoview("geodesic_rect", w=64, h=36, D=24)
    case["params"]["min_depth"] = zmin
    _twoview_vs_oracle(hip_ctx, case, expect_dense=dense)


def test_mvs_more_than_three_neighbours(hip_ctx):
    """num_neighbours is a run-time parameter (the reference's NUM_NEIGHBOURING_VIEWS = 3 is a constant,
    multiviewstereo.cpp:97): four neighbours per view on both MVS paths, against the oracle."""
    case = cases.get_mvs("mvs_five_views")
    case["params"]["num_neighbours"] = 4
    imgs, ocams, op = cases.oracle_inputs(case)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    neigh = O.mvs_neighbours(ocams, op)
    assert [list(map(int, n)) for n in capi.mvs_neighbours(cams, p)] == [list(map(int, n)) for n in neigh]
    assert max(len(n) for n in neigh) == 4
    for v in (0, 2):
        want, n_eval = O.mvs_initial_estimate(imgs, ocams, v, neigh[v], op)
        for generic in (0, 1):
            hip_ctx.set_option("force_generic", generic)
            try:
                hip_ctx.mvs_initial_estimate(v, neigh[v], p)
            finally:
                hip_ctx.set_option("force_generic", 0)
            ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(v), want, 1e-9)
            assert ok, (v, generic, msg)
            assert hip_ctx.stats()["n_eval"] == n_eval


def test_mvs_estimates_in_flight_interleaved_with_other_calls(hip_ctx):
    """srh_mvs_initial_estimate only queues a view (two views in flight on two streams): whatever a caller does next --
    more estimates, a re-upload of a neighbour, downloads, the ordered cross-check -- sees complete maps.  Against a
    second context that waits for every view (option mvs_async 0), bit for bit, and a list capacity that is too small
    on first use (a fresh context's hint) is found out later and the view redone."""
    case = cases.get_mvs("mvs_five_views")
    cams, p = cases.hip_inputs(case)
    neigh = [list(map(int, n)) for n in capi.mvs_neighbours(cams, p)]
    nv = len(cams)
    ref = capi.Context(0)                                       # fresh: its list-capacity hint starts at the default
    try:
        ref.set_option("mvs_async", 0)
        cases.upload_case(ref, case, cams)
        want = []
        for v in range(nv):
            ref.mvs_initial_estimate(v, neigh[v], p)
            want.append(ref.download_depth(v))
        evals_last = ref.stats()["n_eval"]
        for v in range(nv):
            ref.mvs_cross_check(list(range(nv)), v, p)
        want_cc = [ref.download_depth(v) for v in range(nv)]
    finally:
        ref.close()
    ctx = capi.Context(0)
    try:
        cases.upload_case(ctx, case, cams)
        for v in range(nv):                                     # five views queued back to back: slots reused twice
            ctx.mvs_initial_estimate(v, neigh[v], p)
        assert ctx.stats()["n_eval"] == evals_last
        rgba, mask, _, _, _ = case["views"][1]
        ctx.upload_view(1, rgba, mask, cams[1])                 # re-upload of a view while nothing may be in flight any more
        ctx.mvs_initial_estimate(1, neigh[1], p)
        ctx.mvs_initial_estimate(0, neigh[0], p)
        for v in range(nv):
            assert np.array_equal(ctx.download_depth(v).view(np.uint64), want[v].view(np.uint64)), v
        ctx.mvs_initial_estimate(3, neigh[3], p)                # queued, then consumed by the cross-check chain directly
        for v in range(nv):
            ctx.mvs_cross_check(list(range(nv)), v, p)
        for v in range(nv):
            assert np.array_equal(ctx.download_depth(v).view(np.uint64), want_cc[v].view(np.uint64)), v
    finally:
        ctx.close()


def _bits(a):
    return a.view(np.uint64)


def test_band_budget_follows_free_memory_and_survives_allocation_failures(hip_ctx):
    """The band budget is a request: a run plans with at most a quarter of the device memory that is free (option
    mem_limit_mb pretends there is little), and a run whose band buffers are refused (debug_alloc_limit_mb: as if the
    device were out of memory) is repeated with thinner bands instead of failing.  Same bits either way: TwoView dense,
    TwoView candidate lists (refractive), MultiViewStereo list path."""
    tv = cases.get_twoview("geodesic_masks", w=160, h=96, D=24)
    rf = cases.get_twoview("adaptive_refractive", w=128, h=80, D=20, radius=5)
    mv = cases.get_mvs("mvs_geodesic", nviews=3, w=128, h=96, D=24)

    def run_all():
        out = []
        for case in (tv, rf):
            cams, p = cases.hip_inputs(case)
            cases.upload_case(hip_ctx, case, cams)
            hip_ctx.twoview_wta(0, 1, p)
            out.append(hip_ctx.download_depth(0))
        cams, p = cases.hip_inputs(mv)
        cases.upload_case(hip_ctx, mv, cams)
        neigh = capi.mvs_neighbours(cams, p)
        for v in range(3):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        out += [hip_ctx.download_depth(v) for v in range(3)]
        return out

    want = run_all()
    st0 = hip_ctx.stats()
    assert 0 < st0["band_budget_bytes"] <= 32768 << 20
    try:
        hip_ctx.set_option("mem_limit_mb", 16)                    # a quarter of it: 4 MB bands
        got = run_all()
        st = hip_ctx.stats()
        assert st["band_budget_bytes"] == 4 << 20 and st["band_retries"] == st0["band_retries"]
        for a, b in zip(want, got):
            assert np.array_equal(_bits(a), _bits(b))
        hip_ctx.set_option("mem_limit_mb", 0)
        hip_ctx.set_option("band_budget_mb", 32768)
        hip_ctx.set_option("debug_alloc_limit_mb", 2)             # every band buffer above 2 MB is "out of memory"
        got = run_all()
        st = hip_ctx.stats()
        assert st["band_retries"] > st0["band_retries"] and st["band_budget_bytes"] < 64 << 20
        for a, b in zip(want, got):
            assert np.array_equal(_bits(a), _bits(b))
    finally:
        hip_ctx.set_option("debug_alloc_limit_mb", 0)
        hip_ctx.set_option("mem_limit_mb", 0)
        hip_ctx.set_option("band_budget_mb", 32768)               # (forgets the halvings)


def test_queued_view_with_a_cut_list_survives_the_oom_retry_of_the_next_call(hip_ctx):
    """A MultiViewStereo view queued with a list capacity that turns out too small is redone when its slot is settled.
    When the NEXT call runs out of memory in between, with_thinner_bands drains everything and releases the band buffers:
    the queued view must still get its capacity check (round-4 advisor: it was dropped, leaving a depth map made from cut
    lists).  View 0 is queued on 8 rows (small buffers: below the debug allocation limit) with a capacity hint of 8
    candidates; view 1's full-height call is refused its buffers and retried with thinner bands."""
    mv = cases.get_mvs("mvs_geodesic", nviews=3, w=128, h=96, D=24)
    cams, p = cases.hip_inputs(mv)
    neigh = capi.mvs_neighbours(cams, p)
    cases.upload_case(hip_ctx, mv, cams)
    hip_ctx.set_option("mvs_async", 0)
    try:
        hip_ctx.mvs_initial_estimate(0, neigh[0], p, 0, 8)
        want0 = hip_ctx.download_depth(0)[:8].copy()
        hip_ctx.mvs_initial_estimate(1, neigh[1], p)
        want1 = hip_ctx.download_depth(1)
    finally:
        hip_ctx.set_option("mvs_async", 1)
    cases.upload_case(hip_ctx, mv, cams)                          # (fresh depth maps)
    st0 = hip_ctx.stats()
    try:
        hip_ctx.set_option("band_budget_mb", 32768)
        hip_ctx.set_option("debug_alloc_limit_mb", 1)
        hip_ctx.set_option("debug_mvs_cmax_hint", 8)              # far below the ~50 candidates of a curve here
        hip_ctx.mvs_initial_estimate(0, neigh[0], p, 0, 8)        # queued on slot 0: lists cut at 8 entries
        hip_ctx.mvs_initial_estimate(1, neigh[1], p)              # slot 1: buffers refused -> drained, released, retried
        got1 = hip_ctx.download_depth(1)
        got0 = hip_ctx.download_depth(0)[:8]
        assert hip_ctx.stats()["band_retries"] > st0["band_retries"]
    finally:
        hip_ctx.set_option("debug_alloc_limit_mb", 0)
        hip_ctx.set_option("debug_mvs_cmax_hint", 0)
        hip_ctx.set_option("band_budget_mb", 32768)
    assert np.array_equal(_bits(got0), _bits(want0))
    assert np.array_equal(_bits(got1), _bits(want1))


def test_two_contexts_share_one_gpu(hip_ctx):
    """A second context on the same GPU (a GUI host beside a batch job, the loopback transport's shards): each plans its
    bands from what is free when it runs; both give the same bits."""
    case = cases.get_twoview("geodesic_masks", w=160, h=96, D=24)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    a_l, a_r = hip_ctx.twoview_compute(0, 1, p)
    with capi.Context(0) as other:
        cases.upload_case(other, case, cams)
        b_l, b_r = other.twoview_compute(0, 1, p)
        c_l, c_r = hip_ctx.twoview_compute(0, 1, p)               # interleaved with the other context's buffers alive
        assert other.stats()["band_budget_bytes"] > 0
    assert np.array_equal(_bits(a_l), _bits(b_l)) and np.array_equal(_bits(a_r), _bits(b_r))
    assert np.array_equal(_bits(a_l), _bits(c_l)) and np.array_equal(_bits(a_r), _bits(c_r))


def test_twoview_compute_passes_side_by_side_or_one_after_the_other(hip_ctx):
    """srh_twoview_compute queues its second pass on a stream and band buffers of its own beside the first (option
    tv_overlap, default 1): the same bits and the same counters as with the passes one after the other -- on the dense plan
    (where the overlap happens), on a plan the device refutes in both passes (force_dense on a verged rig: both maps are
    redone one after the other), on the list path (no overlap: its passes wait for their counters), and again after the
    band buffers were released under it (allocation failures with the second set in use)."""
    runs = [("geodesic_masks", dict(w=160, h=96, D=24), 0), ("adaptive_rect", dict(w=96, h=64, D=32, radius=5), 0),
            ("adaptive_verged", dict(w=72, h=44, D=20, radius=5), 1), ("adaptive_refractive", dict(w=96, h=60, D=20, radius=5), 0)]
    for name, kw, force_dense in runs:
        case = cases.get_twoview(name, **kw)
        cams, p = cases.hip_inputs(case)
        cases.upload_case(hip_ctx, case, cams)
        hip_ctx.set_option("force_dense", force_dense)
        got = {}
        try:
            for ov in (1, 0, 1):
                hip_ctx.set_option("tv_overlap", ov)
                l, r = hip_ctx.twoview_compute(0, 1, p)
                st = hip_ctx.stats()
                cur = (l, r, st["n_eval"], st["n_pixels"], st["used_dense_path"])
                if ov in got:
                    prev = got[ov]
                    assert np.array_equal(_bits(prev[0]), _bits(cur[0])) and np.array_equal(_bits(prev[1]), _bits(cur[1])) and prev[2:] == cur[2:]
                got[ov] = cur
            assert np.array_equal(_bits(got[0][0]), _bits(got[1][0])) and np.array_equal(_bits(got[0][1]), _bits(got[1][1])), name
            assert got[0][2:] == got[1][2:], (name, got[0][2:], got[1][2:])
        finally:
            hip_ctx.set_option("force_dense", 0)
            hip_ctx.set_option("tv_overlap", 1)
    # allocation failures while both sets of band buffers are wanted: thinner bands, the same bits
    case = cases.get_twoview("geodesic_masks", w=160, h=96, D=24)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want = hip_ctx.twoview_compute(0, 1, p)
    st0 = hip_ctx.stats()
    try:
        hip_ctx.set_option("debug_alloc_limit_mb", 2)
        got2 = hip_ctx.twoview_compute(0, 1, p)
        assert hip_ctx.stats()["band_retries"] > st0["band_retries"]
    finally:
        hip_ctx.set_option("debug_alloc_limit_mb", 0)
    assert np.array_equal(_bits(want[0]), _bits(got2[0])) and np.array_equal(_bits(want[1]), _bits(got2[1]))


def test_caller_owned_stream(hip_ctx):
    """srh_set_stream: the library works on a stream of the caller's (here one of torch's): srh_twoview_compute with its
    second pass on the library's side stream, and queued MultiViewStereo estimates, give the bits of the context's own
    stream; work the caller puts on its stream afterwards is ordered behind them."""
    torch = pytest.importorskip("torch")
    tv = cases.get_twoview("geodesic_masks", w=160, h=96, D=24)
    cams, p = cases.hip_inputs(tv)
    cases.upload_case(hip_ctx, tv, cams)
    want = hip_ctx.twoview_compute(0, 1, p)
    stream = torch.cuda.Stream()
    hip_ctx.set_stream(stream.cuda_stream)
    try:
        got = hip_ctx.twoview_compute(0, 1, p)
        assert np.array_equal(_bits(want[0]), _bits(got[0])) and np.array_equal(_bits(want[1]), _bits(got[1]))
        # a MultiViewStereo estimate queued on a side stream, its map copied device-to-device on the caller's stream
        mv = cases.get_mvs("mvs_geodesic", nviews=3, w=128, h=96, D=24)
        mcams, mp = cases.hip_inputs(mv)
        cases.upload_case(hip_ctx, mv, mcams)
        neigh = capi.mvs_neighbours(mcams, mp)
        out = torch.empty((96, 128), dtype=torch.float64, device="cuda")
        hip_ctx.mvs_initial_estimate(0, neigh[0], mp)                # queued on a side stream, ordered into the caller's
        ptr = hip_ctx.depth_device_ptr(0)
        hip_ctx.copy_depth_to_device(0, out.data_ptr(), 96 * 128 * 8)   # (on the caller's stream, behind the estimate)
        stream.synchronize()
        got0 = out.cpu().numpy()
    finally:
        hip_ctx.set_stream(0)
    hip_ctx.set_option("mvs_async", 0)
    try:
        hip_ctx.mvs_initial_estimate(0, neigh[0], mp)
        ref0 = hip_ctx.download_depth(0)
    finally:
        hip_ctx.set_option("mvs_async", 1)
    assert ptr and np.array_equal(_bits(ref0), _bits(got0))
