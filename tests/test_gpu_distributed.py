"""The N>1 MultiViewStereo flow with the real HIP engine: views sharded over ranks, one all-gather of
the depth maps, the ordered cross-check chain on every rank.  One GPU box has one GPU,
so the two ranks share it and exchange through gloo with host tensors; the RCCL transport itself is
covered by bench.py at N>1 and by the single-rank communicator test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _sequential_oracle(case):
    imgs, ocams, op = cases.oracle_inputs(case)
    neigh = O.mvs_neighbours(ocams, op)
    maps = [O.mvs_initial_estimate(imgs, ocams, v, neigh[v], op)[0] for v in range(len(ocams))]
    for v in range(len(ocams)):
        O.mvs_cross_check(imgs, ocams, v, op, maps)
    return maps


def test_single_rank_engine_is_the_reference_order(hip_ctx):
    from stereoreconstruction_amd import capi
    from stereoreconstruction_amd.distributed import HipMultiViewEngine, multiview_sharded
    case = cases.get_mvs("mvs_geodesic", nviews=3, w=40, h=28, D=12)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    eng = HipMultiViewEngine(hip_ctx, [0, 1, 2], capi.mvs_neighbours(cams, p), p, "cuda:0")
    assert multiview_sharded(eng, 3) == [0, 1, 2]
    for v, want in enumerate(_sequential_oracle(case)):
        ok, msg, _ = cases.compare_depth(hip_ctx.download_depth(v), want, 1e-9)
        assert ok, "view %d: %s" % (v, msg)
    # device-to-device hand-over both ways
    t = eng.depth_tensor(1)
    eng.fence()
    eng.set_depth(2, t)
    eng.fence()
    assert np.array_equal(hip_ctx.download_depth(2).view(np.uint64), hip_ctx.download_depth(1).view(np.uint64))


def _worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cases as cs
    from stereoreconstruction_amd import capi
    from stereoreconstruction_amd.distributed import HipMultiViewEngine, multiview_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = cs.get_mvs("mvs_geodesic", nviews=3, w=40, h=28, D=12)
        cams, p = cs.hip_inputs(case)
        with capi.Context(0) as ctx:
            cs.upload_case(ctx, case, cams)
            eng = HipMultiViewEngine(ctx, [0, 1, 2], capi.mvs_neighbours(cams, p), p, "cpu")
            mine = multiview_sharded(eng, 3)
            q.put((rank, mine, [ctx.download_depth(v) for v in range(3)]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_views():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, mine, maps = q.get(timeout=300)
        got[rank] = (mine, maps)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == [0, 1] and got[1][0] == [2]
    want = _sequential_oracle(cases.get_mvs("mvs_geodesic", nviews=3, w=40, h=28, D=12))
    for v in range(3):
        assert np.array_equal(got[0][1][v].view(np.uint64), got[1][1][v].view(np.uint64)), "ranks disagree on view %d" % v
        ok, msg, _ = cases.compare_depth(got[0][1][v], want[v], 1e-9)
        assert ok, "view %d: %s" % (v, msg)


def _band_worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cases as cs
    from stereoreconstruction_amd import capi
    from stereoreconstruction_amd.distributed import HipTwoViewBandEngine, twoview_rowbands_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = cs.get_twoview("geodesic_masks", w=96, h=45, D=20)
        cams, p = cs.hip_inputs(case)
        with capi.Context(0) as ctx:
            cs.upload_case(ctx, case, cams)
            eng = HipTwoViewBandEngine(ctx, p, "cpu")
            band = twoview_rowbands_sharded(eng, 45)
            q.put((rank, band, [ctx.download_depth(v) for v in range(2)]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_split_one_pair_by_row_bands(hip_ctx):
    """BASELINE.md's C3 row "+ row-band split for 2/4/8": two ranks (sharing this box's one GPU, exchanging through gloo)
    compute 23 + 22 rows of both maps of ONE pair; rank 0 stitches and cross-checks: bit-identical to srh_twoview_compute."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, band, maps = q.get(timeout=300)
        got[rank] = (band, maps)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == (0, 23) and got[1][0] == (23, 45)
    case = cases.get_twoview("geodesic_masks", w=96, h=45, D=20)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    want_l, want_r = hip_ctx.twoview_compute(0, 1, p)
    assert np.array_equal(got[0][1][0].view(np.uint64), want_l.view(np.uint64))
    assert np.array_equal(got[0][1][1].view(np.uint64), want_r.view(np.uint64))


def test_one_rank_native_view_exchange(hip_ctx):
    """srh_comm_allgather_views (the device-resident exchange of the C++ RcclTransport) on a one-rank communicator -- all a
    one-GPU box allows: every view is this rank's own, the maps come back unchanged, bit for bit."""
    from stereoreconstruction_amd import capi
    case = cases.get_mvs("mvs_geodesic", nviews=3, w=40, h=28, D=12)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    neigh = capi.mvs_neighbours(cams, p)
    for v in range(3):
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
    before = [hip_ctx.download_depth(v) for v in range(3)]
    try:
        hip_ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    except capi.StereoHipError as e:
        if e.code == capi.SRH_E_UNSUPPORTED:
            pytest.skip("librccl not available")
        raise
    try:
        hip_ctx.comm_allgather_views([0, 1, 2])
        hip_ctx.synchronize()
        for v in range(3):
            assert np.array_equal(hip_ctx.download_depth(v).view(np.uint64), before[v].view(np.uint64))
    finally:
        hip_ctx.comm_destroy()


def test_host_allgather_waits_for_the_estimates_in_flight(hip_ctx):
    """srh_comm_allgather_host stages [send | recv] in the band scratch -- the buffer a queued MultiViewStereo estimate
    (slot 0 of the two in flight) reads its support windows from on its own stream.  sharded::runMultiView calls it right
    after queueing its views (the status word), with no other entry point in between: the maps must equal those of
    one-view-at-a-time runs bit for bit, i.e. the exchange has to finish the estimates first."""
    from stereoreconstruction_amd import capi
    case = cases.get_mvs("mvs_geodesic", nviews=3, w=320, h=240, D=48)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    neigh = capi.mvs_neighbours(cams, p)
    hip_ctx.set_option("mvs_async", 0)
    try:
        for v in range(3):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)
        want = [hip_ctx.download_depth(v) for v in range(3)]
    finally:
        hip_ctx.set_option("mvs_async", 1)
    try:
        hip_ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    except capi.StereoHipError as e:
        if e.code == capi.SRH_E_UNSUPPORTED:
            pytest.skip("librccl not available")
        raise
    try:
        payload = np.arange(1 << 21, dtype=np.float64)                 # 16 MB over the start of the band scratch, twice
        for v in range(3):
            hip_ctx.mvs_initial_estimate(v, neigh[v], p)               # only queued
        back = hip_ctx.comm_allgather_host(payload)
        assert np.array_equal(back, payload)
        for v in range(3):
            assert np.array_equal(hip_ctx.download_depth(v).view(np.uint64), want[v].view(np.uint64)), v
    finally:
        hip_ctx.comm_destroy()
