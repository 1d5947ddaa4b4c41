"""The process's pool of band buffers (srh_api.hip): a context's multi-megabyte band buffers go to a process-wide pool at
srh_destroy and serve the next context's requests -- a drop-in makes one object per computation (twoviewstereo.cpp:150-227), and
handing gigabytes back to the driver between two of them costs, now and then, a second (profiles/r06_first_call.txt).  The same
bits with and without the pool, across contexts and after the pool has been emptied; the out-of-memory retry still works when
the pool holds memory."""
import numpy as np
import pytest

import cases
from stereoreconstruction_amd import capi

pytestmark = pytest.mark.gpu


def _run_once(case, pool_mb=None, opts=()):
    cams, p = cases.hip_inputs(case)
    with capi.Context(0) as ctx:
        if pool_mb is not None:
            ctx.set_option("band_pool_mb", pool_mb)
        for k, v in opts:
            ctx.set_option(k, v)
        cases.upload_case(ctx, case, cams)
        dl, dr = ctx.twoview_compute(0, 1, p)
        st = ctx.stats()
    return dl, dr, st


def test_results_do_not_depend_on_where_the_band_buffers_came_from():
    big = cases.get_twoview("geodesic_rect", w=640, h=360, D=96)          # dense path, band buffers well above the pool's 4 MB floor
    lists = cases.get_twoview("geodesic_verged_dist_masks", w=320, h=200, D=48, radius=2)
    ref = {}
    for name, case in (("dense", big), ("lists", lists)):
        ref[name] = _run_once(case, pool_mb=0)                             # no pool: fresh hipMalloc, everything freed at destroy
    for rep in range(3):                                                   # pooled: the second and third contexts reuse the first's blocks,
        for name, case in (("dense", big), ("lists", lists)):              # alternating shapes (blocks larger than asked for)
            dl, dr, st = _run_once(case, pool_mb=65536 if rep == 0 and name == "dense" else None)
            assert np.array_equal(dl.view(np.uint64), ref[name][0].view(np.uint64)), (rep, name)
            assert np.array_equal(dr.view(np.uint64), ref[name][1].view(np.uint64)), (rep, name)
            assert st["n_eval"] == ref[name][2]["n_eval"]
    # emptying the pool is harmless, and so is an out-of-memory retry while the pool holds memory
    dl, dr, _ = _run_once(big, pool_mb=0)
    assert np.array_equal(dl.view(np.uint64), ref["dense"][0].view(np.uint64))
    _run_once(big, pool_mb=65536)
    dl, dr, st = _run_once(big, opts=(("debug_alloc_limit_mb", 24),))
    assert st["band_retries"] >= 1 and np.array_equal(dl.view(np.uint64), ref["dense"][0].view(np.uint64))
    with capi.Context(0) as ctx:
        ctx.set_option("band_pool_mb", 65536)                              # (leave the default behind for the tests that follow)
