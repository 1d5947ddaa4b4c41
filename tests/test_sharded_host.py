"""host/sharded.hpp -- the C++ side of the multi-GPU run: MultiViewStereo::runTask over several
srh_contexts, views sharded, one all-gather of the depth maps, the ordered cross-check chain on every shard.
CPU: the sharding / padding / ordering logic on a stand-in engine, 2 and 3 ranks as threads.  GPU: the same scene
on one context and on 2 and 3 contexts of one process (one per GPU where there are several), and through the RCCL
transport on a one-rank communicator -- all bit-identical, and equal to the C-ABI driven from Python."""
import os
import subprocess

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereoreconstruction_amd", "host")
LIBDIR = os.path.join(ROOT, "stereoreconstruction_amd")


def _build(tmp):
    exe = os.path.join(tmp, "sharded_host_test")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
                           os.path.join(ROOT, "tests", "sharded_host_test.cpp"),
                           "-L" + LIBDIR, "-lstereo_recon_hip", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_sharding_logic_on_cpu(tmp_path):
    out = subprocess.run([_build(str(tmp_path)), "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpu ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_sharded_contexts_equal_one_context(tmp_path, hip_ctx):
    from test_gpu_host_api import _write_input
    from stereoreconstruction_amd import capi
    case = cases.get_mvs("mvs_five_views")
    views = []
    for (rgba, mask, cam, dist, plane) in case["views"]:
        im = rgba.copy()
        im[..., 3] = np.where(mask == 1, 255, 51)                 # the driver takes the mask from alpha
        views.append((im, mask, cam, dist, plane))
    case = dict(case, views=views)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_input(inp, case, False)
    r = subprocess.run([_build(str(tmp_path)), "gpu", inp, outp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "gpu ok" in r.stdout, r.stdout + r.stderr
    h, w = views[0][0].shape[:2]
    nv = len(views)
    got = np.fromfile(outp).reshape(nv, h, w)
    cams, p = cases.hip_inputs(case)
    cases.upload_case(hip_ctx, case, cams)
    neigh = capi.mvs_neighbours(cams, p)
    for v in range(nv):
        hip_ctx.mvs_initial_estimate(v, neigh[v], p)
    for v in range(nv):
        hip_ctx.mvs_cross_check(list(range(nv)), v, p)
    for v in range(nv):
        assert np.array_equal(got[v].view(np.uint64), hip_ctx.download_depth(v).view(np.uint64)), v
    assert np.isnan(got).any() and np.isfinite(got).any()
