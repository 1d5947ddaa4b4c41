"""MultiViewStereo with the MRF branch on N synthetic views (sphere rig, geodesic r=2, 64 depth levels): initial estimate
+ MRF one view after the other (srh_mvs_initial_estimate_mrf) against initial estimates with the peaks kept, then the MRF
stage of all views with their sweeps in flight together (srh_mvs_mrf_estimate_views).  Same results either way."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # what the library asks for at load time; set here before torch initialises HIP
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import cases
import oracle_ffi as O
from stereoreconstruction_amd import capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, D = (int(sys.argv[2]), int(sys.argv[3]), 64) if len(sys.argv) > 3 else (640, 480, 64)
case = cases.get_mvs("mvs_geodesic", w=w, h=h, D=D, nviews=N)
imgs, ocams, op = cases.oracle_inputs(case)
neigh = O.mvs_neighbours(ocams, op)
cams, p = cases.hip_inputs(case)
ctx = capi.Context(0)
cases.upload_case(ctx, case, cams)
for rep in range(2):
    t = time.time()
    a = [ctx.mvs_initial_estimate_mrf(v, neigh[v], p) for v in range(N)]
    t_one = time.time() - t
    t = time.time()
    for v in range(N):
        ctx.mvs_initial_estimate_peaks(v, neigh[v], p)
    t_est = time.time() - t
    b = ctx.mvs_mrf_estimate_views(list(range(N)))
    t_all = time.time() - t
assert a == b
print("%d views %dx%d: estimate + MRF one view after the other %.1f ms; estimates %.1f ms + MRF of all views together %.1f ms = %.1f ms; sweeps per view %s"
      % (N, w, h, t_one * 1e3, t_est * 1e3, (t_all - t_est) * 1e3, t_all * 1e3, [i["iterations"] for i in b]))
