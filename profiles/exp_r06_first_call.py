"""What a pair costs the FIRST time (VERDICT r5 missing #2): fresh context, uploads fenced, one srh_twoview_compute -- wall clock,
the kernels the call launched (HIP events of the library's own scopes) and what is left: allocations, host waits, plane builds.
Then the same call again on the same context (steady state) for comparison.  usage: python profiles/exp_r06_first_call.py [c3|c5|c2]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from stereoreconstruction_amd import capi, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
W, H, D, wk, seed = {"c3": (1920, 1080, 256, capi.WEIGHT_GEODESIC, 0x5EED0003), "c5": (1920, 1080, 256, capi.WEIGHT_GEODESIC, 0x5EED0050),
                     "c2": (640, 480, 64, capi.WEIGHT_ADAPTIVE, 0x5EED0002)}[which]
L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, seed)
(Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
zmin, zmax = synthetic.rectified_depth_range(W, D)
plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333) if which == "c5" else ()
p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wk)
# a throw-away context first: the process's code objects are resident afterwards
with capi.Context(0) as warm:
    warm.upload_view(0, L[:64, :64].copy(), ml[:64, :64].copy(), capi.camera_from_krt(Kl, Rl, tl, None, *plane))
    warm.upload_view(1, R[:64, :64].copy(), mr[:64, :64].copy(), capi.camera_from_krt(Kr, Rr, tr, None, *plane))
    warm.synchronize()
for rep in range(2):
    with capi.Context(0) as ctx:
        ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl, None, *plane))
        ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr, None, *plane))
        ctx.synchronize()
        if len(sys.argv) > 2: ctx.set_option("debug_trace", 1)
        for call in range(4):
            ctx.profile_reset(); ctx.profile_enable(True)
            t0 = time.perf_counter()
            ctx.twoview_compute_device(0, 1, p)
            ctx.synchronize()
            ms = (time.perf_counter() - t0)*1e3
            ctx.profile_enable(False)
            prof = ctx.profile()
            ksum = sum(v[0] for v in prof.values())
            print("%s context %d call %d: %.2f ms wall, kernels (sum of scopes, passes may overlap) %.2f ms: %s" % (
                which, rep, call, ms, ksum, {k: round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:9]}))
