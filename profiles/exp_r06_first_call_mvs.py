"""First runTask of MultiViewStereo on a fresh context (C4 rig): wall, kernels, the list path's decisions."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from stereoreconstruction_amd import capi, synthetic
W, H, D, V = 1280, 960, 128, 8
cams3 = synthetic.semicircle_rig(V, W, H, radius=10.0, step_deg=22.5, focal=float(W))
rgba, masks, _ = synthetic.render_sphere_views(cams3, W, H, 0x5EED0004, sphere_radius=2.0, tex_size=1024)
cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
p = capi.params_mvs(min_depth=8.0, max_depth=12.0, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC, cross_check_threshold=2*4.0/(D - 1))
neigh = capi.mvs_neighbours(cams, p)
for rep in range(2):
    with capi.Context(0) as ctx:
        for v in range(V): ctx.upload_view(v, rgba[v], masks[v], cams[v])
        ctx.synchronize()
        if len(sys.argv) > 1: ctx.set_option("debug_trace", 1)
        for call in range(3):
            ctx.profile_reset(); ctx.profile_enable(True)
            t0 = time.perf_counter(); marks = []
            for v in range(V):
                ctx.mvs_initial_estimate(v, neigh[v], p); marks.append(round((time.perf_counter() - t0)*1e3, 2))
            ctx.synchronize(); t1 = (time.perf_counter() - t0)*1e3
            for v in range(V): ctx.mvs_cross_check(list(range(V)), v, p)
            ctx.synchronize()
            ms = (time.perf_counter() - t0)*1e3
            ctx.profile_enable(False)
            prof = ctx.profile()
            print("context %d call %d: %.2f ms wall (estimates %.2f, host returned at %s), kernels %s" % (
                rep, call, ms, t1, marks, {k: round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}))
