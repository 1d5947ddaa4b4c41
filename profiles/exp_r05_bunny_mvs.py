import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from stereoreconstruction_amd import capi
g = np.load('/root/repo/tests/golden/bunny_views.npz')
NV = 8
cams = [capi.camera_from_p(g["P"][v], g["dist"][v]) for v in range(NV)]
p = capi.params_mvs(min_depth=30.0, max_depth=80.0, num_depth_levels=100, image_scale=0.25, cross_check_threshold=1.01)
neigh = capi.mvs_neighbours(cams, p)
with capi.Context(0) as ctx:
    for v in range(NV):
        ctx.upload_view(v, g["rgba"][v], g["mask"][v], cams[v])
    ctx.set_option("mvs_async", 0)
    for arith in (3, 0):
        ctx.set_option("arith", arith)
        for staged in (1, 0):
            ctx.set_option("mvs_staged", staged)
            ctx.mvs_initial_estimate(0, neigh[0], p); ctx.synchronize()
            ctx.profile_reset(); ctx.profile_enable(True)
            t0 = time.perf_counter()
            for v in range(NV):
                ctx.mvs_initial_estimate(v, neigh[v], p)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            ctx.profile_enable(False)
            st = ctx.stats()
            print("arith", arith, "staged", staged, "ms per 8 views %.2f" % (dt*1e3), "last view: n_flagged", st["n_flagged"], "n_pixels", st["n_pixels"], "n_eval", st["n_eval"], "dev", st["n_eval_device"],
                  "waves", st["mvs_waves_staged"], st["mvs_waves_listed"], {k: round(v[0]/v[1], 3) for k, v in ctx.profile().items() if 'mvs' in k})
