"""Template scan against the per-pixel walk, C3 left -> right: tiles settled by either, kernel times."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from stereoreconstruction_amd import capi, synthetic
W, H, D = 1920, 1080, 256
L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0003)
(Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
zmin, zmax = synthetic.rectified_depth_range(W, D)
p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
with capi.Context(0) as ctx:
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    maps = {}
    for ts in (1, 0):
        ctx.set_option("tscan", ts)
        for a, b in ((0, 1), (1, 0)):
            ctx.twoview_wta(a, b, p); ctx.synchronize()
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(3):
                ctx.twoview_wta(a, b, p)
            ctx.synchronize(); ctx.profile_enable(False)
            st = ctx.stats()
            maps[(ts, a)] = ctx.download_depth(a)
            print("tscan", ts, "dir", a, "tiles template", st["scan_tiles_template"], "walked", st["scan_tiles_walked"], "n_eval", st["n_eval"], "flagged", st["n_flagged"],
                  {k: round(v[0]/v[1], 3) for k, v in ctx.profile().items() if "scan" in k or "template" in k})
    for a in (0, 1):
        print("same bits dir", a, np.array_equal(maps[(1, a)].view(np.uint64), maps[(0, a)].view(np.uint64)))
