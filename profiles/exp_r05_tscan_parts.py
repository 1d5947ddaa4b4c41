"""Where the template scan's time goes (experiment build): whole, without the verification, without the look-ups."""
import os, sys
os.environ["SRH_LIBRARY"] = os.path.abspath("profiles/lib/libstereo_recon_hip_exp.so")
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
    for mode in (0, 3, 4, 5, 6):
        ctx.set_option("exp_scan_mode", mode)
        for a, b in ((0, 1), (1, 0)):
            ctx.twoview_wta(a, b, p); ctx.synchronize()
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(3):
                ctx.twoview_wta(a, b, p)
            ctx.synchronize(); ctx.profile_enable(False)
            st = ctx.stats()
            print("exp mode", mode, "dir", a, "tiles template", st["scan_tiles_template"], "walked", st["scan_tiles_walked"],
                  {k: round(v[0]/v[1], 3) for k, v in ctx.profile().items() if "scan" in k or "template" in k})
