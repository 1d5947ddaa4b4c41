"""Where the C5 list kernel's time goes (experiment build): whole, without raster / visitor, without Newton, without refraction."""
import os, sys
os.environ["SRH_LIBRARY"] = os.path.abspath("profiles/lib/libstereo_recon_hip_exp.so")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from stereoreconstruction_amd import capi, synthetic
W, H, D = 1920, 1080, 256
L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0050)
(Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
zmin, zmax = synthetic.rectified_depth_range(W, D)
plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333)
p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
with capi.Context(0) as ctx:
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl, None, *plane))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr, None, *plane))
    ctx.twoview_wta(0, 1, p); ctx.synchronize()
    ctx.set_option("exp_rows_mode", 9)                             # Newton statistics of one pass (printed on stderr by mode -1)
    ctx.twoview_wta(0, 1, p); ctx.synchronize()
    ctx.set_option("exp_rows_mode", -1)
    for mode in (0, 1, 2, 3):
        ctx.set_option("exp_rows_mode", mode)
        ctx.profile_reset(); ctx.profile_enable(True)
        try:
            ctx.twoview_wta(0, 1, p)
        except Exception as e:
            print("mode", mode, "run error (expected for modes that change the lists):", str(e)[:80])
        ctx.synchronize(); ctx.profile_enable(False)
        print("exp rows mode", mode, {k: round(v[0]/v[1], 3) for k, v in ctx.profile().items() if "list" in k})
