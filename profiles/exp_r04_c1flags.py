"""round 4 experiment: how many pixels the certified scan flags on C1 (the bunny pair), per certified form and direction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stereoreconstruction_amd import capi
g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_pair.npz"))
cl = capi.camera_from_krt(g["left_K"], g["left_R"], g["left_t"], g["left_dist"])
cr = capi.camera_from_krt(g["right_K"], g["right_R"], g["right_t"], g["right_dist"])
p = capi.params_twoview(min_depth=30.0, max_depth=80.0, num_depth_levels=100, weight_kind=capi.WEIGHT_GEODESIC, image_scale=float(g["scale"][0]))
with capi.Context(0) as ctx:
    ctx.upload_view(0, g["left_rgba"], g["left_mask"], cl)
    ctx.upload_view(1, g["right_rgba"], g["right_mask"], cr)
    for form in (1, 2):
        ctx.set_option("cert_form", form)
        for a, b in ((0, 1), (1, 0)):
            ctx.twoview_wta(a, b, p)
            st = ctx.stats()
            print("form", form, "ref", a, "certified", st["n_certified"], "flagged", st["n_flagged"], "pixels", st["n_pixels"])
