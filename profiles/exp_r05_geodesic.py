"""Geodesic windows kernel, C3 both directions: kernel time (HIP events of the library's own scopes), step time, a hash
of the depth maps' bits.  Run once per library (SRH_LIBRARY): the variants differ in GEO_AHEAD / the sweep's form only."""
import hashlib, os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from stereoreconstruction_amd import capi, synthetic
W, H, D = 1920, 1080, 256
L, R, ml, mr, _ = synthetic.rectified_pair(W, H, D, 0x5EED0003)
(Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
zmin, zmax = synthetic.rectified_depth_range(W, D)
p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=capi.WEIGHT_GEODESIC)
with capi.Context(0) as ctx:
    if "GEODMA" in os.environ: ctx.set_option("geodma", int(os.environ["GEODMA"]))
    ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
    ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
    h = hashlib.sha256()
    for a, b in ((0, 1), (1, 0)):
        ctx.twoview_wta(a, b, p); ctx.synchronize()
        h.update(ctx.download_depth(a).tobytes())
    rows, _, _ = ctx.twoview_cost_rows(0, 1, p, 500, 524, 0)           # the reference's arithmetic on this build's windows
    hr = hashlib.sha256(rows.tobytes()).hexdigest()[:16]
    rows, _, _ = ctx.twoview_cost_rows(1, 0, p, 0, 12, 0)              # (rows at the image border: windows over the edge)
    hr += " " + hashlib.sha256(rows.tobytes()).hexdigest()[:16]
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(4):
        ctx.twoview_wta(0, 1, p); ctx.twoview_wta(1, 0, p)
    ctx.synchronize(); ctx.profile_enable(False)
    prof = {k: round(v[0]/v[1], 3) for k, v in ctx.profile().items() if "geodesic" in k or "strip" in k}
    print(os.environ.get("SRH_LIBRARY", "default").split("/")[-1], capi.build_id() if hasattr(capi, "build_id") else "", prof, "depth bits", h.hexdigest()[:16], "cost rows", hr)
