"""C4 MultiViewStereo initial estimates with and without the sorted top-K (cost, depth) output (the MRF input).
usage: python3 profiles/mvs_bench.py [small]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from stereoreconstruction_amd import capi, synthetic
W, H, D, NV = 1280, 960, 128, 8
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    W, H, D, NV = 320, 240, 64, 4
cams3 = synthetic.semicircle_rig(NV, W, H, radius=10.0, step_deg=22.5, focal=float(W))
rgba, masks, depth = synthetic.render_sphere_views(cams3, W, H, 0x5EED0004, sphere_radius=2.0, tex_size=1024)
cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
p = capi.params_mvs(min_depth=8.0, max_depth=12.0, num_depth_levels=D, cross_check_threshold=2 * 4.0 / (D - 1))
neigh = capi.mvs_neighbours(cams, p)
torch.cuda.init()
ctx = capi.Context(0)
for v in range(NV):
    ctx.upload_view(v, rgba[v], masks[v], cams[v])
pk = torch.zeros((H, W, p.top_k, 2), dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
for tag, peaks, generic in (("two-stage", 0, 0), ("two-stage + top-K", 1, 0), ("inline kernel + top-K", 1, 1), ("inline kernel", 0, 1)):
    ctx.set_option("force_generic", generic)
    def run():
        for v in range(NV):
            ctx.mvs_initial_estimate(v, neigh[v], p, peaks_dev=pk.data_ptr() if peaks else None)
        ctx.synchronize()
    run()
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    print("%-24s %7.1f ms for %d views" % (tag, dt * 1e3, NV))
