"""Times the MRF stage (srh_mvs_mrf_estimate) on a synthetic 1280x960 top-9 peaks buffer: sweeps forced to a fixed
count so that the per-sweep time can be read off; run under rocprofv3 --kernel-trace --stats for the per-kernel view."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import mrf_cases
from stereoreconstruction_amd import capi

w, h, K = (int(sys.argv[1]), int(sys.argv[2]), 9) if len(sys.argv) > 2 else (1280, 960, 9)
peaks, mask = mrf_cases.peaks_case_fast(w, h, K=K, seed=5)
ctx = capi.Context(0)
rgba = np.zeros((h, w, 4), dtype=np.uint8); rgba[..., 3] = 255
Km = np.array([[100.0, 0, w / 2], [0, 100.0, h / 2], [0, 0, 1]])
ctx.upload_view(0, rgba, mask, capi.camera_from_krt(Km, np.eye(3), np.zeros(3), None))
pk = torch.from_numpy(peaks).to("cuda:0"); torch.cuda.synchronize()
for sweeps in (1, 4, 4):
    t = time.time()
    info = ctx.mvs_mrf_estimate(0, K, pk.data_ptr(), capi.mrf_params(min_energy_drop=-1.0, max_iters=sweeps - 1))
    dt = time.time() - t
    print("%dx%d K=%d sweeps=%d: %.2f ms total, energy %.6f -> %.6f" % (w, h, K, info["iterations"], dt * 1e3, info["energy_initial"], info["energy_final"]))
t = time.time(); info = ctx.mvs_mrf_estimate(0, K, pk.data_ptr()); dt = time.time() - t
print("reference stopping rule: %d sweeps, %.2f ms" % (info["iterations"], dt * 1e3))
