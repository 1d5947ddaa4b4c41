#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: M disparity-hypotheses/s (W x H x D) of
the TwoView cost-volume / support-weight / WTA path on synthetic 1920x1080x256 pairs.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|small|c5|c4|c1]

A "step" is one pass of the hot path over one stereo pair per GPU: WTA left->right,
WTA right->left, cross-check, and the device-side hand-over of the two depth maps
(N>1: RCCL gather to rank 0).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (contract in the task description).  N>1: one rank per GPU, either
under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or,
invoked plainly as `python bench.py --gpus N`, with this process starting the N ranks itself as
child processes (self_launch); pairs are sharded (weak scaling), there is no data-path collective
other than the final gather.

--workload c4 is the MultiViewStereo configuration (8 views 1280x960, 128 levels, r=2): a step is
runTask over all 8 views -- initial estimates, the all-gather of the depth maps (N>1: views are
sharded over the ranks, strong scaling), the ordered cross-check chain.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

from stereoreconstruction_amd import capi, synthetic  # noqa: E402
from stereoreconstruction_amd.distributed import (HipMultiViewEngine, HipTwoViewBandEngine, multiview_sharded,  # noqa: E402
                                                  shard_units, twoview_rowbands_sharded)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (datasheet)
TV_OVERLAP = os.environ.get("SRH_BENCH_TV_OVERLAP", "1") != "0"   # srh_twoview_compute's two passes side by side (the library's default)

WORKLOADS = {
    # name: (W, H, D, weight_kind, seed, description)
    "c3": (1920, 1080, 256, capi.WEIGHT_GEODESIC, 0x5EED0003,
           "C3: synthetic rectified 1920x1080 pair, 256 depth levels, GeodesicWeight r=5"),
    "c2": (640, 480, 64, capi.WEIGHT_ADAPTIVE, 0x5EED0002,
           "C2: synthetic rectified 640x480 pair, 64 depth levels, AdaptiveWeight r=5"),
    "small": (320, 240, 64, capi.WEIGHT_GEODESIC, 0x5EED0009,
              "dev: synthetic rectified 320x240 pair, 64 depth levels, GeodesicWeight r=5"),
    # C5 geometry: C3 + planar refractive interface (normal = optical axis, distance 0.1, ratio 1.333):
    # curved epipolar lines -> the general curve-walk kernel
    "c5": (1920, 1080, 256, capi.WEIGHT_GEODESIC, 0x5EED0050,
           "C5: C3 geometry + refractive interface (dist 0.1, ratio 1.333), one pair per GPU"),
    # C1: the example project's bunny pair as the reference ingests it (tests/golden/bunny_pair.npz: Qt smooth
    # scaling 0.25, alpha masks, lens distortion; cameras 7310085/7310087), 100 levels -> general-geometry kernels
    "c1": (256, 192, 100, capi.WEIGHT_GEODESIC, 0,
           "C1: example 'bunny' pair 1024x768 at scale 0.25 (256x192), 100 depth levels 30-80, GeodesicWeight r=5, "
           "distorted verged cameras"),
    "c4": (1280, 960, 128, capi.WEIGHT_GEODESIC, 0x5EED0004,
           "C4: 8 views on a semicircle (22.5 deg apart) around a textured sphere, 1280x960, 128 uniform depth "
           "levels, MultiViewStereo r=2, 3 neighbours"),
    # C1M: the reference's live entry point on its own example data -- StereoWidget -> MultiViewStereo on every camera of the
    # project (gui/widgets/stereowidget.cpp:974-1002) -- the eight `bunny` views as the reference ingests them
    # (tests/golden/bunny_views.npz: Qt smooth scaling 0.25, alpha masks, lens distortion), 100 levels 30-80
    "c1m": (256, 192, 100, capi.WEIGHT_GEODESIC, 0,
            "C1M: the example project's 8 'bunny' views 1024x768 at scale 0.25 (256x192), 100 uniform depth levels 30-80, "
            "MultiViewStereo r=2, 3 neighbours, distorted cameras from the project's projection matrices"),
}
C4_VIEWS = 8


NO_FMA_CEILING_TLANE = 256 * 4 * 16 * 2.4e9 / 1e12   # separate v_mul_f64 / v_add_f64: 16 lanes per SIMD per clock = 39.3 T lane-op/s
SUSTAINED_NO_FMA_TLANE = 34.5                        # measured: profiles/microbench/fp64_sustained (2 waves/SIMD, 2344 MHz held under load)


def _pmc_section(fname, workload, mode, any_build=False):
    """The section of a committed PMC summary (profiles/) for this workload and arithmetic -- only if it was collected on
    the very build of the library that is loaded now (srh_build_id): counts of another build are not evidence for this run.
    (any_build: the section whatever its build -- the caller labels the figure as taken from another build's counts.)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", fname)))
    except Exception:
        return None
    sec = t.get(workload if mode in ("certified", None) else workload + "_" + mode)
    if not isinstance(sec, dict) or (sec.get("_build_id") != capi.build_id() and not any_build):
        return None
    return sec


def pmc_traffic(workload, kernel, mode=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (profiles/pmc_traffic.json), or None."""
    sec = _pmc_section("pmc_traffic.json", workload, mode)
    return sec.get(kernel) if sec else None


def pmc_executed(workload, kernel, avg_launch_ms, mode=None):
    """What the kernel EXECUTES, from the committed rocprofv3 --pmc instruction counts (profiles/pmc_instr.json:
    SQ_INSTS_VALU_MUL_F64 + ADD_F64 + FMA_F64 wave-instructions per launch, x64 lanes) over the launch time measured
    live, against the FP64 issue ceiling (one instruction per lane and clock, fused or not: 39.3 T lane-op/s)."""
    sec = _pmc_section("pmc_instr.json", workload, mode, any_build=True)
    e = sec.get(kernel) if sec else None
    if not e:
        return None
    lane_ops = 64.0 * (e["mul_f64"] + e["add_f64"] + e.get("fma_f64", 0))
    flops = 64.0 * (e["mul_f64"] + e["add_f64"] + 2 * e.get("fma_f64", 0))     # a fused multiply-add counts twice
    rate = lane_ops / (avg_launch_ms * 1e-3) / 1e12
    return {"f64_flops_per_launch": round(flops), "tflops": round(flops / (avg_launch_ms * 1e-3) / 1e12, 3),
            "frac_of_fp64_peak": round(flops / (avg_launch_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
            "counts_from_this_build": sec.get("_build_id") == capi.build_id(),
            "f64_lane_ops_per_launch": round(lane_ops), "rate": round(rate, 3), "unit": "T lane-op/s",
            "ceiling": round(NO_FMA_CEILING_TLANE, 2), "frac": round(rate / NO_FMA_CEILING_TLANE, 4),
            "sustained_ceiling": SUSTAINED_NO_FMA_TLANE, "frac_of_sustained": round(rate / SUSTAINED_NO_FMA_TLANE, 4),
            "wave_instr_per_launch": {k: e.get(k) for k in ("mul_f64", "add_f64", "fma_f64", "valu")},
            "pmc_build_id": sec.get("_build_id")}


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"model": model, "cores_online": os.cpu_count() or 1, "cores_visible": vis}


def run_c4(args, rank, world, dev, dev_index, backend):
    """MultiViewStereo::runTask on C4; returns the JSON dict on rank 0."""
    import torch
    import torch.distributed as dist
    wl = args.workload if args.workload in ("c4", "c1m") else "c4"
    W, H, D, wkind, seed, desc = WORKLOADS[wl]
    scale = 1.0
    if wl == "c1m":
        g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_views.npz"))
        rgba, masks, scale = list(g["rgba"]), list(g["mask"]), float(g["scale"][0])
        cams = [capi.camera_from_p(g["P"][v], g["dist"][v]) for v in range(C4_VIEWS)]
        zmin, zmax = 30.0, 80.0
    else:
        if os.environ.get("SRH_BENCH_C4_SMALL"):               # dev: quick functional run
            W, H, D = 320, 240, 32
        cams3 = synthetic.semicircle_rig(C4_VIEWS, W, H, radius=10.0, step_deg=22.5, focal=float(W))
        rgba, masks, _ = synthetic.render_sphere_views(cams3, W, H, seed, sphere_radius=2.0, tex_size=1024)
        cams = [capi.camera_from_krt(K, R, t) for (K, R, t) in cams3]
        zmin, zmax = 8.0, 12.0
    p = capi.params_mvs(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind, image_scale=scale,
                        cross_check_threshold=2 * (zmax - zmin) / (D - 1))
    neigh = capi.mvs_neighbours(cams, p)
    links = sum(len(n) for n in neigh)
    ctx = capi.Context(dev_index)
    if os.environ.get("SRH_BENCH_EXP_WALK_MODE"):               # timing experiment build only
        ctx.set_option("exp_walk_mode", int(os.environ["SRH_BENCH_EXP_WALK_MODE"]))
    if args.arith in ("fma", "f32"):
        sys.exit("--arith fma / f32 apply to the dense row-aligned TwoView path (c2, c3, small)")
    ctx.set_option("arith", capi.ARITH_EXACT if args.arith == "exact" else capi.ARITH_CERTIFIED)
    ctx.set_option("mvs_async", int(os.environ.get("SRH_MVS_ASYNC", "1")))   # (0: the rocprofv3 passes, one view in flight; the library itself reads no environment)
    for v in range(C4_VIEWS):                                  # every rank holds all views (a few MB): any neighbour
        ctx.upload_view(v, rgba[v], masks[v], cams[v])
    eng = HipMultiViewEngine(ctx, list(range(C4_VIEWS)), neigh, p, dev if backend == "nccl" else "cpu")

    def fence():
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def first_call():
        """A new image set (StereoWidget re-initializes the task per run, stereowidget.cpp:990-999): fresh context, the eight
        uploads untimed and fenced, then ONE runTask (estimates + all-gather + ordered cross-check) timed with a fence."""
        c = capi.Context(dev_index)
        c.set_option("arith", capi.ARITH_EXACT if args.arith == "exact" else capi.ARITH_CERTIFIED)
        c.set_option("mvs_async", int(os.environ.get("SRH_MVS_ASYNC", "1")))
        for v in range(C4_VIEWS):
            c.upload_view(v, rgba[v], masks[v], cams[v])
        e = HipMultiViewEngine(c, list(range(C4_VIEWS)), neigh, p, dev if backend == "nccl" else "cpu")
        c.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        multiview_sharded(e, C4_VIEWS)
        c.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        c.close()                                                  # (its band buffers wait in the process's pool for the next context)
        return round(ms, 3)

    for _ in range(args.warmup):
        multiview_sharded(eng, C4_VIEWS)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        multiview_sharded(eng, C4_VIEWS)
    fence()
    dt = time.perf_counter() - t0
    # per-kernel durations (roofline): the library keeps two views' estimates in flight on two streams, so the HIP events
    # around a kernel also time its neighbour's share of the GPU; an extra, untimed pass with one view at a time gives
    # each kernel's own duration
    prof_steps = min(args.steps, 2)
    ctx.set_option("mvs_async", 0)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(prof_steps):
        multiview_sharded(eng, C4_VIEWS)
    fence()
    ctx.profile_enable(False)
    ctx.set_option("mvs_async", int(os.environ.get("SRH_MVS_ASYNC", "1")))
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof = ctx.profile()
    first_call_samples = [first_call() for _ in range(3)] if world == 1 and not args.no_first_call else None
    first_call_ms = sorted(first_call_samples)[1] if first_call_samples else None      # (median of three fresh contexts)
    # cost evaluations the reference performs for this rank's views (untimed recount): only masked-in pixels are
    # matched and a curve has as many candidates as its pixel length, so W*H*D*links is neither a bound nor an estimate
    my_views = list(shard_units(C4_VIEWS, world, rank))
    n_eval = 0
    waves = [0, 0]
    for v in my_views:
        ctx.mvs_initial_estimate(v, neigh[v], p)
        st = ctx.stats()
        n_eval += st["n_eval"]
        waves[0] += st["mvs_waves_staged"]
        waves[1] += st["mvs_waves_listed"]
    result = None
    if rank == 0:
        hyp_per_step = W * H * D * links                        # nominal: every pixel x level x neighbour
        T = (2 * p.window_radius + 1) ** 2
        name, (ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
        my_links = sum(len(neigh[v]) for v in my_views)
        flops = n_eval * (15.0 * T + 8) * prof_steps             # rank 0's share, what its profile pass timed
        valu = flops / (ms * 1e-3) / 1e12
        alg_bytes = 14.0 * W * H * my_links * prof_steps
        hbm = alg_bytes / (ms * 1e-3) / 1e9
        executed = pmc_executed(wl, name, ms / launches) if not os.environ.get("SRH_BENCH_C4_SMALL") else None
        result = {
            "metric": "Mdisparity-hypotheses/s (WxHxD)", "value": round(hyp_per_step * args.steps / dt / 1e6, 3),
            "unit": "Mhyp/s", "build_id": capi.build_id(), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "first_call_ms": first_call_ms, "first_call_samples_ms": first_call_samples, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if wl == "c4" else "fixture (the example project's eight bunny views, Qt-scaled)",
            "config": {"workload": desc + "; initial estimates + depth-map all-gather + ordered cross-check",
                       "width": W, "height": H, "depth_levels": D, "views": C4_VIEWS, "view_neighbour_links": links,
                       "window_radius": int(p.window_radius), "weights": "geodesic",
                       "parallelism": ("views sharded, %s all-gather" % ("RCCL" if backend == "nccl" else backend))
                       if world > 1 else "single GPU",
                       "arithmetic": ("certified (default): fused sweeps in the staged cost kernel, every unit's winner certified against "
                                      "a proven error bound, its score recomputed and ambiguous units redone in the reference's arithmetic "
                                      "-- same bits as 'exact' (DESIGN.md 2b)" if args.arith == "certified"
                                      else "exact: the reference's operation order, no contraction, everywhere"),
                       "masked_in_fraction": round(float(np.mean([m.mean() for m in masks])), 4),
                       # which cost kernel did the work: 64-pixel waves (per neighbour link) evaluated from LDS copies of the
                       # other view (mvs_staged_cost_kernel) / by gathers (mvs_list_cost_kernel)
                       "cost_waves_staged": int(waves[0]), "cost_waves_gathering": int(waves[1]),
                       "n_eval_reference_rank0_per_step": int(n_eval)},
            "roofline": {"bound": "valu_fp64", "kernel": name,
                         "achieved": (executed or {}).get("tflops", round(valu, 3)), "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": (executed or {}).get("frac_of_fp64_peak", round(valu / FP64_VALU_PEAK_TFLOPS, 5)),
                         "frac_is": "executed" if executed else "nominal (no instruction counts for this kernel)",
                         "nominal_achieved": round(valu, 3), "nominal_frac": round(valu / FP64_VALU_PEAK_TFLOPS, 5),
                         "executed": executed,
                         "traffic": pmc_traffic(wl, name) if not os.environ.get("SRH_BENCH_C4_SMALL") else None,
                         "avg_launch_ms": round(ms / launches, 4), "launches": launches,
                         "alg_flops_per_launch": round(flops / launches), "flops_per_hyp": 15 * T + 8,
                         "note": "flops = cost evaluations of the reference (n_eval) x 383; value counts nominal W*H*D*links hypotheses; "
                                 "kernel durations from an untimed pass with one view in flight (option mvs_async 0), the timed steps keep two",
                         "hbm": {"achieved": round(hbm, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(hbm / HBM_PEAK_GBS, 6), "alg_bytes_per_launch": round(alg_bytes / launches),
                                 "traffic_over_algorithmic": None}},
            "kernels_ms": {k: [round(v[0], 3), v[1]] for k, v in sorted(prof.items())},
            "cpu_baseline": None,
        }
        if world == 1:
            # the nominal W*H*D*links counts every pixel, level and link; the reference evaluates a cost per CANDIDATE of a
            # masked-in pixel's curve: that is the work there is, and the rate to quote
            result["evaluated"] = {"value": round(n_eval * args.steps / dt / 1e6, 3), "unit": "M cost evaluations/s",
                                   "n_eval_reference_per_step": int(n_eval), "over_nominal": round(n_eval / hyp_per_step, 4)}
        if world == 1 and args.cpu_rows > 0:
            import oracle_ffi as O
            ocams = ([O.camera_set_p(g["P"][v], g["dist"][v]) for v in range(C4_VIEWS)] if wl == "c1m"
                     else [O.camera_set(K, R, t) for (K, R, t) in cams3])
            op = O.params_mvs(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind, image_scale=scale,
                              cross_check_threshold=2 * (zmax - zmin) / (D - 1))
            oimgs = [O.OImage(rgba[v], masks[v]) for v in range(C4_VIEWS)]
            rows = max(args.cpu_rows, 8)
            y0 = H // 2 - rows // 2
            t0 = time.perf_counter()
            want, n_eval = O.mvs_initial_estimate(oimgs, ocams, 0, neigh[0], op, y0, y0 + rows)
            cdt = time.perf_counter() - t0
            ctx.mvs_initial_estimate(0, neigh[0], p)
            got = ctx.download_depth(0)[y0:y0 + rows]
            want = want[y0:y0 + rows]
            fin = np.isfinite(got) & np.isfinite(want)
            same_cls = (np.isnan(got) == np.isnan(want)) & (np.isinf(got) == np.isinf(want))
            close = np.abs(got[fin] - want[fin]) <= 1e-9 * np.maximum(1.0, np.abs(want[fin]))
            result["cpu_baseline"] = dict(
                value=rows * W * D * len(neigh[0]) / cdt / 1e6, unit="Mhyp/s", cores=1, kind="port", host=host_cpu(),
                sample="oracle/sr_oracle.c sro_mvs_initial_estimate, view 0, %d centre rows x %d levels x %d neighbours "
                       "(%d cost evaluations) in %.2f s on 1 host thread" % (rows, D, len(neigh[0]), n_eval, cdt),
                parity_band={"pixels": int(got.size), "class_mismatch": int((~same_cls).sum()),
                             "value_mismatch": int((~close).sum())})
    ctx.close()
    return result


def cpu_baseline(L, R, ml, mr, cam_triples, zmin, zmax, D, weight_kind, rows, plane=None, scale=1.0, dists=(None, None),
                 all_cores=True):
    """Time the oracle (CPU restatement, one thread) on a centre row band of the same pair."""
    import oracle_ffi as O
    (Kl, Rl, tl), (Kr, Rr, tr) = cam_triples
    if plane is None:
        cl, cr = O.camera_set(Kl, Rl, tl, dists[0]), O.camera_set(Kr, Rr, tr, dists[1])
    else:
        cl, cr = O.camera_set(Kl, Rl, tl, None, *plane), O.camera_set(Kr, Rr, tr, None, *plane)
    p = O.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=weight_kind, image_scale=scale)
    li, ri = O.OImage(L, ml), O.OImage(R, mr)
    h, w = L.shape[:2]
    y0 = h // 2 - rows // 2
    t0 = time.perf_counter()
    depth, diag = O.twoview_wta(li, ri, cl, cr, p, y0, y0 + rows, want_diag=True)
    dt = time.perf_counter() - t0
    base = dict(value=rows * w * D / dt / 1e6, unit="Mhyp/s", cores=1, kind="port", host=host_cpu(),
                sample="oracle/sr_oracle.c sro_twoview_wta, left->right, %d full-width centre rows x %d levels "
                       "(%d cost evaluations) in %.2f s on 1 host thread" % (rows, D, diag["n_eval"], dt))
    # the same band on every host core of this box's share (the oracle is re-entrant; ctypes drops the GIL):
    # what the reference's own OpenMP-over-rows build could reach at best -- reported beside, not instead of,
    # the single-thread figure the metric names ("tbb/openmp off")
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    if cores > 1 and all_cores:
        ys = [max(0, min(h - rows, y0 + (i - cores // 2) * rows)) for i in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda yy: O.twoview_wta(li, ri, cl, cr, p, yy, yy + rows), ys))
        dta = time.perf_counter() - t0
        base["all_cores"] = dict(value=cores * rows * w * D / dta / 1e6, unit="Mhyp/s", cores=cores,
                                 sample="%d threads x %d rows each in %.2f s" % (cores, rows, dta))
    return base, depth, y0


def run_twoview(args, workload, rank, world, dev, dev_index, backend):
    """One TwoView workload (c1, c2, c3, c5, small); returns the JSON dict on rank 0."""
    import torch
    import torch.distributed as dist
    W, H, D, wkind, seed, desc = WORKLOADS[workload]
    # weak scaling: `world` pairs in the job, pairs sharded over ranks -> every rank owns one pair
    # (--shard rows: every rank holds the SAME pair and computes its band of rows)
    rows_shard = world > 1 and getattr(args, "shard", "pairs") == "rows"
    (unit,) = [0] if rows_shard else list(shard_units(world, world, rank))
    # C5: the 8 pairs of SURVEY 8(d) are seeds ...50-...57; other workloads: one independent pair per rank
    scale, dist_l, dist_r = 1.0, None, None
    if workload == "c1":
        g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_pair.npz"))
        L, R, ml, mr = g["left_rgba"], g["right_rgba"], g["left_mask"], g["right_mask"]
        cams3 = ((g["left_K"], g["left_R"], g["left_t"]), (g["right_K"], g["right_R"], g["right_t"]))
        zmin, zmax, scale = 30.0, 80.0, float(g["scale"][0])
        dist_l, dist_r = g["left_dist"], g["right_dist"]
    else:
        L, R, ml, mr, disp = synthetic.rectified_pair(W, H, D, seed + (unit if workload == "c5" else 0x10000 * unit))
        cams3 = synthetic.rectified_cameras(W, H)
        zmin, zmax = synthetic.rectified_depth_range(W, D)
    (Kl, Rl, tl), (Kr, Rr, tr) = cams3
    if workload == "c5":
        plane = (np.array([0.0, 0.0, 1.0]), 0.1, 1.333)
        cl = capi.camera_from_krt(Kl, Rl, tl, None, *plane)
        cr = capi.camera_from_krt(Kr, Rr, tr, None, *plane)
    else:
        cl, cr = capi.camera_from_krt(Kl, Rl, tl, dist_l), capi.camera_from_krt(Kr, Rr, tr, dist_r)
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind, image_scale=scale)

    if args.arith in ("fma", "f32") and workload in ("c1", "c4", "c5"):
        sys.exit("--arith fma / f32 apply to the dense row-aligned TwoView path (c2, c3, small)")
    arith_code = {"certified": capi.ARITH_CERTIFIED, "exact": capi.ARITH_EXACT, "fma": capi.ARITH_FMA, "f32": capi.ARITH_F32}[args.arith]

    def first_call(keep=None):
        """What a user of the drop-in sees for a NEW pair (TwoViewStereo computes a pair once per object,
        twoviewstereo.cpp:150-227): fresh srh_create + two uploads (untimed, fenced), then ONE srh_twoview_compute timed
        with a fence -- the padded / full / geo5 planes built on first use, every band allocation, the list paths' first
        sizing of their lists: everything the steady-state steps below no longer pay.  (keep: a list to hand the context back open)"""
        c = capi.Context(dev_index)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_option("arith", arith_code)
        c.upload_view(0, L, ml, cl)
        c.upload_view(1, R, mr, cr)
        c.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.twoview_compute_device(0, 1, p)
        c.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if keep is not None:
            keep.append(c)
        else:
            c.close()
        return round(ms, 3)

    def first_calls(n=3):
        """median of n fresh contexts, one after the other, + the samples.  Each is closed before the next is made: its band
        buffers wait in the process's pool (srh_api.hip) and serve the next context -- what a GUI that makes one stereo object
        per run sees from its second object on; the process's very first call (no pool yet) is first_call_cold_process_ms."""
        v = [first_call() for _ in range(n)]
        return sorted(v)[len(v) // 2], v

    # the process's very first call (C3 in the default run: kernels' code objects not yet resident either)
    first_call_cold = first_call() if not rows_shard and not args.no_first_call else None

    ctx = capi.Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for opt in ("fused", "exp_repeat", "exp_lds_pad", "exp_scan_mode", "strip", "geodma", "rows_masked", "f32_form"):   # tuning knobs (timing only / path choice; results are identical)
        if os.environ.get("SRH_BENCH_" + opt.upper()):
            ctx.set_option(opt, int(os.environ["SRH_BENCH_" + opt.upper()]))
    ctx.upload_view(0, L, ml, cl)
    ctx.upload_view(1, R, mr, cr)
    mismatch = None
    if args.arith != "exact" and workload not in ("c1", "c5") and not args.no_exact_check:
        # untimed: the same pass in the reference's arithmetic, to count the depths the chosen arithmetic changes
        # (certified: must be 0 -- it is the parity mode; fma / f32: the measured winner-mismatch rate)
        ctx.set_option("arith", capi.ARITH_EXACT)
        ctx.twoview_wta(0, 1, p)
        exact_l = ctx.download_depth(0)
        ctx.set_option("arith", arith_code)
        ctx.twoview_wta(0, 1, p)
        mismatch = float((exact_l.view(np.uint64) != ctx.download_depth(0).view(np.uint64)).mean())
    ctx.set_option("arith", arith_code)
    ctx.set_option("tv_overlap", 1 if TV_OVERLAP else 0)
    ctx.set_option("side_weights", 1 if TV_OVERLAP else 0)        # (SRH_BENCH_TV_OVERLAP=0, the rocprofv3 passes: every kernel by itself)
    # Depth hand-over: both maps are copied device-to-device into a staging tensor; with N > 1 ranks they are
    # gathered on rank 0 (RCCL over xGMI).  The gather of step k runs while step k+1 computes (two staging
    # buffers, async collective); everything is drained inside the timed region by fence().
    NBUF = 2
    outs = [torch.empty((2, H, W), dtype=torch.float64, device=dev) for _ in range(NBUF)]
    on_dev = backend == "nccl"
    recv = None
    if world > 1 and rank == 0:
        recv = [[torch.empty((2, H, W), dtype=torch.float64, device=dev if on_dev else "cpu") for _ in range(world)]
                for _ in range(NBUF)]
    pending = [None] * NBUF
    host_src = [None] * NBUF
    state = {"n": 0}

    band_engine = HipTwoViewBandEngine(ctx, p, dev if on_dev else "cpu") if rows_shard else None

    def step():
        if rows_shard:
            twoview_rowbands_sharded(band_engine, H)
            return
        b = state["n"] % NBUF
        state["n"] += 1
        if pending[b] is not None:                     # the gather that last used this staging buffer
            pending[b].wait()
            if on_dev:
                torch.cuda.current_stream().synchronize()   # the library writes the buffer from its own stream
            pending[b] = None
        ctx.twoview_compute_device(0, 1, p)            # srh_twoview_compute: WTA both ways + cross-check (TwoViewStereo::computeDepthMaps)
        ctx.copy_depth_to_device(0, outs[b][0].data_ptr(), H * W * 8)
        ctx.copy_depth_to_device(1, outs[b][1].data_ptr(), H * W * 8)
        if world > 1:
            ctx.synchronize()      # the library's stream and torch's / RCCL's streams are not ordered
            src = outs[b] if on_dev else outs[b].cpu()
            host_src[b] = src
            pending[b] = dist.gather(src, recv[b] if rank == 0 else None, dst=0, async_op=True)

    def fence():
        for b in range(NBUF):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    prof_timed = ctx.profile()
    stats = ctx.stats()
    first_call_warm, first_call_samples = first_calls() if not rows_shard and not args.no_first_call else (None, None)
    # per-kernel durations (roofline): srh_twoview_compute runs its two passes side by side on two streams, so the HIP
    # events around a kernel of the timed steps also time the other pass's share of the GPU; an extra, untimed run of the
    # same step with the passes one after the other (option tv_overlap 0) gives each kernel's own duration
    # (SRH_BENCH_TV_OVERLAP=0: the whole run that way -- the rocprofv3 passes of profiles/collect_r04.sh)
    psteps = args.steps
    if TV_OVERLAP and not rows_shard:
        psteps = min(args.steps, 2)
        ctx.set_option("tv_overlap", 0)
        ctx.set_option("side_weights", 0)                         # (row-run path: the windows kernel behind the list kernel, not beside it)
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(psteps):
            step()
        fence()
        ctx.profile_enable(False)
        ctx.set_option("tv_overlap", 1)
        ctx.set_option("side_weights", 1)
    prof = ctx.profile()
    certified = None
    if stats["n_certified"]:
        # (the last pass of the timed steps: right -> left)
        certified = {"pixels_scanned_on_fused_costs": stats["n_certified"], "flagged_and_redone_exactly": stats["n_flagged"],
                     "flagged_frac": round(stats["n_flagged"] / stats["n_certified"], 8)}
    # both passes, one untimed run each: which path every direction took and what its certified scan flagged (C1: the real
    # photographs of the example project -- the flagged fraction on natural texture)
    passes = []
    if not rows_shard and rank == 0:
        for a, b in ((0, 1), (1, 0)):
            ctx.twoview_wta(a, b, p)
            st = ctx.stats()
            passes.append({"direction": "left->right" if a == 0 else "right->left",
                           "path": "dense" if st["used_dense_path"] else "candidate lists",
                           "pixels": st["n_pixels"], "scanned_on_fused_costs": st["n_certified"], "flagged": st["n_flagged"],
                           "flagged_frac": round(st["n_flagged"] / st["n_certified"], 8) if st["n_certified"] else None,
                           "scan_tiles_template": st["scan_tiles_template"], "scan_tiles_walked": st["scan_tiles_walked"]})
        if certified is None and any(q["scanned_on_fused_costs"] for q in passes):
            certified = {}
        if certified is not None:
            certified["passes"] = passes
    hyp_per_step_per_gpu = 2 * W * H * D
    value = (1 if rows_shard else world) * hyp_per_step_per_gpu * args.steps / dt / 1e6

    result = None
    if rank == 0:
        # dominant kernel by accumulated HIP-event time on the launch stream
        name, (ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
        # algorithmic HBM bytes: 14 B per reference pixel per direction
        # (u8 RGB+mask of the reference view, u8 gray+mask of the other view, f64 depth out);
        # one step covers 2*W*H reference pixels, spread over `launches/steps` launches.
        share = (len(shard_units(H, world, 0)) / H) if rows_shard else 1.0   # what rank 0's profile covers of the pair
        alg_bytes_total = 14.0 * 2 * W * H * psteps * share
        bytes_per_launch = alg_bytes_total / launches
        avg_ms = ms / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        T = (2 * p.window_radius + 1) ** 2
        flops_per_step = hyp_per_step_per_gpu * (15.0 * T + 8) * share
        valu_achieved = flops_per_step * psteps / (ms * 1e-3) / 1e12
        # the PMC files were collected on the exact mode: no traffic / instruction figures are claimed for the opt-in modes
        pmc_mode = args.arith if stats["n_certified"] or args.arith != "certified" else "exact"   # what the dominant kernel really ran
        traffic = pmc_traffic(workload, name, pmc_mode)
        executed = pmc_executed(workload, name, avg_ms, pmc_mode) if args.arith != "f32" else None
        VALU_PEAK = 157.3 if args.arith == "f32" else FP64_VALU_PEAK_TFLOPS
        result = {
            "metric": "Mdisparity-hypotheses/s (WxHxD)", "value": round(value, 3), "unit": "Mhyp/s",
            "build_id": capi.build_id(), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            # ms_per_step is STEADY STATE (the same uploaded pair again, after the warm-up steps); first_call_ms is what a new pair
            # costs: a fresh context, uploads untimed, ONE srh_twoview_compute with a fence (measured after the timed steps, the
            # process's code objects resident); first_call_cold_process_ms: the same as this process's very first call
            "first_call_ms": first_call_warm, "first_call_samples_ms": first_call_samples, "first_call_cold_process_ms": first_call_cold,
            "higher_is_better": True, "scaling": "strong" if rows_shard else "weak", "vs_baseline": None,
            "dtype": ("f32" if args.arith == "f32" else "f64"), "data": "synthetic" if workload != "c1" else "fixture (example project's bunny pair, Qt-scaled)",
            "config": {"workload": desc + "; TwoView WTA both directions + cross-check; one pair per GPU",
                       "width": W, "height": H, "depth_levels": D, "window_radius": int(p.window_radius),
                       "weights": "geodesic" if wkind == capi.WEIGHT_GEODESIC else "adaptive",
                       "pairs_per_gpu": 1,
                       # (weak scaling: rank r computes its OWN pair -- C5: seed + r, SURVEY 8(d)'s eight pairs ...50 to ...57;
                       # other workloads: seed + 0x10000*r)
                       "pair_seed_rank0": seed, "pair_seed_of_rank_r": ("seed + r" if workload == "c5" else "seed + 0x10000*r") if not rows_shard else "one pair",
                       "parallelism": (("ONE pair in %d row bands, %s gather to rank 0, cross-check there" if rows_shard else "%.0s pairs sharded, %s gather")
                                                           % (world, "RCCL" if backend == "nccl" else backend)) if world > 1 else "single GPU",
                       "dense_path": bool(stats["used_dense_path"]),
                       "arithmetic": ("certified (default): fused multiply-adds in the strip kernel's cost loops, every WTA decision checked "
                                      "against a proven error bound, uncovered pixels redone in the reference's arithmetic -- same bits as "
                                      "'exact' (DESIGN.md 2b); kernels without a fused form run the reference's arithmetic" if args.arith == "certified"
                                      else "exact: the reference's operation order, no contraction, everywhere" if args.arith == "exact"
                                      else "fma: multiply-adds of the cost loops fused, UNCHECKED (opt-in, NOT a parity mode)" if args.arith == "fma"
                                      else "f32: cost loops in packed single precision (opt-in, NOT the parity mode, NOT the reference's precision)"),
                       "certified_scan": certified,
                       "winner_mismatch_vs_exact": mismatch,
                       "n_eval_reference_last_pass": stats["n_eval"],
                       "n_eval_device_last_pass": stats["n_eval_device"]},
            # The binding roof is the vector FP64 ALU (intensity ~3e4 flop/B; SQ_INSTS_MFMA = 0).
            # "achieved" prices the NOMINAL algorithmic flops (1823 per hypothesis at r=5) against the 78.6 TFLOP/s
            # FMA datasheet peak; "executed" is what the kernel really issues (PMC multiply + add lane-ops) against the
            # ceiling of a no-contraction instruction mix (39.3 T lane-op/s; 34.5 sustained under load).
            # The HBM view the metric asks for is in "hbm": algorithmic 14 B/pixel against 8 TB/s.
            # (the opt-in f32 mode runs on the FP32 vector ALU: 157.3 TFLOP/s)
            "roofline": {"bound": "valu_fp32" if args.arith == "f32" else "valu_fp64", "kernel": name,
                         # achieved / frac: what the kernel EXECUTES (PMC instruction counts of profiles/pmc_instr.json, a fused
                         # multiply-add two flops) over its launch time measured live; nominal_*: the reference's 15T+8 flops per
                         # hypothesis over the same time -- above one where the certified one-pass form retires them with fewer
                         # operations.  Without counts for the kernel (opt-in modes) achieved / frac fall back to the nominal figure.
                         "achieved": (executed or {}).get("tflops", round(valu_achieved, 3)), "peak": VALU_PEAK, "unit": "TFLOP/s",
                         "frac": (executed or {}).get("frac_of_fp64_peak", round(valu_achieved / VALU_PEAK, 5)),
                         "frac_is": "executed" if executed else "nominal (no instruction counts for this kernel / mode)",
                         "nominal_achieved": round(valu_achieved, 3), "nominal_frac": round(valu_achieved / VALU_PEAK, 5),
                         "traffic": traffic,
                         "executed": executed,
                         "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                         "alg_flops_per_launch": round(flops_per_step * psteps / launches),
                         "flops_per_hyp": 15 * T + 8,
                         "note": "executed (frac) and nominal (nominal_frac) flops against the FMA datasheet peak over the launch time measured in THIS run (HIP events"
                                 + ("; from %d untimed steps with the two passes one after the other, option tv_overlap 0: the timed steps run them "
                                    "side by side and a kernel's events then span the other pass's share of the GPU, kernels_ms_timed" % psteps
                                    if TV_OVERLAP and not rows_shard else "") + ")"
                                 + ("; nominal_frac above 1: the certified one-pass form retires the reference's nominal flops with fewer executed "
                                    "operations -- frac is the utilisation" if valu_achieved > VALU_PEAK else "") + "; 'traffic' and "
                                 "'executed' take their per-launch counts from the committed rocprofv3 --pmc passes (profiles/pmc_*.json), "
                                 "only the time they are divided by is live",
                         "hbm": {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved / HBM_PEAK_GBS, 6),
                                 "alg_bytes_per_launch": round(bytes_per_launch),
                                 "traffic_over_algorithmic": (round(traffic / bytes_per_launch, 1) if traffic else None)}},
            "kernels_ms": {k: [round(v[0], 3), v[1]] for k, v in sorted(prof.items())},
        }
        if TV_OVERLAP and not rows_shard:
            result["kernels_ms_timed"] = {k: [round(v[0], 3), v[1]] for k, v in sorted(prof_timed.items())}
        if world == 1 and args.cpu_rows > 0:
            base, cpu_depth, y0 = cpu_baseline(L, R, ml, mr, cams3, zmin, zmax, D, wkind,
                                                args.cpu_rows if workload != "c1" else max(args.cpu_rows, 24),
                                                (np.array([0.0, 0.0, 1.0]), 0.1, 1.333) if workload == "c5" else None,
                                                scale, (dist_l, dist_r), getattr(args, "cpu_all_cores", True))
            if workload == "c1":
                args.cpu_rows = max(args.cpu_rows, 24)
            # the timed CPU band doubles as a full-size parity spot check of the WTA pass
            ctx.twoview_wta(0, 1, p, y0, y0 + args.cpu_rows)
            got = ctx.download_depth(0)[y0:y0 + args.cpu_rows]
            want = cpu_depth[y0:y0 + args.cpu_rows]
            same_cls = (np.isnan(got) == np.isnan(want)) & (np.isinf(got) == np.isinf(want))
            fin = np.isfinite(got) & np.isfinite(want)
            close = np.abs(got[fin] - want[fin]) <= 1e-9 * np.maximum(1.0, np.abs(want[fin]))
            base["parity_band"] = {"pixels": int(got.size), "class_mismatch": int((~same_cls).sum()),
                                   "value_mismatch": int((~close).sum())}
            result["cpu_baseline"] = base
        else:
            result["cpu_baseline"] = None
    ctx.close()
    return result


def other_configs(args, rank, world, dev, dev_index, backend):
    """Short legs (3 timed steps after 2 warm-ups) of the other BASELINE configurations, run after the headline so that
    every config's number is timed by the same driver run: C5 (refractive pair), C4 (8-view MultiViewStereo), C2, C1.
    Each leg carries its own CPU-oracle band (parity spot check at full size + baseline rate)."""
    import copy
    out = {}
    for w, rows in (("c5", 2), ("c4", 8), ("c2", 4), ("c1", 24), ("c1m", 16)):
        a = copy.copy(args)
        # (two warm-up steps: the first learns the candidate-list capacities of a pair, the second is the first to queue both
        # passes side by side and allocates the second pass's band buffers)
        a.steps, a.warmup, a.cpu_rows, a.workload, a.cpu_all_cores = 3, 2, rows, w, False
        t0 = time.perf_counter()
        try:
            r = run_c4(a, rank, world, dev, dev_index, backend) if w in ("c4", "c1m") else run_twoview(a, w, rank, world, dev, dev_index, backend)
        except Exception as e:                                   # a leg must never take the headline line down with it
            out[w] = {"error": "%s: %s" % (type(e).__name__, e)}
            continue
        cb = r.get("cpu_baseline") or {}
        out[w] = {"workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "first_call_ms": r.get("first_call_ms"),
                  "value": r["value"], "unit": r["unit"],
                  "steps": a.steps, "warmup": a.warmup, "scaling": r["scaling"],
                  "roofline": {k: r["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_is", "nominal_frac", "avg_launch_ms", "launches")},
                  "n_eval_reference": r["config"].get("n_eval_reference_last_pass", r["config"].get("n_eval_reference_rank0_per_step")),
                  "evaluated": r.get("evaluated"), "certified_scan": r["config"].get("certified_scan"),
                  "cost_waves": {k: r["config"].get(k) for k in ("cost_waves_staged", "cost_waves_gathering")} if w in ("c4", "c1m") else None,
                  "winner_mismatch_vs_exact": r["config"].get("winner_mismatch_vs_exact"),
                  "kernels_ms": r["kernels_ms"], "parity_band": cb.get("parity_band"),
                  "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None,
                  "leg_seconds": round(time.perf_counter() - t0, 2)}
    return out


def flat_area_sweep(dev_index, steps=3, fracs=(0.0, 0.01, 0.05, 0.25, 0.60)):
    """The certified arithmetic off friendly input (VERDICT r4 #1c): the C3 pair with a growing share of its area made flat /
    saturated (synthetic.paint_flat_bands), one step = srh_twoview_compute, in the default (certified) arithmetic and in the
    reference's arithmetic everywhere.  Reported per fraction: ms per step of both, the share of pixels the certified scan
    flagged for the exact redo (last pass), and that the two modes' depth maps are the same bits.  No pass is ever repeated
    as a whole: the certified time must grow smoothly from its textured-image value towards the exact mode's."""
    W, H, D, wkind, seed, _ = WORKLOADS["c3"]
    L0, R0, ml, mr, _ = synthetic.rectified_pair(W, H, D, seed)
    (Kl, Rl, tl), (Kr, Rr, tr) = synthetic.rectified_cameras(W, H)
    zmin, zmax = synthetic.rectified_depth_range(W, D)
    p = capi.params_twoview(min_depth=zmin, max_depth=zmax, num_depth_levels=D, weight_kind=wkind)
    out = []
    with capi.Context(dev_index) as ctx:
        for frac in fracs:
            L, R, painted = synthetic.paint_flat_bands(L0, R0, frac)
            ctx.upload_view(0, L, ml, capi.camera_from_krt(Kl, Rl, tl))
            ctx.upload_view(1, R, mr, capi.camera_from_krt(Kr, Rr, tr))
            rec = {"flat_frac": frac, "rows_painted": painted}
            maps = {}
            for mode, code in (("certified", capi.ARITH_CERTIFIED), ("exact", capi.ARITH_EXACT)):
                ctx.set_option("arith", code)
                ctx.twoview_compute_device(0, 1, p)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ctx.twoview_compute_device(0, 1, p)
                ctx.synchronize()
                rec["ms_" + mode] = round((time.perf_counter() - t0) / steps * 1e3, 3)
                st = ctx.stats()
                if mode == "certified":
                    rec["flagged_frac_last_pass"] = round(st["n_flagged"] / max(1, st["n_certified"]), 8)
                    rec["pixels_on_fused_costs_last_pass"] = st["n_certified"]
                maps[mode] = (ctx.download_depth(0), ctx.download_depth(1))
            rec["same_bits_as_exact_mode"] = bool(all(np.array_equal(a.view(np.uint64), b.view(np.uint64))
                                                      for a, b in zip(maps["certified"], maps["exact"])))
            out.append(rec)
    return out


RANK_TIMEOUT_S = float(os.environ.get("SRH_BENCH_RANK_TIMEOUT_S", "300"))   # rendezvous + every collective of a rank


def comm_record(dist, backend, world, dev):
    """What the collective layer saw, machine-readable on the JSON line: backend, world size, the RCCL version torch's
    backend reports and the one the library's own transport would load (ncclGetVersion), and `ranks_seen` = an
    all-reduce of 1 over the job (1 without a process group)."""
    import torch
    rec = {"backend": backend if world > 1 else None, "world_size": world, "rccl_version": None, "rccl_version_library": None,
           "ranks_seen": 1, "rank_timeout_s": RANK_TIMEOUT_S if world > 1 else None}
    try:
        v = torch.cuda.nccl.version()
        rec["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else int(v)
    except Exception as e:                                        # (a torch build without the binding: say so, do not fail the run)
        rec["rccl_version"] = "unavailable: %s" % type(e).__name__
    try:
        from stereoreconstruction_amd import capi
        rec["rccl_version_library"] = capi.comm_version() or None
    except Exception:
        pass
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        rec["ranks_seen"] = int(one.item())
        if rec["ranks_seen"] != world:
            raise SystemExit("bench.py: all-reduce of 1 saw %d ranks of %d" % (rec["ranks_seen"], world))
    return rec


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one -- before
    anything here has initialised HIP or torch.cuda; a process that has touched the GPU is never exec'ed over -- with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR 127.0.0.1, a free
    MASTER_PORT).  Rank 0's JSON line is passed through on stdout; everything else the ranks print joins stderr.  If a
    rank fails, the others are terminated (their exact PIDs) and its exit code is returned."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def pump(stream):
        # stdout carries the JSON line and nothing else: what libraries print there (gloo's connection notes) joins stderr
        for line in stream:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    import signal
    import threading
    reader = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
    reader.start()

    def stop_ranks(grace=5.0):
        # the exact PIDs this launcher started: terminate, then kill what is still there after the grace period
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        t_end = time.monotonic() + grace
        for pr in procs:
            try:
                pr.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                pr.kill()

    # a launcher that is told to stop (timeout -k, the driver, Ctrl-C) takes its ranks with it: left alone they would keep
    # their GPUs and wait in a collective, under the next measurement on the box
    def on_signal(signum, _frame):
        stop_ranks()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_signal)
    # a rank that never arrives (rendezvous, a hung collective) must end the launch, not hang it: every rank's own
    # process-group timeout is RANK_TIMEOUT_S; the launcher gives the whole run a deadline as the last resort
    deadline = time.monotonic() + float(os.environ.get("SRH_BENCH_LAUNCH_DEADLINE_S", "1500"))
    code = 0
    alive = list(procs)
    try:
        while alive:
            for pr in list(alive):
                rc = pr.poll()
                if rc is None:
                    continue
                alive.remove(pr)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    for other in alive:                           # a rank is gone: the others would wait in a collective for ever
                        other.terminate()
            if alive and time.monotonic() > deadline:
                sys.stderr.write("bench.py: ranks still running at the launch deadline, stopping them\n")
                code = code or 124
                break
            if alive:
                time.sleep(0.05)
    finally:
        stop_ranks()
    reader.join(timeout=10)
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-rows", type=int, default=4, help="rows of the CPU-baseline band (0 = skip)")
    ap.add_argument("--arith", default="certified", choices=["certified", "exact", "fma", "f32"],
                    help="certified (default, the library's default): fused multiply-adds with every decision checked against an "
                         "error bound -- the exact mode's bits; exact: the reference's arithmetic operation by operation; fma / f32: "
                         "opt-in unchecked modes -- their winner-mismatch rate against the exact mode is measured and reported")
    ap.add_argument("--shard", default="pairs", choices=["pairs", "rows"],
                    help="N > 1, TwoView workloads: 'pairs' (default) = one pair per GPU, weak scaling; 'rows' = ONE pair cut "
                         "into N row bands, bands gathered on rank 0, cross-check there: strong scaling")
    ap.add_argument("--no-first-call", action="store_true",
                    help="skip the fresh-context first-call measurements (the rocprofv3 passes: only the steady-state launches are counted)")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the untimed pass in the reference's arithmetic that counts the depths the chosen arithmetic changes "
                         "(profiling passes: only the kernels of the timed steps are to be traced)")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline line only: skip the short driver-timed legs of the other BASELINE configurations")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (nothing here has touched HIP yet)
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    # SRH_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the
    # visible GPUs and the gather goes through host memory.  The measured configuration is "nccl" (RCCL).
    backend = os.environ.get("SRH_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import datetime
        # a finite timeout on the rendezvous and on every collective: a rank that never arrives makes the others fail
        # (non-zero exit of these child processes; the launcher then stops the rest), never wait for ever
        tmo = datetime.timedelta(seconds=RANK_TIMEOUT_S)
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")   # a timed-out collective tears the process down
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    comm = comm_record(dist, backend, world, dev) if rank == 0 or world > 1 else None

    if args.workload in ("c4", "c1m"):
        result = run_c4(args, rank, world, dev, dev_index, backend)
        if result is not None:
            result["comm"] = comm
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result))
        return

    result = run_twoview(args, args.workload, rank, world, dev, dev_index, backend)
    if result is not None:
        result["comm"] = comm
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_configs and args.arith in ("certified", "exact"):
        result["configs"] = other_configs(args, rank, world, dev, dev_index, backend)
        if args.arith == "certified" and result["config"].get("certified_scan"):
            result["config"]["certified_scan"]["flat_area_sweep"] = flat_area_sweep(dev_index)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
