#!/usr/bin/env python3
"""Regenerates profiles/README.md from the committed bench lines and rocprofv3 summaries of this round."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
L = lambda n: json.load(open(os.path.join(HERE, n)))
c1, c2, c3, c4, c5 = (L("r01_%s_bench.json" % w) for w in ("c1", "c2", "c3", "c4", "c5"))
s3, s4 = L("r01_c3_summary.json"), L("r01_c4_summary.json")
per = lambda d, k: d["kernels_ms"][k][0] / d["steps"]
txt = f"""# profiles/ — round 1 (one MI355X)

Commands profiled (the bench lines): `python3 bench.py --steps 3 --warmup 1 --cpu-rows 0` (C3, default workload),
`--workload c4` and `--workload c5` under `rocprofv3 --kernel-trace --stats`; `--steps 1 --warmup 0` for the PMC
passes (one counter per pass, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, as `MI355X_MICROARCH.md` §rocprofv3 prescribes).

| file | what |
|---|---|
| `r01_c{{1,2,3,4,5}}_bench.json` | bench lines of this round (`python3 bench.py [--workload …] --steps 5 --warmup 1`, CPU baseline + full-size parity band included) |
| `r01_c3_kernel_stats.csv`, `r01_c4_kernel_stats.csv`, `r01_c5_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv` per-kernel summaries |
| `r01_c3_summary.json`, `r01_c4_summary.json` | kernel stats joined with the `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (`summarize_rocprof.py`) |
| `r01_c3_instruction_mix.txt` | `rocprofv3 --pmc` passes over one C3 step (`SQ_INSTS_VALU/LDS/SALU/MFMA`, `SQ_INSTS_VALU_ADD_F64` / `MUL_F64` / `FMA_F64` / `MFMA_F64`, `SQ_WAVE_CYCLES`, `SQ_BUSY_CU_CYCLES`), summed per kernel (`pmc_table.py`) |
| `pmc_traffic.json` | HBM bytes per launch per kernel = (2·FETCH_SIZE + WRITE_SIZE)·1024 (gfx950 FETCH_SIZE counts half of wide reads); read by `bench.py` for `roofline.traffic` |
| `microbench/fp64_rate.hip`, `fp64_rate_mi355x.txt` | what the chip sustains for separate `v_mul_f64`+`v_add_f64` (the no-contraction mix): 37.6–38.4 T lane-instr/s at ≥2 waves/SIMD, FMA 62.7–70.7 TFLOP/s |
| `summarize_rocprof.py`, `pmc_table.py`, `mvs_bench.py`, `make_readme.py` | the scripts that made the summaries and this file |

## Round-1 numbers

### C3 (headline): synthetic rectified 1920×1080 pair, 256 levels, GeodesicWeight r=5

* **{c3['value']/1e3:.1f} G hypotheses/s** ({c3['value']:.0f} Mhyp/s; {c3['ms_per_step']:.1f} ms per step = both directions + cross-check +
  device hand-over), dense path, one row band per direction (8 GB scratch budget), `n_eval` reference
  {c3['config']['n_eval_reference_last_pass']/1e6:.1f} M / device {c3['config']['n_eval_device_last_pass']/1e6:.1f} M per direction (joint duplicates evaluated once).
* CPU baseline (oracle, 1 thread of the GPU box's host, 2 full-width rows × 256 levels): **{c3['cpu_baseline']['value']:.3f} Mhyp/s**; the
  same band compared at full size: {c3['cpu_baseline']['parity_band']['class_mismatch']} class mismatches, {c3['cpu_baseline']['parity_band']['value_mismatch']} value mismatches.
* Dominant kernel `twoview_dense_cost_kernel`: {c3['roofline']['avg_launch_ms']:.2f} ms per launch by HIP events in `bench.py`,
  {s3['twoview_dense_cost_kernel']['avg_us']/1e3:.2f} ms by rocprofv3 (2 launches per step) = {s3['twoview_dense_cost_kernel']['pct']:.0f} % of device time.
  * nominal FP64 work `15·T+8 = 1823` flop/hypothesis ⇒ {c3['roofline']['achieved']:.1f} TFLOP/s = **{100*c3['roofline']['frac']:.0f} % of the 78.6 TFLOP/s vector-FP64
    datasheet peak**; the kernel issues 968 multiply/add per hypothesis (contraction off), for which the chip's
    measured ceiling is ≈38 T lane-instr/s ⇒ 17.7 ms per direction against {c3['roofline']['avg_launch_ms']:.1f} ms measured (**{100*17.7/c3['roofline']['avg_launch_ms']:.0f} %**).
  * HBM: algorithmic 14 B/pixel ⇒ {c3['roofline']['hbm']['alg_bytes_per_launch']/1e6:.1f} MB per launch; measured traffic {c3['roofline']['traffic']/1e6:.0f} MB per launch —
    the staged cost rows and support windows — i.e. {c3['roofline']['hbm']['traffic_over_algorithmic']:.0f}× the algorithmic bytes, at ≈300 GB/s (3.8 % of HBM peak): it
    does not limit the kernel (DESIGN.md §9 item 1 has the fusion analysis).
* Instruction mix of the two launches of one step (`r01_c3_instruction_mix.txt`): 19.58 G VALU wave-instructions, of
  which 8.33 G `v_mul_f64` + 8.33 G `v_add_f64` (85 %; 1004 per nominal hypothesis against the 968 of the fast form),
  0.36 G `v_fma_f64` (inside the compiler's division / square-root sequences only), 1.27 G LDS instructions,
  **0 MFMA**; 19.58 G × 4 cycles on 1024 SIMDs for 40.5 ms at 2.4 GHz = 79 % of the VALU issue slots.
* CPU, all cores: the same band on {c3['cpu_baseline']['all_cores']['cores']} host threads runs at {c3['cpu_baseline']['all_cores']['value']:.2f} Mhyp/s.
* `twoview_scan_kernel` {per(c3,'twoview_scan_kernel')/2:.2f} ms / launch ({s3['twoview_scan_kernel']['hbm_bytes_per_launch_corrected']/1e9:.1f} GB per launch since the cost rows are stored
  tile-transposed — the 32 pixels' k-th costs contiguous; 11.3 GB before, when every lane pulled its own 64-byte sectors),
  `geodesic_reg_kernel` {per(c3,'geodesic_reg_kernel')/2:.2f} ms / launch.

### C4: MultiViewStereo, 8 views 1280×960, 128 levels, r=2, 3 neighbours (`--workload c4`)

* **{c4['ms_per_step']:.1f} ms per `runTask`** (8 initial estimates + 8 cross-checks) = {c4['value']/1e3:.1f} G nominal hypotheses/s
  (W·H·D·24 links).  The sphere mask covers {100*c4['config']['masked_in_fraction']:.1f} % of the pixels and a curve has 186–402 candidates (its
  pixel length), so the reference performs {c4['config']['n_eval_reference_rank0_per_step']/1e9:.2f} G cost evaluations per step; on those,
  `mvs_list_cost_kernel` runs at {c4['roofline']['achieved']:.1f} TFLOP/s = {100*c4['roofline']['frac']:.0f} % of the FP64 peak ({per(c4,'mvs_list_cost_kernel'):.1f} ms per step),
  `mvs_walk_kernel` {per(c4,'mvs_walk_kernel'):.1f} ms per step.
* CPU baseline (oracle, 1 thread, view 0, 8 centre rows): {c4['cpu_baseline']['value']:.2f} Mhyp/s; same band on the GPU: 0 mismatches.
* History this round (ms per step): 337.8 (general kernel) → 99.7 (`mvs_reg_kernel<2>`, all in registers, 320 VGPR,
  1 wave/SIMD) → 73.0 (winner depth computed lazily: `closestPoints` was a third of the time) → 49.0 (walk and cost
  split into two kernels) → {c4['ms_per_step']:.1f} (weights in LDS, 8 waves/CU).

### C5: C3 geometry + refractive interface (curved epipolar lines, `--workload c5`)

* **{c5['value']/1e3:.2f} G hyp/s, {c5['ms_per_step']:.0f} ms per pair** (row-run candidate lists: cost {per(c5,'twoview_rows_cost_kernel'):.0f} ms, refractive curve walk
  {per(c5,'twoview_rows_list_kernel'):.0f} ms, scan {per(c5,'twoview_rows_scan_kernel'):.1f} ms per step).  History: 1 497 ms (one thread per pixel) → 273 ms (lists evaluated in
  list order, 242 gathers per candidate) → 189 ms (row runs) → 144 ms (one band) → 126 ms (wave-tiled lists and
  tile-transposed cost slots: scan 18 → {per(c5,'twoview_rows_scan_kernel'):.1f} ms) → {c5['ms_per_step']:.0f} ms (select-form blocks compacted over the tile).  Full-width band vs oracle: 0 mismatches.

### C1: the example project's bunny pair (`--workload c1`)

* 1024×768 at scale 0.25 (256×192, Qt-scaled fixture `tests/golden/bunny_pair.npz`), 100 levels over depth 30–80,
  distorted verged cameras: **{c1['ms_per_step']:.1f} ms per pair** (both directions + cross-check), {c1['value']/1e3:.2f} G nominal hyp/s; left→right
  takes the row-run list kernels, right→left has steep curves (> 32 image rows) and takes the list-order kernels —
  the choice is learnt on the first run of a view pair.  CPU baseline (oracle, 1 thread, 24 rows through the
  object): {c1['cpu_baseline']['value']:.3f} Mhyp/s; same band on the GPU: 0 mismatches.

### C2: 640×480, 64 levels, AdaptiveWeight r=5 (`--workload c2`)

* {c2['value']/1e3:.1f} G hyp/s, {c2['ms_per_step']:.2f} ms per pair (small problem: 4 800 tiles per direction, launch-bound tails).

History of the dense kernel in this round (ms per C3 direction): 53.7 (first LDS-tiled version) → 33.5 (large row
bands) → 30.8 (conflict-free 16-byte LDS layout) → 21.9 (balanced two-phase general path; border blocks were
costing 36 % through tail imbalance) → 21.1 → {c3['roofline']['avg_launch_ms']:.1f} (one band per direction).  Per-phase stamps
(`SRH_DENSE_DBG=2`) showed the block loops at ≈98 % of the FP64 issue rate; the remaining loss is staging, the
single-lane per-pixel prologue and the barrier at the end of a tile.
"""
open(os.path.join(HERE, "README.md"), "w").write(txt)
