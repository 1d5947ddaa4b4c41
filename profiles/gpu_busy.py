#!/usr/bin/env python3
"""GPU busy time from a rocprofv3 --kernel-trace CSV: the union of the kernels' [start, end) intervals against the span
from the first start to the last end of a region of the trace, the largest gaps and what follows them.

  gpu_busy.py kernel_trace.csv                      the whole trace
  gpu_busy.py kernel_trace.csv NAME I J             from the dispatch after the I-th to the J-th dispatch (1-based) of
                                                    the kernel whose name contains NAME -- e.g. `mvs_cross_check 16 24`
                                                    is the third step of a C4 run (8 cross-checks end a step).  A bench.py
                                                    run ends in untimed passes (per-kernel durations, recounts): "the last N
                                                    dispatches" is not the timed region.
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows)
if len(sys.argv) > 4:
    hits = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
    ev = ev[hits[int(sys.argv[3]) - 1] + 1: hits[int(sys.argv[4]) - 1] + 1]
t0 = ev[0][0]
span = max(e for _, e, _ in ev) - t0
busy, cur, gaps = 0, t0, []
for s, e, n in ev:
    if s > cur:
        gaps.append((s - cur, n))
    busy += max(0, e - max(s, cur))
    cur = max(cur, e)
print(f"dispatches {len(ev)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %)  "
      f"idle {(span-busy)/1e6:.3f} ms in {len(gaps)} gaps")
for g, n in sorted(gaps, reverse=True)[:10]:
    print(f"  gap {g/1e3:8.1f} us  before {n}")
