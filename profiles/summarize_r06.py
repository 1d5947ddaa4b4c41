#!/usr/bin/env python3
"""profiles/summarize_r06.py <dir with the output of collect_r06.sh>: copies the round's summaries into profiles/ as
r06_* and rebuilds pmc_traffic.json (HBM bytes per launch per kernel) and pmc_instr.json (FP64 multiply / add / fma
wave-instructions per launch) that bench.py reads for roofline.traffic / roofline.executed.

Every workload section records the BUILD (srh_build_id: hash of the library's sources) and the arithmetic it was collected
on; bench.py attaches a section's counts to a run only when the loaded library is that very build.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB and were collected in separate
passes.  On gfx950 FETCH_SIZE reports half of the bytes of WIDE coalesced reads (16 bytes per lane): the read side is
doubled for the kernels whose loads are of that kind (WIDE_READERS below) and taken as it is for the others (8-byte
gathers, byte loads), for which the guide's correction does not apply."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = sys.argv[1]

# kernels whose global reads are 16 bytes per lane (LDS-DMA dwordx4 / dwordx4 loads of whole rows)
WIDE_READERS = {"twoview_strip_cost_kernel", "geodesic_reg_kernel", "geodesic_dma_kernel", "padded_plane_kernel", "fill_kernel", "__amd_rocclr_copyBuffer"}   # (the template scan reads 8 bytes per lane: not doubled)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("srh::", "").split("<")[0]


def per_launch(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    if not os.path.exists(path):
        return {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def build_of(tag):
    try:
        txt = open(os.path.join(SRC, "pmc_%s.build" % tag)).read()
        return txt.split('"')[3]
    except Exception:
        return None


traffic, instr = {}, {}
for tag in ("c3", "c4", "c5", "c2", "c1"):
    f = per_launch(os.path.join(SRC, "pmc_%s_fetch.csv" % tag), "FETCH_SIZE")
    w = per_launch(os.path.join(SRC, "pmc_%s_write.csv" % tag), "WRITE_SIZE")
    builds = {build_of("%s_%s" % (tag, p)) for p in ("fetch", "write", "mix1")}
    if len(builds) != 1 or None in builds:
        print("section %s: passes of different / unknown builds %s: skipped" % (tag, builds))
        continue
    (bid,) = builds
    sec = {k: round(((2.0 if k in WIDE_READERS else 1.0) * f.get(k, 0.0) + w.get(k, 0.0)) * 1024) for k in sorted(set(f) | set(w))}
    sec["_build_id"], sec["_arith"] = bid, "certified"
    sec["_fetch_doubled_for"] = sorted(k for k in sec if k in WIDE_READERS)
    traffic[tag] = sec
    path = os.path.join(SRC, "pmc_%s_mix1.csv" % tag)
    m, a, fm = (per_launch(path, c) for c in ("SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64"))
    v = per_launch(path, "SQ_INSTS_VALU")
    isec = {k: {"mul_f64": round(m.get(k, 0)), "add_f64": round(a.get(k, 0)), "fma_f64": round(fm.get(k, 0)), "valu": round(v.get(k, 0))}
            for k in sorted(set(m) | set(fm)) if m.get(k, 0) + a.get(k, 0) + fm.get(k, 0) > 1e6}
    isec["_build_id"], isec["_arith"] = bid, "certified"
    instr[tag] = isec
traffic["_source"] = ("round 6, profiles/collect_r06.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, bench.py --steps 1), bytes = "
                      "(k*FETCH_SIZE + WRITE_SIZE)*1024 per launch, k = 2 for the kernels listed in _fetch_doubled_for (16-byte-per-lane reads: the "
                      "gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md), k = 1 for the others")
instr["_source"] = ("round 6, profiles/collect_r06.sh: rocprofv3 --pmc SQ_INSTS_VALU_MUL_F64 / ADD_F64 / FMA_F64 / SQ_INSTS_VALU, "
                    "wave-instructions per launch, averaged over the launches of one bench.py --steps 1 run")
json.dump(traffic, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
json.dump(instr, open(os.path.join(HERE, "pmc_instr.json"), "w"), indent=1, sort_keys=True)

for name in sorted(os.listdir(SRC)):
    if name.endswith("_bench.json") or name.endswith("_kernel_stats.csv") or name.endswith("_instruction_mix.txt") \
            or "_phases" in name or name in ("cert_flat_sweep.json", "first_call.txt", "first_call_mvs.txt", "gpu_busy.txt"):
        shutil.copy(os.path.join(SRC, name), os.path.join(HERE, "r06_" + name))
for name, dst in (("mfma_f64_rate.txt", "microbench/mfma_f64_rate_mi355x.txt"), ("fma_bank.txt", "microbench/fma_bank_mi355x.txt")):
    if os.path.exists(os.path.join(SRC, name)) and os.path.getsize(os.path.join(SRC, name)) > 200:
        shutil.copy(os.path.join(SRC, name), os.path.join(HERE, dst))
print(json.dumps({k: v for k, v in traffic.items() if k != "_source"}, indent=1)[:2500])
print(json.dumps({k: v for k, v in instr.items() if k != "_source"}, indent=1)[:2500])
