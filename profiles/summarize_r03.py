#!/usr/bin/env python3
"""profiles/summarize_r03.py <dir with the output of collect_r03.sh>: copies the round's summaries into profiles/ as
r03_* and rebuilds pmc_traffic.json (HBM bytes per launch per kernel) and pmc_instr.json (FP64 multiply / add / fma
wave-instructions per launch) that bench.py reads for roofline.traffic / roofline.executed.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB and were collected in separate
passes; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so the read side is doubled."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = sys.argv[1]


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


traffic = {}
for tag in ("c3", "c4", "c5"):
    f = per_launch(os.path.join(SRC, "pmc_%s_fetch.csv" % tag), "FETCH_SIZE")
    w = per_launch(os.path.join(SRC, "pmc_%s_write.csv" % tag), "WRITE_SIZE")
    traffic[tag] = {k: round((2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024) for k in sorted(set(f) | set(w))}
traffic["_source"] = ("round 3, profiles/collect_r03.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, bench.py --steps 1), "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)")
json.dump(traffic, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)

instr = {}
for tag in ("c3", "c4", "c5"):
    path = os.path.join(SRC, "pmc_%s_mix1.csv" % tag)
    m, a, f = (per_launch(path, c) for c in ("SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64"))
    v = per_launch(path, "SQ_INSTS_VALU")
    instr[tag] = {k: {"mul_f64": round(m[k]), "add_f64": round(a.get(k, 0)), "fma_f64": round(f.get(k, 0)), "valu": round(v.get(k, 0))}
                  for k in sorted(m) if m[k] + a.get(k, 0) > 1e6}
instr["_source"] = ("round 3, profiles/collect_r03.sh: rocprofv3 --pmc SQ_INSTS_VALU_MUL_F64 / ADD_F64 / FMA_F64 / SQ_INSTS_VALU, "
                    "wave-instructions per launch, averaged over the launches of one bench.py --steps 1 run")
json.dump(instr, open(os.path.join(HERE, "pmc_instr.json"), "w"), indent=1, sort_keys=True)

for name in sorted(os.listdir(SRC)):
    if name.endswith("_bench.json") or name.endswith("_kernel_stats.csv") or name.endswith("_instruction_mix.txt") \
            or name.endswith("_phases.txt") or name == "c3_repeat_experiment.txt":
        shutil.copy(os.path.join(SRC, name), os.path.join(HERE, "r03_" + name))
for name in ("fp64_sustained_mi355x.txt", "fp64_dep_distance_mi355x.txt", "fp64_chain_latency_mi355x.txt", "lds_dma_alignment_mi355x.txt"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(HERE, "microbench", name))
print(json.dumps({k: v for k, v in traffic.items() if k != "_source"}, indent=1)[:3000])
print(json.dumps({k: v for k, v in instr.items() if k != "_source"}, indent=1)[:3000])
