#!/bin/bash
# instruction-fetch counters of the geodesic kernel (C3, one step)
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST_ANY\|SQC_TC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail_if.txt
echo >> $OUT/avail_if.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
	tag=$(echo $set | cut -d' ' -f1)
	timeout -k 10 300 rocprofv3 --pmc $set -d $OUT/pmc_if_$tag --output-format csv -- python3 bench.py --workload c3 --steps 1 --warmup 0 --cpu-rows 0 --no-configs --no-exact-check > $OUT/pmc_if_$tag.log 2>&1 || { tail -5 $OUT/pmc_if_$tag.log; continue; }
	f=$(find $OUT/pmc_if_$tag -name '*counter_collection.csv' | head -1)
	python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    if "geodesic" in k or "strip" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in acc.items():
    print(k, {c: int(v) for c, v in d.items()})
PY
	rm -rf $OUT/pmc_if_$tag
done
cat $OUT/avail_if.txt
