#!/bin/bash
# profiles/collect_r03_pmc.sh -- round 3: instruction mix / stall / cache counters of the C4 and C5 cost kernels
# (and C3 for reference), one rocprofv3 --pmc pass per counter set, never combined with a trace.
# usage (repo root, GPU box): bash profiles/collect_r03_pmc.sh gpurun_out/r03pmc "c4 c5 c3"
set -u
OUT=${1:-gpurun_out/r03pmc}
WL=${2:-"c4 c5 c3"}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
pmc() { # tag workload counters...
	local tag=$1 w=$2; shift 2
	rocprofv3 --pmc "$@" -d "$OUT/pmc_$tag" --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --cpu-rows 0 --no-configs > "$OUT/pmc_$tag.log" 2>&1
	cp "$(find "$OUT/pmc_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.csv" 2>/dev/null
	rm -rf "$OUT/pmc_$tag"
	echo "pass $tag done: $(wc -l < "$OUT/pmc_$tag.csv" 2>/dev/null) rows"
}
for w in $WL; do
	pmc ${w}_mix1 $w SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM
	pmc ${w}_mix2 $w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
	pmc ${w}_mix3 $w SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT
	pmc ${w}_tcp $w TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
	pmc ${w}_tcc $w TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
	python3 profiles/pmc_table.py "$OUT"/pmc_${w}_*.csv > "$OUT/${w}_pmc_table.txt" 2>&1
done
ls "$OUT"
