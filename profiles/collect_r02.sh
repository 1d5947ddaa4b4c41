#!/bin/bash
# profiles/collect_r02.sh -- everything profiles/ holds for round 2, in one go on the GPU box:
#   bench lines (c1..c5, c3 in the fused and the fma variants), rocprofv3 kernel stats, PMC traffic and instruction mix.
# usage (from the repo root on the GPU box):  bash profiles/collect_r02.sh gpurun_out/r02
# PMC passes run on their own (never with --kernel-trace --stats in one rocprofv3 command), one counter set per pass.
set -u
OUT=${1:-gpurun_out/r02}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py"
for w in c1 c2 c3 c4 c5; do
	timeout -k 10 300 $B --workload $w --steps 5 --warmup 2 > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"
done
SRH_BENCH_FUSED=1 timeout -k 10 300 $B --workload c3 --steps 5 --warmup 2 --cpu-rows 0 > "$OUT/c3_fused_bench.json" 2>/dev/null
timeout -k 10 300 $B --workload c3 --arith fma --steps 5 --warmup 2 --cpu-rows 0 > "$OUT/c3_fma_bench.json" 2>/dev/null
timeout -k 10 300 $B --workload c2 --arith fma --steps 5 --warmup 2 --cpu-rows 0 > "$OUT/c2_fma_bench.json" 2>/dev/null
timeout -k 10 300 $B --workload c3 --arith f32 --steps 5 --warmup 2 --cpu-rows 0 > "$OUT/c3_f32_bench.json" 2>/dev/null
timeout -k 10 300 $B --workload c2 --arith f32 --steps 5 --warmup 2 --cpu-rows 0 > "$OUT/c2_f32_bench.json" 2>/dev/null
# MRF stage (1280x960, K = 9): wall clock per sweep count + per-kernel times
rocprofv3 --kernel-trace --stats -d "$OUT/stats_mrf" --output-format csv -- python3 profiles/mrf_timing.py > "$OUT/mrf_timing.log" 2>&1
cp "$(find "$OUT/stats_mrf" -name '*kernel_stats.csv' | head -1)" "$OUT/mrf_kernel_stats.csv" 2>/dev/null
grep -E "sweeps|stopping" "$OUT/mrf_timing.log" > "$OUT/mrf_timing.txt"
rm -rf "$OUT/stats_mrf"
# per-kernel time
for w in c3 c4 c5; do
	rocprofv3 --kernel-trace --stats -d "$OUT/stats_$w" --output-format csv -- $B --workload $w --steps 3 --warmup 1 --cpu-rows 0 > "$OUT/stats_$w.log" 2>&1
	cp "$(find "$OUT/stats_$w" -name '*kernel_stats.csv' | head -1)" "$OUT/${w}_kernel_stats.csv" 2>/dev/null
done
SRH_BENCH_FUSED=1 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3_fused" --output-format csv -- $B --workload c3 --steps 3 --warmup 1 --cpu-rows 0 > "$OUT/stats_c3_fused.log" 2>&1
cp "$(find "$OUT/stats_c3_fused" -name '*kernel_stats.csv' | head -1)" "$OUT/c3_fused_kernel_stats.csv" 2>/dev/null
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
pmc() { # tag, env assignment or "-", workload, counters...
	local tag=$1 envs=$2 w=$3; shift 3
	if [ "$envs" = "-" ]; then rocprofv3 --pmc "$@" -d "$OUT/pmc_$tag" --output-format csv -- $B --workload $w --steps 1 --warmup 0 --cpu-rows 0 > "$OUT/pmc_$tag.log" 2>&1
	else env $envs true; export $envs; rocprofv3 --pmc "$@" -d "$OUT/pmc_$tag" --output-format csv -- $B --workload $w --steps 1 --warmup 0 --cpu-rows 0 > "$OUT/pmc_$tag.log" 2>&1; unset ${envs%%=*}; fi
	cp "$(find "$OUT/pmc_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.csv" 2>/dev/null
}
for w in c3 c4 c5; do
	pmc ${w}_fetch - $w FETCH_SIZE
	pmc ${w}_write - $w WRITE_SIZE
done
pmc c3_fused_fetch SRH_BENCH_FUSED=1 c3 FETCH_SIZE
pmc c3_fused_write SRH_BENCH_FUSED=1 c3 WRITE_SIZE
# instruction mix of one C3 step
pmc c3_mix1 - c3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA
pmc c3_mix2 - c3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pmc c3_fused_mix1 SRH_BENCH_FUSED=1 c3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA
pmc c3_fused_mix2 SRH_BENCH_FUSED=1 c3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
python3 profiles/pmc_table.py "$OUT/pmc_c3_mix1.csv" "$OUT/pmc_c3_mix2.csv" > "$OUT/c3_instruction_mix.txt" 2>&1
python3 profiles/pmc_table.py "$OUT/pmc_c3_fused_mix1.csv" "$OUT/pmc_c3_fused_mix2.csv" > "$OUT/c3_fused_instruction_mix.txt" 2>&1
./profiles/microbench/fp64_sustained > "$OUT/fp64_sustained_mi355x.txt" 2>&1
# keep only the summaries (the raw rocprofv3 trees are large)
rm -rf "$OUT"/stats_c3 "$OUT"/stats_c4 "$OUT"/stats_c5 "$OUT"/stats_c3_fused "$OUT"/pmc_*_fetch "$OUT"/pmc_*_write "$OUT"/pmc_c3_mix1 "$OUT"/pmc_c3_mix2 "$OUT"/pmc_c3_fused_mix1 "$OUT"/pmc_c3_fused_mix2
ls "$OUT" | head -80
