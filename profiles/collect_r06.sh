#!/bin/bash
# profiles/collect_r06.sh -- everything profiles/ holds for round 6, in one go on the GPU box:
#   the default bench line (C3 headline + driver-style legs of C5, C4, C2, C1), per-workload bench lines with their CPU
#   baselines (default = certified arithmetic) and the same in the reference's arithmetic (--arith exact), rocprofv3 kernel
#   stats, PMC traffic and instruction mix of one step of C3 / C4 / C5, phase stamps and the loop-repeat experiment of the
#   fused strip kernel.
# usage (from the repo root on the GPU box):  bash profiles/collect_r06.sh gpurun_out/r06
# PMC passes run on their own (never with --kernel-trace --stats in one rocprofv3 command), one counter set per pass.
set -u
OUT=${1:-gpurun_out/r06}
PART=${2:-all}        # all | a (bench lines, kernel stats, experiments, phase stamps) | b (the PMC passes): one gpurun call each
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py"
if [ "$PART" != b ]; then
timeout -k 10 400 $B --steps 20 --warmup 3 > "$OUT/default_bench.json" 2> "$OUT/default_bench.err"
for w in c1 c2 c3 c4 c5 c1m; do
	timeout -k 10 300 $B --workload $w --steps 5 --warmup 2 --no-configs > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"
	timeout -k 10 300 $B --workload $w --steps 5 --warmup 2 --no-configs --arith exact --cpu-rows 0 > "$OUT/${w}_exact_bench.json" 2>/dev/null
done
timeout -k 10 300 $B --workload c3 --arith fma --steps 5 --warmup 2 --cpu-rows 0 --no-configs > "$OUT/c3_fma_bench.json" 2>/dev/null
echo "bench lines done"
# per-kernel time (C4: one view in flight, SRH_MVS_ASYNC=0: bench.py takes its kernels_ms from such a pass as well)
export SRH_MVS_ASYNC=0
# (and the two TwoView passes one after the other: side by side, a kernel's duration includes the other pass's share of the GPU)
export SRH_BENCH_TV_OVERLAP=0
for w in c3 c4 c5 c2 c1; do
	rocprofv3 --kernel-trace --stats -d "$OUT/stats_$w" --output-format csv -- $B --workload $w --steps 3 --warmup 1 --cpu-rows 0 --no-configs --no-exact-check --no-first-call > "$OUT/stats_$w.log" 2>&1
	cp "$(find "$OUT/stats_$w" -name '*kernel_stats.csv' | head -1)" "$OUT/${w}_kernel_stats.csv" 2>/dev/null
	rm -rf "$OUT/stats_$w"
done
echo "kernel stats done"
unset SRH_MVS_ASYNC SRH_BENCH_TV_OVERLAP
fi
pmc() { # tag workload counters...
	local tag=$1 w=$2; shift 2
	rocprofv3 --pmc "$@" -d "$OUT/pmc_$tag" --output-format csv -- $B --workload $w --steps 1 --warmup 0 --cpu-rows 0 --no-configs --no-exact-check --no-first-call > "$OUT/pmc_$tag.log" 2>&1
	cp "$(find "$OUT/pmc_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.csv" 2>/dev/null
	grep -o '"build_id": "[0-9a-f]*"' "$OUT/pmc_$tag.log" | head -1 > "$OUT/pmc_$tag.build"
	rm -rf "$OUT/pmc_$tag"
	echo "pass $tag done"
}
if [ "$PART" != a ]; then
export SRH_MVS_ASYNC=0
export SRH_BENCH_TV_OVERLAP=0
for w in c3 c4 c5 c2 c1; do
	pmc ${w}_fetch $w FETCH_SIZE
	pmc ${w}_write $w WRITE_SIZE
	pmc ${w}_mix1 $w SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA
	pmc ${w}_mix2 $w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
done
unset SRH_MVS_ASYNC SRH_BENCH_TV_OVERLAP
for w in c3 c4 c5 c2 c1; do python3 profiles/pmc_table.py $(ls "$OUT"/pmc_${w}_mix*.csv 2>/dev/null) > "$OUT/${w}_instruction_mix.txt" 2>/dev/null; done
fi
if [ "$PART" != b ]; then
# this round's experiments: the certified arithmetic on flat / saturated areas; what a pair / an image set costs the first time
# (fresh context, the list paths' capacity decisions traced); the FP64-MFMA and FMA-operand-bank microbenchmarks
timeout -k 10 300 python3 profiles/cert_flat_sweep.py > "$OUT/cert_flat_sweep.json" 2>/dev/null
for w in c5 c3 c2; do timeout -k 10 120 python3 profiles/exp_r06_first_call.py $w trace 2>&1 | grep -v amdgpu.ids | cut -c1-400; done > "$OUT/first_call.txt"
timeout -k 10 120 python3 profiles/exp_r06_first_call_mvs.py trace 2>&1 | grep -v amdgpu.ids | cut -c1-400 > "$OUT/first_call_mvs.txt"
[ -x profiles/microbench/mfma_f64_rate ] && timeout -k 10 120 profiles/microbench/mfma_f64_rate > "$OUT/mfma_f64_rate.txt" 2>&1
[ -x profiles/microbench/fma_bank ] && timeout -k 10 120 profiles/microbench/fma_bank > "$OUT/fma_bank.txt" 2>&1
# phase stamps of the strip kernel, the staged cost kernel and the geodesic kernel (diagnostic build), device busy time
if [ -f profiles/lib/libstereo_recon_hip_prof.so ]; then
	bash profiles/exp_r06_phases_busy.sh "$OUT" > /dev/null 2>&1
	SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so timeout -k 10 200 $B --workload c3 --steps 2 --warmup 1 --cpu-rows 0 --no-configs --no-first-call > /dev/null 2> "$OUT/phases_c3.err"
	grep "srh dbg" "$OUT/phases_c3.err" | grep -v rows | tail -9 > "$OUT/c3_strip8_phases_certified.txt"
fi
fi
rm -f "$OUT"/pmc_*.log "$OUT"/stats_*.log "$OUT"/phases_*.json "$OUT"/phases_*.err
ls "$OUT"
