#!/bin/bash
# profiles/collect_r03.sh -- everything profiles/ holds for round 3, in one go on the GPU box:
#   the default bench line (C3 headline + driver-style legs of C5, C4, C2, C1), per-workload bench lines with their CPU
#   baselines, rocprofv3 kernel stats, PMC traffic and instruction mix of one C3 step, per-phase stamps of the strip
#   kernel (diagnostic build), the loop-repeat experiment, microbenchmarks.
# usage (from the repo root on the GPU box):  bash profiles/collect_r03.sh gpurun_out/r03
# PMC passes run on their own (never with --kernel-trace --stats in one rocprofv3 command), one counter set per pass.
set -u
OUT=${1:-gpurun_out/r03}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py"
timeout -k 10 400 $B --steps 20 --warmup 3 > "$OUT/default_bench.json" 2> "$OUT/default_bench.err"
for w in c1 c2 c3 c4 c5; do
	timeout -k 10 300 $B --workload $w --steps 5 --warmup 2 --no-configs > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"
done
timeout -k 10 300 $B --workload c3 --arith fma --steps 5 --warmup 2 --cpu-rows 0 --no-configs > "$OUT/c3_fma_bench.json" 2>/dev/null
timeout -k 10 300 $B --workload c3 --arith f32 --steps 5 --warmup 2 --cpu-rows 0 --no-configs > "$OUT/c3_f32_bench.json" 2>/dev/null
SRH_BENCH_STRIP=0 timeout -k 10 300 $B --workload c3 --steps 5 --warmup 2 --cpu-rows 0 --no-configs > "$OUT/c3_per_tile_kernel_bench.json" 2>/dev/null
echo "bench lines done"
# per-kernel time
# (C4: one view in flight, SRH_MVS_ASYNC=0 -- with the default two, a kernel's traced duration includes its neighbour's share
# of the GPU; bench.py takes its kernels_ms from such a pass as well)
export SRH_MVS_ASYNC=0
for w in c3 c4 c5; do
	rocprofv3 --kernel-trace --stats -d "$OUT/stats_$w" --output-format csv -- $B --workload $w --steps 3 --warmup 1 --cpu-rows 0 --no-configs > "$OUT/stats_$w.log" 2>&1
	cp "$(find "$OUT/stats_$w" -name '*kernel_stats.csv' | head -1)" "$OUT/${w}_kernel_stats.csv" 2>/dev/null
	rm -rf "$OUT/stats_$w"
done
echo "kernel stats done"
pmc() { # tag workload counters...
	local tag=$1 w=$2; shift 2
	rocprofv3 --pmc "$@" -d "$OUT/pmc_$tag" --output-format csv -- $B --workload $w --steps 1 --warmup 0 --cpu-rows 0 --no-configs > "$OUT/pmc_$tag.log" 2>&1
	cp "$(find "$OUT/pmc_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.csv" 2>/dev/null
	rm -rf "$OUT/pmc_$tag"
	echo "pass $tag done"
}
pmc c3_fetch c3 FETCH_SIZE
pmc c3_write c3 WRITE_SIZE
pmc c3_mix1 c3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA
pmc c3_mix2 c3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
for w in c4 c5; do
	pmc ${w}_fetch $w FETCH_SIZE
	pmc ${w}_write $w WRITE_SIZE
	pmc ${w}_mix1 $w SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM
	pmc ${w}_mix2 $w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
	pmc ${w}_mix3 $w SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT
	pmc ${w}_tcp $w TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
	pmc ${w}_tcc $w TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
unset SRH_MVS_ASYNC
for w in c3 c4 c5; do python3 profiles/pmc_table.py $(ls "$OUT"/pmc_${w}_mix*.csv "$OUT"/pmc_${w}_tc*.csv 2>/dev/null) > "$OUT/${w}_instruction_mix.txt" 2>/dev/null; done
# per-phase stamps of the strip kernel (diagnostic build: never the shipped library), both forms
for strip in 8 4; do
	SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so SRH_BENCH_STRIP=$strip timeout -k 10 200 $B --workload c3 --steps 2 --warmup 1 --cpu-rows 0 --no-configs > /dev/null 2> "$OUT/phases_strip$strip.err"
	grep "srh dbg" "$OUT/phases_strip$strip.err" | grep -v rows | tail -9 > "$OUT/c3_strip${strip}_phases.txt"
done
# phases of the geodesic kernel (C3) and of the staged MVS cost kernel (C4), diagnostic build
SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so timeout -k 10 200 $B --workload c3 --steps 1 --warmup 1 --cpu-rows 0 --no-configs 2>&1 >/dev/null | grep "geodesic kernel" | tail -1 > "$OUT/c3_geodesic_phases.txt"
SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so SRH_MVS_ASYNC=0 timeout -k 10 200 $B --workload c4 --steps 1 --warmup 1 --cpu-rows 0 --no-configs 2>&1 >/dev/null | grep "staged MVS" | tail -8 > "$OUT/c4_staged_phases.txt"
# what the block loops cost: every tile's loops repeated 1, 2, 3 times (timing experiment build), per-tile and strip kernels
{
	for strip in 0 8; do for rep in 1 2 3; do
		SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_exp.so SRH_BENCH_STRIP=$strip SRH_BENCH_EXP_REPEAT=$rep timeout -k 10 200 $B --workload c3 --steps 3 --warmup 1 --cpu-rows 0 --no-configs > "$OUT/rep.json" 2>/dev/null
		python3 -c "
import json
d=json.load(open('$OUT/rep.json'))
k=[(n,v) for n,v in d['kernels_ms'].items() if 'cost_kernel' in n][0]
print('strip option $strip  loops x $rep :  %s %.3f ms per launch' % (k[0], k[1][0]/k[1][1]))"
	done; done
} > "$OUT/c3_repeat_experiment.txt" 2>&1
rm -f "$OUT/rep.json"
./profiles/microbench/fp64_sustained > "$OUT/fp64_sustained_mi355x.txt" 2>&1
./profiles/microbench/fp64_dep_distance > "$OUT/fp64_dep_distance_mi355x.txt" 2>&1
./profiles/microbench/fp64_chain_latency > "$OUT/fp64_chain_latency_mi355x.txt" 2>&1
./profiles/microbench/lds_dma_alignment > "$OUT/lds_dma_alignment_mi355x.txt" 2>&1
rm -f "$OUT"/pmc_*.log "$OUT"/stats_*.log
ls "$OUT"
