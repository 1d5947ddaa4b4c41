#!/bin/bash
# profiles/exp_r06_phases_busy.sh OUT -- phase clocks of the staged MultiViewStereo cost kernel and of the geodesic windows
# kernel (diagnostic build, make prof), and the device's busy time over one timed step of C2 / C3 / C4 (kernel trace +
# profiles/gpu_busy.py).  usage (repo root, GPU box): bash profiles/exp_r06_phases_busy.sh gpurun_out/r06
OUT=${1:-gpurun_out/r06}
mkdir -p "$OUT"
export TMPDIR=/tmp
PROF=$PWD/profiles/lib/libstereo_recon_hip_prof.so
for a in certified exact; do
	SRH_LIBRARY=$PROF timeout -k 10 180 python3 bench.py --workload c4 --steps 1 --warmup 0 --cpu-rows 0 --no-configs --no-first-call --arith $a 2>&1 >/dev/null < /dev/null \
		| grep "staged MVS cost" | sort | uniq -c | sort -rn | head -4 | sed "s/^/arith $a: /"
done > "$OUT/c4_staged_phases.txt"
# (SRH_BENCH_GEODMA=0: the register-staged windows kernel, the one with the stamps; the default since round 5 is geodesic_dma_kernel)
SRH_BENCH_GEODMA=0 SRH_LIBRARY=$PROF timeout -k 10 180 python3 bench.py --workload c3 --steps 1 --warmup 0 --cpu-rows 0 --no-configs --no-exact-check --no-first-call 2>&1 >/dev/null < /dev/null \
	| grep "geodesic kernel" | tail -1 > "$OUT/c3_geodesic_phases.txt"
: > "$OUT/gpu_busy.txt"
for spec in "c2 twoview_cross_check 8 10" "c3 twoview_cross_check 8 10" "c4 mvs_cross_check 16 24"; do
	set -- $spec
	timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/kt_$1" --output-format csv -- python3 bench.py --workload $1 --steps 4 --warmup 2 --cpu-rows 0 --no-configs --no-exact-check --no-first-call > "$OUT/kt_$1.log" 2>&1 < /dev/null
	f=$(find "$OUT/kt_$1" -name "*kernel_trace.csv" | head -1)
	echo "== $1: one timed step (dispatches after the $3-th up to the $4-th $2 kernel)" >> "$OUT/gpu_busy.txt"
	[ -n "$f" ] && python3 profiles/gpu_busy.py "$f" $2 $3 $4 >> "$OUT/gpu_busy.txt" < /dev/null
	rm -rf "$OUT/kt_$1" "$OUT/kt_$1.log"
done
cat "$OUT/c4_staged_phases.txt" "$OUT/c3_geodesic_phases.txt" "$OUT/gpu_busy.txt"
