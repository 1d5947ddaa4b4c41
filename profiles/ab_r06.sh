#!/bin/bash
# profiles/ab_r06.sh WORKLOAD "ENV=VAL ..." ... -- one bench line per variant inside ONE gpurun call (box-to-box variance is 1-5 %):
# ms per step with the two passes side by side, and the kernels' own durations from a second run with the passes in sequence.
# A variant is a string of environment assignments (SRH_LIBRARY=... for another build, SRH_BENCH_<OPTION>=... for a knob).
W=$1; shift
for v in "$@"; do
	a=$(env $v timeout -k 10 200 python3 bench.py --workload $W --steps 5 --warmup 2 --cpu-rows 0 --no-configs --no-first-call 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
	b=$(env $v SRH_BENCH_TV_OVERLAP=0 SRH_MVS_ASYNC=0 timeout -k 10 200 python3 bench.py --workload $W --steps 5 --warmup 2 --cpu-rows 0 --no-configs --no-first-call 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v[0]/v[1],3) for k,v in d['kernels_ms'].items() if v[0]/v[1] > 0.2})")
	echo "$W [$v] step $a ms | in sequence: $b"
done
