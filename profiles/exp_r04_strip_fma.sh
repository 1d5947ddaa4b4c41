#!/bin/bash
# round 4: where the certified (fused) strip kernel's time goes -- phase stamps (diagnostic build) and the loop-repeat
# experiment (timing-experiment build), C3, both forms.  usage: bash profiles/exp_r04_strip_fma.sh gpurun_out/r4f
set -u
O=${1:-gpurun_out/r4f}; mkdir -p $O
B="python3 bench.py --workload c3 --cpu-rows 0 --no-configs"
for ar in certified exact; do
  SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so timeout -k 10 200 $B --steps 2 --warmup 1 --arith $ar > $O/phases_$ar.json 2> $O/phases_$ar.err
  grep "srh dbg" $O/phases_$ar.err | grep -v rows | tail -9 > $O/c3_strip8_phases_$ar.txt
done
{
for ar in certified exact; do for rep in 1 2 3; do
  SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_exp.so SRH_BENCH_EXP_REPEAT=$rep timeout -k 10 200 $B --steps 3 --warmup 1 --arith $ar > $O/rep.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/rep.json'))
k=[(n,v) for n,v in d['kernels_ms'].items() if 'cost_kernel' in n][0]
print('arith $ar  loops x $rep :  %s %.3f ms per launch' % (k[0], k[1][0]/k[1][1]))"
done; done
} > $O/c3_repeat_experiment.txt 2>&1
rm -f $O/rep.json
cat $O/c3_repeat_experiment.txt; cat $O/c3_strip8_phases_certified.txt
