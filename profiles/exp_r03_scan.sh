set -u
O=gpurun_out/r3scan; mkdir -p $O
export SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_exp.so
for mode in 0 1 2; do
SRH_BENCH_EXP_SCAN_MODE=$mode timeout -k 10 200 python3 bench.py --workload c3 --steps 3 --warmup 1 --cpu-rows 0 --no-configs > $O/mode$mode.json 2> $O/mode$mode.err
python3 -c "
import json
d=json.load(open('$O/mode$mode.json'))
print('scan mode $mode', {k:(round(v[0]/v[1],3),v[1]) for k,v in d['kernels_ms'].items() if 'scan' in k})"
done
