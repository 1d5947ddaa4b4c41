set -u
O=gpurun_out/r3c5; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload c5 --steps 3 --warmup 1 --cpu-rows 2 --no-configs > $O/c5.json 2> $O/c5.err
python3 -c "
import json
d=json.load(open('$O/c5.json'))
print('c5', d['ms_per_step'], d['value'], d['cpu_baseline']['parity_band']); print({k:(round(v[0]/v[1],3),v[1]) for k,v in d['kernels_ms'].items()})"
