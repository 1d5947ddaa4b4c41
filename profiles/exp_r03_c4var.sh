for a in 0 1; do
SRH_MVS_ASYNC=$a timeout -k 10 200 python3 bench.py --workload c4 --steps 5 --warmup 2 --cpu-rows 8 --no-configs > gpurun_out/c4_async$a.json 2>gpurun_out/c4_async$a.err; python3 -c "
import json
d=json.load(open('gpurun_out/c4_async$a.json'))
print('async=$a', d['ms_per_step'], {k:round(v[0]/v[1],3) for k,v in d['kernels_ms'].items() if 'mvs' in k}, d['cpu_baseline']['parity_band'] if d.get('cpu_baseline') else None)"; done
