timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mrf.py -x -q -k "mvs" 2>&1 | tail -5 &&
for m in 0 1; do SRH_MVS_STAGED=$m timeout -k 10 200 python3 bench.py --workload c4 --steps 3 --warmup 1 --cpu-rows 8 --no-configs > gpurun_out/c4_st$m.json 2>gpurun_out/c4_st$m.err; python3 -c "
import json
d=json.load(open('gpurun_out/c4_st$m.json'))
print('staged=$m', d['ms_per_step'], {k:round(v[0]/v[1],3) for k,v in d['kernels_ms'].items()}, d['cpu_baseline']['parity_band'] if d.get('cpu_baseline') else None)"; done
