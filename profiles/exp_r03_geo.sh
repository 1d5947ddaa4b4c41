timeout -k 10 200 python3 bench.py --workload c3 --steps 4 --warmup 2 --cpu-rows 0 --no-configs > gpurun_out/geo.json 2>/dev/null
python3 -c "
import json
d=json.load(open('gpurun_out/geo.json'))
print(d['ms_per_step'], {k:round(v[0]/v[1],3) for k,v in d['kernels_ms'].items()})"
SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so timeout -k 10 200 python3 bench.py --workload c3 --steps 1 --warmup 1 --cpu-rows 0 --no-configs 2>&1 | grep "geodesic kernel" | tail -1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_strip.py -x -q 2>&1 | tail -2
