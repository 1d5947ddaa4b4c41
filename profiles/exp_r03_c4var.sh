for v in default $VARIANTS; do
L=$PWD/stereoreconstruction_amd/libstereo_recon_hip.so; [ $v != default ] && L=$PWD/stereoreconstruction_amd/libvar_$v.so
SRH_LIBRARY=$L timeout -k 10 200 python3 bench.py --workload c4 --steps 3 --warmup 1 --cpu-rows 8 --no-configs > gpurun_out/c4_$v.json 2>gpurun_out/c4_$v.err; python3 -c "
import json
d=json.load(open('gpurun_out/c4_$v.json'))
print('$v', d['ms_per_step'], {k:round(v[0]/v[1],3) for k,v in d['kernels_ms'].items() if 'mvs' in k}, d['cpu_baseline']['parity_band'] if d.get('cpu_baseline') else None)"; done
