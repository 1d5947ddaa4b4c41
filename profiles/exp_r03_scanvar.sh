for v in 16_5 8_4 8_5 8_6; do
SRH_LIBRARY=$PWD/stereoreconstruction_amd/libvar_$v.so timeout -k 10 200 python3 bench.py --workload c3 --steps 4 --warmup 2 --cpu-rows 0 --no-configs > gpurun_out/var_$v.json 2>/dev/null
python3 -c "
import json
d=json.load(open('gpurun_out/var_$v.json'))
print('$v', d['ms_per_step'], {k:round(v[0]/v[1],3) for k,v in d['kernels_ms'].items() if 'scan' in k})"
done
