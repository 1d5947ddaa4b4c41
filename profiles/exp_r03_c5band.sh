for mb in 8192 32768; do
SRH_WBUF_MB=$mb timeout -k 10 300 python3 bench.py --workload c5 --steps 3 --warmup 1 --cpu-rows 0 --no-configs > gpurun_out/c5_$mb.json 2>/dev/null
python3 -c "
import json
d=json.load(open('gpurun_out/c5_$mb.json'))
print('$mb', d['ms_per_step'], {k:(round(v[0]/v[1],3), v[1]) for k,v in d['kernels_ms'].items() if v[0]>1})"
done
