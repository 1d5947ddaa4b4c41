set -u
O=gpurun_out/r3q; mkdir -p $O
for strip in 0 8; do
SRH_BENCH_STRIP=$strip timeout -k 10 200 python3 bench.py --workload c3 --steps 5 --warmup 2 --cpu-rows 0 --no-configs > $O/c3_strip$strip.json 2> $O/c3_strip$strip.err
python3 -c "
import json
d=json.load(open('$O/c3_strip$strip.json'))
print('strip $strip', d['ms_per_step'], d['value']); print({k:(round(v[0]/v[1],3),v[1]) for k,v in d['kernels_ms'].items()})"
done
