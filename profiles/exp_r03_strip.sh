set -u
O=gpurun_out/r3exp2; mkdir -p $O
export SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_exp.so
for strip in 4 8; do for rep in 1 2; do
  SRH_BENCH_STRIP=$strip SRH_BENCH_EXP_REPEAT=$rep timeout -k 10 200 python3 bench.py --workload c3 --steps 3 --warmup 1 --cpu-rows 0 --no-configs > $O/strip${strip}_rep${rep}.json 2> $O/strip${strip}_rep${rep}.err
  python3 -c "import json,sys; d=json.load(open('$O/strip${strip}_rep${rep}.json')); print('strip',$strip,'rep',$rep,'ms/step',d['ms_per_step'],d['kernels_ms'])"
done; done
unset SRH_LIBRARY
timeout -k 10 200 python3 bench.py --workload c3 --steps 5 --warmup 2 --cpu-rows 4 --no-configs > $O/c3_default.json 2> $O/c3_default.err; cat $O/c3_default.json
