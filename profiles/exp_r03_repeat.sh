set -u
O=gpurun_out/r3exp1; mkdir -p $O
export SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_exp.so
for pad in 0 24000; do for rep in 1 2 3; do
  SRH_BENCH_EXP_REPEAT=$rep SRH_BENCH_EXP_LDS_PAD=$pad timeout -k 10 200 python3 bench.py --workload c3 --steps 3 --warmup 1 --cpu-rows 0 --no-configs > $O/pad${pad}_rep${rep}.json 2> $O/pad${pad}_rep${rep}.err
  python3 -c "import json,sys; d=json.load(open('$O/pad${pad}_rep${rep}.json')); print('pad',$pad,'rep',$rep,'ms/step',d['ms_per_step'],d['kernels_ms'])"
done; done
./profiles/microbench/fp64_sustained > $O/fp64_sustained.txt 2>&1; cat $O/fp64_sustained.txt
