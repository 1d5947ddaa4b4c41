set -u
O=gpurun_out/r3exp3; mkdir -p $O
export SRH_LIBRARY=$PWD/profiles/lib/libstereo_recon_hip_prof.so
for strip in 4 8; do
  SRH_BENCH_STRIP=$strip timeout -k 10 200 python3 bench.py --workload c3 --steps 2 --warmup 1 --cpu-rows 0 --no-configs > $O/strip${strip}.json 2> $O/strip${strip}.err
  grep "srh dbg" $O/strip${strip}.err | tail -2
  python3 -c "import json,sys; d=json.load(open('$O/strip${strip}.json')); print('strip',$strip,'ms/step',d['ms_per_step'],d['kernels_ms'])"
done
