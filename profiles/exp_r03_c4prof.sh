cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c4pmc gpurun_out/c4pmc2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/c4pmc -o c4 --output-format csv -- python3 bench.py --workload c4 --steps 1 --warmup 0 --cpu-rows 0 --no-configs > gpurun_out/c4pmc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM -d gpurun_out/c4pmc2 -o c4 --output-format csv -- python3 bench.py --workload c4 --steps 1 --warmup 0 --cpu-rows 0 --no-configs > gpurun_out/c4pmc2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ('c4pmc','c4pmc2'):
    fs=glob.glob('gpurun_out/%s/**/*counter_collection.csv'%d,recursive=True)
    if not fs: print(d,'no file'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        k='staged' if 'mvs_staged' in k else 'list' if 'mvs_list_cost' in k else 'walk' if 'mvs_walk' in k else None
        if k: acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items(): print(k, {a:int(b) for a,b in v.items()})
PY
