#!/bin/bash
# SQ instruction counters of the split path (spectral_y_kernel / refine_y_kernel), config 3, 0.5 h = 180,000 frames
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_split; rm -rf $O; mkdir -p $O; cd $R
export VBX_SPLIT_CHUNK=180000
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 bench.py --workload ${1:-config3} --hours 0.5 --steps 1 --warmup 0 --no-cpu > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -- python3 bench.py --workload ${1:-config3} --hours 0.5 --steps 1 --warmup 0 --no-cpu > $O/p2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ('p1','p2'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/pmc_split/{name}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if 'spectral' in k[0] or 'refine' in k[0] or 'analyze' in k[0]: print(name,k,len(v),sum(v)/len(v), 'per frame', sum(v)/len(v)/180000)
PY
