#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof3; mkdir -p gpurun_out/prof3
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d gpurun_out/prof3/pmc -- python3 bench.py --workload config3 --hours 0.25 --steps 1 --warmup 0 --no-cpu > gpurun_out/prof3/pmc.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/prof3/pmc/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:30], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if 'pitch' in k[0]: print(k,sum(v)/len(v))
PY
