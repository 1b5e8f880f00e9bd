#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof6; mkdir -p gpurun_out/prof6
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d gpurun_out/prof6/pmc -- python3 bench.py --workload config4 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof6/pmc.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/prof6/pmc/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:30], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if 'burg' in k[0] or 'formant' in k[0] or 'tracker_spec' in k[0]: print(k, sum(v)/len(v))
PY
