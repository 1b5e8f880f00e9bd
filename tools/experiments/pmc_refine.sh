#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof_ref; mkdir -p gpurun_out/prof_ref
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_ref/pmc1 -- python3 tools/experiments/pitch_full.py > gpurun_out/prof_ref/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_ref/pmc2 -- python3 tools/experiments/pitch_full.py > gpurun_out/prof_ref/pmc2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/prof_ref/{name}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:26], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            if 'pitch' in k[0]: print(name,k,len(v),sum(v)/len(v))
PY
