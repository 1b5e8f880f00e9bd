#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof_mfcc; mkdir -p gpurun_out/prof_mfcc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_mfcc/pmc1 -- python3 tools/experiments/mfcc_bench.py > gpurun_out/prof_mfcc/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_mfcc/pmc2 -- python3 tools/experiments/mfcc_bench.py > gpurun_out/prof_mfcc/pmc2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/prof_mfcc/{name}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            if 'mfcc' in k[0]: print(name,k,len(v),sum(v)/len(v))
PY
tail -3 gpurun_out/prof_mfcc/pmc2.log | cut -c1-300
