#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof_c2b; mkdir -p gpurun_out/prof_c2b
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/prof_c2b/pmc1 -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof_c2b/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof_c2b/pmc2 -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof_c2b/pmc2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/prof_c2b/{name}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:30], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            if 'autocorr' in k[0]: print(name,k,len(v),sum(v)/len(v))
PY
tail -2 gpurun_out/prof_c2b/pmc1.log | cut -c1-200
