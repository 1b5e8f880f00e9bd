#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench commands, then PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --hours 2 --steps 3 --warmup 1 --no-cpu > gpurun_out/prof/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace_c2 -- python3 bench.py --workload config2 --steps 5 --warmup 2 --no-cpu > gpurun_out/prof/bench_trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace_c4 -- python3 bench.py --workload config4 --steps 5 --warmup 2 --no-cpu > gpurun_out/prof/bench_trace_c4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch_p -- python3 bench.py --workload config3 --hours 0.5 --steps 1 --warmup 0 --no-cpu > gpurun_out/prof/pmc_fetch_p.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write_p -- python3 bench.py --workload config3 --hours 0.5 --steps 1 --warmup 0 --no-cpu > gpurun_out/prof/pmc_write_p.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for name in ('pmc_fetch','pmc_write','pmc_fetch_p','pmc_write_p'):
    for f in glob.glob(f'gpurun_out/prof/{name}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()): print(name,k,len(v),sum(v)/len(v))
for t in ('trace','trace_c2','trace_c4'):
    for f in glob.glob(f'gpurun_out/prof/{t}/*/*kernel_stats.csv'):
        print('==',t)
        for r in csv.DictReader(open(f)): print('  ',r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3),'ms', r['Percentage'])
PY
