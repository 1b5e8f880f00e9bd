#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command, then PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --hours 0.5 --steps 3 --warmup 1 --no-cpu > gpurun_out/prof/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace_c2 -- python3 bench.py --workload config2 --steps 5 --warmup 2 --no-cpu > gpurun_out/prof/bench_trace_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof/pmc_write.log 2>&1
find gpurun_out/prof -name "*.csv" | head -30
for f in $(find gpurun_out/prof/trace gpurun_out/prof/trace_c2 -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
