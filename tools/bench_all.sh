#!/bin/bash
# Un-profiled bench lines of every workload (the files profiles/<tag>_bench_*.json): default = the 12.5 h pipeline shard with
# the CPU baseline legs.  usage (GPU box): tools/bench_all.sh <tag>
TAG=${1:-r02f}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bench_$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; python3 bench.py "$@" 2>/dev/null | grep '^{"metric' > $O/${TAG}_bench_$name.json; python3 -c "
import json,sys; d=json.loads(open('$O/${TAG}_bench_$name.json').readline()); print('$name', round(d['value']/1e6,2), 'M/s', d.get('kernels_ms'))"; }
run default
run config2 --workload config2 --steps 20 --warmup 5
run config3 --workload config3
run config3_kmax64 --workload config3 --kmax 64 --hours 1 --no-cpu
run config3_kmax302 --workload config3 --kmax 302 --hours 1 --no-cpu
run config4 --workload config4 --steps 10 --warmup 3
run frontend --workload frontend --no-cpu
run pipeline_2048 --frame-len 2048 --hop 1024 --no-cpu
run config3_2048 --workload config3 --frame-len 2048 --hop 1024 --no-cpu
run config3_1024 --workload config3 --frame-len 1024 --hop 512 --no-cpu
run config3_4096 --workload config3 --frame-len 4096 --hop 2048 --no-cpu
