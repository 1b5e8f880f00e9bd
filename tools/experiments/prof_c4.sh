#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof4; mkdir -p gpurun_out/prof4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4/t -- python3 bench.py --workload config4 --steps 3 --warmup 1 --no-cpu > gpurun_out/prof4/log 2>&1
for f in $(find gpurun_out/prof4/t -name "*kernel_stats.csv"); do cut -c1-60,200- $f | head -8; cut -d, -f1-4 $f | cut -c1-200 | head -8; done
