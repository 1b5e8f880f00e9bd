#!/bin/bash
# rocprofv3 kernel stats of the one-pass Burg at two shapes -> gpurun_out/burg_fast/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/burg_fast
for cfg in "512 512 1000000" "1200 480 2000000"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o r -- python3 $R/tools/experiments/dbg_burg_fast_prof.py $cfg > $R/gpurun_out/burg_fast/log_$tag.txt 2>&1
  find /tmp/p_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/burg_fast/stats_$tag.csv \;
done
