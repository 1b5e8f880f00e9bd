#!/bin/bash
# kernel-level times of config 4 with the chunked tracker (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_trk; rm -rf $O; mkdir -p $O; cd $R
export VBX_TRACKER_CHUNKED=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --workload config4 --steps 10 --warmup 3 --no-cpu > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %4s avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
