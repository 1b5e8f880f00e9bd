#!/bin/bash
# kernel timeline of one pipeline step (rocprofv3 --kernel-trace): who runs beside whom, who is the tail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_pipe; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --hours ${1:-6} --steps 1 --warmup 1 --no-cpu --no-sub > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(R + "/gpurun_out/trace_pipe/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last step: everything after the last synth/memset gap -> simply the last 200 kernels
t_end = int(rows[-1]["End_Timestamp"])
out = []
for r in rows[-120:]:
    out.append("%-40s start %9.3f ms  dur %8.3f ms" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t_end) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
open(R + "/gpurun_out/trace_pipe/timeline.txt", "w").write("\n".join(out) + "\n")
PY
rm -rf $O/t
