#!/bin/bash
# VALU instruction counts of the pitch kernel for several library builds (rocprofv3 --pmc, counters only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O; cd $R
for L in "$@"; do
  name=$(basename $L .so)
  VBX_LIB_PATH=$R/$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/$name -- python3 bench.py --workload config3 --hours 1 --steps 1 --warmup 0 --no-cpu --no-sub > $O/$name.log 2>&1
  python3 - "$O/$name" "$name" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "analyze" in k or "pitch" in k:
        print(sys.argv[2], k, {a: round(b) for a, b in v.items()})
PY
  rm -rf $O/$name
done
