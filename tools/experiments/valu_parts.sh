#!/bin/bash
# per-dispatch instruction counts and wait counters of tools/experiments/valu_parts.py (rocprofv3 --pmc, counters only; two passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/valu_parts; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/c1 -- python3 tools/experiments/valu_parts.py ${1:-0.5} > $O/log1.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c2 -- python3 tools/experiments/valu_parts.py ${1:-0.5} > $O/log2.txt 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for fn in glob.glob(sys.argv[1] + "/c*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in sorted(rows.items()):
    if v.get("SQ_WAVES", 0) < 1000: continue
    w = v["SQ_WAVES"]; wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print("%4d %-60s valu/wave %6.0f salu %5.0f lds %4.0f busy %.3f | wave cycles: issuing %.2f wait_inst %.2f wait_any %.2f | lds: active %.3f wait_inst_lds %.3f conflict/idx %.2f" % (
        d, k, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w,
        v["SQ_ACTIVE_INST_VALU"] / (v["GRBM_GUI_ACTIVE"] / 32.0 * 1024.0),
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc,
        v.get("SQ_ACTIVE_INST_LDS", 0) / wc, v.get("SQ_WAIT_INST_LDS", 0) / wc,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
PY
rm -rf $O/c1 $O/c2
