#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof_pitch; mkdir -p gpurun_out/prof_pitch
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pitch/trace -- python3 tools/experiments/pitch_full.py > gpurun_out/prof_pitch/log.txt 2>&1
tail -2 gpurun_out/prof_pitch/log.txt | cut -c1-200
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_pitch/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)): print('  ',r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3),'ms', r['Percentage'])
PY
