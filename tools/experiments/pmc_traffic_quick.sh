#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the pipeline's kernels at 0.5 h (180,000 frames) for the library in VBX_LIB_PATH
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_tq_${1:-x}; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 bench.py --hours 0.5 --steps 1 --warmup 0 --no-cpu > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 bench.py --hours 0.5 --steps 1 --warmup 0 --no-cpu > $O/w.log 2>&1
python3 - "$O" <<'PY'
import csv,glob,collections,sys
O=sys.argv[1]
for name in ('f','w'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'{O}/{name}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if any(t in k[0] for t in ('analyze','burg','formant','tracker')):
            m=sum(v)/len(v); mult = 2048.0 if k[1]=='FETCH_SIZE' else 1024.0
            print(name,k,len(v),'%.0f B/frame' % (m*mult/180000))
PY
