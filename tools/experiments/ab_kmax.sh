#!/bin/bash
# pitch frames/s against kmax for several library builds (VBX_LIB_PATH): usage ab_kmax.sh "k1 k2 ..." lib1.so lib2.so ...
KS=$1; shift
for lib in "$@"; do
  for k in $KS; do
    VBX_LIB_PATH=$PWD/$lib python3 bench.py --workload config3 --kmax $k --hours 1 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[2], 'kmax', sys.argv[1], round(d['value']/1e6,2), 'M frames/s', 'evals/frame', round(r.get('sinc_evals_per_frame',0),1))" $k $(basename $lib)
  done
done
