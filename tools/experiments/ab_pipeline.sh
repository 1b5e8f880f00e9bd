#!/bin/bash
# the default pipeline bench with each library given: tools/experiments/ab_pipeline.sh lib1.so lib2.so ...
for lib in "$@"; do
  VBX_LIB_PATH=$PWD/$lib python3 bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,2), {k: round(v,1) for k,v in d['kernels_ms'].items()})" $lib
done
