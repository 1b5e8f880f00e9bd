#!/bin/bash
# config 4 with each library given (VBX_LIB_PATH), optionally with extra environment: tools/experiments/ab_config4.sh lib1.so lib2.so ...
for lib in "$@"; do
  VBX_LIB_PATH=$PWD/$lib python3 bench.py --workload config4 --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,1), d['kernels_ms'])" $lib
done
