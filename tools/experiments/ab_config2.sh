#!/bin/bash
# config 2 with each library given: tools/experiments/ab_config2.sh lib1.so lib2.so ...
for lib in "$@"; do
  VBX_LIB_PATH=$PWD/$lib python3 bench.py --workload config2 --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,1), d['kernels_ms'], round(d['roofline']['frac'],3))" $lib
done
