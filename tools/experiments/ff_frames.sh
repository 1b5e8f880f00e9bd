#!/bin/bash
# config 4 at other batch sizes: does the slicing rule hold away from 1 M frames?
for f in "$@"; do
  python3 bench.py --workload config4 --frames $f --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,1), round(d['ms_per_step'],3), d['kernels_ms'])" $f
done
