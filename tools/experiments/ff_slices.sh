#!/bin/bash
# config 4 (find_formants, 1 M x 512) against the number of time slices (VBX_FF_SLICES, default 4)
for k in "$@"; do
  VBX_FF_SLICES=$k python3 bench.py --workload config4 --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,1), d['kernels_ms'])" $k
done
