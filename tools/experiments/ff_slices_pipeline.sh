#!/bin/bash
# the pipeline (12.5 h shard unless given) against the number of find_formants time slices
H=${H:-4}
for k in "$@"; do
  VBX_FF_SLICES=$k python3 bench.py --hours $H --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,2), d['kernels_ms'])" $k
done
