#!/bin/bash
# the fused frame loop at other frame shapes: frames/s and the per-kernel times (which stage binds where)
for s in "$@"; do
  IFS=: read n h <<< "$s"
  python3 bench.py --frame-len $n --hop $h --hours 1 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']/1e6,2), 'M/s', d['kernels_ms'])" $s
done
