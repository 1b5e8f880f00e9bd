#!/bin/bash
# the default pipeline bench under environment variants: tools/experiments/ab_env_pipeline.sh "VAR=val" "VAR=val2" ...  ("" = none)
for e in "$@"; do
  env $e python3 bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(repr(sys.argv[1]), round(d['value']/1e6,2), {k: round(v,1) for k,v in d['kernels_ms'].items()})" "$e"
done
