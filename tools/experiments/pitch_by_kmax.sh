#!/bin/bash
# pitch frames/s against kmax (exact top-k pruning: what the caller asks for decides how much is refined)
for k in "$@"; do
  python3 bench.py --workload config3 --kmax $k --hours 1 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('kmax', sys.argv[1], round(d['value']/1e6,2), 'M frames/s', 'evals/frame', round(r.get('sinc_evals_per_frame',0),1))" $k
done
