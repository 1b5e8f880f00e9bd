#!/bin/bash
# Round 6 evidence run (one MI355X): the driver's GPU suite, the profiled bench commands + counter passes (tools/prof_round.sh), the stall
# counters of the headline kernel (tools/prof_stalls.sh), the default bench line + its detail record, every-row LPC check against the
# long-double arbiter, the fast paths' hand-over shares per shape.  Everything lands in gpurun_out/; tools/prof_commit.py <tag> "" r06_headline
# copies what the roofline recomputation needs into profiles/r06_headline/.     usage: tools/evidence_r06.sh [tag] [quick]
TAG=${1:-r06a}; QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/${TAG}_gputest.txt
bash tools/prof_round.sh $TAG $QUICK > gpurun_out/prof_$TAG.log 2>&1
bash tools/prof_stalls.sh $TAG > gpurun_out/stalls_$TAG.log 2>&1
cd $R
python3 bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python3 tools/experiments/lpc_exact_check.py 6000 > gpurun_out/${TAG}_lpc_exact_check.txt 2>&1
python3 tools/experiments/burg_direct_by_shape.py 0.5 > gpurun_out/${TAG}_burg_direct_by_shape.txt 2>&1
cat gpurun_out/${TAG}_gputest.txt; wc -c gpurun_out/${TAG}_bench_default.json; tail -30 gpurun_out/prof_$TAG/summary.txt
