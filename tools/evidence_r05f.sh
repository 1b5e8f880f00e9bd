#!/bin/bash
# Round 5, last evidence run (one MI355X): the driver's GPU suite, the profiled bench commands + counters (tools/prof_round.sh), the stall
# counters of the headline kernel, the default bench line, every-frame soaks at 1200 / 480, at 1103 / 441 (MFCC by interpolated bins) and at 3000 / 1200 (the 4096-point plan as two kernels),
# the interpolated MFCC against the chirp-z form at fifteen shapes.  Everything lands in gpurun_out/; tools/prof_commit.py r05f copies.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r05f_gputest.txt
bash tools/prof_round.sh r05f > gpurun_out/prof_r05f.log 2>&1
bash tools/prof_stalls.sh r05f > gpurun_out/stalls_r05f.log 2>&1
cd $R
python3 bench.py > gpurun_out/r05f_bench_default.json 2> gpurun_out/r05f_bench_default.err
python3 tools/soak_parity.py 100000 gpurun_out/r05f_soak_every_frame_1200.json 1200 480 1 0 > gpurun_out/r05f_soak_1200.log 2>&1
python3 tools/soak_parity.py 60000 gpurun_out/r05f_soak_every_frame_1103.json 1103 441 1 0 > gpurun_out/r05f_soak_1103.log 2>&1
python3 tools/soak_parity.py 20000 gpurun_out/r05f_soak_every_frame_3000.json 3000 1200 1 0 > gpurun_out/r05f_soak_3000.log 2>&1
python3 tools/experiments/mfcc_interp_check.py --hours 2 > gpurun_out/r05f_mfcc_interp_check.txt 2>&1
python3 tools/experiments/interp_parts.py > gpurun_out/r05f_interp_parts.txt 2>&1
python3 tools/experiments/split_check.py --hours 2 > gpurun_out/r05f_split_check.txt 2>&1
cat gpurun_out/r05f_gputest.txt; tail -3 gpurun_out/r05f_soak_1200.log; tail -3 gpurun_out/r05f_soak_1103.log; tail -3 gpurun_out/r05f_soak_3000.log
