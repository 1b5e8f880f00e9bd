#!/bin/bash
# Evidence run of a round (rounds 3 to 5) (one MI355X): rocprofv3 kernel trace + stats of the bench commands, then PMC passes in SEPARATE
# runs (HBM traffic: FETCH_SIZE / WRITE_SIZE, one counter per pass; one SQ pass per workload).  Everything lands in
# gpurun_out/prof_<tag>/; tools/prof_summary.py + tools/prof_commit.py turn it into the files committed under profiles/.
# usage: tools/prof_round.sh [tag] [quick]       (run on the GPU box from the repo root; "quick" skips the other frame shapes)
TAG=${1:-r04c}
QUICK=${2:-}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
B="--no-cpu --no-sub --no-live-traffic"
run() { local name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; grep '^{"metric' $O/$name.log > $O/$name.bench_line.json; }
# kernel traces
run trace_pipeline --kernel-trace --stats --output-format csv -d $O/trace_pipeline -- python3 bench.py --hours 2 --steps 3 --warmup 1 $B
run trace_config2 --kernel-trace --stats --output-format csv -d $O/trace_config2 -- python3 bench.py --workload config2 --steps 5 --warmup 2 $B
run trace_config3 --kernel-trace --stats --output-format csv -d $O/trace_config3 -- python3 bench.py --workload config3 --hours 2 --steps 3 --warmup 1 $B
run trace_config4 --kernel-trace --stats --output-format csv -d $O/trace_config4 -- python3 bench.py --workload config4 --steps 5 --warmup 2 $B
run trace_frontend --kernel-trace --stats --output-format csv -d $O/trace_frontend -- python3 bench.py --workload frontend --hours 1 --steps 3 --warmup 1 $B
if [ -z "$QUICK" ]; then
  # the reference's own frame shapes (examples/pitch_detection.rs:23: 2048 / 1024; tests/lib.rs:56-57: 1024 / 512)
  run trace_pipeline_2048 --kernel-trace --stats --output-format csv -d $O/trace_pipeline_2048 -- python3 bench.py --frame-len 2048 --hop 1024 --hours 2 --steps 3 --warmup 1 $B
  run trace_config3_2048 --kernel-trace --stats --output-format csv -d $O/trace_config3_2048 -- python3 bench.py --workload config3 --frame-len 2048 --hop 1024 --hours 2 --steps 3 --warmup 1 $B
  run trace_config3_1024 --kernel-trace --stats --output-format csv -d $O/trace_config3_1024 -- python3 bench.py --workload config3 --frame-len 1024 --hop 512 --hours 2 --steps 3 --warmup 1 $B
  # the reference's bench frame (benches/periodic.rs:22-25: 4096 samples; hop 2048)
  run trace_pipeline_4096 --kernel-trace --stats --output-format csv -d $O/trace_pipeline_4096 -- python3 bench.py --frame-len 4096 --hop 2048 --hours 2 --steps 3 --warmup 1 $B
  run trace_config3_4096 --kernel-trace --stats --output-format csv -d $O/trace_config3_4096 -- python3 bench.py --workload config3 --frame-len 4096 --hop 2048 --hours 2 --steps 3 --warmup 1 $B
  # 25 ms at 44.1 kHz: a length that does not divide its transform -- MFCC by interpolated bins inside the fused kernel (round 5)
  run trace_pipeline_1103 --kernel-trace --stats --output-format csv -d $O/trace_pipeline_1103 -- python3 bench.py --frame-len 1103 --hop 441 --hours 2 --steps 3 --warmup 1 $B
  # 62.5 ms frames: the 4096-point plan as two kernels (transforms + LPC + interpolated MFCC, then the refinement from the scratch rows)
  run trace_pipeline_3000 --kernel-trace --stats --output-format csv -d $O/trace_pipeline_3000 -- python3 bench.py --frame-len 3000 --hop 1200 --hours 2 --steps 3 --warmup 1 $B
  for w in pipeline_2048:pipeline:2048:1024 config3_2048:config3:2048:1024 config3_1024:config3:1024:512 pipeline_4096:pipeline:4096:2048 config3_4096:config3:4096:2048 pipeline_1103:pipeline:1103:441 pipeline_3000:pipeline:3000:1200; do
    IFS=: read name wl fl hop <<< "$w"
    run pmc_fetch_$name --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$name -- python3 bench.py --workload $wl --frame-len $fl --hop $hop --hours 0.5 --steps 1 --warmup 0 $B
    run pmc_write_$name --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$name -- python3 bench.py --workload $wl --frame-len $fl --hop $hop --hours 0.5 --steps 1 --warmup 0 $B
  done
fi
# HBM traffic, one counter per pass; then one SQ pass (8 SQ slots + GRBM) per workload
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for w in pipeline config3; do
  run pmc_fetch_$w --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --hours 0.5 --steps 1 --warmup 0 $B
  run pmc_write_$w --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --hours 0.5 --steps 1 --warmup 0 $B
  run pmc_sq_$w --pmc $SQ --output-format csv -d $O/pmc_sq_$w -- python3 bench.py --workload $w --hours 0.5 --steps 1 --warmup 0 $B
done
for w in config2 config4; do
  run pmc_fetch_$w --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 $B
  run pmc_write_$w --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 $B
  run pmc_sq_$w --pmc $SQ --output-format csv -d $O/pmc_sq_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 $B
done
run pmc_fetch_frontend --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_frontend -- python3 bench.py --workload frontend --hours 0.5 --steps 1 --warmup 0 $B
run pmc_write_frontend --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_frontend -- python3 bench.py --workload frontend --hours 0.5 --steps 1 --warmup 0 $B
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
tail -60 $O/summary.txt
# keep the merge-back small: the raw traces are large, the stats and counter CSVs are not
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
