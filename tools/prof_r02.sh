#!/bin/bash
# Round-2 evidence run (one MI355X): rocprofv3 kernel trace + stats of the bench commands, then PMC passes in SEPARATE
# runs (HBM traffic: FETCH_SIZE / WRITE_SIZE; SQ counters of the dominant kernel).  Everything lands in
# gpurun_out/prof_r02/; tools/prof_summary.py turns it into the files committed under profiles/.
# usage: tools/prof_r02.sh [tag]        (run on the GPU box from the repo root)
TAG=${1:-r02a}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
run() { local name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; grep '^{"metric' $O/$name.log > $O/$name.bench_line.json; }
# kernel traces
run trace_pipeline --kernel-trace --stats --output-format csv -d $O/trace_pipeline -- python3 bench.py --hours 2 --steps 3 --warmup 1 --no-cpu
run trace_config2 --kernel-trace --stats --output-format csv -d $O/trace_config2 -- python3 bench.py --workload config2 --steps 5 --warmup 2 --no-cpu
run trace_config3 --kernel-trace --stats --output-format csv -d $O/trace_config3 -- python3 bench.py --workload config3 --hours 2 --steps 3 --warmup 1 --no-cpu
run trace_config4 --kernel-trace --stats --output-format csv -d $O/trace_config4 -- python3 bench.py --workload config4 --steps 5 --warmup 2 --no-cpu
# the reference's own frame shapes (examples/pitch_detection.rs:23: 2048 / 1024; tests/lib.rs:56-57: 1024 / 512)
run trace_pipeline_2048 --kernel-trace --stats --output-format csv -d $O/trace_pipeline_2048 -- python3 bench.py --frame-len 2048 --hop 1024 --hours 2 --steps 3 --warmup 1 --no-cpu
run trace_config3_2048 --kernel-trace --stats --output-format csv -d $O/trace_config3_2048 -- python3 bench.py --workload config3 --frame-len 2048 --hop 1024 --hours 2 --steps 3 --warmup 1 --no-cpu
run trace_config3_4096 --kernel-trace --stats --output-format csv -d $O/trace_config3_4096 -- python3 bench.py --workload config3 --frame-len 4096 --hop 2048 --hours 2 --steps 3 --warmup 1 --no-cpu
run trace_config3_1024 --kernel-trace --stats --output-format csv -d $O/trace_config3_1024 -- python3 bench.py --workload config3 --frame-len 1024 --hop 512 --hours 2 --steps 3 --warmup 1 --no-cpu
for w in pipeline_2048:pipeline:2048:1024 config3_2048:config3:2048:1024 config3_1024:config3:1024:512; do
  IFS=: read name wl fl hop <<< "$w"
  run pmc_fetch_$name --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$name -- python3 bench.py --workload $wl --frame-len $fl --hop $hop --hours 0.5 --steps 1 --warmup 0 --no-cpu
  run pmc_write_$name --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$name -- python3 bench.py --workload $wl --frame-len $fl --hop $hop --hours 0.5 --steps 1 --warmup 0 --no-cpu
done
# HBM traffic, one counter per pass
for w in pipeline config3; do
  run pmc_fetch_$w --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --hours 0.5 --steps 1 --warmup 0 --no-cpu
  run pmc_write_$w --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --hours 0.5 --steps 1 --warmup 0 --no-cpu
done
for w in config2 config4; do
  run pmc_fetch_$w --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu
  run pmc_write_$w --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu
done
# SQ counters of the fused spectral kernel (pipeline, 0.5 h = 180,000 frames)
run pmc_sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq1 -- python3 bench.py --hours 0.5 --steps 1 --warmup 0 --no-cpu
run pmc_sq2 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq2 -- python3 bench.py --hours 0.5 --steps 1 --warmup 0 --no-cpu
run pmc_sq3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq3 -- python3 bench.py --hours 0.5 --steps 1 --warmup 0 --no-cpu
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
# keep the merge-back small: the raw traces are large, the stats and counter CSVs are not
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
