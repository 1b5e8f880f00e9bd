#!/bin/bash
# Where the issue slots of the headline kernel go: five SQ passes of the SAME bench command (rocprofv3 --pmc,
# counters only -- no trace domains), per-kernel averages by tools/prof_summary.py; tools/prof_stalls.py turns the
# analyze_kernel rows into profiles/<tag>_analyze_stalls.json.
# usage: tools/prof_stalls.sh [tag] [bench args...]      (run on the GPU box from the repo root)
TAG=${1:-r04a}; shift
ARGS=${@:---hours 0.5 --steps 1 --warmup 0}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stalls_$TAG
rm -rf $O; mkdir -p $O
cd $R
B="--no-cpu --no-sub --no-live-traffic"
run() { local name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; }
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"
P4="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"
P5="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  run pmc_p$i --pmc $P --output-format csv -d $O/pmc_p$i -- python3 bench.py $ARGS $B
  i=$((i+1))
done
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
python3 tools/prof_stalls.py $O $TAG > $O/stalls.txt 2>&1
cat $O/stalls.txt | tail -60
find $O -name '*agent_info.csv' -delete
