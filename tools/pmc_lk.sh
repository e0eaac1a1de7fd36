#!/bin/bash
# Runs on the GPU box: the two SQ counter passes for the tracker kernel only (quick look while tuning).
# Usage: tools/pmc_lk.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
ARGS="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_a -- python3 bench.py $ARGS > $O/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq_b -- python3 bench.py $ARGS > $O/pmc_sq_b.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq_a $O/pmc_sq_b > $O/pmc_summary.txt 2>&1
grep -A17 "k_lk" $O/pmc_summary.txt
