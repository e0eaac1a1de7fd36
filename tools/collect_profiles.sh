#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + the PMC passes the roofline numbers come from, for `python3 bench.py`
# itself.  Usage: tools/collect_profiles.sh <tag> [bench args...]    (e.g. r02_c2, or r02_ref --config ref)
# Counters are collected in passes of their own, each with --kernel-trace only (never together with other trace domains);
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).  The summaries end up in gpurun_out/<tag>/: copy what is to
# be judged into profiles/.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
ARGS="--steps 32 --warmup 4 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $ARGS > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_a -- python3 bench.py $ARGS > $O/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq_b -- python3 bench.py $ARGS > $O/pmc_sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $ARGS > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq_a $O/pmc_sq_b $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
grep -h '"metric"' $O/trace.log | tail -1 > $O/bench_under_trace.json
# the timeline of the TIMED steps: a trace of the run without the per-kernel "alone" launches behind the timed region
rocprofv3 --kernel-trace --output-format csv -d $O/trace_tl -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-kernel-timing $* > $O/trace_tl.log 2>&1
python3 tools/timeline_steps.py $O/trace_tl > $O/timeline.txt 2>&1
tail -3 $O/timeline.txt; head -6 $O/kernel_stats.csv | cut -c1-160
