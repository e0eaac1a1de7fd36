cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pyramid" > gpurun_out/r4_pyr_t.log 2>&1; tail -3 gpurun_out/r4_pyr_t.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pyr_wide -- python3 tools/pyr_alone.py > /dev/null 2>&1
PYR_AHEAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pyr_1w -- python3 tools/pyr_alone.py > /dev/null 2>&1
for d in pyr_wide pyr_1w; do f=$(ls gpurun_out/$d/*/*_kernel_stats.csv | head -1); echo $d; grep -i pyramid $f | cut -c1-150; done
