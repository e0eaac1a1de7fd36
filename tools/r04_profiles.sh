cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh r04_c2 > gpurun_out/r04_c2_collect.txt 2>&1; echo c2 done
bash tools/collect_profiles.sh r04_ref --config ref > gpurun_out/r04_ref_collect.txt 2>&1; echo ref done
bash tools/collect_profiles.sh r04_c5 --config c5 > gpurun_out/r04_c5_collect.txt 2>&1; echo c5 done
python tools/host_calls.py > gpurun_out/r04_host_calls.txt 2>&1; python tools/host_calls.py pcie 6 > gpurun_out/r04_host_calls_pcie.txt 2>&1; echo host done
ICELK_HOST_TAIL=1 python tools/host_calls.py pcie 6 > gpurun_out/r04_host_calls_pcie_hosttail.txt 2>&1
python tools/host_short.py > gpurun_out/r04_host_short_run.txt 2>&1
tail -4 gpurun_out/r04_c2_collect.txt gpurun_out/r04_ref_collect.txt gpurun_out/r04_c5_collect.txt; head -12 gpurun_out/r04_host_calls.txt gpurun_out/r04_host_calls_pcie.txt gpurun_out/r04_host_calls_pcie_hosttail.txt
