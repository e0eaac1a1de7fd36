#!/bin/bash
# Runs on the GPU box: SQ counters of the tracker launches of the timed steps, for one setting of the environment.
#   tools/pmc_lk.sh <tag> [VAR=value ...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/$TAG; mkdir -p "$O"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-kernel-timing > $O/pmc.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_lk_fast" not in n: continue
    key = "%s grid=%s" % (n.split("(")[0][-30:], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
