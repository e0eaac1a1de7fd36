#!/bin/bash
# Runs on the GPU box: kernel + memory-copy trace of the PCIe-fed loop (tools/host_calls.py pcie <depth>), then what the
# device does over three steady periods.   Usage: tools/trace_pcie.sh <tag> [depth]
set -u
TAG=$1; D=${2:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/host_calls.py pcie $D > $O/trace.log 2>&1
head -1 $O/trace.log
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
kf = glob.glob(o + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
mf = glob.glob(o + "/trace/**/*_memory_copy_trace.csv", recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:36]) for r in csv.DictReader(open(kf))]
if mf:
    for r in csv.DictReader(open(mf[0])):
        if int(r.get("Bytes", r.get("Size", "0")) or 0) > 1000000:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s" % r.get("Direction", "")))
rows.sort()
lk = [i for i, r in enumerate(rows) if r[2].startswith("k_lk")]
a, b = lk[-12], lk[-9]
t0 = rows[a][0]
for s, e, n in rows[a:b + 1]:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
s0, s1 = rows[lk[-40]][0], rows[lk[-8]][0]
lkt = sum(rows[i][1] - rows[i][0] for i in lk[-40:-8])
print("steady state: %.1f us per tracker launch, tracker kernels cover %.1f%%" % ((s1 - s0) / 32e3, 100.0 * lkt / (s1 - s0)))
cp = [r for r in rows if r[2].startswith("COPY") and r[0] >= s0 and r[0] < s1]
if cp:
    print("copies: %d, mean %.1f us, start to start %.1f us" % (len(cp), sum(e - s for s, e, _ in cp) / len(cp) / 1e3, (cp[-1][0] - cp[0][0]) / (len(cp) - 1) / 1e3))
PY
