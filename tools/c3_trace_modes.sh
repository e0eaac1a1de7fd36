#!/bin/bash
# Runs on the GPU box: the C3 configuration under a kernel + memory-copy trace, N times; per run the PCIe-inclusive rate and
# what the device did over the last periods of the PCIe-fed loop (kernels and copies on one time axis).
# Usage: tools/c3_trace_modes.sh <n> <tag>
N=${1:-4}; TAG=${2:-c3modes}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq $N); do
  O=gpurun_out/$TAG/run$i; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 bench.py --config c3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo "run $i failed"; tail -3 $O/bench.err; continue; }
  python3 - "$O" <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/bench.json") if l.startswith("{")][-1])
print("== %s: resident %.0f  pcie-inclusive %.0f pairs/s" % (o, d["value"], d["pcie_inclusive"]["value"]))
kf = glob.glob(o + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
mf = glob.glob(o + "/trace/**/*_memory_copy_trace.csv", recursive=True)
def short(n):
    return n.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:30]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "q" + r.get("Queue_Id", "")[-2:]) for r in csv.DictReader(open(kf))]
for r in csv.DictReader(open(mf[0])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 100000:
        rows.append((s, e, "COPY", "s" + r.get("Stream_Id", "")))
rows.sort()
lk = [i for i, r in enumerate(rows) if r[2].startswith("k_lk")]
# the PCIe-fed loop is the last run of tracker launches with copies between them
cps = [r for r in rows if r[2] == "COPY"]
t_first_loop_copy = cps[-60][0] if len(cps) >= 60 else cps[0][0]
lkl = [i for i in lk if rows[i][0] > t_first_loop_copy]
a, b = lkl[-8], lkl[-6]
t0 = rows[a][0]
for s, e, n, q in rows[a:b + 1]:
    print("%9.1f %8.1f  %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
span = rows[lkl[-3]][0] - rows[lkl[-23]][0]
print("steady: %.1f us per tracker launch; tracker kernels %.1f us mean" % (span / 20e3, sum(rows[i][1] - rows[i][0] for i in lkl[-23:-3]) / 20e3))
cc = [r for r in cps if rows[lkl[-23]][0] <= r[0] < rows[lkl[-3]][0]]
gaps = [(cc[k + 1][0] - cc[k][1]) / 1e3 for k in range(len(cc) - 1)]
print("copies: %d, mean %.1f us; idle between consecutive copies (us): %s" % (len(cc), sum(e - s for s, e, _, _ in cc) / len(cc) / 1e3, " ".join("%.0f" % g for g in gaps)))
PY
done
