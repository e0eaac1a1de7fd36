#!/bin/bash
# Runs on the GPU box: tools/c3_stall.py under a kernel + memory-copy trace; then every hole of more than 3 ms in the device's
# activity (no kernel, no copy) with what ran before and after it.
N=${1:-24}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c3stall_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/c3_stall.py $N > $O/log.txt 2>&1
grep "pairs/s" $O/log.txt | cut -c1-150
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
kf = glob.glob(o + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
mf = glob.glob(o + "/trace/**/*_memory_copy_trace.csv", recursive=True)
def short(n):
    return n.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:30]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "q" + r.get("Queue_Id", "")[-2:]) for r in csv.DictReader(open(kf))]
for r in csv.DictReader(open(mf[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s" % r["Direction"][12:], "s" + r.get("Stream_Id", "")))
rows.sort()
# holes inside a run of tracker launches (between reps the device is idle anyway: handle creation)
busy_end = rows[0][1]
for i in range(1, len(rows)):
    s, e, n, q = rows[i]
    if s - busy_end > 3e6:
        before = [r for r in rows[max(0, i - 14):i]]
        lk_near = any(r[2].startswith("k_lk") for r in rows[max(0, i - 40):i]) and any(r[2].startswith("k_lk") for r in rows[i:i + 40])
        if lk_near:
            print("---- hole of %.1f ms" % ((s - busy_end) / 1e6))
            t0 = busy_end
            for r in rows[max(0, i - 14):i + 10]:
                print("%10.1f %8.1f  %-4s %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
    busy_end = max(busy_end, e)
PY
