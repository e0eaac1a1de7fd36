#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch.  Tracker
launches are listed per grid size as well (a joint launch of two segment pairs has twice the workgroups of a single one)."""
import collections
import csv
import glob
import sys


def short(name):
    name = name.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "")
    return name.split("(")[0][:40]


def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                per[(r["Dispatch_Id"], r["Kernel_Name"], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))][r["Counter_Name"]] = float(r["Counter_Value"])
            for (_, k, g, wg), cs in per.items():
                names = [short(k)]
                if "k_lk" in k and g and wg:
                    names.append("%s [%d workgroups]" % (short(k), int(g) // max(int(wg), 1)))
                for c, v in cs.items():
                    for nm in names:
                        agg[nm][c].append(v)
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("    %-24s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
