#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc / --kernel-trace CSV output: per kernel name, mean counter value
per dispatch.  usage: tools/pmc_summary.py DIR [DIR ...] (searches *_counter_collection.csv,
*_kernel_trace.csv)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void rmk::", "").replace("rmk::", "")
    return name[:70]


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            meta = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"),
                           r.get("Grid_Size"), r.get("Workgroup_Size"))
            for k in agg:
                if "rm_" not in k:
                    continue
                print("%s  [%s] vgpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % ((os.path.relpath(f), k) + meta[k]))
                for c in sorted(agg[k]):
                    v = agg[k][c]
                    print("    %-28s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
        for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k, v in dur.items():
                if "rm_" in k:
                    print("%s  [%s] dispatches=%d mean=%.1f us min=%.1f us max=%.1f us" %
                          (os.path.relpath(f), k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))


if __name__ == "__main__":
    main()
