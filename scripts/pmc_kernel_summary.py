#!/usr/bin/env python3
"""Per-kernel means of the counters scripts/pmc_kernel.sh collected: pmc_kernel_summary.py gpurun_out/pmc_<TAG> [name filter]"""
import csv, glob, sys, collections
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void cmps::", "")
        if flt in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:] if len(v) > 2 else v          # drop warm-up launches
        print(f"    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
