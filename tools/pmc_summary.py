#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel dispatch, per kernel name.
usage: tools/pmc_summary.py <dir with pmc*/p_counter_collection.csv> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and filt not in k: continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
