#!/usr/bin/env python3
"""Mean counter values per launch, per kernel, from rocprofv3 --pmc CSV output:  python tools/pmc_summary.py <dir-or-csv> [substr]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "ilx::"
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in acc.items():
    print(k)
    for c, vals in sorted(cs.items()):
        g = max(v[0] for v in vals)
        full = [v[1] for v in vals if v[0] == g]
        print(f"   {c:40s} {sum(full) / len(full):16.1f}   ({len(full)} launches at grid {g})")
