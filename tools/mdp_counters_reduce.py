"""Mean per launch of the counters tools/mdp_counters.sh collected:  python tools/mdp_counters_reduce.py <tag> [<tag> ...]"""
import csv, glob, sys
from collections import defaultdict
for tag in sys.argv[1:]:
    acc, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(f"gpurun_out/{tag}_sq*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "mdp_step" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(tag, {k: round(acc[k] / n[k]) for k in sorted(acc)})
