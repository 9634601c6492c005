"""Summarise a `rocprofv3 --pmc ... --kernel-trace --output-format csv` directory: per kernel, mean of each counter per
launch (summed over the dimension instances rocprofv3 reports).  usage: pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(launches[k])
    print("%s  (%d launches)" % (k, n))
    for c, v in sorted(acc[k].items()):
        print("    %-32s %.4g per launch" % (c, v / n))
