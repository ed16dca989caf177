"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (developer tool)."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        if "b4d" not in k:
            continue
        print(d.split("/")[-1], k, {c: f"{v / cnt[k][c]:.4g}" for c, v in acc[k].items()}, "n=", max(cnt[k].values()))
