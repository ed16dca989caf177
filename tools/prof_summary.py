"""Print the b4d kernels of the newest rocprofv3 kernel_stats.csv under a directory (developer tool)."""
import csv
import glob
import os
import sys

fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    n = r["Name"]
    if "b4d" in n:
        print(n[:70].ljust(72), r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "total %.2f ms" % (float(r["TotalDurationNs"]) / 1e6),
              r["Percentage"])
