"""prints a rocprofv3 kernel_stats.csv with short kernel names:  python scripts/kstats.py <csv> [substring ...]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", n)
    short = m.group(1) if m else n[:40]
    if want and not any(w in short for w in want):
        continue
    print("%-44s calls %4s  avg %9.1f us  min %9.1f  max %9.1f  total %8.2f ms" % (short[:44], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
