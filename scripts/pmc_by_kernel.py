"""per-kernel averages of a rocprofv3 --pmc counter_collection.csv:  python scripts/pmc_by_kernel.py <csv> [substring ...]"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", n)
    short = m.group(1) if m else n[:40]
    if sys.argv[2:] and not any(w in short for w in sys.argv[2:]):
        continue
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[short].add(r["Dispatch_Id"])
for k in sorted(acc):
    c = len(calls[k])
    print("%-28s calls %3d  " % (k[:28], c) + "  ".join("%s=%.4g" % (cn, v / c) for cn, v in sorted(acc[k].items())))
