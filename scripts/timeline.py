"""Timeline of the last complete generation in a rocprofv3 kernel trace (CSV): per-kernel start offset, duration and the gap
to the previous kernel's end (negative: it overlaps work on another stream).
    python scripts/timeline.py <..._kernel_trace.csv>
A generation has exactly one model fit (k_pls_fit*; k_simple_dist for FILTER::SIMPLE) -- the projection kernels also run for the
Wilcoxon rule's scores since round 5 --; it starts at the k_pilot_shift before it."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


marks = [i for i, r in enumerate(rows) if "k_pls_fit" in r["Kernel_Name"] or "k_simple_dist" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit("fewer than two generations in the trace")


def start_of(m):
    i = m
    while i > 0 and "k_pilot_shift" not in rows[i]["Kernel_Name"]:
        i -= 1
    return i


a, b = start_of(marks[-2]), start_of(marks[-1])
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
print("%-46s %9s %9s %9s %s" % ("kernel", "start us", "dur us", "gap us", "queue"))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s %9.1f %9.1f %9.1f %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "")))
    busy += e - s
    prev_end = max(prev_end, e)
print("generation: %.1f us from first start to next generation's first start, %.1f us of kernel time, %d kernels" % (
    (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
