"""Timeline of the last generation in a rocprofv3 kernel trace (CSV): per-kernel start offset, duration and the gap to
the previous kernel's end.
    python scripts/timeline.py <..._kernel_trace.csv> [first-kernel-name-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_pilot_shift"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the pilot shift is launched once per generation for the PLS statistics (and once more by the covariance pass): take the
# generation that starts at the third-last marker of the first kind
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:40]
if len(starts) < 2:
    sys.exit("marker kernel not found")
gens = [starts[0]]
for i in starts[1:]:
    if int(rows[i]["Start_Timestamp"]) - int(rows[gens[-1]]["Start_Timestamp"]) > 300000:      # > 0.3 ms apart: a new generation
        gens.append(i)
a, b = gens[-2], gens[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
print("%-42s %9s %9s %9s" % ("kernel", "start us", "dur us", "gap us"))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-42s %9.1f %9.1f %9.1f" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += e - s
    prev_end = max(prev_end, e)
print("generation: %.1f us from first start to next generation's first start, %.1f us busy, %d kernels" % (
    (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
