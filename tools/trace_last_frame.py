#!/usr/bin/env python3
"""Print the launches of the last frame in a rocprofv3 kernel-trace CSV, in start order:
kernel, duration, gap to the previous launch's end.  python tools/trace_last_frame.py t_kernel_trace.csv"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


# a frame starts at a ray-generating launch (template flag GEN, after TAB, true); the last frame that does not
# count events (last template flag false) is the one a frame loop runs
starts = [i for i, r in enumerate(rows) if re.search(r"k_trace_shade<\d+, true", r["Kernel_Name"])]
if not starts:
    sys.exit("no ray-generating launch found")
def counts_events(name):  # template flag STATS (sixth) of k_trace_shade
    args = [a.strip() for a in re.search(r"k_trace_shade<([^>]*)>", name).group(1).split(",")]
    return len(args) > 5 and args[5] == "true"


plain = [i for i in starts if not counts_events(rows[i]["Kernel_Name"])] or starts
first = plain[-1]
later = [i for i in starts if i > first]
frame = rows[first:(later[0] if later else len(rows))]
t0 = int(frame[0]["Start_Timestamp"])
prev_end = t0
total = 0
for r in frame:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-70s %8.1f us  (gap %6.1f us)  grid %s" % (short(r["Kernel_Name"]), (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", "?")))
    prev_end = max(prev_end, e)
    total += e - s
print("frame: %.1f us wall, %.1f us in kernels" % ((prev_end - t0) / 1e3, total / 1e3))
