#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv per (kernel, counter), plain (non-STATS) k_trace_shade launches listed per
level of the last frame.  python tools/pmc_sum.py file.csv [...]"""
import csv
import re
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(float)
    n = defaultdict(set)
    per_dispatch = defaultdict(dict)
    names = {}
    for row in csv.DictReader(open(path)):
        full = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("f32::", "").split("(")[0]  # f64:: stays
        k = re.sub(r"<.*", "", full)
        if "k_trace_shade" in full:
            args = [a.strip() for a in re.search(r"k_trace_shade<([^>]*)>", full).group(1).split(",")]
            k += " stats" if len(args) > 5 and args[5] == "true" else " plain"  # template flag STATS
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
        d = int(row["Dispatch_Id"])
        per_dispatch[d][row["Counter_Name"]] = per_dispatch[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        names[d] = k
    for (k, c), v in sorted(acc.items()):
        print("%-34s %-28s %.5g  (%d dispatches)" % (k[:34], c, v, len(n[k])))
    plain = sorted(d for d in names if names[d] == "k_trace_shade plain")
    if plain:
        last = plain[-6:] if len(plain) >= 6 else plain
        print("last plain frame, per launch:")
        for d in last:
            print("  dispatch %d: %s" % (d, "  ".join("%s=%.4g" % kv for kv in sorted(per_dispatch[d].items()))))
