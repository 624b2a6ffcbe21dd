#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv per (kernel, counter).  python tools/pmc_sum.py file.csv [...]"""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(float)
    n = defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    for (k, c), v in sorted(acc.items()):
        print("%-34s %-28s %.4g  (%d dispatches)" % (k[:34], c, v, len(n[k])))
