#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<tag>_*.

    python tools/profile_summarise.py r02_a
Writes the kernel-stats CSVs, the bench JSON line, the PMC sums of the headline frame and
profiles/<tag>_traffic.json (HBM bytes per plain k_trace_shade launch of one composite 3840x2160 frame:
2 x FETCH_SIZE + WRITE_SIZE, the gfx950 half-count correction of MI355X_MICROARCH.md applied), which
bench.py reports as roofline.traffic.
"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def pmc(path):
    """(kernel class, counter) -> sum, kernel class -> dispatches.  k_trace_shade is split into the plain
    launches of a frame loop and the event-counting ones of a statistics frame (last template flag)."""
    acc = defaultdict(float)
    n = defaultdict(set)
    for row in csv.DictReader(open(path)):
        full = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("f32::", "").split("(")[0]  # f64:: stays
        k = re.sub(r"<.*", "", full)
        if "k_trace_shade" in full:
            args = [a.strip() for a in re.search(r"k_trace_shade<([^>]*)>", full).group(1).split(",")]
            k += " stats" if len(args) > 5 and args[5] == "true" else " plain"  # template flag STATS
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


shutil.copy(os.path.join(src, "bench", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))            # the profiled command: --headline-only
shutil.copy(os.path.join(src, "bench_full.json"), os.path.join(dst, tag + "_bench_full.json"))  # the default command, unprofiled
for name in ("teapot", "composite4k", "composite8k", "glass_spheres", "fresnel", "cylinder_world"):
    shutil.copy(os.path.join(src, name, "k_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    line = [l for l in open(os.path.join(src, name + ".log")) if " rays " in l][-1]
    open(os.path.join(dst, "%s_%s_frame.txt" % (tag, name)), "w").write(line)

lines = []
traffic = {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0}
for p in ("fetch", "write", "sq", "mem"):
    acc, n = pmc(os.path.join(src, "pmc_composite4k_%s" % p, "pmc_counter_collection.csv"))
    for (k, c), v in sorted(acc.items()):
        if k.startswith("k_"):
            lines.append("%-22s %-24s %.5g  (%d dispatches)" % (k, c, v, n[k]))
        if k == "k_trace_shade plain" and c == "FETCH_SIZE":
            traffic["fetch_kb"] += v
            traffic["launches"] += n[k]
        if k == "k_trace_shade plain" and c == "WRITE_SIZE":
            traffic["write_kb"] += v
open(os.path.join(dst, tag + "_pmc_composite4k.txt"), "w").write(
    "rocprofv3 --pmc, one pass per counter group, RTX_PROFILE=0 RTX_LANES=1 tools/frame.py composite:1 3840 2160 1\n"
    "(two plain frames = 12 'k_trace_shade plain' launches, then one statistics frame = 6 'k_trace_shade stats' launches);\n"
    "FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE not yet doubled)\n" + "\n".join(lines) + "\n")
bytes_all = (2.0 * traffic["fetch_kb"] + traffic["write_kb"]) * 1024.0
bid_path = os.path.join(src, "build_id.txt")
build_id = open(bid_path).read().strip() if os.path.exists(bid_path) else None  # written on the GPU box by profile_round.sh
out = {
    "what": "HBM bytes moved by the plain k_trace_shade launches of composite 3840x2160 frames",
    "library_build": build_id,
    "frame": "composite 3840x2160",
    "fetch_size_kb": traffic["fetch_kb"], "write_size_kb": traffic["write_kb"], "launches": traffic["launches"],
    "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)",
    "bytes_per_launch": bytes_all / max(traffic["launches"], 1),
    "bytes_per_frame": bytes_all / max(traffic["launches"], 1) * 6,
}
json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
