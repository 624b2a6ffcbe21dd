#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<tag>_*.

    python tools/profile_summarise.py r01_d
Writes the kernel-stats CSVs, the bench JSON line, the PMC sums and profiles/<tag>_traffic.json
(HBM bytes per k_trace_shade launch of one bench step: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950
half-count correction of MI355X_MICROARCH.md applied), which bench.py reports as roofline.traffic.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def pmc(path):
    acc = defaultdict(float)
    n = defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


shutil.copy(os.path.join(src, "bench", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
for name in ("teapot", "composite4k", "composite8k"):
    stem = "teapot" if name == "teapot" else "composite"
    shutil.copy(os.path.join(src, name, stem + "_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    line = [l for l in open(os.path.join(src, name + ".log")) if " rays " in l][-1]
    open(os.path.join(dst, "%s_%s_frame.txt" % (tag, name)), "w").write(line)

lines = []
traffic = {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0}
for scene in ("glass_spheres", "fresnel"):
    for p in ("fetch", "write", "sq"):
        acc, n = pmc(os.path.join(src, "pmc_%s_%s" % (scene, p), "pmc_counter_collection.csv"))
        for (k, c), v in sorted(acc.items()):
            if k.startswith("k_"):
                lines.append("%-14s %-14s %-24s %.5g  (%d dispatches)" % (scene, k, c, v, n[k]))
            if k == "k_trace_shade" and c == "FETCH_SIZE":
                traffic["fetch_kb"] += v
                traffic["launches"] += n[k]
            if k == "k_trace_shade" and c == "WRITE_SIZE":
                traffic["write_kb"] += v
open(os.path.join(dst, tag + "_pmc_bench_scenes_1080p.txt"), "w").write(
    "rocprofv3 --pmc, one pass per counter group, tools/frame.py <scene> 1920 1080 1 (one frame = 6 k_trace_shade launches);\n"
    "FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE not yet doubled)\n" + "\n".join(lines) + "\n")
bytes_step = (2.0 * traffic["fetch_kb"] + traffic["write_kb"]) * 1024.0
out = {
    "what": "HBM bytes moved by the k_trace_shade launches of one bench step (one glass_spheres + one fresnel frame at 1920x1080)",
    "fetch_size_kb": traffic["fetch_kb"], "write_size_kb": traffic["write_kb"], "launches": traffic["launches"],
    "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)",
    "bytes_per_step": bytes_step, "bytes_per_launch": bytes_step / max(traffic["launches"], 1),
}
json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
