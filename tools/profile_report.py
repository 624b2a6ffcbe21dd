#!/usr/bin/env python3
"""The numbers DESIGN.md quotes from a tools/profile_round.sh + tools/flop_counters.sh run:
    python tools/profile_report.py r02_f     (reads profiles/<tag>_* and gpurun_out/<tag>/; writes profiles/<tag>_flops.json)"""
import collections
import csv
import json
import re
import statistics
import sys

tag = sys.argv[1]
d = json.loads(open("profiles/%s_bench.json" % tag).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline-only: %.4f ms  %.0f Mrays/s  events avg %.4f ms  frac %.4f  achieved %.2f  flops/frame %.4g" % (
    d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"], r["achieved"], r["flops_per_frame"]))
print(r["events"])
f = json.loads(open("profiles/%s_bench_full.json" % tag).read().strip().splitlines()[-1])
print("full: %.4f ms  %.0f Mrays/s  events avg %.4f  frac %.4f  d2h %.3f  f64 %.3f ms (%.3f of peak)" % (
    f["ms_per_step"], f["value"], f["roofline"]["avg_launch_ms"], f["roofline"]["frac"], f.get("frame_ms_with_d2h") or 0,
    f["f64_report"]["frame_ms"], f["f64_report"]["frac_of_fp64_vector_peak"]))
print("cpu", f["cpu_baseline"]["value"], f["cpu_baseline_all_threads"]["value"])
for c in f.get("configs", []):
    print("  %-32s %-10s %.3f ms (%s in flight; one at a time %s) %6.0f Mrays/s" % (
        c["scene"], c["size"], c["frame_ms"], c.get("frames_in_flight", 1),
        "%.3f ms" % c["frame_ms_one_at_a_time"] if c.get("frame_ms_one_at_a_time") else "-", c["mrays_per_s"]))
rows = list(csv.DictReader(open("profiles/%s_bench_kernel_stats.csv" % tag)))
tot = n = 0
for row in rows[:6]:
    print("  %-90s %5s %12s %6s" % (row["Name"][:90], row["Calls"], row["AverageNs"], row["Percentage"]))
    if "k_trace_shade" in row["Name"] and "true, false>" not in row["Name"]:
        tot += float(row["AverageNs"]) * int(row["Calls"])
        n += int(row["Calls"])
print("rocprof avg k_trace_shade %.1f us over %d launches" % (tot / n / 1e3, n))
rows = sorted(csv.DictReader(open("gpurun_out/%s/composite4k/k_kernel_trace.csv" % tag)), key=lambda x: int(x["Start_Timestamp"]))
frames, cur = [], []
for row in rows:
    if "k_trace_shade" not in row["Kernel_Name"]:
        continue
    args = [a.strip() for a in row["Kernel_Name"].split("<")[1].split(">")[0].split(",")]
    if args[5] == "true":
        continue
    if args[1] == "true":
        if cur:
            frames.append(cur)
        cur = []
    cur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
frames.append(cur)
frames = [x for x in frames if len(x) == 6]
durs = [statistics.mean(x[l] for x in frames) for l in range(6)]
print("levels (us)", [round(x, 1) for x in durs], "sum %.1f" % sum(durs))
comb = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3 for row in rows if "k_combine" in row["Kernel_Name"]]
print("k_combine x5 %.1f us" % (5 * statistics.mean(comb)))
dd = collections.OrderedDict()
for row in csv.DictReader(open("gpurun_out/%s/pmc_composite4k_sq/pmc_counter_collection.csv" % tag)):
    if "k_trace_shade" in row["Kernel_Name"]:
        dd.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
tot = 0
for l, (k, v) in enumerate(list(dd.items())[:6]):
    cyc = durs[l] * 1e-6 * 2.4e9
    print("  level %d VALU %.3g lanes %.2f issue %.2f" % (l, v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64),
                                                          v["SQ_INSTS_VALU"] * 4 / (cyc * 1024)))
    tot += v["SQ_INSTS_VALU"]
print("total VALU %.3g -> %.3f ms at 100 %% issue" % (tot, tot * 4 / 1024 / 2.4e9 * 1e3))
res = {}
for which in ("f32", "f64"):
    per = collections.OrderedDict()
    for row in csv.DictReader(open("gpurun_out/%s/flops_%s/pmc_counter_collection.csv" % (tag, which))):
        nm = row["Kernel_Name"]
        if "k_trace_shade" not in nm:
            continue
        args = [a.strip() for a in re.search(r"k_trace_shade<([^>]*)>", nm).group(1).split(",")]
        if args[5] == "true":
            continue
        per.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
    nf = len(per) / 6
    t = collections.Counter()
    for v in per.values():
        for k, x in v.items():
            t[k] += x
    S = "F32" if which == "f32" else "F64"
    add, mul, fma, tr = [t["SQ_INSTS_VALU_%s_%s" % (x, S)] / nf for x in ("ADD", "MUL", "FMA", "TRANS")]
    lanes = t["SQ_THREAD_CYCLES_VALU"] / (t["SQ_ACTIVE_INST_VALU"] * 64)
    slots = (add + mul + 2 * fma + tr) * 64
    res[which] = {"add": round(add), "mul": round(mul), "fma": round(fma), "trans": round(tr), "valu": round(t["SQ_INSTS_VALU"] / nf),
                  "lanes_active": round(lanes, 3), "flop_slots": round(slots), "flops_active_lanes": round(slots * lanes), "launches": len(per)}
    print(which, res[which])
import os
bid = "gpurun_out/%s/build_id.txt" % tag  # written on the GPU box by tools/profile_round.sh
json.dump({"what": "hardware flop-instruction counters of the plain k_trace_shade launches of one composite 3840x2160 frame "
                   "(rocprofv3 --pmc, tools/flop_counters.sh)", "frame": "composite 3840x2160", "f32": res["f32"], "f64": res["f64"],
           "library_build": open(bid).read().strip() if os.path.exists(bid) else None,
           "unit": "wave-instructions per frame (add, mul, fma, trans, valu); flops = 64 x (add + mul + 2 fma + trans)"},
          open("profiles/%s_flops.json" % tag, "w"), indent=1)
