#!/usr/bin/env python3
"""Static VALU instruction count per source line of one k_trace_shade instantiation (CPU box).

    python tools/isa_lines.py [mangled-name-substring]      default: the bench kernel (queue-fed, TRACK, LDS, C, no BVH)

Compiles rtx_device.hip to assembly with line tables and attributes every v_* instruction to the
last .loc seen.  Static counts: a guide to where the instruction budget of straight-line phases goes.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ray-tracer_amd", "csrc", "rtx_device.hip")
want = sys.argv[1] if len(sys.argv) > 1 else "k_trace_shadeILb0ELb1ELb1ELb1ELb0E"
asm = "/tmp/isa_lines.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                       "-fno-slp-vectorize", "-DRT_BUILD", "-gline-tables-only", "--cuda-device-only", "-S", "-o", asm, SRC],
                      stderr=subprocess.DEVNULL)
lines = open(SRC).read().splitlines()
count = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
inside = False
cur = 0
total = 0
for l in open(asm):
    if not inside:
        if l.startswith("_ZN") and want in l and l.rstrip().endswith(":") or (l.startswith("_ZN") and want in l.split(":")[0]):
            inside = True
        continue
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    m = re.match(r"\s+(v_\w+)", l)
    if m:
        count[cur] += 1
        kinds[cur][m.group(1)] += 1
        total += 1
print("total static VALU instructions:", total)
for ln, c in count.most_common(60):
    src = lines[ln - 1].strip() if 0 < ln <= len(lines) else "?"
    top = ", ".join("%s x%d" % (k.replace("_e32", "").replace("_e64", ""), v) for k, v in kinds[ln].most_common(3))
    print("%5d  %4d  %-70s  [%s]" % (ln, c, src[:70], top))
