#!/usr/bin/env python3
"""Static instruction count per source line of one k_trace_shade instantiation (CPU box).

    python tools/isa_lines.py [mangled-name-substring] [top-n]    default: the headline's level-0 kernel

Compiles rtx_device.hip to assembly with line tables and attributes every instruction to the last .loc seen
(file and line: the arithmetic lives in rtx_trace.inc).  Static counts: a guide to where the instruction budget
of straight-line phases goes — loops execute their lines many times.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ray-tracer_amd", "csrc", "rtx_device.hip")
want = sys.argv[1] if len(sys.argv) > 1 else "k_trace_shadeILi2ELb1ELb1ELb1ELb1ELb0ELb0ELb0E"
top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
asm = "/tmp/isa_lines.s"
if not os.environ.get("ISA_REUSE") or not os.path.exists(asm):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                           "-fno-slp-vectorize", "-DRT_BUILD", "-gline-tables-only", "--cuda-device-only", "-S", "-o", asm, SRC],
                          stderr=subprocess.DEVNULL)
files = {}
count = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
inside = False
cur = (0, 0)
tot = collections.Counter()
for l in open(asm):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
        continue
    if not inside:
        if l.startswith("_ZN") and want in l.split(":")[0]:
            inside = True
        continue
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+((?:v|s|ds|global|buffer|flat|scratch)_\w+)", l)
    if m:
        op = m.group(1)
        cls = "valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else "mem")
        count[cur] += 1
        kinds[cur][op] += 1
        tot[cls] += 1
print("static instructions:", dict(tot))
cache = {}
def src_line(f, n):
    path = files.get(f, "")
    if path not in cache:
        try:
            cache[path] = open(path).read().splitlines()
        except OSError:
            cache[path] = []
    ls = cache[path]
    return (os.path.basename(path), ls[n - 1].strip() if 0 < n <= len(ls) else "?")
for (f, ln), c in count.most_common(top_n):
    name, text = src_line(f, ln)
    top = ", ".join("%s x%d" % (k.replace("_e32", "").replace("_e64", ""), v) for k, v in kinds[(f, ln)].most_common(3))
    print("%-16s %5d  %4d  %-80s  [%s]" % (name[:16], ln, c, text[:80], top))
