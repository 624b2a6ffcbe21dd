#!/usr/bin/env python3
"""Per-phase cycle shares of k_trace_shade from the diagnostic build (make -C ray-tracer_amd diag).

    python tools/diag_phases.py <sample> <W> <H>
"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402
rta.LIB_PATH = os.path.join(rta.HERE, "libraytracer_amd_diag.so")
L = rta.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "glass_spheres"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
world, cam, keep = load(name, w, h)
scene = rta.Scene(world)
scene.render(cam)
buf = (C.c_ulonglong * 16)()
L.rtx_diag_read(buf, 1)
canvas, st = scene.render(cam)
L.rtx_diag_read(buf, 1)
names = ["ray fetch/gen", "closest traversal", "prepare + shadow setup", "shadow traversal", "lighting/refract/schlick",
         "leaf delivery", "slot allocation atomic", "node + child stores"]
tot = sum(buf[:8])
iters = sum(st.level_rays) / 64.0
print(name, w, h, "ray-wave iterations ~%.0f, fenced cycles per iteration %.0f" % (iters, tot / iters))
for n, v in zip(names, buf[:8]):
    print("  %-26s %5.1f%%  %8.0f cycles/iteration" % (n, 100.0 * v / tot, v / iters))
for which, base in (("closest", 8), ("shadow", 12)):
    for n, v in zip(("gate tests", "linear shapes", "BVH walk", "node record fetch"), buf[base:base + 4]):
        print("    %-8s %-14s %5.1f%%  %8.0f cycles/iteration" % (which, n, 100.0 * v / tot, v / iters))
