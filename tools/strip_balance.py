#!/usr/bin/env python3
"""Load balance of the static strip interleave (one GPU stands in for every rank in turn).

    python tools/strip_balance.py <scene> <W> <H> <ranks> [strip_rows]

Renders the strips each of `ranks` ranks would own (strip s belongs to rank s % ranks) and prints
per-rank rays and device time, then max / mean — the efficiency a frame split that way can reach
before the gather.  Scenes as in tools/frame.py (sample names, teapot:L, composite:L).
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rta = importlib.import_module("ray-tracer_amd")

name, w, h, ranks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
strip_rows = int(sys.argv[5]) if len(sys.argv) > 5 else 16
if name.split(":")[0] in ("teapot", "composite"):
    level = name.split(":")[1] if ":" in name else "1"
    text = rta.teapot_obj_text(int(level))
    world, cam, keep = (rta.teapot_scene if name.startswith("teapot") else rta.composite_scene)(text, w, h)
else:
    s = rta.Sample(name, w, h)
    world, cam = s.world, s.camera
scene = rta.Scene(world)
_, full = scene.render(cam)
_, full = scene.render(cam)
ms, rays = [], []
for r in range(ranks):
    scene.render(cam, strips=(strip_rows, ranks, r), compact_rows=True)
    best = None
    for _ in range(3):
        _, st = scene.render(cam, strips=(strip_rows, ranks, r), compact_rows=True)
        best = st.ms_total if best is None else min(best, st.ms_total)
    ms.append(best)
    rays.append(st.rays_total)
    print("rank %d: %9d rays  %.3f ms" % (r, st.rays_total, best))
print("%s %dx%d, %d ranks x %d-row strips: whole frame %.3f ms on one GPU; per-rank max %.3f ms, mean %.3f ms "
      "(balance %.3f), rays max/mean %.3f; speed-up bound %.2fx" % (
          name, w, h, ranks, strip_rows, full.ms_total, max(ms), sum(ms) / ranks, (sum(ms) / ranks) / max(ms),
          max(rays) / (sum(rays) / ranks), full.ms_total / max(ms)))
scene.close()
