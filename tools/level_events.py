#!/usr/bin/env python3
"""Event counts per ray-tree level of one frame: renders with rtx_opts.depth = 1..5 and differences the statistics
(levels 0 and 1 come together: depth 0 means the default).  python tools/level_events.py <scene> <W> <H>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "composite:1"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
world, cam, keep = load(name, w, h)
scene = rta.Scene(world)
prev = None
for depth in range(1, 6):
    _, st = scene.render(cam, depth=depth)
    cur = {"gate": st.gate_tests, "box": st.bvh_box_tests, "tri": st.bvh_triangle_tests, "sphere": st.shape_tests[0], "plane": st.shape_tests[1],
           "hits": st.rays_shadow, "skipped": st.shadow_rays_skipped}
    rays = list(st.level_rays)
    label = "levels 0-1" if prev is None else "level %d" % depth
    d = {k: cur[k] - (prev[k] if prev else 0) for k in cur}
    print("%-10s rays %s  %s  box/gate-test %.1f" % (label, rays[: depth + 1] if prev is None else rays[depth], d, d["box"] / max(d["gate"], 1)))
    prev = cur
scene.close()
