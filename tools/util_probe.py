#!/usr/bin/env python3
"""Event counts of one frame (GPU box): python tools/util_probe.py <scene> <W> <H>.
With RTX_LIB pointing at a -DRTX_EXP_SLOTS build the same fields hold lane slots, so
events / slots is the lane utilisation of each loop."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
world, cam, keep = load(name, w, h)
scene = rta.Scene(world)
_, st = scene.render(cam)
print("EVENTS", name, "box", st.bvh_box_tests, "tri", st.bvh_triangle_tests, "gate", st.gate_tests, "shape", list(st.shape_tests),
      "rays", st.rays_primary + st.rays_shadow + st.rays_secondary, "shadow", st.rays_shadow, "shadow_skipped", st.shadow_rays_skipped)
scene.close()
