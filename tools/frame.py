#!/usr/bin/env python3
"""Render one sample scene a few times on cuda:0 (profiling driver for rocprofv3).

    python tools/frame.py <sample> <W> <H> [frames] [batch_pixels]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rta = importlib.import_module("ray-tracer_amd")

name = sys.argv[1] if len(sys.argv) > 1 else "glass_spheres"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 0
no_bvh = os.environ.get("RTX_NO_BVH") == "1"
if name.split(":")[0] in ("teapot", "composite"):
    level = name.split(":")[1] if ":" in name else "1"
    text = rta.teapot_obj_text(int(level))  # in-process: nothing may be spawned under rocprofv3
    world, cam, keep = (rta.teapot_scene if name.startswith("teapot") else rta.composite_scene)(text, w, h)
else:
    s = rta.Sample(name, w, h)
    world, cam = s.world, s.camera
scene = rta.Scene(world)
print("shapes", scene.n_shapes)
for i in range(frames):
    canvas, st = scene.render(cam, batch_pixels=batch, profile=os.environ.get("RTX_PROFILE", "1") != "0", no_bvh=no_bvh)
if os.environ.get("RTX_SAVE"):
    import numpy as np

    np.save(os.environ["RTX_SAVE"], rta.to_ppm_bytes(canvas))
print(name, w, h, "rays", st.rays_primary, st.rays_shadow, st.rays_secondary, "levels", list(st.level_rays),
      "hits", list(st.level_hits), "ms_total %.3f trace %.3f (%d launches) combine %.3f" % (
          st.ms_total, st.ms_trace_shade, st.launches_trace_shade, st.ms_combine))
scene.close()
