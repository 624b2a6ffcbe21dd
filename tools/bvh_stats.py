#!/usr/bin/env python3
"""Tree-quality statistics of the triangle BVH on the host (no GPU): box tests and triangle tests
per camera ray of the teapot scene.    python tools/bvh_stats.py [level] [W H]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rta = importlib.import_module("ray-tracer_amd")
st = rta.st

level = sys.argv[1] if len(sys.argv) > 1 else "1"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 192
h = int(sys.argv[3]) if len(sys.argv) > 3 else 108
text = rta.teapot_obj_text(int(level))
world, cam, keep = rta.teapot_scene(text, w, h)
flat = rta.FlatScene(world)
rays_o, rays_d = [], []
r = st.Ray()
for y in range(h):
    for x in range(w):
        rta.lib().rt_ray_for_pixel(C.byref(cam), x, y, C.byref(r))
        rays_o.append(list(r.origin))
        rays_d.append(list(r.direction))
o = np.array(rays_o, np.float32)
d = np.array(rays_d, np.float32)
nn, nt, bad = C.c_uint32(), C.c_uint32(), C.c_uint32()
assert rta.lib().rtx_bvh_selftest(C.byref(flat.desc), rta.fptr(o), rta.fptr(d), len(o), C.byref(nn), C.byref(nt), C.byref(bad)) == 0
boxes, tris = C.c_uint64(), C.c_uint64()
rta.lib().rtx_bvh_selftest_work(C.byref(boxes), C.byref(tris))
print("teapot level %s: %d shapes, %d BVH nodes, %d triangles in leaves; %d camera rays (one tree per gate class): "
      "%.1f box tests and %.2f triangle tests per ray, %d mismatches against brute force" % (
          level, flat.desc.n_shapes, nn.value, nt.value, len(o), boxes.value / len(o), tris.value / len(o), bad.value))
