#!/usr/bin/env python3
"""Where the composite frame's time goes, by leaving parts of the scene out (GPU box):
    python tools/composite_parts.py [W H]
Prints frame time, rays and ns per 1000 rays for: the whole scene, without the mesh group, without the spheres."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
world, cam, keep = load("composite:1", W, H)
fcam = rta.flatten_camera(cam)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
n_groups, n_spheres, n_planes = world.n_groups, world.n_spheres, world.n_planes
for label, g, s, p in (("whole scene", n_groups, n_spheres, n_planes), ("no mesh group", 0, n_spheres, n_planes),
                       ("no spheres", n_groups, 0, n_planes), ("mesh only", n_groups, 0, 0), ("plane only", 0, 0, n_planes)):
    world.n_groups, world.n_spheres, world.n_planes = g, s, p
    scene = rta.Scene(world)
    st = scene.render_device(fcam, rgb.data_ptr(), None, stream=stream.cuda_stream, want_stats=True)
    for _ in range(3):
        scene.render_device(fcam, rgb.data_ptr(), None, stream=stream.cuda_stream, lanes=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(30):
        scene.render_device(fcam, rgb.data_ptr(), None, stream=stream.cuda_stream, lanes=1)
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 30
    rays = st.rays_primary + st.rays_shadow + st.rays_secondary
    print("PART %-14s %.3f ms  rays %5.1f M (shadow %5.1f M, skipped %4.1f M, secondary %5.1f M)  %.1f ps/ray  levels %s" % (
        label, ms, rays / 1e6, st.rays_shadow / 1e6, st.shadow_rays_skipped / 1e6, st.rays_secondary / 1e6, ms * 1e9 / rays,
        [round(x / 1e6, 2) for x in st.level_rays]))
    scene.close()
world.n_groups, world.n_spheres, world.n_planes = n_groups, n_spheres, n_planes
