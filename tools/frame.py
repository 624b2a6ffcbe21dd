#!/usr/bin/env python3
"""Render one scene a few times on cuda:0 (profiling driver for rocprofv3).

    python tools/frame.py <sample|teapot[:level]|composite[:level]> <W> <H> [frames] [batch_pixels]

RTX_PROFILE=0 (what the tracing scripts set): the frames are plain device-resident frames without
statistics (the kernels a frame loop runs), timed together with HIP events; one counted frame at the
end supplies the ray counts.  Otherwise every frame is a blocking rtx_render with per-launch events.
"""
import os
import sys

if os.environ.get("RTX_PROFILE", "1") == "0":
    import torch  # before the library: both bring a HIP runtime, and torch's must be the one that gets loaded

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "glass_spheres"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 0
no_bvh = os.environ.get("RTX_NO_BVH") == "1"
f64 = os.environ.get("RTX_F64") == "1"  # rtx_opts.real = RTX_REAL_F64 for every frame
world, cam, keep = load(name, w, h)
scene = rta.Scene(world)
print("shapes", scene.n_shapes)
ms_frame = None
if os.environ.get("RTX_PROFILE", "1") == "0":
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream, batch_pixels=batch, no_bvh=no_bvh, f64=f64)  # warm-up, allocations
    a.record(stream)
    for i in range(frames):
        scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream, batch_pixels=batch, no_bvh=no_bvh, f64=f64)
    b.record(stream)
    torch.cuda.synchronize()
    ms_frame = a.elapsed_time(b) / frames
    canvas, st = scene.render(cam, batch_pixels=batch, no_bvh=no_bvh, f64=f64)
else:
    for i in range(frames):
        canvas, st = scene.render(cam, batch_pixels=batch, profile=True, no_bvh=no_bvh, f64=f64)
if os.environ.get("RTX_SAVE"):
    import numpy as np

    np.save(os.environ["RTX_SAVE"], rta.to_ppm_bytes(canvas))
print(name, w, h, "rays", st.rays_primary, st.rays_shadow, st.rays_secondary, "levels", list(st.level_rays),
      "hits", list(st.level_hits), "ms_total %.3f trace %.3f (%d launches) combine %.3f" % (
          ms_frame if ms_frame is not None else st.ms_total, st.ms_trace_shade, st.launches_trace_shade, st.ms_combine))
scene.close()
