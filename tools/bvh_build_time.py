#!/usr/bin/env python3
"""Host (binned SAH) against device (LBVH) build of the teapot's triangle BVH: scene-creation time and the
frame time of the resulting tree at 3840x2160.   python tools/bvh_build_time.py [levels...]"""
import ctypes as C
import os
import sys
import time

import torch  # before the library: both bring a HIP runtime

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import rta  # noqa: E402

levels = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]
W, H = 3840, 2160
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
for level in levels:
    world, cam, keep = rta.teapot_scene(rta.teapot_obj_text(level), W, H)
    flat = rta.FlatScene(world)
    row = "level %2d: %7d shapes" % (level, flat.desc.n_shapes)
    canvases = []
    for device_bvh in (False, True):
        best = 1e9
        flat.desc.flags = 1 if device_bvh else 0  # RTX_SCENE_BVH_DEVICE
        for _ in range(3):  # rtx_scene_create alone: the flattening of the world (host, the same for both) is not the builders' time
            h = C.c_void_p()
            t0 = time.perf_counter()
            rc = rta.lib().rtx_scene_create(C.byref(flat.desc), 0, C.byref(h))
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, rta.lib().rtx_last_error()
            rta.lib().rtx_scene_destroy(h)
        scene = rta.Scene(world, device_bvh=device_bvh)
        scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream, lanes=1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(10):
            scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream, lanes=1)
        b.record(stream)
        torch.cuda.synchronize()
        canvases.append(rgb.cpu())
        row += " | %s: rtx_scene_create %8.2f ms, frame %.3f ms" % ("device LBVH" if device_bvh else "host SAH   ", best * 1e3, a.elapsed_time(b) / 10)
        scene.close()
    row += " | canvases equal: %s" % bool(torch.equal(canvases[0], canvases[1]))
    print(row, flush=True)
