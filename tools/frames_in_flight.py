#!/usr/bin/env python3
"""Throughput of a frame loop with 1, 2 or 3 frames in flight (GPU box): frame k goes to scene k % n on stream k % n (every scene
has its own work buffers), so the deep, latency-bound levels of one frame share the chip with the next frame's level 0.
    python tools/frames_in_flight.py [scene] [W H] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "composite:1"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 60
world, cam, keep = load(name, W, H)
fcam = rta.flatten_camera(cam)
for n in (1, 2, 3):
    scenes = [rta.Scene(world) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    rgb = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
    rgb8 = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    for k in range(2 * n):
        scenes[k % n].render_device(fcam, rgb[k % n].data_ptr(), rgb8[k % n].data_ptr(), stream=streams[k % n].cuda_stream, lanes=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(frames):
        scenes[k % n].render_device(fcam, rgb[k % n].data_ptr(), rgb8[k % n].data_ptr(), stream=streams[k % n].cuda_stream, lanes=1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    same = all(torch.equal(rgb8[0], x) for x in rgb8)
    print("IN FLIGHT %d: %.3f ms per frame (%s)" % (n, ms, "canvases identical" if same else "CANVASES DIFFER"), flush=True)
    for s in scenes:
        s.close()
    del rgb, rgb8
