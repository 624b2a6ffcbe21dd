#!/usr/bin/env python3
"""What one rank of an N-GPU split frame costs (GPU box): the composite 3840x2160 frame's 16-row strips dealt
round-robin to N ranks, rank 0's share rendered alone in a frame loop.   python tools/strip_share.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

W, H = 3840, 2160
world, cam, keep = load("composite:1", W, H)
scene = rta.Scene(world)
fcam = rta.flatten_camera(cam)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
base = None
for n in ([int(os.environ["SHARE_N"])] if os.environ.get("SHARE_N") else (1, 2, 4, 8)):
    strips = (16, n, 0) if n > 1 else None
    for _ in range(3):
        scene.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), stream=stream.cuda_stream, strips=strips, compact_rows=n > 1, lanes=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(50):
        scene.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), stream=stream.cuda_stream, strips=strips, compact_rows=n > 1, lanes=1)
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    base = base or ms
    print("SHARE 1/%d of the frame: %.3f ms  (ideal %.3f, efficiency if nothing else costs: %.0f %%)" % (n, ms, base / n, 100 * base / n / ms))
scene.close()
