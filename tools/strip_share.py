#!/usr/bin/env python3
"""What one rank of an N-GPU split frame costs (GPU box): the composite 3840x2160 frame's 16-row strips dealt
round-robin to N ranks, rank 0's share rendered alone in a frame loop — one frame at a time, and with two and three
frames in flight (frame k on work set and stream k % F, rtx_opts.work_set).   python tools/strip_share.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

W, H = 3840, 2160
FRAMES = 60
world, cam, keep = load("composite:1", W, H)
scene = rta.Scene(world)
fcam = rta.flatten_camera(cam)
rgb = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(3)]
rgb8 = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
base = {}
for n in ([int(os.environ["SHARE_N"])] if os.environ.get("SHARE_N") else (1, 2, 4, 8)):
    strips = (16, n, 0) if n > 1 else None
    line = []
    for flight in (1, 2, 3):
        def frame(k):
            s = k % flight
            scene.render_device(fcam, rgb[s].data_ptr(), rgb8[s].data_ptr(), stream=streams[s].cuda_stream, strips=strips,
                                compact_rows=n > 1, lanes=1, work_set=s)
        for k in range(2 * flight):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(FRAMES):
            frame(k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / FRAMES * 1e3
        base.setdefault(flight, ms)
        line.append("%d in flight %.3f ms (%.0f %% of ideal)" % (flight, ms, 100 * base[flight] / n / ms))
        assert all(torch.equal(rgb8[0], rgb8[s]) for s in range(flight)), "canvases of the frame slots differ"
    print("SHARE 1/%d of the frame: %s" % (n, "; ".join(line)), flush=True)
scene.close()
