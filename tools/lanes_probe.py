#!/usr/bin/env python3
"""Why do two lanes lose in a frame loop?  Wall and event time of composite 3840x2160 frame loops with and
without the u8 canvas, lanes 1 / 2.   python tools/lanes_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

NF = int(os.environ.get("PROBE_FRAMES", "50"))
W, H = 3840, 2160
if os.environ.get("PROBE_STREAM_FIRST") == "1":
    torch.cuda.set_device(0)
    early = torch.cuda.Stream()
world, cam, keep = load("composite:1", W, H)
scene = rta.Scene(world, device=0) if os.environ.get("PROBE_DEVARG") == "1" else rta.Scene(world)
fcam = rta.flatten_camera(cam)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
rgb8b = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
ALT = os.environ.get("PROBE_ALT") == "1"
stream = early if os.environ.get("PROBE_STREAM_FIRST") == "1" else torch.cuda.Stream()
torch.cuda.set_stream(stream)
if os.environ.get("PROBE_STATS") == "1":
    scene.render_device(fcam, rgb.data_ptr(), None, stream=stream.cuda_stream, want_stats=True)
if os.environ.get("PROBE_DEV") == "1":
    scene.close()
    scene = rta.Scene(world, device=0)
for lanes in ((2, 1) if os.environ.get("PROBE_ORDER") == "21" else (1, 2)):
    for with8 in (False, True):
        for _ in range(3):
            scene.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr() if with8 else None, stream=stream.cuda_stream, lanes=lanes)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(stream)
        for k in range(NF):
            scene.render_device(fcam, rgb.data_ptr(), ((rgb8b if ALT and k & 1 else rgb8).data_ptr()) if with8 else None, stream=stream.cuda_stream, lanes=lanes)
        b.record(stream)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("lanes %d u8 %-5s: enqueue %.3f ms/frame, wall %.3f ms/frame, events %.3f ms/frame" % (
            lanes, with8, t_enq / NF * 1e3, t_all / NF * 1e3, a.elapsed_time(b) / NF))
scene.close()
