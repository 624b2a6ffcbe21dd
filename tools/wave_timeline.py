#!/usr/bin/env python3
"""Where a launch's empty wave slots come from: start / end of every wave of every level launch of one frame and the
length of its longest iteration, on the chip-wide 100 MHz clock (library variant: tools/build_variant.sh tl -DRTX_EXP_TIMELINE).

    RTX_LIB=$PWD/ray-tracer_amd/libx_tl.so python tools/wave_timeline.py <scene> <W> <H>
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # before the library: both bring a HIP runtime, and torch's must be the one that gets loaded

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scenes import load, rta  # noqa: E402

L = rta.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "composite:1"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
world, cam, keep = load(name, w, h)
scene = rta.Scene(world)
LEVELS, WAVES = 6, 8192
buf = np.zeros((LEVELS, WAVES, 12), dtype=np.uint64)
ptr = buf.ctypes.data_as(C.POINTER(C.c_ulonglong))
rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
for _ in range(3):  # plain device-resident frames, like a frame loop's
    scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream)
torch.cuda.synchronize()
L.rtx_exp_timeline_read(ptr, 1)
scene.render_device(cam, rgb.data_ptr(), None, stream=stream.cuda_stream)
torch.cuda.synchronize()
L.rtx_exp_timeline_read(ptr, 1)
TICK_US = 0.01  # 100 MHz
for lv in range(LEVELS):
    b = buf[lv]
    used = b[:, 1] > 0
    n = int(used.sum())
    if n == 0:
        continue
    start, end, longest, iters = [b[used, k].astype(np.int64) for k in range(4)]
    t0, t1 = start.min(), end.max()
    dur = (t1 - t0) * TICK_US
    busy = ((end - start).sum() * TICK_US) / (n * dur)
    print("level %d: %d waves, launch %.1f us; waves start within %.1f us (median %.1f); mean wave residency %.0f %% of the launch" % (
        lv, n, dur, (start.max() - t0) * TICK_US, float(np.median(start - t0)) * TICK_US, 100 * busy))
    left = (t1 - end) * TICK_US  # how long before the launch's end each wave left
    qs = [50, 75, 90, 99, 100]
    print("   waves leave before the end by (us): " + ", ".join("p%d %.1f" % (q, float(np.percentile(left, q))) for q in qs))
    it = np.maximum(iters - 1, 0)
    total_iters = int(it.sum())
    mean_iter = ((end - start).sum() * TICK_US) / max(total_iters, 1)
    print("   iterations %d (per wave min %d median %d max %d), mean %.1f us each; longest iteration of a wave (us): " % (
        total_iters, int(it.min()), int(np.median(it)), int(it.max()), mean_iter)
        + ", ".join("p%d %.1f" % (q, float(np.percentile(longest, q)) * TICK_US) for q in qs))
    # the waves with the longest iterations: what those iterations were made of
    order = np.argsort(-longest)[:8]
    bu = b[used]
    for k in order:
        r = bu[k].astype(np.int64)
        print("     longest iteration %.1f us: closest %.1f, predecessor walk %.1f, prepare + shadow %.1f us; box-loop turns %d, triangle-loop turns %d  (%.2f us per turn of the three walks)" % (
            r[2] * TICK_US, r[4] * TICK_US, r[5] * TICK_US, r[6] * TICK_US, r[7], r[8], (r[4] + r[5] + r[6]) * TICK_US / max(r[7] + r[8], 1)))
    print("   all iterations: box-loop turns %d, triangle-loop turns %d" % (int(bu[:, 9].sum()), int(bu[:, 10].sum())))
    # resident waves over time, in tenths of the launch
    edges = np.linspace(t0, t1, 11)
    occ = []
    for a, bb in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(end, bb) - np.maximum(start, a), 0, None).sum() / (bb - a)
        occ.append(overlap / n)
    print("   share of the launch's waves resident per tenth of the launch: " + " ".join("%.2f" % o for o in occ))
scene.close()
