#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the north-star target.

A step = one pass of the hot path (`Camera.render -> World.colorAt`) over one frame of synthetic
input: the frozen composite scene (generated teapot mesh + the two glass spheres of
samples/glass_spheres.zig + the reflective floor of samples/reflective_floor.zig, under the camera and
light of samples/teapot.zig:21-36) at 3840x2160, reflection/refraction depth 5 — BASELINE.json's
"Target: teapot + glass scene at 3840x2160, depth 5".  The scene is resident in HBM, the canvas stays
on the device (f32 + the u8 quantisation of canvas.zig:45-47).  metric = Mrays/s, a ray being one
colorAtInternal or one isShadowed call.

    python bench.py --gpus N --steps K --warmup W

N > 1 — one rank per GPU (torch.distributed.run; started from here when WORLD_SIZE is not set), every
rank holds the scene:
* default, "scaling": "strong" — the metric's "frame ms at 1/2/4/8": every frame is split into 16-row
  strips, rank r renders the strips s with s % N == r, the u8 strips are gathered on rank 0 (one RCCL
  gather per step, overlapped with the next step's rendering) and put back in place.
* --whole-frames, "scaling": "weak" — a step is N frames, rank r renders frame r whole; every u8
  canvas is gathered on rank 0.
Time is the max over ranks between two barrier+synchronize pairs; all gathers are inside it.
With fewer visible GPUs than ranks the run is a REHEARSAL (all ranks on GPU 0, gloo): it checks the N > 1
code path (what rank 0 received equals the whole-frame render) and says "rehearsal": true — not a measurement.

Frames in flight (--frames-in-flight F; default 2 at N = 1, 3 at N > 1): frame k runs through work set k % F of the scene
(rtx_opts.work_set: its own ray queues, nodes and counters) on stream k % F into canvases (and, at N > 1, exchange buffers) of its
own, so the deep levels of one frame — few rays, latency-bound, 58-62 % of the VALU issue slots used — share the chip with the next
frame's level 0.  A rank's share of a split frame fills ever less of its chip per launch, so the ranks of an N > 1 run keep three
in flight (tools/strip_share.py: rank 0's eighth of the frame takes 0.47 / 0.28 / 0.23 ms per frame with 1 / 2 / 3).  `value` and
`ms_per_step` are that loop's throughput; `config.frame_ms_one_at_a_time` is the same frame with nothing else in flight.  Timed
steps that carry per-launch HIP events run alone (the other slot is drained first and waits), so `roofline` prices a kernel by
itself; `--headline-only` (the command profiles/ traces) keeps one frame in flight throughout.

Also in the same JSON line (N = 1):
* `roofline` for the dominant kernel, k_trace_shade: algorithmic flops = SURVEY §8d's table x the
  device's own event counts (rtx_stats: shape tests by type, gate tests, BVH box and triangle tests,
  hits, ...), over the HIP-event duration of its launches inside the timed region (every EVENT_EVERY-th
  timed step carries the events); `frac` is quoted against 78.6 TFLOP/s (SURVEY §8d / BASELINE.md §4:
  the f32 vector peak for unpacked per-lane code); the same figure against 39.3 (unfused issue rate,
  what the parity contract allows) and 157.3 (packed FMA, the spec number) is alongside.
* `configs`: frame ms and Mrays/s of BASELINE.json's other single-GPU configurations, measured in this run.
* `frame_ms_with_d2h`: the same frame loop with the u8 canvas copied to pinned host memory inside the
  timed region (double-buffered on a copy stream) — SURVEY §8d's "gather + D2H included" frame time.
* `cpu_baseline`: the oracle (a C port of the reference's algorithm; 8 static row bands like
  src/camera.zig:87-99) timed on this box on a bounded row sample of the same frame, and
  `cpu_baseline_all_threads`: the same on every usable hardware thread.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WIDTH, HEIGHT = 3840, 2160
STRIP_ROWS = 16
EVENT_EVERY = 20  # timed steps whose kernel launches are bracketed by HIP events (roofline.avg_launch_ms)
VALU_PEAK_FMA_TFLOPS = 78.6     # the contract denominator: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz x 2 (FMA), unpacked
VALU_PEAK_UNFUSED_TFLOPS = 39.3  # one flop per VALU instruction: contraction is forbidden by parity (measured 38.7 T lane-ops/s,
                                 # profiles/r01_m_valu_rate.txt)
VALU_PEAK_SPEC_TFLOPS = 157.3   # packed FMA, MI355X_MICROARCH.md "Peak FP32 (vector)"

# SURVEY.md §8(d) flop table (add/sub/mul/div/sqrt = 1, fma = 2, pow = 1, compares = 0; 4-lane dot = 7, mat-vec = 28).
F_RAYGEN = 80
F_XFORM = 56
F_TEST = {0: 44, 1: 2, 2: 16, 3: 48, 4: 58, 5: 55}  # sphere, plane, cube, cylinder, cone, triangle
F_BOX = 16           # AABB / BVH node test
F_BVH_TRIANGLE = 55  # triangle test on a BVH leaf; the mesh's composed transform is the identity (+0)
F_PREPARE = 98       # prepareComputations incl. position
F_SHADOW_SETUP = 24
F_LIGHTING = 70
F_PATTERN = 62       # object-space transform + pattern-space transform + pattern
F_REFRACT = 60       # refraction set-up + Schlick
F_COMBINE = 12


def trace_shade_flops(st):
    """Algorithmic flops of all k_trace_shade launches of one frame, from the device's event counts."""
    f = st.rays_primary * F_RAYGEN
    f += sum(st.shape_tests[t] * (F_XFORM + F_TEST[t]) for t in range(6))  # closest-hit and shadow traversals
    f += (st.gate_tests + st.bvh_box_tests) * F_BOX
    f += st.bvh_triangle_tests * F_BVH_TRIANGLE
    f += st.rays_shadow * (F_PREPARE + F_SHADOW_SETUP + F_LIGHTING + F_COMBINE)  # one per hit
    f += st.hits_patterned * F_PATTERN
    f += st.hits_transparent * F_REFRACT
    return f


def measured_traffic():
    """HBM bytes per k_trace_shade launch of the headline frame from the newest committed PMC pass
    (profiles/r*_traffic.json, written by tools/profile_summarise.py from separate `rocprofv3 --pmc`
    runs: 2 x FETCH_SIZE + WRITE_SIZE).  Counters cannot be read from inside this process, so the
    value is the offline one; None if no pass of this frame has been committed."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("frame") == "composite %dx%d" % (WIDTH, HEIGHT):
            return float(d["bytes_per_launch"]), os.path.relpath(path, ROOT), d.get("library_build")
    return None, None, None


def measured_flops():
    """Hardware flop slots (64 x (ADD + MUL + 2 FMA + TRANS) f32 wave-instructions) of one headline frame's
    k_trace_shade launches from the newest committed counter pass (profiles/r*_flops.json, tools/flop_counters.sh):
    the cross-check of the algorithmic count.  (None, None) if none has been committed."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flops.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("frame") == "composite %dx%d" % (WIDTH, HEIGHT):
            return d["f32"], os.path.relpath(path, ROOT), d.get("library_build")
    return None, None, None


def usable_threads():
    """Hardware threads this process may use: the affinity mask, capped by the box's own hint
    (OMP_NUM_THREADS is set to the CPU share of a one-GPU box; os.cpu_count() shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    hint = os.environ.get("OMP_NUM_THREADS", "")
    if hint.isdigit() and int(hint) > 0:
        n = min(n, int(hint))
    return max(1, n)


def cpu_baseline(cam, world, threads, label, n_shapes):
    """The oracle (C port of the reference algorithm) on a bounded row sample of the headline frame:
    the first row of each of `threads` static row bands (src/camera.zig:87-99)."""
    import oracle_lib as o

    band = HEIGHT // threads
    t0 = time.perf_counter()
    _, counts = o.render(cam, world, rows_quirk=True, row_step=band, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": counts.total / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "machine_threads": os.cpu_count(),
        "sample": "%s: the first row of each of %d static row bands of the headline frame (%d of %d rows, %d rays, %.1f s); "
                  "C restatement of the Zig reference (brute force over all %d shapes, per-test inverse), not the Zig binary"
                  % (label, threads, threads, HEIGHT, counts.total, dt, n_shapes),
    }


def self_launch(args):
    """--gpus N without a torch.distributed.run environment: start the N ranks from here, before
    anything in this process touches a GPU, and pass their output through."""
    import torch  # importing torch and counting devices do not initialise the GPU

    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus:
        env["RTX_BENCH_REHEARSE"] = "1"
        print("bench.py: %d GPU(s) visible, %d ranks asked for: REHEARSAL (all ranks on GPU 0, gloo) — checks the "
              "N > 1 path, measures nothing" % (torch.cuda.device_count(), args.gpus), file=sys.stderr, flush=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def in_process(args):
    """--gpus N --in-process: the same frame through rtx_multi_render_device — one process, one host thread per GPU, strip sets
    handed out from a shared counter (stealing when chunks_per_gpu > 1), peers copying their strips into the canvas on
    GPU 0 over their own xGMI link.  Each call blocks until the frame is complete, so a step is a whole frame, exchange
    included.  With fewer visible GPUs than asked for, device 0 is listed N times: a rehearsal of the path, not a measurement."""
    import torch

    rta = importlib.import_module("ray-tracer_amd")
    n = args.gpus
    visible = torch.cuda.device_count()
    rehearse = visible < n
    devices = [0] * n if rehearse else list(range(n))
    torch.cuda.set_device(0)
    obj_text = rta.teapot_obj_text(1)
    world, cam, keep = rta.composite_scene(obj_text, WIDTH, HEIGHT)
    fcam = rta.flatten_camera(cam)
    multi = rta.MultiScene(world, devices=devices)
    rgb = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device="cuda:0")
    rgb8 = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device="cuda:0")
    stats = multi.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=args.chunks_per_gpu, want_stats=True)
    for _ in range(args.warmup):
        multi.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=args.chunks_per_gpu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        multi.render_device(fcam, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=args.chunks_per_gpu)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rehearse:  # the canvas the peers assembled must be the single-GPU canvas
        single = rta.Scene(world, device=0)
        want = rta.to_ppm_bytes(single.render(fcam)[0])
        single.close()
        assert (rgb8.cpu().numpy() == want).all(), "rtx_multi_render_device: canvas differs from the single-GPU render"
    out = {
        "metric": "Mrays/s (primary+shadow+secondary)", "value": stats.rays_total * args.steps / dt / 1e6, "unit": "Mrays/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "composite (generated teapot.obj, 1568 triangles + glass_spheres' two spheres + reflective_floor's plane and sphere; "
                        "camera/light of samples/teapot.zig) at %dx%d, depth 5, one step = one frame" % (WIDTH, HEIGHT),
            "rays_per_step": int(stats.rays_total), "frame_ms": dt / args.steps * 1e3,
            "parallelism": "one process, %d GPUs: rtx_multi_render_device, %d sets of 16-row strips per GPU taken from a shared counter "
                           "(dynamic stealing), peer copies into the canvas on GPU 0 (peer links enabled: %d); every call blocks until "
                           "the frame is complete" % (n, args.chunks_per_gpu, multi.peer_links),
        },
        "library_build": rta.build_id(),
    }
    if rehearse:
        out["rehearsal"] = True
        out["note"] = "device 0 listed %d times: a check of the in-process N > 1 path, not a measurement" % n
    print(json.dumps(out), flush=True)
    multi.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frame k runs through work set k %% F of the scene (rtx_opts.work_set) on stream k %% F, so the deep, "
                         "latency-bound levels of one frame share the chip with the next frame's level 0; 1 = one frame at a time; "
                         "default: 2 at N = 1, 3 at N > 1 (a rank's share of a split frame fills less of its chip per launch)")
    ap.add_argument("--lanes", type=int, default=1, help="rtx_opts.lanes for the timed frames (2 is an experiment: see rtx.h)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed frame loop: no D2H leg, no f64 report, no other configs, no CPU legs (what profiles/ traces: "
                         "every k_trace_shade launch of the run is then one of the headline frame, running alone)")
    ap.add_argument("--no-events", action="store_true", help="experiment: no per-launch HIP events in the timed loop")
    ap.add_argument("--in-process", action="store_true",
                    help="N > 1: one process drives all N GPUs through rtx_multi_render_device (one host thread per GPU, strip sets "
                         "taken from a shared counter: the path with dynamic stealing; peer copies into a canvas on GPU 0) instead of "
                         "one rank per GPU over RCCL")
    ap.add_argument("--chunks-per-gpu", type=int, default=2, help="--in-process: strip sets per GPU (> 1: stealing)")
    ap.add_argument("--whole-frames", action="store_true",
                    help="N > 1: deal whole frames to the ranks (weak scaling) instead of splitting every frame into strips")
    args = ap.parse_args()
    n = args.gpus
    if args.headline_only:
        args.frames_in_flight = 1  # every launch of the run is then a kernel running alone: what the rocprofv3 averages under profiles/ are of
    if n > 1 and args.in_process:
        sys.exit(in_process(args))
    if n > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    rta = importlib.import_module("ray-tracer_amd")
    tiling = importlib.import_module("ray-tracer_amd.tiling")
    rehearse = os.environ.get("RTX_BENCH_REHEARSE") == "1"  # not a measurement mode; see the module docstring

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if n > 1:
        import torch.distributed as dist

        if world_size != n:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (n, world_size))
        if rehearse:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    # One explicit stream for renders and gathers (torch's default stream has handle 0, which the
    # C ABI reads as "use the scene's own stream").
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    # the copy stream of the D2H leg, created next to the render stream: the runtime deals streams to its few hardware queues in
    # creation order, and a copy stream that lands on the render stream's queue serialises with it (2.7 instead of 1.8 ms per frame)
    copy_stream = torch.cuda.Stream(device=dev)

    obj_text = rta.teapot_obj_text(1)
    world, cam, keep = rta.composite_scene(obj_text, WIDTH, HEIGHT)
    scene = rta.Scene(world, device=dev.index)
    fcam = rta.flatten_camera(cam)
    split = n > 1 and not args.whole_frames
    strips = (STRIP_ROWS, n, rank) if split else None
    max_rows = tiling.slot_rows(HEIGHT, STRIP_ROWS, n) if split else HEIGHT  # equal-size gather slots
    # F frames in flight, each on a stream, canvases, an exchange buffer set and a work set (rtx_opts.work_set) of its own; slot 0
    # is the stream above.  N > 1 defaults to three: a rank's share of a split frame fills ever less of its chip per launch
    # (tools/strip_share.py: rank 0's eighth of the frame takes 0.47 / 0.28 / 0.23 ms with 1 / 2 / 3 in flight).
    F = args.frames_in_flight if args.frames_in_flight else (2 if n == 1 else 3)
    F = max(1, min(3, F))
    n_sets = max(F, 2)  # the D2H leg and the scaling extras alternate between two canvases whatever F is
    slot_streams = [tstream] + [torch.cuda.Stream(device=dev) for _ in range(F - 1)]
    slot_rgb = [torch.zeros((max_rows, WIDTH, 3), dtype=torch.float32, device=dev) for _ in range(F)]
    rgb = slot_rgb[0]
    # u8 canvas sets: while set b travels (to rank 0, or to the host) later steps render into the others.
    rgb8 = [torch.zeros((1, max_rows, WIDTH, 3), dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    slot_done = [torch.cuda.Event() for _ in range(F)]
    exchange = whole_slots = None
    if split:
        exchange = tiling.StripExchange(HEIGHT, WIDTH, STRIP_ROWS, frames=1, device=dev, sets=n_sets)
    elif n > 1:
        gdev = torch.device("cpu") if rehearse else dev
        whole_slots = [[torch.zeros(rgb8[0].shape, dtype=torch.uint8, device=gdev) for _ in range(n)] for _ in range(n_sets)] if rank == 0 else None
    pending = {}

    def finish(b):
        if exchange is not None:
            exchange.finish(b)
        else:
            work, _src = pending.pop(b, (None, None))
            if work is not None:
                work.wait()

    def frame(k, profile):
        # A frame that carries per-launch events runs alone (slot 0 waits for the others, the others wait for it): its
        # kernels must not share the chip with another frame's if their durations are to be those of a kernel.
        slot = 0 if profile else k % F
        if profile:
            for other in range(1, F):
                slot_streams[0].wait_event(slot_done[other])
        with torch.cuda.stream(slot_streams[slot]):  # the exchange's collectives and waits follow torch's current stream
            finish(slot)  # set `slot` is free again once its last gather has landed
            scene.render_device(fcam, slot_rgb[slot].data_ptr(), rgb8[slot].data_ptr(), stream=slot_streams[slot].cuda_stream,
                                strips=strips, compact_rows=split, profile=profile, lanes=args.lanes, work_set=slot)
            if exchange is not None:  # canvas exchange: every peer sends its u8 strips straight to rank 0
                exchange.submit(slot, rgb8[slot])
            elif n > 1:
                src = rgb8[slot].cpu() if rehearse else rgb8[slot]
                pending[slot] = (dist.gather(src, whole_slots[slot] if rank == 0 else None, dst=0, async_op=True), src)
        slot_done[slot].record(slot_streams[slot])
        if profile:
            for other in range(1, F):
                slot_streams[other].wait_event(slot_done[0])

    def fence():
        for b in range(F):
            with torch.cuda.stream(slot_streams[b]):
                finish(b)
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Ray and event counts of this rank's share (deterministic per frame): one counted pass, untimed.
    stats = scene.render_device(fcam, rgb.data_ptr(), None, stream=stream, strips=strips, compact_rows=split, want_stats=True)
    rays_local = stats.rays_total

    for k in range(args.warmup):
        frame(k, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # Per-launch HIP events break the back-to-back dispatch (and force one lane), so only every
        # EVENT_EVERY-th timed step carries them — counted back from the last one, which has to drain the frames in
        # flight anyway (and so that a run of fewer steps still has one).
        frame(k, not args.no_events and (args.steps - 1 - k) % EVENT_EVERY == 0)
    fence()
    dt = time.perf_counter() - t0
    prof = scene.profile_collect()
    frame_ms_alone = None
    if n == 1 and F > 1:  # the same frames one at a time (untimed extra): what a single frame takes from first launch to last
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for k in range(20):
            scene.render_device(fcam, rgb.data_ptr(), rgb8[0].data_ptr(), stream=stream, lanes=args.lanes)
        torch.cuda.synchronize()
        frame_ms_alone = (time.perf_counter() - t_a) / 20 * 1e3
        assert all(torch.equal(rgb8[0], x) for x in rgb8[1:F]), "the canvases of the frame slots differ"

    if rehearse and n > 1 and rank == 0:  # what arrived must be the whole-frame render, byte for byte
        whole = rta.to_ppm_bytes(scene.render(fcam)[0])[None]
        if split:
            assert np.array_equal(exchange.full.numpy(), whole), "re-interleaved strips differ from the whole-frame render"
        else:
            for b in range(F):
                for r in range(n):
                    assert np.array_equal(whole_slots[b][r].numpy(), whole), "canvas set %d of rank %d differs" % (b, r)
        print("rehearsal (%s): the canvases gathered from %d ranks equal the whole-frame render" % (
            "split frames" if split else "whole frames", n), file=sys.stderr, flush=True)

    # What a scaling record needs next to the headline: every rank's own render time for its share (device time of the
    # render calls alone, nothing else in flight) and what one canvas exchange costs when nothing hides it.  Untimed
    # extras, after the measured loop.
    scaling_detail = None
    if n > 1:
        reps = max(3, min(20, args.steps))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record(tstream)
        for k in range(reps):
            scene.render_device(fcam, rgb.data_ptr(), rgb8[k & 1].data_ptr(), stream=stream, strips=strips, compact_rows=split, lanes=args.lanes)
        e1.record(tstream)
        torch.cuda.synchronize()
        render_ms = e0.elapsed_time(e1) / reps
        fence()
        t_x = time.perf_counter()
        for k in range(reps):
            if exchange is not None:
                exchange.submit(0, rgb8[0])
                exchange.finish(0)
            else:
                src = rgb8[0].cpu() if rehearse else rgb8[0]
                dist.gather(src, whole_slots[0] if rank == 0 else None, dst=0)
            torch.cuda.synchronize()
        exchange_ms = (time.perf_counter() - t_x) / reps * 1e3
        mine = torch.tensor([render_ms, exchange_ms, float(rays_local)], dtype=torch.float64, device=torch.device("cpu") if rehearse else dev)
        everyone = [torch.zeros_like(mine) for _ in range(n)]
        dist.all_gather(everyone, mine)
        scaling_detail = {
            "per_rank": [{"rank": r, "render_ms": float(v[0]), "exchange_ms_unhidden": float(v[1]), "rays": int(v[2])} for r, v in enumerate(everyone)],
            "note": "render_ms: device time of a rank's own render calls with nothing else in flight; exchange_ms_unhidden: wall time of one "
                    "canvas exchange (gather on rank 0 + re-interleave) with nothing to overlap it; in the measured loop the exchange of "
                    "step k overlaps the rendering of step k + 1",
        }

    t = torch.tensor([dt, float(rays_local)], dtype=torch.float64, device=torch.device("cpu") if rehearse or n == 1 else dev)
    if n > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rays_step = float(tmax[0]), float(t[1])
    else:
        rays_step = float(rays_local)

    extra = {}
    if args.headline_only:
        args.no_cpu = args.no_configs = True
    if n == 1 and rank == 0 and not args.headline_only:
        # --- the same frame loop with the u8 canvas landing in pinned host memory (SURVEY §8d's frame time) ---
        host8 = [torch.empty(rgb8[0].shape, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        rendered = [torch.cuda.Event() for _ in range(2)]
        copied = [torch.cuda.Event() for _ in range(2)]

        def frame_d2h(k):
            b = k & 1
            if k >= 2:
                tstream.wait_event(copied[b])  # the copy of this set's previous frame has left the device buffer
            scene.render_device(fcam, rgb.data_ptr(), rgb8[b].data_ptr(), stream=stream, lanes=args.lanes)
            rendered[b].record(tstream)
            copy_stream.wait_event(rendered[b])
            with torch.cuda.stream(copy_stream):
                host8[b].copy_(rgb8[b], non_blocking=True)
                copied[b].record(copy_stream)

        d2h_steps = max(10, args.steps // 2)
        for k in range(4):
            frame_d2h(k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(d2h_steps):
            frame_d2h(k)
        torch.cuda.synchronize()
        extra["frame_ms_with_d2h"] = (time.perf_counter() - t1) / d2h_steps * 1e3
        want8 = rgb8[(d2h_steps - 1) & 1].cpu()
        assert torch.equal(host8[(d2h_steps - 1) & 1], want8), "the host copy differs from the device canvas"

        # --- the f64 reporting instantiation on the same frame (north_star's FP64 question; parity mode stays f32) ---
        try:
            f64_frames = 5
            scene.render_device(fcam, rgb.data_ptr(), rgb8[0].data_ptr(), stream=stream, lanes=1, f64=True)  # allocates the double queues
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(f64_frames):
                scene.render_device(fcam, rgb.data_ptr(), rgb8[0].data_ptr(), stream=stream, lanes=1, f64=True)
            torch.cuda.synchronize()
            ms64 = (time.perf_counter() - t2) / f64_frames * 1e3
            img64 = rgb8[0].cpu().numpy()
            scene.render_device(fcam, rgb.data_ptr(), rgb8[0].data_ptr(), stream=stream, lanes=1)
            torch.cuda.synchronize()
            img32 = rgb8[0].cpu().numpy()
            diff = np.abs(img64.astype(np.int32) - img32.astype(np.int32)).max(axis=-1)
            extra["f64_report"] = {
                "frame_ms": ms64, "mrays_per_s": rays_step / ms64 / 1e3,
                "achieved_tflops_whole_frame": trace_shade_flops(stats) / (ms64 * 1e-3) / 1e12,
                "fp64_vector_peak_tflops": 78.6,
                "frac_of_fp64_vector_peak": trace_shade_flops(stats) / (ms64 * 1e-3) / 1e12 / 78.6,
                "pixels_within_1_lsb_of_f32": float((diff <= 1).mean()), "pixels_identical_to_f32": float((diff == 0).mean()),
                "note": "rtx_opts.real = RTX_REAL_F64: every computed quantity, the ray queues and the ray-tree nodes in double, scene "
                        "records f32; canonical-w variants with the hit records in LDS at 3 waves per SIMD where the scene allows them "
                        "(this frame), else the plainest variants at 2; flops are "
                        "the f32 frame's event counts priced with the same table, over the whole frame time (not one kernel)",
            }
        except rta.RtxError as e:  # e.g. not enough memory for the double queues next to the f32 ones
            extra["f64_report"] = {"error": str(e)}

        # --- BASELINE.json's other single-GPU configurations, a few frames each ---
        if not args.no_configs:
            extra["configs"] = other_configs(rta, torch, dev, stream, obj_text, in_flight=F)

    if rank == 0:
        launches = prof.launches_trace_shade
        avg_ms = prof.ms_trace_shade / max(launches, 1)
        flops_frame = trace_shade_flops(stats)
        profiled_steps = (args.steps + EVENT_EVERY - 1) // EVENT_EVERY if not args.no_events else 0
        flops_per_launch = flops_frame * profiled_steps / max(launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, traffic_src, traffic_build = measured_traffic() if n == 1 else (None, None, None)  # the PMC pass is of whole frames
        hw_flops, hw_flops_src, hw_flops_build = measured_flops() if n == 1 else (None, None, None)
        build = rta.build_id()  # hash of the library's sources: counter files quoted here say which build they measured
        frames_per_step = 1 if n == 1 or split else n
        out = {
            "metric": "Mrays/s (primary+shadow+secondary)",
            "value": rays_step * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if n == 1 or split else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "composite (generated teapot.obj, 1568 triangles + glass_spheres' two spheres + reflective_floor's plane and sphere; "
                            "camera/light of samples/teapot.zig) at %dx%d, depth 5, one step = %s" % (
                                WIDTH, HEIGHT, "one frame" if frames_per_step == 1 else "%d frames (one per rank)" % n),
                "rays_per_step": int(rays_step),
                "frame_ms": dt / args.steps * 1e3 / frames_per_step,
                "frames_in_flight": F,
                "frame_ms_one_at_a_time": frame_ms_alone,
                "kernel_variant": {"canonical_w": bool(stats.kernel_flags & 1), "bvh": bool(stats.kernel_flags & 2),
                                   "lds_tables": bool(stats.kernel_flags & 4), "track_n1": bool(stats.kernel_flags & 8),
                                   "lds_hit_records": bool(stats.kernel_flags & 16)},
                "parallelism": ("1 GPU, %d frames in flight (frame k on work set and stream k %% %d: a frame's deep, latency-bound levels share "
                                "the chip with the next frame's level 0; frames with per-launch events run alone)" % (F, F) if F > 1 else "1 GPU") if n == 1 else (
                    "%d ranks x 16-row strips of every frame, RCCL gather + re-interleave of the u8 strips on rank 0, "
                    "overlapped with the next steps; %d frames in flight per rank (work set, stream and exchange buffers k %% %d)" % (n, F, F) if split
                    else "%d ranks x whole frames, one RCCL gather of the u8 canvases per step on rank 0, overlapped "
                         "with the next steps; %d frames in flight per rank" % (n, F)),
            },
            "roofline": {
                "bound": "valu", "kernel": "k_trace_shade",
                "achieved": achieved, "peak": VALU_PEAK_FMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / VALU_PEAK_FMA_TFLOPS,
                "frac_of_unfused_issue_rate": achieved / VALU_PEAK_UNFUSED_TFLOPS, "frac_of_spec_peak": achieved / VALU_PEAK_SPEC_TFLOPS,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "traffic_stale": (traffic_build != build) if traffic is not None else None,
                "hw_flop_counters": ({"flop_slots_per_frame": hw_flops["flop_slots"], "flops_on_active_lanes_per_frame": hw_flops["flops_active_lanes"],
                                      "lanes_active": hw_flops["lanes_active"], "source": hw_flops_src,
                                      "stale": hw_flops_build != build} if hw_flops else None),
                "flops_per_launch": flops_per_launch, "flops_per_frame": flops_frame, "avg_launch_ms": avg_ms, "launches": launches,
                "share_of_step": (prof.ms_trace_shade / profiled_steps) / (dt / args.steps * 1e3) if dt > 0 and profiled_steps else None,
                "events": {"shape_tests": list(stats.shape_tests), "gate_tests": stats.gate_tests, "bvh_box_tests": stats.bvh_box_tests,
                           "bvh_triangle_tests": stats.bvh_triangle_tests, "hits": stats.rays_shadow,
                           "hits_patterned": stats.hits_patterned, "hits_transparent": stats.hits_transparent,
                           "shadow_rays_skipped": stats.shadow_rays_skipped},
                "note": "f32 VALU roofline (no MFMA: divergent per-lane scalar math); peak = 78.6 TFLOP/s, the f32 vector peak for "
                        "unpacked per-lane code (SURVEY 8d); frac_of_unfused_issue_rate prices the same figure against 39.3 "
                        "(parity forbids FMA contraction: one flop per VALU instruction), frac_of_spec_peak against 157.3 (packed "
                        "FMA); flops = SURVEY 8d's table x the device's event counts (`events`, of this rank's share of one frame); "
                        "durations: rank 0's launches of every %d-th timed step, bracketed by HIP events on the render stream (those "
                        "frames run on one stream); traffic = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed "
                        "PMC pass named in traffic_source; *_stale: that file was measured on another build of the library than the one "
                        "loaded now (library_build)" % EVENT_EVERY,
            },
            "library_build": build,
        }
        if scaling_detail is not None:
            out["scaling_detail"] = scaling_detail
        if rehearse and n > 1:
            out["rehearsal"] = True
            out["note"] = "all %d ranks shared GPU 0 over gloo: a check of the N > 1 code path, not a measurement" % n
        out.update(extra)
        if n == 1 and not args.no_cpu:
            import oracle_lib as o  # the checker, here as the reported CPU baseline (never the thing measured above)

            ow, ocam, okeep = world, cam, keep  # the oracle reads the same by-value structs
            out["cpu_baseline"] = cpu_baseline(ocam, ow, 8, "8 threads as src/camera.zig:87", scene.n_shapes)
            nthreads = usable_threads()
            if nthreads != 8:
                out["cpu_baseline_all_threads"] = cpu_baseline(ocam, ow, nthreads, "every usable hardware thread", scene.n_shapes)
        print(json.dumps(out), flush=True)
    scene.close()
    if n > 1:
        dist.destroy_process_group()


def other_configs(rta, torch, dev, stream, obj_text, frames=20, in_flight=2):
    """Frame ms and Mrays/s of BASELINE.json's other single-GPU configurations (device-resident canvases): a frame loop like the
    headline's, `in_flight` frames in flight (frame k on work set and stream k % in_flight), and the same frames one at a time."""
    rows = []
    streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]

    def measure(label, config, world, cam, w, h):
        scene = rta.Scene(world, device=dev.index)
        fcam = rta.flatten_camera(cam)
        rgb = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in range(in_flight)]
        rgb8 = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(in_flight)]
        st = scene.render_device(fcam, rgb[0].data_ptr(), rgb8[0].data_ptr(), stream=stream, want_stats=True)
        for _ in range(2):
            scene.render_device(fcam, rgb[0].data_ptr(), rgb8[0].data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            scene.render_device(fcam, rgb[0].data_ptr(), rgb8[0].data_ptr(), stream=stream)
        torch.cuda.synchronize()
        ms_alone = (time.perf_counter() - t0) / frames * 1e3

        def frame(k):
            b = k % in_flight
            scene.render_device(fcam, rgb[b].data_ptr(), rgb8[b].data_ptr(), stream=streams[b].cuda_stream, work_set=b)

        for k in range(2 * in_flight):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            frame(k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        assert all(torch.equal(rgb8[0], x) for x in rgb8[1:]), "%s: the canvases of the frame slots differ" % label
        flops = trace_shade_flops(st)
        rows.append({"config": config, "scene": label, "size": "%dx%d" % (w, h), "shapes": scene.n_shapes, "rays_per_frame": st.rays_total,
                     "frame_ms": ms, "frames_in_flight": in_flight, "frame_ms_one_at_a_time": ms_alone,
                     "mrays_per_s": st.rays_total / ms / 1e3, "flops_per_frame": flops,
                     # the frame's algorithmic flops over the WHOLE frame time (level kernels + combine + quantise) of the frame
                     # running alone, so a lower bound on the dominant kernel's own fraction
                     "kernel": "k_trace_shade", "achieved_tflops_whole_frame": flops / (ms_alone * 1e-3) / 1e12,
                     "frac": flops / (ms_alone * 1e-3) / 1e12 / VALU_PEAK_FMA_TFLOPS})
        scene.close()
        del rgb, rgb8

    for name in ("glass_spheres", "fresnel"):
        s = rta.Sample(name, 1920, 1080)
        measure(name, 1, s.world, s.camera, 1920, 1080)
    for name in ("cube", "cylinder_world", "cone", "striped_scene"):
        s = rta.Sample(name, 3840, 2160)
        measure(name, 2, s.world, s.camera, 3840, 2160)
    world, cam, keep = rta.teapot_scene(obj_text, 3840, 2160)
    measure("teapot (generated teapot.obj)", 3, world, cam, 3840, 2160)
    world, cam, keep = rta.composite_scene(obj_text, 7680, 4320)
    measure("composite, on one GPU", 4, world, cam, 7680, 4320)
    return rows


if __name__ == "__main__":
    main()
