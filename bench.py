#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its configs[1] workload.

A step = one pass of the hot path (`Camera.render -> World.colorAt`) over one batch of
synthetic input: the reference's glass_spheres AND fresnel scenes, each rendered at
1920x1080 with reflection/refraction depth 5 (BASELINE.json configs[1]), scenes already
resident in HBM, canvases left on the device (f32 + the u8 quantisation of canvas.zig).
metric = Mrays/s, a ray being one colorAtInternal or one isShadowed call.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU; every rank holds the scenes):
* default, "scaling": "weak" — a step is a batch of N frame pairs, rank r renders pair r whole
  and every u8 canvas is gathered on rank 0 with one RCCL gather per step, overlapped with the
  next step's rendering (two canvas buffers).  A configs[1] frame takes 0.4 ms on one GPU, below
  the grain at which splitting it pays (launch floors bound the split at 3.8x on
  8 GPUs), so frames, not strips, are the unit a node deals out at this size.
* --split-frames, "scaling": "strong" — north_star's tile sharding: every frame is split into
  16-row strips, rank r renders the strips s with s % N == r, the strips are gathered and
  re-interleaved on rank 0 (the tool for 4K/8K frames; bound 6.2x at 8K).
Time is the max over ranks between two barrier+synchronize pairs; all gathers are inside it.

Also reported in the same JSON line: `roofline` for the dominant kernel (k_trace_shade:
algorithmic flops from SURVEY §8d's table x device event counts, over the HIP-event duration
of its launches inside the timed region — every EVENT_EVERY-th timed step carries the events,
since bracketing every launch slows the step by 14 % — plus the HBM bytes per launch of the
newest committed PMC pass) and `cpu_baseline` (the oracle — a C port of the
reference's algorithm, 8 static row bands like src/camera.zig:87-99 — timed on this box on
a bounded row sample of the same workload; rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

rta = importlib.import_module("ray-tracer_amd")
tiling = importlib.import_module("ray-tracer_amd.tiling")

WIDTH, HEIGHT = 1920, 1080
SCENES = ["glass_spheres", "fresnel"]
STRIP_ROWS = 16
REHEARSE = os.environ.get("RTX_BENCH_REHEARSE") == "1"  # not a measurement mode; see main()
EVENT_EVERY = 10  # timed steps whose kernel launches are bracketed by HIP events (roofline.avg_launch_ms)
VALU_PEAK_TFLOPS = 39.3  # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz, one flop per VALU instruction: FMA
                         # contraction is forbidden by parity, so a multiply and an add issue
                         # separately, and an unpacked f32 VALU wave-instruction occupies its SIMD
                         # for 4 cycles whatever it does — measured on the box, 4.06-4.23 cycles at
                         # 6 waves per SIMD for mul/add, fma and cmp/cndmask alike
                         # (tools/microbench/valu_rate.hip, profiles/r01_m_valu_rate.txt).  The spec's
                         # 157.3 TFLOP/s is that rate x 2 (FMA) x 2 (v_pk_* packed pairs).
VALU_PEAK_FMA_TFLOPS = 78.6     # same issue rate with every instruction an FMA (not reachable under the parity contract)
VALU_PEAK_SPEC_TFLOPS = 157.3   # packed FMA, MI355X_MICROARCH.md "Peak FP32 (vector)"

# SURVEY.md §8(d) flop table (add/sub/mul/div/sqrt = 1, fma = 2, pow = 1, compares = 0;
# 4-lane dot = 7, mat-vec = 28).
F_RAYGEN = 80
F_XFORM = 56
F_TEST = {0: 44, 1: 2, 2: 16, 3: 48, 4: 58, 5: 55}  # sphere, plane, cube, cylinder, cone, triangle
F_PREPARE = 98       # prepareComputations incl. position
F_SHADOW_SETUP = 24
F_LIGHTING = 70
F_PATTERN = 62       # object-space transform + pattern-space transform + pattern
F_REFRACT = 60       # refraction set-up + Schlick
F_COMBINE = 12


def shape_test_flops(world):
    """Flops to test one ray against every shape of a group-free world."""
    counts = [world.n_spheres, world.n_planes, world.n_cubes, world.n_cylinders, world.n_cones, world.n_triangles]
    return sum(n * (F_XFORM + F_TEST[t]) for t, n in enumerate(counts))


def trace_shade_flops(world, st):
    """Algorithmic flops of all k_trace_shade launches of one frame, from device event counts."""
    per_ray = shape_test_flops(world)
    rays = sum(st.level_rays[l] for l in range(6))
    hits = st.rays_shadow
    f = st.rays_primary * F_RAYGEN
    f += rays * per_ray                       # closest-hit traversal
    f += hits * (F_PREPARE + F_SHADOW_SETUP + per_ray + F_LIGHTING + F_COMBINE)  # shadow any-hit is a full pass
    f += st.hits_patterned * F_PATTERN
    f += st.hits_transparent * F_REFRACT
    return f


def measured_traffic():
    """HBM bytes per k_trace_shade launch from the newest committed PMC pass (profiles/r*_traffic.json,
    written by tools/profile_summarise.py from separate `rocprofv3 --pmc` runs of the same frames:
    2 x FETCH_SIZE + WRITE_SIZE).  Counters cannot be read from inside this process, so the value is
    the offline one; None if no pass has been committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return float(json.load(f)["bytes_per_launch"]), os.path.relpath(files[-1], ROOT)


def cpu_baseline(samples):
    """The oracle (C port of the reference algorithm) on a bounded row sample of the workload."""
    import oracle_lib as o

    row_step = 1
    rays = 0
    t0 = time.perf_counter()
    for s in samples:
        osamp = o.Sample(s.name, WIDTH, HEIGHT)
        _, counts = o.render(osamp.camera, osamp.world, rows_quirk=True, row_step=row_step)
        rays += counts.total
    dt = time.perf_counter() - t0
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 8, "kind": "port",
        "sample": "the whole workload once (row step %d; %d rays, %.1f s), 8 threads on static row bands "
                  "as src/camera.zig:87-99; C restatement of the Zig reference, not the Zig binary" % (row_step, rays, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 80 ms of GPU time: short runs sit 1.5 % lower (clock ramp)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-events", action="store_true", help="experiment: no per-launch HIP events in the timed loop")
    ap.add_argument("--overlap-frames", action="store_true",
                    help="render the two frames of a step on two streams (+6 %% throughput; per-kernel profiles no longer isolated)")
    ap.add_argument("--split-frames", action="store_true",
                    help="N > 1: split every frame into strips across the ranks (strong scaling) instead of dealing whole frames")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    dist = None
    if n > 1:
        import torch.distributed as dist

        assert world == n, "launch with torch.distributed.run --nproc-per-node %d" % n
        if REHEARSE:  # every rank on GPU 0, gloo between them: exercises this file's N > 1 logic on a 1-GPU box
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    # One explicit stream for renders and gathers (torch's default stream has handle 0, which
    # the C ABI reads as "use the scene's own stream"): kernels serialise, so the per-launch
    # HIP-event durations are those of each kernel running alone.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    # --overlap-frames: the two frames of a step are independent, so the second scene can render on
    # a stream of its own and its kernels fill the tails (and the empty deep levels) of the first
    # one's: +6 % Mrays/s.  Off by default: overlapped kernels stretch each other, and the
    # per-kernel durations of `rocprofv3 --kernel-trace --stats` over this command (profiles/)
    # would no longer be those of the kernel running alone, which is what the roofline uses.
    side = torch.cuda.Stream(device=dev) if args.overlap_frames else None

    samples = [rta.Sample(name, WIDTH, HEIGHT) for name in SCENES]
    scenes = [rta.Scene(s.world, device=dev.index) for s in samples]
    cams = [rta.flatten_camera(s.camera) for s in samples]
    split = n > 1 and args.split_frames
    strips = (STRIP_ROWS, n, rank) if split else None
    rows = [rta.rows_for(c, strips=strips) for c in cams]
    max_rows = tiling.slot_rows(HEIGHT, STRIP_ROWS, n) if split else HEIGHT  # equal-size gather slots
    rgb = [torch.zeros((max_rows, WIDTH, 3), dtype=torch.float32, device=dev) for _ in scenes]
    # Two u8 canvas sets: while set b travels to rank 0 the next step renders into set 1 - b.
    rgb8 = [torch.zeros((len(scenes), max_rows, WIDTH, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    gdev = torch.device("cpu") if REHEARSE else dev  # gloo gathers host tensors
    slots = full8 = None
    if n > 1 and rank == 0:
        slots = [[torch.zeros(rgb8[0].shape, dtype=torch.uint8, device=gdev) for _ in range(n)] for _ in range(2)]
        if split:
            full8 = torch.zeros((len(scenes), HEIGHT, WIDTH, 3), dtype=torch.uint8, device=gdev)
    pending = {}  # canvas set -> the gather in flight on it

    def finish(b):
        """Wait for the gather of canvas set b; rank 0 of a split run puts the strips back in place."""
        work, _src = pending.pop(b, (None, None))  # _src: keeps the rehearsal's host copy alive
        if work is None:
            return
        work.wait()  # RCCL: the render stream waits, the host does not
        if split and rank == 0:
            for r in range(n):
                idx = tiling.row_index(HEIGHT, STRIP_ROWS, n, r, True, gdev)
                if idx.numel():
                    full8.index_copy_(1, idx, slots[b][r][:, : idx.numel()])

    def frame(k, profile):
        b = k & 1
        finish(b)  # set b is free again once its last gather has landed
        if side is not None:
            side.wait_stream(tstream)
        for i, sc in enumerate(scenes):
            st_i = side.cuda_stream if (side is not None and i == 1 and not profile) else stream
            sc.render_device(cams[i], rgb[i].data_ptr(), rgb8[b][i].data_ptr(), stream=st_i, strips=strips,
                             compact_rows=split, profile=profile)
        if side is not None:
            tstream.wait_stream(side)
        if n > 1:  # canvas exchange: every peer sends its u8 canvases (or strips) straight to rank 0
            src = rgb8[b].cpu() if REHEARSE else rgb8[b]
            pending[b] = (dist.gather(src, slots[b] if rank == 0 else None, dst=0, async_op=True), src)

    def fence():
        for b in list(pending):
            finish(b)
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Ray counts of this rank's share (deterministic per frame): one counted pass, untimed.
    stats = [sc.render_device(cams[i], rgb[i].data_ptr(), None, stream=stream, strips=strips, compact_rows=split,
                              want_stats=True) for i, sc in enumerate(scenes)]
    rays_local = sum(s.rays_total for s in stats)

    for k in range(args.warmup):
        frame(k, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # Per-launch HIP events cost ~14 % of the step when every launch is bracketed (they break
        # the back-to-back dispatch), so every EVENT_EVERY-th timed step carries them.
        frame(k, not args.no_events and k % EVENT_EVERY == EVENT_EVERY // 2)
    fence()
    dt = time.perf_counter() - t0

    prof = [sc.profile_collect() for sc in scenes]
    if REHEARSE and n > 1 and rank == 0:  # what arrived must be the whole-frame renders, byte for byte
        whole = np.stack([rta.to_ppm_bytes(sc.render(cams[i])[0]) for i, sc in enumerate(scenes)])
        for b in range(2):
            if split:
                assert np.array_equal(full8.numpy(), whole), "re-interleaved strips differ from the whole-frame render"
            else:
                for r in range(n):
                    assert np.array_equal(slots[b][r].numpy(), whole), "canvas set %d of rank %d differs" % (b, r)
        print("rehearsal (%s): the canvases gathered from %d ranks equal the whole-frame renders" % (
            "split frames" if split else "whole frames", n), file=sys.stderr, flush=True)
    t = torch.tensor([dt, float(rays_local)], dtype=torch.float64, device=gdev)
    if n > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rays_step = float(tmax[0]), float(t[1])
    else:
        rays_step = float(rays_local)

    if rank == 0:
        ms_trace = sum(p.ms_trace_shade for p in prof)
        launches = sum(p.launches_trace_shade for p in prof)
        flops_step = sum(trace_shade_flops(s.world, st) for s, st in zip(samples, stats))
        avg_ms = ms_trace / max(launches, 1)
        profiled_steps = len(range(EVENT_EVERY // 2, args.steps, EVENT_EVERY))
        flops_per_launch = flops_step * profiled_steps / max(launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic() if n == 1 else (None, None)  # the PMC pass is of whole frames
        out = {
            "metric": "Mrays/s (primary+shadow+secondary)",
            "value": rays_step * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "glass_spheres + fresnel (samples/*.zig scenes) at 1920x1080, depth 5, one step = "
                            + ("both frames" if n == 1 or split else "%d frame pairs (one per rank)" % n),
                "rays_per_step": int(rays_step),
                "frame_ms": dt / args.steps * 1e3 / (len(SCENES) * (1 if n == 1 or split else n)),
                "parallelism": "1 GPU" if n == 1 else (
                    "%d ranks x 16-row strips of every frame, RCCL gather + re-interleave of the u8 strips on rank 0" % n if split
                    else "%d ranks x whole frames, one RCCL gather of the u8 canvases per step on rank 0, overlapped "
                         "with the next step" % n),
            },
            "roofline": {
                "bound": "valu", "kernel": "k_trace_shade",
                "achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / VALU_PEAK_TFLOPS,
                "frac_of_fma_peak": achieved / VALU_PEAK_FMA_TFLOPS, "frac_of_spec_peak": achieved / VALU_PEAK_SPEC_TFLOPS,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "flops_per_launch": flops_per_launch, "avg_launch_ms": avg_ms, "launches": launches,
                "share_of_step": (ms_trace / profiled_steps) / (dt / args.steps * 1e3) if dt > 0 and profiled_steps else None,
                "note": "f32 VALU roofline (no MFMA: divergent per-lane scalar math); peak = unfused, unpacked f32 issue rate, "
                        "39.3 T lane-ops/s (measured 38.7: profiles/r01_m_valu_rate.txt); frac_of_fma_peak / frac_of_spec_peak "
                        "price the same achieved figure against 78.6 (all-FMA) and 157.3 (packed FMA, the spec number); "
                        "rank 0's launches of every %d-th timed step, bracketed by HIP events on the render stream; "
                        "traffic = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed PMC pass "
                        "named in traffic_source" % EVENT_EVERY,
            },
        }
        if n == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(samples)
        print(json.dumps(out), flush=True)
    for sc in scenes:
        sc.close()
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
