"""N > 1 path on CPU: two gloo ranks partition a frame into interleaved strips, gather them on
rank 0 and must reassemble exactly the single-rank canvas.  The strip arithmetic is also checked
against the library's own rtx_rows_for (host code, no GPU needed)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

rta = importlib.import_module("ray-tracer_amd")
tiling = importlib.import_module("ray-tracer_amd.tiling")


@pytest.mark.parametrize("height,strip,world", [(1080, 16, 2), (1080, 16, 8), (500, 8, 4), (70, 16, 3), (64, 64, 2)])
def test_rows_of_rank_matches_library(height, strip, world):
    cam = rta.Sample("cone", 32, height).camera
    seen = []
    for r in range(world):
        rows = tiling.rows_of_rank(height, strip, world, r)
        assert len(rows) == rta.rows_for(cam, strips=(strip, world, r))
        assert len(rows) <= tiling.slot_rows(height, strip, world)
        seen += rows
    assert sorted(seen) == list(range(8 * (height // 8)))  # every rendered row exactly once


def _worker(rank, world, port, height, width, strip, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # "render": pixel value is a function of its global coordinates, as a deterministic renderer's is
        full = (np.arange(height)[:, None, None] * 7 + np.arange(width)[None, :, None] * 3 + np.arange(3)[None, None, :]) % 251
        full = full.astype(np.uint8)
        full[8 * (height // 8):] = 0
        rows = tiling.rows_of_rank(height, strip, world, rank)
        local = torch.zeros((tiling.slot_rows(height, strip, world), width, 3), dtype=torch.uint8)
        local[: len(rows)] = torch.from_numpy(full[rows])
        out = tiling.gather_canvas(local, height, strip)
        if rank == 0:
            ok.value = int(np.array_equal(out.numpy(), full))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,strip", [(1080, 16), (70, 16)])
def test_two_ranks_gather_reassembles_the_canvas(height, strip):
    ctx = mp.get_context("spawn")
    ok = ctx.Value("i", 0)
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, 48, strip, ok)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok.value == 1
