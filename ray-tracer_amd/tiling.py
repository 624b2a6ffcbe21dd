"""Row-strip sharding of a frame across ranks and the canvas gather (one process per GPU).

The reference scales by static row bands over 8 threads (src/camera.zig:87-99); across GPUs
the bands become interleaved strips: rank r renders strips s with s % world == r, a strip
being `strip_rows` consecutive rows (rtx_opts.strip_rows / strip_count / strip_index), into a
compact buffer, and rank 0 gathers the strips (RCCL when the tensors live on GPUs, gloo in the
CPU tests) and puts every row back in its place.  Pure index arithmetic + one collective.
"""
import torch
import torch.distributed as dist


def rendered_rows(height, rows_quirk=True):
    """Rows the reference renders at all: 8*floor(H/8) (src/camera.zig:87-88)."""
    return 8 * (height // 8) if rows_quirk else height


def rows_of_rank(height, strip_rows, world, rank, rows_quirk=True):
    """Global row indices rank `rank` renders, in the order of its compact buffer."""
    total = rendered_rows(height, rows_quirk)
    rows = []
    s = rank
    while s * strip_rows < total:
        rows.extend(range(s * strip_rows, min((s + 1) * strip_rows, total)))
        s += world
    return rows


def slot_rows(height, strip_rows, world):
    """Rows of the equal-size gather slot (the largest share any rank can have)."""
    strips = -(-height // strip_rows)
    return -(-strips // world) * strip_rows


_ROW_INDEX = {}  # (height, strip_rows, world, rank, rows_quirk, device) -> long tensor of global rows


def row_index(height, strip_rows, world, rank, rows_quirk, device):
    """rows_of_rank as a device tensor, built once: the frame loop must not copy indices per frame."""
    key = (height, strip_rows, world, rank, rows_quirk, str(device))
    idx = _ROW_INDEX.get(key)
    if idx is None:
        idx = torch.as_tensor(rows_of_rank(height, strip_rows, world, rank, rows_quirk), dtype=torch.long, device=device)
        _ROW_INDEX[key] = idx
    return idx


def gather_canvas(local, height, strip_rows, rows_quirk=True, group=None, out=None, slots=None):
    """Gather compact per-rank strips [slot_rows, W, C] on rank 0 into a full [H, W, C] canvas.

    Returns the canvas on rank 0 (rows never rendered stay zero), None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == 0 and slots is None:
        slots = [torch.zeros_like(local) for _ in range(world)]
    dist.gather(local, slots if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None
    if out is None:
        out = torch.zeros((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = row_index(height, strip_rows, world, r, rows_quirk, local.device)
        if idx.numel():
            out.index_copy_(0, idx, slots[r][: idx.numel()])
    return out
