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


class StripExchange:
    """The canvas exchange of a split frame loop: every rank hands over its compact u8 strips
    [frames, slot_rows, W, C]; rank 0 gathers them (one collective per step, asynchronous: RCCL
    runs it on its own stream) and puts every row back in place in `fulls[b]` [frames, H, W, C]
    (`full` is the set finished last).

    `sets` buffer sets (default 2): set b may travel while later steps render into the other sets; `finish(b)`
    must be called before set b's source buffer is written again.  submit / finish act on torch's CURRENT
    stream: a loop that keeps several frames in flight calls them inside `torch.cuda.stream(...)` of the frame's
    own stream; the sets share nothing but the process group, whose collectives run in the order they are issued
    (the same on every rank).  With the gloo backend (CPU tests, single-GPU rehearsals) device tensors are staged
    through host memory.
    """

    def __init__(self, height, width, strip_rows, frames=1, channels=3, rows_quirk=True, device="cpu", group=None, sets=2):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.group = group
        self.height, self.strip_rows, self.rows_quirk = height, strip_rows, rows_quirk
        self.slot_rows = slot_rows(height, strip_rows, self.world)
        self.local_rows = len(rows_of_rank(height, strip_rows, self.world, self.rank, rows_quirk))
        self.device = torch.device(device)
        self.staged = dist.get_backend(group) == "gloo" and self.device.type != "cpu"
        self.gather_device = torch.device("cpu") if self.staged else self.device
        self.sets = sets
        shape = (frames, self.slot_rows, width, channels)
        self.slots = self.fulls = self.full = None
        if self.rank == 0:
            self.slots = [[torch.zeros(shape, dtype=torch.uint8, device=self.gather_device) for _ in range(self.world)] for _ in range(sets)]
            self.fulls = [torch.zeros((frames, height, width, channels), dtype=torch.uint8, device=self.gather_device) for _ in range(sets)]
            self.full = self.fulls[0]
        self.pending = {}

    def submit(self, b, local):
        """Start the gather of canvas set b (a [frames, slot_rows, W, C] u8 tensor on `device`)."""
        assert 0 <= b < self.sets, "set %d of %d" % (b, self.sets)
        assert b not in self.pending, "finish(%d) first" % b
        src = local.cpu() if self.staged else local
        work = dist.gather(src, self.slots[b] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        self.pending[b] = (work, src)  # src: keeps a staged host copy alive until the gather has landed

    def finish(self, b):
        """Wait for set b's gather; on rank 0 re-interleave the strips into `fulls[b]`."""
        work, _src = self.pending.pop(b, (None, None))
        if work is None:
            return
        work.wait()  # RCCL: the current stream waits, the host does not
        if self.rank == 0:
            for r in range(self.world):
                idx = row_index(self.height, self.strip_rows, self.world, r, self.rows_quirk, self.gather_device)
                if idx.numel():
                    self.fulls[b].index_copy_(1, idx, self.slots[b][r][:, : idx.numel()])
            self.full = self.fulls[b]

    def drain(self):
        for b in list(self.pending):
            self.finish(b)
