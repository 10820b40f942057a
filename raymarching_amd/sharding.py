"""Row-band sharding of one frame over the GPUs of a node (one process per GPU).

Every pixel is independent (render::CPU's loops carry no state between pixels,
/root/reference/source/render.cpp:122-139), so a frame shards by rows with one exchange step:
the finished bands are gathered to rank 0.  Bands are narrow and dealt cyclically
(band b -> rank b % world) because cost per row varies by more than 10x between sky and sponge;
interleaving gives every rank a statistically equal share without a shared queue, and keeps all
shards the same size, so a plain gather suffices.
"""
from . import binding


def band_plan(height, band_rows, world, rank):
    """Rows descriptor of `rank`'s share and the number of local rows (padded to whole bands).

    All ranks get the same band_count; bands that lie below the image are skipped by the kernel
    and stay zero.
    """
    if world == 1:
        return binding.Rows(height, 0, 1, 1), height
    n_bands = (height + band_rows - 1) // band_rows
    per_rank = (n_bands + world - 1) // world
    return binding.Rows(band_rows, rank, world, per_rank), per_rank * band_rows


def gather_frame(shard, height, band_rows, world, rank, gathered=None):
    """Gathers equal-sized shards to rank 0 and returns the assembled [height, W, C] frame there
    (None on other ranks).  `shard` is a torch tensor [local_rows, W, C] on the backend's device."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return shard[:height]
    if rank == 0 and gathered is None:
        gathered = [torch.empty_like(shard) for _ in range(world)]
    dist.gather(shard, gathered if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    return assemble(torch.stack(gathered), height, band_rows)


def assemble(stacked, height, band_rows):
    """[world, local_rows, W, C] shards (cyclic bands) -> [height, W, C] frame."""
    world, local_rows, width, ch = stacked.shape
    g = stacked.view(world, local_rows // band_rows, band_rows, width, ch).permute(1, 0, 2, 3, 4)
    return g.reshape(-1, width, ch)[:height]


def band_costs_from_tiles(tile_costs, tile_side, kernel, band_rows, n_local_bands):
    """Cost of each of a rank's bands from the per-tile durations of its last tile-timing render
    (Renderer.tile_costs(): [tiles_y, tiles_x]; a tile row covers tile_side sample rows of the rank's compact
    sample image, a band band_rows * kernel of them)."""
    import numpy as np

    per_tile_row = np.asarray(tile_costs, np.float64).sum(axis=1)
    rows_per_band = band_rows * kernel
    costs = np.zeros(n_local_bands, np.float64)
    for t, c in enumerate(per_tile_row):
        for r in range(t * tile_side, (t + 1) * tile_side):          # the tile row's sample rows, equal shares
            b = r // rows_per_band
            if b < n_local_bands:
                costs[b] += c / tile_side
    return costs


def balanced_band_lists(costs, world):
    """Deals len(costs) bands (a multiple of world) to `world` ranks, the same number to each, so that the summed
    costs are as equal as a greedy longest-first assignment makes them.  Returns an int array [world, per_rank]
    of image band indices, ascending within a rank.  Deterministic: every rank computes the same plan."""
    import numpy as np

    costs = np.asarray(costs, np.float64)
    assert len(costs) % world == 0
    per_rank = len(costs) // world
    load = np.zeros(world)
    taken = [[] for _ in range(world)]
    for b in np.argsort(-costs, kind="stable"):
        open_ranks = [r for r in range(world) if len(taken[r]) < per_rank]
        r = min(open_ranks, key=lambda q: (load[q], q))
        taken[r].append(int(b))
        load[r] += costs[b]
    return np.array([sorted(t) for t in taken], np.int64)


class FrameGatherer:
    """Gathers the shards of one frame -- or of a batch of G frames in one collective -- to rank 0 without
    blocking the host, into buffers allocated once.

    start(shard) issues the gather (with a device backend the call only queues it: the collective is ordered
    after the work queued so far on the current stream, and the current stream waits for it) and, on rank 0,
    queues the de-interleaving copy behind it.  wait() makes the current stream wait for everything start() queued, which is what a stream
    needs before it overwrites its part of `shard` with a later frame.  frame(j) is rank 0's assembled
    [height, W, C] tensor of the batch's j-th frame, valid once the stream has reached the copy.

    One collective costs about 0.1 ms of host time whatever its size; batching G frames per collective keeps
    that off the critical path when a rank's share of a frame takes less than that to render.
    """

    def __init__(self, shard, height, band_rows, world, rank, stage_on_host=False, band_lists=None, always_collective=False):
        """shard: [local_rows, W, C], or [G, local_rows, W, C] for G frames per collective.
        stage_on_host: gather through host copies, blocking, for backends without device collectives
        (gloo with GPU shards: rehearsals and tests only).
        band_lists: [world, per_rank] image band indices per rank (balanced_band_lists) when bands are not dealt
        cyclically; rank r's shard then holds its bands in the order of band_lists[r].
        always_collective: go through the collective and the assembly even when world == 1 (a one-rank group
        exercises the backend's gather on a machine with a single GPU)."""
        import torch

        self.height, self.band_rows, self.world, self.rank = height, band_rows, world, rank
        self.stage_on_host = stage_on_host
        self.collective = world > 1 or always_collective
        self.batched = shard.dim() == 4
        self.frames = shard.shape[0] if self.batched else 1
        local_rows, width, ch = shard.shape[-3:]
        self.row_index = None
        if band_lists is not None and self.collective and rank == 0:
            lists = torch.as_tensor(band_lists, dtype=torch.int64)
            rows = (lists.reshape(-1, 1) * band_rows + torch.arange(band_rows).reshape(1, -1)).reshape(-1)
            self.row_index = rows.to(shard.device)      # frame row of every row of the stacked shards
        self.stack = self.out = None
        if self.collective and rank == 0:
            self.stack = torch.empty((world, self.frames, local_rows, width, ch), dtype=shard.dtype, device=shard.device)
            self.parts = [t.view(shard.shape) for t in self.stack.unbind(0)]
            self.out = torch.empty((self.frames, local_rows * world, width, ch), dtype=shard.dtype, device=shard.device)
        self.single = None
        self.done = torch.cuda.Event() if shard.is_cuda else None

    def start(self, shard):
        import torch.distributed as dist

        if not self.collective:
            self.single = shard
            return None
        if self.stage_on_host:
            import torch
            host = shard.cpu()   # waits for the current stream
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0)
            work = None
            if self.rank == 0:
                self.stack.copy_(torch.stack(parts).view(self.stack.shape))
        else:
            # With NCCL/RCCL the call returns once the collective is queued: it runs on the backend's stream behind
            # the work queued on the current stream so far, and the current stream waits for it -- the host does not.
            # (async_op=True plus wait() does the same through a Work object, at 20 us more per call.)
            dist.gather(shard, self.parts if self.rank == 0 else None, dst=0)
            work = None
        if self.rank == 0:
            world, frames, local_rows, width, ch = self.stack.shape
            if self.row_index is not None:
                for j in range(frames):
                    self.out[j].index_copy_(0, self.row_index, self.stack[:, j].reshape(-1, width, ch))
            else:
                nb = local_rows // self.band_rows
                src = self.stack.view(world, frames, nb, self.band_rows, width, ch).permute(1, 2, 0, 3, 4, 5)
                self.out.view(frames, nb, world, self.band_rows, width, ch).copy_(src)
        if self.done is not None:
            self.done.record()
        return work

    def wait(self, stream=None):
        """`stream` (default: the current one) waits for what the last start() queued (no-op before the first
        start and on CPU)."""
        if self.done is not None and self.collective:
            if stream is None:
                import torch
                stream = torch.cuda.current_stream()
            stream.wait_event(self.done)

    def frame(self, j=0):
        if not self.collective:
            one = self.single[j] if self.batched else self.single
            return one[:self.height]
        return self.out[j][:self.height] if self.rank == 0 else None
