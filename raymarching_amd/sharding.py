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
