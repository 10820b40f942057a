"""The multi-GPU path's host logic on CPU: 2 gloo ranks shard a frame by cyclic row bands, gather to
rank 0 and must reassemble the original frame exactly (ragged last band and surplus bands included)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, height, width, band_rows, out_path):
    sys.path.insert(0, ROOT)
    from raymarching_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (height, width, 4), dtype=np.uint8))
        rows, local_rows = sharding.band_plan(height, band_rows, world, rank)
        assert (rows.band_rows, rows.first_band, rows.band_stride) == (band_rows, rank, world)
        # what the kernel writes for this rank: its bands, compact; rows below the image stay zero
        shard = torch.zeros((local_rows, width, 4), dtype=torch.uint8)
        for i in range(rows.band_count):
            r0 = (rows.first_band + i * rows.band_stride) * band_rows
            r1 = min(r0 + band_rows, height)
            if r0 < height:
                shard[i * band_rows: i * band_rows + (r1 - r0)] = full[r0:r1]
        frame = sharding.gather_frame(shard, height, band_rows, world, rank)
        # the pipelined path of bench.py: two frames in flight through two gatherers
        gatherers = [sharding.FrameGatherer(shard, height, band_rows, world, rank) for _ in range(2)]
        shards = [shard.clone(), (255 - shard)]
        works = [g.start(sh) for g, sh in zip(gatherers, shards)]
        for w in works:
            if w is not None:
                w.wait()
        # three frames per collective
        batch = torch.stack([shard, 255 - shard, shard // 2])
        gb = sharding.FrameGatherer(batch, height, band_rows, world, rank)
        wb = gb.start(batch)
        if wb is not None:
            wb.wait()
        # bands dealt by cost instead of cyclically: every rank computes the same plan from the same costs
        n_bands = world * rows.band_count
        costs = np.random.default_rng(9).uniform(0, 1, n_bands) ** 3
        plan = sharding.balanced_band_lists(costs, world)
        assert sorted(plan.reshape(-1).tolist()) == list(range(n_bands))
        loads = costs[plan].sum(axis=1)
        assert loads.max() <= costs[np.arange(n_bands).reshape(-1, world).T].sum(axis=1).max() + 1e-12   # no worse than cyclic
        dealt = torch.zeros_like(shard)
        for i, b in enumerate(plan[rank].tolist()):
            r0, r1 = b * band_rows, min((b + 1) * band_rows, height)
            if r0 < height:
                dealt[i * band_rows: i * band_rows + (r1 - r0)] = full[r0:r1]
        g = sharding.FrameGatherer(dealt, height, band_rows, world, rank, band_lists=plan)
        w = g.start(dealt)
        if w is not None:
            w.wait()
        if rank == 0:
            assert frame.shape == full.shape
            ok = (frame == full).all().item() and (g.frame() == full).all().item()
            ok = ok and (gatherers[0].frame() == full).all().item() and (gatherers[1].frame() == 255 - full).all().item()
            ok = ok and all((gb.frame(j) == want).all().item() for j, want in enumerate((full, 255 - full, full // 2)))
            np.save(out_path, ok)
        else:
            assert frame is None and gatherers[0].frame() is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,band_rows", [(37, 4), (64, 8), (5, 4)])
def test_two_rank_gather_reassembles_frame(tmp_path, height, band_rows):
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() + height) % 2000
    mp.spawn(_worker, args=(2, port, height, 19, band_rows, out), nprocs=2, join=True)
    assert bool(np.load(out))


def test_band_plan_covers_every_row_once():
    sys.path.insert(0, ROOT)
    from raymarching_amd import sharding
    for height, band_rows, world in [(768, 4, 8), (2160, 4, 8), (100, 8, 3), (7, 4, 2)]:
        seen = np.zeros(height, int)
        for rank in range(world):
            rows, local_rows = sharding.band_plan(height, band_rows, world, rank)
            assert local_rows == rows.band_count * band_rows
            for i in range(rows.band_count):
                r0 = (rows.first_band + i * rows.band_stride) * band_rows
                seen[r0:min(r0 + band_rows, height)] += 1
        assert (seen == 1).all()
    rows, local_rows = sharding.band_plan(768, 4, 1, 0)
    assert (rows.band_rows, rows.band_count, local_rows) == (768, 1, 768)
