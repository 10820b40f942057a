"""The RCCL leg of the sharded path on one GPU: a one-rank "nccl" group runs the same calls a rank of an N-GPU
job makes -- the library's own stream seen as a torch stream, a frame enqueued on it, the asynchronous gather
ordered behind it, the assembly behind the gather -- and the frame that comes out must be the rendered one.
(More than one rank needs more than one GPU: RCCL refuses two ranks on one device.)"""
import os

import pytest
import torch
import torch.distributed as dist

from raymarching_amd import binding, sharding

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def one_rank_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_gather_behind_the_renderers_stream(one_rank_group):
    W, H, band_rows = 256, 192, 4
    params = binding.default_params(width=W, height=H, kernel=2)
    scene = binding.Scene(os.path.join(ROOT, "scenes", "menger2.txt"))
    rs = []
    for _ in range(2):                                  # two frames in flight, as bench.py keeps them
        r = binding.Renderer(0)
        r.set_params(params)
        r.set_scene(scene)
        rs.append(r)
    rows, local_rows = sharding.band_plan(H, band_rows, 1, 0)
    shards = [torch.zeros((local_rows, W, 4), dtype=torch.uint8, device="cuda") for _ in rs]
    gatherers = [sharding.FrameGatherer(s, H, band_rows, 1, 0, always_collective=True) for s in shards]
    streams = [torch.cuda.ExternalStream(r.stream) for r in rs]
    dist.barrier()
    for i in range(6):
        f = i % 2
        with torch.cuda.stream(streams[f]):
            gatherers[f].wait()                         # the previous gather has read this shard
            rs[f].enqueue_device(rows, None, shards[f].data_ptr())
            gatherers[f].start(shards[f])
    for r in rs:
        r.wait()
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    expected = torch.from_numpy(rs[0].render_rgba8()).cuda()
    for g in gatherers:
        assert torch.equal(g.frame(), expected)
    t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [1.5, 2.5]
