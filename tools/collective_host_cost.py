"""Host time per call of the collectives that could carry a frame's shards to rank 0, in a one-rank RCCL group
(the launch path is the same at any world size): python tools/collective_host_cost.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
shard = torch.zeros((4, 96, 1024, 4), dtype=torch.uint8, device="cuda")
stack = torch.empty((1,) + tuple(shard.shape), dtype=torch.uint8, device="cuda")
parts = list(stack.unbind(0))
side = torch.cuda.Stream()
ev = torch.cuda.Event()


def timed(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print(f"{name:50s} {t * 1e6:7.1f} us per call", flush=True)


def gather_async():
    w = dist.gather(shard, parts, dst=0, async_op=True)
    w.wait()


def in_side_stream(fn):
    def run():
        with torch.cuda.stream(side):
            fn()
    return run


timed("dist.gather async + wait", gather_async)
timed("dist.gather", lambda: dist.gather(shard, parts, dst=0))
timed("dist.all_gather_into_tensor", lambda: dist.all_gather_into_tensor(stack, shard))
timed("dist.all_gather_into_tensor async + wait", lambda: dist.all_gather_into_tensor(stack, shard, async_op=True).wait())
timed("dist.all_gather_into_tensor in a side stream", in_side_stream(lambda: dist.all_gather_into_tensor(stack, shard)))
timed("dist.broadcast", lambda: dist.broadcast(shard, 0))
timed("event record + stream wait", lambda: (ev.record(), side.wait_event(ev)))
timed("copy_ of the permuted stack", lambda: shard.copy_(stack[0]))
timed("with torch.cuda.stream", in_side_stream(lambda: None))
dist.destroy_process_group()
