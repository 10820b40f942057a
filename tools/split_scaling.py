"""Kernel time of one rank's share against the split factor N, on one GPU (fit t = W/N + T)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarching_amd as rm  # noqa: E402
from raymarching_amd import sharding  # noqa: E402

r = rm.Renderer(0)
p = rm.default_params()
r.set_params(p)
r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", (sys.argv[1] if len(sys.argv) > 1 else "default") + ".txt")))
buf = torch.empty((p.height + 64, p.width, 4), dtype=torch.uint8, device="cuda")
for world in (1, 2, 4, 8, 16, 32, 64):
    ts = []
    for rank in range(min(world, 4)):
        rows, _ = sharding.band_plan(p.height, 2, world, rank)
        r.render_device(rows, None, buf.data_ptr())
        ms = []
        for _ in range(6):
            r.render_device(rows, None, buf.data_ptr())
            ms.append(r.counters()["kernel_ms"])
        ts.append(min(ms))
    print(f"N={world:3d}: share kernel {sum(ts) / len(ts):.4f} ms   x N = {sum(ts) / len(ts) * world:.3f}", flush=True)
