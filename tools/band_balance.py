"""Load balance of cyclic row bands, measured on one GPU: renders every rank's share of an N-way split in
turn and prints max/mean of the trace-kernel times.  python tools/band_balance.py [workload] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarching_amd as rm  # noqa: E402
from raymarching_amd import sharding  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "default"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kw = dict(width=3840, height=2160, kernel=4) if workload == "uhd" else {}
scene = "default" if workload == "uhd" else workload
r = rm.Renderer(0)
p = rm.default_params(**kw)
r.set_params(p)
r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", scene + ".txt")))
whole, _ = sharding.band_plan(p.height, 4, 1, 0)
buf = torch.empty((p.height + 64, p.width, 4), dtype=torch.uint8, device="cuda")
for _ in range(3):
    r.render_device(whole, None, buf.data_ptr())
t_whole = min(_t for _t in [(r.render_device(whole, None, buf.data_ptr()), r.counters()["kernel_ms"])[1] for _ in range(5)])
print(f"{workload}: whole frame {t_whole:.3f} ms; ideal per rank at N={world}: {t_whole / world:.3f} ms", flush=True)
for band_rows in (1, 2, 4, 8, 16):
    times = []
    for rank in range(world):
        rows, local_rows = sharding.band_plan(p.height, band_rows, world, rank)
        r.render_device(rows, None, buf.data_ptr())
        ms = []
        for _ in range(4):
            r.render_device(rows, None, buf.data_ptr())
            c = r.counters()
            ms.append(c["kernel_ms"] + c["resolve_ms"])
        times.append(min(ms))
    mean = sum(times) / len(times)
    print(f"  band_rows {band_rows:2d}: max {max(times):.3f} mean {mean:.3f} ms  max/mean {max(times) / mean:.3f}  "
          f"speed-up over whole frame {t_whole / max(times):.2f}x", flush=True)

# ---- the same split with the bands re-dealt by measured cost (what bench.py does at N > 1)
import numpy as np  # noqa: E402

band_rows = 4
per_rank = sharding.band_plan(p.height, band_rows, world, 0)[0].band_count
costs = np.zeros(world * per_rank)
for rank in range(world):
    rows, local_rows = sharding.band_plan(p.height, band_rows, world, rank)
    r.set_tile_scheduling(False)
    r.set_tile_scheduling(True)                       # forget the order: the next render times its tiles
    r.render_device(rows, None, buf.data_ptr())
    tiles = r.tile_costs()
    side = 8 if tiles.shape[0] * 8 >= local_rows * p.kernel else 16
    costs[rank::world] = sharding.band_costs_from_tiles(tiles, side, p.kernel, band_rows, rows.band_count)
plan = sharding.balanced_band_lists(costs, world)
times = []
rows = rm.binding.Rows(band_rows, 0, 1, per_rank)
for rank in range(world):
    r.set_band_list(plan[rank])
    ms = []
    for _ in range(5):
        r.render_device(rows, None, buf.data_ptr())
        c = r.counters()
        ms.append(c["kernel_ms"] + c["resolve_ms"])
    times.append(min(ms))
r.set_band_list(None)
mean = sum(times) / len(times)
print(f"  band_rows {band_rows:2d}, dealt by measured cost: max {max(times):.3f} mean {mean:.3f} ms  max/mean {max(times) / mean:.3f}  "
      f"(planned cost max/mean {costs[plan].sum(axis=1).max() / costs[plan].sum(axis=1).mean():.3f}, "
      f"cyclic {max(costs[q::world].sum() for q in range(world)) / (costs.sum() / world):.3f})", flush=True)
