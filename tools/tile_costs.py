"""Cost map of a frame: the per-tile durations the first render of a configuration measures, as statistics, the
most expensive tiles, and a heat map PNG (white = the most expensive tile).

    python tools/tile_costs.py [scene name under scenes/] [--map gpurun_out/tile_costs.png]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import raymarching_amd as rm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="default")
ap.add_argument("--map", default=os.path.join(ROOT, "gpurun_out", "tile_costs.png"))
a = ap.parse_args()

r = rm.Renderer(0)
r.set_params(rm.default_params())
r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", a.scene + ".txt")))
r.render_rgba8()
first = r.counters()["kernel_ms"]
cost = r.tile_costs().astype(np.float64) / 100.0   # microseconds
r.render_rgba8()
ordered = r.counters()["kernel_ms"]
print(f"{a.scene}: {cost.shape[1]} x {cost.shape[0]} tiles; kernel {first:.3f} ms in natural order, {ordered:.3f} ms cost-ordered")
print(f"tile duration us: mean {cost.mean():.1f}  median {np.median(cost):.1f}  p99 {np.percentile(cost, 99):.1f}  max {cost.max():.1f}"
      f"  sum/(256 CUs x 16 waves) {cost.sum() / 4096:.1f}")
for idx in np.argsort(-cost.ravel())[:8]:
    y, x = divmod(int(idx), cost.shape[1])
    print(f"  tile ({x},{y}): {cost[y, x]:.1f} us")
grey = (cost / cost.max() * 255).astype(np.uint8)
os.makedirs(os.path.dirname(a.map), exist_ok=True)
rm.write_png(a.map, np.stack([grey, grey, grey, np.full_like(grey, 255)], axis=2))
r.close()
