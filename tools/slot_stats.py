#!/usr/bin/env python3
"""Which workgroups a frame waits for: per-slot durations, start times and march lengths of a cost-ordered frame
(rm_set_slot_stats).  python tools/slot_stats.py [workload] [--share N] [--top 24]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import raymarching_amd as rm  # noqa: E402
from raymarching_amd import sharding  # noqa: E402
from tools.workloads import WORKLOADS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="default")
ap.add_argument("--share", type=int, default=1, help="rank 0's share of an N-way band split")
ap.add_argument("--top", type=int, default=24)
a = ap.parse_args()
name, kw, _ = WORKLOADS[a.workload]
p = rm.default_params(**kw)
r = rm.Renderer(0)
r.set_params(p)
r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", name)))
rows, local = sharding.band_plan(p.height, 4, a.share, 0)
out = torch.zeros((local, p.width, 4), dtype=torch.uint8, device="cuda")
for _ in range(4):
    r.render_device(rows, None, out.data_ptr())
r.set_slot_stats(True)
best = None
for _ in range(5):
    r.render_device(rows, None, out.data_ptr())
    ms = r.counters()["kernel_ms"]
    st = r.slot_stats()
    if best is None or ms < best[0]:
        best = (ms, st)
ms, st = best
entry, ticks, turns, start = st[:, 0], st[:, 1].astype(np.int64), st[:, 2], st[:, 3].astype(np.int64)
t0 = start.min()
begin = (start - t0) * 0.01           # us
end = begin + ticks * 0.01
kind = np.where(entry >> 27 == 0, "tile", np.where(entry >> 27 < 5, "quarter", "sixteenth"))
print(f"{a.workload} share 1/{a.share}: kernel {ms * 1e3:.1f} us, {len(st)} slots, last end {end.max():.1f} us; "
      f"slots: {np.sum(kind == 'tile')} tiles, {np.sum(kind == 'quarter')} quarters, {np.sum(kind == 'sixteenth')} sixteenths")
order = np.argsort(-end)[:a.top]
print("the slots that end last:  slot kind  begin_us  end_us  dur_us  camera_turns  shadow_turns  us/turn")
for i in order:
    cam, shd = int(turns[i] & 0xffff), int(turns[i] >> 16)
    print(f"  {i:7d} {kind[i]:9s} {begin[i]:8.1f} {end[i]:8.1f} {ticks[i] * 0.01:7.1f} {cam:6d} {shd:6d}   {ticks[i] * 0.01 / max(1, cam + shd):.3f}")
for k in ("sixteenth", "quarter", "tile"):
    m = kind == k
    if m.any():
        d = ticks[m] * 0.01
        c = (turns[m] & 0xffff) + (turns[m] >> 16)
        print(f"{k:9s}: n {m.sum():6d}  duration us median {np.median(d):7.1f} p99 {np.percentile(d, 99):7.1f} max {d.max():7.1f}; "
              f"us per march turn (slots over 100 turns) {np.median((d / np.maximum(1, c))[c > 100]) if (c > 100).any() else float('nan'):.3f}")
