"""Trace-kernel time and work counts with and without the colourless-surface shortcut (rm_set_shading), in one
process: python tools/shading_time.py [--scene default] [--rounds 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from raymarching_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="default")
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()

rs = {}
for coloured_only in (False, True):
    r = binding.Renderer(0)
    r.set_params(binding.default_params(**(dict(kernel=1) if a.scene == "menger2" else {})))
    r.set_shading(coloured_only)
    r.set_scene(binding.Scene(os.path.join(ROOT, "scenes", a.scene + ".txt")))
    rs[coloured_only] = r
times = {k: [] for k in rs}
for rnd in range(a.rounds + 2):
    for k, r in rs.items():
        r.render_rgba8()
        if rnd >= 2:
            times[k].append(r.counters()["kernel_ms"])
for k, r in rs.items():
    r.set_counting(True)
    r.render_rgba8()
    c = r.counters()
    ms = sorted(times[k])
    print(f"{'coloured only' if k else 'every sample '}: min {ms[0]:.3f} median {ms[len(ms) // 2]:.3f} ms; rays {c['rays']} "
          f"sdf_calls {c['sdf_calls']} march_steps {c['march_steps']} flops_executed {c['flops_executed']}", flush=True)
