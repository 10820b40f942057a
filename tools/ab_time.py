"""A/B timing of library builds in ONE process on ONE device (devices differ by >10 %, so numbers from
separate runs do not compare): python tools/ab_time.py libA.so libB.so ... [--scene default] [--rounds 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from raymarching_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--scene", default="default")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--exhaustive", action="store_true")
a = ap.parse_args()

renderers = []
for path in a.libs:
    binding._lib = None
    binding.LIB_PATH = os.path.abspath(path)
    L = binding.lib()
    r = binding.Renderer(0)
    kw = dict(kernel=1) if a.scene == "menger2" else {}
    r.set_params(binding.default_params(**kw))
    r.set_evaluator(not a.exhaustive)
    r.set_scene(binding.Scene(os.path.join(ROOT, "scenes", a.scene + ".txt")))
    renderers.append((path, L, r))

times = {p: [] for p, _, _ in renderers}
for rnd in range(a.rounds + 1):
    for path, L, r in renderers:
        binding._lib = L
        r.render_rgba8()
        if rnd:
            times[path].append(r.counters()["kernel_ms"])
for path, ms in times.items():
    ms.sort()
    print(f"{os.path.basename(path):28s} min {ms[0]:.3f}  median {ms[len(ms) // 2]:.3f}  ms  {[round(m, 2) for m in ms]}", flush=True)
