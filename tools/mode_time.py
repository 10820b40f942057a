"""Times the compiled and the interpreted kernels on the same scenes in ONE process (devices differ by
>10 %, so numbers from separate runs do not compare): python tools/mode_time.py [scene ...] [--rounds 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from raymarching_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="*", default=["default"])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--exhaustive", action="store_true")
ap.add_argument("--kernel", type=int, default=0)
a = ap.parse_args()

for scene in a.scenes:
    rs = []
    for compiled in (True, False):
        r = binding.Renderer(0)
        kw = dict(kernel=1) if scene == "menger2" else {}
        if a.kernel:
            kw["kernel"] = a.kernel
        r.set_params(binding.default_params(**kw))
        r.set_evaluator(not a.exhaustive)
        r.set_kernel_mode(compiled)
        t0 = time.time()
        r.set_scene(binding.Scene(os.path.join(ROOT, "scenes", scene + ".txt")))
        rs.append(("compiled" if compiled else "interpreted", r, time.time() - t0))
    frames = []
    times = {n: [] for n, _, _ in rs}
    for rnd in range(a.rounds + 1):
        for name, r, _ in rs:
            img = r.render_rgba8()
            if rnd == 0:
                frames.append(img)
            else:
                times[name].append(r.counters()["kernel_ms"])
    same = (frames[0] == frames[1]).all()
    for name, r, setup in rs:
        ms = sorted(times[name])
        c = r.counters()
        print(f"{scene:14s} {name:12s} min {ms[0]:9.3f}  median {ms[len(ms) // 2]:9.3f} ms   set_scene {setup:6.2f} s "
              f"(compile {c['compile_ms']:.0f} ms, ran compiled: {c['compiled_kernel']})  frames equal: {same}", flush=True)
