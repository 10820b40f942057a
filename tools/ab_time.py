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
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--kernel", type=int, default=0, help="SSAA kernel (with --width/--height: another frame size of the scene)")
ap.add_argument("--in-flight", type=int, default=0, help="also time the frame rate with this many contexts alternating frames")
a = ap.parse_args()

renderers = []
for path in a.libs:
    binding._lib = None
    binding.LIB_PATH = os.path.abspath(path)
    L = binding.lib()
    r = binding.Renderer(0)
    kw = dict(kernel=1) if a.scene == "menger2" else {}
    if a.width:
        kw.update(width=a.width, height=a.height)
    if a.kernel:
        kw.update(kernel=a.kernel)
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
if a.in_flight:
    import time
    for path, L, r in renderers:
        binding._lib = L
        rs = [r]
        for _ in range(a.in_flight - 1):
            q = binding.Renderer(0)
            q.set_params(r.params)
            q.set_evaluator(not a.exhaustive)
            q.set_scene(binding.Scene(os.path.join(ROOT, "scenes", a.scene + ".txt")))
            rs.append(q)
        rows = binding.Rows(r.params.height, 0, 1, 1)
        bufs = [torch.empty((r.params.height, r.params.width, 4), dtype=torch.uint8, device="cuda") for _ in rs]
        for q, b in zip(rs, bufs):
            for _ in range(3):
                q.render_device(rows, None, b.data_ptr())
        best = 1e9
        for rnd in range(a.rounds):
            n = 40
            t0 = time.perf_counter()
            for i in range(n):
                rs[i % len(rs)].enqueue_device(rows, None, bufs[i % len(rs)].data_ptr())
            for q in rs:
                q.wait()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        print(f"{os.path.basename(path):28s} {a.in_flight} frames in flight: {best:.4f} ms per frame", flush=True)
for path, ms in times.items():
    ms.sort()
    print(f"{os.path.basename(path):28s} min {ms[0]:.3f}  median {ms[len(ms) // 2]:.3f}  ms  {[round(m, 2) for m in ms]}", flush=True)
