"""Kernel time of the default frame against the march iteration cap (diagnostic: how much of the frame is
the tail of rays that never converge)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import raymarching_amd as rm  # noqa: E402

r = rm.Renderer(0)
sc = rm.Scene(os.path.join(ROOT, "scenes", "default.txt"))
for it in (1000, 500, 250, 125, 64, 32):
    r.set_params(rm.default_params(iterations=it))
    r.set_scene(sc)
    r.set_counting(True)
    r.render_rgba8()
    c = r.counters()
    r.set_counting(False)
    ms = []
    for _ in range(5):
        r.render_rgba8()
        ms.append(r.counters()["kernel_ms"])
    print(f"iterations {it:5d}: kernel {min(ms):.3f} ms  sdf_calls {c['sdf_calls']}  steps/ray {c['march_steps'] / c['rays']:.2f}", flush=True)
