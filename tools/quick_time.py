"""Quick device timing of the trace kernel on the two headline configurations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402

r = rm.Renderer(0)
for name, kw in (("menger2.txt", dict(kernel=1)), ("default.txt", dict())):
    p = rm.default_params(**kw)
    r.set_params(p)
    sc = rm.Scene(os.path.join(ROOT, "scenes", name))
    r.set_scene(sc)
    rays = p.width * p.height * p.kernel ** 2 * (1 + len(sc.lights()))
    for rep in range(3):
        r.render_rgba8()
        c = r.counters()
        print(f"{name} k={p.kernel}: trace {c['kernel_ms']:.3f} ms resolve {c['resolve_ms']:.3f} ms "
              f"-> {rays / c['kernel_ms'] / 1e3:.1f} Mrays/s", flush=True)
r.close()
