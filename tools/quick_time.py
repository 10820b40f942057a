"""Quick device timing of the trace kernel: both evaluators on the headline configurations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402

which = sys.argv[1:] or ["menger2", "default"]
r = rm.Renderer(0)
for name in which:
    kw = dict(kernel=1) if name == "menger2" else {}
    p = rm.default_params(**kw)
    r.set_params(p)
    sc = rm.Scene(os.path.join(ROOT, "scenes", name + ".txt"))
    rays = p.width * p.height * p.kernel ** 2 * (1 + len(sc.lights()))
    for lattice in (True, False):
        r.set_evaluator(lattice)
        r.set_scene(sc)
        r.set_counting(True)
        r.render_rgba8()
        c = r.counters()
        r.set_counting(False)
        ms = []
        for rep in range(4):
            r.render_rgba8()
            ms.append(r.counters()["kernel_ms"])
        best = min(ms)
        print(f"{name} k={p.kernel} {'lattice' if lattice else 'exhaustive'}: trace {best:.3f} ms (runs {[round(m, 2) for m in ms]}) "
              f"-> {rays / best / 1e3:.1f} Mrays/s; sdf_calls {c['sdf_calls']} prim_evals {c['prim_evals']} "
              f"({c['prim_evals'] / max(1, c['sdf_calls']):.1f}/call)", flush=True)
r.set_evaluator(True)
r.close()
