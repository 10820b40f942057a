"""Times several scene files with one library in one process: python tools/scene_time.py a.txt b.txt ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402

r = rm.Renderer(0)
r.set_params(rm.default_params())
for path in sys.argv[1:]:
    sc = rm.Scene(path)
    r.set_scene(sc)
    r.set_counting(True)
    r.render_rgba8()
    c = r.counters()
    r.set_counting(False)
    ms = []
    for _ in range(5):
        r.render_rgba8()
        ms.append(r.counters()["kernel_ms"])
    print(f"{os.path.basename(path):20s} min {min(ms):.3f} ms  sdf_calls {c['sdf_calls']}  evals/call {c['prim_evals'] / max(1, c['sdf_calls']):.2f}"
          f"  ns/1000 calls {min(ms) * 1e6 / c['sdf_calls'] * 1000:.1f}  march lane utilisation {c['march_steps'] / max(1, c['march_slots']):.3f}"
          f"  steps/ray {c['march_steps'] / max(1, c['rays']):.1f}", flush=True)
r.close()
