"""Lane utilisation of the sphere-tracing loop (march_steps / march_slots) and work per frame of a workload."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import raymarching_amd as rm
from tools.workloads import WORKLOADS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

for name in sys.argv[1:] or ["default"]:
    scene_name, kw, _ = WORKLOADS[name]
    r = rm.Renderer(0)
    p = rm.default_params(**kw)
    r.set_params(p)
    r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", scene_name)))
    buf = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
    rows = rm.Rows(p.height, 0, 1, 1)
    for _ in range(3):
        r.render_device(rows, None, buf.data_ptr())
    ms = r.counters()["kernel_ms"]
    r.set_counting(True)
    r.render_device(rows, None, buf.data_ptr())
    c = r.counters()
    print(json.dumps({"workload": name, "kernel_ms": round(ms, 4), "march_util": round(c["march_steps"] / max(1, c["march_slots"]), 4),
                      "regions(lanes/slots)": {n: (round(c["probe"][2 * i] / max(1, c["probe"][2 * i + 1]), 3), c["probe"][2 * i + 1] // 64)
                                               for i, n in enumerate(["level0", "level1", "level2", "deeper", "row_cells", "grad_taps", "-", "scene_calls"])},
                      "clustered_runs_per_wave(hull tests, rows folded)": c["probe"][12:14],
                      **{k: c[k] for k in ("sdf_calls", "march_steps", "march_slots", "rays", "prim_evals", "flops_executed")}}))
    r.close()
