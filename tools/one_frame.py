"""Renders N frames of a workload (for profiler runs): python tools/one_frame.py [workload of tools/workloads.py] [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402
from tools.workloads import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "default"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene_file, kw, _ = WORKLOADS[name]
r = rm.Renderer(0)
p = rm.default_params(**kw)
r.set_params(p)
r.set_scene(rm.Scene(os.path.join(ROOT, "scenes", scene_file)))
for _ in range(frames):
    r.render_rgba8()
    print(r.counters(), flush=True)
r.close()
