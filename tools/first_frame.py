"""Cold-start cost of one process: python tools/first_frame.py <workload> <cache dir or "off">
Prints one JSON line: library load, context, scene upload (kernel compile or cache hit), and the first frame into a
host image -- what the reference's main() times as "Copy to GPU" and "Render with GPU" (main.cpp:62-73) -- then the
second and third frames.  Run twice with the same directory to see the warm-cache figures."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
workload, cache = sys.argv[1], sys.argv[2]
os.environ["RM_KERNEL_CACHE_DIR"] = cache
t0 = time.perf_counter()
import raymarching_amd as rm  # noqa: E402
from tools.workloads import WORKLOADS  # noqa: E402
scene_name, kw, _ = WORKLOADS[workload]
rm.lib()
t1 = time.perf_counter()
scene = rm.Scene(os.path.join(ROOT, "scenes", scene_name))
rm.precompile(scene)     # the kernels start compiling while the context is made (as the render:: adapter does it)
r = rm.Renderer(0)
p = rm.default_params(**kw)
r.set_params(p)
t2 = time.perf_counter()
r.set_scene(scene)
t3 = time.perf_counter()
r.reserve()
t4 = time.perf_counter()
img = rm.HostBuffer((p.height, p.width, 4), "uint8")
frames = []
for _ in range(3):
    t = time.perf_counter()
    rc = rm.lib().rm_render_rgba8(r._h, img.ptr)
    frames.append((time.perf_counter() - t) * 1e3)
    assert rc == 0
c = r.counters()
print(json.dumps({"import_and_load_s": round(t1 - t0, 3), "context_and_parse_ms": round((t2 - t1) * 1e3, 2),
                  "set_scene_ms": round((t3 - t2) * 1e3, 2), "reserve_ms": round((t4 - t3) * 1e3, 2), "kernel_compile_ms": round(c["compile_ms"], 1),
                  "first_frame_ms": round(frames[0], 3), "second_frame_ms": round(frames[1], 3), "third_frame_ms": round(frames[2], 3),
                  "third_frame_in_cost_order": c["tiles_by_cost"]}))
r.close()
img.free()
