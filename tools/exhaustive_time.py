"""Times the reference's algorithm as it stands (every primitive at every SDF call) on the record-walking kernel, default scene."""
import os, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import torch
import raymarching_amd as rm
r=rm.Renderer(0); r.set_params(rm.default_params()); r.set_evaluator(False); r.set_kernel_mode(False)
r.set_scene(rm.Scene(os.path.join(ROOT,"scenes","default.txt")))
r.set_counting(True); r.render_rgba8(); c=r.counters(); r.set_counting(False)
ms=[]
for _ in range(5):
    r.render_rgba8(); ms.append(r.counters()["kernel_ms"])
fl=c["sdf_calls"]*c["flops_per_sdf"]+7*c["march_steps"]
print("exhaustive record walker default:", ms, "TFLOP/s", fl/min(ms)/1e9, "frac", fl/min(ms)/1e9/157.3)
