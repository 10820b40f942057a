"""In-process comparison of library builds on the critical path: the trace-kernel time of one rank's share of the default
frame at fine splits (N = 16, 64), where the longest-running tiles decide.  python tools/ab_split.py libA.so libB.so"""
import os, sys
ROOT="/root/repo" if os.path.isdir("/root/repo/raymarching_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
import torch
from raymarching_amd import binding, sharding
res={}
for path in sys.argv[1:]:
    binding._lib=None; binding.LIB_PATH=os.path.abspath(path); L=binding.lib()
    r=binding.Renderer(0); p=binding.default_params(); r.set_params(p)
    r.set_scene(binding.Scene(os.path.join(ROOT,"scenes","default.txt")))
    buf=torch.empty((p.height+64,p.width,4),dtype=torch.uint8,device="cuda")
    out=[]
    for world in (16,64):
        ts=[]
        for rank in range(4):
            rows,_=sharding.band_plan(p.height,2,world,rank)
            r.render_device(rows,None,buf.data_ptr())
            ms=[]
            for _ in range(6):
                r.render_device(rows,None,buf.data_ptr()); ms.append(r.counters()["kernel_ms"])
            ts.append(min(ms))
        out.append(round(sum(ts)/len(ts),4))
    print(os.path.basename(path),"share kernel ms at N=16,64:",out,flush=True)
