#!/usr/bin/env python3
"""Registers, scratch (spill) bytes, LDS and occupancy of the kernels generated for a scene, from the compiler's
own resource report (needs no GPU): the source rm_set_scene would generate is compiled with hipcc for gfx950
with the library's flags and -Rpass-analysis=kernel-resource-usage.

    python tools/kernel_resources.py [workload ...] [--counting] [--json FILE]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402
from tools.workloads import WORKLOADS  # noqa: E402

CSRC = os.path.join(ROOT, "raymarching_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage", "-c"]
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "static_lds_bytes",
          "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}


def resources(source):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene_kernels.hip")
        open(path, "w").write(source)
        out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", CSRC, path, "-o", os.path.join(d, "k.o")],
                             capture_output=True, text=True)
        if out.returncode:
            raise SystemExit(out.stderr[-3000:])
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.+?): (\d+)", line) or re.search(r"remark:\s+(.+?): (\d+)", line)
        if m and name and m.group(1).strip() in FIELDS:
            kernels[name][FIELDS[m.group(1).strip()]] = int(m.group(2))
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["default", "menger2", "closeup6", "synthetic"])
    ap.add_argument("--counting", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    report = {}
    for w in a.workloads:
        scene = rm.Scene(os.path.join(ROOT, "scenes", WORKLOADS[w][0]))
        src = rm.kernel_source(scene, counting=a.counting)
        report[w] = {"source_bytes": len(src), "kernels": resources(src)}
        for k, v in report[w]["kernels"].items():
            print(f"{w:12s} {k:22s} " + "  ".join(f"{f}={v.get(f)}" for f in FIELDS.values() if f in v))
    if a.json:
        json.dump(report, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
