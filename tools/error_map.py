"""Parity report of a GPU frame against a reference image: share of pixels within 1/255 per channel, the maximum
error, and the max-error map written as a PNG (per pixel: the largest channel difference, scaled so that the
frame's maximum error, at least 1/255, is white).

    python tools/error_map.py [--scene scenes/default.txt] [--reference tests/golden/out_cpu.png]
                              [--map gpurun_out/error_map.png] [--json gpurun_out/parity.json]

The default arguments compare the shipped configuration with the reference's own CPU render (out_cpu.png)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import raymarching_amd as rm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "default.txt"))
ap.add_argument("--reference", default=os.path.join(ROOT, "tests", "golden", "out_cpu.png"))
ap.add_argument("--map", default=os.path.join(ROOT, "gpurun_out", "error_map.png"))
ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "parity.json"))
a = ap.parse_args()

from PIL import Image  # noqa: E402

gold = np.array(Image.open(a.reference).convert("RGBA"))
h, w = gold.shape[:2]
r = rm.Renderer(0)
r.set_params(rm.default_params(width=w, height=h))
r.set_scene(rm.Scene(a.scene))
img = r.render_rgba8()
r.close()
err = np.abs(img.astype(np.int32) - gold.astype(np.int32)).max(axis=2)
scale = max(1, int(err.max()))
grey = (err * 255 // scale).astype(np.uint8)
out = np.stack([grey, grey, grey, np.full_like(grey, 255)], axis=2)
os.makedirs(os.path.dirname(a.map), exist_ok=True)
rm.write_png(a.map, out)
report = {
    "scene": os.path.relpath(a.scene, ROOT), "reference": os.path.relpath(a.reference, ROOT), "width": w, "height": h,
    "pixels": int(err.size), "pixels_exact": int((err == 0).sum()), "pixels_within_1_of_255": int((err <= 1).sum()),
    "share_within_1_of_255": float((err <= 1).mean()), "max_error_of_255": int(err.max()),
    "black_pixel_share": float((gold[..., :3].max(axis=2) == 0).mean()), "map": os.path.relpath(a.map, ROOT),
}
json.dump(report, open(a.json, "w"), indent=1)
print(json.dumps(report))
