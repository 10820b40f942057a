"""How sharply the reference's committed render (out_cpu.png) pins the oracle's arithmetic: renders the full shipped frame
with one operation at a time replaced by a plausible alternative form and counts the pixels that leave the golden image.
Writes tests/golden/variants.json (pixel counts and, per variant, a few rows that hold differing pixels, which
tests/test_oracle_golden.py re-renders).  About 15 minutes per variant on 8 cores:  python tests/golden/make_variants.py"""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import rmo  # noqa: E402

VARIANTS = {
    "normalize_by_division": dict(normalize_div=1),      # a / len instead of a * (1 / len)
    "cross_as_exact_median": dict(variant=1),             # med3 instead of dx + dy + dz - min - max
    "lerp_weighted": dict(variant=2),                     # (1 - t) u + t v instead of u + t (v - u)
    "dot_products_fused": dict(variant=4),                # what -ffp-contract=fast would compile
}

gold = np.array(Image.open(os.path.join(HERE, "out_cpu.png")))
scene = rmo.Scene(os.path.join(ROOT, "scenes", "default.txt"))
out = {}
for name, kw in VARIANTS.items():
    t0 = time.time()
    img, _ = scene.render(rmo.default_params(**kw), threads=int(os.environ.get("THREADS", "0")))
    bad = (rmo.quantize(img) != gold).any(axis=2)
    rows = np.flatnonzero(bad.any(axis=1))
    pick = rows[np.linspace(0, len(rows) - 1, min(4, len(rows))).astype(int)].tolist() if len(rows) else []
    out[name] = {"params": kw, "pixels_off": int(bad.sum()), "rows_with_pixels_off": int(len(rows)), "sample_rows": pick,
                 "pixels_off_in_sample_rows": [int(bad[r].sum()) for r in pick],
                 "max_error": int(np.abs(rmo.quantize(img).astype(int) - gold.astype(int)).max())}
    print(name, out[name], f"{time.time() - t0:.0f} s", flush=True)
    json.dump(out, open(os.path.join(HERE, "variants.json"), "w"), indent=1)
