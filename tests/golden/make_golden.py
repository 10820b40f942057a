"""Regenerates tests/golden/default_1024x768_k3.json: hashes and work counters of the ORACLE's
full frame of the shipped scene (scenes/default.txt, 1024x768, SSAA 3 -- the configuration of the
reference's committed out_cpu.png).  Takes ~10-15 minutes of CPU; run from the repo root:

    python tests/golden/make_golden.py [threads]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rmo  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    scene = rmo.Scene(os.path.join(ROOT, "scenes", "default.txt"))
    params = rmo.default_params()
    t0 = time.time()
    img, counters = scene.render(params, threads=threads)
    seconds = time.time() - t0
    q = rmo.quantize(img)
    out = {
        "scene": "scenes/default.txt", "width": params.width, "height": params.height, "kernel": params.kernel,
        "sha256_f32": hashlib.sha256(img.tobytes()).hexdigest(),
        "sha256_rgba8": hashlib.sha256(q.tobytes()).hexdigest(),
        "counters": counters,
        "oracle_seconds": round(seconds, 1), "oracle_threads": threads or rmo.max_threads(),
    }
    try:
        from PIL import Image
        gold = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "out_cpu.png")))
        err = np.abs(q.astype(np.int32) - gold.astype(np.int32)).max(axis=2)
        out["vs_out_cpu_png"] = {"pixels_exact": int((err == 0).sum()), "pixels": int(err.size), "max_error": int(err.max())}
    except ImportError:
        pass
    with open(os.path.join(ROOT, "tests", "golden", "default_1024x768_k3.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
