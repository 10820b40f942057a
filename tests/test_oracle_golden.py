"""The oracle against the reference's own committed CPU render (tests/golden/out_cpu.png =
/root/reference/out_cpu.png: shipped scene, 1024x768, SSAA 3x3).  The full frame was checked once
(tests/golden/make_golden.py -> default_1024x768_k3.json: 786432 of 786432 pixels exact); here a
row subset keeps the CPU suite short."""
import json
import os

import numpy as np

from conftest import GOLDEN, scene_path


def gold_png():
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, "out_cpu.png")))


def test_golden_png_facts():
    g = gold_png()
    assert g.shape == (768, 1024, 4) and (g[..., 3] == 255).all()
    assert abs(float((g[..., :3].max(axis=2) == 0).mean()) - 0.6240) < 5e-4   # sky = black bounds box
    for (x, y), rgb in {(0, 0): (0, 0, 0), (512, 384): (178, 178, 178), (240, 450): (255, 0, 0),
                        (370, 580): (0, 0, 187), (573, 608): (0, 246, 0), (230, 530): (13, 13, 13)}.items():
        assert tuple(g[y, x, :3]) == rgb


def test_oracle_rows_equal_reference_png(oracle):
    g = gold_png()
    sc = oracle.Scene(scene_path("default.txt"))
    rows = (20, 768, 68)            # 11 rows across sky, spheres, sponge and floor
    img, _ = sc.render(oracle.default_params(), rows=rows)
    sel = np.arange(*rows)
    np.testing.assert_array_equal(oracle.quantize(img)[sel], g[sel])


def test_oracle_sample_pixels(oracle):
    g = gold_png()
    sc = oracle.Scene(scene_path("default.txt"))
    p = oracle.default_params()
    for x, y in [(495, 400), (600, 600), (100, 500), (500, 300), (850, 500), (573, 608), (370, 580)]:
        s = sc.pixel_samples(p, x, y)
        total = np.zeros(3, np.float32)
        for c in s:                       # render.cpp:114: total += color, in loop order
            total = (total + c).astype(np.float32)
        px = np.append(total / np.float32(9), np.float32(1)).astype(np.float32)
        assert oracle.quantize(px[None, :])[0].tolist() == g[y, x].tolist()


def test_full_frame_record_is_exact():
    rec = json.load(open(os.path.join(GOLDEN, "default_1024x768_k3.json")))
    assert rec["vs_out_cpu_png"] == {"pixels_exact": 786432, "pixels": 786432, "max_error": 0}
    assert rec["counters"]["rays"] == 1024 * 768 * 9 * 3
    import hashlib
    assert hashlib.sha256(gold_png().tobytes()).hexdigest() == rec["sha256_rgba8"]


def test_the_golden_render_tells_the_oracle_from_its_plausible_variants(oracle):
    """The pin discriminates: with one operation replaced by another plausible form -- normalize by division, the cross as an exact
    median, lerp as a weighted sum, fused dot products -- the oracle leaves the reference's committed render (tests/golden/variants.json
    holds the full-frame counts, from tests/golden/make_variants.py: 61, 14, 128 and 143 of 786 432 pixels), while the forms the oracle
    uses reproduce every pixel.  Re-rendered here: for each variant a few of the rows that hold its differing pixels."""
    rec = json.load(open(os.path.join(GOLDEN, "variants.json")))
    assert set(rec) == {"normalize_by_division", "cross_as_exact_median", "lerp_weighted", "dot_products_fused"}
    g = gold_png()
    sc = oracle.Scene(scene_path("default.txt"))
    for name, v in rec.items():
        assert v["pixels_off"] > 0, name
        rows = v["sample_rows"][:2]
        xy = [(x, y) for y in rows for x in range(1024)]
        base = oracle.quantize(sc.render_pixels(oracle.default_params(), xy)).reshape(len(rows), 1024, 4)
        np.testing.assert_array_equal(base, g[rows])
        wrong = oracle.quantize(sc.render_pixels(oracle.default_params(**v["params"]), xy)).reshape(len(rows), 1024, 4)
        off = [int((wrong[i] != g[r]).any(axis=1).sum()) for i, r in enumerate(rows)]
        assert off == v["pixels_off_in_sample_rows"][:2], (name, off)
