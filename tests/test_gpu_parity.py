"""Parity of the HIP render path against the CPU oracle, through the C ABI.

Bar: the float framebuffer is BIT-IDENTICAL to the oracle's (the kernels keep the
reference's fp32 operation order with no fused multiply-add), and against the
reference's committed render out_cpu.png at least 99.9 % of the pixels are within
1/255 per 8-bit channel.
"""
import os

import numpy as np
import pytest

from conftest import same_but_zero_signs, GOLDEN, scene_path

pytestmark = pytest.mark.gpu

LIST, SPHERE, BOX, CROSS = 0, 1, 2, 3
UNION, COMPLEMENT, INTERSECTION, DIFFERENCE = 0, 1, 2, 3


def both_params(rm, oracle, **kw):
    return rm.default_params(**kw), oracle.default_params(**kw)


def assert_bit_identical(gpu, cpu):
    gpu = np.ascontiguousarray(gpu, np.float32)
    cpu = np.ascontiguousarray(cpu, np.float32)
    if gpu.tobytes() == cpu.tobytes():
        return
    diff = np.abs(gpu.astype(np.float64) - cpu.astype(np.float64))
    bad = np.argwhere(gpu.view(np.uint32) != cpu.view(np.uint32))
    raise AssertionError(f"{len(bad)} of {gpu.size} floats differ; max abs diff {np.nanmax(diff):.3g}; "
                         f"first at {bad[0].tolist()}: gpu {gpu[tuple(bad[0])]!r} cpu {cpu[tuple(bad[0])]!r}")


def random_points(rng, n):
    pts = rng.uniform([-60, -5, -110], [60, 45, 10], (n, 3)).astype(np.float32)
    # a share of the points right at the sponge / floor / spheres, where ties and sign changes live
    pts[: n // 4] = rng.uniform([-10.5, -0.5, -60.5], [10.5, 20.5, -39.5], (n // 4, 3)).astype(np.float32)
    return pts


@pytest.mark.parametrize("name", ["default.txt", "menger2.txt"])
def test_scene_sdf_bitwise(rm, oracle, renderer, name):
    prod, orc = rm.Scene(scene_path(name)), oracle.Scene(scene_path(name))
    renderer.set_scene(prod)
    pts = random_points(np.random.default_rng(1), 20000)
    sd, prim, sd_march = renderer.eval_sdf_both(pts)
    assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
    osd, orgb = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    colours = np.array([tuple(p.color) for p in orc.flat_prims()], np.float32)
    np.testing.assert_array_equal(colours[prim], orgb)


def _mk(mod_node, spec):
    import ctypes as C
    f3 = C.c_float * 3
    return [mod_node(t, m, n, f3(*c), f3(*s), f3(*col)) for (t, m, n, c, s, col) in spec]


def _random_tree(rng, depth, max_children):
    """Random CSG tree in pre-order, all four list modes, anisotropic boxes/crosses, nesting up to depth."""
    spec = []

    def prim():
        t = int(rng.integers(1, 4))
        c = rng.uniform(-8, 8, 3)
        if t == SPHERE:
            s = (rng.uniform(0.5, 4), 0, 0)
        elif rng.random() < 0.5:
            e = rng.uniform(0.5, 6)
            s = (e, e, e)
        else:
            s = rng.uniform(0.5, 6, 3)
        spec.append((t, 0, 0, c, s, rng.uniform(0, 1, 3)))

    def lst(d):
        n = int(rng.integers(0 if d > 0 else 1, max_children + 1))
        at = len(spec)
        spec.append((LIST, int(rng.integers(0, 4)), n, (0, 0, 0), (0, 0, 0), (0, 0, 0)))
        for _ in range(n):
            if d < depth and rng.random() < 0.3:
                lst(d + 1)
            else:
                prim()
        return at

    lst(0)
    return spec


@pytest.mark.parametrize("seed", range(6))
def test_random_trees_all_modes_bitwise(rm, oracle, renderer, seed):
    rng = np.random.default_rng(100 + seed)
    spec = _random_tree(rng, depth=3 if seed % 2 else 1, max_children=9)
    pnodes, onodes = _mk(rm.Node, spec), _mk(oracle.Node, spec)
    orc = oracle.Scene(nodes=onodes)
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    renderer.set_scene_raw(pnodes, np.zeros((0, 3), np.float32), ident, 1.0)
    pts = rng.uniform(-12, 12, (8000, 3)).astype(np.float32)
    sd, prim, sd_march = renderer.eval_sdf_both(pts)
    assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
    osd, orgb = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    colours = np.array([tuple(p.color) for p in orc.flat_prims()] + [(0, 0, 0)], np.float32)
    idx = np.where(prim == 0xFFFFFFFF, len(colours) - 1, prim)
    np.testing.assert_array_equal(colours[idx], orgb)


@pytest.mark.parametrize("kernel,width,height", [(1, 256, 192), (2, 100, 75), (3, 64, 48)])
def test_small_frames_bit_identical(rm, oracle, renderer, kernel, width, height):
    """configs[0] (Menger iterations 2) at reduced sizes, ragged (non multiple-of-16) included."""
    pp, op = both_params(rm, oracle, width=width, height=height, kernel=kernel)
    renderer.set_params(pp)
    renderer.set_scene(rm.Scene(scene_path("menger2.txt")))
    gpu = renderer.render_f32()
    cpu, _ = oracle.Scene(scene_path("menger2.txt")).render(op)
    assert_bit_identical(gpu, cpu)
    np.testing.assert_array_equal(renderer.render_rgba8(), oracle.quantize(cpu))


def test_default_scene_rows_bit_identical(rm, oracle, renderer):
    """configs[1] (shipped scene, 1024x768, SSAA 3): every 64th row against the oracle."""
    pp, op = both_params(rm, oracle)
    renderer.set_params(pp)
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    gpu = renderer.render_f32()
    rows = (8, 768, 64)
    cpu, _ = oracle.Scene(scene_path("default.txt")).render(op, rows=rows)
    sel = np.arange(*rows)
    assert_bit_identical(gpu[sel], cpu[sel])


def test_default_scene_full_frame_hash(rm, renderer):
    """configs[1] at full size: the float frame hashes to the oracle's full-frame hash (the oracle's
    frame is itself pixel-exact against the reference's out_cpu.png, see make_golden.py)."""
    import hashlib
    import json
    rec = json.load(open(os.path.join(GOLDEN, "default_1024x768_k3.json")))
    renderer.set_params(rm.default_params())
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    renderer.set_counting(True)
    renderer.set_shading(False)                  # every sample does the reference's full work: counts compare
    try:
        img = renderer.render_f32()
        c = renderer.counters()
        renderer.set_shading(True)               # the default: colourless samples skip normal and shadow rays
        img_default = renderer.render_f32()
        c_default = renderer.counters()
    finally:
        renderer.set_counting(False)
        renderer.set_shading(True)
    assert hashlib.sha256(img.tobytes()).hexdigest() == rec["sha256_f32"]
    assert hashlib.sha256(img_default.tobytes()).hexdigest() == rec["sha256_f32"]
    assert hashlib.sha256(renderer.render_rgba8().tobytes()).hexdigest() == rec["sha256_rgba8"]
    for key in ("sdf_calls", "march_steps", "rays"):
        assert c[key] == rec["counters"][key]
    # 62.4 % of the frame is the colourless wall: those samples cast no shadow rays
    assert c_default["rays"] < 0.6 * c["rays"] and c_default["sdf_calls"] < c["sdf_calls"]


def test_default_scene_against_reference_png(rm, renderer):
    """The reference's own CPU render: >= 99.9 % of pixels within 1/255 per channel (tolerance of the task)."""
    from PIL import Image
    gold = np.array(Image.open(os.path.join(GOLDEN, "out_cpu.png")))
    renderer.set_params(rm.default_params())
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    img = renderer.render_rgba8()
    assert img.shape == gold.shape == (768, 1024, 4)
    err = np.abs(img.astype(np.int32) - gold.astype(np.int32)).max(axis=2)
    within1 = float((err <= 1).mean())
    assert within1 >= 0.999, f"only {within1:.5f} of pixels within 1/255 (max error {err.max()})"
    assert (img[..., 3] == 255).all()


def test_counters_match_oracle(rm, oracle, renderer):
    pp, op = both_params(rm, oracle, width=96, height=72, kernel=2)
    renderer.set_params(pp)
    renderer.set_scene(rm.Scene(scene_path("menger2.txt")))
    renderer.set_counting(True)
    renderer.set_shading(False)
    try:
        renderer.render_f32()
        c = renderer.counters()
    finally:
        renderer.set_counting(False)
        renderer.set_shading(True)
    _, oc = oracle.Scene(scene_path("menger2.txt")).render(op)
    assert c["rays"] == oc["rays"] == 96 * 72 * 4 * 3
    assert c["march_steps"] == oc["march_steps"]
    assert c["sdf_calls"] == oc["sdf_calls"]
    assert c["prims"] == 28


def test_row_bands_equal_whole_frame(rm, renderer):
    """Sharding by row bands (what multi-GPU runs do) reproduces the one-launch frame bit for bit."""
    import torch
    p = rm.default_params(width=160, height=100, kernel=2)
    renderer.set_params(p)
    renderer.set_scene(rm.Scene(scene_path("menger2.txt")))
    whole = renderer.render_f32()
    band_rows, shards = 8, 3
    n_bands = (p.height + band_rows - 1) // band_rows
    frame = np.zeros_like(whole)
    for rank in range(shards):
        count = len(range(rank, n_bands, shards))
        buf = torch.zeros((count * band_rows, p.width, 4), dtype=torch.float32, device="cuda")
        renderer.render_device(rm.Rows(band_rows, rank, shards, count), buf.data_ptr(), None)
        host = buf.cpu().numpy()
        for i, band in enumerate(range(rank, n_bands, shards)):
            r0 = band * band_rows
            r1 = min(r0 + band_rows, p.height)
            frame[r0:r1] = host[i * band_rows: i * band_rows + (r1 - r0)]
    assert frame.tobytes() == whole.tobytes()


def test_empty_scene_and_zero_lights(rm, oracle, renderer):
    pp, op = both_params(rm, oracle, width=32, height=16, kernel=1, iterations=20)
    renderer.set_params(pp)
    for text in ("", "Bounds 50\nSphere 0 0 -10 2\n"):
        renderer.set_scene(rm.Scene(text=text))
        gpu = renderer.render_f32()
        cpu, _ = oracle.Scene(text=text).render(op)
        assert_bit_identical(gpu, cpu)


def test_errors_are_reported(rm):
    r = rm.Renderer(0)
    try:
        with pytest.raises(rm.RaymarchError, match="rm_set_params"):
            r.render_f32() if r.params else r.set_params(rm.default_params(width=0))
        r.set_params(rm.default_params(width=8, height=8))
        with pytest.raises(rm.RaymarchError, match="rm_set_scene has not been called"):
            r.render_f32()
        bad = _mk(rm.Node, [(LIST, 0, 3, (0, 0, 0), (0, 0, 0), (0, 0, 0)), (SPHERE, 0, 0, (0, 0, 0), (1, 0, 0), (1, 1, 1))])
        with pytest.raises(rm.RaymarchError, match="child_count"):
            r.set_scene_raw(bad, np.zeros((0, 3)), np.eye(4).reshape(-1), 1.0)
    finally:
        r.close()
    with pytest.raises(rm.RaymarchError, match="no HIP device"):
        rm.Renderer(99)


@pytest.mark.parametrize("copies", [1, 5])
@pytest.mark.parametrize("mode", [UNION, INTERSECTION, COMPLEMENT, DIFFERENCE])
def test_ties_between_kinds_keep_the_earlier_child(rm, oracle, renderer, mode, copies):
    """Primitives of one list are regrouped per kind before they are folded; of two children at exactly
    the same distance the earlier one (in the list's order) must still win, as in the sequential fold.
    copies = 1 exercises single records, copies = 5 the table RUNs."""
    spec = [(LIST, mode, 0, (0, 0, 0), (0, 0, 0), (0, 0, 0)),
            (SPHERE, 0, 0, (0, 40, 0), (1, 0, 0), (0.1, 0.1, 0.1))]       # first child, far away
    pts = []
    for c in range(copies):
        z = 10.0 * c
        # box (half 1) at x=+3 comes BEFORE the sphere (r=1) at x=-3 and the cross at y=+3: all are at
        # distance 2 from (0,0,z); spheres are folded before boxes after regrouping
        spec.append((BOX, 0, 0, (3, 0, z), (2, 2, 2), (1, 0, 0)))
        spec.append((SPHERE, 0, 0, (-3, 0, z), (1, 0, 0), (0, 1, 0)))
        spec.append((CROSS, 0, 0, (0, 3, z + 0.0), (2, 2, 2), (0, 0, 1)))
        pts.append((0, 0, z))
    spec[0] = (LIST, mode, len(spec) - 1, (0, 0, 0), (0, 0, 0), (0, 0, 0))
    pnodes, onodes = _mk(rm.Node, spec), _mk(oracle.Node, spec)
    renderer.set_scene_raw(pnodes, np.zeros((0, 3), np.float32), np.eye(4, dtype=np.float32).reshape(-1), 1.0)
    pts = np.array(pts, np.float32)
    sd, prim, sd_march = renderer.eval_sdf_both(pts)
    assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
    orc = oracle.Scene(nodes=onodes)
    osd, orgb = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    colours = np.array([tuple(p.color) for p in orc.flat_prims()], np.float32)
    np.testing.assert_array_equal(colours[prim], orgb)


@pytest.mark.parametrize("size", [dict(width=320, height=200, kernel=2), dict(width=163, height=99, kernel=3)], ids=["320x200k2", "163x99k3"])
def test_cost_ordered_tiles_render_the_same_frame(rm, renderer, size):
    """The second and later renders of a configuration start expensive tiles first: same bytes, flag set
    (the second size has ragged tiles at the right and bottom edges)."""
    renderer.set_params(rm.default_params(**size))
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    first = renderer.render_f32()
    assert renderer.counters()["tiles_by_cost"] == 0          # the measuring frame runs in natural order
    for _ in range(2):
        again = renderer.render_f32()
        c = renderer.counters()
        assert c["tiles_by_cost"] == 1
        # the sponge's longest tiles run as quarters or sixteenths (kernels of one wave per workgroup take such entries)
        assert (c["tile_parts"] > 0) == (renderer.kernel_mode == "compiled")
        assert again.tobytes() == first.tobytes()
    renderer.set_tile_scheduling(False)
    try:
        natural = renderer.render_f32()
        assert renderer.counters()["tiles_by_cost"] == 0 and renderer.counters()["tile_parts"] == 0
        assert natural.tobytes() == first.tobytes()
    finally:
        renderer.set_tile_scheduling(True)


def test_queued_frames_on_caller_streams_equal_synchronous_render(rm):
    """rm_enqueue_device / rm_wait / rm_set_stream: frames of two contexts in flight on two torch streams."""
    import torch
    p = rm.default_params(width=256, height=160, kernel=2)
    scene = rm.Scene(scene_path("default.txt"))
    rows = rm.binding.Rows(160, 0, 1, 1)
    ctxs, streams, bufs = [], [], []
    try:
        for _ in range(2):
            r = rm.Renderer(0)
            r.set_params(p)
            r.set_scene(scene)
            ctxs.append(r)
            streams.append(torch.cuda.Stream())
            bufs.append(torch.zeros((160, 256, 4), dtype=torch.float32, device="cuda"))
        ref = ctxs[0].render_f32()                                 # synchronous; also times the tiles
        ctxs[1].render_device(rows, bufs[1].data_ptr(), None)
        for r, st in zip(ctxs, streams):
            r.set_stream(st.cuda_stream)
        for rep in range(3):
            for r, st, b in zip(ctxs, streams, bufs):
                with torch.cuda.stream(st):
                    b.zero_()                                      # ordered before the kernels: same stream
                    r.enqueue_device(rows, b.data_ptr(), None)
        for r in ctxs:
            r.wait()
            c = r.counters()
            assert c["kernel_ms"] > 0 and c["compiled_kernel"] == 1 and c["tiles_by_cost"] == 1
        for b in bufs:
            assert np.array_equal(b.cpu().numpy(), ref)
        for r in ctxs:
            r.set_stream(None)
        assert np.array_equal(ctxs[0].render_f32(), ref)
    finally:
        for r in ctxs:
            r.close()


def test_band_list_renders_the_listed_bands(rm, renderer):
    """rm_set_band_list: an arbitrary deal of image bands lands, in list order, in the compact buffer."""
    import torch
    p = rm.default_params(width=160, height=90, kernel=2)      # 90 rows: the last band of 4 is ragged
    renderer.set_params(p)
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    whole = renderer.render_f32()
    bands = [22, 3, 17, 0, 9, 23]                              # 23 lies partly, 24+ wholly outside the image
    buf = torch.full((len(bands) * 4, 160, 4), -1.0, dtype=torch.float32, device="cuda")
    rows = rm.binding.Rows(4, 0, 1, len(bands))
    renderer.set_band_list(bands)
    try:
        for _ in range(2):                                     # natural tile order, then cost order
            renderer.render_device(rows, buf.data_ptr(), None)
            got = buf.cpu().numpy()
            for i, b in enumerate(bands):
                r0, r1 = b * 4, min(b * 4 + 4, 90)
                assert np.array_equal(got[i * 4: i * 4 + (r1 - r0)], whole[r0:r1]), f"band {b}"
        with pytest.raises(rm.RaymarchError, match="band list"):
            renderer.render_device(rm.binding.Rows(4, 0, 1, 3), buf.data_ptr(), None)
    finally:
        renderer.set_band_list(None)
    assert np.array_equal(renderer.render_f32(), whole)
