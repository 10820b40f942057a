"""The LATTICE evaluator (exact, sub-linear evaluation of recognised Menger lattices) against the
brute-force oracle and against the library's own exhaustive evaluator: results must be bit-identical."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import same_but_zero_signs, GOLDEN, scene_path

pytestmark = pytest.mark.gpu


def sponge_points(rng, n, pos, size):
    """Points inside, on and around the sponge: uniform in an enlarged box, plus points snapped to the
    lattice planes of every level (ties and exact zeros) and points far away."""
    pos = np.asarray(pos, np.float32)
    lo, hi = pos - 0.75 * size, pos + 0.75 * size
    pts = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    m = n // 4
    # snap one or two coordinates of a share of the points onto multiples of size / 3^l / 2 from the centre
    for lvl in range(1, 7):
        step = np.float32(size) / np.float32(3 ** lvl) / np.float32(2)
        sel = slice((lvl - 1) * (m // 6), lvl * (m // 6))
        snapped = pos + np.round((pts[sel] - pos) / step) * step
        mask = rng.random((m // 6, 3)) < 0.5
        pts[sel] = np.where(mask, snapped, pts[sel]).astype(np.float32)
    pts[m:m + 64] = rng.uniform(-500, 500, (64, 3)).astype(np.float32)
    return pts


@pytest.mark.parametrize("levels,pos,size,n_pts", [
    (2, (0, 10, -50), 20.0, 20000), (3, (0, 10, -50), 20.0, 20000), (4, (0, 10, -50), 20.0, 6000),
    (3, (1.37, -2.91, 3.3), 7.77, 20000), (4, (-123.456, 78.9, 0.001), 0.37, 4000),
    (5, (0, 10, -50), 20.0, 600),
])
def test_lattice_sdf_bitwise(rm, oracle, renderer, levels, pos, size, n_pts):
    text = f"Bounds 2000\nColor 0.3 0.6 0.9\nMengerSponge {pos[0]} {pos[1]} {pos[2]} {size} {levels}\n"
    orc = oracle.Scene(text=text)
    pts = sponge_points(np.random.default_rng(levels), n_pts, pos, size)
    osd, orgb = orc.sdf(pts)
    colours = np.array([tuple(p.color) for p in orc.flat_prims()], np.float32)
    for lattice in (True, False):
        if not lattice and levels >= 5:
            continue                      # the exhaustive device evaluator is covered at levels <= 4
        renderer.set_evaluator(lattice)
        try:
            renderer.set_scene(rm.Scene(text=text))
            sd, prim, sd_march = renderer.eval_sdf_both(pts)
            assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
        finally:
            renderer.set_evaluator(True)
        bad = np.flatnonzero(sd.view(np.uint32) != osd.view(np.uint32))
        assert len(bad) == 0, (f"lattice={lattice}: {len(bad)} of {len(pts)} distances differ, first at "
                               f"{pts[bad[0]].tolist()}: {sd[bad[0]]!r} vs {osd[bad[0]]!r}")
        np.testing.assert_array_equal(colours[prim], orgb)


def _nodes_of(rm, text):
    return list(rm.Scene(text=text).nodes())


def test_near_miss_lists_are_not_lattices_and_stay_exact(rm, oracle, renderer):
    """A sponge with one cross nudged, one recoloured, or one missing is not a complete lattice; the
    compiler must fall back to plain runs (results still exact)."""
    text = "Bounds 500\nMengerSponge 0 0 0 9 3\n"
    base = _nodes_of(rm, text)
    lights = np.zeros((0, 3), np.float32)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    pts = sponge_points(np.random.default_rng(9), 6000, (0, 0, 0), 9.0)

    def run(nodes):
        onodes = [oracle.Node(n.type, n.mode, n.child_count, n.center, n.size, n.color) for n in nodes]
        osd, _ = oracle.Scene(nodes=onodes).sdf(pts)
        renderer.set_scene_raw(nodes, lights, ident, 1.0)
        sd, _, sd_march = renderer.eval_sdf_both(pts)
        assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
        assert sd.tobytes() == osd.tobytes()

    def copy(nodes):
        return [rm.Node(n.type, n.mode, n.child_count, n.center, n.size, n.color) for n in nodes]

    # find the sponge list and its crosses
    sponge_at = max(i for i, n in enumerate(base) if n.type == 0)
    first_cross = sponge_at + 2
    nudged = copy(base)
    nudged[first_cross + 57].center[1] = np.nextafter(np.float32(nudged[first_cross + 57].center[1]), np.float32(10))
    run(nudged)
    recoloured = copy(base)
    recoloured[first_cross + 200].color[0] = 0.5
    run(recoloured)
    shorter = copy(base)
    del shorter[-1]
    shorter[sponge_at].child_count -= 1
    run(shorter)
    run(copy(base))


def test_default_frame_both_evaluators_hash(rm, renderer):
    rec = json.load(open(os.path.join(GOLDEN, "default_1024x768_k3.json")))
    renderer.set_params(rm.default_params())
    evals = {}
    for lattice in (True, False):
        renderer.set_evaluator(lattice)
        try:
            renderer.set_scene(rm.Scene(scene_path("default.txt")))
            renderer.set_counting(True)
            renderer.set_shading(False)          # the reference's full work, for the comparison of counts
            img = renderer.render_f32()
            c = renderer.counters()
        finally:
            renderer.set_counting(False)
            renderer.set_shading(True)
            renderer.set_evaluator(True)
        assert hashlib.sha256(img.tobytes()).hexdigest() == rec["sha256_f32"], f"lattice={lattice}"
        assert c["sdf_calls"] == rec["counters"]["sdf_calls"]
        evals[lattice] = c["prim_evals"]
        assert c["compiled_kernel"] == (1 if renderer.kernel_mode == "compiled" else 0)
        if not lattice and renderer.kernel_mode == "interpreted":
            # the record walker visits every primitive at every call
            assert c["prim_evals"] == rec["counters"]["prim_evals"]
            assert c["flops_executed"] == c["sdf_calls"] * c["flops_per_sdf"]
        else:
            assert 0 < c["flops_executed"] < c["sdf_calls"] * c["flops_per_sdf"]
    # the compiled kernels skip lists a bounding child rules out, even with every primitive listed
    assert evals[False] <= rec["counters"]["prim_evals"]
    assert evals[True] < rec["counters"]["prim_evals"] // 5


def test_menger4_frame_bit_identical(rm, oracle, renderer):
    """Menger iterations 4 (8421 crosses), small frame, against the brute-force oracle."""
    text = open(scene_path("default.txt")).read().replace("20.0 3", "20.0 4")
    kw = dict(width=96, height=72, kernel=1)
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    gpu = renderer.render_f32()
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()


def test_closeup6_points_and_tiny_frame(rm, oracle, renderer):
    """configs[3]: Menger iterations 6 (3 368 421 crosses) with the close-up camera.  The brute-force
    oracle needs ~70 ms per SDF call here, so the check is a handful of points and a 6x4 frame."""
    path = scene_path("closeup6.txt")
    prod, orc = rm.Scene(path), oracle.Scene(path)
    assert orc.prim_count == 7 + 3368421
    renderer.set_scene(prod)
    rng = np.random.default_rng(6)
    pts = sponge_points(rng, 96, (0, 10, -50), 20.0)
    sd, _, sd_march = renderer.eval_sdf_both(pts)
    assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
    osd, _ = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    kw = dict(width=6, height=4, kernel=1)
    renderer.set_params(rm.default_params(**kw))
    gpu = renderer.render_f32()
    cpu, _ = orc.render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()


@pytest.mark.parametrize("name", ["synthetic1024.txt", "synthetic1024_wide.txt"])
def test_synthetic1024_frame_bit_identical(rm, oracle, renderer, name):
    """configs[4]: 1024 mixed primitives in the root union, 8 lights; small frame against the oracle.  The second file has
    every size, the crosses' bar widths included, in [1, 4]: a thicket (long cell lists, most points near many bars)."""
    path = scene_path(name)
    kw = dict(width=80, height=60, kernel=2)
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(path))
    gpu = renderer.render_f32()
    orc = oracle.Scene(path)
    assert (orc.prim_count, orc.light_count) == (1026, 8)
    cpu, _ = orc.render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()
    pts = np.random.default_rng(3).uniform([-70, -2, -120], [70, 40, 20], (5000, 3)).astype(np.float32)
    sd, prim, sd_march = renderer.eval_sdf_both(pts)
    assert same_but_zero_signs(sd_march, sd)   # the march loop's pruning evaluator agrees (zero signs aside)
    osd, orgb = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    colours = np.array([tuple(p.color) for p in orc.flat_prims()], np.float32)
    np.testing.assert_array_equal(colours[prim], orgb)
