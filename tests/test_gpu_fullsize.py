"""Every benchmark configuration at the size the benchmark runs it, checked through properties that need no full CPU frame.

configs[3] (Menger iterations 6 close-up, 1024x768, SSAA 3; 3 368 421 crosses under Body::List::SDF,
/root/reference/source/body.cpp:66-111): the frame of the kernel generated for the scene in natural tile order, the same
kernel with the tiles in cost order and the longest ones split, and the record-walking kernel must be the same bytes; a few
pixels of it are checked against the brute-force oracle (70 ms per scene-SDF call there, hence few).
configs[4] (1024 loose primitives, 8 lights, both scene files, 1024x768, SSAA 3): generated kernel == record walker,
cell grid == hull walk (the grid switched off), cost order == natural order; oracle pixels.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

FULL = dict(width=1024, height=768, kernel=3)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _frames(rm, scene, compiled, evaluator_lattice=True, renders=1):
    """Float frames of `renders` successive renders in a fresh context (the first times its tiles in natural order)."""
    r = rm.Renderer(0)
    try:
        r.set_kernel_mode(compiled)
        r.set_evaluator(evaluator_lattice)
        r.set_params(rm.default_params(**FULL))
        r.set_scene(scene)
        out = []
        for _ in range(renders):
            out.append(r.render_f32())
            out[-1] = (out[-1], r.counters())
        return out
    finally:
        r.close()


def test_closeup6_full_frame_is_the_same_whatever_renders_it(rm, oracle):
    path = scene_path("closeup6.txt")
    scene = rm.Scene(path)
    (natural, c0), (ordered, c1), (again, c2) = _frames(rm, scene, True, renders=3)
    assert c0["tiles_by_cost"] == 0 and c0["compiled_kernel"] == 1
    assert c1["tiles_by_cost"] == 1 and c2["tiles_by_cost"] == 1
    assert _sha(ordered) == _sha(natural) and _sha(again) == _sha(natural)
    (walked, cw), = _frames(rm, scene, False)
    assert cw["compiled_kernel"] == 0
    assert _sha(walked) == _sha(natural)
    # oracle pixels: sponge face, a tunnel's rim, the floor in front, the frame's last pixel
    orc = oracle.Scene(path)
    xy = [(512, 300), (200, 420), (830, 700), (1023, 767)]
    cpu = orc.render_pixels(oracle.default_params(**FULL), xy)
    got = np.stack([natural[y, x] for x, y in xy])
    assert got.tobytes() == cpu.tobytes()


@pytest.mark.parametrize("name", ["synthetic1024.txt", "synthetic1024_wide.txt"])
def test_synthetic1024_full_frame_is_the_same_whatever_renders_it(rm, oracle, name, monkeypatch):
    path = scene_path(name)
    scene = rm.Scene(path)
    (natural, c0), (ordered, c1) = _frames(rm, scene, True, renders=2)
    assert c0["tiles_by_cost"] == 0 and c1["tiles_by_cost"] == 1
    assert _sha(ordered) == _sha(natural)
    # the cell grid switched off: lanes take the hull walk everywhere
    monkeypatch.setenv("RM_GRID_MAX_ENTRIES", "0")
    (hulls, ch), = _frames(rm, scene, True)
    monkeypatch.delenv("RM_GRID_MAX_ENTRIES")
    assert ch["compiled_kernel"] == 1
    assert _sha(hulls) == _sha(natural)
    (walked, cw), = _frames(rm, scene, False)
    assert cw["compiled_kernel"] == 0
    assert _sha(walked) == _sha(natural)
    orc = oracle.Scene(path)
    rng = np.random.default_rng(11)
    xy = [(int(x), int(y)) for x, y in zip(rng.integers(0, 1024, 160), rng.integers(0, 768, 160))] + [(1023, 767), (0, 0)]
    cpu = orc.render_pixels(oracle.default_params(**FULL), xy)
    got = np.stack([natural[y, x] for x, y in xy])
    assert got.tobytes() == cpu.tobytes()


def test_grid_falls_back_when_its_lists_would_not_fit(rm, oracle):
    """Four thousand coincident spheres (plus loose ones): every cell they can be nearest in lists all of them, the grid's
    entry count passes 2^32 (it is counted in 64 bits; a 32-bit sum would wrap to a small number and the fill pass would
    write past its buffer) and the scene is left to the hull walk; the frame is the oracle's."""
    rng = np.random.default_rng(5)
    lines = ["Bounds 200.0", "Camera Position 0.0 8.0 30.0", "Camera Direction 0.0 -0.2 -1.0", "Camera Up 0.0 1.0 0.0", "Camera FOV 60",
             "Light 20.0 40.0 20.0", "Color 0.8 0.6 0.2"]
    lines += ["Sphere 0.0 3.0 0.0 3.0"] * 4000
    lines += [f"Sphere {x:.3f} {y:.3f} {z:.3f} 0.7" for x, y, z in rng.uniform([-12, 0, -12], [12, 8, 12], (40, 3))]
    text = "\n".join(lines) + "\n"
    kw = dict(width=96, height=64, kernel=1)
    r = rm.Renderer(0)
    try:
        r.set_params(rm.default_params(**kw))
        r.set_scene(rm.Scene(text=text))
        gpu = r.render_f32()
    finally:
        r.close()
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()


def test_tables_too_long_for_a_list_entry_take_the_hull_walk(rm, oracle):
    """A list entry of the cell grid names its table row in 14 bits: a scene of 17 000 loose spheres gets no grid (every lane
    takes the hull walk) and its frame is still the oracle's; one of 3 000 spheres gets the grid, and the same frame with it
    switched off."""
    rng = np.random.default_rng(17)
    head = ["Bounds 300.0", "Camera Position 0.0 30.0 120.0", "Camera Direction 0.0 -0.25 -1.0", "Camera Up 0.0 1.0 0.0", "Camera FOV 60",
            "Light 60.0 120.0 80.0", "Color 0.3 0.7 0.9"]

    def text(n):
        c = rng.uniform([-80, 0, -80], [80, 40, 80], (n, 3))
        r = rng.uniform(0.2, 1.2, n)
        return "\n".join(head + [f"Sphere {x:.3f} {y:.3f} {z:.3f} {q:.3f}" for (x, y, z), q in zip(c, r)]) + "\n"

    kw = dict(width=40, height=24, kernel=1)

    def frame(t):
        r = rm.Renderer(0)
        try:
            r.set_params(rm.default_params(**kw))
            r.set_scene(rm.Scene(text=t))
            return r.render_f32()
        finally:
            r.close()

    many = text(17000)
    cpu, _ = oracle.Scene(text=many).render(oracle.default_params(**kw))
    assert frame(many).tobytes() == cpu.tobytes()
    fewer = text(3000)
    with_grid = frame(fewer)
    os.environ["RM_GRID_MAX_ENTRIES"] = "0"
    try:
        without = frame(fewer)
    finally:
        del os.environ["RM_GRID_MAX_ENTRIES"]
    assert with_grid.tobytes() == without.tobytes()
    cpu, _ = oracle.Scene(text=fewer).render(oracle.default_params(**kw))
    assert with_grid.tobytes() == cpu.tobytes()
