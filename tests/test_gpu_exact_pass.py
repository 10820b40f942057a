"""The two kernels of a scene with lattices: the frame's kernel decides a point's distance to a lattice from the crosses
round the nearest centres and marks the tiles where that was not enough; the exact kernel behind it renders the marked
tiles again with the full walk of Body::List::SDF (/root/reference/source/body.cpp:66-111).  Rays as good as never need
it, so these tests make them: a build of the kernels that marks every tile, and a camera inside the sponge's material."""
import hashlib

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _render(rm, scene, compiled, renders=1, **kw):
    r = rm.Renderer(0)
    try:
        r.set_kernel_mode(compiled)
        r.set_params(rm.default_params(**kw))
        r.set_scene(scene)
        frames = [r.render_f32() for _ in range(renders)]
        return frames, r.exact_tiles()
    finally:
        r.close()


@pytest.mark.parametrize("kw", [dict(width=203, height=117, kernel=3), dict(width=160, height=96, kernel=2),
                                dict(width=96, height=64, kernel=5), dict(width=64, height=40, kernel=1),
                                dict(width=37, height=22, kernel=9)],
                         ids=["k3", "k2", "k5", "k1", "k9"])
def test_every_tile_marked_gives_the_same_frame(rm, monkeypatch, kw):
    """A build of the scene's kernels in which the frame's kernel marks every tile: the exact kernel renders the whole
    frame over it -- with and without a sample image between the kernels, in natural and in cost order -- and the frame is
    the record walker's.  k = 3, 5: the exact kernel's launch also resolves the sample image (one workgroup per k x k tiles);
    k = 9: too many tiles per workgroup for that, the resolve kernel follows it; k = 1, 2: no sample image."""
    scene = rm.Scene(scene_path("default.txt"))
    (walked,), none = _render(rm, scene, False, **kw)
    assert none == 0
    (plain, plain2), few = _render(rm, scene, True, renders=2, **kw)
    assert _sha(plain) == _sha(walked) and _sha(plain2) == _sha(walked)
    monkeypatch.setenv("RM_JIT_EXTRA", "-DRM_FLAG_EVERY_TILE")
    frames, tiles = _render(rm, scene, True, renders=3, **kw)
    for f in frames:
        assert _sha(f) == _sha(walked)
    assert tiles >= 3 and tiles % 3 == 0 and tiles > few
    # the count is the launch's tile count, whatever the tile shape the library chose
    sw, sh = kw["width"] * kw["kernel"], kw["height"] * kw["kernel"]
    assert tiles // 3 in {((sw + s - 1) // s) * ((sh + s - 1) // s) for s in (8, 16, 24)}


def test_every_tile_marked_static_scratch(rm, monkeypatch):
    """The same through the entry point that allocates nothing: its marks live behind the sample image in the static array."""
    import torch
    monkeypatch.setenv("RM_JIT_EXTRA", "-DRM_FLAG_EVERY_TILE")
    r = rm.Renderer(0)
    try:
        r.set_scene(rm.Scene(scene_path("default.txt")))
        for kw in (dict(width=203, height=117, kernel=3), dict(width=64, height=40, kernel=2)):
            p = rm.default_params(**kw)
            r.set_params(p)
            rows = rm.Rows(p.height, 0, 1, 1)
            a = torch.zeros((p.height, p.width, 4), dtype=torch.float32, device="cuda")
            b = torch.zeros_like(a)
            torch.cuda.synchronize()
            before = r.exact_tiles()
            r.enqueue_device_static(rows, a.data_ptr(), None)
            r.wait()
            assert r.exact_tiles() > before
            r.render_device(rows, b.data_ptr(), None)
            assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
    finally:
        r.close()


INSIDE = """Bounds 200.0
Camera Position 9.9 19.9 -40.1
Camera Direction -1.0 -0.5 -1.0
Camera Up 0.0 1.0 0.0
Camera FOV 75
Light -15.0 40.0 -20.0
Box 0.0 -1.0 -50.0 75.0 2.0 75.0
Color 1.0 1.0 1.0
MengerSponge 0.0 10.0 -50.0 20.0 4
"""


def test_camera_inside_the_material(rm, oracle):
    """Every ray starts deep inside the sponge's material (the corner cell of every level): the scene distance there is
    negative, the march ends at once and the normal's taps and shadow rays sit where no cross is near."""
    kw = dict(width=48, height=32, kernel=2)
    scene = rm.Scene(text=INSIDE)
    (gpu,), tiles = _render(rm, scene, True, **kw)
    (walked,), _ = _render(rm, scene, False, **kw)
    cpu, _ = oracle.Scene(text=INSIDE).render(oracle.default_params(**kw))
    assert walked.tobytes() == cpu.tobytes()
    assert gpu.tobytes() == cpu.tobytes()
    print("tiles rendered again:", tiles)


@pytest.mark.parametrize("iterations,camera,whole", [
    (5, ("-1.1134 5.3313 -44.9354", "0.4565 0.6824 0.5709"), True),
    (6, ("8.7631 10.1043 -51.0847", "-0.1227 0.3563 0.9263"), False)], ids=["inside-5", "grazing-6"])
def test_cameras_that_need_the_exact_kernel(rm, iterations, camera, whole):
    """Two cameras a random search found (tools/soak_lattice.py): one inside the material of a Menger-5 sponge, where the frame's
    kernel marks every tile, one at a Menger-6 sponge where two tiles are marked.  The frames are the record walker's (its own
    lattice walk), in natural and in cost order."""
    kw = dict(width=96, height=64, kernel=2)
    text = "\n".join(["Bounds 200.0", "Camera Position " + camera[0], "Camera Direction " + camera[1], "Camera Up 0.0 1.0 0.0", "Camera FOV 70",
                      "Light -15.0 40.0 -20.0", "Light 45.0 50.0 -35.0", "Box 0.0 -1.0 -50.0 75.0 2.0 75.0", "Color 1.0 1.0 1.0",
                      f"MengerSponge 0.0 10.0 -50.0 20.0 {iterations}"]) + "\n"
    scene = rm.Scene(text=text)
    (first, second), tiles = _render(rm, scene, True, renders=2, **kw)
    (walked,), _ = _render(rm, scene, False, **kw)
    assert first.tobytes() == walked.tobytes() and second.tobytes() == walked.tobytes()
    assert tiles > 0 and tiles % 2 == 0
    if whole:
        assert tiles == 2 * (96 * 2 // 8) * (64 * 2 // 8)


def test_kernels_come_from_the_program_beside_the_library(rm, tmp_path):
    """A scene no cache has seen: its kernels are compiled by rm_jit (three programs side by side, three modules), the code
    objects land in the cache directory, and a second context finds them in the process's cache; the frame is the record
    walker's either way."""
    kw = dict(width=64, height=48, kernel=3)
    text = INSIDE.replace("MengerSponge 0.0 10.0 -50.0 20.0 4", "MengerSponge 0.25 10.0 -50.0 20.0 3").replace("9.9 19.9 -40.1", "30.0 25.0 -20.0")
    rm.set_kernel_cache_dir(str(tmp_path))
    try:
        scene = rm.Scene(text=text)
        (gpu,), _ = _render(rm, scene, True, **kw)
        assert rm.kernel_compiler().startswith("rm_jit, a process of its own: "), rm.kernel_compiler()
        assert len([f for f in tmp_path.iterdir() if f.name.endswith(".gfx950.bin")]) >= 3   # frame, exact pass, point evaluation
        (again,), _ = _render(rm, scene, True, **kw)
        (walked,), _ = _render(rm, scene, False, **kw)
        assert gpu.tobytes() == walked.tobytes() and again.tobytes() == walked.tobytes()
    finally:
        rm.set_kernel_cache_dir(None)


def test_compile_started_ahead_of_the_context(rm, tmp_path):
    """rm_precompile_scene: the compile of a scene's kernels starts before there is a context; rm_set_scene waits for it
    instead of compiling (every code object in the cache directory was written once), and the frame is the record walker's."""
    kw = dict(width=64, height=48, kernel=2)
    text = INSIDE.replace("MengerSponge 0.0 10.0 -50.0 20.0 4", "MengerSponge 0.5 10.0 -50.0 20.0 3").replace("9.9 19.9 -40.1", "31.0 25.0 -20.0")
    rm.set_kernel_cache_dir(str(tmp_path))
    try:
        scene = rm.Scene(text=text)
        rm.precompile(scene)
        rm.precompile(scene)                      # a second announcement starts nothing new
        (gpu,), _ = _render(rm, scene, True, **kw)
        objects = sorted(f.name for f in tmp_path.iterdir() if f.name.endswith(".gfx950.bin"))
        assert len(objects) == 3, objects
        (walked,), _ = _render(rm, scene, False, **kw)
        assert gpu.tobytes() == walked.tobytes()
        with pytest.raises(rm.RaymarchError):
            rm.lib() and rm.binding._check(rm.lib().rm_precompile_scene(None, 0, 1), "rm_precompile_scene")
    finally:
        rm.set_kernel_cache_dir(None)
