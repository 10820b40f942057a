"""Edge cases of the render path against the oracle: many lights (the reference's GL path stops at 16, its CPU path
does not: /root/reference/source/scene.cpp:53-62), large SSAA kernels, iteration caps that leave rays unconverged,
one-pixel and one-row frames, loose tolerances, the deepest nesting the device walkers take."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

LIST, SPHERE, BOX, CROSS = 0, 1, 2, 3

BASE = open(scene_path("menger2.txt")).read()


def _frame_pair(rm, oracle, renderer, text, **kw):
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    gpu = renderer.render_f32()
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    return gpu, cpu


def test_twenty_lights(rm, oracle, renderer):
    rng = np.random.default_rng(3)
    lights = "".join(f"Light {x:.3f} {y:.3f} {z:.3f}\n" for x, y, z in rng.uniform([-50, 30, -80], [50, 60, 0], (18, 3)))
    gpu, cpu = _frame_pair(rm, oracle, renderer, BASE + "\n" + lights, width=96, height=64, kernel=1)
    assert gpu.tobytes() == cpu.tobytes()


@pytest.mark.parametrize("n_lights", [3, 5, 18])
def test_split_tiles_march_their_shadow_rays_side_by_side(rm, oracle, renderer, n_lights):
    """In the parts of a split tile (the longest tiles of a cost-ordered frame) the shadow rays of a sample march on
    different lanes of the wave, several lights per round, and the sample's lane adds the contributions up in the order
    of the lights (scene.cpp:52-62): 3 lights fit one round, 5 take two rounds in a quarter (four groups of 16 lanes)
    with the last round short, 18 two rounds in a sixteenth too.  Same frame as the oracle's, parts or not."""
    rng = np.random.default_rng(40 + n_lights)
    lights = "".join(f"Light {x:.3f} {y:.3f} {z:.3f}\n" for x, y, z in rng.uniform([-50, 30, -80], [50, 60, 0], (n_lights - 2, 3)))
    text = open(scene_path("default.txt")).read() + "\n" + lights
    kw = dict(width=160, height=100, kernel=2)
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    assert renderer.render_f32().tobytes() == cpu.tobytes()            # natural order, whole tiles
    for _ in range(2):
        again = renderer.render_f32()
        c = renderer.counters()
        assert c["tiles_by_cost"] == 1 and (c["tile_parts"] > 0) == (renderer.kernel_mode == "compiled")
        assert again.tobytes() == cpu.tobytes()


@pytest.mark.parametrize("kw", [dict(width=40, height=30, kernel=5), dict(width=45, height=31, kernel=4), dict(width=21, height=13, kernel=8),
                                dict(width=3, height=2, kernel=4), dict(width=1, height=1, kernel=3), dict(width=257, height=1, kernel=2),
                                dict(width=3, height=129, kernel=1), dict(width=64, height=48, kernel=2, iterations=7),
                                dict(width=64, height=48, kernel=1, iterations=1), dict(width=48, height=32, kernel=2, iterations=0),
                                dict(width=64, height=48, kernel=2, surface_eps=0.05, offset_eps=0.2),
                                dict(width=64, height=48, kernel=1, saturation=0.5)])
def test_unusual_parameters(rm, oracle, renderer, kw):
    gpu, cpu = _frame_pair(rm, oracle, renderer, BASE, **kw)
    assert gpu.shape == (kw["height"], kw["width"], 4)
    assert gpu.tobytes() == cpu.tobytes()


@pytest.mark.parametrize("saturation", [0.05, 0.0, -0.25])
def test_colourless_surfaces_in_both_shading_modes(rm, oracle, renderer, saturation):
    """Samples on a surface of colour (0, 0, 0) may skip normal and shadow rays (rm_set_shading; scene.cpp:31 multiplies
    the lighting by the colour): a black sphere between coloured bodies, waves that mix both kinds, and lower
    clamps of the lighting (scene.cpp:60) down to a negative one, under which the shortcut does not apply."""
    text = BASE.replace("Color 1.0 0.0 0.0", "Color 0.0 0.0 0.0")
    assert text != BASE
    kw = dict(width=96, height=72, kernel=2, saturation=saturation)
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    try:
        for coloured_only in (True, False):
            renderer.set_shading(coloured_only)
            assert renderer.render_f32().tobytes() == cpu.tobytes(), f"coloured_only={coloured_only}"
    finally:
        renderer.set_shading(True)


def test_lights_in_awkward_places_in_both_shading_modes(rm, oracle, renderer):
    """Shadow rays may stop once they are past their light (rm_set_shading): lights inside a body, a hair above the
    floor, outside the room (never passed), next to the camera, and a room ten times the size (longer steps, larger
    coordinates) must give the oracle's frame with and without that exit."""
    lights = "Light -9.5 7.0 -25.5\nLight 3.0 0.004 -30.0\nLight 0.0 500.0 0.0\nLight 49.0 34.5 -3.0\nLight 0.0 10.0 -50.0\n"
    texts = [BASE + "\n" + lights, (BASE + "\n" + lights).replace("Bounds 200.0", "Bounds 2000.0")]
    assert texts[1] != texts[0]
    kw = dict(width=80, height=60, kernel=2)
    for text in texts:
        cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
        renderer.set_params(rm.default_params(**kw))
        renderer.set_scene(rm.Scene(text=text))
        try:
            for coloured_only in (True, False):
                renderer.set_shading(coloured_only)
                assert renderer.render_f32().tobytes() == cpu.tobytes(), f"coloured_only={coloured_only}"
        finally:
            renderer.set_shading(True)


def test_camera_inside_and_looking_away(rm, oracle, renderer):
    """A camera inside the sponge's bounding box looking along an axis (ties between symmetric crosses), and one
    that looks straight up (rays that only ever meet the room)."""
    inside = BASE.replace("Camera Position 50.0 35.0 -2.5", "Camera Position 0.0 10.0 -50.0").replace(
        "Camera Direction -1.0 -0.5 -1.0", "Camera Direction 0.0 0.0 -1.0")
    up = BASE.replace("Camera Direction -1.0 -0.5 -1.0", "Camera Direction 0.0 1.0 0.001")
    for text in (inside, up):
        assert text != BASE
        gpu, cpu = _frame_pair(rm, oracle, renderer, text, width=80, height=60, kernel=2)
        assert gpu.tobytes() == cpu.tobytes()


def _nested(rm_node, depth):
    f3 = C.c_float * 3
    zero = f3(0, 0, 0)
    nodes = []
    for d in range(depth):
        nodes.append(rm_node(LIST, d % 4, 2 if d + 1 < depth else 1, zero, zero, zero))
        nodes.append(rm_node(SPHERE, 0, 0, f3(1.5 * d, 0.5 * d, -d), f3(2.0 + 0.25 * d, 0, 0), f3(0.1 * d, 1, 1)))
    return nodes


def test_deepest_nesting_and_one_level_more(rm, oracle, renderer):
    """The root list plus 8 nested levels is the limit of the device walkers; one more is refused with a message."""
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    pts = np.random.default_rng(4).uniform(-14, 14, (4000, 3)).astype(np.float32)
    ok = _nested(rm.Node, 9)
    renderer.set_scene_raw(ok, np.zeros((0, 3), np.float32), ident, 1.0)
    sd, _, sd_march = renderer.eval_sdf_both(pts)
    osd, _ = oracle.Scene(nodes=_nested(oracle.Node, 9)).sdf(pts)
    assert sd.tobytes() == osd.tobytes() and sd_march.tobytes() == osd.tobytes()
    with pytest.raises(rm.RaymarchError, match="nested deeper"):
        renderer.set_scene_raw(_nested(rm.Node, 10), np.zeros((0, 3), np.float32), ident, 1.0)
    renderer.set_scene_raw(ok, np.zeros((0, 3), np.float32), ident, 1.0)   # the context is still usable


@pytest.mark.parametrize("lights", [1, 2])
def test_coloured_sample_with_a_nan_normal(rm, oracle, renderer, lights):
    """normalize(grad) of a zero gradient is NaN (scene.cpp:28) and the lighting clamp then decides the sample
    (scene.cpp:60).  The camera sits at the centre of a coloured cube so small that the first step (applied before the
    test, scene.cpp:37-39) does not move the position: every ray ends where it began, all six gradient taps give the
    same value, the normal is NaN, every shadow ray starts at NaN and runs its full iteration count, and the sample is
    saturation x colour -- with the reference's min(max(saturation, NaN), 1).  Both shading modes, both kernel modes."""
    f3 = C.c_float * 3

    def nodes(node):
        return [node(LIST, 0, 3, f3(0, 0, 0), f3(0, 0, 0), f3(0, 0, 0)),
                node(LIST, 1, 1, f3(0, 0, 0), f3(0, 0, 0), f3(0, 0, 0)),            # Bounds: COMPLEMENT { box }
                node(BOX, 0, 0, f3(0, 0, 0), f3(200, 200, 200), f3(0, 0, 0)),
                node(BOX, 0, 0, f3(0, 0, 0), f3(2e-30, 2e-30, 2e-30), f3(1.0, 0.5, 0.25)),
                node(SPHERE, 0, 0, f3(0, -30, -40), f3(10, 0, 0), f3(0, 1, 0))]
    light_xyz = np.array([[30, 40, 10], [-20, 50, -30]], np.float32)[:lights]
    for iterations in (1, 25):
        kw = dict(width=24, height=10, kernel=2, iterations=iterations)
        # camera at the origin (the cube's centre), looking down -z
        orc = oracle.Scene(nodes=nodes(oracle.Node), lights=light_xyz, cam_position=(0, 0, 0), cam_direction=(0, 0, -1), cam_fov=70.0)
        transform, focal = orc.camera()
        cpu, _ = orc.render(oracle.default_params(**kw))
        assert np.allclose(cpu[..., :3], np.float32(0.05) * np.array([1.0, 0.5, 0.25], np.float32)), "the oracle's samples are saturation x colour"
        renderer.set_params(rm.default_params(**kw))
        renderer.set_scene_raw(nodes(rm.Node), light_xyz, transform, float(focal))
        try:
            for coloured_only in (True, False):
                renderer.set_shading(coloured_only)
                assert renderer.render_f32().tobytes() == cpu.tobytes(), f"iterations={iterations} coloured_only={coloured_only}"
        finally:
            renderer.set_shading(True)


def test_slot_statistics_cover_every_workgroup(rm, renderer):
    """rm_set_slot_stats: one record per workgroup of the render, in launch order -- every tile once (whole, or as its 4 or
    16 parts), durations and march lengths filled in; the frame is the same with the statistics on."""
    kw = dict(width=160, height=100, kernel=2)
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(scene_path("default.txt")))
    plain = [renderer.render_f32() for _ in range(2)][-1]
    renderer.set_slot_stats(True)
    try:
        frame = renderer.render_f32()
        st = renderer.slot_stats()
    finally:
        renderer.set_slot_stats(False)
    assert frame.tobytes() == plain.tobytes()
    side = 8 if renderer.kernel_mode == "compiled" else 16
    tiles_x = (160 * 2 + (side if side == 8 else 32) - 1) // (side if side == 8 else 32)
    n_tiles = tiles_x * ((100 * 2 + side - 1) // side)
    entry, ticks, turns = st[:, 0], st[:, 1], st[:, 2]
    part = entry >> 27
    covered = np.zeros(n_tiles)
    np.add.at(covered, entry & ((1 << 27) - 1), np.where(part == 0, 1.0, np.where(part < 5, 0.25, 1 / 16)))
    assert np.allclose(covered, 1.0)
    assert (ticks > 0).all() and ((turns & 0xffff) > 0).all()
    assert len(st) == n_tiles + renderer.counters()["tile_parts"]
    renderer.render_f32()
    assert len(renderer.slot_stats()) == 0          # off again: nothing recorded


def test_host_image_written_by_the_resolve_pass(rm):
    """rm_render_rgba8 into page-locked memory: where a resolve pass follows the trace kernel (SSAA 3) that pass writes the
    host image itself, elsewhere (SSAA 2: the trace kernel writes the pixels) and for pageable memory a copy follows the
    kernels.  Memory from rm_host_alloc, pages of the caller's own (registered where the machine allows it), an address
    inside them, a pageable array: the same bytes every time."""
    import ctypes as C
    r = rm.Renderer(0)
    try:
        r.set_scene(rm.Scene(scene_path("default.txt")))
        for kw in (dict(width=203, height=117, kernel=3), dict(width=160, height=96, kernel=2)):
            p = rm.default_params(**kw)
            r.set_params(p)
            want = r.render_rgba8()                                   # pageable
            n = p.width * p.height * 4
            locked = rm.HostBuffer((p.height, p.width, 4), np.uint8)
            try:
                for _ in range(2):                                    # first render times the tiles, second runs in cost order
                    locked.array[...] = 7
                    assert rm.lib().rm_render_rgba8(r._h, locked.ptr) == 0
                    assert locked.array.tobytes() == want.tobytes(), kw
            finally:
                locked.free()
            import mmap
            pages = mmap.mmap(-1, (n + 4096 + 4095) // 4096 * 4096)   # whole pages of the caller's own (a small heap block does not register)
            own = np.frombuffer(pages, np.uint8)
            registered = rm.lib().rm_host_register(own.ctypes.data, own.nbytes) == 0   # (some machines refuse to lock a process's own pages)
            try:
                for offset in (0, 1024):                              # (1024: not the start of the registered range)
                    own[...] = 9
                    assert rm.lib().rm_render_rgba8(r._h, own.ctypes.data + offset) == 0
                    assert own[offset:offset + n].tobytes() == want.tobytes(), (kw, offset, registered)
                    assert (own[:offset] == 9).all() and (own[offset + n:] == 9).all()
            finally:
                if registered:
                    assert rm.lib().rm_host_unregister(own.ctypes.data) == 0
                del own
                pages.close()
    finally:
        r.close()
