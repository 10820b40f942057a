"""Known-answer tests of the CPU oracle, derived by hand from the reference's formulas
(body.cpp:32-57 primitives, :66-111 list modes, :114-169 generators; object.cpp:23-36 camera)."""
import numpy as np

from conftest import scene_path

SPHERE, BOX, CROSS, LIST = 1, 2, 3, 0
UNION, COMPLEMENT, INTERSECTION, DIFFERENCE = 0, 1, 2, 3


def node(oracle, type_, mode=0, n=0, c=(0, 0, 0), s=(0, 0, 0), col=(1, 1, 1)):
    return oracle.Node(type_, mode, n, (oracle.C.c_float * 3)(*c), (oracle.C.c_float * 3)(*s),
                       (oracle.C.c_float * 3)(*col))


def sd_of(oracle, nodes, pts):
    sc = oracle.Scene(nodes=nodes)
    return sc.sdf(np.array(pts, np.float32))


def test_sphere(oracle):
    sd, _ = sd_of(oracle, [node(oracle, LIST, UNION, 1), node(oracle, SPHERE, s=(1, 0, 0))], [(2, 0, 0), (0, 0, 0)])
    assert sd.tolist() == [1.0, -1.0]


def test_box_is_chebyshev_with_full_edge_size(oracle):
    sd, _ = sd_of(oracle, [node(oracle, LIST, UNION, 1), node(oracle, BOX, s=(2, 2, 2))],
                  [(2, 0, 0), (0, 0, 0), (2, 3, 0)])
    assert sd.tolist() == [1.0, -1.0, 2.0]


def test_cross_is_median_of_slab_distances(oracle):
    sd, _ = sd_of(oracle, [node(oracle, LIST, UNION, 1), node(oracle, CROSS, s=(2, 2, 2))],
                  [(3, 0, 0), (3, 3, 0), (0.5, 0.25, 0)])
    assert sd.tolist() == [-1.0, 2.0, -0.75]


def _two(oracle, mode):
    # A: sphere radius 1 at x=2 (SD at origin 1.0, red); B: sphere radius 1 at x=3 (SD 2.0, green)
    nodes = [node(oracle, LIST, mode, 2),
             node(oracle, SPHERE, c=(2, 0, 0), s=(1, 0, 0), col=(1, 0, 0)),
             node(oracle, SPHERE, c=(3, 0, 0), s=(1, 0, 0), col=(0, 1, 0))]
    sd, rgb = sd_of(oracle, nodes, [(0, 0, 0)])
    return float(sd[0]), rgb[0].tolist()


def test_list_modes(oracle):
    assert _two(oracle, UNION) == (1.0, [1, 0, 0])
    assert _two(oracle, COMPLEMENT) == (-2.0, [0, 1, 0])        # min(-1, -2)
    assert _two(oracle, INTERSECTION) == (2.0, [0, 1, 0])       # max(1, 2)
    assert _two(oracle, DIFFERENCE) == (1.0, [1, 0, 0])         # max(1, -2)


def test_single_child_complement_negates_and_empty_list_is_inf_black(oracle):
    sd, _ = sd_of(oracle, [node(oracle, LIST, COMPLEMENT, 1), node(oracle, SPHERE, s=(1, 0, 0))], [(2, 0, 0)])
    assert sd.tolist() == [-1.0]
    sd, rgb = sd_of(oracle, [node(oracle, LIST, UNION, 0)], [(0, 0, 0)])
    assert np.isposinf(sd[0]) and rgb[0].tolist() == [0, 0, 0]


def test_ties_keep_the_earlier_operand(oracle):
    nodes = [node(oracle, LIST, UNION, 2),
             node(oracle, SPHERE, c=(2, 0, 0), s=(1, 0, 0), col=(1, 0, 0)),
             node(oracle, SPHERE, c=(-2, 0, 0), s=(1, 0, 0), col=(0, 1, 0))]
    _, rgb = sd_of(oracle, nodes, [(0, 0, 0)])
    assert rgb[0].tolist() == [1, 0, 0]


def test_bounds_is_complement_of_black_box(oracle):
    sc = oracle.Scene(text="Bounds 200.0\n")
    sd, rgb = sc.sdf(np.zeros((1, 3), np.float32))
    assert sd.tolist() == [100.0] and rgb[0].tolist() == [0, 0, 0]


def test_menger_counts_and_first_centres(oracle):
    for it, crosses in ((1, 1), (2, 21), (3, 421), (4, 8421)):
        sc = oracle.Scene(text=f"MengerSponge 0 10 -50 20 {it}\n")
        assert sc.prim_count == 1 + crosses
    prims = oracle.Scene(text="MengerSponge 0 10 -50 20 2\n").flat_prims()
    d = np.float32(20.0) / np.float32(3)
    assert prims[0].type == BOX and list(prims[0].size) == [20.0, 20.0, 20.0]
    assert prims[1].type == CROSS and list(prims[1].center) == [0.0, 10.0, -50.0]
    assert list(prims[1].size) == [d, d, d]
    d2 = d / np.float32(3)
    assert list(prims[2].center) == [np.float32(0) + d, np.float32(10) - d, np.float32(-50) - d]
    assert list(prims[2].size) == [d2, d2, d2]
    assert list(prims[3].center) == [0.0, np.float32(10) - d, np.float32(-50) - d]


def test_deathstar_second_sphere(oracle):
    prims = oracle.Scene(text="DeathStar 12.5 4.5 -25 4.5\n").flat_prims()
    assert list(prims[0].center) == [12.5, 4.5, -25.0]
    assert list(prims[1].center) == [12.5 + 1.5 * 4.5, 4.5, -25.0]
    assert prims[0].size[0] == prims[1].size[0] == 4.5


def test_default_scene_structure(oracle):
    sc = oracle.Scene(scene_path("default.txt"))
    assert (sc.prim_count, sc.list_count, sc.light_count) == (428, 4, 2)
    modes, parents = sc.list_modes()
    assert modes.tolist() == [UNION, COMPLEMENT, DIFFERENCE, DIFFERENCE]
    assert parents.tolist() == [-1, 0, 0, 0]
    assert sc.lights().tolist() == [[-15.0, 40.0, -20.0], [45.0, 50.0, -35.0]]
    # sticky colour: white floor, blue death star, red and green spheres, white sponge
    cols = [tuple(p.color) for p in sc.flat_prims()]
    assert cols[0] == (0, 0, 0) and cols[1] == (1, 1, 1)
    assert cols[2] == cols[3] == (0, 0, 1) and cols[4] == (1, 0, 0) and cols[5] == (0, 1, 0)
    assert set(cols[6:]) == {(1, 1, 1)}


def test_default_camera(oracle):
    t, focal = oracle.Scene(scene_path("default.txt")).camera()
    np.testing.assert_allclose(t[0:3], [0.70711, 0, -0.70711], atol=1e-5)
    np.testing.assert_allclose(t[4:7], [-0.23570, 0.94281, -0.23570], atol=1e-5)
    np.testing.assert_allclose(t[8:11], [0.66667, 0.33333, 0.66667], atol=1e-5)
    assert t[12:16].tolist() == [50.0, 35.0, -2.5, 1.0]
    assert abs(float(focal) - 1.5346540) < 1e-6
    # tanf and tan-in-double agree for this FOV, so the overload question is moot here
    half = np.float32(75) * (np.float32(np.pi) / np.float32(180)) / np.float32(2)
    assert np.float32(2) * np.float32(np.tan(np.float64(half))) == focal


def test_quantisation_rule(oracle):
    px = np.array([[0.05, 1.0, 0.0, 1.0], [2.0, -1.0, 0.5, 1.0]], np.float32)
    assert oracle.quantize(px).tolist() == [[13, 255, 0, 255], [255, 0, 128, 255]]


def test_unknown_commands_blank_lines_and_missing_file(oracle):
    sc = oracle.Scene(text="\nFoo 1 2 3\nColor 1 0 0\n\nSphere 0 0 0 1\n")
    assert sc.prim_count == 1 and tuple(sc.flat_prims()[0].color) == (1, 0, 0)
    assert oracle.Scene("/nonexistent/objects.txt").prim_count == 0


def test_ray_count_identity(oracle):
    sc = oracle.Scene(scene_path("menger2.txt"))
    p = oracle.default_params(width=32, height=24, kernel=2)
    _, cnt = sc.render(p, threads=1)
    assert cnt["rays"] == 32 * 24 * 4 * (1 + 2)
    assert cnt["sdf_calls"] == cnt["march_steps"] + 6 * 32 * 24 * 4
