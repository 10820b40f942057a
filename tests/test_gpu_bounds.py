"""Long primitive runs with bounding hulls, rejected spheres and bounded nested lists: the distance the march
loop uses must equal the exhaustive fold of Body::List::SDF (/root/reference/source/body.cpp:66-111) bit for
bit, wherever the point lies."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

LIST, SPHERE, BOX, CROSS = 0, 1, 2, 3
UNION, COMPLEMENT, INTERSECTION, DIFFERENCE = 0, 1, 2, 3


def _nodes(mod_node, spec):
    f3 = C.c_float * 3
    return [mod_node(t, m, n, f3(*c), f3(*s), f3(*col)) for (t, m, n, c, s, col) in spec]


def _prims(rng, n, kinds, spread, iso_share=0.5):
    out = []
    for _ in range(n):
        t = int(rng.choice(kinds))
        c = rng.uniform(-spread, spread, 3)
        if t == SPHERE:
            s = (rng.uniform(0.3, 3), 0, 0)
        elif rng.random() < iso_share:
            e = rng.uniform(0.3, 4)
            s = (e, e, e)
        else:
            s = rng.uniform(0.3, 4, 3)
        out.append((t, 0, 0, c, s, rng.uniform(0, 1, 3)))
    return out


def _points(rng, spec, n, spread):
    pts = rng.uniform(-1.3 * spread, 1.3 * spread, (n, 3)).astype(np.float32)
    # a third of the points on or just off primitive centres and faces: inside material, ties, sign changes
    prims = [s for s in spec if s[0] != LIST]
    for i in range(0, n // 3):
        t, _, _, c, s, _ = prims[int(rng.integers(len(prims)))]
        pts[i] = np.asarray(c, np.float32) + rng.normal(0, 1, 3).astype(np.float32) * np.float32(0.6 * max(s))
    return pts


def _check(rm, oracle, renderer, spec, pts):
    orc = oracle.Scene(nodes=_nodes(oracle.Node, spec))
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    renderer.set_scene_raw(_nodes(rm.Node, spec), np.zeros((0, 3), np.float32), ident, 1.0)
    sd, prim, sd_march = renderer.eval_sdf_both(pts)
    osd, orgb = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes()
    assert sd_march.tobytes() == osd.tobytes()
    colours = np.array([tuple(p.color) for p in orc.flat_prims()] + [(0, 0, 0)], np.float32)
    np.testing.assert_array_equal(colours[np.where(prim == 0xFFFFFFFF, len(colours) - 1, prim)], orgb)


@pytest.mark.parametrize("seed,n,mode", [(1, 40, UNION), (2, 300, UNION), (3, 1500, UNION), (4, 300, DIFFERENCE),
                                         (5, 300, INTERSECTION), (6, 300, COMPLEMENT), (7, 17, UNION)])
def test_long_runs_of_every_kind(rm, oracle, renderer, seed, n, mode):
    rng = np.random.default_rng(seed)
    spread = 40.0
    spec = [(LIST, mode, n, (0, 0, 0), (0, 0, 0), (0, 0, 0))] + _prims(rng, n, [SPHERE, BOX, CROSS], spread)
    _check(rm, oracle, renderer, spec, _points(rng, spec, 6000, spread))


def test_nested_lists_with_long_runs(rm, oracle, renderer):
    """A union of: a room (complement box), a difference list with 200 subtracted primitives, an intersection of
    a box with a union of 120 spheres, a single-child list and 90 loose boxes."""
    rng = np.random.default_rng(11)
    spread = 30.0
    zero = (0, 0, 0)
    sub = _prims(rng, 200, [SPHERE, BOX, CROSS], spread)
    balls = _prims(rng, 120, [SPHERE], spread)
    loose = _prims(rng, 90, [BOX], spread)
    spec = [(LIST, UNION, 4 + len(loose), zero, zero, zero)]
    spec += [(LIST, COMPLEMENT, 1, zero, zero, zero), (BOX, 0, 0, zero, (90, 90, 90), (1, 1, 1))]
    spec += [(LIST, DIFFERENCE, 1 + len(sub), zero, zero, zero), (BOX, 0, 0, (5, 0, 0), (50, 50, 50), (1, 0, 0))] + sub
    spec += [(LIST, INTERSECTION, 2, zero, zero, zero), (BOX, 0, 0, (-10, 5, 0), (30, 30, 30), (0, 1, 0)),
             (LIST, UNION, len(balls), zero, zero, zero)] + balls
    spec += [(LIST, UNION, 1, zero, zero, zero), (SPHERE, 0, 0, (20, 20, 20), (3, 0, 0), (0, 0, 1))]
    spec += loose
    _check(rm, oracle, renderer, spec, _points(rng, spec, 8000, spread))


def test_identical_and_degenerate_rows(rm, oracle, renderer):
    """Runs whose rows coincide (zero-extent bounding boxes, Morton keys all equal) or have zero size."""
    rng = np.random.default_rng(12)
    zero = (0, 0, 0)
    same = [(SPHERE, 0, 0, (1, 2, 3), (2, 0, 0), (0.1 * i % 1, 0, 0)) for i in range(40)]
    flat = [(BOX, 0, 0, (float(i), 0, 0), (0, 0, 0), (0, 1, 0)) for i in range(40)]
    spec = [(LIST, UNION, len(same) + len(flat), zero, zero, zero)] + same + flat
    pts = rng.uniform(-10, 45, (3000, 3)).astype(np.float32)
    pts[:50] = np.float32([1, 2, 3])
    _check(rm, oracle, renderer, spec, pts)


def test_synthetic_scene_points(rm, oracle, renderer):
    prod, orc = rm.Scene(scene_path("synthetic1024.txt")), oracle.Scene(scene_path("synthetic1024.txt"))
    renderer.set_scene(prod)
    rng = np.random.default_rng(13)
    pts = rng.uniform([-70, -5, -120], [70, 45, 20], (6000, 3)).astype(np.float32)
    sd, _, sd_march = renderer.eval_sdf_both(pts)
    osd, _ = orc.sdf(pts)
    assert sd.tobytes() == osd.tobytes() and sd_march.tobytes() == osd.tobytes()
    sd10, _ = renderer.eval_sdf(pts[:10])          # the plain entry point gives the same numbers
    assert sd10.tobytes() == osd[:10].tobytes()


@pytest.mark.parametrize("seed", [21, 22])
def test_frames_of_large_flat_scenes(rm, oracle, renderer, seed):
    """Whole frames (march loop, gradient taps, shadow rays) of a room with 400 loose primitives: the march loop
    hands the scene walker an upper bound of the next value (the SDF is 1-Lipschitz), which lets root-level runs
    skip groups before the accumulator has tightened; the frame must stay bit-identical."""
    rng = np.random.default_rng(seed)
    lines = ["Bounds 120.0", "Camera Position 30.0 20.0 30.0", "Camera Direction -1.0 -0.6 -1.0", "Camera Up 0.0 1.0 0.0",
             "Camera FOV 70", "Light 10.0 50.0 20.0", "Light -30.0 40.0 -10.0", "Box 0.0 -1.0 0.0 120.0 2.0 120.0"]
    for _ in range(400):
        t = ("Sphere", "Box", "Cross")[int(rng.integers(3))]
        c = rng.uniform([-40, 1, -40], [40, 25, 40])
        lines.append("Color %.3f %.3f %.3f" % tuple(rng.uniform(0, 1, 3)))
        if t == "Sphere":
            lines.append("Sphere %.4f %.4f %.4f %.4f" % (*c, rng.uniform(0.5, 3)))
        elif t == "Box":
            lines.append("Box %.4f %.4f %.4f %.4f %.4f %.4f" % (*c, *rng.uniform(0.5, 4, 3)))
        else:
            lines.append("Cross %.4f %.4f %.4f %.4f %.4f %.4f" % (*c, *rng.uniform(0.05, 0.4, 3)))
    text = "\n".join(lines) + "\n"
    kw = dict(width=72, height=48, kernel=2)
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    gpu = renderer.render_f32()
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()
