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


@pytest.mark.parametrize("seed,shift", [(31, 0.0), (32, 0.0), (33, 3000.0), (34, -70000.0)])
def test_spheres_of_the_root_behind_their_common_box_test(rm, oracle, renderer, seed, shift):
    """The root union's spheres and sphere-bounded lists are folded last, behind one test of the point against the box
    around them.  Points where that test is closest to wrong: the floor's distance equals the distance to the box's
    face, on the axis through a sphere's centre (the sphere's value equals the box excess there), a few ulps either
    side, and nearer by shares from 1e-6 to 1e-2; the whole scene also far from the origin, where roundings are largest.
    (Checked once against a build whose test accepted points 1e-4 too early: it fails here.)"""
    rng = np.random.default_rng(seed)
    zero = (0, 0, 0)

    def ball():
        return (SPHERE, 0, 0, (shift + rng.uniform(-30, 30), rng.uniform(0.5, 12), rng.uniform(-30, 30)), (rng.uniform(0.3, 6), 0, 0),
                rng.uniform(0.1, 1, 3))

    loose = [ball() for _ in range(int(rng.integers(2, 4)))]           # fewer than four: records of their own, not a run
    wrapped = [ball() for _ in range(int(rng.integers(1, 6)))]         # each the only child of a list
    balls = loose + wrapped
    star = [(LIST, DIFFERENCE, 2, zero, zero, zero), (SPHERE, 0, 0, (shift + 8.0, 5.0, -12.0), (5, 0, 0), (0, 0, 1)),
            (SPHERE, 0, 0, (shift + 11.0, 6.0, -12.0), (3, 0, 0), (0, 0, 1))]
    holed = [(LIST, DIFFERENCE, 3, zero, zero, zero), (BOX, 0, 0, (shift - 45.0, 10.0, 40.0), (20, 20, 20), (1, 1, 1)),
             (CROSS, 0, 0, (shift - 45.0, 10.0, 40.0), (6, 6, 6), (1, 1, 1)), (SPHERE, 0, 0, (shift - 45.0, 22.0, 40.0), (4, 0, 0), (1, 1, 1))]
    spec = [(LIST, UNION, 4 + len(balls), zero, zero, zero)]
    spec += [(LIST, COMPLEMENT, 1, zero, zero, zero), (BOX, 0, 0, (shift, 0, 0), (400, 400, 400), zero)]
    spec += [(BOX, 0, 0, (shift, -1.0, 0.0), (300, 2, 300), (0.5, 0.5, 0.5))]
    spec += loose + star + holed
    for w in wrapped:
        spec += [(LIST, UNION, 1, zero, zero, zero), w]

    class AsScene:
        def nodes(self):
            nodes = _nodes(rm.Node, spec)
            return (rm.Node * len(nodes))(*nodes)
    assert "group_wanted" in rm.kernel_source(AsScene())   # the scene does exercise the common test
    spread_pts = _points(rng, spec, 3072, 60.0)
    spread_pts[1024:] += np.float32([shift, 0, 0])   # the first third lies at the primitives, wherever they are
    # The test is made per wave (one lane that needs the spheres makes all evaluate them), so probe points of one kind
    # fill whole workgroups of the evaluation kernel (256 points): a wave that skips wrongly is then not covered
    # up by a neighbour that needs the spheres anyway.
    kinds = {}
    members = balls + [star[1]]
    for axis, sign in ((0, 1), (0, -1), (2, 1), (2, -1)):
        # the sphere that reaches furthest this way: the box's face is its extreme point, so along its axis the
        # sphere's value equals the excess over the box (elsewhere the point would lie inside the box)
        (_, _, _, c, size, _) = max(members, key=lambda m: sign * m[3][axis] + m[4][0])
        c = np.asarray(c, np.float64)
        q = c.astype(np.float32)
        q[axis] = np.float32(c[axis] + sign * (size[0] + c[1]))   # as far from the sphere as the floor is below
        for ulps in range(-6, 7):
            v = q.copy()
            for _ in range(abs(ulps)):
                v[axis] = np.nextafter(v[axis], np.float32(np.inf if ulps > 0 else -np.inf))
            kinds.setdefault(("ulps", ulps), []).append(v)
            w = v.copy()
            w[1] = np.nextafter(w[1], np.float32(np.inf if ulps % 2 else -np.inf))
            kinds.setdefault(("ulps", ulps), []).append(w)
        # and a little nearer or further: a test that skipped the spheres too early by any of these shares would
        # lose the sphere where it is the nearer surface
        for share in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2):
            for side in (-1, 1):
                v = q.copy()
                v[axis] = np.float32(c[axis] + sign * (size[0] + c[1] * (1.0 + side * share)))
                kinds.setdefault((share, side), []).append(v)
    pts = [spread_pts] + [np.resize(np.asarray(v, np.float32), (256, 3)) for v in kinds.values()]
    _check(rm, oracle, renderer, spec, np.concatenate(pts).astype(np.float32))


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


def test_frame_of_a_flat_scene_on_a_huge_ground_sphere(rm, oracle, renderer):
    """The upper bound the march loop hands the scene walker carries a margin for the rounding of the value it was
    derived from; when that value is a primitive's whose centre lies thousands of units away (a ground sphere of
    radius 5000), the margin has to grow with that primitive's magnitude, not only with the point's: a room, the
    ground sphere and 300 loose primitives just above it, frame bit-identical to the oracle's."""
    rng = np.random.default_rng(23)
    lines = ["Bounds 400.0", "Camera Position 30.0 12.0 30.0", "Camera Direction -1.0 -0.35 -1.0", "Camera Up 0.0 1.0 0.0",
             "Camera FOV 70", "Light 10.0 50.0 20.0", "Light -30.0 40.0 -10.0", "Color 0.4 0.5 0.3", "Sphere 0.0 -5000.0 0.0 5000.0"]
    for _ in range(300):
        t = ("Sphere", "Box", "Cross")[int(rng.integers(3))]
        c = rng.uniform([-40, 0.2, -40], [40, 6, 40])
        lines.append("Color %.3f %.3f %.3f" % tuple(rng.uniform(0, 1, 3)))
        if t == "Sphere":
            lines.append("Sphere %.4f %.4f %.4f %.4f" % (*c, rng.uniform(0.3, 2)))
        elif t == "Box":
            lines.append("Box %.4f %.4f %.4f %.4f %.4f %.4f" % (*c, *rng.uniform(0.3, 3, 3)))
        else:
            lines.append("Cross %.4f %.4f %.4f %.4f %.4f %.4f" % (*c, *rng.uniform(0.05, 0.4, 3)))
    text = "\n".join(lines) + "\n"
    kw = dict(width=72, height=48, kernel=2)
    renderer.set_params(rm.default_params(**kw))
    renderer.set_scene(rm.Scene(text=text))
    gpu = renderer.render_f32()
    cpu, _ = oracle.Scene(text=text).render(oracle.default_params(**kw))
    assert gpu.tobytes() == cpu.tobytes()
    rng = np.random.default_rng(24)
    pts = rng.uniform([-50, -0.5, -50], [50, 12, 50], (6000, 3)).astype(np.float32)
    sd, _, sd_march = renderer.eval_sdf_both(pts)
    osd, _ = oracle.Scene(text=text).sdf(pts)
    assert sd.tobytes() == osd.tobytes() and sd_march.tobytes() == osd.tobytes()
