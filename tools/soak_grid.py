"""Random flat scenes of a few hundred to a few thousand loose spheres, boxes and crosses with random cameras and lights: the
frame of the kernel generated for the scene (cell grid) against the record walker's (hull walk).
python tools/soak_grid.py [scenes] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
kw = dict(width=96, height=64, kernel=2)
bad = 0
for sc in range(n_scenes):
    n = int(rng.choice([150, 400, 1024, 3000]))
    extent = float(rng.choice([10.0, 30.0, 80.0]))
    small, large = rng.choice([(0.05, 0.5), (0.3, 2.0), (1.0, 4.0)])
    lines = ["Bounds 300.0"]
    pos = rng.uniform(-1.2, 1.2, 3) * extent
    d = -pos + rng.normal(size=3) * 0.3 * extent
    d /= np.linalg.norm(d)
    lines += [f"Camera Position {pos[0]:.4f} {pos[1]:.4f} {pos[2]:.4f}", f"Camera Direction {d[0]:.4f} {d[1]:.4f} {d[2]:.4f}", "Camera Up 0.0 1.0 0.0", "Camera FOV 65"]
    for _ in range(int(rng.integers(1, 5))):
        q = rng.uniform(-2.0, 2.0, 3) * extent
        lines.append(f"Light {q[0]:.3f} {q[1]:.3f} {q[2]:.3f}")
    for i in range(n):
        if i % 97 == 0:
            c = rng.uniform(0.1, 1.0, 3)
            lines.append(f"Color {c[0]:.3f} {c[1]:.3f} {c[2]:.3f}")
        c = rng.uniform(-1.0, 1.0, 3) * extent
        kind = int(rng.integers(0, 3))
        if kind == 0:
            lines.append(f"Sphere {c[0]:.4f} {c[1]:.4f} {c[2]:.4f} {rng.uniform(small, large):.4f}")
        else:
            h = rng.uniform(small, large, 3)
            lines.append(f"{'Box' if kind == 1 else 'Cross'} {c[0]:.4f} {c[1]:.4f} {c[2]:.4f} {h[0]:.4f} {h[1]:.4f} {h[2]:.4f}")
    scene = rm.Scene(text="\n".join(lines) + "\n")
    frames = []
    for compiled in (True, False):
        r = rm.Renderer(0)
        r.set_kernel_mode(compiled)
        r.set_params(rm.default_params(**kw))
        r.set_scene(scene)
        frames.append(r.render_f32())
        if compiled:
            frames.append(r.render_f32())
        r.close()
    same = frames[0].tobytes() == frames[2].tobytes() and frames[1].tobytes() == frames[2].tobytes()
    bad += 0 if same else 1
    print(f"scene {sc}: {n} primitives, extent {extent}, sizes {small}-{large}: {'same' if same else 'DIFFERENT'}; lit pixels {int((frames[2][..., :3].sum(-1) > 0).sum())}", flush=True)
print("different frames:", bad)
sys.exit(1 if bad else 0)
