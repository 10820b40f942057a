"""Random cameras in and around Menger sponges: the frame of the kernel generated for the scene against the record walker's
(exhaustive evaluator for iterations 4, its own lattice walk for 5 and 6).  python tools/soak_lattice.py [cameras per sponge] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raymarching_amd as rm  # noqa: E402

n_cameras = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
kw = dict(width=96, height=64, kernel=2)
bad = 0
for iterations, exhaustive in ((4, True), (5, False), (6, False)):
    for c in range(n_cameras):
        # positions from well outside the sponge (centre (0, 10, -50), size 20) to inside its tunnels
        pos = np.array([0.0, 10.0, -50.0]) + rng.uniform(-1.0, 1.0, 3) * rng.choice([4.0, 9.0, 14.0, 25.0])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        text = "\n".join([
            "Bounds 200.0", f"Camera Position {pos[0]:.4f} {pos[1]:.4f} {pos[2]:.4f}", f"Camera Direction {d[0]:.4f} {d[1]:.4f} {d[2]:.4f}",
            "Camera Up 0.0 1.0 0.0", "Camera FOV 70", "Light -15.0 40.0 -20.0", "Light 45.0 50.0 -35.0",
            "Box 0.0 -1.0 -50.0 75.0 2.0 75.0", "Color 1.0 1.0 1.0", f"MengerSponge 0.0 10.0 -50.0 20.0 {iterations}"]) + "\n"
        scene = rm.Scene(text=text)
        frames = []
        for compiled in (True, False):
            r = rm.Renderer(0)
            r.set_kernel_mode(compiled)
            r.set_evaluator(True if compiled else not exhaustive)
            r.set_params(rm.default_params(**kw))
            r.set_scene(scene)
            frames.append(r.render_f32())
            if compiled:
                frames.append(r.render_f32())       # in cost order, parts and all
                again = r.exact_tiles()
            r.close()
        same = frames[0].tobytes() == frames[2].tobytes() and frames[1].tobytes() == frames[2].tobytes()
        bad += 0 if same else 1
        print(f"iterations {iterations} camera {c}: {'same' if same else 'DIFFERENT'}; tiles rendered again {again}; lit pixels {int((frames[2][..., :3].sum(-1) > 0).sum())}", flush=True)
print("different frames:", bad)
sys.exit(1 if bad else 0)
