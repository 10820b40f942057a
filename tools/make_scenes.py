"""Writes the committed benchmark scenes that are not the reference's own file:

  scenes/closeup6.txt       the shipped scene with MengerSponge iterations 6 and a close-up camera
                            (step-count / divergence stress; 3 368 421 crosses)
  scenes/synthetic1024.txt  1024 primitives cycling Sphere / Box / Cross, 8 lights (seed 12345); sphere radii in [0.5, 2],
                            box sizes in [1, 4], cross bars 0.05-0.3 thick
  scenes/synthetic1024_wide.txt  the same with every size, the crosses' bars included, uniform in [1, 4]

Deterministic; run from the repo root:  python tools/make_scenes.py
"""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def closeup6():
    src = open(os.path.join(ROOT, "scenes", "default.txt")).read()
    src = src.replace("Camera Position 50.0 35.0 -2.5", "Camera Position 14.0 16.0 -36.0")
    src = src.replace("Camera Direction -1.0 -0.5 -1.0", "Camera Direction -1.0 -0.45 -1.0")
    src = src.replace("Camera FOV 75", "Camera FOV 40")
    src = src.replace("MengerSponge 0.0 10.0 -50.0 20.0 3", "MengerSponge 0.0 10.0 -50.0 20.0 6")
    return src


def synthetic(n=1024, seed=12345, wide=False):
    rnd = random.Random(seed)
    u = rnd.uniform
    out = ["Bounds 200.0", "", "Camera Position 50.0 35.0 -2.5", "Camera Direction -1.0 -0.5 -1.0",
           "Camera Up 0.0 1.0 0.0", "Camera FOV 75", ""]
    for _ in range(8):
        out.append("Light %.4f %.4f %.4f" % (u(-50, 50), u(40, 60), u(-60, 0)))
    out += ["", "Box 0.0 -1.0 -50.0 150.0 2.0 150.0", ""]
    for i in range(n):
        out.append("Color %.3f %.3f %.3f" % (u(0.1, 1), u(0.1, 1), u(0.1, 1)))
        c = (u(-60, 60), u(1, 30), u(-110, 10))
        kind = i % 3
        if kind == 0:
            out.append("Sphere %.4f %.4f %.4f %.4f" % (c + ((u(1, 4) if wide else u(0.5, 2)),)))
        elif kind == 1:
            out.append("Box %.4f %.4f %.4f %.4f %.4f %.4f" % (c + (u(1, 4), u(1, 4), u(1, 4))))
        else:
            # a cross is unbounded along its three bars: the first scene keeps the bars thin, the `wide` one does not
            # (every size uniform in [1, 4]: a thicket of bars that fills much of the room)
            lo, hi = (1, 4) if wide else (0.05, 0.3)
            out.append("Cross %.4f %.4f %.4f %.4f %.4f %.4f" % (c + (u(lo, hi), u(lo, hi), u(lo, hi))))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    open(os.path.join(ROOT, "scenes", "closeup6.txt"), "w").write(closeup6())
    open(os.path.join(ROOT, "scenes", "synthetic1024.txt"), "w").write(synthetic())
    open(os.path.join(ROOT, "scenes", "synthetic1024_wide.txt"), "w").write(synthetic(wide=True))
    print("written")
