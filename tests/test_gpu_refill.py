"""The lane-refill kernel (rm_set_lane_refill: a wave owns a tile of samples, a lane whose ray has ended takes the next piece
of work): same frames and same work counts as the oracle, whatever the tile shape."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", [(3, 3), (4, 4), (5, 3)], ids=["8x8", "16x16", "32x8"])
def test_refill_frames_and_counts_equal_the_oracle(rm, oracle, tile):
    r = rm.Renderer(0)
    try:
        r.set_lane_refill(*tile)
        for name, kw in [("default.txt", dict(width=150, height=85, kernel=3)), ("menger2.txt", dict(width=96, height=72, kernel=2)),
                         ("synthetic1024.txt", dict(width=40, height=30, kernel=1))]:
            r.set_params(rm.default_params(**kw))
            r.set_scene(rm.Scene(scene_path(name)))
            cpu, oc = oracle.Scene(scene_path(name)).render(oracle.default_params(**kw))
            for _ in range(2):                                   # natural tile order, then cost order
                assert r.render_f32().tobytes() == cpu.tobytes(), (name, tile)
            np.testing.assert_array_equal(r.render_rgba8(), oracle.quantize(cpu))
            r.set_counting(True)
            r.set_shading(False)                                 # every sample does the reference's full work
            try:
                assert r.render_f32().tobytes() == cpu.tobytes()
                c = r.counters()
            finally:
                r.set_counting(False)
                r.set_shading(True)
            assert (c["rays"], c["march_steps"], c["sdf_calls"]) == (oc["rays"], oc["march_steps"], oc["sdf_calls"])
            assert c["march_slots"] >= c["march_steps"]
    finally:
        r.close()


def test_refill_arguments_are_checked(rm):
    r = rm.Renderer(0)
    try:
        with pytest.raises(rm.RaymarchError, match="tile sides"):
            r.set_lane_refill(2, 4)
        r.set_lane_refill(0, 0)
    finally:
        r.close()
