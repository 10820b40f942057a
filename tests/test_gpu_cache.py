"""The cache directory: compiled kernels and measured tile orders survive the context (and the process) that made them."""
import glob
import os

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def test_tile_order_and_kernels_come_back_from_the_cache_directory(rm, tmp_path):
    rm.set_kernel_cache_dir(str(tmp_path))
    try:
        p = rm.default_params(width=333, height=222, kernel=2)
        scene = rm.Scene(scene_path("default.txt"))
        a = rm.Renderer(0)
        a.set_params(p)
        a.set_scene(scene)
        first = a.render_f32()
        assert a.counters()["tiles_by_cost"] == 0            # nothing known: natural order, tiles timed
        second = a.render_f32()
        ca = a.counters()
        assert ca["tiles_by_cost"] == 1
        a.close()
        files = glob.glob(os.path.join(str(tmp_path), "rm_*.gfx950.bin"))
        assert len(files) >= 1                               # the tile order (and the code object, unless this process had compiled it before)
        b = rm.Renderer(0)                                   # what a later process would do
        b.set_params(p)
        b.set_scene(scene)
        again = b.render_f32()
        cb = b.counters()
        assert cb["tiles_by_cost"] == 1 and cb["tile_parts"] == ca["tile_parts"]   # first frame already in the measured order
        assert again.tobytes() == first.tobytes() == second.tobytes()
        # rm_reserve clears the tile tables: the order is read again, whichever of the two calls comes last,
        # and a reserve between two frames costs a frame in natural order at most, never a wrong one
        for reserve_first in (True, False):
            d = rm.Renderer(0)
            d.set_params(p)
            if reserve_first:
                d.reserve()
            d.set_scene(scene)
            if not reserve_first:
                d.reserve()
            assert d.render_f32().tobytes() == first.tobytes()
            assert d.counters()["tiles_by_cost"] == 1
            d.reserve()
            assert d.render_f32().tobytes() == first.tobytes()
            assert d.render_rgba8().tobytes() == b.render_rgba8().tobytes()
            d.close()
        # another camera is another configuration
        moved = rm.Scene(text=open(scene_path("default.txt")).read().replace("Camera Position 50.0", "Camera Position 49.0"))
        b.set_scene(moved)
        b.render_f32()
        assert b.counters()["tiles_by_cost"] == 0
        b.close()
        # a damaged order file is not trusted: the frame is timed again and stays the same
        order_file = min(files, key=os.path.getsize)           # 4704 entries; the code objects are far larger
        assert os.path.getsize(order_file) < 40000
        data = bytearray(open(order_file, "rb").read())
        data[-8:] = b"\xff" * 8                                # two entries that name no tile
        open(order_file, "wb").write(bytes(data))
        c = rm.Renderer(0)
        c.set_params(p)
        c.set_scene(scene)
        assert c.render_f32().tobytes() == first.tobytes()
        c.close()
    finally:
        rm.set_kernel_cache_dir(os.environ["RM_KERNEL_CACHE_DIR"])
