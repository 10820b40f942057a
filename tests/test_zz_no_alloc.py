"""rm_enqueue_device_static: the render entry point that allocates nothing (its scratch -- the SSAA sample image
between the trace and the resolve kernel -- is a static device array).  Its frames must equal rm_render_device's
bit for bit, on two calls in a row, the second with a smaller input, in both kernel modes."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _frame(r, rm, p, static):
    import torch
    rows = rm.Rows(p.height, 0, 1, 1)
    f32 = torch.full((p.height, p.width, 4), -7.0, dtype=torch.float32, device="cuda")
    rgba = torch.full((p.height, p.width, 4), 99, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    if static:
        r.enqueue_device_static(rows, f32.data_ptr(), rgba.data_ptr())
        r.wait()
    else:
        r.render_device(rows, f32.data_ptr(), rgba.data_ptr())
    return f32.cpu().numpy(), rgba.cpu().numpy()


@pytest.mark.parametrize("mode", ["compiled", "interpreted"])
def test_static_scratch_render_equals_allocating_render(rm, mode):
    r = rm.Renderer(0)
    try:
        r.set_kernel_mode(mode == "compiled")
        r.set_scene(rm.Scene(scene_path("default.txt")))
        # kernel 3 and 5: the resolve kernel reads the sample image from the scratch in both modes
        sizes = [dict(width=640, height=480, kernel=3), dict(width=203, height=117, kernel=3), dict(width=96, height=64, kernel=5)]
        for kw in sizes:                         # static first: it must not lean on buffers an earlier render left
            p = rm.default_params(**kw)
            r.set_params(p)
            got = _frame(r, rm, p, True)
            assert r.counters()["kernel_ms"] > 0
            want = _frame(r, rm, p, False)
            assert got[0].tobytes() == want[0].tobytes(), kw
            assert got[1].tobytes() == want[1].tobytes(), kw
        # and after the allocating renders, which left a tile order behind: two static calls in a row, large then small
        for kw in sizes[:2]:
            p = rm.default_params(**kw)
            r.set_params(p)
            want = _frame(r, rm, p, False)
            got = _frame(r, rm, p, True)
            assert got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes(), kw
    finally:
        r.close()


def test_static_scratch_too_small_falls_back_to_pixels_from_the_trace_kernel(rm):
    """SSAA 3 on a frame whose sample image (78 MB) exceeds the scratch: the scene's own kernel then runs 3 x 3-wave
    workgroups that hold whole pixels and needs no sample image at all."""
    r = rm.Renderer(0)
    try:
        r.set_scene(rm.Scene(scene_path("default.txt")))
        p = rm.default_params(width=1200, height=900, kernel=3)
        assert p.width * p.height * 9 * 8 > rm.lib().rm_static_scratch_bytes()
        r.set_params(p)
        got = _frame(r, rm, p, True)
        assert r.counters()["resolve_ms"] < 0.01           # no resolve kernel ran
        want = _frame(r, rm, p, False)
        assert got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes()
        p = rm.default_params(width=1200, height=900, kernel=6)
        r.set_params(p)
        got = _frame(r, rm, p, True)
        want = _frame(r, rm, p, False)
        assert got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes()
    finally:
        r.close()


def test_static_scratch_limits_are_reported(rm):
    r = rm.Renderer(0)
    try:
        r.set_kernel_mode(False)
        r.set_scene(rm.Scene(scene_path("menger2.txt")))
        p = rm.default_params(width=1024, height=768, kernel=4)     # 12.6 M samples x 8 bytes > 64 MiB
        r.set_params(p)
        assert rm.lib().rm_static_scratch_bytes() == 64 << 20
        with pytest.raises(rm.RaymarchError, match="static buffer holds"):
            r.enqueue_device_static(rm.Rows(p.height, 0, 1, 1), None, None)
        # a share of the rows fits
        import torch
        part = torch.zeros((256, 1024, 4), dtype=torch.uint8, device="cuda")
        whole = torch.zeros((768, 1024, 4), dtype=torch.uint8, device="cuda")
        r.render_device(rm.Rows(768, 0, 1, 1), None, whole.data_ptr())
        r.enqueue_device_static(rm.Rows(256, 1, 1, 1), None, part.data_ptr())
        r.wait()
        assert torch.equal(part, whole[256:512])
        r.set_counting(True)
        with pytest.raises(rm.RaymarchError, match="counting"):
            r.enqueue_device_static(rm.Rows(256, 1, 1, 1), None, part.data_ptr())
    finally:
        r.close()
