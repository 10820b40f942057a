"""rm_multi_*: one frame over several ranks from one process (the native counterpart of bench.py's one-process-per-GPU
sharding).  A GPU box has one GPU, so the device list names it several times: the ranks store their rows straight into
the frame, or their shares are gathered by copies; the frame must be the frame one context renders (every pixel is independent: render::CPU's loops,
/root/reference/source/render.cpp:122-129), whatever the number of ranks and however the image's height cuts the bands."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _single(rm, scene, params):
    r = rm.Renderer(0)
    try:
        r.set_params(params)
        r.set_scene(scene)
        return r.render_rgba8()
    finally:
        r.close()


@pytest.mark.parametrize("ranks,kw", [(3, dict(width=200, height=150, kernel=3)), (1, dict(width=64, height=33, kernel=2)),
                                      (5, dict(width=97, height=77, kernel=2)), (8, dict(width=64, height=20, kernel=1)),
                                      (4, dict(width=33, height=3, kernel=4))])
@pytest.mark.parametrize("transport", ["copies", "direct"])
def test_frame_gathered_from_ranks_on_one_gpu_equals_single_launch(rm, oracle, ranks, kw, transport):
    scene = rm.Scene(scene_path("default.txt"))
    p = rm.default_params(**kw)
    want = _single(rm, scene, p)
    # left to itself the library takes direct stores here (one device: every rank can write the first one's memory)
    m = rm.MultiRenderer([0] * ranks, rm.MULTI_COPY) if transport == "copies" else rm.MultiRenderer([0] * ranks)
    try:
        assert m.ranks == ranks and m.transport == (rm.MULTI_COPY if transport == "copies" else rm.MULTI_DIRECT)
        m.set_params(p)
        m.set_scene(scene)
        for i in range(3):                       # rank after rank with the tiles timed, then all ranks at once in cost order
            got = m.render_rgba8()
            assert got.tobytes() == want.tobytes(), f"frame {i}"
        if ranks == 3:
            c = m.context(1).counters()
            assert c["tiles_by_cost"] == 1 and c["compiled_kernel"] == 1
            # another frame size on the same object: buffers and plan follow
            p2 = rm.default_params(width=120, height=91, kernel=1)
            m.set_params(p2)
            assert m.render_rgba8().tobytes() == _single(rm, scene, p2).tobytes()
    finally:
        m.close()
    cpu, _ = oracle.Scene(scene_path("default.txt")).render(oracle.default_params(**kw))
    np.testing.assert_array_equal(want, oracle.quantize(cpu))


def test_multi_into_device_memory_and_settings_per_rank(rm):
    import torch
    scene = rm.Scene(scene_path("menger2.txt"))
    p = rm.default_params(width=160, height=101, kernel=2)
    want = _single(rm, scene, p)
    m = rm.MultiRenderer([0, 0])
    try:
        for r in range(2):
            m.context(r).set_kernel_mode(False)          # the record walker on every rank
        m.set_params(p)
        m.set_scene(scene)
        out = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            out.zero_()
            m.render_device(out.data_ptr())
            assert out.cpu().numpy().tobytes() == want.tobytes()
        assert m.context(0).counters()["compiled_kernel"] == 0
    finally:
        m.close()


def test_rows_stored_at_their_place_in_the_frame(rm):
    """rm_set_row_placement(RM_ROWS_IN_FRAME): contexts with different bands fill one frame between them -- float and RGBA8
    outputs, cyclic bands and a band list, a height that cuts the last band short and leaves whole bands below the image
    (not stored: the words behind the frame stay as they were), the entry point that allocates nothing, and the kernels that
    write pixels themselves (k = 2), the resolve kernel (record walker) and the exact kernel's resolve (k = 3)."""
    import torch
    scene = rm.Scene(scene_path("default.txt"))
    for kw, compiled in ((dict(width=90, height=50, kernel=2), True), (dict(width=90, height=50, kernel=3), True),
                         (dict(width=70, height=45, kernel=3), False), (dict(width=40, height=9, kernel=1), True)):
        p = rm.default_params(**kw)
        single = rm.Renderer(0)
        single.set_kernel_mode(compiled)
        single.set_params(p)
        single.set_scene(scene)
        want8, want32 = single.render_rgba8(), single.render_f32()
        single.close()
        n, band_rows = 3, 4
        n_bands = (p.height + band_rows - 1) // band_rows
        per_rank = (n_bands + n - 1) // n
        guard = 4096
        f8 = torch.full((p.height * p.width * 4 + guard,), 7, dtype=torch.uint8, device="cuda")
        f32 = torch.full((p.height * p.width * 4 + guard,), -3.0, dtype=torch.float32, device="cuda")
        ctxs = [rm.Renderer(0) for _ in range(n)]
        try:
            for r, c in enumerate(ctxs):
                c.set_kernel_mode(compiled)
                c.set_row_placement(True)
                c.set_params(p)
                c.set_scene(scene)
            for mode in ("cyclic", "list", "static"):
                f8.fill_(7)
                f32.fill_(-3.0)
                torch.cuda.synchronize()
                for r, c in enumerate(ctxs):
                    if mode == "list":                       # rank r takes the bands r, r + n, ... in falling order
                        c.set_band_list([(r + b * n) for b in range(per_rank)][::-1])
                        rows = rm.Rows(band_rows, 0, 1, per_rank)
                    else:
                        c.set_band_list(None)
                        rows = rm.Rows(band_rows, r, n, per_rank)
                    if mode == "static":
                        c.enqueue_device_static(rows, f32.data_ptr(), f8.data_ptr())
                        c.wait()
                    else:
                        c.render_device(rows, f32.data_ptr(), f8.data_ptr())
                got8, got32 = f8.cpu().numpy(), f32.cpu().numpy()
                assert got8[:-guard].tobytes() == want8.tobytes(), (kw, mode)
                assert got32[:-guard].tobytes() == want32.tobytes(), (kw, mode)
                assert (got8[-guard:] == 7).all() and (got32[-guard:] == -3.0).all(), (kw, mode)
            # back to a compact image of the call's bands
            ctxs[1].set_band_list(None)
            ctxs[1].set_row_placement(False)
            share = torch.zeros((per_rank * band_rows, p.width, 4), dtype=torch.uint8, device="cuda")
            ctxs[1].render_device(rm.Rows(band_rows, 1, n, per_rank), None, share.data_ptr())
            first = share.cpu().numpy()[:band_rows]
            assert first.tobytes() == want8[band_rows:2 * band_rows].tobytes()
        finally:
            for c in ctxs:
                c.close()


def test_multi_argument_errors(rm):
    with pytest.raises(rm.RaymarchError, match="names a device twice"):
        rm.MultiRenderer([0, 0], rm.MULTI_RCCL)
    with pytest.raises(rm.RaymarchError, match="no HIP device"):
        rm.MultiRenderer([0, 99])
    m = rm.MultiRenderer([0])
    try:
        with pytest.raises(rm.RaymarchError, match="rm_multi_set_params has not been called"):
            m.render_rgba8(out=np.zeros(4, np.uint8).ctypes.data)
    finally:
        m.close()


def test_multi_frames_in_flight(rm):
    """rm_multi_enqueue_* / rm_multi_wait: two multi-rank objects over the same GPU take turns, frames into device memory and
    into page-locked host memory; every frame equals the single launch's; a second enqueue without a wait is refused, and
    the setters wait for the frame in flight."""
    import torch
    p = rm.default_params(width=203, height=117, kernel=3)
    scene = rm.Scene(scene_path("default.txt"))
    r = rm.Renderer(0)
    r.set_params(p)
    r.set_scene(scene)
    want = r.render_rgba8()
    r.close()
    ms = [rm.MultiRenderer([0, 0, 0]) for _ in range(2)]
    host = [rm.HostBuffer((p.height, p.width, 4), np.uint8) for _ in range(2)]
    try:
        for m in ms:
            m.set_params(p)
            m.set_scene(scene)
        dev = [torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for i in range(8):                        # the first frame of each object is its synchronous, tile-timing one
            m = ms[i % 2]
            m.wait()
            if i >= 2:
                assert dev[i % 2].cpu().numpy().tobytes() == want.tobytes(), i
                dev[i % 2].zero_()
            m.enqueue_device(dev[i % 2].data_ptr())
        with pytest.raises(rm.RaymarchError):
            ms[0].enqueue_device(dev[0].data_ptr())          # one frame at a time per object
        for m in ms:
            m.wait()
        for i in range(4):
            m = ms[i % 2]
            m.wait()
            if i >= 2:
                assert host[i % 2].array.tobytes() == want.tobytes(), i
            host[i % 2].array[...] = 0
            m.enqueue_rgba8(host[i % 2].ptr)
        ms[0].set_params(p)                                   # waits for the frame in flight
        assert host[0].array.tobytes() == want.tobytes()
        ms[1].wait()
        assert host[1].array.tobytes() == want.tobytes()
        ms[0].set_scene(scene)
        assert ms[0].render_rgba8().tobytes() == want.tobytes()
    finally:
        for m in ms:
            m.close()
        for h in host:
            h.free()
