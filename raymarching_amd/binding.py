"""ctypes declarations for include/raymarch_hip.h and thin owning wrappers."""
import ctypes as C
import os

import numpy as np

# RM_LIB_PATH: another build of the same library (A/B timing of two builds on one GPU box)
LIB_PATH = os.environ.get("RM_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libraymarch_hip.so")

# every symbol include/raymarch_hip.h declares
EXPORTS = (
    "rm_create", "rm_destroy", "rm_set_params", "rm_reserve", "rm_set_scene", "rm_render_rgba8", "rm_render_f32",
    "rm_render_device", "rm_enqueue_host", "rm_host_alloc", "rm_host_free", "rm_host_register", "rm_host_unregister", "rm_set_band_list", "rm_set_row_placement", "rm_enqueue_device", "rm_enqueue_device_static", "rm_static_scratch_bytes", "rm_wait", "rm_set_stream", "rm_eval_sdf", "rm_eval_sdf_both", "rm_kernel_source", "rm_kernel_source_compiles", "rm_kernel_compiler", "rm_precompile_scene", "rm_set_kernel_cache_dir", "rm_set_evaluator", "rm_set_kernel_mode", "rm_set_tile_scheduling", "rm_set_shading", "rm_set_lane_refill", "rm_get_tile_costs", "rm_set_counting", "rm_get_counters", "rm_set_slot_stats", "rm_get_slot_stats", "rm_get_exact_tiles", "rm_stream", "rm_last_error",
    "rm_scene_load", "rm_scene_parse", "rm_scene_free", "rm_scene_node_count", "rm_scene_light_count",
    "rm_scene_nodes", "rm_scene_lights", "rm_scene_camera", "rm_set_scene_from", "rm_write_png",
    "rm_multi_create", "rm_multi_destroy", "rm_multi_set_params", "rm_multi_set_scene", "rm_multi_set_scene_from",
    "rm_multi_render_rgba8", "rm_multi_render_device", "rm_multi_enqueue_rgba8", "rm_multi_enqueue_device", "rm_multi_wait", "rm_multi_context", "rm_multi_ranks", "rm_multi_transport_in_use",
)


class RaymarchError(RuntimeError):
    pass


class Node(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mode", C.c_uint32), ("child_count", C.c_uint32),
                ("center", C.c_float * 3), ("size", C.c_float * 3), ("color", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("kernel", C.c_uint32),
                ("iterations", C.c_int32), ("saturation", C.c_float),
                ("surface_eps", C.c_float), ("offset_eps", C.c_float)]


class Rows(C.Structure):
    _fields_ = [("band_rows", C.c_uint32), ("first_band", C.c_uint32),
                ("band_stride", C.c_uint32), ("band_count", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("sdf_calls", C.c_uint64), ("march_steps", C.c_uint64), ("rays", C.c_uint64),
                ("prims", C.c_uint64), ("prim_evals", C.c_uint64), ("lattice_prims", C.c_uint64),
                ("flops_per_sdf", C.c_uint64), ("kernel_ms", C.c_double), ("resolve_ms", C.c_double),
                ("compile_ms", C.c_double), ("compiled_kernel", C.c_uint32), ("tiles_by_cost", C.c_uint32), ("march_slots", C.c_uint64), ("flops_executed", C.c_uint64), ("tile_parts", C.c_uint64), ("probe", C.c_uint64 * 16), ("flops_shared_first", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["probe"] = list(d["probe"])
        return d


_lib = None


def lib():
    """Loads libraymarch_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # A process must hold ONE HIP runtime.  PyTorch ships its own libamdhip64 (SONAME
    # libamdhip64.so.7); when torch is imported first, the dynamic loader resolves this library's
    # dependency of the same SONAME to that copy, and tensors, streams and RCCL buffers can be
    # shared with the renderer.  Loaded the other way round the process ends up with two runtimes
    # and torch finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RaymarchError(
            f"{LIB_PATH} is missing: build it with `make -C raymarching_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rm_last_error.restype = C.c_char_p
    L.rm_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rm_destroy.argtypes = [vp]
    L.rm_set_params.argtypes = [vp, C.POINTER(Params)]
    L.rm_reserve.argtypes = [vp]
    L.rm_set_scene.argtypes = [vp, C.POINTER(Node), C.c_uint32, vp, C.c_uint32, vp, C.c_float]
    L.rm_render_rgba8.argtypes = [vp, vp]
    L.rm_render_f32.argtypes = [vp, vp]
    L.rm_render_device.argtypes = [vp, C.POINTER(Rows), vp, vp]
    L.rm_enqueue_device.argtypes = [vp, C.POINTER(Rows), vp, vp]
    L.rm_enqueue_host.argtypes = [vp, C.POINTER(Rows), vp, vp]
    L.rm_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.rm_host_free.argtypes = [vp]
    L.rm_host_register.argtypes = [vp, C.c_size_t]
    L.rm_host_unregister.argtypes = [vp]
    L.rm_enqueue_device_static.argtypes = [vp, C.POINTER(Rows), vp, vp]
    L.rm_static_scratch_bytes.restype = C.c_size_t
    L.rm_wait.argtypes = [vp]
    L.rm_set_band_list.argtypes = [vp, vp, C.c_uint32]
    L.rm_set_row_placement.argtypes = [vp, C.c_int]
    L.rm_set_stream.argtypes = [vp, vp]
    L.rm_eval_sdf.argtypes = [vp, vp, C.c_uint32, vp, vp]
    L.rm_eval_sdf_both.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
    L.rm_set_counting.argtypes = [vp, C.c_int]
    L.rm_set_evaluator.argtypes = [vp, C.c_int]
    L.rm_kernel_source.argtypes = [vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rm_kernel_source_compiles.argtypes = [C.c_char_p]
    L.rm_kernel_compiler.restype = C.c_char_p
    L.rm_precompile_scene.argtypes = [vp, C.c_uint32, C.c_int]
    L.rm_set_kernel_cache_dir.argtypes = [C.c_char_p]
    L.rm_set_kernel_mode.argtypes = [vp, C.c_int]
    L.rm_set_tile_scheduling.argtypes = [vp, C.c_int]
    L.rm_set_shading.argtypes = [vp, C.c_int]
    L.rm_set_lane_refill.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.rm_get_tile_costs.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rm_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.rm_set_slot_stats.argtypes = [vp, C.c_int]
    L.rm_get_exact_tiles.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rm_get_slot_stats.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rm_stream.argtypes = [vp]
    L.rm_stream.restype = vp
    L.rm_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rm_scene_parse.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rm_scene_free.argtypes = [vp]
    L.rm_scene_free.restype = None
    L.rm_scene_node_count.argtypes = [vp]
    L.rm_scene_node_count.restype = C.c_uint32
    L.rm_scene_light_count.argtypes = [vp]
    L.rm_scene_light_count.restype = C.c_uint32
    L.rm_scene_nodes.argtypes = [vp, C.POINTER(Node), C.c_uint32]
    L.rm_scene_lights.argtypes = [vp, vp, C.c_uint32]
    L.rm_scene_camera.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.rm_set_scene_from.argtypes = [vp, vp]
    L.rm_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rm_multi_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.c_int, C.POINTER(vp)]
    L.rm_multi_destroy.argtypes = [vp]
    L.rm_multi_set_params.argtypes = [vp, C.POINTER(Params)]
    L.rm_multi_set_scene.argtypes = [vp, C.POINTER(Node), C.c_uint32, vp, C.c_uint32, vp, C.c_float]
    L.rm_multi_set_scene_from.argtypes = [vp, vp]
    L.rm_multi_render_rgba8.argtypes = [vp, vp]
    L.rm_multi_render_device.argtypes = [vp, vp]
    L.rm_multi_enqueue_rgba8.argtypes = [vp, vp]
    L.rm_multi_enqueue_device.argtypes = [vp, vp]
    L.rm_multi_wait.argtypes = [vp]
    L.rm_multi_context.argtypes = [vp, C.c_uint32]
    L.rm_multi_context.restype = vp
    L.rm_multi_ranks.argtypes = [vp]
    L.rm_multi_ranks.restype = C.c_uint32
    L.rm_multi_transport_in_use.argtypes = [vp]
    _lib = L
    return L


def exported_symbols():
    """Names from EXPORTS the loaded library really exports."""
    L = lib()
    return [n for n in EXPORTS if hasattr(L, n)]


def _check(rc, what):
    if rc != 0:
        raise RaymarchError(f"{what} failed ({rc}): {lib().rm_last_error().decode(errors='replace')}")


def default_params(**over):
    """The reference's constants.h values."""
    p = Params(1024, 768, 3, 1000, 0.05, 1e-3, 1e-3)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Scene:
    """Host scene parsed by the library's C++ loader (scene::load)."""

    def __init__(self, path=None, text=None):
        self._h = C.c_void_p()
        if text is not None:
            _check(lib().rm_scene_parse(text.encode(), C.byref(self._h)), "rm_scene_parse")
        else:
            _check(lib().rm_scene_load(os.fsencode(path), C.byref(self._h)), "rm_scene_load")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rm_scene_free(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def nodes(self):
        n = lib().rm_scene_node_count(self._h)
        arr = (Node * n)()
        _check(lib().rm_scene_nodes(self._h, arr, n), "rm_scene_nodes")
        return arr

    def lights(self):
        n = lib().rm_scene_light_count(self._h)
        out = np.zeros((n, 3), np.float32)
        _check(lib().rm_scene_lights(self._h, out.ctypes.data, n), "rm_scene_lights")
        return out

    def camera(self):
        t = np.zeros(16, np.float32)
        f = C.c_float()
        _check(lib().rm_scene_camera(self._h, t.ctypes.data, C.byref(f)), "rm_scene_camera")
        return t, np.float32(f.value)


def kernel_source(scene, lattice=True, counting=False):
    """HIP source of the kernels rm_set_scene generates for `scene` (a Scene); needs no GPU."""
    arr = scene.nodes()
    n = C.c_size_t()
    _check(lib().rm_kernel_source(arr, len(arr), 1 if lattice else 0, int(counting), None, 0, C.byref(n)), "rm_kernel_source")
    buf = C.create_string_buffer(n.value + 1)
    _check(lib().rm_kernel_source(arr, len(arr), 1 if lattice else 0, int(counting), buf, n.value + 1, C.byref(n)), "rm_kernel_source")
    return buf.value.decode()


def precompile(scene, lattice=True):
    """Starts compiling the kernels a Renderer.set_scene(scene) will want and returns at once (rm_precompile_scene)."""
    arr = scene.nodes()
    _check(lib().rm_precompile_scene(arr, len(arr), 1 if lattice else 0), "rm_precompile_scene")


def set_kernel_cache_dir(path):
    """Directory for compiled code objects shared between processes (None: off)."""
    _check(lib().rm_set_kernel_cache_dir(os.fsencode(path) if path else None), "rm_set_kernel_cache_dir")


def kernel_compiler():
    """How this process last obtained a scene's code object (rm_kernel_compiler)."""
    return lib().rm_kernel_compiler().decode()


def kernel_source_compiles(source):
    """Raises RaymarchError with the compiler log unless `source` compiles for gfx950 (hiprtc; needs no GPU)."""
    _check(lib().rm_kernel_source_compiles(source.encode()), "rm_kernel_source_compiles")


class Renderer:
    """Owning wrapper of an rm_context."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().rm_create(device, C.byref(self._h)), "rm_create")
        self.params = None

    def close(self):
        if getattr(self, "_h", None):
            lib().rm_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None and _lib is not None:   # module globals are gone at interpreter exit
            self.close()

    def set_params(self, params):
        _check(lib().rm_set_params(self._h, C.byref(params)), "rm_set_params")
        self.params = params

    def reserve(self):
        _check(lib().rm_reserve(self._h), "rm_reserve")

    def set_scene(self, scene):
        _check(lib().rm_set_scene_from(self._h, scene.handle), "rm_set_scene_from")

    def set_scene_raw(self, nodes, lights, transform, focal):
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 3)
        transform = np.ascontiguousarray(transform, np.float32)
        arr = (Node * len(nodes))(*nodes) if not isinstance(nodes, C.Array) else nodes
        _check(lib().rm_set_scene(self._h, arr, len(arr), lights.ctypes.data, len(lights),
                                  transform.ctypes.data, C.c_float(focal)), "rm_set_scene")

    def render_f32(self):
        p = self.params
        out = np.zeros((p.height, p.width, 4), np.float32)
        _check(lib().rm_render_f32(self._h, out.ctypes.data), "rm_render_f32")
        return out

    def render_rgba8(self):
        p = self.params
        out = np.zeros((p.height, p.width, 4), np.uint8)
        _check(lib().rm_render_rgba8(self._h, out.ctypes.data), "rm_render_rgba8")
        return out

    def render_device(self, rows, f32_ptr=None, rgba8_ptr=None):
        """rows: Rows; pointers are raw device addresses (e.g. torch.Tensor.data_ptr())."""
        _check(lib().rm_render_device(self._h, C.byref(rows), f32_ptr, rgba8_ptr), "rm_render_device")

    def enqueue_device(self, rows, f32_ptr=None, rgba8_ptr=None):
        """render_device without waiting; pair with wait()."""
        _check(lib().rm_enqueue_device(self._h, C.byref(rows), f32_ptr, rgba8_ptr), "rm_enqueue_device")

    def enqueue_host(self, rows, f32_ptr=None, rgba8_ptr=None):
        """Queues a frame whose finished rows are copied into host memory at the given addresses; pair with wait()."""
        _check(lib().rm_enqueue_host(self._h, C.byref(rows), f32_ptr, rgba8_ptr), "rm_enqueue_host")

    def enqueue_device_static(self, rows, f32_ptr=None, rgba8_ptr=None):
        """enqueue_device through the library's static scratch: the call allocates nothing."""
        _check(lib().rm_enqueue_device_static(self._h, C.byref(rows), f32_ptr, rgba8_ptr), "rm_enqueue_device_static")

    def set_band_list(self, image_bands):
        """image_bands: sequence of image band indices this context renders (None or empty: arithmetic bands)."""
        arr = np.ascontiguousarray(image_bands if image_bands is not None else [], np.uint32)
        _check(lib().rm_set_band_list(self._h, arr.ctypes.data if len(arr) else None, len(arr)), "rm_set_band_list")

    def set_row_placement(self, in_frame):
        """in_frame: the device-buffer renders take whole frames and store every rendered row at its own place in them
        (contexts with different bands then fill one frame between them); False: a compact image of the call's bands."""
        _check(lib().rm_set_row_placement(self._h, 1 if in_frame else 0), "rm_set_row_placement")

    def wait(self):
        _check(lib().rm_wait(self._h), "rm_wait")

    def set_stream(self, hip_stream):
        """hip_stream: raw hipStream_t (e.g. torch.cuda.Stream().cuda_stream) or None for the context's own."""
        _check(lib().rm_set_stream(self._h, hip_stream), "rm_set_stream")

    def eval_sdf(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        sd = np.zeros(len(xyz), np.float32)
        prim = np.zeros(len(xyz), np.uint32)
        _check(lib().rm_eval_sdf(self._h, xyz.ctypes.data, len(xyz), sd.ctypes.data, prim.ctypes.data), "rm_eval_sdf")
        return sd, prim

    def eval_sdf_both(self, xyz):
        """(sd, prim, sd_march): the colour-tracking evaluator's result and the distance-only evaluator's."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        sd = np.zeros(len(xyz), np.float32)
        sd_march = np.zeros(len(xyz), np.float32)
        prim = np.zeros(len(xyz), np.uint32)
        _check(lib().rm_eval_sdf_both(self._h, xyz.ctypes.data, len(xyz), sd.ctypes.data, prim.ctypes.data, sd_march.ctypes.data),
               "rm_eval_sdf_both")
        return sd, prim, sd_march

    def set_kernel_mode(self, compiled):
        """True (default): kernels generated and compiled per scene; False: the record-walking kernel.
        Takes effect at the next set_scene."""
        _check(lib().rm_set_kernel_mode(self._h, 1 if compiled else 0), "rm_set_kernel_mode")

    def set_tile_scheduling(self, by_cost):
        """True (default): after the first frame of a configuration, expensive tiles are started first."""
        _check(lib().rm_set_tile_scheduling(self._h, 1 if by_cost else 0), "rm_set_tile_scheduling")

    def set_shading(self, coloured_only):
        """True (default): samples on colourless surfaces skip the gradient taps and shadow rays (their product
        with the lighting is zero) and shadow rays stop once past their light; False: every sample and ray
        does the reference's full work."""
        _check(lib().rm_set_shading(self._h, 1 if coloured_only else 0), "rm_set_shading")

    def set_lane_refill(self, tile_w_log2, tile_h_log2):
        """Tiles of 2^w x 2^h samples whose lanes take the next piece of work when their ray has ended (0, 0: off, the
        default; measured slower).  Applies to the next set_scene."""
        _check(lib().rm_set_lane_refill(self._h, tile_w_log2, tile_h_log2), "rm_set_lane_refill")

    def tile_costs(self):
        """[tiles_y, tiles_x] uint32 durations (10 ns ticks) of the last render that timed its tiles."""
        tx, ty = C.c_uint32(), C.c_uint32()
        _check(lib().rm_get_tile_costs(self._h, None, 0, C.byref(tx), C.byref(ty)), "rm_get_tile_costs")
        out = np.zeros((ty.value, tx.value), np.uint32)
        _check(lib().rm_get_tile_costs(self._h, out.ctypes.data, out.size, C.byref(tx), C.byref(ty)), "rm_get_tile_costs")
        return out

    def set_evaluator(self, lattice):
        """True: recognise Menger lattices (default); False: visit every primitive.  Applies to the next set_scene."""
        _check(lib().rm_set_evaluator(self._h, 1 if lattice else 0), "rm_set_evaluator")

    def set_counting(self, enabled):
        _check(lib().rm_set_counting(self._h, int(bool(enabled))), "rm_set_counting")

    def exact_tiles(self):
        """Tiles the scene's exact kernel has rendered again since set_scene (rm_get_exact_tiles)."""
        n = C.c_uint64(0)
        _check(lib().rm_get_exact_tiles(self._h, C.byref(n)), "rm_get_exact_tiles")
        return int(n.value)

    def set_slot_stats(self, enabled):
        _check(lib().rm_set_slot_stats(self._h, int(bool(enabled))), "rm_set_slot_stats")

    def slot_stats(self):
        """[slots, 4] uint32 of the last render: tile-order entry, duration (10 ns ticks), camera-march | shadow-march << 16
        turns, start (low 32 bits of the 100 MHz clock)."""
        n = C.c_size_t()
        _check(lib().rm_get_slot_stats(self._h, None, 0, C.byref(n)), "rm_get_slot_stats")
        out = np.zeros((n.value, 4), np.uint32)
        _check(lib().rm_get_slot_stats(self._h, out.ctypes.data, n.value, C.byref(n)), "rm_get_slot_stats")
        return out

    def counters(self):
        c = Counters()
        _check(lib().rm_get_counters(self._h, C.byref(c)), "rm_get_counters")
        return c.as_dict()

    @property
    def stream(self):
        return lib().rm_stream(self._h)


class _BorrowedRenderer(Renderer):
    """A rank's context of a MultiRenderer: settings and counters; the MultiRenderer owns it."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self.params = None

    def close(self):
        self._h = None


MULTI_AUTO, MULTI_RCCL, MULTI_COPY, MULTI_DIRECT = 0, 1, 2, 3


class MultiRenderer:
    """Owning wrapper of an rm_multi: one frame over the listed devices (a device may be listed more than once)."""

    def __init__(self, devices, transport=MULTI_AUTO):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().rm_multi_create(arr, len(devices), transport, C.byref(self._h)), "rm_multi_create")
        self.params = None

    def close(self):
        if getattr(self, "_h", None):
            lib().rm_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None and _lib is not None:
            self.close()

    @property
    def ranks(self):
        return lib().rm_multi_ranks(self._h)

    @property
    def transport(self):
        return lib().rm_multi_transport_in_use(self._h)

    def context(self, rank):
        h = lib().rm_multi_context(self._h, rank)
        if not h:
            raise RaymarchError(f"rm_multi_context: no rank {rank}")
        return _BorrowedRenderer(h)

    def set_params(self, params):
        _check(lib().rm_multi_set_params(self._h, C.byref(params)), "rm_multi_set_params")
        self.params = params

    def set_scene(self, scene):
        _check(lib().rm_multi_set_scene_from(self._h, scene.handle), "rm_multi_set_scene_from")

    def render_rgba8(self, out=None):
        """The frame in host memory: a new numpy array, or the page-locked / ordinary buffer at address `out`."""
        p = self.params
        if out is not None:
            _check(lib().rm_multi_render_rgba8(self._h, out), "rm_multi_render_rgba8")
            return None
        img = np.zeros((p.height, p.width, 4), np.uint8)
        _check(lib().rm_multi_render_rgba8(self._h, img.ctypes.data), "rm_multi_render_rgba8")
        return img

    def render_device(self, rgba8_ptr):
        _check(lib().rm_multi_render_device(self._h, rgba8_ptr), "rm_multi_render_device")

    def enqueue_device(self, rgba8_ptr):
        """render_device without waiting; pair with wait()."""
        _check(lib().rm_multi_enqueue_device(self._h, rgba8_ptr), "rm_multi_enqueue_device")

    def enqueue_rgba8(self, host_ptr):
        """The frame into (page-locked) host memory at address host_ptr without waiting; pair with wait()."""
        _check(lib().rm_multi_enqueue_rgba8(self._h, host_ptr), "rm_multi_enqueue_rgba8")

    def wait(self):
        _check(lib().rm_multi_wait(self._h), "rm_multi_wait")


class HostBuffer:
    """Page-locked host memory from rm_host_alloc, seen as a numpy array."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        _check(lib().rm_host_alloc(self.nbytes, C.byref(self._p)), "rm_host_alloc")
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p.value), dtype=self.dtype).reshape(self.shape)

    @property
    def ptr(self):
        return self._p.value

    def free(self):
        if self._p:
            self.array = None
            _check(lib().rm_host_free(self._p), "rm_host_free")
            self._p = C.c_void_p()


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    _check(lib().rm_write_png(os.fsencode(path), rgba.ctypes.data, w, h), "rm_write_png")
