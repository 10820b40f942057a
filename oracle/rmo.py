"""ctypes binding of the CPU oracle (oracle/librm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
benchmark's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librm_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("kernel", C.c_uint32),
        ("iterations", C.c_int32), ("saturation", C.c_float),
        ("surface_eps", C.c_float), ("offset_eps", C.c_float),
        ("normalize_div", C.c_int32), ("variant", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [("sdf_calls", C.c_uint64), ("prim_evals", C.c_uint64),
                ("march_steps", C.c_uint64), ("rays", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class FlatPrim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("list", C.c_uint32),
                ("center", C.c_float * 3), ("size", C.c_float * 3), ("color", C.c_float * 3)]


class Node(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mode", C.c_uint32), ("child_count", C.c_uint32),
                ("center", C.c_float * 3), ("size", C.c_float * 3), ("color", C.c_float * 3)]


def build():
    """Compile the oracle with its Makefile (gcc only, no GPU needed)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.rmo_scene_load.restype = C.c_void_p
        L.rmo_scene_load.argtypes = [C.c_char_p]
        L.rmo_scene_parse.restype = C.c_void_p
        L.rmo_scene_parse.argtypes = [C.c_char_p]
        L.rmo_scene_free.argtypes = [C.c_void_p]
        L.rmo_scene_from_nodes.restype = C.c_void_p
        L.rmo_scene_from_nodes.argtypes = [C.POINTER(Node), C.c_uint32, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        for f in ("rmo_scene_prim_count", "rmo_scene_list_count", "rmo_scene_light_count"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.rmo_scene_flat_prims.argtypes = [C.c_void_p, C.POINTER(FlatPrim)]
        L.rmo_scene_list_modes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_scene_lights.argtypes = [C.c_void_p, C.c_void_p]
        L.rmo_scene_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_sdf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rmo_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_int, C.c_void_p, C.POINTER(Counters)]
        L.rmo_render_pixels.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.rmo_pixel_samples.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_void_p]
        L.rmo_quantize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.rmo_default_params.argtypes = [C.POINTER(Params)]
        L.rmo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def default_params(**over):
    p = Params()
    lib().rmo_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Scene:
    """Owning handle on an oracle scene (scene::load restated)."""

    def __init__(self, path=None, text=None, nodes=None, lights=(), cam_position=(0, 0, 0),
                 cam_direction=(0, 0, -1), cam_up=(0, 1, 0), cam_fov=90.0):
        L = lib()
        if nodes is not None:
            arr = (Node * len(nodes))(*nodes)
            lt = np.ascontiguousarray(lights, np.float32).reshape(-1, 3)
            cp, cd, cu = (np.ascontiguousarray(v, np.float32) for v in (cam_position, cam_direction, cam_up))
            self._h = L.rmo_scene_from_nodes(arr, len(arr), lt.ctypes.data, len(lt), cp.ctypes.data,
                                             cd.ctypes.data, cu.ctypes.data, C.c_float(cam_fov))
        elif text is not None:
            self._h = L.rmo_scene_parse(text.encode())
        else:
            self._h = L.rmo_scene_load(os.fsencode(path))
        if not self._h:
            raise RuntimeError("oracle: scene load failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rmo_scene_free(self._h)
            self._h = None

    @property
    def prim_count(self):
        return lib().rmo_scene_prim_count(self._h)

    @property
    def list_count(self):
        return lib().rmo_scene_list_count(self._h)

    @property
    def light_count(self):
        return lib().rmo_scene_light_count(self._h)

    def flat_prims(self):
        n = self.prim_count
        arr = (FlatPrim * n)()
        lib().rmo_scene_flat_prims(self._h, arr)
        return arr

    def list_modes(self):
        n = self.list_count
        modes = np.zeros(n, np.uint32)
        parents = np.zeros(n, np.int32)
        lib().rmo_scene_list_modes(self._h, modes.ctypes.data, parents.ctypes.data)
        return modes, parents

    def lights(self):
        out = np.zeros((self.light_count, 3), np.float32)
        lib().rmo_scene_lights(self._h, out.ctypes.data)
        return out

    def camera(self):
        t = np.zeros(16, np.float32)
        f = C.c_float()
        lib().rmo_scene_camera(self._h, t.ctypes.data, C.byref(f))
        return t, np.float32(f.value)

    def sdf(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        sd = np.zeros(len(xyz), np.float32)
        rgb = np.zeros((len(xyz), 3), np.float32)
        lib().rmo_sdf(self._h, xyz.ctypes.data, len(xyz), sd.ctypes.data, rgb.ctypes.data)
        return sd, rgb

    def render(self, params, rows=None, threads=0):
        """Returns (float32 [H,W,4] framebuffer, counters dict).

        rows = (row0, row1, step) restricts the rendered rows (others stay 0).
        threads: 0 = all OpenMP threads, 1 = the serial loop.
        """
        if threads == 0:
            threads = lib().rmo_max_threads()
        r0, r1, st = rows if rows is not None else (0, params.height, 1)
        out = np.zeros((params.height, params.width, 4), np.float32)
        cnt = Counters()
        lib().rmo_render(self._h, C.byref(params), r0, r1, st, threads, out.ctypes.data, C.byref(cnt))
        return out, cnt.as_dict()

    def render_pixels(self, params, xy, threads=0):
        """RGBA floats of the pixels xy = [(x, y), ...] (any subset of the frame), rendered in parallel."""
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        out = np.zeros((len(xy), 4), np.float32)
        lib().rmo_render_pixels(self._h, C.byref(params), xy.ctypes.data, len(xy), threads or lib().rmo_max_threads(), out.ctypes.data)
        return out

    def pixel_samples(self, params, px, py):
        k = params.kernel
        out = np.zeros((k * k, 3), np.float32)
        lib().rmo_pixel_samples(self._h, C.byref(params), px, py, out.ctypes.data)
        return out


def quantize(rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros(rgba.shape, np.uint8)
    lib().rmo_quantize(rgba.ctypes.data, rgba.size // 4, out.ctypes.data)
    return out


def max_threads():
    return lib().rmo_max_threads()
