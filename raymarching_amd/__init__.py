"""raymarching_amd -- ctypes view of libraymarch_hip.so (the C ABI in include/raymarch_hip.h).

The product is the shared library: host scene loader (C++), C ABI and gfx950
kernels.  This module only binds it for the tests and the benchmark; there is
no Python or CPU implementation of the render path behind it, and importing
the binding without the built library raises.
"""
from .binding import (  # noqa: F401
    LIB_PATH, Node, Params, Rows, Counters, Scene, Renderer, RaymarchError,
    default_params, exported_symbols, HostBuffer, kernel_source, kernel_source_compiles, kernel_compiler, precompile, set_kernel_cache_dir, lib, write_png,
    MultiRenderer, MULTI_AUTO, MULTI_RCCL, MULTI_COPY, MULTI_DIRECT,
)
