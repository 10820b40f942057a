import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The library keeps compiled kernels and measured tile orders in a cache directory (default: under the user's home):
# the tests get a directory of their own, so that what an earlier run left behind cannot change what they observe.
import tempfile  # noqa: E402
os.environ["RM_KERNEL_CACHE_DIR"] = tempfile.mkdtemp(prefix="rm_test_cache_")

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (built on demand with gcc)."""
    from oracle import rmo
    rmo.lib()
    return rmo


@pytest.fixture(scope="session")
def rm():
    """The product binding (libraymarch_hip.so must have been built)."""
    import raymarching_amd
    raymarching_amd.lib()
    return raymarching_amd


@pytest.fixture(scope="session", params=["compiled", "interpreted"])
def renderer(rm, request):
    """One context per kernel mode: kernels generated and compiled for each uploaded scene (the
    default), and the ahead-of-time kernel that walks the scene's record stream."""
    r = rm.Renderer(0)
    r.set_kernel_mode(request.param == "compiled")
    r.kernel_mode = request.param
    yield r
    r.close()


def scene_path(name):
    return os.path.join(SCENES, name)


def same_but_zero_signs(a, b):
    """True if two float32 arrays are equal bit for bit except that a zero may differ in sign.  The march loop's
    evaluator may fold equal candidates in another order than the colour-tracking one (cooperative lattice rows);
    max and min of the same set of floats then agree in everything but the sign of a zero result."""
    import numpy as np
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))).all())
