"""Host side of the product library (no GPU needed): the C ABI loads and exports every
declared symbol, and the C++ scene loader / generators / camera agree bit for bit with the oracle."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path


def test_library_exports_every_declared_symbol(rm):
    header = open(os.path.join(ROOT, "include", "raymarch_hip.h")).read()
    declared = set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed from the header"
    from raymarching_amd import binding
    assert declared == set(binding.EXPORTS)
    assert set(rm.exported_symbols()) == declared


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 with nothing but itself."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "raymarch_hip.h"\nint main(void) { rm_params p = { 1024, 768, 3, 1000, 0.05f, 1e-3f, 1e-3f }; '
                   '(void)p; return RM_OK; }\n')
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                          "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def _flatten_product(nodes):
    """(type, list index, centre, size, colour) per primitive, lists numbered in DFS order."""
    out, modes = [], []
    pos = [0]

    def rec():
        n = nodes[pos[0]]; pos[0] += 1
        modes.append(n.mode)
        my_id = len(modes) - 1
        for _ in range(n.child_count):
            k = nodes[pos[0]]
            if k.type == 0:
                rec()
            else:
                pos[0] += 1
                out.append((k.type, my_id, tuple(k.center), tuple(k.size), tuple(k.color)))
    rec()
    assert pos[0] == len(nodes)
    return out, modes


@pytest.mark.parametrize("name", ["default.txt", "menger2.txt"])
def test_loader_matches_oracle_bitwise(rm, oracle, name):
    prod = rm.Scene(scene_path(name))
    orc = oracle.Scene(scene_path(name))
    prims, modes = _flatten_product(prod.nodes())
    oprims = orc.flat_prims()
    assert len(prims) == len(oprims) == orc.prim_count
    for a, b in zip(prims, oprims):
        assert a == (b.type, b.list, tuple(b.center), tuple(b.size), tuple(b.color))
    assert modes == orc.list_modes()[0].tolist()
    np.testing.assert_array_equal(prod.lights(), orc.lights())
    pt, pf = prod.camera()
    ot, of = orc.camera()
    assert pt.tobytes() == ot.tobytes() and pf.tobytes() == of.tobytes()


def test_loader_menger4_generator_order(rm, oracle):
    text = "Color 0.25 0.5 0.75\nMengerSponge 1.5 -2.25 3.125 7.0 4\n"
    prims, _ = _flatten_product(rm.Scene(text=text).nodes())
    oprims = oracle.Scene(text=text).flat_prims()
    assert len(prims) == 1 + 8421
    for a, b in zip(prims, oprims):
        assert a == (b.type, b.list, tuple(b.center), tuple(b.size), tuple(b.color))


def test_loader_edge_cases(rm, oracle):
    text = "\n\nNonsense 1 2\nCamera FOV 40\nCamera Position 1 2 3\nLight 1 1 1\nCross 0 0 0 1 2 3\nBounds 10\n"
    prod, orc = rm.Scene(text=text), oracle.Scene(text=text)
    prims, modes = _flatten_product(prod.nodes())
    assert [(p[0], p[1]) for p in prims] == [(3, 0), (2, 1)] and modes == [0, 1]
    assert prod.camera()[0].tobytes() == orc.camera()[0].tobytes()
    assert prod.camera()[1] == orc.camera()[1]
    empty = rm.Scene("/nonexistent/objects.txt")
    assert len(empty.nodes()) == 1 and empty.nodes()[0].child_count == 0 and len(empty.lights()) == 0


def test_png_writer_roundtrip(rm, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    path = tmp_path / "x.png"
    rm.write_png(str(path), img)
    back = np.array(Image.open(path))
    np.testing.assert_array_equal(back, img)


def test_missing_library_is_an_error(rm, monkeypatch):
    from raymarching_amd import binding
    monkeypatch.setattr(binding, "_lib", None)
    monkeypatch.setattr(binding, "LIB_PATH", "/nonexistent/libraymarch_hip.so")
    with pytest.raises(binding.RaymarchError):
        binding.lib()
