"""The compiled kernel mode's host half, without a GPU: source generation for every shipped scene and a gfx950
compile of it with hiprtc (what rm_set_scene does before it loads the code object)."""
import os

import pytest

from conftest import scene_path


@pytest.mark.parametrize("name,lattice", [("default.txt", True), ("default.txt", False), ("menger2.txt", True),
                                          ("closeup6.txt", True), ("synthetic1024.txt", True)])
def test_generated_kernels_compile_for_gfx950(rm, name, lattice):
    src = rm.kernel_source(rm.Scene(scene_path(name)), lattice=lattice)
    assert 'extern "C" __global__' in src and "rm_trace" in src and "rm_eval_sdf" in src
    assert "trace_body<CompiledSceneT<false>, 1, false>" in src
    assert ("lattice_level<" in src) == (lattice and name != "synthetic1024.txt")
    assert ("rm_trace_exact" in src) == ("lattice_level<" in src)   # scenes with a lattice: the optimistic kernel's exact pass
    assert ("grid_surface<" in src) == ("grid_fold<" in src) == (name == "synthetic1024.txt")
    assert ("fold_run_clustered<" in src) == (name == "synthetic1024.txt")
    rm.kernel_source_compiles(src)


def test_counting_variant_and_compile_errors(rm):
    scene = rm.Scene(scene_path("menger2.txt"))
    counted = rm.kernel_source(scene, counting=True)
    assert "trace_body<CompiledSceneT<true>, 1, true>" in counted and "rm_trace_exact" not in counted
    rm.kernel_source_compiles(counted)
    with pytest.raises(rm.RaymarchError) as e:
        rm.kernel_source_compiles(counted.replace("single_sd<", "no_such_function<", 1))
    assert "no_such_function" in str(e.value)


def test_code_objects_are_kept_on_disk_when_asked(rm, tmp_path, monkeypatch):
    import time
    monkeypatch.setenv("AMD_COMGR_CACHE", "0")   # the compiler's own cache would make the compiles below as quick as the loads
    src = rm.kernel_source(rm.Scene(scene_path("menger2.txt")))
    rm.set_kernel_cache_dir(str(tmp_path))
    try:
        t0 = time.perf_counter()
        rm.kernel_source_compiles(src)
        first = time.perf_counter() - t0
        files = list(tmp_path.iterdir())
        assert len(files) == 1 and files[0].name.endswith(".gfx950.bin") and files[0].stat().st_size > 10000
        t0 = time.perf_counter()
        rm.kernel_source_compiles(src)
        assert time.perf_counter() - t0 < first / 5          # loaded, not compiled
        # a damaged file is not trusted: the key stored in it no longer matches, so the source is compiled again
        data = bytearray(files[0].read_bytes())
        data[40] ^= 0xFF
        files[0].write_bytes(bytes(data))
        t0 = time.perf_counter()
        rm.kernel_source_compiles(src)
        assert time.perf_counter() - t0 > first / 5
        rm.kernel_source_compiles(src.replace("rm_eval_sdf", "rm_eval_sdf2"))   # another source, another file
        assert len(list(tmp_path.iterdir())) == 2
    finally:
        rm.set_kernel_cache_dir(None)


def test_cache_directory_stays_below_its_limit(rm, tmp_path, monkeypatch):
    """The files unused for longest go once the directory outgrows $RM_KERNEL_CACHE_MAX_MB."""
    import os
    import time
    base = rm.kernel_source(rm.Scene(scene_path("menger2.txt")))
    rm.set_kernel_cache_dir(str(tmp_path))
    try:
        rm.kernel_source_compiles(base)
        one = sum(f.stat().st_size for f in tmp_path.iterdir())
        assert one > 10000
        monkeypatch.setenv("RM_KERNEL_CACHE_MAX_MB", "1")
        n = (1 << 20) // one + 3                       # enough sources to outgrow one megabyte
        names = []
        for i in range(n):
            rm.kernel_source_compiles(base.replace("rm_eval_sdf", f"rm_eval_sdf_{i}"))
            newest = max(tmp_path.iterdir(), key=lambda f: f.stat().st_mtime_ns)
            names.append(newest.name)
            old = time.time() - 1000 + i               # distinct, rising times of last use
            os.utime(newest, (old, old))
        left = {f.name for f in tmp_path.iterdir()}
        assert sum(f.stat().st_size for f in tmp_path.iterdir()) <= (1 << 20)
        assert len(left) < n + 1 and names[-1] in left   # the newest survives, some older ones are gone
        assert all(name.startswith("rm_") and name.endswith(".gfx950.bin") for name in left)
    finally:
        rm.set_kernel_cache_dir(None)


def test_kernels_are_compiled_by_the_program_beside_the_library(rm):
    """The library compiles a scene's kernels with rm_jit, a process of its own linked against the ROCm release the library was
    built with; where that program is switched off (or missing) it compiles in the calling process.  Both accept and refuse
    the same sources."""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(rm.LIB_PATH), "rm_jit")
    assert os.access(helper, os.X_OK), "raymarching_amd/rm_jit has not been built (make -C raymarching_amd/csrc)"
    # a program built with other device headers than the library's refuses to compile for it (the library then compiles itself)
    assert subprocess.run([helper, "0123456789abcdef", os.devnull, os.devnull]).returncode == 11
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import os, signal\n"
        "if os.environ.get('TEST_IGNORE_SIGCHLD'): signal.signal(signal.SIGCHLD, signal.SIG_IGN)\n"
        "import raymarching_amd as rm\n"
        "rm.set_kernel_cache_dir(None)\n"
        "src = rm.kernel_source(rm.Scene(%r))\n"
        "rm.kernel_source_compiles(src)\n"
        "print(rm.kernel_compiler())\n"
        "try:\n"
        "    rm.kernel_source_compiles(src.replace('single_sd<', 'no_such_function<', 1))\n"
        "except rm.RaymarchError as e:\n"
        "    print('refused' if 'no_such_function' in str(e) else 'refused without a log')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(rm.__file__))), scene_path("menger2.txt"))
    # "ignore": the calling program ignores SIGCHLD, so the library never learns the program's exit status: it goes by the code
    # object file, and compiles the refused source once more in its own process, which refuses it too
    for setting, want in ((None, "rm_jit, a process of its own: " + helper), ("off", "hiprtc in this process"),
                          ("/no/such/program", "hiprtc in this process"), ("ignore", None)):
        env = dict(os.environ)
        env.pop("RM_JIT_HELPER", None)
        if setting == "ignore":
            env["TEST_IGNORE_SIGCHLD"] = "1"
            want = "rm_jit, a process of its own: " + helper
        elif setting is not None:
            env["RM_JIT_HELPER"] = setting
        out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.splitlines()[-2:] == [want, "refused"], (setting, out.stdout)
