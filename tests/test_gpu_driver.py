"""The command-line driver (raymarching_amd/csrc/main.cpp) against the oracle: same report lines as the
reference's main() (/root/reference/source/main.cpp:43-77) and a frame identical to the CPU restatement's."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

DRIVER = os.path.join(ROOT, "raymarching_amd", "raymarch")


def test_driver_report_and_frame(oracle, tmp_path):
    if not os.path.isfile(DRIVER):
        pytest.skip("raymarching_amd/raymarch has not been built (make -C raymarching_amd/csrc)")
    raw, png = tmp_path / "out.rgba", tmp_path / "out.png"
    cmd = [DRIVER, "--scene", scene_path("menger2.txt"), "--width", "128", "--height", "96", "--ssaa", "2",
           "--frames", "2", "--raw", str(raw), "--out", str(png)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    for line in ("...Loading scene", "...Setting up context", "...Compiling shaders", "...Generating buffers", "...Rendering",
                 "Render with GPU:", "Copy to GPU:", "Render + Copy on GPU:", "Mrays/s"):
        assert line in out.stdout, out.stdout
    assert "[Error]" not in out.stdout
    img = np.fromfile(raw, dtype=np.uint8).reshape(96, 128, 4)
    ref, _ = oracle.Scene(scene_path("menger2.txt")).render(oracle.default_params(width=128, height=96, kernel=2))
    assert np.array_equal(img, oracle.quantize(ref))
    assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"


def test_driver_shards_a_frame_over_listed_devices(oracle, tmp_path):
    """raymarch --devices 0,0,0: render::setup::devices + render::GPU over three ranks on this one GPU (every rank stores
    its rows straight into the first device's frame); the frame is the one-device frame."""
    if not os.path.isfile(DRIVER):
        pytest.skip("raymarching_amd/raymarch has not been built (make -C raymarching_amd/csrc)")
    raw = tmp_path / "out.rgba"
    cmd = [DRIVER, "--scene", scene_path("menger2.txt"), "--width", "128", "--height", "90", "--ssaa", "2", "--devices", "0,0,0",
           "--frames", "3", "--raw", str(raw), "--out", ""]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "[Error]" not in out.stdout and "3 ranks, rows stored straight into the first device's frame" in out.stdout, out.stdout
    img = np.fromfile(raw, dtype=np.uint8).reshape(90, 128, 4)
    ref, _ = oracle.Scene(scene_path("menger2.txt")).render(oracle.default_params(width=128, height=90, kernel=2))
    assert np.array_equal(img, oracle.quantize(ref))


def test_driver_reports_a_missing_scene_like_the_reference(tmp_path):
    """scene::load leaves the scene empty when the file is missing (scene.cpp:95-98); the frame is then black."""
    if not os.path.isfile(DRIVER):
        pytest.skip("raymarching_amd/raymarch has not been built (make -C raymarching_amd/csrc)")
    raw = tmp_path / "out.rgba"
    out = subprocess.run([DRIVER, "--scene", str(tmp_path / "absent.txt"), "--width", "32", "--height", "16", "--ssaa", "1",
                          "--raw", str(raw), "--out", ""], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    img = np.fromfile(raw, dtype=np.uint8).reshape(16, 32, 4)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 255).all()


@pytest.mark.parametrize("extra,batch", [([], []), (["--balance"], []), ([], ["--frames-per-gather", "2"])],
                         ids=["cyclic", "dealt-by-cost", "two-frames-per-gather"])
def test_two_rank_bench_rehearsal_gathers_the_single_launch_frame(extra, batch):
    """bench.py's N > 1 path (cyclic bands, frames in flight, gather to rank 0, assembly) with both ranks on this
    one GPU and gloo in place of RCCL: the gathered frames must equal the frame a single launch renders."""
    import json
    import sys
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "3",
           "--workload", "menger2", "--rehearse-on-one-gpu", "--no-cpu-baseline", "--no-exhaustive-leg"] + extra + batch
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["gathered_frame_equals_single_launch"] is True
    assert line["value"] > 0 and line["config"]["frames_in_flight"] == 4 and line["config"]["frames_per_gather"] == (2 if batch else 1)
    assert ("dealt by measured cost" in line["config"]["sharding"]) == bool(extra)
