#!/usr/bin/env python3
"""Benchmark of the sphere-tracing hot path: Mrays/s (primary + shadow rays) and frame time.

    python bench.py --gpus N --steps K --warmup W [--workload default|menger2|uhd|closeup6|synthetic]
                    [--evaluator lattice|exhaustive] [--kernel-mode compiled|interpreted]

A step is one frame of the workload.  Frames are independent, so --frames-in-flight contexts (default 4; 8 from
eight GPUs on), each on its own stream, alternate frames: the timed region holds exactly
K whole frames, all finished and gathered (streams drained) before the clock stops.  At N > 1 (one process per
GPU, launched by torch.distributed.run) the frame's row bands are dealt cyclically to the ranks, every rank
renders its bands into device memory and the RGBA8 bands are gathered to rank 0 with RCCL (the one exchange
step of the path), --frames-per-gather frames to a collective; the same frame is rendered whatever N is, so
scaling is "strong".  Rank 0 prints ONE JSON line.

    value         = frames * W*H*k*k*(1+L) rays / wall time (max over ranks), in Mrays/s: the frame's nominal
                    ray count (scene.cpp:26-31, 53-62: one camera ray and L shadow rays per sample).  With
                    --shading coloured (the library's default) samples that end on a surface of colour
                    (0, 0, 0) do not march their shadow rays, whose result is multiplied by that colour;
                    mrays_cast_per_s counts the rays actually marched
    roofline      = FLOPs the trace kernel EXECUTED for the frame (its counting variant's tally of the
                    primitive evaluations it executed x FLOPs per primitive + 7 per march step) / its
                    HIP-event duration, against the fp32 vector peak.  The default kernels never visit most
                    primitives (lattice evaluator, lists skipped by a bounding child); `brute_force_equivalent` gives the reference algorithm's FLOPs (counted
                    scene-SDF calls x FLOPs of one exhaustive call) over the same time, a speed-up
                    figure, not a roofline fraction.
    roofline_exhaustive = the reference's algorithm as it stands (every primitive at every SDF call) on the
                    record-walking kernel, timed in the same run on a few frames: executed == algorithmic
                    FLOPs there.
    traffic       = HBM bytes of one trace launch from rocprofv3's TCC counters (2 x FETCH_SIZE + WRITE_SIZE,
                    the gfx950 correction for wide reads), read from profiles/<round>/traffic.json, which
                    tools/collect_traffic.py writes from a counter run of this workload (counters cannot be
                    collected from inside the timed process); null when no such run is on file.
    cpu_baseline  = the CPU oracle (a port of the reference's render::OMP / render::CPU) timed on
                    this host on a stated row sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: starts one rank per GPU with torch.distributed.run as a CHILD
    process -- before this process has imported torch or touched the GPU -- and hands on the child's output (rank 0's
    one JSON line) and its exit status."""
    import socket
    import subprocess
    with socket.socket() as s:               # a port that is free right now
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1 \
        and "--rehearse-on-one-gpu" not in sys.argv:
    sys.exit(_launch_ranks(_requested_gpus(sys.argv[1:])))

# Frames in flight run on one HIP stream each; HIP maps streams onto this many hardware queues per process
# (its default, 4, would let only four frames overlap).  Read when the runtime starts, hence set before torch.
# A sharded run keeps 4 frames in flight (up to 16 in sweeps), and the collective's own stream must not share a queue
# with a frame's: a gather waiting for its frame would hold up the frames queued behind it.
_QUEUES = 24 if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--emulate-share" in sys.argv or "--force-collective" in sys.argv) else 8
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < _QUEUES:
    os.environ["GPU_MAX_HW_QUEUES"] = str(_QUEUES)

import torch  # noqa: E402  (first: the renderer must share torch's HIP runtime)
import torch.distributed as dist  # noqa: E402

import raymarching_amd as rm  # noqa: E402
from raymarching_amd import sharding  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3        # MI355X datasheet: an FMA in every issue slot; code without fusable multiply-adds tops out near half of it
VALU_ISSUE_CEILING_T = 69.0            # T lane-instructions/s the 13-instruction cross body sustains (tools/valu_bench.hip)
from tools.workloads import WORKLOADS  # noqa: E402  (name -> scene file, parameter overrides, description)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="default", choices=sorted(WORKLOADS))
    ap.add_argument("--evaluator", default="lattice", choices=["lattice", "exhaustive"])
    ap.add_argument("--kernel-mode", default="compiled", choices=["compiled", "interpreted"])
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="contexts/streams per GPU that alternate frames (default: 4 at every rank count)")
    ap.add_argument("--frames-per-gather", type=int, default=0,
                    help="frames whose shards go to rank 0 in one collective (default 1)")
    ap.add_argument("--emulate-share", type=int, default=0,
                    help="diagnostic, 1 GPU only: render rank 0's share of an N-way split (no gather); value is "
                         "this one GPU's rate times N, an upper bound for the N-GPU run")
    ap.add_argument("--shading", choices=["coloured", "all"], default="coloured",
                    help="coloured (the library's default): samples on surfaces of colour (0, 0, 0) skip the normal and the "
                         "shadow rays, whose product with that colour is zero; all: every sample does the reference's full work")
    ap.add_argument("--no-every-ray-leg", action="store_true",
                    help="skip the extra timing of the same frames with the shading cuts off (keeps a kernel profile of the run to the main setting's launches)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: still form the RCCL group and send every frame through the gather and the "
                         "assembly (exercises the N > 1 code path's RCCL calls on a single GPU)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank run with every rank on cuda:0 and the gloo backend: exercises the N > 1 code path "
                         "(sharding, asynchronous gather, assembly) where only one GPU is at hand; its numbers mean nothing")
    ap.add_argument("--balance", action="store_true",
                    help="N > 1: re-deal the bands to the ranks by measured cost (off by default: cyclic 4-row bands leave "
                         "under 2 %% of cost imbalance at N = 8 on the shipped scenes, tools/band_balance.py)")
    ap.add_argument("--band-rows", type=int, default=4, help="rows per band when sharding over GPUs")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the timing of frames delivered into host memory")
    ap.add_argument("--no-first-frame-leg", action="store_true", help="skip the cold-start measurement in fresh processes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exhaustive-leg", action="store_true")
    return ap.parse_args()


def cpu_baseline(scene_file, kw):
    """Times the oracle on a bounded row sample of the workload; returns the JSON object."""
    from oracle import rmo
    sc = rmo.Scene(scene_file)
    if sc.prim_count > 20000:
        return {"value": None, "unit": "Mrays/s", "cores": rmo.max_threads(), "kind": "port",
                "sample": f"skipped: {sc.prim_count} primitives per SDF call are out of reach of the brute-force CPU path"}
    p = rmo.default_params(**kw)
    threads = rmo.max_threads()
    if sc.prim_count * p.kernel * p.kernel <= 40:
        # cheap enough for the whole frame on both legs, as the reference's main() times them (main.cpp:39-50)
        t0 = time.perf_counter()
        _, cnt = sc.render(p, threads=threads)
        t_omp = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, cnt1 = sc.render(p, threads=1)
        t_1 = time.perf_counter() - t0
        return {
            "value": round(cnt["rays"] / t_omp / 1e6, 5), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"render::OMP restatement, the whole {p.width}x{p.height} frame ({cnt['rays']} rays, {cnt['sdf_calls']} SDF calls) in {t_omp:.2f} s",
            "frame_s": round(t_omp, 3),
            "single_thread": {"value": round(cnt1["rays"] / t_1 / 1e6, 5), "unit": "Mrays/s", "cores": 1, "frame_s": round(t_1, 3),
                              "sample": f"render::CPU restatement, the whole frame ({cnt1['rays']} rays) in {t_1:.2f} s"},
        }
    step = max(1, p.height // 24)
    rows = (step // 2, p.height, step)                 # ~24 rows spread over the frame
    t0 = time.perf_counter()
    _, cnt = sc.render(p, rows=rows, threads=threads)
    t_omp = time.perf_counter() - t0
    rows1 = (step // 2 + 8 * step, p.height, 8 * step)  # ~3 of those rows for the serial loop
    t0 = time.perf_counter()
    _, cnt1 = sc.render(p, rows=rows1, threads=1)
    t_1 = time.perf_counter() - t0
    return {
        "value": round(cnt["rays"] / t_omp / 1e6, 5), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"render::OMP restatement on rows {rows[0]}::{rows[2]} ({len(range(*rows))} of {p.height} rows, "
                  f"{cnt['rays']} rays, {cnt['sdf_calls']} SDF calls) in {t_omp:.2f} s",
        "single_thread": {
            "value": round(cnt1["rays"] / t_1 / 1e6, 5), "unit": "Mrays/s", "cores": 1,
            "sample": f"render::CPU restatement on rows {rows1[0]}::{rows1[2]} ({len(range(*rows1))} rows, {cnt1['rays']} rays) in {t_1:.2f} s",
        },
    }


def executed_flops(c):
    """FLOPs the trace kernel executed for the counted frame: the kernel's own tally of the primitive
    evaluations it executed (11 per sphere, 9 per box, 15 per cross; lists skipped by a bound and lattice
    crosses never visited are not in it; nor is the camera rays' shared first evaluation, which the timed kernel
    skips) + 7 per march step."""
    return c["flops_executed"] - c.get("flops_shared_first", 0) + 7 * c["march_steps"]


def recorded_traffic(workload, evaluator, kernel_mode):
    """HBM bytes per trace launch from the newest committed counter run of this configuration."""
    base = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(base), reverse=True) if os.path.isdir(base) else []:
        path = os.path.join(base, rnd, "traffic.json")
        if os.path.isfile(path):
            for row in json.load(open(path)):
                if (row["workload"], row["evaluator"], row["kernel_mode"]) == (workload, evaluator, kernel_mode):
                    return row["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), row.get("valu_wave_insts_per_launch")
    return None, None, None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        raise SystemExit("--gpus N > 1 runs one process per GPU (python bench.py --gpus N starts them itself; "
                         "--rehearse-on-one-gpu must be launched with torch.distributed.run --nproc-per-node N)")
    if world > 1 and not a.rehearse_on_one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: this node shows {torch.cuda.device_count()} GPU(s); "
                         "one rank per GPU is the only sharded mode (RCCL refuses two ranks on one device)")
    if a.rehearse_on_one_gpu:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(60, exit=True)   # a rehearsal that hangs says where
    torch.cuda.set_device(local)
    group = world > 1 or a.force_collective       # a process group exists and frames go through the gather
    if group:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29573")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    scene_name, kw, desc = WORKLOADS[a.workload]
    scene_file = os.path.join(ROOT, "scenes", scene_name)
    params = rm.default_params(**kw)
    scene = rm.Scene(scene_file)
    n_lights = len(scene.lights())
    # ---- contexts: one per frame in flight, each on its own stream (frames are independent; the tail of one
    # frame's long-running tiles overlaps the start of the next frame, and the gather of frame i overlaps the
    # render of frame i + 1)
    # Four frames in flight hide one frame's longest tiles at every share size measured (rank 0's eighth of the default
    # frame through a one-rank gather, 400 timed frames, ms per frame: 2 in flight 0.110, 3 0.092, 4 0.081, 8 0.093,
    # 16 0.089; tools/gather_sweep.sh).
    F = a.frames_in_flight if a.frames_in_flight > 0 else 4
    # frames per collective: one.  Frames gathered together wait for each other (the same share, 16 in flight: one frame
    # per gather 0.090 ms per frame, two 0.092, four 0.100, eight 0.103).
    G = a.frames_per_gather if a.frames_per_gather > 0 else 1
    if F % G:
        raise SystemExit("--frames-in-flight must be a multiple of --frames-per-gather")
    if a.balance:
        # dealing by cost needs the tiles' measured durations; a tile order read from the cache directory comes without them
        rm.set_kernel_cache_dir(None)
    rs, set_scene_s, compiled_by = [], 0.0, None
    for f in range(F):
        rr = rm.Renderer(local)
        rr.set_params(params)
        rr.set_evaluator(a.evaluator == "lattice")
        rr.set_kernel_mode(a.kernel_mode == "compiled")
        rr.set_shading(a.shading == "coloured")
        t0 = time.perf_counter()
        rr.set_scene(scene)
        if f == 0:
            set_scene_s = time.perf_counter() - t0   # later contexts find the kernels in the code cache
            compiled_by = rm.kernel_compiler()
        rs.append(rr)
    r = rs[0]

    W, H, k = params.width, params.height, params.kernel
    rays_per_frame = W * H * k * k * (1 + n_lights)

    # ---- sharding: bands of band_rows rows, band b belongs to rank b % world
    rows, local_rows = sharding.band_plan(H, a.band_rows, world, rank)
    if a.emulate_share > 1:
        if world != 1:
            raise SystemExit("--emulate-share is a single-GPU diagnostic")
        rows, local_rows = sharding.band_plan(H, a.band_rows, a.emulate_share, 0)
    # Frame slot f renders into slice f % G of batch f // G; a batch's G frames go to rank 0 in one collective,
    # queued on the stream of the batch's last frame behind the other frames of the batch.
    def make_gatherers(band_lists=None):
        return [sharding.FrameGatherer(b, H, a.band_rows, world, rank, stage_on_host=a.rehearse_on_one_gpu,
                                       always_collective=group, band_lists=band_lists) for b in batches]
    # Two sets of batch buffers alternate between rounds of F frames, so that a stream's next frame does not wait
    # for the collective that is reading its last one.
    NB = F // G
    batches = [torch.empty((G, local_rows, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2 * NB)]
    shards = [batches[f // G][f % G] for f in range(F)]
    shard = shards[0]
    gatherers = make_gatherers()
    # every context's own HIP stream, seen from PyTorch (streams the library creates land on distinct hardware
    # queues in creation order; streams of PyTorch's pool do not)
    streams = [torch.cuda.ExternalStream(rr.stream) for rr in rs]
    rendered = [torch.cuda.Event() for _ in range(F)]

    # ---- per context: one synchronous frame times the tiles, the next ones run in cost order; the trace
    # kernel's duration is taken here, with nothing else on the GPU
    iso = []
    for rr, sh in zip(rs, shards):
        for i in range(4):
            rr.render_device(rows, None, sh.data_ptr())
            if i >= 2:
                iso.append(rr.counters()["kernel_ms"])
    kernel_ms = sum(iso) / len(iso)

    # ---- cross-GPU balance by measured cost: every rank turns the tile durations it has just measured into a cost
    # per band, the costs are exchanged, and all ranks compute the same plan that deals the bands -- the same number
    # to each rank -- so that the summed costs are equal (opt-in: on the shipped scenes cyclic dealing is within 2 % already)
    band_lists = None
    if world > 1 and a.balance:
        import numpy as np
        tiles = r.tile_costs()
        side = 8 if tiles.shape[0] * 8 >= local_rows * k else 16
        mine = sharding.band_costs_from_tiles(tiles, side, k, a.band_rows, rows.band_count)
        dev = "cpu" if a.rehearse_on_one_gpu else "cuda"
        mine_t = torch.tensor(mine, dtype=torch.float64, device=dev)
        every = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(every, mine_t)
        costs = np.zeros(world * rows.band_count)
        for q, t in enumerate(every):
            costs[q::world] = t.cpu().numpy()           # rank q holds image bands q, q + world, ...
        band_lists = sharding.balanced_band_lists(costs, world)
        for rr, sh in zip(rs, shards):
            rr.set_band_list(band_lists[rank])
            for i in range(3):                          # times the tiles of the new share, then runs in cost order
                rr.render_device(rows, None, sh.data_ptr())
        gatherers = make_gatherers(band_lists)
        kernel_ms = rs[0].counters()["kernel_ms"]

    # ---- work count of this rank's share (counting kernel variant, outside the timed region)
    r.set_counting(True)
    r.render_device(rows, None, shard.data_ptr())
    cnt = r.counters()
    r.set_counting(False)
    local_exec = executed_flops(cnt)
    local_algo = cnt["sdf_calls"] * cnt["flops_per_sdf"] + 7 * cnt["march_steps"]

    timed_kernel_ms = []

    gathered = set()

    def frame(i, last):
        f = i % F
        b, slot = divmod(f, G)
        b += NB * ((i // F) & 1)
        if group:
            gatherers[b].wait(streams[f])      # the collective that read this buffer two rounds ago
        rs[f].enqueue_device(rows, None, batches[b][slot].data_ptr())     # on the context's own stream, streams[f]
        if slot == G - 1 or last:              # the batch is complete (or the loop ends): one collective for its frames,
            for other in range(f - slot, f):   # behind every one of them
                streams[f].wait_event(rendered[other])
            with torch.cuda.stream(streams[f]):
                gatherers[b].start(batches[b])
            gathered.update((b, j) for j in range(slot + 1))
        elif group:
            rendered[f].record(streams[f])
        if F == 1:   # nothing overlaps: every frame's own kernel time can be booked
            rs[0].wait()
            timed_kernel_ms.append(rs[0].counters()["kernel_ms"])

    def drain():
        for rr in rs:
            rr.wait()
        for st in streams:
            st.synchronize()
        torch.cuda.synchronize()

    # set-up, like the tile timing above: every frame slot and both of its buffers carry one frame through the
    # gather once (first use of a buffer in a collective registers it with the backend), whatever W is
    if group:
        for i in range(2 * F):
            frame(i, i == 2 * F - 1)
        drain()
    # Before the W warm-up frames: the GPU under load long enough to reach its steady clocks.  The tile timing above runs
    # one frame at a time with the GPU idle in between; 20 timed frames right after 5 warm-up frames measure 0.545 ms per
    # default frame, after 50 or more 0.515 ms -- the rate of every longer run.  The same count on every rank (the
    # frames go through the collective); a lone process goes on until 40 ms have passed.
    pre_roll = 64
    t_pre = time.perf_counter()
    for i in range(pre_roll):
        frame(i, i == pre_roll - 1)
    drain()
    if not group:
        while time.perf_counter() - t_pre < 0.040 and pre_roll < 4096:
            for i in range(F):
                frame(i, i == F - 1)
            drain()
            pre_roll += F
    gathered.clear()
    for i in range(a.warmup):
        frame(i, i == a.warmup - 1)
    drain()
    gathered.clear()
    if group:
        dist.barrier()
    timed_kernel_ms.clear()
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame(i, i == a.steps - 1)
    host_issue_ms = (time.perf_counter() - t0) / a.steps * 1e3   # host time to queue a frame (launches, collective)
    drain()
    if group:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # tiles the scene's exact kernel rendered again behind the optimistic one, in all frames of this process so far
    exact_tiles = sum(rr.exact_tiles() for rr in rs) if a.kernel_mode == "compiled" else 0
    overlapped_ms = sum(rr.counters()["kernel_ms"] for rr in rs) / F   # last timed frame of every context, co-scheduled
    kernel_ms_before = kernel_ms
    if timed_kernel_ms:
        kernel_ms = overlapped_ms = sum(timed_kernel_ms) / len(timed_kernel_ms)   # one frame in flight: the timed frames themselves
    else:
        # The trace kernel's own duration for the roofline: frames in flight stretch each other's spans, and the frames timed
        # during set-up ran on a GPU that idled between them (lower clocks: 0.48 against 0.45 ms for the default frame).  So:
        # right behind the timed region, with the GPU still under load, frames one at a time, back to back, for at least
        # 16 frames and 20 ms; the mean of the second half.
        alone, t_alone = [], time.perf_counter()
        while len(alone) < 16 or (time.perf_counter() - t_alone < 0.020 and len(alone) < 4096):
            r.render_device(rows, None, shard.data_ptr())
            alone.append(r.counters()["kernel_ms"])
        alone = alone[len(alone) // 2:]
        kernel_ms = sum(alone) / len(alone)
    # ---- the same frames with every sample and every ray doing the reference's full work (one GPU only; after the
    # timed region): the rate at which the frame's nominal rays are all marched
    all_marched = None
    if world == 1 and not group and a.emulate_share <= 1 and a.shading == "coloured" and not a.no_every_ray_leg:
        for rr, sh in zip(rs, shards):
            rr.set_shading(False)
            for _ in range(3):                        # times the tiles of the new setting, then runs in cost order
                rr.render_device(rows, None, sh.data_ptr())
        all_kernel_ms = r.counters()["kernel_ms"]
        r.set_counting(True)
        r.render_device(rows, None, shard.data_ptr())
        cnt_all_rays = r.counters()
        r.set_counting(False)
        n_all = max(F, min(a.steps, 100))
        for i in range(F):
            frame(i, i == F - 1)
        drain()
        t1 = time.perf_counter()
        for i in range(n_all):
            frame(i, i == n_all - 1)
        drain()
        all_ms = (time.perf_counter() - t1) / n_all * 1e3
        all_tflops = executed_flops(cnt_all_rays) / (all_kernel_ms * 1e-3) / 1e12
        all_marched = {"value": round(rays_per_frame / all_ms / 1e3, 3), "unit": "Mrays/s", "ms_per_step": round(all_ms, 4), "steps": n_all,
                       "roofline": {"bound": "valu_fp32", "achieved": round(all_tflops, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(all_tflops / FP32_VECTOR_PEAK_TFLOPS, 4), "kernel_ms": round(all_kernel_ms, 4),
                                    "achieved_frames_in_flight": round(executed_flops(cnt_all_rays) / (all_ms * 1e-3) / 1e12, 3),
                                    "executed_flops_per_frame": executed_flops(cnt_all_rays), "sdf_calls_per_frame": cnt_all_rays["sdf_calls"],
                                    "rays_cast_per_frame": cnt_all_rays["rays"]},
                       "note": "rm_set_shading(RM_SHADE_ALL): no sample skips its normal and shadow rays, no shadow ray stops past its "
                               "light; same frames, timed right after the main region in the same process"}
        for rr in rs:
            rr.set_shading(True)
        timed_kernel_ms.clear()
    # ---- self-check of the sharded path (outside the timed region): the frames gathered on rank 0 must equal
    # the frame one launch renders
    gathered_ok = None
    for rr in rs:
        rr.set_band_list(None)
    if group and rank == 0:
        whole, _ = sharding.band_plan(H, a.band_rows, 1, 0)
        single = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        r.render_device(whole, None, single.data_ptr())
        gathered_ok = all(bool(torch.equal(gatherers[b].frame(j), single)) for b, j in sorted(gathered))

    stats = torch.tensor([elapsed, kernel_ms, float(local_exec), float(local_algo), float(cnt["sdf_calls"]), float(cnt["rays"])],
                         dtype=torch.float64, device="cuda")
    if group:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms_max = float(mx[0]), float(mx[1])
        total_exec, total_algo, total_sdf, total_rays = float(sm[2]), float(sm[3]), float(sm[4]), float(sm[5])
    else:
        kernel_ms_max, total_exec, total_algo, total_sdf, total_rays = kernel_ms, float(local_exec), float(local_algo), float(cnt["sdf_calls"]), float(cnt["rays"])

    # ---- the reference's algorithm as it stands -- every primitive evaluated at every SDF call -- on the
    # record-walking kernel (rank 0's share, a few frames): executed == algorithmic FLOPs there
    exhaustive = None
    if not a.no_exhaustive_leg and cnt["prims"] <= 20000:
        r.set_evaluator(False)
        r.set_kernel_mode(False)
        r.set_shading(False)                      # and every sample shaded, as the reference does
        r.set_scene(scene)
        r.set_counting(True)
        r.render_device(rows, None, shard.data_ptr())
        cnt_all = r.counters()
        r.set_counting(False)
        algo_all = cnt_all["sdf_calls"] * cnt_all["flops_per_sdf"] + 7 * cnt_all["march_steps"]
        for _ in range(2):
            r.render_device(rows, None, shard.data_ptr())
        ms = []
        for _ in range(3):
            r.render_device(rows, None, shard.data_ptr())
            ms.append(r.counters()["kernel_ms"])
        ex_ms = sum(ms) / len(ms)
        achieved = algo_all / (ex_ms * 1e-3) / 1e12
        exhaustive = {"bound": "valu_fp32", "kernel": "trace_kernel (record walker), every primitive visited at every SDF call, every sample shaded",
                      "sdf_calls_per_frame": cnt_all["sdf_calls"], "rays_cast_per_frame": cnt_all["rays"],
                      "achieved": round(achieved, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4), "traffic": None, "kernel_ms": round(ex_ms, 4),
                      "mrays_per_s": round(rays_per_frame / world / ex_ms / 1e3, 2) if world == 1 else None}
        r.set_evaluator(a.evaluator == "lattice")
        r.set_kernel_mode(a.kernel_mode == "compiled")
        r.set_shading(a.shading == "coloured")
        r.set_scene(scene)

    # ---- one frame at a time, and frames delivered into host memory (render::GPU's definition of a frame,
    # render.cpp:468-475: dispatch + read-back), one GPU only; after the timed region
    host = None
    if world == 1 and not group and a.emulate_share <= 1 and not a.no_host_leg:
        n_h = max(F, min(a.steps, 100))
        t1 = time.perf_counter()
        for _ in range(n_h):
            r.render_device(rows, None, shard.data_ptr())
        alone_ms = (time.perf_counter() - t1) / n_h * 1e3
        bufs = [rm.HostBuffer((H, W, 4), "uint8") for _ in rs]
        whole = rm.Rows(H, 0, 1, 1)
        for rr, hb in zip(rs, bufs):
            rr.enqueue_host(whole, None, hb.ptr)
        drain()
        t1 = time.perf_counter()
        for i in range(n_h):
            rs[i % F].enqueue_host(whole, None, bufs[i % F].ptr)
        drain()
        host_ms = (time.perf_counter() - t1) / n_h * 1e3
        t1 = time.perf_counter()
        for _ in range(n_h):
            rm.lib().rm_render_rgba8(r._h, bufs[0].ptr)
        host_alone_ms = (time.perf_counter() - t1) / n_h * 1e3
        pageable = __import__("numpy").zeros((H, W, 4), "uint8")
        t1 = time.perf_counter()
        for _ in range(max(3, n_h // 4)):
            rm.lib().rm_render_rgba8(r._h, pageable.ctypes.data)
        pageable_ms = (time.perf_counter() - t1) / max(3, n_h // 4) * 1e3
        single = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        r.render_device(whole, None, single.data_ptr())
        same = all(bool((torch.from_numpy(hb.array) == single.cpu()).all()) for hb in bufs)
        host = {"value_host": round(rays_per_frame / host_ms / 1e3, 3), "unit": "Mrays/s", "frame_ms_host": round(host_ms, 4), "steps": n_h,
                "frames_in_flight": F, "host_frames_equal_device_frame": same,
                "frame_alone_ms": round(alone_ms, 4), "frame_alone_ms_host": round(host_alone_ms, 4),
                "frame_alone_ms_host_pageable": round(pageable_ms, 4),
                "note": "value_host: the same frames with every frame's RGBA8 image copied into page-locked host memory behind its kernels "
                        "(rm_enqueue_host), frames in flight as in the main region; frame_alone_ms*: one frame at a time, wall clock "
                        "per synchronous call (rm_render_device / rm_render_rgba8 into page-locked / pageable memory)"}
        for hb in bufs:
            hb.free()
    # ---- the one-process sharded path (rm_multi_*): four ranks, all on this GPU -- the ranks' kernels store their rows
    # straight into the caller's frame (what the library chooses where every device can write the first one's memory), and
    # for comparison shares gathered by copies (with distinct devices without peer access the same calls gather with RCCL
    # send/recv) -- one frame at a time and three in flight, into device memory; after the timed region
    native_multi = None
    if world == 1 and not group and a.emulate_share <= 1 and not a.no_host_leg:
        whole_rows = rm.Rows(H, 0, 1, 1)
        single = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        r.render_device(whole_rows, None, single.data_ptr())
        names = {rm.MULTI_COPY: "copies", rm.MULTI_RCCL: "rccl", rm.MULTI_DIRECT: "direct stores"}

        def multi_leg(transport):
            ring, outs = [], []
            try:
                for _ in range(3):
                    mm = rm.MultiRenderer([local] * 4, transport)
                    ring.append(mm)
                    mm.set_params(params)
                    mm.set_scene(scene)
                    outs.append(torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"))
                    for _ in range(4):
                        mm.render_device(outs[-1].data_ptr())
                m = ring[0]
                n_m = max(8, min(a.steps, 50))
                t1 = time.perf_counter()
                for _ in range(n_m):
                    m.render_device(outs[0].data_ptr())
                multi_ms = (time.perf_counter() - t1) / n_m * 1e3
                one_equal = bool(torch.equal(outs[0], single))
                # three frames in flight: three such objects take turns (rm_multi_enqueue_device / rm_multi_wait)
                for o in outs:
                    o.zero_()
                n_p = 3 * max(4, min(a.steps, 60) // 3)
                t1 = time.perf_counter()
                for i in range(n_p):
                    q = ring[i % 3]
                    q.wait()
                    q.enqueue_device(outs[i % 3].data_ptr())
                for q in ring:
                    q.wait()
                piped_ms = (time.perf_counter() - t1) / n_p * 1e3
                return {"transport": names[m.transport], "frame_ms": round(multi_ms, 4), "frame_equals_single_launch": one_equal,
                        "frame_ms_three_in_flight": round(piped_ms, 4),
                        "frames_in_flight_equal_single_launch": all(bool(torch.equal(o, single)) for o in outs)}
            finally:
                for mm in ring:
                    mm.close()

        native_multi = {"ranks": 4, "devices": [local] * 4}
        native_multi.update(multi_leg(rm.MULTI_AUTO))
        native_multi["gathered_by_copies"] = multi_leg(rm.MULTI_COPY)
        native_multi["note"] = ("rm_multi_render_device / rm_multi_enqueue_device: the frame's 4-row bands dealt to four contexts of this one "
                                "GPU, every rank storing its rows at their place in the caller's frame (no exchange step), and for comparison "
                                "shares gathered and de-interleaved by copies on rank 0 (a functional rehearsal of the one-process multi-GPU "
                                "path, not a scaling figure)")
    # ---- cold start: a fresh process uploads the scene and renders its first frame, with an empty cache directory
    # and again with what the first run left in it (compiled kernels, measured tile order)
    first = None
    if world == 1 and rank == 0 and not group and a.emulate_share <= 1 and not a.no_first_frame_leg:
        import subprocess
        import tempfile
        import shutil
        d = tempfile.mkdtemp(prefix="rm_bench_cache_")
        try:
            runs = []
            for leg in range(2):
                # the compiler keeps a cache of its own under the user's cache directory (this process has just compiled the
                # same kernels): off for the run that is to show a compile
                env = dict(os.environ, AMD_COMGR_CACHE="0") if leg == 0 else None
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_frame.py"), a.workload, d],
                                     capture_output=True, text=True, timeout=300, env=env)
                runs.append(json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else {"error": out.stderr[-300:]})
            first = {"empty_cache": runs[0], "warm_cache": runs[1],
                     "note": "tools/first_frame.py in a fresh process: rm_set_scene, then three rm_render_rgba8 into page-locked memory "
                             "(empty_cache: no code object anywhere, the compiler's own cache off too)"}
        finally:
            shutil.rmtree(d, ignore_errors=True)

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = rays_per_frame * a.steps / elapsed / 1e6
        if a.emulate_share > 1:
            desc += f" -- EMULATED: rank 0's share of a {a.emulate_share}-way split on one GPU, rate x {a.emulate_share}, no gather"
        # dominant kernel: the trace kernel of rank 0, its own executed flops over its own duration
        achieved = local_exec / (kernel_ms * 1e-3) / 1e12
        traffic, traffic_src, valu_insts = recorded_traffic(a.workload, a.evaluator, a.kernel_mode) if world == 1 and a.shading == "coloured" else (None, None, None)
        valu_issue = None
        if valu_insts:
            rate = valu_insts * 64 / (kernel_ms * 1e-3) / 1e12
            valu_issue = {"wave_insts_per_launch": valu_insts, "lane_insts_per_s_T": round(rate, 2), "measured_ceiling_T": VALU_ISSUE_CEILING_T,
                          "frac_of_ceiling": round(rate / VALU_ISSUE_CEILING_T, 3),
                          "frac_of_ceiling_frames_in_flight": round(valu_insts * 64 / (ms_per_step * 1e-3) / 1e12 / VALU_ISSUE_CEILING_T, 3)
                                                              if a.emulate_share <= 1 else None,
                          "note": "SQ_INSTS_VALU of a counter run of this workload (same file as traffic) over this run's kernel time "
                                  "(one frame alone on the GPU, bounded by its longest tiles) and over the time per frame "
                                  "with frames in flight, against the lane-instruction rate tools/valu_bench.hip sustains"}
        line = {
            "metric": "Mrays/s (primary+shadow), default scene", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "scene": scene_name, "width": W, "height": H, "ssaa_kernel": k,
                       "lights": n_lights, "primitives": cnt["prims"], "rays_per_frame": rays_per_frame,
                       "rays_cast_per_frame": int(total_rays),
                       "shading": ("samples on surfaces of colour (0,0,0) skip normal and shadow rays (product with the colour is zero); "
                                   "value counts the frame's nominal rays, rays_cast_per_frame those marched") if a.shading == "coloured"
                                  else "every sample (the reference's full work)",
                       "evaluator": a.evaluator, "kernel_mode": a.kernel_mode,
                       "set_scene_s": round(set_scene_s, 3), "kernel_compile_ms": round(cnt["compile_ms"], 1), "kernels_compiled_by": compiled_by,
                       "frames_in_flight": F, "frames_per_gather": G, "untimed_frames_before_warmup": pre_roll,
                       "sharding": ("single launch" + (", sent through a one-rank RCCL gather" if group else "")) if world == 1 else f"row bands of {a.band_rows} over {world} ranks, " + ("dealt by measured cost" if band_lists is not None else "cyclic") + ", RCCL gather of RGBA8 to rank 0"},
            "roofline": {
                "bound": "valu_fp32", "kernel": "rm_trace (generated for the scene)" if a.kernel_mode == "compiled" else "trace_kernel (record walker)", "achieved": round(achieved, 3), "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "achieved_frames_in_flight": round(total_exec / (ms_per_step * 1e-3) / 1e12, 3),
                "frac_frames_in_flight": round(total_exec / (ms_per_step * 1e-3) / 1e12 / (FP32_VECTOR_PEAK_TFLOPS * world), 4),
                "in_flight_note": "executed FLOPs of a frame over the time per frame of the timed region (all GPUs; resolve and gather included): "
                                  "the sustained rate when the tails of one frame's longest tiles are filled by other frames",
                "valu_issue": valu_issue,
                "kernel_ms": round(kernel_ms, 4), "kernel_ms_in_timed_region": round(overlapped_ms, 4), "kernel_ms_during_set_up": round(kernel_ms_before, 4),
                "kernel_ms_note": "kernel_ms: HIP events around the trace kernel with one frame on the GPU (used for achieved): the "
                                  "timed frames themselves with --frames-in-flight 1; otherwise frames rendered one at a time, back to "
                                  "back, right behind the timed region (frames in flight on several streams stretch each other's spans; "
                                  "kernel_ms_during_set_up: the same measured before the region, on a GPU that idled between frames)",
                "flops": "executed (counted primitive evaluations)",
                "executed_flops_per_frame": int(total_exec), "sdf_calls_per_frame": int(total_sdf),
                "prim_evals_per_sdf_call": round(cnt["prim_evals"] / max(1, cnt["sdf_calls"]), 2),
                "brute_force_equivalent": {"flops_per_sdf_call": cnt["flops_per_sdf"], "flops_per_frame": int(total_algo),
                                           "tflops": round(local_algo / (kernel_ms * 1e-3) / 1e12, 2)},
                "note": "peak is the FMA-every-slot datasheet figure; this path has no multiply-adds to fuse, its instruction mix tops out near 80 TFLOP/s (tools/valu_bench.hip)",
            },
            "gathered_frame_equals_single_launch": gathered_ok,
            "frame_ms": round(ms_per_step, 4),
            # the reference's definition of a frame (render::GPU, render.cpp:468-475: dispatch + read-back into the caller's
            # host image), one frame at a time, page-locked image; and the rate of every ray marched -- beside `value`
            "frame_alone_host_ms": host["frame_alone_ms_host"] if host else None,
            "value_frame_alone_host": round(rays_per_frame / host["frame_alone_ms_host"] / 1e3, 3) if host else None,
            "frame_alone_device_ms": host["frame_alone_ms"] if host else None,
            "value_every_ray_marched": all_marched["value"] if all_marched else None,
            "host_issue_ms_per_frame": round(host_issue_ms, 4),
            "tiles_rendered_again_by_the_exact_kernel": exact_tiles,
            "mrays_cast_per_s": round(total_rays * a.steps / elapsed / 1e6, 3),
            "every_ray_marched": all_marched,
            "host_image": host,
            "first_frame": first,
            "native_multi": native_multi,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "trace_kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
        }
        if exhaustive:
            line["roofline_exhaustive"] = exhaustive
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene_file, kw)
        print(json.dumps(line), flush=True)
    for rr in rs:
        rr.close()
    if group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
