#!/usr/bin/env python3
"""Animated camera fly-through of the single-GPU benchmark scene, frame-partitioned.

    python tools/flythrough.py --frames 120 --out /tmp/fly
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/flythrough.py --frames 120 --out /tmp/fly

120 frames of 1280 x 720 at 128 spp (n = 16, m = 8) by default; the camera runs once around a
closed curve in front of and inside the open Cornell box, looking at the mesh.  Frame f is rendered
by rank f mod N on that rank's GPU (every rank holds the whole scene; no collective while
rendering).  A finished frame leaves the rendering stream through a FrameSink: its copy to the
host and its Motion-JPEG encoding into the AVI (one GPU) run on a worker thread while the next
frame renders; with several ranks rank 0 gathers the 8-bit frames and writes them in order.
Prints one JSON line: frames/s and Mrays/s of the whole job."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import render_params  # noqa: E402
from raytracer_amd.parallel import FrameSink, dist_env, frame_slice  # noqa: E402


def camera_on_path(u, aspect):
    """Camera at parameter u in [0, 1) of the closed curve."""
    return hip.camera_init(**scenes.flythrough_camera(u, aspect))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rank, local_rank, world = dist_env()
    torch.cuda.set_device(local_rank)
    hip.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo")                 # only finished 8-bit frames and counters travel
    desc, _ = scenes.bunny_cornell_scene()
    sc = hip.HipScene(desc)
    W, H = a.width, a.height
    prm = render_params(W, H, a.n, a.m, seed=1243)
    img = torch.zeros(H, W, 3, dtype=torch.float64, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    frames = {}
    avi = None
    if a.out and world == 1:
        os.makedirs(a.out, exist_ok=True)
        avi = hip.AviWriter(os.path.join(a.out, "flythrough.avi"), W, H, 15.0, mjpg_quality=95)

    def consume(f, arr):
        """Runs on the sink's worker thread while the next frame renders: one GPU writes the file as
        the frames arrive (they arrive in order), several ranks keep them for the gather."""
        if avi is not None:
            avi.add(arr)
            if f == a.frames // 3:
                hip.save_bmp(arr, os.path.join(a.out, "frame.bmp"))
        else:
            frames[f] = arr.copy()

    def run(nframes, sink):
        for f in frame_slice(rank, world, nframes):
            img.zero_()
            sc.render_dev(camera_on_path(f / a.frames, W / H), prm, img.data_ptr(), cnt.data_ptr(), stream)
            bgr = sink.device_buffer(f // world)
            hip.resolve_bgr8_dev(img.data_ptr(), W, H, a.n * a.m, bgr.data_ptr(), stream)
            sink.submit(f // world)

    warm = FrameSink(H, W, lambda f, arr: None, dev)
    run(min(world, a.frames), warm)                      # warm-up (workspace allocation)
    warm.close()
    cnt.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mine = frame_slice(rank, world, a.frames)
    sink = FrameSink(H, W, lambda k, arr: consume(mine[k], arr), dev)
    run(a.frames, sink)
    sink.close()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if avi is not None:
        avi.close()
    sc.check()
    rays = torch.tensor([float(cnt[0].item())], dtype=torch.float64)
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        parts = [None] * world if rank == 0 else None
        dist.gather_object(frames, parts, dst=0)
        if rank == 0 and a.out:
            os.makedirs(a.out, exist_ok=True)
            allf = {f: v for part in parts for f, v in part.items()}
            with hip.AviWriter(os.path.join(a.out, "flythrough.avi"), W, H, 15.0, mjpg_quality=95) as w:
                for f in range(a.frames):
                    w.add(allf[f])
    if rank == 0:
        dt = float(tmax.item())
        print(json.dumps({"workload": f"fly-through, {a.frames} frames {W}x{H} @ {a.n * a.m} spp, frame f on rank f mod {world}",
                          "n_gpus": world, "seconds": dt, "frames_per_s": a.frames / dt,
                          "value": float(rays.item()) / dt / 1e6, "unit": "Mrays/s",
                          "includes": "render + resolve + copy to the host + (one GPU) Motion-JPEG AVI write, the last three overlapped with the next frame's rendering"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
