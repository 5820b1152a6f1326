#!/usr/bin/env python3
"""Frame-partitioned animation on one node: frame f is rendered by GPU f mod N.

    python examples/animation.py --frames 16 --out /tmp/anim
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \\
        examples/animation.py --frames 134 --out /tmp/anim

The loop of the reference's driver (main.cpp:83-101: `time += 0.015`, setTime, render, append
the frame) with the ChangeFovCamera and the MovableFaceLight applied per frame.  Every rank
holds the whole scene; frames are independent, so nothing is exchanged while rendering and
rank 0 only collects the finished 8-bit frames to write them in order (BMP files)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import render_params  # noqa: E402
from raytracer_amd.parallel import dist_env, render_frames  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--spp-root", type=int, default=4)
    ap.add_argument("--out", default="anim")
    a = ap.parse_args()
    rank, local_rank, world = dist_env()
    torch.cuda.set_device(local_rank)
    hip.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo")                 # finished frames are host byte arrays
    desc, cam, li = scenes.animation_scene(a.width, a.height)
    sc = hip.HipScene(desc)
    c0 = hip.camera_init(**cam)
    l0 = desc.lights[li].copy()
    prm = render_params(a.width, a.height, a.spp_root, seed=1243)

    def render_frame(f, time):
        sc.set_light(li, hip.light_move_face(l0, time))
        img, _ = sc.render(hip.camera_change_fov(c0, time), prm)
        return hip.resolve_bgr8_host(img, a.spp_root * a.spp_root)

    frames = render_frames(render_frame, a.frames, rank, world)
    if rank == 0:
        os.makedirs(a.out, exist_ok=True)
        with hip.AviWriter(os.path.join(a.out, "anim.avi"), a.width, a.height, 15.0, mjpg_quality=95) as avi:
            for f, img in enumerate(frames):
                hip.save_bmp(img, os.path.join(a.out, f"frame_{f:03d}.bmp"))
                avi.add(img)
        print(f"{len(frames)} frames -> {a.out}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
