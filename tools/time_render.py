"""Time the path-tracing and primary kernels on one GPU (development aid)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import render_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--nu", type=int, default=186)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--soup", type=int, default=1000000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.scene == "bunny":
        desc, cam = scenes.bunny_cornell_scene(nu=a.nu, nv=a.nu + 1)
        W = H = a.size
    elif a.scene == "atrium":
        desc, cam = scenes.atrium_scene()
        W, H = a.size * 16 // 9, a.size
    else:
        desc, cam = scenes.soup_scene(a.soup)
        W, H = a.size * 16 // 9, a.size
    t0 = time.time()
    sc = hip.HipScene(desc)
    print("scene create s", time.time() - t0, [sc.octree_stats(i) for i in range(len(desc.meshes))], flush=True)
    c = hip.camera_init(**cam)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    if a.scene == "soup":
        prim = torch.empty(H * W, dtype=torch.int32, device=dev)
        tt = torch.empty(H * W, dtype=torch.float64, device=dev)
        for r in range(a.reps):
            cnt.zero_()
            torch.cuda.synchronize(); t0 = time.time()
            sc.trace_primary_dev(c, W, H, prim.data_ptr(), tt.data_ptr(), cnt.data_ptr())
            torch.cuda.synchronize(); dt = time.time() - t0
            k = cnt.cpu().numpy()
            print(f"primary {W}x{H}: {dt*1e3:.2f} ms  rays {k[0]} inner {k[1]} leaf {k[2]} prim {k[3]}  Mrays/s {k[0]/dt/1e6:.1f}  hits {(prim != -1).sum().item()}", flush=True)
        return
    img = torch.zeros(H, W, 3, dtype=torch.float64, device=dev)
    prm = render_params(W, H, a.n, seed=1)
    for r in range(a.reps):
        cnt.zero_(); img.zero_()
        torch.cuda.synchronize(); t0 = time.time()
        sc.render_dev(c, prm, img.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t0
        sc.check()
        k = cnt.cpu().numpy()
        print(f"render {W}x{H}@{a.n*a.n}: {dt*1e3:.1f} ms rays {k[0]} inner {k[1]} leaf {k[2]} prim {k[3]} Mrays/s {k[0]/dt/1e6:.1f} "
              f"Mpaths/s {W*H*a.n*a.n/dt/1e6:.2f} mean {img.mean().item()/(a.n*a.n):.4f}", flush=True)


if __name__ == "__main__":
    main()
