"""Configuration C1 at its own size -- the nine smallpt spheres at 512 x 384, 64 samples per pixel
(n = m = 8), face-forward shading -- on the GPU and through the CPU oracle on the box's host cores
(development aid; run on the GPU box).  Prints one JSON line for profiles/: both times, and the
GPU frame against the oracle's on evenly spaced image rows."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402  (the checker; never the thing measured as the product)
from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import render_params  # noqa: E402

W, H, n = 512, 384, 8
desc, cam = scenes.smallpt_scene()
sg, so = hip.HipScene(desc), orc.OracleScene(desc)
c = hip.camera_init(**cam)
prm = render_params(W, H, n, seed=1243, face_forward=1)
sg.render(c, prm)                                     # warm-up (workspace allocation)
times = []
for _ in range(3):
    t0 = time.perf_counter()
    img, k = sg.render(c, prm)
    times.append(time.perf_counter() - t0)
gpu_s = min(times)
threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
rows = np.unique(np.linspace(0, H - 1, 48).astype(int))
t0 = time.perf_counter()
worst, rays_cpu, rms = 0.0, 0, []
for y in rows:
    ref, kc = so.render(orc.camera_init(**cam), prm, threads=threads, pix_range=(W * int(y), W * int(y) + W))
    rays_cpu += int(kc[0])
    a, b = ref.reshape(H, W, 3)[y], img.reshape(H, W, 3)[y]
    rel = np.abs(a - b) / np.maximum(1e-9, np.maximum(np.abs(a), np.abs(b)))
    worst = max(worst, float(rel.max()))
    rms.append(float(np.sqrt((((a - b) / (n * n)) ** 2).mean())))
cpu_s = time.perf_counter() - t0
print(json.dumps({"workload": "C1: nine smallpt spheres as reference Spheres, 512x384 @ 64 spp (n = m = 8), face-forward, sphere light",
                  "gpu_ms": gpu_s * 1e3, "gpu_rays": int(k[0]), "gpu_Mrays_per_s": k[0] / gpu_s / 1e6,
                  "cpu_oracle": {"rows": int(len(rows)), "of": H, "seconds": cpu_s, "threads": threads, "rays": rays_cpu,
                                 "Mrays_per_s": rays_cpu / cpu_s / 1e6, "whole_frame_seconds_extrapolated": cpu_s * H / len(rows)},
                  "gpu_vs_oracle_on_those_rows": {"max_rel_pixel_sum": worst, "max_row_rmse_radiance": max(rms)},
                  "note": "pixel sums of 64 samples compared; the per-path and RMSE bars are in tests/test_gpu_parity.py"}))
