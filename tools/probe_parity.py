"""Quick GPU-vs-oracle probe (development aid; the real checks live in tests/)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import MISS, RAY_DT, render_params  # noqa: E402


def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-12, np.maximum(np.abs(a), np.abs(b)))


def main():
    print("devices", hip.device_count(), flush=True)
    rng = np.random.default_rng(0)
    # 1. soup
    desc, cam = scenes.soup_scene(20000, seed=3)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    print("stats", so.octree_stats(), sg.octree_stats(), flush=True)
    R = 50000
    rays = np.zeros(R, RAY_DT)
    rays["e"] = rng.random((R, 3)) * 3 - 1
    d = rng.random((R, 3)) - rays["e"]
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    p1, t1, n1, c1 = so.trace_rays(rays)
    t0 = time.time(); p2, t2, n2, c2 = sg.trace_rays(rays); print("gpu trace_rays s", time.time() - t0)
    print("soup: hits", (p1 != MISS).sum(), "id mismatch", (p1 != p2).sum(), "t mismatch", (t1 != t2).sum(),
          "normal mismatch", (n1 != n2).any(axis=1).sum(), "counters", c1, c2, flush=True)
    co = orc.camera_init(**cam); cg = hip.camera_init(**cam)
    assert bytes(co) == bytes(cg), "camera mismatch"
    a1, b1, _ = so.trace_primary(co, 320, 180)
    a2, b2, _ = sg.trace_primary(cg, 320, 180)
    print("soup primary: hits", (a1 != MISS).sum(), "id mismatch", (a1 != a2).sum(), "t mismatch", (b1 != b2).sum(), flush=True)

    # 2. analytic scene, per-path radiance
    for light in ("sphere", "face"):
        desc, cam = scenes.three_sphere_scene(light)
        so, sg = orc.OracleScene(desc), hip.HipScene(desc)
        co = orc.camera_init(**cam)
        W, H = 96, 72
        a1, b1, _ = so.trace_primary(co, W, H); a2, b2, _ = sg.trace_primary(co, W, H)
        print(light, "primary id mismatch", (a1 != a2).sum(), "t mismatch", (b1 != b2).sum(), "max rel t", rel(b1, b2).max())
        rays = np.zeros(W * H, RAY_DT)
        for i in range(W * H):
            r = orc.camera_get_ray(co, ((i % W) + .5) / W, ((i // W) + .5) / H)
            rays[i] = r
        prm = render_params(W, H, 1, seed=77)
        t0 = time.time(); c1, k1 = so.ray_color(rays, prm); to = time.time() - t0
        t0 = time.time(); c2, k2 = sg.ray_color(rays, prm); tg = time.time() - t0
        r = rel(c1, c2).max(axis=1)
        print(light, "ray_color: oracle s %.3f gpu s %.3f rays" % (to, tg), k1[0], k2[0], "max rel", r.max(), "frac>1e-4", (r > 1e-4).mean(),
              "frac>1e-9", (r > 1e-9).mean(), "mean", c1.mean(), c2.mean(), flush=True)
        prm = render_params(W, H, 4, seed=5)
        t0 = time.time(); i1, k1 = so.render(co, prm); to = time.time() - t0
        t0 = time.time(); i2, k2 = sg.render(co, prm); tg = time.time() - t0
        rr = rel(i1, i2)
        print(light, "render 4x4: oracle s %.3f gpu s %.3f" % (to, tg), "rays", k1[0], k2[0], "max rel", rr.max(), "rmse", np.sqrt(((i1 - i2) ** 2).mean()) / 16, flush=True)

    # 3. mesh scene
    desc, cam = scenes.bunny_cornell_scene(nu=48, nv=48, extras=True)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    print("bunny stats", so.octree_stats(3), sg.octree_stats(3))
    co = orc.camera_init(**cam)
    W = H = 96
    a1, b1, _ = so.trace_primary(co, W, H); a2, b2, _ = sg.trace_primary(co, W, H)
    print("bunny primary id mismatch", (a1 != a2).sum(), "t mismatch", (b1 != b2).sum())
    prm = render_params(W, H, 4, seed=9)
    t0 = time.time(); i1, k1 = so.render(co, prm); to = time.time() - t0
    t0 = time.time(); i2, k2 = sg.render(co, prm); tg = time.time() - t0
    rr = rel(i1, i2)
    print("bunny render 4x4: oracle s %.3f gpu s %.3f" % (to, tg), "rays", k1, k2, "max rel", rr.max(), "frac>1e-6", (rr > 1e-6).mean(),
          "rmse", np.sqrt(((i1 - i2) ** 2).mean()) / 16, "mean", i1.mean() / 16, flush=True)


if __name__ == "__main__":
    main()
