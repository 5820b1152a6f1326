"""Randomised differential run of whole frames of the HIP path tracer against the CPU oracle (development
aid; run on the GPU box; scenes as in tools/soak_radiance.py): random small scenes -- spheres, planes, blob and soup meshes, Monte-Carlo materials with any
mix of the diffuse, mirror (sharp or blurred) and dielectric lobes or Phong materials, one to three
lights of the three kinds -- seen through a random camera (a quarter of them with a lens) at 24..72 pixels a side, with square and
rectangular jitter tables, whole frames and frames put together from two sample slices.  Ray counts must be
the oracle's exactly and every pixel's sum within the per-path tolerance for path trees up to five deep
(deeper ones: see the comment at the check), the resolved bytes equal on all but a thousandth of the pixels.

    python tools/soak_render.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402  (the checker)
from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import (LIGHT_DT, MAT_NORMAL, MAT_SPECULAR, MAT_TRANSPARENT, MATERIAL_DT, OBJ_GROUND, OBJECT_DT, RAY_DT,  # noqa: E402
                                SceneDesc, render_params)

PATH_RTOL = 1e-4


def random_material(rng, phong_ok):
    if phong_ok and rng.random() < 0.25:
        return scenes.phong_material(ambient=tuple(rng.uniform(0, .3, 3)), diffuse=tuple(rng.uniform(.1, .9, 3)), specular=tuple(rng.uniform(0, .8, 3)),
                                     shine=float(rng.uniform(1, 40)), mirror=tuple(rng.uniform(0, .6, 3)) if rng.random() < 0.4 else None)
    kind = 0
    while kind == 0:
        kind = (MAT_NORMAL if rng.random() < 0.7 else 0) | (MAT_SPECULAR if rng.random() < 0.3 else 0) | (MAT_TRANSPARENT if rng.random() < 0.25 else 0)
    return scenes.material(kind, emission=tuple(rng.uniform(0, .3, 3)) if rng.random() < 0.2 else (0, 0, 0), color=tuple(rng.uniform(.2, .95, 3)),
                           blur=float(rng.uniform(0.01, 0.3)) if rng.random() < 0.3 else 0.0, atten=tuple(rng.uniform(0, .3, 3)), nr=float(rng.uniform(1.1, 1.8)))


def random_scene(rng):
    phong_ok = rng.random() < 0.4
    nmat = int(rng.integers(2, 6))
    mats = [random_material(rng, phong_ok) for _ in range(nmat)]
    objs, meshes = [], []
    objs.append(scenes.plane(OBJ_GROUND, 0.0, int(rng.integers(nmat))))
    for _ in range(int(rng.integers(1, 5))):
        r = float(rng.uniform(0.3, 1.2))
        objs.append(scenes.sphere((float(rng.uniform(-3, 3)), r + float(rng.uniform(0, 1.5)), float(rng.uniform(-3, 3))), r, int(rng.integers(nmat))))
    for _ in range(int(rng.integers(0, 3))):
        if rng.random() < 0.6:
            nu = int(rng.integers(6, 40))
            tris = scenes.blob((float(rng.uniform(-3, 3)), float(rng.uniform(1, 2.5)), float(rng.uniform(-3, 3))), float(rng.uniform(0.5, 1.2)), nu, nu + 1, seed=int(rng.integers(1 << 30)))
        else:
            tris = scenes.soup(int(rng.integers(50, 4000)), seed=int(rng.integers(1 << 30)), edge_lo=0.02, edge_hi=0.25) * 3.0 + np.tile((-1.5, 0.2, -1.5), 3)
        objs.append(scenes.mesh_object(len(meshes), int(rng.integers(nmat))))
        meshes.append(np.ascontiguousarray(tris))
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        k = rng.random()
        pos = (float(rng.uniform(-4, 4)), float(rng.uniform(4, 7)), float(rng.uniform(-4, 4)))
        if k < 0.4:
            rad = float(rng.uniform(0.2, 0.9))
            em = tuple(rng.uniform(5, 30, 3))
            lights.append(scenes.sphere_light(pos, rad, em))
            mats.append(scenes.material(MAT_NORMAL, emission=em, color=(0, 0, 0)))
            objs.append(scenes.sphere(pos, rad, len(mats) - 1))                  # Scene::addLight appends the light's geometry
        elif k < 0.8:
            lights.append(scenes.face_light(pos, (float(rng.uniform(-.5, .5)), -1, float(rng.uniform(-.5, .5))), (0, 0, 1), float(rng.uniform(.5, 3)), float(rng.uniform(.5, 3)),
                                            tuple(rng.uniform(3, 15, 3)), ambient=tuple(rng.uniform(0, .3, 3)), specular=tuple(rng.uniform(0, 1, 3))))
        else:
            lights.append(scenes.point_light(pos, tuple(rng.uniform(0, .4, 3)), tuple(rng.uniform(.3, 1, 3)), tuple(rng.uniform(0, 1, 3))))
    desc = SceneDesc(scenes._stack(objs, OBJECT_DT), scenes._stack(mats, MATERIAL_DT), scenes._stack(lights, LIGHT_DT), meshes,
                     background=tuple(rng.uniform(0, .3, 3)), octree_max_size=int(rng.choice([4, 16, 64, 128])))
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="evaluate this frame of the sequence alone")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    k = 0
    worst = 0.0
    while time.time() < t_end:
        desc = random_scene(rng)
        W, H = int(rng.integers(24, 73)), int(rng.integers(24, 73))
        n = int(rng.integers(1, 4))
        m = n if rng.random() < 0.6 else int(rng.integers(1, 5))
        mj = int(rng.integers(2, 10))
        eye = (float(rng.uniform(-7, 7)), float(rng.uniform(0.5, 6)), float(rng.uniform(5, 9)) * (1 if rng.random() < 0.5 else -1))
        cam = hip.camera_init(eye, (float(rng.uniform(-1, 1)), float(rng.uniform(0.5, 2)), float(rng.uniform(-1, 1))), (0, 1, 0), float(rng.uniform(25, 70)), W / H)
        if rng.random() < 0.25:
            cam.aperture = float(rng.uniform(0.02, 0.3))
        seed = int(rng.integers(1 << 30))
        common = dict(seed=seed, max_jump=mj, rus_jump=int(rng.integers(1, 8)), face_forward=bool(rng.random() < 0.3))
        limit = int(rng.uniform(0.8e9, 2e9)) if rng.random() < 0.5 else 0
        spp = n * m
        sliced = bool(rng.random() < 0.5 and spp > 1)
        cut = int(rng.integers(1, spp)) if sliced else 0
        if a.only >= 0 and k != a.only:
            k += 1
            if k > a.only:
                break
            continue
        so, sg = orc.OracleScene(desc), hip.HipScene(desc)
        if limit:
            sg.set_workspace_limit(limit)
        i1, k1 = so.render(cam, render_params(W, H, n, m, **common))
        if sliced:
            i2 = np.zeros((H, W, 3))
            _, ka = sg.render(cam, render_params(W, H, n, m, sample_begin=0, sample_end=cut, **common), i2)
            _, kb = sg.render(cam, render_params(W, H, n, m, sample_begin=cut, sample_end=spp, **common), i2)
            k2 = ka + kb
        else:
            i2, k2 = sg.render(cam, render_params(W, H, n, m, **common))
        nan1, nan2 = ~np.isfinite(i1), ~np.isfinite(i2)
        f1, f2 = np.where(nan1, 0.0, i1), np.where(nan2, 0.0, i2)
        rel = np.abs(f1 - f2) / np.maximum(1e-9, np.maximum(np.abs(f1), np.abs(f2)))
        worst = max(worst, float(rel.max()))
        same_tree = np.array_equal(k1, k2) if mj <= 5 else abs(int(k1[0]) - int(k2[0])) <= 1e-5 * int(k1[0])
        bytes_off = float((orc.resolve_bgr8(f1, spp) != orc.resolve_bgr8(f2, spp)).mean())
        # Deep trees: one ray in 10^6 or so ends differently on the two sides (see tools/soak_radiance.py; seen here
        # at depth 6 of one pixel in 2 500 frames, everything equal to 1e-14 up to depth 5), so a pixel in a thousand
        # may be off there, and the image as a whole must still agree (RMSE over the averaged frame).
        over = rel.max(axis=2) > PATH_RTOL
        rmse = float(np.sqrt((((f1 - f2) / spp) ** 2).mean()))
        pixels_ok = not over.any() if mj <= 5 else (over.mean() <= 1e-3 and rmse <= 1e-3)
        if not same_tree or not pixels_ok or not np.array_equal(nan1, nan2) or bytes_off > 1e-3:
            print("DIFFERENCE in scene", k, "seed", a.seed, (W, H, n, m), common, "counters", k1.tolist(), k2.tolist(), "max rel", float(rel.max()),
                  "not finite", int(nan1.sum()), int(nan2.sum()), "bytes off", bytes_off, flush=True)
            bad = np.argwhere(rel.max(axis=2) > PATH_RTOL)
            for y, x in bad[:4]:
                print(" pixel", (int(x), int(y)), "oracle", i1[y, x].tolist(), "device", i2[y, x].tolist(), flush=True)
            # the shallowest depth limit at which the two sides part
            for d in range(1, mj + 1):
                q = dict(common, max_jump=d)
                ia, kc = so.render(cam, render_params(W, H, n, m, **q))
                ib, kd = sg.render(cam, render_params(W, H, n, m, **q))
                rr = np.abs(ia - ib) / np.maximum(1e-9, np.maximum(np.abs(ia), np.abs(ib)))
                print("  max_jump", d, "rays", int(kc[0]), int(kd[0]), "max rel", float(np.nanmax(rr)), flush=True)
            sys.exit(1)
        print(f"scene {k}: {W}x{H} @ {n}x{m} depth {mj} objects {len(desc.objects)} meshes {[len(x) for x in desc.meshes]} lights {len(desc.lights)} "
              f"rays {int(k1[0])} max rel {float(rel.max()):.1e} ok", flush=True)
        k += 1
    print("all", k, "frames within tolerance; worst relative difference", worst, flush=True)


if __name__ == "__main__":
    main()
