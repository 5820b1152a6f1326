"""Randomised differential run of the HIP closest-hit path against the CPU oracle (development aid; run on
the GPU box): scenes of one to three random meshes with random octree parameters between spheres and
planes, rays of several kinds (uniform, axis-aligned, aimed at vertices and at octree splitting planes,
from inside), two windows each.  Every result must be the oracle's bit for bit: primitive ids, distances,
normals and the four work counters.  Prints one line per scene and stops at the first difference.

    python tools/soak_parity.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402  (the checker)
from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import LIGHT_DT, MAT_NORMAL, MATERIAL_DT, MISS, OBJ_GENUS2, OBJ_GROUND, OBJ_HEART, OBJECT_DT, RAY_DT, SceneDesc  # noqa: E402

OCTREE_KNOBS = False          # also vary the insert policy's overlap tolerance and the triangles' box padding (--knobs)
IMPLICIT_SHARE = 0.0          # share of the scenes that hold implicit surfaces (--implicit)


def random_scene(rng):
    nmesh = int(rng.integers(1, 4))
    meshes, objs = [], []
    mats = scenes._stack([scenes.material(MAT_NORMAL, color=(.5, .5, .5))], MATERIAL_DT)
    if rng.random() < 0.4:
        objs.append(scenes.sphere(tuple(rng.uniform(-0.5, 1.5, 3)), float(rng.uniform(0.05, 0.4)), 0))
    for k in range(nmesh):
        ntri = int(np.exp(rng.uniform(np.log(1), np.log(60000))))
        lo = float(np.exp(rng.uniform(np.log(1e-4), np.log(5e-2))))
        hi = lo * float(np.exp(rng.uniform(0, np.log(30))))
        tris = scenes.soup(ntri, seed=int(rng.integers(1 << 30)), edge_lo=lo, edge_hi=hi)
        if rng.random() < 0.3:                                    # vertices on a coarse lattice: many exact coincidences with splitting planes
            q = float(2.0 ** -int(rng.integers(2, 7)))
            tris = np.round(tris / q) * q
            if rng.random() < 0.5:
                # ... and moved off it by next to nothing: triangles that are collinear, or coplanar with a
                # splitting plane, to within 1e-13 .. 1e-4
                tris = tris + rng.normal(size=tris.shape) * float(10.0 ** -int(rng.integers(4, 14)))
        if rng.random() < 0.3:
            tris = tris * float(np.exp(rng.uniform(np.log(1e-3), np.log(1e3)))) + np.tile(rng.uniform(-1, 1, 3) * float(np.exp(rng.uniform(0, np.log(1e4)))), 3)
        meshes.append(np.ascontiguousarray(tris))
        objs.append(scenes.mesh_object(k, 0))
        if rng.random() < 0.3:
            objs.append(scenes.plane(OBJ_GROUND, float(rng.uniform(-1, 0)), 0))
    kw = dict(octree_max_size=int(rng.choice([2, 4, 8, 16, 32, 64, 128])), octree_max_depth=int(rng.integers(3, 16)))
    if OCTREE_KNOBS:
        # (a tolerance of 7 or 8 children, or a padding that is large against the smallest nodes, copies every
        # triangle into all children level after level: trees that nobody can build)
        kw["octree_overlap_tol"] = int(rng.integers(1, 6))
        kw["tri_box_pad"] = float(rng.choice([0.0, 1e-12, 1e-7, 1e-5]))
        kw["octree_max_depth"] = min(kw["octree_max_depth"], 9)
    if rng.random() < IMPLICIT_SHARE:
        # one or two implicit surfaces (voxel octrees) among the meshes, about as large as they are
        allv = np.concatenate([m.reshape(-1, 3) for m in meshes])
        lo, hi = allv.min(axis=0), allv.max(axis=0)
        for _ in range(int(rng.integers(1, 3))):
            objs.insert(int(rng.integers(0, len(objs) + 1)),
                        scenes.implicit_object(OBJ_HEART if rng.random() < 0.5 else OBJ_GENUS2, tuple(rng.uniform(lo, hi)), float(max(np.max(hi - lo), 1e-3) * rng.uniform(0.15, 0.5)), 0))
        kw["implicit_min_size"] = float(rng.uniform(2e-2, 5e-2))
    desc = SceneDesc(scenes._stack(objs, OBJECT_DT), mats, np.zeros(0, LIGHT_DT), meshes, **kw)
    return desc, meshes, kw


def random_rays(rng, meshes, n):
    allv = np.concatenate([m.reshape(-1, 3) for m in meshes])
    lo, hi = allv.min(axis=0), allv.max(axis=0)
    ext = np.maximum(hi - lo, 1e-6)
    r = np.zeros(n, RAY_DT)
    kind = rng.integers(0, 8, n)                                   # 5-7: at a point inside a triangle
    e = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, (n, 3))
    inside = kind == 3
    e[inside] = rng.uniform(lo, hi, (int(inside.sum()), 3))
    target = rng.uniform(lo, hi, (n, 3))
    pick = allv[rng.integers(0, len(allv), n)]
    target[kind == 1] = pick[kind == 1]                            # straight at a vertex
    tri = np.concatenate([m.reshape(-1, 3, 3) for m in meshes])
    w = rng.dirichlet((1, 1, 1), n)
    inner = (tri[rng.integers(0, len(tri), n)] * w[:, :, None]).sum(axis=1)
    target[kind >= 5] = inner[kind >= 5]
    # at a point of a splitting plane of the first mesh's octree (its bounds halved k times)
    m0 = meshes[0].reshape(-1, 3)
    l0, h0 = m0.min(axis=0), m0.max(axis=0)
    sp = kind == 2
    frac = rng.integers(0, 17, (n, 3)) / 16.0
    target[sp] = (l0 + frac * (h0 - l0))[sp]
    d = target - e
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
    ax = kind == 4                                                 # one or two exact zeros
    z = rng.integers(0, 3, n)
    d[ax, z[ax]] = 0.0
    z2 = rng.integers(0, 3, n)
    two = ax & (rng.random(n) < 0.3) & (z2 != z)
    d[two, z2[two]] = 0.0
    # a fifth of the rays between points of the 1/8 lattice (exactly rational directions, origins on splitting planes)
    lat = rng.random(n) < 0.2
    pts = np.round(rng.uniform(lo, hi, (n, 3)) * 8.0) / 8.0
    pts2 = np.round(rng.uniform(lo, hi, (n, 3)) * 8.0) / 8.0
    dl = pts2 - pts
    nl = np.linalg.norm(dl, axis=1, keepdims=True)
    lat &= nl[:, 0] > 0
    e[lat] = pts[lat]
    d[lat] = (dl / np.maximum(nl, 1e-300))[lat]
    bad = np.linalg.norm(d, axis=1) == 0
    d[bad] = (0, 0, 1)
    # one ray in twenty from far away, 1e2 .. 1e7 times the meshes' size back along its own line
    far = rng.random(n) < 0.05
    back = np.exp(rng.uniform(np.log(1e2), np.log(1e7), n)) * float(np.max(ext))
    e[far] = (e - d * back[:, None])[far]
    r["e"], r["d"] = e, d
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rays", type=int, default=20000)
    ap.add_argument("--implicit", type=float, default=0.0, help="share of the scenes with one or two implicit surfaces among the meshes")
    ap.add_argument("--knobs", action="store_true", help="also vary octree_overlap_tol and tri_box_pad")
    ap.add_argument("--only", type=int, default=-1, help="evaluate this scene of the sequence alone")
    ap.add_argument("--oracle-only", action="store_true", help="build the scenes and run the oracle alone (checks this script without a GPU)")
    a = ap.parse_args()
    global IMPLICIT_SHARE, OCTREE_KNOBS
    IMPLICIT_SHARE = a.implicit
    OCTREE_KNOBS = a.knobs
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    k = 0
    while time.time() < t_end:
        desc, meshes, kw = random_scene(rng)
        rays = random_rays(rng, meshes, a.rays)
        cam_draw = (rng.normal(size=3), float(rng.uniform(0.2, 2.0)), int(rng.integers(16, 200)), int(rng.integers(16, 200)), float(rng.uniform(10, 100)))
        if a.only >= 0 and k != a.only:
            k += 1
            if k > a.only:
                break
            continue
        so = orc.OracleScene(desc)
        sg = so if a.oracle_only else hip.HipScene(desc)
        ext = float(np.ptp(np.concatenate([m.reshape(-1, 3) for m in meshes])))
        for t0, t1 in ((0.0, 1e30), (1e-9 * ext, 0.7 * ext)):
            p1, d1, n1, c1 = so.trace_rays(rays, t0, t1)
            p2, d2, n2, c2 = sg.trace_rays(rays, t0, t1)
            same = np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2, equal_nan=True) and np.array_equal(c1, c2)
            if not same:
                bad = np.nonzero((p1 != p2) | (d1 != d2))[0]
                print("DIFFERENCE in scene", k, "seed", a.seed, kw, [len(m) for m in meshes], "window", (t0, t1), "rays", bad[:10], "counters", c1, c2, flush=True)
                for i in bad[:4]:
                    one = rays[i:i + 1]
                    print(" ray", i, "e", one["e"][0].tolist(), "d", one["d"][0].tolist())
                    print("  oracle", [x.tolist() for x in so.trace_rays(one, t0, t1)])
                    print("  device", [x.tolist() for x in sg.trace_rays(one, t0, t1)], flush=True)
                np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"soak_diff_{a.seed}_{k}.npz"),
                         rays=rays[bad[:16]], objects=desc.objects, t0=t0, t1=t1, max_size=kw["octree_max_size"], max_depth=kw["octree_max_depth"],
                         **{f"mesh{j}": m for j, m in enumerate(meshes)})
                sys.exit(1)
        # the primary-ray id buffer through a camera that looks at the meshes (single-precision pixel coordinates)
        allv = np.concatenate([m.reshape(-1, 3) for m in meshes])
        mid, span = (allv.min(axis=0) + allv.max(axis=0)) / 2, float(max(np.ptp(allv), 1e-3))
        eye = mid + cam_draw[0] * span * cam_draw[1]
        if np.linalg.norm(np.cross(mid - eye, (0, 1, 0))) > 1e-6 * span:
            W, H = cam_draw[2], cam_draw[3]
            cam = hip.camera_init(tuple(eye), tuple(mid), (0, 1, 0), cam_draw[4], W / H)
            a1, b1, q1 = so.trace_primary(cam, W, H)
            a2, b2, q2 = sg.trace_primary(cam, W, H)
            if not (np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(q1, q2)):
                print("DIFFERENCE in scene", k, "seed", a.seed, kw, "primary rays", (W, H), "pixels", np.nonzero((a1 != a2) | (b1 != b2))[0][:10], "counters", q1, q2, flush=True)
                sys.exit(1)
        print(f"scene {k}: meshes {[len(m) for m in meshes]} objects {len(desc.objects)} {kw} hit {float((p1 != MISS).mean()):.2f} counters {c1.tolist()} ok", flush=True)
        k += 1
    print("all", k, "scenes identical", flush=True)


if __name__ == "__main__":
    main()
