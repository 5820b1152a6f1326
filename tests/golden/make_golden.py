"""Generates the golden vectors in this directory from the UNMODIFIED reference.

Run in the build container (needs /root/reference and `make -C oracle`):
    python tests/golden/make_golden.py
Every .npz holds seeded inputs and the outputs the reference's own objects produced for them
through oracle/_ref/libref_harness.so (ref_harness.cpp): data only, no reference source.
The GPU box has no reference checkout; there the oracle and the HIP kernels are checked
against these files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.desc import RAY_DT  # noqa: E402


def rand_rays(rng, n, lo=-1.0, hi=2.0, axis_frac=0.05):
    r = np.zeros(n, RAY_DT)
    r["e"] = rng.uniform(lo, hi, (n, 3))
    d = rng.uniform(0, 1, (n, 3)) - r["e"]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # some rays with exactly-zero direction components (division by zero in the slab test)
    k = int(n * axis_frac)
    idx = rng.choice(n, k, replace=False)
    for j, i in enumerate(idx):
        d[i, j % 3] = 0.0
        if j % 5 == 0:
            d[i, (j + 1) % 3] = 0.0
    r["d"] = d
    return r


def shell_rays(rng, n, radius, target):
    """Rays from a shell of the given radius around the origin towards a small ball at its centre."""
    r = np.zeros(n, RAY_DT)
    e = rng.normal(size=(n, 3)); e *= radius / np.linalg.norm(e, axis=1, keepdims=True)
    d = rng.uniform(-target, target, (n, 3)) - e
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::17, 0] = 0.0
    r["e"], r["d"] = e, d
    return r


def edge_cases():
    """Constructed inputs for branches that random draws do not reach:
    Sphere::hit's tangent branch, which skips the [t0, t1] window (Surface.cpp:79-81) -- discriminants
    of exactly zero and of magnitude below / above 1e-9, with the tangent point inside and outside the
    window; SphereLight::getPosition's two "do not sample" exits (light.cpp:146-149: the shading point
    on the light's surface, light.cpp:153-156: cosmax >= 1 - 1e-9) with points on either side of both
    thresholds; the planes with the ray's origin exactly on, just above and just below the plane
    (Surface.cpp:124-133, 194-204, 217-226)."""
    # spheres with exactly representable centre / radius; rays along an axis that pass the centre at a
    # distance of radius + delta, so that the discriminant is (close to) -2 * radius * delta * |d|^2
    sc, sr, se, sd, st0, st1, sout = [], [], [], [], [], [], []
    for c, r in (((0.0, 0.0, 0.0), 1.0), ((2.0, -4.0, 8.0), 0.5), ((-16.0, 32.0, 1.0), 4.0)):
        for axis in range(3):
            for scale in (1.0, 2.0, 0.25):
                for delta in (0.0, 1e-11, -1e-11, 2e-10, -2e-10, 4e-10 / r, -4e-10 / r, 3e-9, -3e-9, 1e-6, -1e-6):
                    off = (axis + 1) % 3
                    e = np.array(c, float); e[axis] -= 5.0; e[off] += r + delta
                    d = np.zeros(3); d[axis] = scale
                    tt = 5.0 / scale                              # the tangent point's parameter
                    for t0, t1 in ((1e-9, 1e10), (1e-9, tt - 0.5), (tt + 0.5, 1e10), (1e-9, tt), (tt, 1e10)):
                        h, t, nn = orc.ref_sphere_hit(c, r, e, d, t0, t1)
                        sc.append(c); sr.append(r); se.append(e); sd.append(d); st0.append(t0); st1.append(t1)
                        sout.append((h, t if h else 0.0, *(nn if h else (0, 0, 0))))
    # sphere lights: shading points at distance radius + delta from the centre, and far points around
    # the distance where cosmax reaches 1 - 1e-9 (radius / distance = sqrt(2e-9) = 4.47e-5)
    lp, lr, lx, luv, lout = [], [], [], [], []
    for pos, r in (((0.0, 0.0, 0.0), 1.0), ((4.0, 8.0, -2.0), 0.5), ((-3.0, 1.0, 64.0), 2.0)):
        dirs = [np.array(v, float) for v in ((1, 0, 0), (0, -1, 0), (0, 0, 1), (0.6, 0.8, 0.0), (0.0, -0.6, 0.8))]
        for dv in dirs:
            for delta in (0.0, 5e-10, -5e-10, 9e-10, 2e-9, -2e-9, 1e-6, -1e-6):
                lp.append(pos); lr.append(r); lx.append(np.array(pos) + dv * (r + delta)); luv.append((0.3, 0.7))
            for ratio in (1e-6, 2e-5, 4.4e-5, 4.4721e-5, 4.4722e-5, 4.5e-5, 6e-5, 1e-3):
                lp.append(pos); lr.append(r); lx.append(np.array(pos) + dv * (r / ratio)); luv.append((0.62, 0.18))
    for i in range(len(lp)):
        p, nl, pdf = orc.ref_sphere_light_sample(lp[i], lr[i], lx[i], luv[i][0], luv[i][1])
        lout.append((*p, *nl, pdf))
    # planes: the origin's coordinate on the plane's axis is coord + delta
    pk, pc, pe, pd, pt0, pt1, pout = [], [], [], [], [], [], []
    for kind, axis in ((2, 1), (3, 2), (4, 0)):                    # Ground (y), Wall_z, Wall_x
        for coord in (0.0, 1.5, -8.0):
            for delta in (0.0, 1e-300, 5e-324, 1e-12, 5e-10, 2e-9, -1e-12, -1.0, 2.0):
                for dcomp in (-1.0, -1e-9, -0.0, 0.0, 1e-9, 1.0):
                    e = np.array([0.3, 0.4, 0.5]); e[axis] = coord + delta
                    d = np.array([0.2, 0.3, 0.1]); d[axis] = dcomp
                    for t0, t1 in ((1e-9, 1e10), (0.0, 1e10), (1e-9, 1.0)):
                        h, t, nn = orc.ref_plane_hit(kind, coord, e, d, t0, t1)
                        pk.append(kind); pc.append(coord); pe.append(e); pd.append(d); pt0.append(t0); pt1.append(t1)
                        pout.append((h, t if h else 0.0))
    np.savez_compressed(os.path.join(HERE, "edge_cases.npz"),
                        sph_c=np.array(sc), sph_r=np.array(sr), sph_e=np.array(se), sph_d=np.array(sd), sph_t0=np.array(st0), sph_t1=np.array(st1), sph_out=np.array(sout, float),
                        light_pos=np.array(lp), light_r=np.array(lr), light_x=np.array(lx), light_uv=np.array(luv), light_out=np.array(lout, float),
                        pl_kind=np.array(pk), pl_coord=np.array(pc), pl_e=np.array(pe), pl_d=np.array(pd), pl_t0=np.array(pt0), pl_t1=np.array(pt1), pl_out=np.array(pout, float))
    so = np.array(sout, float); lo_ = np.array(lout, float)
    print("edge cases: sphere", len(sout), "hits", int(so[:, 0].sum()), "| light", len(lout), "skipped", int((lo_[:, 6] > 1e10).sum()),
          "nan", int(np.isnan(lo_).any(axis=1).sum()), "| plane", len(pout), "hits", int(np.array(pout)[:, 0].sum()))


def main():
    assert orc.ref_available(), "build oracle/_ref first (make -C oracle)"
    if len(sys.argv) > 1 and sys.argv[1] == "edge":
        edge_cases()
        return
    rng = np.random.default_rng(20260101)

    # 1. hitBox (algorithms.h:223-236)
    n = 4000
    lo = rng.uniform(-1, 0.5, (n, 3)); hi = lo + rng.uniform(0.01, 1.5, (n, 3))
    rays = rand_rays(rng, n, -2, 2, 0.15)
    t0 = np.where(rng.random(n) < 0.5, 0.0, 1e-9); t1 = np.where(rng.random(n) < 0.7, 1e10, rng.uniform(0.1, 3, n))
    out = np.zeros((n, 3))
    for i in range(n):
        out[i] = orc.ref_hit_box(lo[i], hi[i], rays["e"][i], rays["d"][i], t0[i], t1[i])
    np.savez_compressed(os.path.join(HERE, "hit_box.npz"), lo=lo, hi=hi, e=rays["e"], d=rays["d"], t0=t0, t1=t1, out=out)

    # 2. Sphere::hit + normal, planes (Surface.cpp:73-111,124-133,194-204,217-226)
    n = 3000
    c = rng.uniform(-1, 1, (n, 3)); rad = rng.uniform(0.05, 1.2, n)
    rays = rand_rays(rng, n, -3, 3, 0.05)
    # aim most rays at the sphere, un-normalised directions included
    aim = c + rng.normal(0, 0.6, (n, 3)) * rad[:, None]
    d = aim - rays["e"]; d *= rng.uniform(0.3, 3.0, (n, 1))
    rays["d"] = d
    t1 = np.where(rng.random(n) < 0.6, 1e10, rng.uniform(0.2, 4, n))
    sph = np.zeros((n, 5))
    for i in range(n):
        h, t, nn = orc.ref_sphere_hit(c[i], rad[i], rays["e"][i], rays["d"][i], 1e-9, t1[i])
        sph[i] = (h, t if h else 0, *(nn if h else (0, 0, 0)))
    kinds = rng.integers(2, 5, n); coord = rng.uniform(-1, 1, n)
    pl = np.zeros((n, 2))
    for i in range(n):
        h, t, nn = orc.ref_plane_hit(int(kinds[i]), coord[i], rays["e"][i], rays["d"][i], 1e-9, t1[i])
        pl[i] = (h, t if h else 0)
    np.savez_compressed(os.path.join(HERE, "sphere_plane.npz"), c=c, rad=rad, e=rays["e"], d=rays["d"], t1=t1, sph=sph,
                        kinds=kinds, coord=coord, pl=pl)

    # 3. getChildBound (octree.h:82-120)
    n = 500
    lo = rng.uniform(-5, 5, (n, 3)); hi = lo + rng.uniform(1e-3, 7, (n, 3))
    cb = np.zeros((n, 8, 6))
    for i in range(n):
        for k in range(8):
            a, b = orc.ref_child_bound(lo[i], hi[i], k)
            cb[i, k, :3], cb[i, k, 3:] = a, b
    np.savez_compressed(os.path.join(HERE, "child_bound.npz"), lo=lo, hi=hi, out=cb)

    # 4. Camera::init / getRay (camera.cpp:24-48)
    n = 200
    frm = rng.uniform(-50, 50, (n, 3)); at = frm + rng.uniform(-30, 30, (n, 3)) + 1.0
    vfov = rng.uniform(20, 100, n); asp = rng.uniform(0.5, 2.5, n)
    cams = np.zeros((n, 14)); st = rng.random((n, 2)); crays = np.zeros((n, 6))
    for i in range(n):
        cams[i] = orc.ref_camera_init(frm[i], at[i], (0, 1, 0), vfov[i], asp[i])
        crays[i] = orc.ref_camera_get_ray(frm[i], at[i], (0, 1, 0), vfov[i], asp[i], st[i, 0], st[i, 1])
    np.savez_compressed(os.path.join(HERE, "camera.npz"), frm=frm, at=at, vfov=vfov, asp=asp, cams=cams, st=st, rays=crays)

    # 5. SphereLight / FaceLight::getPosition (light.cpp:76-86,138-190)
    n = 2000
    pos = rng.uniform(-5, 5, (n, 3)); rad = rng.uniform(0.1, 2, n)
    x = pos + rng.normal(0, 1, (n, 3)) * rng.uniform(1.0, 8.0, (n, 1)) * rad[:, None]
    x[:20] = pos[:20] + np.array([1.0, 0, 0]) * rad[:20, None] * 0.5      # inside the light sphere
    uv = rng.random((n, 2))
    sl = np.zeros((n, 7))
    for i in range(n):
        p, nl, pdf = orc.ref_sphere_light_sample(pos[i], rad[i], x[i], uv[i, 0], uv[i, 1])
        sl[i] = (*p, *nl, pdf)
    nrm = rng.normal(0, 1, (n, 3)); up = rng.normal(0, 1, (n, 3)); wh = rng.uniform(0.5, 5, (n, 2))
    fl = np.zeros((n, 10))
    for i in range(n):
        p, nl, pdf, right = orc.ref_face_light_sample(pos[i], nrm[i], up[i], wh[i, 0], wh[i, 1], uv[i, 0], uv[i, 1])
        fl[i] = (*p, *nl, pdf, *right)
    np.savez_compressed(os.path.join(HERE, "lights.npz"), pos=pos, rad=rad, x=x, uv=uv, sl=sl, nrm=nrm, up=up, wh=wh, fl=fl)

    # 6. the reference Octree (insertObject + hit, octree.cpp:61-108,285-376) over triangles
    for name, ntri, nray, max_size, seed in (("octree_soup_a", 6000, 6000, 128, 5), ("octree_soup_b", 3000, 4000, 16, 6)):
        tris = scenes.soup(ntri, seed=seed, edge_lo=5e-3, edge_hi=8e-2)
        tree = orc.RefTree(tris, pad=1e-7, max_size=max_size)
        rays = rand_rays(np.random.default_rng(seed), nray, -0.5, 1.5, 0.04)
        prim, t, tests = tree.hit(rays, 0.0, 1e10)
        prim2, t2, _ = tree.hit(rays, 1e-9, 0.9)
        st_ = tree.stats()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, ntri=ntri, max_size=max_size,
                            e=rays["e"], d=rays["d"], prim=prim, t=t, prim_clip=prim2, t_clip=t2, tests=tests,
                            stats=np.array([st_[k] for k in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")]))
        tree.close()

    # 7. Group::hit over spheres and planes (Surface.cpp:160-181)
    desc, cam = scenes.three_sphere_scene("face")
    objs = np.zeros((len(desc.objects), 6))
    for i, o in enumerate(desc.objects):
        objs[i, 0] = o["type"]
        objs[i, 1:5] = o["p"] if int(o["type"]) == 1 else (o["p"][0], 0, 0, 0)
    rays = rand_rays(rng, 5000, -6, 8, 0.03)
    rays["d"] = (rng.uniform(-2.5, 2.5, (5000, 3)) + np.array([0, 1, 0])) - rays["e"]
    which, t, nrm = orc.ref_group_hit(objs, rays, 1e-9, 1e10)
    np.savez_compressed(os.path.join(HERE, "group_hit.npz"), e=rays["e"], d=rays["d"], which=which, t=t, normal=nrm)
    # 8. implicit surfaces: the reference's voxel octree (ImplicitSurfaceOctree::build, octree.cpp:148-254;
    #    BoxNode::calHit, Surface.cpp:43-59) and the drawables HeartShape / SurfaceOfGenus2 through
    #    Node::calHit at the reference's default voxel size (normals, localp)
    out = {}
    for kind, min_size in ((6, 2e-2), (7, 1.5e-2)):
        tree = orc.RefImplicit(kind, min_size)
        st_ = tree.stats()
        boxes = tree.boxes()
        rr = np.random.default_rng(100 + kind)
        rays = np.concatenate([shell_rays(rr, 5000, 4.0, 0.8), shell_rays(rr, 1500, 0.3, 2.0)])
        h1, t1 = tree.hit(rays, 0.0, 1e10)
        h2, t2 = tree.hit(rays, 1e-9, 3.6)
        k = f"k{kind}_"
        out.update({k + "min_size": min_size, k + "stats": np.array([st_[x] for x in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")]),
                    k + "nbox": len(boxes), k + "boxes_every_16th": boxes[::16].copy(), k + "box_sum": boxes.sum(axis=0),
                    k + "e": rays["e"], k + "d": rays["d"], k + "hit": h1, k + "t": t1, k + "hit_clip": h2, k + "t_clip": t2})
        center, scale = (0.4, -0.3, 1.1), 1.7
        rays = shell_rays(rr, 2500, 7.0, 1.2)
        rays["e"] += center
        h, t, nrm, lp = orc.ref_implicit_object_hit(kind, center, scale, rays, 0.0, 1e10)
        out.update({k + "center": np.array(center), k + "scale": scale, k + "obj_e": rays["e"], k + "obj_d": rays["d"],
                    k + "obj_hit": h, k + "obj_t": t, k + "obj_normal": nrm, k + "obj_localp": lp})
    np.savez_compressed(os.path.join(HERE, "implicit.npz"), **out)
    # 9. Perlin noise and the Perlin materials (algorithms.h:56-87, material.h:95-164, material.cpp:67-87)
    #    over the tables the reference's generator builds in this container
    P, G = orc.ref_perlin_tables(256)
    rr = np.random.default_rng(909)
    xyz = np.concatenate([rr.uniform(-300, 300, (300, 3)), rr.uniform(-3, 3, (300, 3))])
    sc = rr.choice([1.0, 0.37, 2.5], len(xyz))
    noise = np.array([orc.ref_perlin_noise(x, y, z, s_, P, G) for (x, y, z), s_ in zip(xyz, sc)])
    amb, dif, spe = (.2, .1, .1), (.7, .5, .4), (.5, .6, .7)
    pos = rr.uniform(-4, 4, (400, 3))
    nrm = rr.normal(size=(400, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[::10] = (0.0, 0.0, 1.0)
    mats = np.stack([np.stack([orc.ref_perlin_material(kind, amb, dif, spe, 20.0, 0.6, P, G, pos[i], nrm[i]) for i in range(len(pos))])
                     for kind in (1, 2)])
    # the generator's vectors after 20 frames of the driver's clock (algorithms.cpp:57-77)
    Gt = G.copy()
    count = stage = 0
    stime = prev = tt = 0.0
    for _ in range(20):
        count, stage, stime = orc.ref_perlin_time_changed(count, stage, stime, prev, tt, Gt)
        prev, tt = tt, tt + 0.015
    np.savez_compressed(os.path.join(HERE, "perlin.npz"), G_after_20=Gt, clock_after_20=np.array([count, stage, stime]), P=P, G=G, xyz=xyz, scale=sc, noise=noise, amb=np.array(amb), dif=np.array(dif),
                        spe=np.array(spe), shine=20.0, mscale=0.6, pos=pos, normal=nrm, mats=mats)
    # 10. tone reproduction operators (tonereproductionmethod.h:11-35)
    x = rr.uniform(-0.5, 6.0, (400, 3))
    np.savez_compressed(os.path.join(HERE, "tone.npz"), x=x, sigmoid=np.stack([orc.ref_tone(1, v) for v in x]),
                        sqrt=np.stack([orc.ref_tone(2, v) for v in x]))
    # 11. ChangeFovCamera / MovableFaceLight under RTObject::setTime along the driver's clock
    #     (camera.cpp:88-94, light.cpp:101-107, main.cpp:83)
    cam = dict(look_from=(1, 2, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=40.0, aspect=1.5)
    times, cams, lpos, lsample = [], [], [], []
    tt = 0.0
    for _ in range(140):
        times.append(tt)
        cams.append(orc.ref_change_fov_camera(time=tt, **cam))
        smp, pos = orc.ref_movable_face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, tt, 0.3, 0.8)
        lpos.append(pos); lsample.append(smp)
        tt += 0.015
    np.savez_compressed(os.path.join(HERE, "animation.npz"), times=np.array(times), cams=np.stack(cams), light_pos=np.stack(lpos), light_sample=np.stack(lsample))
    # 12. constructed edge cases
    edge_cases()
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
