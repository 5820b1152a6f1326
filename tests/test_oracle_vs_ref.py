"""The CPU oracle against the unmodified reference compiled in this container
(oracle/_ref/libref_harness.so), on fresh seeded inputs.  Skipped where the reference build is
absent (the GPU box has no reference checkout; the golden vectors cover that case)."""
import numpy as np
import pytest

from oracle import orc
from raytracer_amd import scenes
from raytracer_amd.desc import LIGHT_DT, MATERIAL_DT, MISS, OBJECT_DT, RAY_DT, SceneDesc

pytestmark = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (no reference checkout here)")


def _rays(rng, n, lo, hi, zero_frac=0.05):
    r = np.zeros(n, RAY_DT)
    r["e"] = rng.uniform(lo, hi, (n, 3))
    d = rng.uniform(0, 1, (n, 3)) - r["e"]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    idx = rng.choice(n, int(n * zero_frac), replace=False)
    for j, i in enumerate(idx):
        d[i, j % 3] = 0.0
    r["d"] = d
    return r


def test_constants():
    R = orc.ref()
    assert R.ref_const_zero() == 1e-9 and R.ref_const_infinite() == 1e10 and R.ref_const_pi() == 3.1415926


@pytest.mark.parametrize("seed,ntri,max_size,edge", [(11, 20000, 128, (3e-3, 5e-2)), (12, 5000, 8, (1e-2, 1e-1)),
                                                     (13, 300, 128, (1e-2, 3e-1)), (14, 40000, 64, (1e-3, 2e-2))])
def test_octree_matches_reference(seed, ntri, max_size, edge):
    """Tree shape, closest hit (id and t bit-exact) and number of primitive tests."""
    tris = scenes.soup(ntri, seed=seed, edge_lo=edge[0], edge_hi=edge[1])
    tree = orc.RefTree(tris, pad=1e-7, max_size=max_size)
    desc = SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT),
                     scenes._stack([scenes.material(1, color=(.5, .5, .5))], MATERIAL_DT), np.zeros(0, LIGHT_DT), [tris],
                     octree_max_size=max_size, octree_max_depth=15)
    s = orc.OracleScene(desc)
    assert s.octree_stats(0) == tree.stats()
    rays = _rays(np.random.default_rng(seed), 8000, -0.5, 1.5)
    for t0, t1 in ((0.0, 1e10), (1e-9, 0.7)):
        p1, d1, _, cnt = s.trace_rays(rays, t0, t1)
        p2, d2, tests = tree.hit(rays, t0, t1)
        assert np.array_equal(p1, p2) and np.array_equal(d1, d2)
        assert int(cnt[3]) == tests
    assert (p2 != MISS).sum() > 50


def test_cornell_walls_match_reference():
    """Large axis-aligned triangles (flat boxes, many stay at inner nodes)."""
    white, _, _ = scenes.cornell_box_meshes(100.0, tess=20)
    tree = orc.RefTree(white, pad=1e-7, max_size=16)
    desc = SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT),
                     scenes._stack([scenes.material(1, color=(.5, .5, .5))], MATERIAL_DT), np.zeros(0, LIGHT_DT), [white],
                     octree_max_size=16, octree_max_depth=15)
    s = orc.OracleScene(desc)
    assert s.octree_stats(0) == tree.stats()
    rng = np.random.default_rng(5)
    rays = np.zeros(6000, RAY_DT)
    rays["e"] = rng.uniform(5, 95, (6000, 3))
    d = rng.normal(0, 1, (6000, 3)); rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    p1, d1, _, _ = s.trace_rays(rays, 1e-9, 1e10)
    p2, d2, _ = tree.hit(rays, 1e-9, 1e10)
    assert np.array_equal(p1, p2) and np.array_equal(d1, d2)
    assert (p1 != MISS).mean() > 0.4        # three of the six box faces are in this mesh


def test_primitives_match_reference():
    rng = np.random.default_rng(21)
    for _ in range(1500):
        lo = rng.uniform(-1, 1, 3); hi = lo + rng.uniform(0.01, 2, 3)
        e = rng.uniform(-3, 3, 3); d = rng.normal(0, 1, 3)
        if rng.random() < 0.2:
            d[rng.integers(0, 3)] = 0.0
        t1 = 1e10 if rng.random() < 0.5 else rng.uniform(0.1, 5)
        assert orc.hit_box(lo, hi, e, d, 1e-9, t1)[0] == orc.ref_hit_box(lo, hi, e, d, 1e-9, t1)[0]
        a, b = orc.hit_box(lo, hi, e, d, 1e-9, t1), orc.ref_hit_box(lo, hi, e, d, 1e-9, t1)
        if a[0]:
            assert a == b
        c = rng.uniform(-1, 1, 3); rad = rng.uniform(0.1, 1.5)
        d2 = (c + rng.normal(0, 0.5, 3) * rad - e) * rng.uniform(0.2, 4)
        h1, tt1 = orc.sphere_hit(c, rad, e, d2, 1e-9, t1)
        h2, tt2, _ = orc.ref_sphere_hit(c, rad, e, d2, 1e-9, t1)
        assert h1 == h2 and (not h1 or tt1 == tt2)
        kind = int(rng.integers(2, 5)); coord = rng.uniform(-1, 1)
        h1, tt1 = orc.plane_hit(kind, coord, e, d, 1e-9, t1)
        h2, tt2, _ = orc.ref_plane_hit(kind, coord, e, d, 1e-9, t1)
        assert h1 == h2 and (not h1 or tt1 == tt2)


def test_camera_and_lights_match_reference():
    rng = np.random.default_rng(22)
    for _ in range(300):
        frm = rng.uniform(-20, 20, 3); at = frm + rng.uniform(-10, 10, 3) + 0.5
        vfov, asp = rng.uniform(20, 110), rng.uniform(0.4, 3)
        cam = orc.camera_init(frm, at, (0, 1, 0), vfov, asp)
        got = np.array(list(cam.e) + list(cam.llc) + list(cam.u) + list(cam.v) + [cam.width, cam.height])
        assert np.array_equal(got, orc.ref_camera_init(frm, at, (0, 1, 0), vfov, asp))
        s, t = rng.random(2)
        r = orc.camera_get_ray(cam, s, t)
        assert np.array_equal(np.concatenate([r["e"], r["d"]]), orc.ref_camera_get_ray(frm, at, (0, 1, 0), vfov, asp, s, t))
        pos = rng.uniform(-5, 5, 3); rad = rng.uniform(0.1, 2)
        x = pos + rng.normal(0, 1, 3) * rng.uniform(1.5, 8) * rad
        u, v = rng.random(2)
        l = scenes.sphere_light(pos, rad, (1, 1, 1))
        p1, n1, pdf1 = orc.light_sample(l, x, u, v)
        p2, n2, pdf2 = orc.ref_sphere_light_sample(pos, rad, x, u, v)
        if pdf2 > 1e10 or np.isnan(pdf2):
            assert pdf1 > 1e10 or np.isnan(pdf1)
        else:
            assert np.array_equal(p1, p2) and np.array_equal(n1, n2) and pdf1 == pdf2


def _implicit_desc(kind, center=(0, 0, 0), scale=1.0, min_size=0.0, max_size=128):
    from raytracer_amd.desc import OBJ_HEART  # noqa: F401
    return SceneDesc(scenes._stack([scenes.implicit_object(kind, center, scale, 0)], OBJECT_DT),
                     scenes._stack([scenes.phong_material((.1, .1, .1), (.6, .3, .3), (.4, .4, .4), 16.0)], MATERIAL_DT),
                     np.zeros(0, LIGHT_DT), [], implicit_min_size=min_size, octree_max_size=max_size, octree_max_depth=15)


def _rays_at(rng, n, radius, target=0.8):
    """Rays from a shell of the given radius towards points of a small ball at the origin."""
    r = np.zeros(n, RAY_DT)
    e = rng.normal(size=(n, 3)); e *= radius / np.linalg.norm(e, axis=1, keepdims=True)
    d = rng.uniform(-target, target, (n, 3)) - e
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::17, 0] = 0.0
    r["e"], r["d"] = e, d
    return r


@pytest.mark.parametrize("kind,min_size", [(6, 2e-2), (6, 7e-3), (7, 1.5e-2)])
def test_implicit_voxel_octree_matches_reference(kind, min_size):
    """ImplicitSurfaceOctree::build (octree.cpp:148-254) and Octree::hit over BoxNode voxels
    (Surface.cpp:43-59): the voxel list in set order, the tree shape and every hit distance are
    bit-identical to the reference's."""
    tree = orc.RefImplicit(kind, min_size)
    s = orc.OracleScene(_implicit_desc(kind, min_size=min_size))
    assert s.implicit_stats(kind) == tree.stats()
    a, b = s.implicit_boxes(kind), tree.boxes()
    assert a.shape == b.shape and a.shape[0] > 1000 and np.array_equal(a, b)
    rng = np.random.default_rng(kind)
    rays = np.concatenate([_rays_at(rng, 6000, 4.0), _rays_at(rng, 2000, 0.3, target=2.0)])   # from outside and from inside
    for t0, t1 in ((0.0, 1e10), (1e-9, 3.6)):
        p1, d1, _, _ = s.trace_rays(rays, t0, t1)
        h2, d2 = tree.hit(rays, t0, t1)
        assert np.array_equal(p1 != MISS, h2 != 0) and np.array_equal(d1, d2)
    assert 500 < (p1 != MISS).sum() < len(rays)
    for pos in rng.uniform(-1.5, 1.5, (200, 3)):
        assert orc.implicit_func(kind, pos) == orc.ref_implicit_func(kind, pos)


def test_default_heart_tree_has_the_reference_shape():
    """HeartShape's shared tree at the reference's default voxel size (3e-3): 475 052 voxels."""
    s = orc.OracleScene(_implicit_desc(6))
    st = s.implicit_stats(6)
    assert (st["nodes"], st["leaves"], st["refs"]) == (30745, 26902, 743759)


@pytest.mark.parametrize("kind", [6, 7])
def test_implicit_drawables_match_reference(kind):
    """HeartShape / SurfaceOfGenus2 through Node::calHit (ray into the surface's frame, analytic
    normal, localp) at the default voxel size, moved and scaled."""
    center, scale = (0.4, -0.3, 1.1), 1.7
    s = orc.OracleScene(_implicit_desc(kind, center, scale))
    rng = np.random.default_rng(40 + kind)
    rays = _rays_at(rng, 3000, 7.0, target=1.2)
    rays["e"] += center
    p1, t1, n1, _ = s.trace_rays(rays, 0.0, 1e10)
    h2, t2, n2, lp2 = orc.ref_implicit_object_hit(kind, center, scale, rays, 0.0, 1e10)
    assert np.array_equal(p1 != MISS, h2 != 0) and np.array_equal(t1, t2) and np.array_equal(n1, n2)
    assert (h2 != 0).sum() > 300


def test_perlin_noise_and_materials_match_reference():
    """algorithm::perlinNoise over the generator's tables, PerlinNoiseNormalMaterial and the
    _Blood variant (colours and tilted normal) -- algorithms.h:56-87, material.h:95-164,
    material.cpp:67-87."""
    P, G = orc.ref_perlin_tables(256)
    assert sorted(P) == list(range(256)) and np.allclose(np.linalg.norm(G, axis=1), 1.0)
    rng = np.random.default_rng(77)
    for _ in range(400):
        x, y, z = rng.uniform(-300, 300, 3) if rng.random() < 0.5 else rng.uniform(-3, 3, 3)
        sc = float(rng.choice([1.0, 0.37, 2.5]))
        assert orc.perlin_noise(x, y, z, sc, P, G) == orc.ref_perlin_noise(x, y, z, sc, P, G)
    amb, dif, spe = (.2, .1, .1), (.7, .5, .4), (.5, .6, .7)
    for kind in (1, 2):
        m = scenes.perlin_material(kind, amb, dif, spe, 20.0, 0.6)
        for _ in range(300):
            pos = rng.uniform(-4, 4, 3)
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            if rng.random() < 0.1:
                nrm = np.array([0.0, 0.0, 1.0])
            a = orc.perlin_material(m, P, G, pos, nrm)
            b = orc.ref_perlin_material(kind, amb, dif, spe, 20.0, 0.6, P, G, pos, nrm)
            assert np.array_equal(a, b), (kind, pos, a, b)


def test_tone_operators_match_reference():
    """SigmoidMethod / SqrtMethod::reproduction, tonereproductionmethod.h:11-35."""
    rng = np.random.default_rng(5)
    for kind in (1, 2):
        for _ in range(300):
            rgb = rng.uniform(-0.5, 6.0, 3)
            assert np.array_equal([orc.tone(kind, v) for v in rgb], orc.ref_tone(kind, rgb))


def test_perlin_time_evolution_matches_reference():
    """PerlinNoiseGeneratorWithTime::timeChanged (algorithms.cpp:57-77) along the driver's clock."""
    from raytracer_amd import hip
    _, G0 = orc.ref_perlin_tables(256)
    Ga, Gb, Gc = G0.copy(), G0.copy(), G0.copy()
    ca, cb = hip.PerlinClockC(), hip.PerlinClockC()
    count = stage = 0
    stime = prev = 0.0
    t = 0.0
    for _ in range(20):
        orc.perlin_time_changed(ca, prev, t, Ga)
        hip.perlin_time_changed(cb, prev, t, Gb)
        count, stage, stime = orc.ref_perlin_time_changed(count, stage, stime, prev, t, Gc)
        assert (ca.count, ca.stage, ca.stime) == (count, stage, stime) == (cb.count, cb.stage, cb.stime)
        assert np.array_equal(Ga, Gc) and np.array_equal(Gb, Gc)
        prev, t = t, t + 0.015
    assert stage == 4 and not np.array_equal(Gc, G0) and np.allclose(np.linalg.norm(Gc, axis=1), 1.0)


def test_animated_camera_and_light_match_reference():
    """ChangeFovCamera::timeChanged (camera.cpp:88-94) and MovableFaceLight::getPosition
    (light.cpp:101-107) under RTObject::setTime, along the driver's clock."""
    cam = dict(look_from=(1, 2, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=40.0, aspect=1.5)
    c0 = orc.camera_init(**cam)
    l0 = scenes.face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, (.5, .5, .6))
    t = 0.0
    for _ in range(140):
        c = orc.camera_change_fov(c0, t)
        got = np.array(list(c.e) + list(c.llc) + list(c.u) + list(c.v) + [c.width, c.height])
        assert np.array_equal(got, orc.ref_change_fov_camera(time=t, **cam))
        l = orc.light_move_face(l0, t)
        sample, pos = orc.ref_movable_face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, t, 0.3, 0.8)
        assert np.array_equal(l["pos"], pos)
        p, _, _ = orc.light_sample(l, (0, 0, 0), 0.3, 0.8)
        assert np.array_equal(p, sample)
        t += 0.015


def test_depth_camera_ray_matches_reference():
    """DepthCamera::getRay (camera.cpp:62-72) for the lens point the reference's randFromDisk drew."""
    cam = dict(look_from=(1, 2, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=40.0, aspect=1.5)
    c = orc.camera_init(**cam)
    c.aperture = 0.35
    rng = np.random.default_rng(2)
    for seed in range(200):
        s, t = rng.uniform(0, 1, 2)
        ray, xy = orc.ref_depth_camera_ray(aperture=0.35, seed=seed, s=s, t=t, **cam)
        assert xy[0] ** 2 + xy[1] ** 2 < 1.0
        r = orc.camera_get_ray_disk(c, s, t, xy[0], xy[1])
        assert np.array_equal(np.concatenate([r["e"], r["d"]]), ray)
