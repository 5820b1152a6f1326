"""Properties of the CPU oracle that no reference fixture pins: the counter-based sampler,
refraction identities, analytic known-answer scenes for the integrator, sample-slice
additivity.  The whole-path radiance of ray_color is "parity unpinned" against the reference
(scene.cpp does not compile here); these tests pin it against closed forms instead."""
import math

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import scenes
from raytracer_amd.desc import (LIGHT_DT, MAT_NORMAL, MAT_SPECULAR, MAT_TRANSPARENT, MATERIAL_DT, OBJECT_DT, RAY_DT,
                                SceneDesc, render_params)


def test_permute_is_a_bijection():
    for l in (1, 2, 3, 16, 100, 256, 400, 1024):
        for p in (0, 1, 0xDEADBEEF, 12345):
            out = sorted(orc.lib().orc_permute(i, l, p) for i in range(l))
            assert out == list(range(l))


@pytest.mark.parametrize("n,m", [(4, 4), (16, 8), (20, 20), (1, 1), (5, 3)])
def test_jitter_table_structure(n, m):
    """RTObject::calRandom (RTObject.cpp:46-69): n x m strata, each used exactly once per pixel
    and object, pair k inside its stratum."""
    for pixel, obj in ((0, 0), (777, 0), (777, 1), (123456, 2)):
        cells = set()
        for k in range(n * m):
            a, b = orc.jitter_pair(1243, pixel, obj, k, n, m)
            assert 0.0 <= a < 1.0 and 0.0 <= b < 1.0
            cells.add((int(a * n), int(b * m)))
        assert len(cells) == n * m
    # different pixels / objects / seeds give different tables
    t = lambda s, p, o: [orc.jitter_pair(s, p, o, k, n, m) for k in range(n * m)]
    if n * m > 1:
        assert t(1, 5, 0) != t(1, 6, 0) and t(1, 5, 0) != t(1, 5, 1) and t(1, 5, 0) != t(2, 5, 0)
    assert t(1, 5, 0) == t(1, 5, 0)


def test_uniform_moments():
    L = orc.lib()
    key = L.orc_path_root_key(99, 1234, 5)
    u = np.array([L.orc_u01(L.orc_hash2(key, 16 + i)) for i in range(20000)])
    assert 0 <= u.min() and u.max() < 1
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_refract_identities():
    """refract (scene.cpp:384-395): Snell's law, unit length, total internal reflection."""
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = np.array([0.0, 1.0, 0.0])
        th = rng.uniform(0, 1.5)
        d = np.array([math.sin(th), -math.cos(th), 0.0])
        rc, t = orc.refract(d, n, 1.5)
        assert rc == 0 and abs(np.linalg.norm(t) - 1) < 1e-14
        assert abs(math.sin(th) - 1.5 * t[0]) < 1e-13          # sin(i) = n sin(t)
    # leaving glass beyond the critical angle (41.8 deg): reflected direction, code -1
    th = math.radians(60)
    d = np.array([math.sin(th), math.cos(th), 0.0])
    rc, t = orc.refract(d, np.array([0.0, -1.0, 0.0]), 1 / 1.5)
    assert rc == -1 and np.allclose(t, [math.sin(th), -math.cos(th), 0.0], atol=1e-15)
    # hand-computed: normal incidence keeps the direction
    rc, t = orc.refract(np.array([0.0, -1.0, 0.0]), np.array([0.0, 1.0, 0.0]), 1.5)
    assert rc == 0 and np.array_equal(t, [0.0, -1.0, 0.0])


def _one_sphere(mat, background=(0, 0, 0), lights=()):
    return SceneDesc(scenes._stack([scenes.sphere((0, 0, 0), 1.0, 0)], OBJECT_DT), scenes._stack([mat], MATERIAL_DT),
                     scenes._stack(list(lights), LIGHT_DT), [], background=background)


def _rays_at_sphere(n, seed=0):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, RAY_DT)
    e = rng.normal(0, 1, (n, 3)); e = 4 * e / np.linalg.norm(e, axis=1, keepdims=True)
    tgt = rng.normal(0, 0.15, (n, 3))
    d = tgt - e
    r["e"], r["d"] = e, d / np.linalg.norm(d, axis=1, keepdims=True)
    return r


def test_known_answer_emissive_and_miss():
    """Emission*E on the first hit (scene.cpp:247) and the clear colour on a miss (:378-380)."""
    mat = scenes.material(MAT_NORMAL, emission=(2, 3, 4), color=(0, 0, 0))
    s = orc.OracleScene(_one_sphere(mat, background=(.1, .2, .3)))
    rays = _rays_at_sphere(50)
    prm = render_params(1, 1, 1, seed=1)
    out, cnt = s.ray_color(rays, prm)
    # colour 0: nothing but emission survives; RR (colour 0 -> probability 0) ends the chain at depth 13
    assert np.array_equal(out, np.tile([2.0, 3.0, 4.0], (50, 1)))
    away = rays.copy(); away["d"] = -away["d"]
    out, _ = s.ray_color(away, prm)
    assert np.array_equal(out, np.tile([.1, .2, .3], (50, 1)))
    # depth cut-off: jumptime >= max_jump returns the clear colour without tracing (scene.cpp:196-198)
    out, cnt = s.ray_color(rays, prm, jumptime=40)
    assert np.array_equal(out, np.tile([.1, .2, .3], (50, 1))) and cnt[0] == 0


def test_known_answer_diffuse_under_uniform_sky():
    """A diffuse sphere with colour c under a uniform clear colour B and no lights: every
    bounce ray leaves the convex sphere, so radiance = c * B exactly (no cosine/pi factors in
    the reference's estimator, scene.cpp:247)."""
    c, B = np.array([.5, .25, .75]), np.array([.8, .4, .2])
    s = orc.OracleScene(_one_sphere(scenes.material(MAT_NORMAL, color=c), background=B))
    out, cnt = s.ray_color(_rays_at_sphere(200, 1), render_params(1, 1, 1, seed=5))
    assert np.allclose(out, np.tile(c * B, (200, 1)), rtol=1e-15, atol=0)
    assert cnt[0] == 400                      # one camera ray + one bounce ray each, no shadow rays


def test_known_answer_mirror():
    c, B = np.array([.9, .8, .7]), np.array([.3, .6, .9])
    s = orc.OracleScene(_one_sphere(scenes.material(MAT_SPECULAR, color=c), background=B))
    out, _ = s.ray_color(_rays_at_sphere(100, 2), render_params(1, 1, 1, seed=5))
    assert np.allclose(out, np.tile(c * B, (100, 1)), rtol=1e-15)


def test_known_answer_glass_conserves_energy():
    """Dielectric with no absorption under a uniform sky: R + (1-R) = 1 at every vertex, so the
    radiance is B (up to Russian-roulette terminations, which need depth > 12)."""
    B = np.array([.7, .7, .7])
    mat = scenes.material(MAT_TRANSPARENT, color=(1, 1, 1), atten=(0, 0, 0), nr=1.5)
    s = orc.OracleScene(_one_sphere(mat, background=B))
    out, _ = s.ray_color(_rays_at_sphere(40, 3), render_params(1, 1, 1, seed=5, max_jump=10, rus_jump=12))
    # with max_jump 10 the deepest branches return the clear colour too: still exactly B
    assert np.allclose(out, np.tile(B, (40, 1)), rtol=1e-12)


def test_nee_single_face_light_closed_form():
    """One diffuse ground plane, one small rectangular light straight above, first vertex only:
    Ld * cos * cos / (pdf * dist^2) with pdf = 1/(w*h) (scene.cpp:238, light.cpp:82)."""
    mats = scenes._stack([scenes.material(MAT_NORMAL, color=(1, 1, 1))], MATERIAL_DT)
    objs = scenes._stack([scenes.plane(2, 0.0, 0)], OBJECT_DT)
    w = h = 1e-3
    light = scenes.face_light((0, 2, 0), (0, -1, 0), (0, 0, 1), w, h, (5, 6, 7))
    s = orc.OracleScene(SceneDesc(objs, mats, scenes._stack([light], LIGHT_DT), [], background=(0, 0, 0)))
    r = np.zeros(1, RAY_DT); r["e"][0] = (0, 1, 0); r["d"][0] = (0, -1, 0)
    out, cnt = s.ray_color(r, render_params(1, 1, 1, seed=1, max_jump=1))
    expect = np.array([5, 6, 7]) * (w * h) / 4.0
    assert np.allclose(out[0], expect, rtol=1e-6)
    assert cnt[0] == 2                        # camera ray + shadow ray (bounce is cut by max_jump)


def test_phong_branch_closed_form():
    """Phong / Whitted branch (scene.cpp:305-375): a ground plane seen straight down, one point
    Light: ambient + diffuse * max(0, n.l) + specular * ((l - d)^ . n)^shine; a blocker between
    the point and the light leaves the ambient term only."""
    amb, dif, spe, shi = np.array([.1, .2, .3]), np.array([.6, .5, .4]), np.array([.9, .8, .7]), 5.0
    la, ld_, ls = np.array([.5, .5, .5]), np.array([2., 3., 4.]), np.array([1., .5, .25])
    mats = scenes._stack([scenes.phong_material(amb, dif, spe, shi), scenes.phong_material()], MATERIAL_DT)
    light = scenes.point_light((3, 4, 0), la, ld_, ls)
    r = np.zeros(1, RAY_DT); r["e"][0] = (0, 1, 0); r["d"][0] = (0, -1, 0)
    prm = render_params(1, 1, 1, seed=1)
    s = orc.OracleScene(SceneDesc(scenes._stack([scenes.plane(2, 0.0, 0)], OBJECT_DT), mats, scenes._stack([light], LIGHT_DT), [],
                                  background=(0, 0, 0)))
    out, cnt = s.ray_color(r, prm)
    l = np.array([3, 4, 0]) / 5.0
    n, d = np.array([0, 1, 0.]), np.array([0, -1, 0.])
    hv = (l - d) / np.linalg.norm(l - d)
    expect = la * amb + dif * ld_ * max(0.0, n @ l) + spe * ls * (hv @ n) ** shi
    assert np.allclose(out[0], expect, rtol=1e-14)
    assert cnt[0] == 2                        # the ray and one shadow ray
    blocked = scenes._stack([scenes.plane(2, 0.0, 0), scenes.sphere((1.5, 2, 0), 0.5, 1)], OBJECT_DT)
    s = orc.OracleScene(SceneDesc(blocked, mats, scenes._stack([light], LIGHT_DT), [], background=(0, 0, 0)))
    out, _ = s.ray_color(r, prm)
    assert np.allclose(out[0], la * amb, rtol=1e-15)


def test_phong_mirror_and_glass_weights():
    """Whitted mirror: getMirror() x the reflected radiance, nothing else (scene.cpp:341); glass
    (Material(a, nr)) without absorption under a uniform sky returns the sky (scene.cpp:343-373)."""
    B, mir = np.array([.3, .6, .9]), np.array([.5, .25, .125])
    mat = scenes.phong_material(mirror=mir)
    mat["type"] = MAT_SPECULAR                       # mirror lobe only
    s = orc.OracleScene(_one_sphere(mat, background=B))
    out, _ = s.ray_color(_rays_at_sphere(50, 2), render_params(1, 1, 1, seed=5))
    assert np.allclose(out, np.tile(mir * B, (50, 1)), rtol=1e-15)
    s = orc.OracleScene(_one_sphere(scenes.glass_material((0, 0, 0), 1.5), background=B))
    out, _ = s.ray_color(_rays_at_sphere(40, 3), render_params(1, 1, 1, seed=5, max_jump=10))
    assert np.allclose(out, np.tile(B, (40, 1)), rtol=1e-12)


def test_sample_slices_add_up():
    desc, cam = scenes.three_sphere_scene("sphere")
    s = orc.OracleScene(desc)
    c = orc.camera_init(**cam)
    W, H = 24, 18
    whole, k0 = s.render(c, render_params(W, H, 4, seed=3), threads=4)
    a, k1 = s.render(c, render_params(W, H, 4, sample_begin=0, sample_end=7, seed=3), threads=4)
    b, k2 = s.render(c, render_params(W, H, 4, sample_begin=7, sample_end=16, seed=3), threads=4)
    assert np.allclose(a + b, whole, rtol=1e-13, atol=1e-15)
    assert int(k1[0] + k2[0]) == int(k0[0])
    # thread count does not change anything (the sampler is keyed, not sequential)
    again, _ = s.render(c, render_params(W, H, 4, seed=3), threads=1)
    assert np.array_equal(again, whole)


def test_resolve_matches_reference_formula():
    rng = np.random.default_rng(1)
    img = rng.uniform(-0.2, 40, (5, 7, 3))
    out = orc.resolve_bgr8(img, 16)
    avg = np.clip(img * (1.0 / 16), 0, 1)
    exp = (avg * 255).astype(np.int64).astype(np.uint8)[:, :, ::-1]
    assert np.array_equal(out, exp)


def test_flat_triangles_survive_the_octree():
    """Axis-aligned triangles have flat AABBs; the strict overlap test (algorithms.h:329-333)
    would drop them without the box padding."""
    white, red, green = scenes.cornell_box_meshes(100.0, tess=12)
    desc = SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT),
                     scenes._stack([scenes.material(MAT_NORMAL, color=(.5, .5, .5))], MATERIAL_DT),
                     np.zeros(0, LIGHT_DT), [white], octree_max_size=32)
    s = orc.OracleScene(desc)
    st = s.octree_stats(0)
    assert st["lost"] == 0 and st["nodes"] > 1
    lost = SceneDesc(desc.objects, desc.materials, desc.lights, desc.meshes, tri_box_pad=0.0, octree_max_size=32)
    assert orc.OracleScene(lost).octree_stats(0)["lost"] > 0


def test_depth_camera_lens_properties():
    """DepthCamera::getRay (camera.cpp:62-72): every sample's ray starts on the lens disk
    (|offset| < aperture in the u, v plane) and passes through its film point, so a surface in
    the focal plane renders exactly as with the pinhole camera while others blur."""
    from raytracer_amd.desc import render_params
    # a textured-by-geometry scene would do; the three-sphere scene has edges at several depths
    desc, cam = scenes.three_sphere_scene("face")
    so = orc.OracleScene(desc)
    pin = orc.camera_init(**cam)
    lens = orc.camera_init(**cam)
    lens.aperture = 0.35
    W, H, n = 48, 36, 4
    prm = render_params(W, H, n, seed=5)
    a, ka = so.render(pin, prm)
    b, kb = so.render(lens, prm)
    assert not np.array_equal(a, b) and abs(a.mean() - b.mean()) < 0.25 * a.mean()
    # aperture 0 is the pinhole camera, bit for bit
    lens.aperture = 0.0
    c, kc = so.render(lens, prm)
    assert np.array_equal(a, c) and np.array_equal(ka, kc)
    # a tiny aperture converges to the pinhole image
    lens.aperture = 1e-9
    d, _ = so.render(lens, prm)
    assert np.abs(d - a).max() < 1e-3 * max(a.max(), 1.0) or (np.abs(d - a) > 1e-6).mean() < 0.02
