"""The CPU oracle against the golden vectors captured from the unmodified reference
(tests/golden/make_golden.py).  Everything here is bit-exact: same fp64 expression order."""
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import scenes
from raytracer_amd.desc import LIGHT_DT, LIGHT_FACE, LIGHT_SPHERE, MISS, RAY_DT, OBJECT_DT, MATERIAL_DT, SceneDesc


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def test_hit_box_bit_exact(golden_dir):
    g = _load(golden_dir, "hit_box")
    n = len(g["lo"])
    nan_cases = 0
    for i in range(n):
        h, r0, r1 = orc.hit_box(g["lo"][i], g["hi"][i], g["e"][i], g["d"][i], float(g["t0"][i]), float(g["t1"][i]))
        eh, e0, e1 = g["out"][i]
        assert h == int(eh)
        if h:
            assert r0 == e0 and r1 == e1
        nan_cases += int(np.any(g["d"][i] == 0))
    assert nan_cases > 100          # the zero-direction (inf / NaN) cases really are in there


def test_sphere_and_planes_bit_exact(golden_dir):
    g = _load(golden_dir, "sphere_plane")
    n = len(g["c"])
    hits = 0
    rays = np.zeros(n, RAY_DT); rays["e"] = g["e"]; rays["d"] = g["d"]
    for i in range(n):
        h, t = orc.sphere_hit(g["c"][i], float(g["rad"][i]), g["e"][i], g["d"][i], 1e-9, float(g["t1"][i]))
        assert h == int(g["sph"][i, 0])
        if h:
            hits += 1
            assert t == g["sph"][i, 1]
        h, t = orc.plane_hit(int(g["kinds"][i]), float(g["coord"][i]), g["e"][i], g["d"][i], 1e-9, float(g["t1"][i]))
        assert h == int(g["pl"][i, 0])
        if h:
            assert t == g["pl"][i, 1]
    assert hits > 300


def test_sphere_normal_through_scene(golden_dir):
    """Normals stored by Drawable::doAfterHit: one-sphere scenes through the oracle's group path."""
    g = _load(golden_dir, "sphere_plane")
    mats = scenes._stack([scenes.material(1, color=(.5, .5, .5))], MATERIAL_DT)
    for i in range(0, 400):
        if not g["sph"][i, 0]:
            continue
        desc = SceneDesc(scenes._stack([scenes.sphere(g["c"][i], float(g["rad"][i]), 0)], OBJECT_DT), mats, np.zeros(0, LIGHT_DT), [])
        s = orc.OracleScene(desc)
        r = np.zeros(1, RAY_DT); r["e"][0] = g["e"][i]; r["d"][0] = g["d"][i]
        prim, t, nrm, _ = s.trace_rays(r, 1e-9, float(g["t1"][i]))
        assert prim[0] == 0 and t[0] == g["sph"][i, 1]
        assert np.array_equal(nrm[0], g["sph"][i, 2:5])
        s.close()


def test_child_bound_bit_exact(golden_dir):
    g = _load(golden_dir, "child_bound")
    for i in range(len(g["lo"])):
        for k in range(8):
            a, b = orc.child_bound(g["lo"][i], g["hi"][i], k)
            assert np.array_equal(a, g["out"][i, k, :3]) and np.array_equal(b, g["out"][i, k, 3:])


def test_camera_bit_exact(golden_dir):
    g = _load(golden_dir, "camera")
    for i in range(len(g["frm"])):
        cam = orc.camera_init(g["frm"][i], g["at"][i], (0, 1, 0), float(g["vfov"][i]), float(g["asp"][i]))
        got = np.array(list(cam.e) + list(cam.llc) + list(cam.u) + list(cam.v) + [cam.width, cam.height])
        assert np.array_equal(got, g["cams"][i])
        r = orc.camera_get_ray(cam, float(g["st"][i, 0]), float(g["st"][i, 1]))
        assert np.array_equal(np.concatenate([r["e"], r["d"]]), g["rays"][i])


def test_lights_bit_exact(golden_dir):
    g = _load(golden_dir, "lights")
    n = len(g["pos"])
    skipped = 0
    for i in range(n):
        l = np.zeros((), LIGHT_DT)
        l["type"], l["pos"], l["radius"] = LIGHT_SPHERE, g["pos"][i], g["rad"][i]
        p, nl, pdf = orc.light_sample(l, g["x"][i], float(g["uv"][i, 0]), float(g["uv"][i, 1]))
        exp = g["sl"][i]
        if exp[6] > 1e10:                         # "do not sample" sentinel (light.cpp:146-156)
            skipped += 1
            assert pdf > 1e10
            continue
        if np.isnan(exp).any():                   # shading point inside the light: NaN on both sides
            assert np.isnan(np.concatenate([p, nl, [pdf]])).any()
            continue
        assert np.array_equal(np.concatenate([p, nl, [pdf]]), exp)
        fl = scenes.face_light(g["pos"][i], g["nrm"][i], g["up"][i], float(g["wh"][i, 0]), float(g["wh"][i, 1]), (1, 1, 1))
        assert np.array_equal(fl["right"], g["fl"][i, 7:10])          # right = up ^ normal (light.cpp:51)
        assert np.array_equal(fl["normal"], g["fl"][i, 3:6])          # normalised normal
        p, nl, pdf = orc.light_sample(fl, g["x"][i], float(g["uv"][i, 0]), float(g["uv"][i, 1]))
        assert np.array_equal(np.concatenate([p, nl, [pdf]]), g["fl"][i, :7])
    assert skipped < n // 2


def test_constructed_edge_cases_bit_exact(golden_dir):
    """Branches that random inputs do not reach, with inputs built for them and the reference's own
    answers (tests/golden/make_golden.py, edge_cases): Sphere::hit's tangent branch, which reports a
    tangent point whatever the window (Surface.cpp:79-81), for discriminants of zero and on either
    side of 1e-9; SphereLight::getPosition's two "do not sample" exits (light.cpp:146-149, 153-156)
    with points on either side of both thresholds; planes with the origin on the plane
    (Surface.cpp:124-133, 194-204, 217-226)."""
    g = _load(golden_dir, "edge_cases")
    o = g["sph_out"]
    outside = 0
    for i in range(len(o)):
        t0, t1 = float(g["sph_t0"][i]), float(g["sph_t1"][i])
        h, t = orc.sphere_hit(g["sph_c"][i], float(g["sph_r"][i]), g["sph_e"][i], g["sph_d"][i], t0, t1)
        assert h == int(o[i, 0])
        if h:
            assert t == o[i, 1]
            outside += int(t < t0 or t > t1)
    assert outside > 100                       # tangent points reported although the window excludes them
    skipped = nans = 0
    for i in range(len(g["light_pos"])):
        l = np.zeros((), LIGHT_DT)
        l["type"], l["pos"], l["radius"] = LIGHT_SPHERE, g["light_pos"][i], g["light_r"][i]
        p, nl, pdf = orc.light_sample(l, g["light_x"][i], float(g["light_uv"][i, 0]), float(g["light_uv"][i, 1]))
        exp, got = g["light_out"][i], np.concatenate([p, nl, [pdf]])
        if exp[6] > 1e10:
            skipped += 1
            assert pdf > 1e10
        elif np.isnan(exp).any():              # a point inside the light: NaN on both sides
            nans += 1
            assert np.isnan(got).any()
        else:
            assert np.array_equal(got, exp)
    assert skipped > 50 and nans > 10 and skipped + nans < len(g["light_pos"])
    hits = 0
    for i in range(len(g["pl_kind"])):
        h, t = orc.plane_hit(int(g["pl_kind"][i]), float(g["pl_coord"][i]), g["pl_e"][i], g["pl_d"][i], float(g["pl_t0"][i]), float(g["pl_t1"][i]))
        assert h == int(g["pl_out"][i, 0])
        if h:
            hits += 1
            assert t == g["pl_out"][i, 1]
    assert hits > 50


@pytest.mark.parametrize("name", ["octree_soup_a", "octree_soup_b"])
def test_octree_build_and_traversal_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    tris = scenes.soup(int(g["ntri"]), seed=int(g["seed"]), edge_lo=5e-3, edge_hi=8e-2)
    desc = SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT),
                     scenes._stack([scenes.material(1, color=(.5, .5, .5))], MATERIAL_DT), np.zeros(0, LIGHT_DT), [tris],
                     octree_max_size=int(g["max_size"]), octree_max_depth=15)
    s = orc.OracleScene(desc)
    st = s.octree_stats(0)
    assert [st[k] for k in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")] == list(g["stats"])
    rays = np.zeros(len(g["e"]), RAY_DT); rays["e"] = g["e"]; rays["d"] = g["d"]
    prim, t, _, cnt = s.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"])
    assert int(cnt[3]) == int(g["tests"])          # same number of primitive tests as the reference made
    prim, t, _, _ = s.trace_rays(rays, 1e-9, 0.9)
    assert np.array_equal(prim, g["prim_clip"]) and np.array_equal(t, g["t_clip"])
    assert (g["prim"] != MISS).sum() > 100


def test_group_hit_bit_exact(golden_dir):
    g = _load(golden_dir, "group_hit")
    desc, _ = scenes.three_sphere_scene("face")
    s = orc.OracleScene(desc)
    rays = np.zeros(len(g["e"]), RAY_DT); rays["e"] = g["e"]; rays["d"] = g["d"]
    prim, t, nrm, _ = s.trace_rays(rays, 1e-9, 1e10)
    which = np.where(prim == MISS, -1, prim.astype(np.int64)).astype(np.int32)   # one primitive per object here
    assert np.array_equal(which, g["which"])
    hit = which >= 0
    assert np.array_equal(t[hit], g["t"][hit]) and np.array_equal(nrm[hit], g["normal"][hit])
    assert hit.sum() > 500


def _implicit_desc(kind, center=(0, 0, 0), scale=1.0, min_size=0.0):
    return SceneDesc(scenes._stack([scenes.implicit_object(kind, center, scale, 0)], OBJECT_DT),
                     scenes._stack([scenes.phong_material((.1, .1, .1), (.6, .3, .3), (.4, .4, .4), 16.0)], MATERIAL_DT),
                     np.zeros(0, LIGHT_DT), [], implicit_min_size=min_size, octree_max_depth=15)


@pytest.mark.parametrize("kind", [6, 7])
def test_implicit_surfaces_bit_exact(golden_dir, kind):
    """Voxelisation, voxel octree and BoxNode hits of the reference's implicit surfaces, then the
    drawables (frame change, analytic normals) at the reference's default voxel size."""
    g = _load(golden_dir, "implicit")
    k = f"k{kind}_"
    s = orc.OracleScene(_implicit_desc(kind, min_size=float(g[k + "min_size"])))
    st = s.implicit_stats(kind)
    assert [st[x] for x in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")] == list(g[k + "stats"])
    boxes = s.implicit_boxes(kind)
    assert len(boxes) == int(g[k + "nbox"]) and np.array_equal(boxes[::16], g[k + "boxes_every_16th"])
    assert np.array_equal(boxes.sum(axis=0), g[k + "box_sum"])
    rays = np.zeros(len(g[k + "e"]), RAY_DT); rays["e"] = g[k + "e"]; rays["d"] = g[k + "d"]
    prim, t, _, _ = s.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(prim != MISS, g[k + "hit"] != 0) and np.array_equal(t, g[k + "t"])
    prim, t, _, _ = s.trace_rays(rays, 1e-9, 3.6)
    assert np.array_equal(prim != MISS, g[k + "hit_clip"] != 0) and np.array_equal(t, g[k + "t_clip"])
    s = orc.OracleScene(_implicit_desc(kind, tuple(g[k + "center"]), float(g[k + "scale"])))
    rays = np.zeros(len(g[k + "obj_e"]), RAY_DT); rays["e"] = g[k + "obj_e"]; rays["d"] = g[k + "obj_d"]
    prim, t, nrm, _ = s.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(prim != MISS, g[k + "obj_hit"] != 0) and np.array_equal(t, g[k + "obj_t"])
    assert np.array_equal(nrm, g[k + "obj_normal"]) and (g[k + "obj_hit"] != 0).sum() > 200


def test_perlin_bit_exact(golden_dir):
    g = _load(golden_dir, "perlin")
    P, G = g["P"], g["G"]
    for (x, y, z), sc, ref in zip(g["xyz"], g["scale"], g["noise"]):
        assert orc.perlin_noise(x, y, z, float(sc), P, G) == ref
    for ki, kind in enumerate((1, 2)):
        m = scenes.perlin_material(kind, g["amb"], g["dif"], g["spe"], float(g["shine"]), float(g["mscale"]))
        for i in range(len(g["pos"])):
            assert np.array_equal(orc.perlin_material(m, P, G, g["pos"][i], g["normal"][i]), g["mats"][ki, i])
    # the generator replayed with the arithmetic of MSVC's rand(): a permutation and unit vectors
    P, G = orc.perlin_tables()
    assert sorted(P) == list(range(256)) and np.allclose(np.linalg.norm(G, axis=1), 1.0, atol=1e-15)


def test_tone_operators_bit_exact(golden_dir):
    g = _load(golden_dir, "tone")
    for x, a, b in zip(g["x"], g["sigmoid"], g["sqrt"]):
        assert np.array_equal([orc.tone(1, v) for v in x], a) and np.array_equal([orc.tone(2, v) for v in x], b)
