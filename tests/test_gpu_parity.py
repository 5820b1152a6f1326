"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors.

Bars: hit primitive ids, hit distances, normals and work counters are BIT-EXACT (integer /
index work and fp64 expressions evaluated in the same order on both sides); radiance is
compared with a tolerance because the device's sin / cos / exp / pow differ from glibc's in
the last bits and per-pixel sums are added in a different order: relative error <= 1e-4 per
path, image RMSE <= 1e-3 (the north-star tolerances), and in practice ~1e-9.
"""
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import hip, scenes
from raytracer_amd.desc import (LIGHT_DT, MAT_NORMAL, MAT_SPECULAR, MAT_TRANSPARENT, MATERIAL_DT, MISS, OBJ_GROUND, OBJECT_DT, RAY_DT,
                                SceneDesc, render_params)

pytestmark = pytest.mark.gpu

PATH_RTOL = 1e-4      # per-path relative tolerance
RMSE_TOL = 1e-3       # image RMSE tolerance (radiance units)


def _rel(a, b):
    return np.abs(a - b) / np.maximum(1e-9, np.maximum(np.abs(a), np.abs(b)))


def _rays(rng, n, lo, hi, zero_frac=0.04):
    r = np.zeros(n, RAY_DT)
    r["e"] = rng.uniform(lo, hi, (n, 3))
    d = rng.uniform(0, 1, (n, 3)) - r["e"]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    idx = rng.choice(n, int(n * zero_frac), replace=False)
    for j, i in enumerate(idx):
        d[i, j % 3] = 0.0                 # exact zeros: the slab test divides by them
    r["d"] = d
    return r


def _camera_rays(cam, W, H):
    rays = np.zeros(W * H, RAY_DT)
    for i in range(W * H):
        rays[i] = orc.camera_get_ray(cam, ((i % W) + .5) / W, ((i // W) + .5) / H)
    return rays


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def _mesh_desc(tris, **kw):
    return SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT),
                     scenes._stack([scenes.material(MAT_NORMAL, color=(.5, .5, .5))], MATERIAL_DT), np.zeros(0, LIGHT_DT),
                     [tris], **kw)


def _bmp_pixels(raw, W, H):
    """The (H, W, 3) payload of a 24-bit BMP whose rows are padded to four bytes."""
    stride = (3 * W + 3) & ~3
    return np.frombuffer(raw[54:], np.uint8).reshape(H, stride)[:, :3 * W].reshape(H, W, 3)


def _example(name):
    """Path of a built program under examples/ (built here with g++ if the tree came without it)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")], timeout=300)
    assert os.path.exists(exe), f"examples/{name} is not built (__graft_entry__.build())"
    return exe


def test_device_present_and_native_library_loaded():
    assert hip.device_count() >= 1
    assert os.path.exists(hip.LIB_PATH)


@pytest.mark.parametrize("name", ["octree_soup_a", "octree_soup_b"])
def test_octree_hits_match_reference_golden(golden_dir, name):
    """HIP traversal against hits recorded from the reference's own Octree::hit."""
    g = _load(golden_dir, name)
    tris = scenes.soup(int(g["ntri"]), seed=int(g["seed"]), edge_lo=5e-3, edge_hi=8e-2)
    s = hip.HipScene(_mesh_desc(tris, octree_max_size=int(g["max_size"]), octree_max_depth=15))
    st = s.octree_stats(0)
    assert [st[k] for k in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")] == list(g["stats"])
    rays = np.zeros(len(g["e"]), RAY_DT); rays["e"] = g["e"]; rays["d"] = g["d"]
    prim, t, _, cnt = s.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"])
    assert int(cnt[3]) == int(g["tests"])
    prim, t, _, _ = s.trace_rays(rays, 1e-9, 0.9)
    assert np.array_equal(prim, g["prim_clip"]) and np.array_equal(t, g["t_clip"])


def test_group_hit_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "group_hit")
    desc, _ = scenes.three_sphere_scene("face")
    s = hip.HipScene(desc)
    rays = np.zeros(len(g["e"]), RAY_DT); rays["e"] = g["e"]; rays["d"] = g["d"]
    prim, t, nrm, _ = s.trace_rays(rays, 1e-9, 1e10)
    which = np.where(prim == MISS, -1, prim.astype(np.int64)).astype(np.int32)
    assert np.array_equal(which, g["which"])
    hit = which >= 0
    assert np.array_equal(t[hit], g["t"][hit]) and np.array_equal(nrm[hit], g["normal"][hit])


def test_rays_that_meet_splitting_planes_exactly():
    """A node step finds the children a ray passes through from the order of the plane distances, and
    falls back to the reference's eight slab tests and its sort when two of the numbers that matter are
    equal (a middle plane met exactly at the window's end, two middle planes met together): rays that
    start on splitting planes with t0 = 0, rays through node centres and along node diagonals give such
    ties; ids, distances, normals and all four work counters must stay the oracle's."""
    rng = np.random.default_rng(23)
    tris = np.clip(scenes.soup(20000, seed=97, edge_lo=4e-3, edge_hi=4e-2), 0.0, 1.0)
    corners = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [1, 1, 1, 0, 1, 1, 1, 0, 1]], float)      # the mesh's bounds are the unit cube: splits at k / 2^d
    desc = _mesh_desc(np.ascontiguousarray(np.concatenate([corners, tris])), octree_max_size=32, octree_max_depth=12)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    n = 24000
    rays = np.zeros(n, RAY_DT)
    e = rng.integers(1, 32, (n, 3)) / 32.0                     # origins on splitting planes of every depth down to 5
    e[: n // 4] += rng.uniform(-0.3, 0.3, (n // 4, 3)) * (rng.random((n // 4, 3)) < 0.5)   # some coordinates off the planes
    target = rng.integers(0, 17, (n, 3)) / 16.0                # through node corners and centres
    target[n // 2:] = rng.uniform(0, 1, (n - n // 2, 3))
    d = target - e
    d[np.abs(d).sum(axis=1) == 0] = (1.0, 1.0, 1.0)
    diag = rng.random(n) < 0.2
    d[diag] = np.sign(d[diag] + 1e-30) * np.abs(d[diag]).max(axis=1, keepdims=True)        # exact diagonals: two or three middle planes met together
    rays["e"], rays["d"] = e, d
    for t0, t1 in ((0.0, 1e10), (0.0, 0.5), (0.25, 1.0)):
        p1, d1, n1, c1 = so.trace_rays(rays, t0, t1)
        p2, d2, n2, c2 = sg.trace_rays(rays, t0, t1)
        assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)
    assert (p1 != MISS).mean() > 0.2


def test_reference_edge_branches_on_device(golden_dir):
    """The constructed edge cases of the golden vectors through the kernels: tangent rays, whose hit is
    reported whatever the window (Surface.cpp:79-81), and rays that start on a plane
    (Surface.cpp:124-133, 194-204, 217-226) -- distances bit for bit the reference's; the two "do not
    sample" exits of SphereLight::getPosition (light.cpp:146-156) through per-path radiance."""
    g = _load(golden_dir, "edge_cases")
    mats = scenes._stack([scenes.material(MAT_NORMAL, color=(.5, .5, .5))], MATERIAL_DT)
    groups = {}
    for i in range(len(g["sph_out"])):
        groups.setdefault((tuple(g["sph_c"][i]), float(g["sph_r"][i]), float(g["sph_t0"][i]), float(g["sph_t1"][i])), []).append(i)
    outside = 0
    scene_of = {}
    for (c, r, t0, t1), idx in groups.items():
        if (c, r) not in scene_of:
            scene_of[(c, r)] = hip.HipScene(SceneDesc(scenes._stack([scenes.sphere(c, r, 0)], OBJECT_DT), mats, np.zeros(0, LIGHT_DT), []))
        rays = np.zeros(len(idx), RAY_DT); rays["e"] = g["sph_e"][idx]; rays["d"] = g["sph_d"][idx]
        prim, t, nrm, _ = scene_of[(c, r)].trace_rays(rays, t0, t1)
        exp = g["sph_out"][idx]
        assert np.array_equal(prim != MISS, exp[:, 0] > 0)
        hit = exp[:, 0] > 0
        assert np.array_equal(t[hit], exp[hit, 1]) and np.array_equal(nrm[hit], exp[hit, 2:5])
        outside += int(((t[hit] < t0) | (t[hit] > t1)).sum())
    assert outside > 100
    groups = {}
    for i in range(len(g["pl_out"])):
        groups.setdefault((int(g["pl_kind"][i]), float(g["pl_coord"][i]), float(g["pl_t0"][i]), float(g["pl_t1"][i])), []).append(i)
    scene_of = {}
    hits = 0
    for (kind, coord, t0, t1), idx in groups.items():
        if (kind, coord) not in scene_of:
            scene_of[(kind, coord)] = hip.HipScene(SceneDesc(scenes._stack([scenes.plane(kind, coord, 0)], OBJECT_DT), mats, np.zeros(0, LIGHT_DT), []))
        rays = np.zeros(len(idx), RAY_DT); rays["e"] = g["pl_e"][idx]; rays["d"] = g["pl_d"][idx]
        prim, t, _, _ = scene_of[(kind, coord)].trace_rays(rays, t0, t1)
        exp = g["pl_out"][idx]
        hit = exp[:, 0] > 0
        assert np.array_equal(prim != MISS, hit) and np.array_equal(t[hit], exp[hit, 1])
        hits += int(hit.sum())
    assert hits > 50
    # light.cpp:146-149: a ray that hits the light's own sphere is shaded on the light's surface, where the
    # light is not sampled: the path returns the sphere's emission and nothing else, on both sides alike
    lmats = scenes._stack([scenes.material(MAT_NORMAL, color=(.6, .6, .6)), scenes.material(MAT_NORMAL, emission=(7, 5, 3), color=(0, 0, 0))], MATERIAL_DT)
    desc = SceneDesc(scenes._stack([scenes.plane(OBJ_GROUND, 0.0, 0), scenes.sphere((0, 3, 0), 1.0, 1)], OBJECT_DT), lmats,
                     scenes._stack([scenes.sphere_light((0, 3, 0), 1.0, (7, 5, 3))], LIGHT_DT), [], background=(0, 0, 0))
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    rng = np.random.default_rng(5)
    n = 4000
    rays = np.zeros(n, RAY_DT)
    rays["e"] = rng.normal(size=(n, 3)); rays["e"] *= 6.0 / np.linalg.norm(rays["e"], axis=1, keepdims=True); rays["e"] += (0, 3, 0)
    d = (0, 3, 0) + rng.uniform(-0.6, 0.6, (n, 3)) - rays["e"]
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    prm = render_params(64, 64, 1, seed=3)
    c1, k1 = so.ray_color(rays, prm)
    c2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2)
    assert np.array_equal(c2, np.tile((7.0, 5.0, 3.0), (n, 1)))
    # light.cpp:153-156: a light so small and far that cosmax >= 1 - 1e-9 is not sampled either: the ground
    # under it stays black, while the same light a little larger lights it
    for radius, lit in ((1e-3, False), (1e-2, True)):
        desc = SceneDesc(scenes._stack([scenes.plane(OBJ_GROUND, 0.0, 0), scenes.sphere((0, 50, 0), radius, 1)], OBJECT_DT), lmats,
                         scenes._stack([scenes.sphere_light((0, 50, 0), radius, (7e6, 5e6, 3e6))], LIGHT_DT), [], background=(0, 0, 0))
        so, sg = orc.OracleScene(desc), hip.HipScene(desc)
        rays = np.zeros(n, RAY_DT)
        rays["e"] = rng.uniform(-3, 3, (n, 3)) + (0, 6, 0)
        d = rng.uniform(-4, 4, (n, 3)) * (1, 0, 1) - rays["e"]
        rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        c1, k1 = so.ray_color(rays, prm)
        c2, k2 = sg.ray_color(rays, prm)
        assert np.array_equal(k1, k2) and _rel(c1, c2).max() <= PATH_RTOL
        assert (c2.max() > 0) == lit


@pytest.mark.parametrize("seed,ntri,max_size,edge", [(31, 30000, 128, (3e-3, 5e-2)), (32, 4000, 8, (1e-2, 1e-1)),
                                                     (33, 200, 128, (1e-2, 3e-1)), (34, 60000, 32, (1e-3, 2e-2)),
                                                     (35, 0, 128, (1e-2, 1e-1)), (36, 1, 128, (2e-1, 5e-1))])
def test_trace_rays_bit_exact_vs_oracle(seed, ntri, max_size, edge):
    tris = scenes.soup(ntri, seed=seed, edge_lo=edge[0], edge_hi=edge[1])
    desc = _mesh_desc(tris, octree_max_size=max_size, octree_max_depth=15)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    assert so.octree_stats(0) == sg.octree_stats(0)
    rays = _rays(np.random.default_rng(seed), 20000, -0.5, 1.5)
    for t0, t1 in ((0.0, 1e10), (1e-9, 0.6)):
        p1, d1, n1, c1 = so.trace_rays(rays, t0, t1)
        p2, d2, n2, c2 = sg.trace_rays(rays, t0, t1)
        assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2)
        assert np.array_equal(c1, c2)         # rays, inner visits, leaf visits, primitive tests


@pytest.mark.parametrize("case", ["far", "slivers", "flat", "mixed_sizes", "on_split_planes"])
def test_box_pretest_never_rejects_a_hit(case):
    """The single-precision box pre-test works in a per-node integer grid; geometry that is hard
    on it must still give the oracle's ids, distances and test counts bit for bit: a mesh far from
    the origin (coordinates 1e8 times the triangle size), needle triangles, a mesh with no extent
    along one axis, triangles from 1e-4 to 1 of the mesh's size in one octree, and walls across the
    whole mesh that lie exactly in the octree's splitting planes (a record's box bounds the part
    of its triangle inside the node's box: here that part is a face of the box)."""
    rng = np.random.default_rng(91)
    if case == "far":
        tris = scenes.soup(8000, seed=91, edge_lo=1e-3, edge_hi=1e-2)
        off = np.array([1.0e5, -2.0e5, 3.0e5])
        tris = tris + np.tile(off, 3)
        lo, hi = off - 0.5, off + 1.5
    elif case == "slivers":
        tris = scenes.soup(6000, seed=92, edge_lo=1e-2, edge_hi=2e-1)
        # third vertex next to the middle of the first edge: needles with an angle of 1e-3 or so at v0 (thinner ones,
        # below 1e-5, are no triangles to the primitive test: its determinant would be within reach of its rounding error)
        tris[:, 6:9] = tris[:, 0:3] + (tris[:, 3:6] - tris[:, 0:3]) * 0.5 + rng.normal(0, 1e-4, (len(tris), 3))
        lo, hi = np.full(3, -0.5), np.full(3, 1.5)
    elif case == "flat":
        tris = scenes.soup(5000, seed=93, edge_lo=5e-3, edge_hi=5e-2)
        tris[:, 1::3] = 0.25                                   # every vertex at y = 0.25
        lo, hi = np.full(3, -0.5), np.full(3, 1.5)
    elif case == "mixed_sizes":
        tris = np.concatenate([scenes.soup(5000, seed=94, edge_lo=1e-4, edge_hi=1e-3), scenes.soup(300, seed=95, edge_lo=0.3, edge_hi=1.0)])
        lo, hi = np.full(3, -0.5), np.full(3, 1.5)
    else:
        walls = []
        for axis, c in ((0, 0.5), (1, 0.25), (2, 0.75), (0, 0.125), (1, 0.5), (2, 0.5), (0, 0.0), (0, 1.0), (1, 0.0), (1, 1.0), (2, 0.0), (2, 1.0)):
            a, b = [k for k in range(3) if k != axis]
            q = np.zeros((4, 3))
            q[:, axis] = c
            q[:, a] = (0, 1, 1, 0)
            q[:, b] = (0, 0, 1, 1)
            walls += [np.concatenate([q[0], q[1], q[2]]), np.concatenate([q[0], q[2], q[3]])]
        pebbles = np.clip(scenes.soup(6000, seed=96, edge_lo=2e-3, edge_hi=2e-2), 0.0, 1.0)       # the mesh's bounds are the unit cube: splits at k / 2^d
        tris = np.concatenate([np.array(walls), pebbles])
        lo, hi = np.full(3, -0.25), np.full(3, 1.25)
    desc = _mesh_desc(np.ascontiguousarray(tris), octree_max_size=16, octree_max_depth=15)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    assert so.octree_stats(0) == sg.octree_stats(0)
    n = 30000
    rays = np.zeros(n, RAY_DT)
    rays["e"] = rng.uniform(lo, hi, (n, 3))
    # aim at points of the mesh so that most rays hit something, many of them at grazing distance
    v = tris.reshape(-1, 3, 3)
    pick = rng.integers(0, len(v), n)
    w = rng.dirichlet((1, 1, 1), n)
    target = (v[pick] * w[:, :, None]).sum(axis=1)
    d = target - rays["e"]
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    p1, d1, n1, c1 = so.trace_rays(rays, 0.0, 1e10)
    p2, d2, n2, c2 = sg.trace_rays(rays, 0.0, 1e10)
    assert (p1 != MISS).mean() > 0.5
    assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)


def test_determinants_that_are_rounding_noise_are_no_hits():
    """Two inputs whose Moeller-Trumbore determinant is rounding noise instead of the zero it stands for, so that
    the quotients of the test are noise over noise: a triangle with two corners in one place (its edge is not
    axis-aligned, so the products do not cancel exactly), which used to be "hit" at distance -0.0 by this ray
    from 0.7 units away; and a ray between two lattice points that runs in the plane of a lattice triangle it
    does not touch, with numerators of exactly zero, which used to "hit" it at t = 0.  Both are rejected by the
    test itself now (the determinant must stand clear of its own rounding error), on both sides alike; before,
    the oracle accepted them and the device's box pre-test, which follows the geometry, did not."""
    T = np.array([[0.125, 0.5, 0.5, 0.0, 0.375, 0.625, 0.0, 0.375, 0.625]])
    fill = np.round(scenes.soup(60, seed=98, edge_lo=0.1, edge_hi=0.3) * 8.0) / 8.0
    c0 = np.array([[-0.125, -0.125, -0.125, 0, -0.125, -0.125, -0.125, 0, -0.125]])         # pin the mesh's bounds to [-1/8, 9/8]^3
    c1 = np.array([[1.125, 1.125, 1.125, 1, 1.125, 1.125, 1.125, 1, 1.125]])
    # the lattice triangle of the second case: normal (1, 1, 2), in the plane x + y + 2 z = 15/8
    P = np.array([[0.15625, 0.46875, 0.625, 0.09375, 0.46875, 0.65625, 0.125, 0.4375, 0.65625]])
    desc = _mesh_desc(np.ascontiguousarray(np.concatenate([fill[:30], T, fill[30:], P, c0, c1])), octree_max_size=4, octree_max_depth=10)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    assert so.octree_stats(0) == sg.octree_stats(0)
    rays = np.zeros(2, RAY_DT)
    rays["e"][0] = [0.9286631233312823, 0.07916902727209418, 0.3983703657320463]
    rays["d"][0] = [-0.6274751611070144, 0.7602633351402673, -0.16815047854562842]
    rays["e"][1] = [0.125, 0.25, 0.75]                                                      # in the plane, direction (5, 1, -3) / sqrt(35)
    rays["d"][1] = [0.8451542547285166, 0.1690308509457033, -0.50709255283711]
    assert orc.triangle_hit(T[0], rays["e"][0], rays["d"][0], 0.0, 1e30)[0] == 0
    assert orc.triangle_hit(P[0], rays["e"][1], rays["d"][1], 0.0, 1e30)[0] == 0
    p1, d1, n1, c1_ = so.trace_rays(rays, 0.0, 1e30)
    p2, d2, n2, c2_ = sg.trace_rays(rays, 0.0, 1e30)
    assert p1[0] != 30 and p1[1] != 61
    assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2, equal_nan=True) and np.array_equal(c1_, c2_)
    # and among many rays through the same mesh, a fifth of them between lattice points
    more = _rays(np.random.default_rng(98), 20000, -0.1, 1.1)
    rng = np.random.default_rng(99)
    a, b = np.round(rng.uniform(0, 1, (4000, 3)) * 8) / 8, np.round(rng.uniform(0, 1, (4000, 3)) * 8) / 8
    ok = np.linalg.norm(b - a, axis=1) > 0
    more["e"][:4000][ok] = a[ok]
    more["d"][:4000][ok] = ((b - a) / np.maximum(np.linalg.norm(b - a, axis=1, keepdims=True), 1e-300))[ok]
    p1, d1, n1, c1_ = so.trace_rays(more, 0.0, 1e30)
    p2, d2, n2, c2_ = sg.trace_rays(more, 0.0, 1e30)
    assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(n1, n2, equal_nan=True) and np.array_equal(c1_, c2_)


def test_primary_id_buffer_bit_exact_small():
    desc, cam = scenes.soup_scene(50000, seed=9)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    assert bytes(c) == bytes(orc.camera_init(**cam))
    for W, H in ((640, 360), (37, 21)):       # a ragged size too: not a multiple of the tile
        a1, b1, k1 = so.trace_primary(c, W, H)
        a2, b2, k2 = sg.trace_primary(c, W, H)
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(k1, k2)
        assert (a1 != MISS).sum() > (50 if W > 100 else 0)


def test_primary_id_buffer_full_size_properties():
    """The benchmark-size id buffer (1 M triangles, 3840 x 2160): checked through properties
    that need no full CPU render -- a random subset of pixels against the oracle (bit-exact),
    and for EVERY hit pixel the reported t re-derived from the reported triangle alone."""
    desc, cam = scenes.soup_scene(1_000_000, seed=42)
    sg = hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 3840, 2160
    prim, t, cnt = sg.trace_primary(c, W, H)
    assert int(cnt[0]) == W * H
    hit = prim != MISS
    assert 0.05 < hit.mean() < 0.95
    assert np.all(t[~hit] == 1e10) and np.all(t[hit] > 0) and np.all(prim[hit] < 1_000_000)
    # subset vs oracle
    so = orc.OracleScene(desc)
    rng = np.random.default_rng(1)
    ys, xs = rng.integers(0, H, 20000), rng.integers(0, W, 20000)
    rays = np.zeros(20000, RAY_DT)
    for i in range(20000):
        rays[i] = orc.camera_get_ray(c, (xs[i] + .5) / W, (ys[i] + .5) / H)
    p1, t1, _, _ = so.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(p1, prim[ys, xs]) and np.array_equal(t1, t[ys, xs])
    # every hit: t equals the single-triangle test of the reported primitive (numpy restatement)
    tri = desc.meshes[0]
    yy, xx = np.nonzero(hit)
    sel = rng.choice(len(yy), 200000, replace=False)
    yy, xx = yy[sel], xx[sel]
    s_ = ((0.5 + xx * 1.0) / W).astype(np.float32).astype(np.float64)
    t_ = ((0.5 + yy * 1.0) / H).astype(np.float32).astype(np.float64)
    e = np.array(list(c.e)); llc = np.array(list(c.llc)); u = np.array(list(c.u)); v = np.array(list(c.v))
    look = llc + (u[None] * s_[:, None]) * c.width + (v[None] * t_[:, None]) * c.height
    d = look - e
    d = d * (1 / np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]))[:, None]
    tr = tri[prim[yy, xx]]
    v0, e1, e2 = tr[:, 0:3], tr[:, 3:6] - tr[:, 0:3], tr[:, 6:9] - tr[:, 0:3]
    cross = lambda a, b: np.stack([a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]], 1)
    dot = lambda a, b: a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]
    p = cross(d, e2); det = dot(e1, p); sv = e[None] - v0; q = cross(sv, e1)
    tt = dot(e2, q) / det
    assert np.array_equal(tt, t[yy, xx])
    us = np.where(det < 0, -dot(sv, p), dot(sv, p)); vs = np.where(det < 0, -dot(d, q), dot(d, q))
    assert np.all((us >= 0) & (vs >= 0) & (us + vs <= np.abs(det)))


@pytest.mark.parametrize("light", ["sphere", "face"])
def test_ray_color_per_path(light):
    """The getRayColor hook: one radiance value per ray, all three lobes and both light types."""
    desc, cam = scenes.three_sphere_scene(light)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 160, 120
    rays = _camera_rays(c, W, H)
    prm = render_params(W, H, 1, seed=77)
    c1, k1 = so.ray_color(rays, prm)
    c2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2)                         # identical path trees: same rays traced
    r = _rel(c1, c2).max(axis=1)
    assert (r > PATH_RTOL).mean() <= 1e-4, r.max()
    assert np.sqrt(((c1 - c2) ** 2).mean()) <= RMSE_TOL
    assert c1.mean() > 0.05
    # non-default window and start depth
    c1, k1 = so.ray_color(rays[:2000], prm, t0=1e-9, t1=9.5, jumptime=5)
    c2, k2 = sg.ray_color(rays[:2000], prm, t0=1e-9, t1=9.5, jumptime=5)
    assert np.array_equal(k1, k2) and _rel(c1, c2).max() <= PATH_RTOL


@pytest.mark.parametrize("light", ["sphere", "face"])
def test_render_mesh_scene(light):
    """Bunny-class mesh in a triangle Cornell box with mirror and glass balls."""
    desc, cam = scenes.bunny_cornell_scene(nu=40, nv=41, light=light, extras=True)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    for i in range(4):
        assert so.octree_stats(i) == sg.octree_stats(i)
    c = hip.camera_init(**cam)
    W = H = 72
    a1, b1, _ = so.trace_primary(c, W, H); a2, b2, _ = sg.trace_primary(c, W, H)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    prm = render_params(W, H, 3, seed=9)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2)
    spp = 9
    assert _rel(i1, i2).max() <= PATH_RTOL
    assert np.sqrt((((i1 - i2) / spp) ** 2).mean()) <= RMSE_TOL
    assert (orc.resolve_bgr8(i1, spp) != orc.resolve_bgr8(i2, spp)).mean() < 1e-3   # a last-bit difference may flip a byte
    assert np.array_equal(hip.resolve_bgr8_host(i2, spp), orc.resolve_bgr8(i2, spp))   # resolve itself is bit-exact


def test_render_smallpt_spheres_face_forward():
    """The nine-sphere box seen from inside its wall spheres (needs the face-forward switch)."""
    desc, cam = scenes.smallpt_scene()
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 64, 48
    prm = render_params(W, H, 2, seed=4, face_forward=True, max_jump=16)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2)
    assert _rel(i1, i2).max() <= PATH_RTOL and np.sqrt((((i1 - i2) / 4) ** 2).mean()) <= RMSE_TOL
    off, _ = sg.render(c, render_params(W, H, 2, seed=4, face_forward=False, max_jump=16))
    assert off.mean() < 0.5 * i2.mean()                   # raw normals: the walls shade black


def test_multi_lobe_material_and_blurred_mirror():
    """Type flags are OR-able and every lobe adds its term (scene.cpp:213,249,267)."""
    mats = [scenes.material(MAT_NORMAL | MAT_SPECULAR, emission=(.1, .1, .1), color=(.6, .5, .4), blur=0.2),
            scenes.material(MAT_NORMAL | MAT_TRANSPARENT, color=(.7, .7, .7), atten=(.1, .1, .1), nr=1.3),
            scenes.material(MAT_NORMAL, color=(.5, .5, .5)), scenes.material(MAT_NORMAL, emission=(9, 9, 9), color=(0, 0, 0))]
    objs = [scenes.sphere((0, 1, 0), 1.0, 0), scenes.sphere((2.1, 1, 0), 1.0, 1), scenes.plane(2, 0.0, 2),
            scenes.sphere((0, 5, 1), 0.7, 3)]
    lights = [scenes.sphere_light((0, 5, 1), 0.7, (9, 9, 9)), scenes.face_light((3, 4, 2), (-1, -1, -1), (0, 1, 0), 1.0, 1.0, (7, 6, 5))]
    desc = SceneDesc(scenes._stack(objs, OBJECT_DT), scenes._stack(mats, MATERIAL_DT), scenes._stack(lights, LIGHT_DT), [],
                     background=(.02, .02, .03))
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init((1, 2, 7), (1, 1, 0), (0, 1, 0), 40, 1.0)
    prm = render_params(48, 48, 2, seed=8, max_jump=14)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2) and _rel(i1, i2).max() <= PATH_RTOL


def test_sample_slices_accumulate_and_match_whole_frame():
    """What the multi-GPU sample partition relies on: slices rendered into one buffer add up."""
    desc, cam = scenes.bunny_cornell_scene(nu=24, nv=25, light="face")
    sg = hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W = H = 48
    whole, k0 = sg.render(c, render_params(W, H, 4, seed=3))
    buf = np.zeros((H, W, 3))
    _, k1 = sg.render(c, render_params(W, H, 4, sample_begin=0, sample_end=5, seed=3), buf)
    _, k2 = sg.render(c, render_params(W, H, 4, sample_begin=5, sample_end=16, seed=3), buf)
    assert np.allclose(buf, whole, rtol=1e-12, atol=1e-14)
    assert np.array_equal(k0, k1 + k2)
    again, _ = sg.render(c, render_params(W, H, 4, seed=3))
    assert np.allclose(again, whole, rtol=1e-12, atol=1e-14)     # only the atomic add order may differ


def test_edge_cases():
    # empty scene: every ray returns the clear colour, nothing is hit
    desc = SceneDesc(np.zeros(0, OBJECT_DT), np.zeros(0, MATERIAL_DT), np.zeros(0, LIGHT_DT), [], background=(.25, .5, .75))
    sg = hip.HipScene(desc)
    c = hip.camera_init((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0)
    img, cnt = sg.render(c, render_params(9, 5, 2, seed=1))
    assert np.array_equal(img, np.tile(np.array([.25, .5, .75]) * 4, (5, 9, 1))) and int(cnt[0]) == 9 * 5 * 4
    prim, t, _ = sg.trace_primary(c, 9, 5)
    assert np.all(prim == MISS)
    # empty batches and empty sample slices
    out, cnt = sg.ray_color(np.zeros(0, RAY_DT), render_params(1, 1, 1))
    assert out.shape == (0, 3)
    img, cnt = sg.render(c, render_params(9, 5, 2, sample_begin=3, sample_end=3))
    assert not img.any() and int(cnt[0]) == 0
    # depth limits: max_jump 0 -> clear colour without tracing; rus_jump 0 -> roulette from the first hit
    desc, cam = scenes.three_sphere_scene("sphere")
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    for kw in (dict(max_jump=0), dict(max_jump=1), dict(max_jump=7, rus_jump=0), dict(max_jump=60, rus_jump=50)):
        prm = render_params(32, 24, 2, seed=2, **kw)
        i1, k1 = so.render(c, prm); i2, k2 = sg.render(c, prm)
        assert np.array_equal(k1, k2) and _rel(i1, i2).max() <= PATH_RTOL, kw
    # bad parameters are errors
    with pytest.raises(hip.RtError):
        sg.render(c, render_params(32, 24, 2, sample_begin=3, sample_end=99))
    with pytest.raises(hip.RtError):
        sg.render(c, render_params(32, 24, 2, max_jump=100))


def test_camera_inside_mesh_and_axis_aligned_rays():
    """Rays that start inside the octree's root box and rays with zero direction components."""
    white, red, green = scenes.cornell_box_meshes(100.0, tess=16)
    desc = _mesh_desc(np.concatenate([white, red, green]), octree_max_size=16, octree_max_depth=15)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    rng = np.random.default_rng(3)
    rays = np.zeros(30000, RAY_DT)
    rays["e"] = rng.uniform(1, 99, (30000, 3))
    d = rng.normal(0, 1, (30000, 3))
    d[:3000, 0] = 0; d[3000:6000, 1] = 0; d[6000:8000, (0, 2)] = 0
    d[8000:9000] = np.sign(d[8000:9000]) * np.array([1, 0, 0])          # exactly along an axis
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["e"][9000:9500] = np.round(rays["e"][9000:9500] / 6.25) * 6.25   # origins on grid planes
    p1, t1, n1, c1 = so.trace_rays(rays, 1e-9, 1e10)
    p2, t2, n2, c2 = sg.trace_rays(rays, 1e-9, 1e10)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)
    assert (p1 != MISS).mean() > 0.6


def test_two_large_meshes_between_simple_objects():
    """Object order sphere, large mesh, one-leaf mesh, plane, large mesh, sphere: every kind of
    hand-over between the object sweep and the octree steps, for closest hits and for path tracing."""
    a = scenes.blob((30, 30, 40), 18.0, 30, 31, seed=2)
    b = scenes.blob((70, 35, 55), 20.0, 28, 29, seed=3)
    small = scenes.quad((10, 0.5, 10), (90, 0.5, 10), (90, 0.5, 90), (10, 0.5, 90))
    mats = [scenes.material(MAT_NORMAL, color=(.7, .6, .5)), scenes.material(MAT_SPECULAR, color=(.9, .9, .9)),
            scenes.material(MAT_TRANSPARENT, color=(.95, .95, .95), atten=(.01, .02, .03), nr=1.4),
            scenes.material(MAT_NORMAL, emission=(20, 20, 20), color=(0, 0, 0))]
    objs = [scenes.sphere((50, 12, 30), 10.0, 1), scenes.mesh_object(0, 0), scenes.mesh_object(2, 0), scenes.plane(OBJ_GROUND, 0.0, 0),
            scenes.mesh_object(1, 2), scenes.sphere((50, 90, 50), 8.0, 3)]
    lights = [scenes.sphere_light((50, 90, 50), 8.0, (20, 20, 20))]
    desc = SceneDesc(scenes._stack(objs, OBJECT_DT), scenes._stack(mats, MATERIAL_DT), scenes._stack(lights, LIGHT_DT),
                     [a, b, small], background=(.1, .1, .12))
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    assert so.octree_stats(0) == sg.octree_stats(0) and sg.octree_stats(0)["nodes"] > 1 and sg.octree_stats(2)["nodes"] == 1
    rays = _rays(np.random.default_rng(5), 40000, 0.0, 100.0)
    d = np.random.default_rng(6).uniform(5, 95, (len(rays), 3)) - rays["e"]      # aim at the scene, keep the exact zeros
    rays["d"] = np.where(rays["d"] == 0.0, 0.0, d / np.linalg.norm(d, axis=1, keepdims=True))
    p1, t1, n1, c1 = so.trace_rays(rays, 1e-9, 1e10)
    p2, t2, n2, c2 = sg.trace_rays(rays, 1e-9, 1e10)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)
    assert (p1 != MISS).mean() > 0.5
    c = hip.camera_init((50, 45, 230), (50, 35, 50), (0, 1, 0), 35, 1.0)
    prm = render_params(64, 64, 2, seed=12, max_jump=20)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2) and _rel(i1, i2).max() <= PATH_RTOL


@pytest.mark.parametrize("scale", [1e-30, 1e60, 1e101])
def test_scaled_scene_bit_exact(scale):
    """Coordinates far from 1: up to 1e100 the slab tests use the reciprocal form, beyond it true
    divisions; both must reproduce the oracle bit for bit."""
    tris = scenes.soup(3000, seed=41, edge_lo=2e-2, edge_hi=2e-1) * scale
    desc = _mesh_desc(tris, octree_max_size=16, octree_max_depth=15, tri_box_pad=1e-7 * scale)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    assert so.octree_stats(0) == sg.octree_stats(0)
    rays = _rays(np.random.default_rng(42), 20000, -0.5, 1.5)
    rays["e"] *= scale
    p1, t1, n1, c1 = so.trace_rays(rays, 0.0, 1e10 * scale)
    p2, t2, n2, c2 = sg.trace_rays(rays, 0.0, 1e10 * scale)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)
    assert (p1 != MISS).mean() > 0.2


@pytest.mark.parametrize("light", ["point", "face"])
def test_phong_whitted_branch(light):
    """Materials without the MONTECARLO flag: ambient / diffuse / specular per light with shadow
    rays, Whitted mirror (blurred) and glass, point Lights and a FaceLight (scene.cpp:305-375)."""
    desc, cam = scenes.phong_scene(light)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 96, 72
    rays = _camera_rays(c, W, H)
    prm = render_params(W, H, 1, seed=21, max_jump=10)
    c1, k1 = so.ray_color(rays, prm)
    c2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2)                         # identical ray trees
    r = _rel(c1, c2).max(axis=1)
    assert (r > PATH_RTOL).mean() <= 1e-4, r.max()
    assert c1.mean() > 0.1
    prm = render_params(W, H, 2, seed=22, max_jump=8)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2) and np.sqrt((((i1 - i2) / 4) ** 2).mean()) <= RMSE_TOL
    assert (_rel(i1, i2) > PATH_RTOL).mean() <= 1e-4


def test_profile_counters_are_consistent():
    desc, cam = scenes.bunny_cornell_scene(nu=24, nv=25)
    sg = hip.HipScene(desc)
    sg.set_profiling(True)
    c = hip.camera_init(**cam)
    _, cnt = sg.render(c, render_params(64, 64, 2, seed=1))
    p = sg.get_profile()
    # every round traces and shades one chunk of bounce rays; shadow-ray launches come on top
    assert p["shade_launches"] >= 1 and p["trace_ms"] > 0 and p["trace_launches"] >= p["shade_launches"]
    assert int(cnt[0]) > 64 * 64 * 4
    # the per-stage counters add up to the call's own four, launch for launch and ray for ray; the
    # engine stages made box tests and fewer fp64 tests than the primitives their walks covered
    ks = sg.get_kernel_stats()
    for j, name in enumerate(("rays", "inner", "leaf", "prim")):
        assert sum(int(v[name]) for v in ks.values()) == int(cnt[j])
    assert sum(v["launches"] for v in ks.values()) == p["trace_launches"]
    assert abs(sum(v["ms"] for v in ks.values()) - p["trace_ms"]) < 1e-6 * max(1.0, p["trace_ms"])
    eng = ks["k_trace"]
    assert eng["box_tests"] > 0 and 0 < eng["fp64_tests"] < eng["prim"]
    sw = ks["k_sweep+k_trace_retry"]
    assert sw["rays"] > 0 and sw["box_tests"] == 0 and sw["fp64_tests"] == sw["prim"]    # the walls' lists are walked whole
    assert all(v["launches"] == 0 and v["rays"] == 0 for v in sg.get_kernel_stats().values())   # reading reset them


def test_cpp_scene_api_example_matches_oracle(tmp_path):
    """examples/main_kaleidoscope (the reference driver's scene written against
    include/rt_scene.hpp, run as its own process) against the oracle's render of that scene."""
    import subprocess
    exe = _example("main_kaleidoscope")
    W, H, n = 102, 63, 3
    out = str(tmp_path / "k.bmp")
    subprocess.check_call([exe, str(W), str(H), str(n), out], timeout=120)
    raw = open(out, "rb").read()
    got = _bmp_pixels(raw, W, H)
    desc, cam = scenes.kaleidoscope_scene(W, H)
    so = orc.OracleScene(desc)
    img, _ = so.render(orc.camera_init(**cam), render_params(W, H, n, seed=1243))
    exp = orc.resolve_bgr8(img, n * n)
    assert (got != exp).mean() < 2e-3           # a last-bit radiance difference may flip a byte
    assert exp.max() > 20


def test_depth_camera_lens_render():
    """DepthCamera (camera.cpp:50-72): the ray origin is drawn on the lens disk per sample."""
    desc, cam = scenes.three_sphere_scene("face")
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    co, cg = orc.camera_init(**cam), hip.camera_init(**cam)
    co.aperture = cg.aperture = 0.35
    W, H, n = 64, 48, 4
    prm = render_params(W, H, n, seed=5)
    i1, k1 = so.render(co, prm)
    i2, k2 = sg.render(cg, prm)
    assert np.array_equal(k1, k2)
    assert np.sqrt((((i1 - i2) / (n * n)) ** 2).mean()) <= RMSE_TOL
    assert (_rel(i1, i2) > PATH_RTOL).mean() <= 1e-3
    pin, _ = sg.render(hip.camera_init(**cam), prm)
    assert not np.array_equal(pin, i2)


@pytest.mark.parametrize("which", ["phong", "montecarlo"])
def test_animation_frames_moving_light_and_changing_fov(which):
    """The driver's animation loop (main.cpp:83-101): per frame the MovableFaceLight is patched
    into the uploaded scene (rt_scene_set_light) and the ChangeFovCamera rescales its film; every
    frame must equal the oracle's render of a scene built afresh for that time."""
    from raytracer_amd.parallel import frame_times
    if which == "phong":
        desc, cam, li = scenes.animation_scene(64, 48)
    else:
        desc, cam = scenes.three_sphere_scene("face")
        li = 0
    sg = hip.HipScene(desc)
    c0 = hip.camera_init(**cam)
    l0 = desc.lights[li].copy()
    W, H, n = 64, 48, 2
    prm = render_params(W, H, n, seed=9, max_jump=10)
    frames = []
    for f, time in enumerate(frame_times(40)):
        if f % 13:
            continue
        lt = hip.light_move_face(l0, time)
        sg.set_light(li, lt)
        i2, k2 = sg.render(hip.camera_change_fov(c0, time), prm)
        desc.lights[li] = lt
        i1, k1 = orc.OracleScene(desc).render(orc.camera_change_fov(orc.camera_init(**cam), time), prm)
        assert np.array_equal(k1, k2), f
        assert np.sqrt((((i1 - i2) / (n * n)) ** 2).mean()) <= RMSE_TOL
        assert (_rel(i1, i2) > PATH_RTOL).mean() <= 1e-3
        frames.append(i2)
    assert not np.array_equal(frames[0], frames[1])
    with pytest.raises(RuntimeError):
        sg.set_light(7, l0)


def test_cpp_animation_example_matches_oracle(tmp_path):
    """examples/main_animation (ChangeFovCamera + MovableFaceLight through Scene::setTime of
    include/rt_scene.hpp, its own process) against the oracle, frame by frame."""
    import subprocess
    from raytracer_amd.parallel import frame_times
    exe = _example("main_animation")
    W, H, n, nframes = 80, 60, 2, 4
    subprocess.check_call([exe, str(W), str(H), str(n), str(nframes), str(tmp_path / "f_")], timeout=120)
    desc, cam, li = scenes.animation_scene(W, H)
    l0 = desc.lights[li].copy()
    c0 = orc.camera_init(**cam)
    prev = None
    for f, time in enumerate(frame_times(nframes)):
        raw = open(tmp_path / f"f_{f:03d}.bmp", "rb").read()
        got = _bmp_pixels(raw, W, H)
        desc.lights[li] = orc.light_move_face(l0, time)
        img, _ = orc.OracleScene(desc).render(orc.camera_change_fov(c0, time), render_params(W, H, n, seed=1243))
        exp = orc.resolve_bgr8(img, n * n)
        assert (got != exp).mean() < 2e-3, f
        assert prev is None or not np.array_equal(prev, got)
        prev = got
    # the same frames went into the Motion-JPEG AVI through AviSaver: its last frame decodes to the last BMP
    import io
    import struct
    from PIL import Image
    raw = open(tmp_path / "f_anim.avi", "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and raw.count(b"vidsMJPG") == 1
    movi, idx = raw.index(b"movi"), raw.rindex(b"idx1")
    assert struct.unpack_from("<I", raw, idx + 4)[0] == 16 * nframes
    _, _, off, sz = struct.unpack_from("<4sIII", raw, idx + 8 + 16 * (nframes - 1))
    dec = np.asarray(Image.open(io.BytesIO(raw[movi + off + 8:movi + off + 8 + sz])).convert("RGB")).astype(float)
    src = got[::-1, :, ::-1].astype(float)
    assert 10 * np.log10(255.0 ** 2 / ((dec - src) ** 2).mean()) > 32


def _implicit_desc(kind, center=(0, 0, 0), scale=1.0, min_size=0.0):
    from raytracer_amd.desc import LIGHT_DT
    return SceneDesc(scenes._stack([scenes.implicit_object(kind, center, scale, 0)], OBJECT_DT),
                     scenes._stack([scenes.phong_material((.1, .1, .1), (.6, .3, .3), (.4, .4, .4), 16.0)], MATERIAL_DT),
                     np.zeros(0, LIGHT_DT), [], implicit_min_size=min_size, octree_max_depth=15)


@pytest.mark.parametrize("kind", [6, 7])
def test_implicit_surfaces_match_reference_golden(golden_dir, kind):
    """The reference's implicit surfaces (HeartShape, SurfaceOfGenus2): voxel set and voxel octree
    built on the host (ImplicitSurfaceOctree::build, octree.cpp:148-254), BoxNode hits on the GPU
    (Surface.cpp:43-59), the drawables' frame change and analytic normals -- against vectors
    captured from the unmodified reference, and against the oracle."""
    g = _load(golden_dir, "implicit")
    k = f"k{kind}_"
    desc = _implicit_desc(kind, min_size=float(g[k + "min_size"]))
    sg, so = hip.HipScene(desc), orc.OracleScene(desc)
    st = sg.implicit_stats(kind)
    assert [st[x] for x in ("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost")] == list(g[k + "stats"])
    boxes = sg.implicit_boxes(kind)
    assert len(boxes) == int(g[k + "nbox"]) and np.array_equal(boxes, so.implicit_boxes(kind))
    rays = np.zeros(len(g[k + "e"]), RAY_DT); rays["e"] = g[k + "e"]; rays["d"] = g[k + "d"]
    for (t0, t1), hk, tk in (((0.0, 1e10), "hit", "t"), ((1e-9, 3.6), "hit_clip", "t_clip")):
        prim, t, _, cnt = sg.trace_rays(rays, t0, t1)
        assert np.array_equal(prim != MISS, g[k + hk] != 0) and np.array_equal(t, g[k + tk])
        p1, t1_, _, c1 = so.trace_rays(rays, t0, t1)
        assert np.array_equal(prim, p1) and np.array_equal(cnt, c1)      # same voxel, same work
    # the drawable at the reference's default voxel size, moved and scaled
    desc = _implicit_desc(kind, tuple(g[k + "center"]), float(g[k + "scale"]))
    sg = hip.HipScene(desc)
    rays = np.zeros(len(g[k + "obj_e"]), RAY_DT); rays["e"] = g[k + "obj_e"]; rays["d"] = g[k + "obj_d"]
    prim, t, nrm, _ = sg.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(prim != MISS, g[k + "obj_hit"] != 0) and np.array_equal(t, g[k + "obj_t"])
    assert np.array_equal(nrm, g[k + "obj_normal"])
    if kind == 6:
        st = sg.implicit_stats(6)
        assert (st["nodes"], st["leaves"], st["refs"]) == (30745, 26902, 743759)


@pytest.mark.parametrize("variant", ["montecarlo", "check"])
def test_implicit_scene_render(variant):
    """Two hearts sharing one voxel tree and a genus-2 surface among planes: closest hits
    bit-exact (voxel ids, t, counters), per-path radiance and the rendered frame within tolerance,
    for Monte-Carlo materials and for the NormalCheck / PositionCheck materials."""
    desc, cam = scenes.implicit_scene(variant)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 96, 72
    rays = _camera_rays(c, W, H)
    p1, t1, n1, c1 = so.trace_rays(rays, 0.0, 1e10)
    p2, t2, n2, c2 = sg.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(c1, c2)
    assert np.array_equal(n1, n2)
    nvox = len(sg.implicit_boxes(6))
    assert ((p1 < nvox).sum() > 200) and ((p1 >= nvox) & (p1 < 2 * nvox)).sum() > 100      # both hearts are seen
    prm = render_params(W, H, 1, seed=31, max_jump=10)
    a1, k1 = so.ray_color(rays, prm)
    a2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2)
    r = _rel(a1, a2).max(axis=1)
    assert (r > PATH_RTOL).mean() <= 1e-3, r.max()
    prm = render_params(W, H, 2, seed=32, max_jump=8)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2) and np.sqrt((((i1 - i2) / 4) ** 2).mean()) <= RMSE_TOL
    assert i1.mean() > 0.05


def test_cpp_heart_example_matches_oracle(tmp_path):
    """examples/main_heart (HeartShape, SurfaceOfGenus2, NormalCheckMaterial through
    include/rt_scene.hpp, its own process) against the oracle's render of that scene."""
    import subprocess
    exe = _example("main_heart")
    W, H, n, ms = 96, 72, 2, 2e-2
    out = str(tmp_path / "h.bmp")
    subprocess.check_call([exe, str(W), str(H), str(n), str(ms), out], timeout=120)
    got = np.frombuffer(open(out, "rb").read()[54:], np.uint8).reshape(H, W, 3)
    desc, cam = scenes.heart_example_scene(W, H, ms)
    img, _ = orc.OracleScene(desc).render(orc.camera_init(**cam), render_params(W, H, n, seed=1243))
    exp = orc.resolve_bgr8(img, n * n)
    assert (got != exp).mean() < 2e-3
    assert exp.std() > 20


@pytest.mark.parametrize("tables", ["default", "golden"])
def test_perlin_materials(golden_dir, tables):
    """PerlinNoiseNormalMaterial and PerlinNoiseNormalMaterial_Blood (material.h:95-164): colours
    from the noise at the hit, tilted normals (trace_rays normals bit-exact), per-path radiance
    and the frame within tolerance; with the library's own tables and with the tables the
    reference's generator produced for the golden vectors."""
    tb = None
    if tables == "golden":
        g = _load(golden_dir, "perlin")
        tb = (g["P"], g["G"])
    desc, cam = scenes.perlin_scene(tb)
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H = 96, 72
    rays = _camera_rays(c, W, H)
    p1, t1, n1, c1 = so.trace_rays(rays, 0.0, 1e10)
    p2, t2, n2, c2 = sg.trace_rays(rays, 0.0, 1e10)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(c1, c2)
    assert np.array_equal(n1, n2)                     # the tilted normals too: division, floor, fabs and sqrt are exact on both sides
    prm = render_params(W, H, 1, seed=41, max_jump=10)
    a1, k1 = so.ray_color(rays, prm)
    a2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2)
    r = _rel(a1, a2).max(axis=1)
    assert (r > PATH_RTOL).mean() <= 1e-3, r.max()
    prm = render_params(W, H, 2, seed=42, max_jump=8)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2) and np.sqrt((((i1 - i2) / 4) ** 2).mean()) <= RMSE_TOL
    assert i1.std() > 0.1


def test_resolve_with_tone_operators():
    """rt_resolve_tone_bgr8: SigmoidMethod / SqrtMethod on the averaged colour, then clamp and
    int(c*255) -- bytes equal to the oracle's except where the device's exp / pow differ from
    glibc's in the last bit right at a byte boundary."""
    rng = np.random.default_rng(8)
    img = rng.uniform(-0.2, 5.0, (64, 96, 3)) * 16
    for tone in (hip.TONE_NONE, hip.TONE_SIGMOID, hip.TONE_SQRT):
        a = hip.resolve_bgr8_host(img, 16, tone)
        b = orc.resolve_bgr8(img, 16, tone)
        assert (a != b).mean() < 1e-3 and np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert not np.array_equal(hip.resolve_bgr8_host(img, 16, hip.TONE_SIGMOID), hip.resolve_bgr8_host(img, 16))


def test_benchmark_frame_full_size_properties():
    """The benchmark frame itself (69 574-triangle mesh in the triangle Cornell box, 1024 x 1024
    at 256 spp): whole image rows against the oracle at full sample count, additivity over
    sample slices (what the multi-GPU reduce relies on), run-to-run agreement, and the ray
    count as the sum of the slices' counts."""
    desc, cam = scenes.bunny_cornell_scene()
    sg, so = hip.HipScene(desc), orc.OracleScene(desc)
    c = hip.camera_init(**cam)
    W = H = 1024
    n = 16
    prm = render_params(W, H, n, seed=1243)
    whole, k = sg.render(c, prm)
    assert int(k[0]) > W * H * n * n * 5                     # several rays per path
    # three whole rows at the full 256 spp against the CPU oracle
    for y in (97, 512, 901):
        row, ko = so.render(orc.camera_init(**cam), prm, threads=16, pix_range=(W * y, W * y + W))
        got, exp = whole[y], row[y]
        assert np.sqrt((((got - exp) / (n * n)) ** 2).mean()) <= RMSE_TOL
        assert (_rel(got, exp) > PATH_RTOL).mean() <= 1e-3
        assert exp.mean() > 1.0
    # two sample slices (ranks of a 2-GPU job) add up to the whole frame, rays included
    acc = np.zeros_like(whole)
    rays = 0
    for b, e in ((0, 128), (128, 256)):
        _, kk = sg.render(c, render_params(W, H, n, sample_begin=b, sample_end=e, seed=1243), acc)
        rays += int(kk[0])
    assert rays == int(k[0])
    assert np.abs(acc - whole).max() <= 1e-9 * max(1.0, np.abs(whole).max())
    # a second run differs by the order of the atomic adds only
    again, k2 = sg.render(c, prm)
    assert np.array_equal(k, k2) and np.abs(again - whole).max() <= 1e-9 * max(1.0, np.abs(whole).max())


def test_atrium_scene_ids_and_radiance():
    """The Sponza-class atrium (three octrees with inner nodes, rectangular light): tree shape and
    primary ids / distances bit-exact at full detail, a small frame against the oracle."""
    desc, cam = scenes.atrium_scene()
    so, sg = orc.OracleScene(desc), hip.HipScene(desc)
    for i in range(len(desc.meshes)):
        assert so.octree_stats(i) == sg.octree_stats(i)
    c = hip.camera_init(**cam)
    W, H = 160, 90
    a1, b1, _ = so.trace_primary(c, W, H); a2, b2, _ = sg.trace_primary(c, W, H)
    assert (a1 != MISS).mean() > 0.99
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    W, H, n = 64, 36, 2
    prm = render_params(W, H, n, seed=1243)
    i1, k1 = so.render(c, prm)
    i2, k2 = sg.render(c, prm)
    assert np.array_equal(k1, k2)
    assert (_rel(i1, i2) > PATH_RTOL).mean() <= 1e-3
    assert np.sqrt((((i1 - i2) / (n * n)) ** 2).mean()) <= RMSE_TOL
    assert i1.mean() > 0.0


def test_perlin_tables_drift_with_time():
    """rt_scene_set_perlin: the generator's vectors after some frames of the driver's clock are
    patched into the uploaded scene; the frame equals the oracle's for a scene built with them."""
    P, G = hip.perlin_tables()
    desc, cam = scenes.perlin_scene((P, G.copy()))
    sg = hip.HipScene(desc)
    c = hip.camera_init(**cam)
    W, H, n = 64, 48, 2
    prm = render_params(W, H, n, seed=7, max_jump=8)
    first, _ = sg.render(c, prm)
    clk = hip.PerlinClockC()
    prev = t = 0.0
    for _ in range(9):
        hip.perlin_time_changed(clk, prev, t, G)
        prev, t = t, t + 0.015
    sg.set_perlin(P, G)
    i2, k2 = sg.render(c, prm)
    desc2, _ = scenes.perlin_scene((P, G))
    i1, k1 = orc.OracleScene(desc2).render(orc.camera_init(**cam), prm)
    assert np.array_equal(k1, k2) and np.sqrt((((i1 - i2) / (n * n)) ** 2).mean()) <= RMSE_TOL
    assert np.abs(first - i2).max() > 1e-3
    with pytest.raises(hip.RtError):
        sg.set_perlin(P[:64], G[:64])


def test_rays_handed_on_from_the_sweep_kernel(monkeypatch):
    """A ray's class (does it reach an octree's root box?) only picks the kernel: class 0 goes to the
    object-sweep kernel, which hands a ray that reaches the box after all on to the engine.  The
    guess is built to err towards class 1, so the hand-on never happens by itself; with the guess
    switched off (RT_CLASS_GUESS=0: every ray is class 0) every ray that enters the mesh takes that
    way, and paths, frames and work counters must still be the oracle's."""
    desc, cam = scenes.bunny_cornell_scene(nu=40, nv=41, extras=True)
    so = orc.OracleScene(desc)
    monkeypatch.setenv("RT_CLASS_GUESS", "0")
    sg = hip.HipScene(desc)
    monkeypatch.delenv("RT_CLASS_GUESS")
    c = hip.camera_init(**cam)
    W, H = 96, 64
    rays = _camera_rays(c, W, H)
    prm = render_params(W, H, 1, seed=9)
    c1, k1 = so.ray_color(rays, prm)
    c2, k2 = sg.ray_color(rays, prm)
    assert np.array_equal(k1, k2)
    r = _rel(c1, c2).max(axis=1)
    assert (r > PATH_RTOL).mean() <= 1e-4, r.max()
    prm = render_params(W, H, 3, seed=9)
    i1, q1 = so.render(orc.camera_init(**cam), prm)
    i2, q2 = sg.render(c, prm)
    assert np.array_equal(q1, q2)
    assert np.sqrt((((i1 - i2) / 9) ** 2).mean()) <= RMSE_TOL
