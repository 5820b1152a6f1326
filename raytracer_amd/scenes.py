"""Deterministic synthetic scenes for the benchmark configurations and the parity tests.

No mesh asset ships with the reference (it has no triangle primitive at all) and none can
be downloaded, so "bunny-class" / "Sponza-class" geometry is generated procedurally with
the triangle counts the benchmark configurations name.  Everything is seeded; the same call
gives the same arrays on every machine.
"""
import math

import numpy as np

from .desc import (LIGHT_DT, LIGHT_FACE, LIGHT_POINT, LIGHT_SPHERE, MAT_MONTECARLO, MAT_NORMAL, MAT_SPECULAR,
                   MAT_TRANSPARENT, MATERIAL_DT, OBJ_GENUS2, OBJ_GROUND, OBJ_HEART, OBJ_MESH, OBJ_SPHERE,
                   OBJ_WALL_X, OBJ_WALL_Z, OBJECT_DT, CHECK_NORMAL, CHECK_POSITION, SceneDesc)


# ----------------------------------------------------------------------------- record helpers
def material(kind, emission=(0, 0, 0), color=(0, 0, 0), blur=0.0, atten=(0, 0, 0), nr=1.0):
    """Material(emission, color, type) -- material.cpp:14-17 (MONTECARLO is OR-ed in)."""
    m = np.zeros((), MATERIAL_DT)
    m["type"] = int(kind) | MAT_MONTECARLO
    m["emission"], m["color"], m["blur"], m["atten"], m["nr"] = emission, color, blur, atten, nr
    return m


def phong_material(ambient=(0, 0, 0), diffuse=(0, 0, 0), specular=(0, 0, 0), shine=1.0, mirror=None, blur=0.0):
    """Material(amb, dif, spe, shi) / Material(amb, dif, spe, mir, shi) -- material.cpp:2-12: the
    materials of the Phong / Whitted branch (no MONTECARLO flag)."""
    m = np.zeros((), MATERIAL_DT)
    m["type"] = MAT_NORMAL if mirror is None else (MAT_NORMAL | MAT_SPECULAR)
    m["ambient"], m["color"], m["specular"], m["shine"], m["blur"], m["nr"] = ambient, diffuse, specular, shine, blur, 1.0
    if mirror is not None:
        m["mirror"] = mirror
    return m


def glass_material(atten, nr):
    """Material(a, nr) -- material.cpp:19-23: TRANSPARENT, shaded by the Whitted branch."""
    m = np.zeros((), MATERIAL_DT)
    m["type"], m["atten"], m["nr"] = MAT_TRANSPARENT, atten, nr
    return m


def point_light(pos, ambient, diffuse, specular):
    """Light(pos, ambient, diffuse, specular) -- light.cpp:8-11."""
    l = np.zeros((), LIGHT_DT)
    l["type"], l["pos"], l["ambient"], l["diffuse"], l["specular"] = LIGHT_POINT, pos, ambient, diffuse, specular
    return l


def sphere(center, radius, mat):
    o = np.zeros((), OBJECT_DT)
    o["type"], o["material"], o["mesh"] = OBJ_SPHERE, mat, -1
    o["p"] = (center[0], center[1], center[2], radius)
    return o


def plane(kind, coord, mat):
    o = np.zeros((), OBJECT_DT)
    o["type"], o["material"], o["mesh"] = kind, mat, -1
    o["p"] = (coord, 0, 0, 0)
    return o


def mesh_object(mesh_index, mat):
    o = np.zeros((), OBJECT_DT)
    o["type"], o["material"], o["mesh"] = OBJ_MESH, mat, mesh_index
    return o


def sphere_light(pos, radius, emission):
    """SphereLight(pos, radius, emission) -- light.cpp:109-116."""
    l = np.zeros((), LIGHT_DT)
    l["type"], l["pos"], l["radius"], l["diffuse"] = LIGHT_SPHERE, pos, radius, emission
    return l


def face_light(pos, normal, up, width, height, diffuse, ambient=(0, 0, 0), specular=(0, 0, 0)):
    """FaceLight(pos, normal, up, width, height, ...) -- light.cpp:50-56: right = up ^ normal,
    normal normalised as v * (1/|v|)."""
    l = np.zeros((), LIGHT_DT)
    n = np.asarray(normal, np.float64)
    u = np.asarray(up, np.float64)
    right = np.array([u[1] * n[2] - n[1] * u[2], u[2] * n[0] - u[0] * n[2], u[0] * n[1] - n[0] * u[1]])
    inv = 1 / math.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
    l["type"], l["pos"], l["normal"], l["up"], l["right"] = LIGHT_FACE, pos, n * inv, u, right
    l["width"], l["height"], l["diffuse"], l["ambient"], l["specular"] = width, height, diffuse, ambient, specular
    return l


def _stack(recs, dt):
    out = np.zeros(len(recs), dt)
    for i, r in enumerate(recs):
        out[i] = r
    return out


# ----------------------------------------------------------------------------- mesh helpers
def quad(a, b, c, d, nu=1, nv=1):
    """Tessellated quad a-b-c-d (counter-clockwise seen from the side its normal points to)."""
    a, b, c, d = (np.asarray(x, np.float64) for x in (a, b, c, d))
    u = np.linspace(0, 1, nu + 1)[:, None, None]
    v = np.linspace(0, 1, nv + 1)[None, :, None]
    p = (1 - u) * (1 - v) * a + u * (1 - v) * b + u * v * c + (1 - u) * v * d
    return grid_tris(p)


def grid_tris(p):
    """Triangles of a (nu+1, nv+1, 3) vertex grid; normal = d/du x d/dv."""
    p00, p10, p11, p01 = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.concatenate([p00, p10, p11], axis=-1).reshape(-1, 9)
    t2 = np.concatenate([p00, p11, p01], axis=-1).reshape(-1, 9)
    return np.concatenate([t1, t2], axis=0)


def flip(tris):
    return np.ascontiguousarray(tris.reshape(-1, 3, 3)[:, [0, 2, 1], :].reshape(-1, 9))


def blob(center, radius, nu, nv, seed=1, bump=0.18):
    """Closed, outward-facing displaced sphere with 2*nu*nv triangles ("bunny-class")."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, math.pi, nv + 1)[None, :]        # polar
    ph = np.linspace(0, 2 * math.pi, nu + 1)[:, None]    # azimuth
    r = np.ones((nu + 1, nv + 1))
    for _ in range(6):
        a, b = rng.integers(1, 6, 2)
        r += bump / 3 * rng.uniform(0.3, 1) * np.sin(a * ph + rng.uniform(0, 6.28)) * np.sin(b * th + rng.uniform(0, 6.28)) * np.sin(th)
    r[-1] = r[0]
    x = r * np.sin(th) * np.cos(ph)
    y = r * np.cos(th)
    z = r * np.sin(th) * np.sin(ph)
    p = np.stack([x, y, z], axis=-1) * radius + np.asarray(center, np.float64)
    return grid_tris(p)      # d/dph x d/dth = sin(th) * radial direction: outward


def cylinder(base, radius, height, nseg, nh=1):
    """Open, outward-facing cylinder along +y."""
    ph = np.linspace(0, 2 * math.pi, nseg + 1)[:, None]
    hh = np.linspace(0, height, nh + 1)[None, :]
    x = radius * np.cos(ph) + 0 * hh
    z = radius * np.sin(ph) + 0 * hh
    y = hh + 0 * ph
    p = np.stack([x, y, z], axis=-1) + np.asarray(base, np.float64)
    return flip(grid_tris(p))      # d/dph x d/dy points at the axis -> flip to face outward


def box_tris(lo, hi):
    """Outward-facing axis-aligned box (12 triangles)."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    return np.concatenate([
        quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),   # +z
        quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),   # -z
        quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),   # +x
        quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),   # -x
        quad((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),   # +y
        quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),   # -y
    ])


def soup(n, seed=42, edge_lo=1e-3, edge_hi=1e-2):
    """n triangles: centres uniform in [0,1]^3, edge length log-uniform in [edge_lo, edge_hi]."""
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.random((n, 3))
    e = np.exp(rng.uniform(math.log(edge_lo), math.log(edge_hi), (n, 1)))
    v = [c + e * (rng.random((n, 3)) - 0.5) for _ in range(3)]
    return np.ascontiguousarray(np.concatenate(v, axis=1))


# ----------------------------------------------------------------------------- scenes
def soup_scene(n=1_000_000, seed=42, width=3840, height=2160):
    """Primary-ray id-buffer workload: a triangle soup seen from outside the unit cube."""
    mats = _stack([material(MAT_NORMAL, color=(.7, .7, .7))], MATERIAL_DT)
    objs = _stack([mesh_object(0, 0)], OBJECT_DT)
    desc = SceneDesc(objs, mats, np.zeros(0, LIGHT_DT), [soup(n, seed)])
    cam = dict(look_from=(1.9, 1.3, 2.4), look_at=(0.5, 0.5, 0.5), vup=(0, 1, 0), vfov=34.0, aspect=width / height)
    return desc, cam


def cornell_box_meshes(size=100.0, tess=1):
    """Five inward-facing walls of a box [0,size]^3, open towards +z. Returns (white, red, green)."""
    s = size
    floor = quad((0, 0, s), (s, 0, s), (s, 0, 0), (0, 0, 0), tess, tess)        # normal +y
    ceil_ = quad((0, s, 0), (s, s, 0), (s, s, s), (0, s, s), tess, tess)        # normal -y
    back = quad((0, 0, 0), (s, 0, 0), (s, s, 0), (0, s, 0), tess, tess)         # normal +z
    left = quad((0, 0, s), (0, 0, 0), (0, s, 0), (0, s, s), tess, tess)         # normal +x
    right = quad((s, 0, 0), (s, 0, s), (s, s, s), (s, s, 0), tess, tess)        # normal -x
    return np.concatenate([floor, ceil_, back]), left, right


def bunny_cornell_scene(nu=186, nv=187, light="sphere", extras=False, wall_tess=1):
    """Bunny-class closed mesh (2*nu*nv triangles; default 69 564) inside a Cornell box made of
    triangle meshes with one sphere (or rectangle) light; extras=True adds a mirror ball and a
    glass ball (used by the parity tests to cover the specular and dielectric lobes)."""
    white, red, green = cornell_box_meshes(100.0, wall_tess)
    body = blob((50, 32, 45), 24.0, nu, nv, seed=1)
    mats = [material(MAT_NORMAL, color=(.75, .75, .75)),        # 0 white
            material(MAT_NORMAL, color=(.75, .25, .25)),        # 1 red
            material(MAT_NORMAL, color=(.25, .25, .75)),        # 2 blue
            material(MAT_NORMAL, color=(.55, .70, .45)),        # 3 body
            material(MAT_SPECULAR, color=(.95, .95, .95)),      # 4 mirror
            material(MAT_TRANSPARENT, color=(.96, .96, .96), atten=(.004, .002, .006), nr=1.5),  # 5 glass
            material(MAT_NORMAL, emission=(36, 36, 36), color=(0, 0, 0))]   # 6 light geometry (light.cpp:113-115)
    objs = [mesh_object(0, 0), mesh_object(1, 1), mesh_object(2, 2), mesh_object(3, 3)]
    if extras:
        objs.append(sphere((22, 12, 70), 12.0, 4))
        objs.append(sphere((78, 11, 78), 11.0, 5))
    lights = []
    if light == "sphere":
        lights.append(sphere_light((50, 88, 50), 6.0, (36, 36, 36)))
        objs.append(sphere((50, 88, 50), 6.0, 6))               # Scene::addLight appends the light's geometry
    else:
        lights.append(face_light((50, 99.5, 50), (0, -1, 0), (0, 0, 1), 30.0, 30.0, (18, 18, 18)))
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT),
                     [white, red, green, body])
    cam = dict(look_from=(50, 50, 235), look_at=(50, 48, 50), vup=(0, 1, 0), vfov=40.0, aspect=1.0)
    return desc, cam


def flythrough_camera(u, aspect):
    """Camera at parameter u in [0, 1) of the closed fly-through curve around the Cornell box of
    bunny_cornell_scene (in front of the open side and into it, looking at the mesh): the keyword
    arguments of Camera(lookFrom, lookAt, vup, vfov, aspect)."""
    a = 2 * math.pi * u
    eye = (50 + 32 * math.cos(a), 50 + 12 * math.sin(2 * a), 185 + 45 * math.sin(a))
    return dict(look_from=eye, look_at=(50, 38, 48), vup=(0, 1, 0), vfov=40.0, aspect=aspect)


def smallpt_scene():
    """The nine smallpt spheres (smallpt.cpp:31-39 values) as reference Spheres; the camera sits
    inside the front wall sphere.  Needs face_forward=True: the wall spheres are seen from inside."""
    D, S, R = MAT_NORMAL, MAT_SPECULAR, MAT_TRANSPARENT
    spec = [  # radius, centre, emission, colour, kind
        (1e5, (1e5 + 1, 40.8, 81.6), (0, 0, 0), (.75, .25, .25), D),
        (1e5, (-1e5 + 99, 40.8, 81.6), (0, 0, 0), (.25, .25, .75), D),
        (1e5, (50, 40.8, 1e5), (0, 0, 0), (.75, .75, .75), D),
        (1e5, (50, 40.8, -1e5 + 170), (0, 0, 0), (0, 0, 0), D),
        (1e5, (50, 1e5, 81.6), (0, 0, 0), (.75, .75, .75), D),
        (1e5, (50, -1e5 + 81.6, 81.6), (0, 0, 0), (.75, .75, .75), D),
        (16.5, (27, 16.5, 47), (0, 0, 0), (.999, .999, .999), S),
        (16.5, (73, 16.5, 78), (0, 0, 0), (.999, .999, .999), R),
    ]
    mats, objs = [], []
    for i, (r, c, e, col, kind) in enumerate(spec):
        mats.append(material(kind, emission=e, color=col, nr=1.5))
        objs.append(sphere(c, r, i))
    mats.append(material(D, emission=(400, 400, 400), color=(0, 0, 0)))
    objs.append(sphere((50, 65.1, 81.6), 1.5, len(mats) - 1))
    lights = [sphere_light((50, 65.1, 81.6), 1.5, (400, 400, 400))]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [])
    cam = dict(look_from=(50, 52, 165), look_at=(50, 48, 80), vup=(0, 1, 0), vfov=60.0, aspect=512 / 384)
    return desc, cam


def atrium_scene(detail=1.0, seed=7):
    """Sponza-class procedural atrium (default ~262 k triangles): a long hall with big floor /
    wall / ceiling quads, two colonnades of finely tessellated columns, arches, hanging
    wavy drapes and a field of small clutter, lit by one rectangular light in the roof."""
    rng = np.random.default_rng(seed)
    L, W, H = 300.0, 120.0, 90.0
    stone, cloth, metal = [], [], []
    # shell: large triangles that overlap many octree children
    stone += [quad((0, 0, W), (L, 0, W), (L, 0, 0), (0, 0, 0), 2, 1),          # floor  (+y)
              quad((0, H, 0), (L, H, 0), (L, H, W), (0, H, W), 2, 1),          # ceiling(-y)
              quad((0, 0, 0), (L, 0, 0), (L, H, 0), (0, H, 0), 2, 1),          # back   (+z)
              quad((L, 0, W), (0, 0, W), (0, H, W), (L, H, W), 2, 1),          # front  (-z)
              quad((0, 0, W), (0, 0, 0), (0, H, 0), (0, H, W), 1, 1),          # left   (+x)
              quad((L, 0, 0), (L, 0, W), (L, H, W), (L, H, 0), 1, 1)]          # right  (-x)
    nseg = max(8, int(48 * detail)); nh = max(4, int(40 * detail))
    for i in range(10):
        x = 20 + i * 28.0
        for z in (28.0, W - 28.0):
            stone.append(cylinder((x, 0, z), 4.0, 58.0, nseg, nh))
            stone.append(box_tris((x - 6, 58, z - 6), (x + 6, 62, z + 6)))
            stone.append(box_tris((x - 5.5, 0, z - 5.5), (x + 5.5, 2.5, z + 5.5)))
    # arches between columns: half-tori approximated by swept grids
    na, nb = max(6, int(28 * detail)), max(4, int(12 * detail))
    for i in range(9):
        x0 = 20 + i * 28.0
        for z in (28.0, W - 28.0):
            a = np.linspace(0, math.pi, na + 1)[:, None]
            b = np.linspace(0, 2 * math.pi, nb + 1)[None, :]
            rr = 14.0 + 2.0 * np.cos(b)
            p = np.stack([x0 + 14.0 - rr * np.cos(a), 62 + rr * np.sin(a), z + 2.0 * np.sin(b) + 0 * a], axis=-1)
            stone.append(grid_tris(p))
    # drapes: wavy cloth sheets
    nd = max(8, int(64 * detail))
    for i in range(6):
        x0 = 30 + i * 44.0
        u = np.linspace(0, 1, nd + 1)[:, None]
        v = np.linspace(0, 1, nd + 1)[None, :]
        px = x0 + 30 * u + 0 * v
        py = 80 - 45 * v - 4 * np.sin(3.14159 * u) * v
        pz = W / 2 + 3.0 * np.sin(9 * u * 6.28 + i) * (0.3 + v) + 0 * u
        sheet = grid_tris(np.stack([px, py, pz], axis=-1))
        cloth.append(sheet)
        cloth.append(flip(sheet) + np.array([0, 0, 0.05] * 3))
    # clutter: small blobs on the floor
    nblob = max(2, int(24 * detail))
    for i in range(nblob):
        c = (rng.uniform(15, L - 15), 0, rng.uniform(40, W - 40))
        r = rng.uniform(2.0, 5.0)
        nbu = max(6, int(40 * detail))
        metal.append(blob((c[0], r * 0.9, c[2]), r, nbu, nbu, seed=100 + i))
    stone, cloth, metal = (np.concatenate(x) for x in (stone, cloth, metal))
    mats = [material(MAT_NORMAL, color=(.72, .68, .60)),      # stone
            material(MAT_NORMAL, color=(.70, .18, .15)),      # cloth
            material(MAT_SPECULAR, color=(.85, .80, .55), blur=0.05),   # brass-like blobs (blurred mirror)
            ]
    objs = [mesh_object(0, 0), mesh_object(1, 1), mesh_object(2, 2)]
    lights = [face_light((L / 2, H - 0.5, W / 2), (0, -1, 0), (0, 0, 1), 120.0, 40.0, (9, 9, 8))]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT),
                     [stone, cloth, metal], octree_max_depth=14)
    cam = dict(look_from=(12, 38, W / 2 + 6), look_at=(L - 40, 30, W / 2 - 4), vup=(0, 1, 0), vfov=62.0, aspect=1920 / 1080)
    return desc, cam


def three_sphere_scene(light="sphere"):
    """Small analytic scene (spheres + the three reference planes) used by unit parity tests."""
    mats = [material(MAT_NORMAL, color=(.8, .3, .3)), material(MAT_SPECULAR, color=(.9, .9, .9)),
            material(MAT_TRANSPARENT, color=(.95, .95, .95), atten=(.01, .02, .03), nr=1.5),
            material(MAT_NORMAL, color=(.6, .6, .6)),
            material(MAT_NORMAL, emission=(20, 20, 20), color=(0, 0, 0)),
            material(MAT_SPECULAR, color=(.8, .8, .8), blur=0.08)]
    objs = [sphere((0, 1, 0), 1.0, 0), sphere((2.2, 1, 0.3), 1.0, 1), sphere((-2.2, 1, 0.5), 1.0, 2),
            plane(OBJ_GROUND, 0.0, 3), plane(OBJ_WALL_Z, -4.0, 3), plane(OBJ_WALL_X, -6.0, 5)]
    if light == "sphere":
        lights = [sphere_light((0, 6, 2), 0.8, (20, 20, 20))]
        objs.append(sphere((0, 6, 2), 0.8, 4))
    else:
        lights = [face_light((0, 7, 1), (0, -1, 0), (0, 0, 1), 3.0, 2.0, (12, 12, 12))]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [],
                     background=(0.05, 0.07, 0.1))
    cam = dict(look_from=(0, 2.5, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=40.0, aspect=4 / 3)
    return desc, cam


def phong_scene(light="point"):
    """Scene for the Phong / Whitted branch of Scene::rayColor (scene.cpp:305-375): a Phong
    sphere, a Phong + mirror sphere (blurred), a glass sphere (Material(a, nr)), a small Phong
    mesh, Phong floor and walls, lit by two point Lights or by a point Light and a FaceLight."""
    mats = [phong_material((.10, .04, .04), (.70, .25, .25), (.5, .5, .5), 24.0),
            phong_material((.02, .02, .02), (.10, .10, .12), (.6, .6, .6), 60.0, mirror=(.75, .75, .8), blur=0.03),
            glass_material((.02, .03, .05), 1.5),
            phong_material((.05, .06, .05), (.45, .55, .40), (.2, .2, .2), 8.0),
            phong_material((.05, .05, .06), (.50, .50, .60), (.1, .1, .1), 4.0, mirror=(.25, .25, .25))]
    objs = [sphere((0, 1, 0), 1.0, 0), sphere((2.3, 1, 0.4), 1.0, 1), sphere((-2.3, 1, 0.8), 1.0, 2),
            mesh_object(0, 3), plane(OBJ_GROUND, 0.0, 4), plane(OBJ_WALL_Z, -4.0, 3), plane(OBJ_WALL_X, -6.0, 3)]
    lights = [point_light((4, 7, 5), (.5, .5, .5), (.9, .9, .8), (1, 1, 1))]
    if light == "point":
        lights.append(point_light((-5, 4, 6), (.2, .2, .3), (.3, .3, .5), (.4, .4, .6)))
    else:
        lights.append(face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, (.5, .5, .6), ambient=(.2, .2, .2), specular=(.5, .5, .5)))
    body = blob((0.3, 2.9, 0.2), 0.7, 14, 15, seed=5)
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [body],
                     background=(0.05, 0.07, 0.1))
    cam = dict(look_from=(0.5, 2.8, 9), look_at=(0, 1.2, 0), vup=(0, 1, 0), vfov=42.0, aspect=4 / 3)
    return desc, cam


def animation_scene(width=160, height=120):
    """Scene of examples/main_animation.cpp: Phong sphere, Phong + mirror sphere, glass sphere on
    a Phong ground in front of two walls, a point Light (index 0) and a MovableFaceLight (index 1,
    light.cpp:87-107), seen through a ChangeFovCamera (camera.cpp:74-94).  Returns the
    description at time 0, the camera arguments and the index of the moving light."""
    mats = [phong_material((.10, .04, .04), (.70, .25, .25), (.5, .5, .5), 24.0),
            phong_material((.02, .02, .02), (.10, .10, .12), (.6, .6, .6), 60.0, mirror=(.75, .75, .8)),
            glass_material((.02, .03, .05), 1.5),
            phong_material((.05, .05, .06), (.50, .50, .60), (.1, .1, .1), 4.0),
            phong_material((.05, .06, .05), (.45, .55, .40), (.2, .2, .2), 8.0)]
    objs = [sphere((0, 1, 0), 1.0, 0), sphere((2.3, 1, 0.4), 1.0, 1), sphere((-2.3, 1, 0.8), 1.0, 2),
            plane(OBJ_GROUND, 0.0, 3), plane(OBJ_WALL_Z, -4.0, 4), plane(OBJ_WALL_X, -6.0, 4)]
    lights = [point_light((4, 7, 5), (.5, .5, .5), (.9, .9, .8), (1, 1, 1)),
              face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, (.5, .5, .6), ambient=(.2, .2, .2), specular=(.5, .5, .5))]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [],
                     background=(0.05, 0.07, 0.1))
    cam = dict(look_from=(0.5, 2.8, 9), look_at=(0, 1.2, 0), vup=(0, 1, 0), vfov=30.0, aspect=width * 1.0 / height)
    return desc, cam, 1


def kaleidoscope_scene(width=1020, height=630):
    """The reference driver's own scene (main.cpp:29-72): two huge mirror spheres and three
    coloured, mirror-finished SphereLights."""
    PI = 3.1415926
    R, T, Z = 300.0, 30 * PI / 180., 62.0
    D = R / math.cos(T)
    r, g, b = (1.2, 1, 1), (1, 1.1, 1), (1, 1, 1)
    C, emission = (0.3, 0.7, 0.8), 0.8e-1
    em = lambda k: tuple(C[i] * emission * k[i] for i in range(3))
    q = R * 2 * math.sqrt(2. / 3.)
    base = np.array([50, 28, Z])
    mats = [material(MAT_SPECULAR, emission=(0, 0, 0), color=(.95, .95, .95)),      # mback
            material(MAT_SPECULAR, emission=(0, 0, 0), color=(.93, .93, .93)),      # mfront
            material(MAT_SPECULAR, emission=em(r), color=(1.15, .996, .996)),
            material(MAT_SPECULAR, emission=em(g), color=(.99, 1.03, .99)),
            material(MAT_SPECULAR, emission=em(b), color=(.996, .996, .996))]
    lpos = [base + np.array([math.cos(T), math.sin(T), 0]) * D, base + np.array([-math.cos(T), math.sin(T), 0]) * D,
            base + np.array([0, -1, 0]) * D]
    objs = [sphere(base + np.array([0, 0, -1]) * q, R, 0), sphere(base + np.array([0, 0, -q / 3.]), 2 * 2 * q - q / 3., 1)]
    lights = []
    for i, (pos, k) in enumerate(zip(lpos, (r, g, b))):
        lights.append(sphere_light(pos, R, em(k)))
        objs.append(sphere(pos, R, 2 + i))
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [], background=(0, 0, 0))
    cam = dict(look_from=(50, 40, -120), look_at=(50, 40, Z), vup=(0, 1, 0), vfov=95.0, aspect=width * 1.0 / height)
    return desc, cam


def implicit_object(kind, center, scale, mat):
    """HeartShape(center, scale) / SurfaceOfGenus2(center, scale) -- Surface.h:151-174, 131-147."""
    o = np.zeros((), OBJECT_DT)
    o["type"], o["material"], o["mesh"] = kind, mat, -1
    o["p"] = (center[0], center[1], center[2], scale)
    return o


def check_material(kind, box=None):
    """NormalCheckMaterial() / PositionCheckMaterial(box) -- material.h:70-92: Phong materials with
    black colours whose ambient term shows |normal| or the hit's position inside `box`."""
    m = phong_material((0, 0, 0), (0, 0, 0), (0, 0, 0), 1.0)
    m["check"] = kind
    if box is not None:
        m["check_box"] = tuple(box[0]) + tuple(box[1])
    return m


def implicit_scene(variant="montecarlo", min_size=1e-2):
    """Implicit surfaces through their voxel octrees (HeartShape, SurfaceOfGenus2): two hearts
    (one tree, two placements) and a genus-2 surface on a ground plane in front of two walls.
    variant "montecarlo": diffuse / mirror Monte-Carlo materials under a FaceLight;
    variant "check": the reference's NormalCheck / PositionCheck materials and a Phong material
    under a point Light (the reference's debug view of an implicit surface)."""
    if variant == "montecarlo":
        mats = [material(MAT_NORMAL | MAT_MONTECARLO, color=(.75, .25, .25)),
                material(MAT_SPECULAR | MAT_MONTECARLO, color=(.9, .9, .9)),
                material(MAT_NORMAL | MAT_MONTECARLO, color=(.3, .6, .35)),
                material(MAT_NORMAL | MAT_MONTECARLO, color=(.7, .7, .7))]
        lights = [face_light((0, 5.5, 2), (0, -1, 0), (0, 0, 1), 3.0, 3.0, (14, 14, 13))]
    else:
        mats = [check_material(CHECK_NORMAL), check_material(CHECK_POSITION, ((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5))),
                phong_material((.05, .08, .05), (.3, .6, .35), (.4, .4, .4), 20.0, mirror=(.3, .3, .3)),
                phong_material((.06, .06, .06), (.6, .6, .6), (.1, .1, .1), 4.0)]
        lights = [point_light((3, 7, 6), (1, 1, 1), (.9, .9, .8), (1, 1, 1))]
    objs = [implicit_object(OBJ_HEART, (-1.6, 1.6, 0.2), 1.1, 0), implicit_object(OBJ_HEART, (1.9, 1.2, -0.6), 0.8, 1),
            implicit_object(OBJ_GENUS2, (0.3, 1.1, 1.8), 0.9, 2),
            plane(OBJ_GROUND, 0.0, 3), plane(OBJ_WALL_Z, -3.5, 3), plane(OBJ_WALL_X, -5.0, 3)]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [],
                     background=(0.06, 0.07, 0.1), implicit_min_size=min_size)
    cam = dict(look_from=(0.6, 3.0, 8.5), look_at=(0.1, 1.3, 0), vup=(0, 1, 0), vfov=40.0, aspect=4 / 3)
    return desc, cam


def heart_example_scene(width=320, height=240, min_size=1e-2):
    """The scene of examples/main_heart.cpp."""
    mats = [check_material(CHECK_NORMAL), material(MAT_NORMAL, color=(.75, .25, .25)),
            phong_material((.05, .08, .05), (.3, .6, .35), (.4, .4, .4), 20.0, mirror=(.3, .3, .3)),
            phong_material((.06, .06, .06), (.6, .6, .6), (.1, .1, .1), 4.0)]
    objs = [implicit_object(OBJ_HEART, (-1.6, 1.6, 0.2), 1.1, 0), implicit_object(OBJ_HEART, (1.9, 1.2, -0.6), 0.8, 1),
            implicit_object(OBJ_GENUS2, (0.3, 1.1, 1.8), 0.9, 2), plane(OBJ_GROUND, 0.0, 3), plane(OBJ_WALL_Z, -3.5, 3)]
    lights = [point_light((3, 7, 6), (1, 1, 1), (.9, .9, .8), (1, 1, 1)),
              face_light((0, 5.5, 2), (0, -1, 0), (0, 0, 1), 3.0, 3.0, (.6, .6, .55), ambient=(.1, .1, .1), specular=(.3, .3, .3))]
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [],
                     background=(0.06, 0.07, 0.1), implicit_min_size=min_size)
    cam = dict(look_from=(0.6, 3.0, 8.5), look_at=(0.1, 1.3, 0), vup=(0, 1, 0), vfov=40.0, aspect=width * 1.0 / height)
    return desc, cam


def perlin_material(kind, ambient, diffuse, specular, shine, scale=1.0):
    """PerlinNoiseNormalMaterial / PerlinNoiseNormalMaterial_Blood (amb, dif, spe, shi, scale) --
    material.h:95-164."""
    m = phong_material(ambient, diffuse, specular, shine)
    m["perlin"], m["perlin_scale"] = kind, scale
    return m


def perlin_scene(tables=None):
    """Perlin materials (material.h:95-164) on a sphere, a small mesh, a heart and the ground, under
    a point Light and a FaceLight: PerlinNoiseNormalMaterial and the _Blood variant (noise-tilted
    normals, so the mirror of the Phong + mirror sphere next to them sees perturbed surfaces)."""
    from .desc import PERLIN_BLOOD, PERLIN_NORMAL
    mats = [perlin_material(PERLIN_NORMAL, (.10, .08, .05), (.80, .70, .40), (.5, .5, .5), 24.0, 0.35),
            perlin_material(PERLIN_BLOOD, (.30, .25, .25), (.80, .60, .55), (.6, .6, .6), 30.0, 0.5),
            phong_material((.02, .02, .02), (.10, .10, .12), (.6, .6, .6), 60.0, mirror=(.75, .75, .8)),
            perlin_material(PERLIN_NORMAL, (.06, .06, .08), (.55, .55, .70), (.1, .1, .1), 4.0, 1.3),
            perlin_material(PERLIN_BLOOD, (.25, .2, .2), (.7, .5, .5), (.4, .4, .4), 12.0, 0.8)]
    objs = [sphere((0, 1, 0), 1.0, 0), sphere((2.3, 1, 0.4), 1.0, 2), implicit_object(OBJ_HEART, (-2.4, 1.5, 0.6), 0.9, 1),
            mesh_object(0, 4), plane(OBJ_GROUND, 0.0, 3), plane(OBJ_WALL_Z, -4.0, 3)]
    lights = [point_light((4, 7, 5), (.5, .5, .5), (.9, .9, .8), (1, 1, 1)),
              face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, (.5, .5, .6), ambient=(.2, .2, .2), specular=(.5, .5, .5))]
    body = blob((0.3, 2.9, 0.2), 0.7, 14, 15, seed=5)
    desc = SceneDesc(_stack(objs, OBJECT_DT), _stack(mats, MATERIAL_DT), _stack(lights, LIGHT_DT), [body],
                     background=(0.05, 0.07, 0.1), implicit_min_size=1.5e-2, perlin_tables=tables)
    cam = dict(look_from=(0.5, 2.8, 9), look_at=(0, 1.2, 0), vup=(0, 1, 0), vfov=42.0, aspect=4 / 3)
    return desc, cam
