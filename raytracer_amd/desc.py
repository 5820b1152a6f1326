"""Plain-data scene description shared by every caller of the C boundary.

The records mirror, field for field, the POD structs declared in ``include/rt_hip.h``
(numpy structured dtypes with C alignment + ctypes mirrors of the top-level structs).
Nothing here computes anything: it is the wire format between the host-side scene API
(``raytracer_amd.scene``) and the shared library.

Reference facts encoded here: material type bits (material.h:10-16), node kinds
(Surface.h:50-56), sampler / integrator defaults (scene.h:52-54), octree defaults
(octree.h:26,50).
"""
import ctypes as C

import numpy as np

# material type bits -- material.h:10-16
MAT_NORMAL, MAT_SPECULAR, MAT_TRANSPARENT, MAT_MONTECARLO = 1, 2, 4, 8
# object kinds -- Surface.h:50-56 (+ the triangle mesh this build adds, + the implicit surfaces
# HeartShape, Surface.h:151-174, and SurfaceOfGenus2, Surface.h:121-147: p = centre, scale)
OBJ_SPHERE, OBJ_GROUND, OBJ_WALL_Z, OBJ_WALL_X, OBJ_MESH, OBJ_HEART, OBJ_GENUS2 = 1, 2, 3, 4, 5, 6, 7
# NormalCheckMaterial / PositionCheckMaterial (material.h:70-92)
CHECK_NONE, CHECK_NORMAL, CHECK_POSITION = 0, 1, 2
# PerlinNoiseNormalMaterial / PerlinNoiseNormalMaterial_Blood (material.h:95-164)
PERLIN_NONE, PERLIN_NORMAL, PERLIN_BLOOD = 0, 1, 2
LIGHT_SPHERE, LIGHT_FACE, LIGHT_POINT = 1, 2, 3
MISS = 0xFFFFFFFF

INFINITE = 1e10   # definitions.h:5
ZERO = 1e-9       # definitions.h:6
PI = 3.1415926    # definitions.h:7 (7 digits, kept on purpose)

OBJECT_DT = np.dtype([("type", "<i4"), ("material", "<i4"), ("mesh", "<i4"), ("reserved", "<i4"),
                      ("p", "<f8", (4,))], align=True)
MATERIAL_DT = np.dtype([("type", "<i4"), ("check", "<i4"), ("emission", "<f8", (3,)),
                        ("color", "<f8", (3,)), ("blur", "<f8"), ("atten", "<f8", (3,)),
                        ("nr", "<f8"), ("ambient", "<f8", (3,)), ("specular", "<f8", (3,)),
                        ("mirror", "<f8", (3,)), ("shine", "<f8"), ("check_box", "<f8", (6,)),
                        ("perlin", "<i4"), ("reserved", "<i4"), ("perlin_scale", "<f8")], align=True)
LIGHT_DT = np.dtype([("type", "<i4"), ("reserved", "<i4"), ("pos", "<f8", (3,)), ("radius", "<f8"),
                     ("normal", "<f8", (3,)), ("up", "<f8", (3,)), ("right", "<f8", (3,)),
                     ("width", "<f8"), ("height", "<f8"), ("diffuse", "<f8", (3,)),
                     ("ambient", "<f8", (3,)), ("specular", "<f8", (3,))], align=True)
RAY_DT = np.dtype([("e", "<f8", (3,)), ("d", "<f8", (3,))], align=True)

assert OBJECT_DT.itemsize == 48 and MATERIAL_DT.itemsize == 240 and LIGHT_DT.itemsize == 200


class MeshC(C.Structure):
    _fields_ = [("tris", C.c_void_p), ("ntris", C.c_uint64)]


class SceneDescC(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("nobjects", C.c_uint64),
                ("materials", C.c_void_p), ("nmaterials", C.c_uint64),
                ("lights", C.c_void_p), ("nlights", C.c_uint64),
                ("meshes", C.c_void_p), ("nmeshes", C.c_uint64),
                ("background", C.c_double * 3),
                ("tri_box_pad", C.c_double),
                ("octree_max_size", C.c_int32), ("octree_overlap_tol", C.c_int32),
                ("octree_max_depth", C.c_int32), ("reserved", C.c_int32),
                ("implicit_min_size", C.c_double),
                ("perlin_p", C.c_void_p), ("perlin_g", C.c_void_p), ("perlin_n", C.c_int32), ("reserved2", C.c_int32)]


class CameraC(C.Structure):
    _fields_ = [("e", C.c_double * 3), ("llc", C.c_double * 3), ("u", C.c_double * 3),
                ("v", C.c_double * 3), ("width", C.c_double), ("height", C.c_double),
                ("look_at", C.c_double * 3), ("aperture", C.c_double)]


class RenderParamsC(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("max_jump", C.c_int32), ("rus_jump", C.c_int32),
                ("seed", C.c_uint64), ("face_forward", C.c_int32), ("reserved", C.c_int32)]


class SceneDesc:
    """A flattened scene: objects in Group child order, materials, lights, triangle meshes."""

    def __init__(self, objects, materials, lights, meshes, background=(0.0, 0.0, 0.0),
                 tri_box_pad=1e-7, octree_max_size=128, octree_overlap_tol=3, octree_max_depth=12,
                 implicit_min_size=0.0, perlin_tables=None):
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DT)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        self.lights = np.ascontiguousarray(lights, dtype=LIGHT_DT)
        self.meshes = [np.ascontiguousarray(m, dtype=np.float64).reshape(-1, 9) for m in meshes]
        self.background = tuple(float(x) for x in background)
        self.tri_box_pad = float(tri_box_pad)
        self.octree_max_size = int(octree_max_size)
        self.octree_overlap_tol = int(octree_overlap_tol)
        self.octree_max_depth = int(octree_max_depth)
        self.implicit_min_size = float(implicit_min_size)   # 0 = the reference's 3e-3
        # (P, G) of the Perlin generator, or None for the library's replay of the reference's own
        self.perlin_tables = None
        if perlin_tables is not None:
            P = np.ascontiguousarray(perlin_tables[0], np.int32)
            G = np.ascontiguousarray(perlin_tables[1], np.float64).reshape(-1, 3)
            assert len(P) == len(G)
            self.perlin_tables = (P, G)

    def to_c(self):
        """Return (SceneDescC, keepalive) -- keepalive must outlive the call that reads it."""
        mesh_arr = (MeshC * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            mesh_arr[i].tris = m.ctypes.data
            mesh_arr[i].ntris = m.shape[0]
        d = SceneDescC()
        d.objects, d.nobjects = self.objects.ctypes.data, len(self.objects)
        d.materials, d.nmaterials = self.materials.ctypes.data, len(self.materials)
        d.lights, d.nlights = self.lights.ctypes.data, len(self.lights)
        d.meshes, d.nmeshes = C.addressof(mesh_arr), len(self.meshes)
        d.background = (C.c_double * 3)(*self.background)
        d.tri_box_pad = self.tri_box_pad
        d.octree_max_size = self.octree_max_size
        d.octree_overlap_tol = self.octree_overlap_tol
        d.octree_max_depth = self.octree_max_depth
        d.implicit_min_size = self.implicit_min_size
        if self.perlin_tables is not None:
            d.perlin_p, d.perlin_g = self.perlin_tables[0].ctypes.data, self.perlin_tables[1].ctypes.data
            d.perlin_n = len(self.perlin_tables[0])
        return d, (mesh_arr, self)

    @property
    def prim_base(self):
        """Global primitive id of each object's first primitive (mesh: one id per triangle).
        Scenes with implicit surfaces have one id per voxel; their count is only known to the
        library that built the voxels, so this helper refuses them."""
        base, out = 0, []
        for o in self.objects:
            out.append(base)
            if int(o["type"]) in (OBJ_HEART, OBJ_GENUS2):
                raise ValueError("prim_base: the scene holds implicit surfaces")
            base += self.meshes[int(o["mesh"])].shape[0] if int(o["type"]) == OBJ_MESH else 1
        return np.asarray(out + [base], dtype=np.int64)


def render_params(width, height, n, m=None, sample_begin=0, sample_end=None, max_jump=40,
                  rus_jump=12, seed=1243, face_forward=False):
    """Sampler/integrator parameters; defaults are the reference's (scene.h:52-54, scene.cpp:22)."""
    m = n if m is None else m
    p = RenderParamsC()
    p.width, p.height, p.n, p.m = int(width), int(height), int(n), int(m)
    p.sample_begin = int(sample_begin)
    p.sample_end = int(n * m if sample_end is None else sample_end)
    p.max_jump, p.rus_jump = int(max_jump), int(rus_jump)
    p.seed = int(seed)
    p.face_forward = 1 if face_forward else 0
    return p
