"""ctypes bindings for the CPU oracle (librt_oracle.so) and, when present, the compiled
reference harness (oracle/_ref/libref_harness.so).

TEST INFRASTRUCTURE ONLY: imported from tests/, from __graft_entry__.smoke() and from the
cpu_baseline leg of bench.py -- never from the product package.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd.desc import CameraC, RAY_DT, SceneDescC, RenderParamsC, LIGHT_DT

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "librt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_harness.so")

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _d3(x):
    return (C.c_double * 3)(*[float(v) for v in x])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDescC)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_octree_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_implicit_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_implicit_boxes.restype = C.c_int64
        L.orc_implicit_boxes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_implicit_func.restype = C.c_double
        L.orc_implicit_func.argtypes = [C.c_int, _dp]
        L.orc_implicit_normal.argtypes = [C.c_int, _dp, _dp]
        L.orc_perlin_noise.restype = C.c_double
        L.orc_perlin_noise.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_perlin_tables.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_perlin_time_changed.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_perlin_material.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _dp, _dp, _dp]
        L.orc_camera_init.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(CameraC)]
        L.orc_camera_get_ray.argtypes = [C.POINTER(CameraC), C.c_double, C.c_double, C.c_void_p]
        L.orc_camera_get_ray_disk.argtypes = [C.POINTER(CameraC), C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_camera_change_fov.argtypes = [C.POINTER(CameraC), C.c_double, C.POINTER(CameraC)]
        L.orc_light_move_face.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_primary.argtypes = [C.c_void_p, C.POINTER(CameraC), C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_ray_color.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_int,
                                    C.POINTER(RenderParamsC), C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraC), C.POINTER(RenderParamsC),
                                 C.c_void_p, C.c_void_p, C.c_int]
        L.orc_render_range.argtypes = [C.c_void_p, C.POINTER(CameraC), C.POINTER(RenderParamsC),
                                       C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_resolve_bgr8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_resolve_tone_bgr8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_tone.restype = C.c_double
        L.orc_tone.argtypes = [C.c_int, C.c_double]
        L.orc_hit_box.argtypes = [_dp, _dp, C.c_void_p, C.c_double, C.c_double, _dp, _dp]
        L.orc_sphere_hit.argtypes = [_dp, C.c_double, C.c_void_p, C.c_double, C.c_double, _dp]
        L.orc_plane_hit.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_double, C.c_double, _dp]
        L.orc_triangle_hit.argtypes = [_dp, C.c_void_p, C.c_double, C.c_double, _dp]
        L.orc_refract.argtypes = [_dp, _dp, C.c_double, _dp]
        L.orc_child_bound.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        L.orc_light_sample.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
        L.orc_jitter_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _dp]
        L.orc_hash2.restype = C.c_uint64
        L.orc_hash2.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint64]
        L.orc_permute.restype = C.c_uint32
        L.orc_permute.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_path_root_key.restype = C.c_uint64
        L.orc_path_root_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        _lib = L
    return _lib


class OracleScene:
    """Owns an orc_scene built from a raytracer_amd.desc.SceneDesc."""

    def __init__(self, desc):
        self.desc = desc
        d, keep = desc.to_c()
        self._h = lib().orc_scene_create(C.byref(d))
        del keep
        if not self._h:
            raise RuntimeError("orc_scene_create: " + lib().orc_last_error().decode())

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: " + lib().orc_last_error().decode())

    def octree_stats(self, mesh=0):
        out = (C.c_uint64 * 6)()
        self._chk(lib().orc_octree_stats(self._h, mesh, out), "orc_octree_stats")
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    def implicit_stats(self, kind):
        out = (C.c_uint64 * 6)()
        self._chk(lib().orc_implicit_stats(self._h, kind, out), "orc_implicit_stats")
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    def implicit_boxes(self, kind):
        n = lib().orc_implicit_boxes(self._h, kind, None, 0)
        if n < 0:
            raise RuntimeError("orc_implicit_boxes: " + lib().orc_last_error().decode())
        out = np.empty((n, 6), np.float64)
        lib().orc_implicit_boxes(self._h, kind, _p(out), n)
        return out

    def trace_rays(self, rays, t0=0.0, t1=1e10):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        n = len(rays)
        prim = np.empty(n, np.uint32); t = np.empty(n, np.float64); nrm = np.empty((n, 3), np.float64)
        cnt = np.zeros(4, np.uint64)
        self._chk(lib().orc_trace_rays(self._h, _p(rays), n, t0, t1, _p(prim), _p(t), _p(nrm), _p(cnt)), "orc_trace_rays")
        return prim, t, nrm, cnt

    def trace_primary(self, cam, w, h, threads=8):
        prim = np.empty(w * h, np.uint32); t = np.empty(w * h, np.float64)
        cnt = np.zeros(4, np.uint64)
        self._chk(lib().orc_trace_primary(self._h, C.byref(cam), w, h, _p(prim), _p(t), _p(cnt), threads), "orc_trace_primary")
        return prim.reshape(h, w), t.reshape(h, w), cnt

    def ray_color(self, rays, params, t0=0.0, t1=1e10, jumptime=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        out = np.zeros((len(rays), 3), np.float64)
        cnt = np.zeros(4, np.uint64)
        self._chk(lib().orc_ray_color(self._h, _p(rays), len(rays), t0, t1, jumptime, C.byref(params), _p(out), _p(cnt)), "orc_ray_color")
        return out, cnt

    def render(self, cam, params, threads=8, pix_range=None):
        out = np.zeros((params.height, params.width, 3), np.float64)
        cnt = np.zeros(4, np.uint64)
        if pix_range is None:
            self._chk(lib().orc_render(self._h, C.byref(cam), C.byref(params), _p(out), _p(cnt), threads), "orc_render")
        else:
            self._chk(lib().orc_render_range(self._h, C.byref(cam), C.byref(params), pix_range[0], pix_range[1],
                                             _p(out), _p(cnt), threads), "orc_render_range")
        return out, cnt


def camera_init(look_from, look_at, vup, vfov, aspect):
    cam = CameraC()
    lib().orc_camera_init(_d3(look_from), _d3(look_at), _d3(vup), float(vfov), float(aspect), C.byref(cam))
    return cam


def camera_change_fov(origin, time):
    out = CameraC()
    lib().orc_camera_change_fov(C.byref(origin), C.c_double(time), C.byref(out))
    return out


def light_move_face(origin, time):
    from raytracer_amd.desc import LIGHT_DT
    src = np.ascontiguousarray(np.asarray(origin, LIGHT_DT).reshape(1))
    out = np.zeros(1, LIGHT_DT)
    lib().orc_light_move_face(src.ctypes.data_as(C.c_void_p), C.c_double(time), out.ctypes.data_as(C.c_void_p))
    return out[0]


def implicit_func(kind, pos):
    return lib().orc_implicit_func(kind, _d3(pos))


def implicit_normal(kind, pos):
    out = (C.c_double * 3)()
    lib().orc_implicit_normal(kind, _d3(pos), out)
    return np.array(out[:])


def perlin_noise(x, y, z, scale, P, G):
    P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
    return lib().orc_perlin_noise(x, y, z, scale, _p(P), _p(G), len(P))


def perlin_tables(seed=12325, n=256):
    P = np.zeros(n, np.int32); G = np.zeros((n, 3), np.float64)
    lib().orc_perlin_tables(seed, n, _p(P), _p(G))
    return P, G


def perlin_time_changed(clock, prev_time, time, G):
    """clock: a raytracer_amd.hip.PerlinClockC-compatible ctypes struct; G updated in place."""
    assert G.dtype == np.float64 and G.flags.c_contiguous
    lib().orc_perlin_time_changed(C.byref(clock), float(prev_time), float(time), len(G), _p(G))


def perlin_material(mat, P, G, pos, normal):
    """ambient, diffuse, specular, adjusted normal of a MATERIAL_DT record at `pos`."""
    from raytracer_amd.desc import MATERIAL_DT
    m = np.zeros(1, MATERIAL_DT); m[0] = mat
    P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
    out = (C.c_double * 12)()
    lib().orc_perlin_material(_p(m), _p(P), _p(G), len(P), _d3(pos), _d3(normal), out)
    return np.array(out[:]).reshape(4, 3)


def camera_get_ray(cam, s, t):
    r = np.zeros(1, RAY_DT)
    lib().orc_camera_get_ray(C.byref(cam), float(s), float(t), _p(r))
    return r[0]


def camera_get_ray_disk(cam, s, t, x, y):
    r = np.zeros(1, RAY_DT)
    lib().orc_camera_get_ray_disk(C.byref(cam), float(s), float(t), float(x), float(y), _p(r))
    return r[0]


def resolve_bgr8(radiance_sum, spp, tone=0):
    h, w, _ = radiance_sum.shape
    a = np.ascontiguousarray(radiance_sum, np.float64)
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_resolve_tone_bgr8(_p(a), w, h, spp, tone, _p(out))
    return out


def tone(kind, x):
    return lib().orc_tone(kind, float(x))


def _ray1(e, d):
    r = np.zeros(1, RAY_DT)
    r["e"][0] = e; r["d"][0] = d
    return r


def hit_box(lo, hi, e, d, t0, t1):
    r0, r1 = C.c_double(), C.c_double()
    h = lib().orc_hit_box(_d3(lo), _d3(hi), _p(_ray1(e, d)), t0, t1, C.byref(r0), C.byref(r1))
    return h, r0.value, r1.value


def sphere_hit(c, radius, e, d, t0, t1):
    t = C.c_double()
    h = lib().orc_sphere_hit(_d3(c), radius, _p(_ray1(e, d)), t0, t1, C.byref(t))
    return h, t.value


def plane_hit(kind, coord, e, d, t0, t1):
    t = C.c_double()
    h = lib().orc_plane_hit(kind, coord, _p(_ray1(e, d)), t0, t1, C.byref(t))
    return h, t.value


def triangle_hit(tri, e, d, t0, t1):
    t = C.c_double()
    a = (C.c_double * 9)(*[float(x) for x in np.asarray(tri).reshape(9)])
    h = lib().orc_triangle_hit(a, _p(_ray1(e, d)), t0, t1, C.byref(t))
    return h, t.value


def refract(d, n, eta):
    out = (C.c_double * 3)()
    rc = lib().orc_refract(_d3(d), _d3(n), eta, out)
    return rc, np.array(out[:])


def child_bound(lo, hi, child):
    a, b = (C.c_double * 3)(), (C.c_double * 3)()
    lib().orc_child_bound(_d3(lo), _d3(hi), child, a, b)
    return np.array(a[:]), np.array(b[:])


def light_sample(light_rec, x, u, v):
    arr = np.zeros(1, LIGHT_DT)
    arr[0] = light_rec
    pos, nl, pdf = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    lib().orc_light_sample(_p(arr), _d3(x), u, v, pos, nl, C.byref(pdf))
    return np.array(pos[:]), np.array(nl[:]), pdf.value


def jitter_pair(seed, pixel, object_id, k, n, m):
    out = (C.c_double * 2)()
    lib().orc_jitter_pair(seed, pixel, object_id, k, n, m, out)
    return out[0], out[1]


# ----------------------------------------------------------------------------- reference harness
_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref():
    """The unmodified reference objects behind a C ABI (only where oracle/_ref was built)."""
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_hit_box.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp]
        R.ref_sphere_hit.argtypes = [_dp, C.c_double, _dp, C.c_double, C.c_double, _dp, _dp]
        R.ref_plane_hit.argtypes = [C.c_int, C.c_double, _dp, C.c_double, C.c_double, _dp, _dp]
        R.ref_child_bound.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        R.ref_box_overlap.argtypes = [_dp, _dp, _dp, _dp]
        R.ref_camera_init.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, _dp]
        R.ref_camera_get_ray.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        R.ref_sphere_light_sample.argtypes = [_dp, C.c_double, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
        R.ref_face_light_sample.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
        R.ref_reflact.argtypes = [_dp, _dp, _dp]
        R.ref_normalize.argtypes = [_dp, _dp]
        R.ref_cross.argtypes = [_dp, _dp, _dp]
        R.ref_first_intersection_sphere.argtypes = [_dp, _dp, C.c_double, _dp, _dp]
        R.ref_tree_create.restype = C.c_void_p
        R.ref_tree_create.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_int]
        R.ref_tree_destroy.argtypes = [C.c_void_p]
        R.ref_tree_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        R.ref_tree_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_group_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_implicit_create.restype = C.c_void_p
        R.ref_implicit_create.argtypes = [C.c_int, C.c_double]
        R.ref_implicit_destroy.argtypes = [C.c_void_p]
        R.ref_implicit_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        R.ref_implicit_boxes.restype = C.c_int64
        R.ref_implicit_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        R.ref_implicit_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        R.ref_implicit_func.restype = C.c_double
        R.ref_implicit_func.argtypes = [C.c_int, _dp]
        R.ref_implicit_object_hit.argtypes = [C.c_int, _dp, C.c_double, C.c_void_p, C.c_uint64, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_perlin_noise.restype = C.c_double
        R.ref_perlin_noise.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        R.ref_perlin_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        R.ref_perlin_material.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, _dp, _dp, _dp]
        R.ref_perlin_time_changed.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_void_p]
        R.ref_change_fov_camera.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp]
        R.ref_movable_face_light.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp]
        R.ref_depth_camera_ray.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_uint, C.c_double, C.c_double, _dp, _dp]
        R.ref_tone.argtypes = [C.c_int, _dp, _dp]
        for f in ("ref_const_zero", "ref_const_infinite", "ref_const_pi"):
            getattr(R, f).restype = C.c_double
        _ref = R
    return _ref


def _d6(e, d):
    return (C.c_double * 6)(*[float(v) for v in list(e) + list(d)])


def ref_hit_box(lo, hi, e, d, t0, t1):
    r0, r1 = C.c_double(), C.c_double()
    h = ref().ref_hit_box(_d3(lo), _d3(hi), _d6(e, d), t0, t1, C.byref(r0), C.byref(r1))
    return h, r0.value, r1.value


def ref_sphere_hit(c, radius, e, d, t0, t1):
    t, n = C.c_double(), (C.c_double * 3)()
    h = ref().ref_sphere_hit(_d3(c), radius, _d6(e, d), t0, t1, C.byref(t), n)
    return h, t.value, np.array(n[:])


def ref_plane_hit(kind, coord, e, d, t0, t1):
    t, n = C.c_double(), (C.c_double * 3)()
    h = ref().ref_plane_hit(kind, coord, _d6(e, d), t0, t1, C.byref(t), n)
    return h, t.value, np.array(n[:])


def ref_child_bound(lo, hi, child):
    a, b = (C.c_double * 3)(), (C.c_double * 3)()
    ref().ref_child_bound(_d3(lo), _d3(hi), child, a, b)
    return np.array(a[:]), np.array(b[:])


def ref_camera_init(look_from, look_at, vup, vfov, aspect):
    out = (C.c_double * 14)()
    ref().ref_camera_init(_d3(look_from), _d3(look_at), _d3(vup), vfov, aspect, out)
    return np.array(out[:])


def ref_camera_get_ray(look_from, look_at, vup, vfov, aspect, s, t):
    out = (C.c_double * 6)()
    ref().ref_camera_get_ray(_d3(look_from), _d3(look_at), _d3(vup), vfov, aspect, s, t, out)
    return np.array(out[:])


def ref_sphere_light_sample(pos, radius, x, u, v):
    p, n, pdf = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    ref().ref_sphere_light_sample(_d3(pos), radius, _d3(x), u, v, p, n, C.byref(pdf))
    return np.array(p[:]), np.array(n[:]), pdf.value


def ref_face_light_sample(pos, normal, up, w, h, u, v):
    p, n, pdf, right = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double(), (C.c_double * 3)()
    ref().ref_face_light_sample(_d3(pos), _d3(normal), _d3(up), w, h, u, v, p, n, C.byref(pdf), right)
    return np.array(p[:]), np.array(n[:]), pdf.value, np.array(right[:])


class RefTree:
    """The reference's own Octree filled with harness-side triangles."""

    def __init__(self, tris, pad=1e-7, max_size=128):
        self.tris = np.ascontiguousarray(tris, np.float64).reshape(-1, 9)
        self._h = ref().ref_tree_create(_p(self.tris), len(self.tris), pad, max_size)

    def stats(self):
        out = (C.c_uint64 * 6)()
        ref().ref_tree_stats(self._h, out)
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    def hit(self, rays, t0=0.0, t1=1e10):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        n = len(rays)
        prim = np.empty(n, np.uint32); t = np.empty(n, np.float64)
        tests = C.c_uint64(0)
        ref().ref_tree_hit(self._h, _p(rays), n, t0, t1, _p(prim), _p(t), C.byref(tests))
        return prim, t, tests.value

    def close(self):
        if self._h:
            ref().ref_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ref_group_hit(objs, rays, t0=0.0, t1=1e10):
    """objs: rows {type, p0..p3, unused}; returns (child index or -1, t, normal)."""
    objs = np.ascontiguousarray(objs, np.float64).reshape(-1, 6)
    rays = np.ascontiguousarray(rays, dtype=RAY_DT)
    n = len(rays)
    which = np.empty(n, np.int32); t = np.empty(n, np.float64); nrm = np.empty((n, 3), np.float64)
    ref().ref_group_hit(_p(objs), len(objs), _p(rays), n, t0, t1, _p(which), _p(t), _p(nrm))
    return which, t, nrm


def _rays_flat(rays):
    return np.ascontiguousarray(rays, dtype=RAY_DT)


class RefImplicit:
    """The reference's ImplicitSurfaceOctree over one of its own implicit functions."""

    def __init__(self, kind, min_size):
        self._h = ref().ref_implicit_create(kind, float(min_size))
        if not self._h:
            raise RuntimeError("ref_implicit_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            ref().ref_implicit_destroy(self._h)
            self._h = None

    def stats(self):
        out = (C.c_uint64 * 6)()
        ref().ref_implicit_stats(self._h, out)
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    def boxes(self):
        n = ref().ref_implicit_boxes(self._h, None, 0)
        out = np.empty((n, 6), np.float64)
        ref().ref_implicit_boxes(self._h, _p(out), n)
        return out

    def hit(self, rays, t0=0.0, t1=1e10):
        flat = _rays_flat(rays)
        n = len(flat)
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float64)
        ref().ref_implicit_hit(self._h, _p(flat), n, t0, t1, _p(hit), _p(t))
        return hit, t


def ref_implicit_func(kind, pos):
    return ref().ref_implicit_func(kind, _d3(pos))


def ref_implicit_object_hit(kind, center, scale, rays, t0=0.0, t1=1e10):
    """HeartShape / SurfaceOfGenus2 through Node::calHit: hit flags, t, normals, localp."""
    flat = _rays_flat(rays)
    n = len(flat)
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float64)
    nrm = np.zeros((n, 3), np.float64); lp = np.zeros((n, 3), np.float64)
    ref().ref_implicit_object_hit(kind, _d3(center), float(scale), _p(flat), n, t0, t1, _p(hit), _p(t), _p(nrm), _p(lp))
    return hit, t, nrm, lp


def ref_perlin_tables(n=256):
    P = np.zeros(n, np.int32); G = np.zeros((n, 3), np.float64)
    ref().ref_perlin_tables(n, _p(P), _p(G))
    return P, G


def ref_perlin_noise(x, y, z, scale, P, G):
    P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
    return ref().ref_perlin_noise(x, y, z, scale, _p(P), _p(G), len(P))


def ref_perlin_material(kind, amb, dif, spe, shine, scale, P, G, pos, normal):
    P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
    out = (C.c_double * 12)()
    ref().ref_perlin_material(kind, _d3(amb), _d3(dif), _d3(spe), shine, scale, _p(P), _p(G), len(P), _d3(pos), _d3(normal), out)
    return np.array(out[:]).reshape(4, 3)


def ref_tone(kind, rgb):
    out = (C.c_double * 3)()
    ref().ref_tone(kind, _d3(rgb), out)
    return np.array(out[:])


def ref_perlin_time_changed(count, stage, stime, prev_time, time, G):
    """Returns (count, stage, stime) after the reference's timeChanged; G updated in place."""
    clk = (C.c_int32 * 2)(count, stage)
    st = C.c_double(stime)
    ref().ref_perlin_time_changed(clk, C.byref(st), prev_time, time, len(G), _p(G))
    return clk[0], clk[1], st.value


def ref_change_fov_camera(look_from, look_at, vup, vfov, aspect, time):
    out = (C.c_double * 14)()
    ref().ref_change_fov_camera(_d3(look_from), _d3(look_at), _d3(vup), vfov, aspect, time, out)
    return np.array(out[:])


def ref_movable_face_light(pos, normal, up, w, h, time, u, v):
    a, b = (C.c_double * 3)(), (C.c_double * 3)()
    ref().ref_movable_face_light(_d3(pos), _d3(normal), _d3(up), w, h, time, u, v, a, b)
    return np.array(a[:]), np.array(b[:])


def ref_depth_camera_ray(look_from, look_at, vup, vfov, aspect, aperture, seed, s, t):
    ray, xy = (C.c_double * 6)(), (C.c_double * 2)()
    ref().ref_depth_camera_ray(_d3(look_from), _d3(look_at), _d3(vup), vfov, aspect, aperture, seed, s, t, ray, xy)
    return np.array(ray[:]), np.array(xy[:])
