"""ctypes binding of librt_hip.so (the C ABI declared in include/rt_hip.h).

There is no fallback: if the shared library is missing or no HIP device is usable, every
call raises.  torch is imported first (when importable) only so that this process ends up
with ONE HIP runtime: the library and torch both resolve the soname libamdhip64.so.7 and
the first one loaded wins; loading them in the other order would give torch tensors and
this library different runtimes.
"""
import ctypes as C
import os

import numpy as np

from .desc import CameraC, RAY_DT, RenderParamsC, SceneDescC

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")   # RT_HIP_LIB: A/B a differently built library

EXPORTS = (
    "rt_last_error", "rt_device_count", "rt_set_device", "rt_scene_create", "rt_scene_destroy",
    "rt_scene_octree_stats", "rt_camera_init", "rt_trace_rays", "rt_trace_rays_host",
    "rt_trace_primary", "rt_trace_primary_host", "rt_ray_color", "rt_ray_color_host", "rt_ray_color_noalloc", "rt_render",
    "rt_render_host", "rt_scene_check", "rt_scene_set_profiling", "rt_scene_get_profile", "rt_scene_get_kernel_stats", "rt_scene_get_ray_flow", "rt_scene_set_workspace_limit", "rt_scene_workspace_bytes", "rt_resolve_bgr8", "rt_resolve_bgr8_host", "rt_save_bmp",
    "rt_camera_change_fov", "rt_light_move_face", "rt_scene_set_light",
    "rt_scene_implicit_stats", "rt_scene_implicit_boxes", "rt_perlin_tables", "rt_perlin_noise",
    "rt_resolve_tone_bgr8", "rt_resolve_tone_bgr8_host", "rt_avi_open", "rt_avi_open_mjpg", "rt_avi_add_frame", "rt_avi_close",
    "rt_perlin_time_changed", "rt_scene_set_perlin",
    "rt_comm_create", "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_size", "rt_comm_destroy", "rt_reduce_and_resolve",
    "rt_multi_create", "rt_multi_size", "rt_multi_scene", "rt_multi_render", "rt_multi_destroy",
)

_lib = None
_vp, _u64, _dbl, _int = C.c_void_p, C.c_uint64, C.c_double, C.c_int
_dp = C.POINTER(C.c_double)


def lib():
    """Load librt_hip.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    try:  # see module docstring
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.rt_last_error.restype = C.c_char_p
    L.rt_set_device.argtypes = [_int]
    L.rt_scene_create.argtypes = [C.POINTER(SceneDescC), C.POINTER(_vp)]
    L.rt_scene_destroy.argtypes = [_vp]
    L.rt_scene_octree_stats.argtypes = [_vp, _int, C.POINTER(_u64)]
    L.rt_camera_init.argtypes = [_dp, _dp, _dp, _dbl, _dbl, C.POINTER(CameraC)]
    L.rt_trace_rays.argtypes = [_vp, _vp, _u64, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp]
    L.rt_trace_rays_host.argtypes = [_vp, _vp, _u64, _dbl, _dbl, _vp, _vp, _vp, _vp]
    L.rt_trace_primary.argtypes = [_vp, C.POINTER(CameraC), _int, _int, _vp, _vp, _vp, _vp]
    L.rt_trace_primary_host.argtypes = [_vp, C.POINTER(CameraC), _int, _int, _vp, _vp, _vp]
    L.rt_ray_color.argtypes = [_vp, _vp, _u64, _dbl, _dbl, _int, C.POINTER(RenderParamsC), _vp, _vp, _vp]
    L.rt_ray_color_noalloc.argtypes = [_vp, _vp, _u64, _dbl, _dbl, _int, C.POINTER(RenderParamsC), _vp, _vp, _vp]
    L.rt_ray_color_host.argtypes = [_vp, _vp, _u64, _dbl, _dbl, _int, C.POINTER(RenderParamsC), _vp, _vp]
    L.rt_render.argtypes = [_vp, C.POINTER(CameraC), C.POINTER(RenderParamsC), _vp, _vp, _vp]
    L.rt_render_host.argtypes = [_vp, C.POINTER(CameraC), C.POINTER(RenderParamsC), _vp, _vp]
    L.rt_scene_check.argtypes = [_vp, _vp]
    L.rt_scene_set_profiling.argtypes = [_vp, _int]
    L.rt_scene_get_profile.argtypes = [_vp, _dp, _int]
    L.rt_scene_get_ray_flow.argtypes = [_vp, _dp, _int]
    L.rt_scene_get_kernel_stats.argtypes = [_vp, _dp, _int]
    L.rt_scene_set_workspace_limit.argtypes = [_vp, _u64]
    L.rt_scene_workspace_bytes.restype = C.c_int64
    L.rt_scene_workspace_bytes.argtypes = [_vp]
    L.rt_resolve_bgr8.argtypes = [_vp, _int, _int, _int, _vp, _vp]
    L.rt_resolve_bgr8_host.argtypes = [_vp, _int, _int, _int, _vp]
    L.rt_save_bmp.argtypes = [_vp, _int, _int, C.c_char_p]
    L.rt_camera_change_fov.argtypes = [C.POINTER(CameraC), _dbl, C.POINTER(CameraC)]
    L.rt_light_move_face.argtypes = [_vp, _dbl, _vp]
    L.rt_scene_set_light.argtypes = [_vp, _u64, _vp, _vp]
    L.rt_resolve_tone_bgr8.argtypes = [_vp, _int, _int, _int, _int, _vp, _vp]
    L.rt_resolve_tone_bgr8_host.argtypes = [_vp, _int, _int, _int, _int, _vp]
    L.rt_avi_open.argtypes = [C.c_char_p, _int, _int, _dbl, C.POINTER(_vp)]
    L.rt_avi_open_mjpg.argtypes = [C.c_char_p, _int, _int, _dbl, _int, C.POINTER(_vp)]
    L.rt_avi_add_frame.argtypes = [_vp, _vp]
    L.rt_avi_close.argtypes = [_vp]
    L.rt_perlin_time_changed.argtypes = [C.POINTER(PerlinClockC), _dbl, _dbl, _int, _vp]
    L.rt_scene_set_perlin.argtypes = [_vp, _vp, _vp, _int, _vp]
    L.rt_perlin_tables.argtypes = [C.c_uint32, _int, _vp, _vp]
    L.rt_perlin_noise.restype = _dbl
    L.rt_perlin_noise.argtypes = [_dbl, _dbl, _dbl, _dbl, _vp, _vp, _int]
    L.rt_scene_implicit_stats.argtypes = [_vp, _int, C.POINTER(C.c_uint64)]
    L.rt_scene_implicit_boxes.restype = C.c_int64
    L.rt_scene_implicit_boxes.argtypes = [_vp, _int, _vp, _u64]
    L.rt_comm_create.argtypes = [_int, C.POINTER(_int), C.POINTER(_vp)]
    L.rt_comm_unique_id.argtypes = [_vp]
    L.rt_comm_init_rank.argtypes = [_vp, _int, _int, _int, C.POINTER(_vp)]
    L.rt_comm_size.argtypes = [_vp]
    L.rt_comm_destroy.argtypes = [_vp]
    L.rt_reduce_and_resolve.argtypes = [_vp, _int, _vp, _int, _int, _int, _int, _vp, _vp]
    L.rt_multi_create.argtypes = [C.POINTER(SceneDescC), C.POINTER(_int), _int, _int, C.POINTER(_vp)]
    L.rt_multi_size.argtypes = [_vp]
    L.rt_multi_scene.restype = _vp
    L.rt_multi_scene.argtypes = [_vp, _int]
    L.rt_multi_render.argtypes = [_vp, C.POINTER(CameraC), C.POINTER(RenderParamsC), _int, _vp, _vp, _vp]
    L.rt_multi_destroy.argtypes = [_vp]
    _lib = L
    return L


class RtError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise RtError(f"{what} failed ({rc}): {lib().rt_last_error().decode(errors='replace')}")


def _p(a):
    return a.ctypes.data_as(_vp)


def _d3(x):
    return (C.c_double * 3)(*[float(v) for v in x])


def device_count():
    n = lib().rt_device_count()
    if n < 0:
        raise RtError("rt_device_count: " + lib().rt_last_error().decode())
    return n


def set_device(i):
    _chk(lib().rt_set_device(int(i)), "rt_set_device")


def camera_init(look_from, look_at, vup, vfov, aspect):
    """Camera(lookFrom, lookAt, vup, vfov, aspect) -- camera.cpp:5-8,37-48."""
    cam = CameraC()
    _chk(lib().rt_camera_init(_d3(look_from), _d3(look_at), _d3(vup), float(vfov), float(aspect), C.byref(cam)),
         "rt_camera_init")
    return cam


def camera_change_fov(origin, time):
    """ChangeFovCamera::timeChanged(time) applied to the camera `origin` -- camera.cpp:88-94."""
    out = CameraC()
    _chk(lib().rt_camera_change_fov(C.byref(origin), C.c_double(time), C.byref(out)), "rt_camera_change_fov")
    return out


def light_move_face(origin, time):
    """MovableFaceLight at `time` (light.cpp:87-107); `origin` and the result are LIGHT_DT records."""
    from .desc import LIGHT_DT
    src = np.ascontiguousarray(np.asarray(origin, LIGHT_DT).reshape(1))
    out = np.zeros(1, LIGHT_DT)
    _chk(lib().rt_light_move_face(src.ctypes.data_as(C.c_void_p), C.c_double(time), out.ctypes.data_as(C.c_void_p)), "rt_light_move_face")
    return out[0]


class PerlinClockC(C.Structure):
    """PerlinNoiseGeneratorWithTime's static count / stage / stime (algorithms.cpp:42-45)."""
    _fields_ = [("count", C.c_int32), ("stage", C.c_int32), ("stime", C.c_double)]


def perlin_time_changed(clock, prev_time, time, G):
    """PerlinNoiseGeneratorWithTime::timeChanged (algorithms.cpp:57-77): G is updated in place."""
    assert G.dtype == np.float64 and G.flags.c_contiguous
    _chk(lib().rt_perlin_time_changed(C.byref(clock), float(prev_time), float(time), len(G), _p(G)), "rt_perlin_time_changed")


def perlin_tables(seed=12325, n=256):
    """PerlinNoiseGenerator's tables (algorithms.cpp:4-19; the reference's SEED and size by default)."""
    P = np.zeros(n, np.int32); G = np.zeros((n, 3), np.float64)
    _chk(lib().rt_perlin_tables(seed, n, _p(P), _p(G)), "rt_perlin_tables")
    return P, G


def perlin_noise(x, y, z, scale, P, G):
    P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
    return lib().rt_perlin_noise(x, y, z, scale, _p(P), _p(G), len(P))


class AviWriter:
    """AviSaver (AviSaver.cpp:9-75) without a codec: uncompressed frames in an AVI container.
    Frames are (h, w, 3) uint8 arrays in the layout of resolve_bgr8 (row 0 = bottom, B,G,R)."""

    def __init__(self, path, width, height, fps=15.0, mjpg_quality=0):
        """mjpg_quality 0: uncompressed frames; 1..100: Motion-JPEG (the fourcc AviSaver asks for)."""
        self._h = _vp()
        self.shape = (height, width, 3)
        if mjpg_quality:
            _chk(lib().rt_avi_open_mjpg(os.fsencode(path), width, height, float(fps), int(mjpg_quality), C.byref(self._h)), "rt_avi_open_mjpg")
        else:
            _chk(lib().rt_avi_open(os.fsencode(path), width, height, float(fps), C.byref(self._h)), "rt_avi_open")

    def add(self, bgr):
        a = np.ascontiguousarray(bgr, np.uint8)
        assert a.shape == self.shape
        _chk(lib().rt_avi_add_frame(self._h, _p(a)), "rt_avi_add_frame")

    def close(self):
        if self._h:
            h, self._h = self._h, None
            _chk(lib().rt_avi_close(h), "rt_avi_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def save_bmp(bgr, filename):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    _chk(lib().rt_save_bmp(_p(bgr), w, h, os.fsencode(filename)), "rt_save_bmp")


TONE_NONE, TONE_SIGMOID, TONE_SQRT = 0, 1, 2


def resolve_bgr8_host(radiance_sum, spp, tone=TONE_NONE):
    a = np.ascontiguousarray(radiance_sum, np.float64)
    h, w, _ = a.shape
    out = np.empty((h, w, 3), np.uint8)
    _chk(lib().rt_resolve_tone_bgr8_host(_p(a), w, h, int(spp), int(tone), _p(out)), "rt_resolve_tone_bgr8_host")
    return out


class HipScene:
    """Device-resident scene (octrees flattened and uploaded once)."""

    def __init__(self, desc):
        self.desc = desc
        d, keep = desc.to_c()
        h = _vp()
        _chk(lib().rt_scene_create(C.byref(d), C.byref(h)), "rt_scene_create")
        del keep
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def octree_stats(self, mesh=0):
        out = (_u64 * 6)()
        _chk(lib().rt_scene_octree_stats(self._h, mesh, out), "rt_scene_octree_stats")
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    # ---- host-buffer entry points (numpy in / numpy out)
    def trace_rays(self, rays, t0=0.0, t1=1e10):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        n = len(rays)
        prim = np.empty(n, np.uint32); t = np.empty(n, np.float64); nrm = np.empty((n, 3), np.float64)
        cnt = np.zeros(4, np.uint64)
        _chk(lib().rt_trace_rays_host(self._h, _p(rays), n, t0, t1, _p(prim), _p(t), _p(nrm), _p(cnt)), "rt_trace_rays_host")
        return prim, t, nrm, cnt

    def trace_primary(self, cam, w, h):
        prim = np.empty((h, w), np.uint32); t = np.empty((h, w), np.float64)
        cnt = np.zeros(4, np.uint64)
        _chk(lib().rt_trace_primary_host(self._h, C.byref(cam), w, h, _p(prim), _p(t), _p(cnt)), "rt_trace_primary_host")
        return prim, t, cnt

    def ray_color(self, rays, params, t0=0.0, t1=1e10, jumptime=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        out = np.zeros((len(rays), 3), np.float64)
        cnt = np.zeros(4, np.uint64)
        _chk(lib().rt_ray_color_host(self._h, _p(rays), len(rays), t0, t1, jumptime, C.byref(params), _p(out), _p(cnt)),
             "rt_ray_color_host")
        return out, cnt

    def render(self, cam, params, radiance_sum=None):
        if radiance_sum is None:
            radiance_sum = np.zeros((params.height, params.width, 3), np.float64)
        assert radiance_sum.shape == (params.height, params.width, 3) and radiance_sum.dtype == np.float64
        cnt = np.zeros(4, np.uint64)
        _chk(lib().rt_render_host(self._h, C.byref(cam), C.byref(params), _p(radiance_sum), _p(cnt)), "rt_render_host")
        return radiance_sum, cnt

    # ---- device-pointer entry points (raw addresses, e.g. torch tensors' data_ptr())
    def render_dev(self, cam, params, radiance_sum_ptr, counters_ptr=0, stream=0):
        _chk(lib().rt_render(self._h, C.byref(cam), C.byref(params), radiance_sum_ptr, counters_ptr or None, stream or None),
             "rt_render")

    def trace_primary_dev(self, cam, w, h, prim_ptr, t_ptr=0, counters_ptr=0, stream=0):
        _chk(lib().rt_trace_primary(self._h, C.byref(cam), w, h, prim_ptr, t_ptr or None, counters_ptr or None, stream or None),
             "rt_trace_primary")

    def set_profiling(self, enable=True):
        _chk(lib().rt_scene_set_profiling(self._h, 1 if enable else 0), "rt_scene_set_profiling")

    def get_profile(self, reset=True):
        out = (C.c_double * 6)()
        _chk(lib().rt_scene_get_profile(self._h, out, 1 if reset else 0), "rt_scene_get_profile")
        return dict(zip(("trace_launches", "trace_ms", "shade_launches", "shade_ms", "generate_launches", "rounds"), out[:]))

    KERNEL_STAGES = ("k_trace", "k_trace_shadow", "k_sweep+k_trace_retry", "k_sweep_shadow+k_trace_shadow_retry")

    def get_kernel_stats(self, reset=True):
        """Per stage of the closest-hit search (profiling on): launches, ms, work counters, executed tests."""
        out = (C.c_double * 40)()
        _chk(lib().rt_scene_get_kernel_stats(self._h, out, 1 if reset else 0), "rt_scene_get_kernel_stats")
        names = ("launches", "ms", "rays", "inner", "leaf", "prim", "box_tests", "fp64_tests")
        return {st: dict(zip(names, out[10 * i:10 * i + 8])) for i, st in enumerate(self.KERNEL_STAGES)}

    def get_ray_flow(self, reset=True):
        out = (C.c_double * 6)()
        _chk(lib().rt_scene_get_ray_flow(self._h, out, 1 if reset else 0), "rt_scene_get_ray_flow")
        return dict(zip(("first_rays", "shaded_in", "bounce_class0", "bounce_class1", "shadow_class0", "shadow_class1"), out[:]))

    def set_workspace_limit(self, nbytes):
        """Upper bound on the device memory of the pending-ray stacks (0 = default policy)."""
        _chk(lib().rt_scene_set_workspace_limit(self._h, int(nbytes)), "rt_scene_set_workspace_limit")

    def workspace_bytes(self):
        return int(lib().rt_scene_workspace_bytes(self._h))

    def implicit_stats(self, kind):
        out = (C.c_uint64 * 6)()
        _chk(lib().rt_scene_implicit_stats(self._h, kind, out), "rt_scene_implicit_stats")
        return dict(zip(("nodes", "leaves", "refs", "inner_refs", "max_depth", "lost"), [int(x) for x in out]))

    def implicit_boxes(self, kind):
        """The voxels of the scene's implicit surfaces of `kind`, in insertion order: (n, 6)."""
        n = lib().rt_scene_implicit_boxes(self._h, kind, None, 0)
        if n < 0:
            _chk(-1, "rt_scene_implicit_boxes")
        out = np.empty((n, 6), np.float64)
        lib().rt_scene_implicit_boxes(self._h, kind, _p(out), n)
        return out

    def set_perlin(self, P, G, stream=0):
        """Replace the Perlin tables of the uploaded scene (animated noise)."""
        P = np.ascontiguousarray(P, np.int32); G = np.ascontiguousarray(G, np.float64)
        _chk(lib().rt_scene_set_perlin(self._h, _p(P), _p(G), len(P), stream or None), "rt_scene_set_perlin")

    def set_light(self, index, light, stream=0):
        """Replace light `index` (a LIGHT_DT record) of the uploaded scene."""
        from .desc import LIGHT_DT
        rec = np.ascontiguousarray(np.asarray(light, LIGHT_DT).reshape(1))
        _chk(lib().rt_scene_set_light(self._h, C.c_uint64(index), rec.ctypes.data_as(C.c_void_p), stream or None), "rt_scene_set_light")

    def check(self, stream=0):
        _chk(lib().rt_scene_check(self._h, stream or None), "rt_scene_check")


def resolve_bgr8_dev(radiance_sum_ptr, w, h, spp, bgr_ptr, stream=0):
    _chk(lib().rt_resolve_bgr8(radiance_sum_ptr, w, h, int(spp), bgr_ptr, stream or None), "rt_resolve_bgr8")


REDUCE_RCCL, REDUCE_HOST = 0, 1


class Comm:
    """rt_comm: the RCCL communicator of the frame's one reduce.  Comm(devices=[...]) for members in
    this process; Comm(unique_id=..., nranks=, rank=, device=) for one process per GPU."""

    def __init__(self, devices=None, unique_id=None, nranks=0, rank=0, device=0):
        self._h = _vp()
        if devices is not None:
            arr = (_int * len(devices))(*devices)
            _chk(lib().rt_comm_create(len(devices), arr, C.byref(self._h)), "rt_comm_create")
        else:
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            _chk(lib().rt_comm_init_rank(buf, int(nranks), int(rank), int(device), C.byref(self._h)), "rt_comm_init_rank")

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _chk(lib().rt_comm_unique_id(buf), "rt_comm_unique_id")
        return bytes(buf)

    def size(self):
        return lib().rt_comm_size(self._h)

    def reduce_and_resolve(self, member, radiance_sum_ptr, w, h, spp, bgr_ptr=0, tone=TONE_NONE, stream=0):
        _chk(lib().rt_reduce_and_resolve(self._h, int(member), radiance_sum_ptr, w, h, int(spp), int(tone), bgr_ptr or None, stream or None),
             "rt_reduce_and_resolve")

    def close(self):
        if self._h:
            h, self._h = self._h, None
            lib().rt_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """rt_multi: one scene handle and one host thread per listed device, samples dealt to them."""

    def __init__(self, desc, devices, reduce_mode=REDUCE_RCCL):
        d, keep = desc.to_c()
        arr = (_int * len(devices))(*devices)
        self._h = _vp()
        _chk(lib().rt_multi_create(C.byref(d), arr, len(devices), int(reduce_mode), C.byref(self._h)), "rt_multi_create")
        del keep

    def render(self, cam, params, tone=TONE_NONE):
        """-> (radiance sums (h, w, 3), bytes (h, w, 3), counters)"""
        img = np.zeros((params.height, params.width, 3), np.float64)
        bgr = np.zeros((params.height, params.width, 3), np.uint8)
        cnt = np.zeros(4, np.uint64)
        _chk(lib().rt_multi_render(self._h, C.byref(cam), C.byref(params), int(tone), _p(img), _p(bgr), _p(cnt)), "rt_multi_render")
        return img, bgr, cnt

    def close(self):
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            lib().rt_multi_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
