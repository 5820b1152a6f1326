"""MI355X-native octree ray intersection + Monte-Carlo path tracing behind the reference
renderer's scene API (WingerZeng/RayTracer).  Host side: `raytracer_amd.scene`; device side:
hand-written HIP kernels for gfx950 in `raytracer_amd/csrc`, reached through the C ABI
declared in `include/rt_hip.h`."""
__all__ = ["desc", "scene", "scenes", "hip"]
