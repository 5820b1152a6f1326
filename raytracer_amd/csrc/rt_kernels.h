// rt_kernels.h -- host-callable launchers of the gfx950 kernels in rt_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "rt_device.h"

namespace rt {

// The closest-hit kernels run persistent lanes that pull rays through `next`, a device word
// owned by the scene (zeroed by the launcher on `st`); num_cu sizes the resident grid.
hipError_t launch_trace_rays(const DevScene &S, int num_cu, uint32_t *next, const rt_ray *rays, uint64_t n, double t0, double t1,
                             uint32_t *prim_id, double *t, double *normal, uint64_t *counters, hipStream_t st);
hipError_t launch_trace_primary(const DevScene &S, int num_cu, uint32_t *next, const rt_camera &cam, int w, int h, uint32_t *prim_id,
                                double *t, uint64_t *counters, hipStream_t st);
// wavefront path-tracing stages (see rt_device.h)
hipError_t launch_generate(const DevScene &S, const WaveArgs &A, const RayDest &dst, uint64_t first, uint32_t count, hipStream_t st);
// ray_class: the class of the chunk's rays (see RayDest).  `ctrl` = the scene's control words ([4] and
// [5] are zeroed and used); `retry` = room for `count` ray indices: class-0 rays that reach an octree
// after all are listed there and traced again by the engine.
hipError_t launch_trace(const DevScene &S, int num_cu, uint32_t *ctrl, uint32_t *retry, const WaveArgs &A, const RayStack &q, uint64_t base, uint32_t count,
                        int ray_class, hipStream_t st);
// the shadow test of a chunk of a shadow stack; unblocked rays add their radiance to A.out
hipError_t launch_trace_shadow(const DevScene &S, int num_cu, uint32_t *ctrl, uint32_t *retry, const WaveArgs &A, const ShadowStack &q, uint64_t base,
                               uint32_t count, int ray_class, hipStream_t st);
hipError_t launch_shade(const DevScene &S, const WaveArgs &A, const RayStack &src, uint64_t base, uint32_t count,
                        const RayDest &dst, hipStream_t st);
// Address, on the CURRENT device, of the library's static workspace (RT_STATIC_SCRATCH_BYTES bytes,
// 256-byte aligned; every device holds its own copy from the moment the code object is loaded there
// until the process ends).  Looked up once per device; no allocation is made.
hipError_t static_scratch(void **ptr);
hipError_t launch_resolve(const double *sum, uint64_t npix, int spp, int tone, uint8_t *bgr, hipStream_t st);

} // namespace rt
