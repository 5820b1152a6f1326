// rt_kernels.h -- host-callable launchers of the gfx950 kernels in rt_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "rt_device.h"

namespace rt {

hipError_t launch_trace_rays(const DevScene &S, const rt_ray *rays, uint64_t n, double t0, double t1,
                             uint32_t *prim_id, double *t, double *normal, uint64_t *counters, hipStream_t st);
hipError_t launch_trace_primary(const DevScene &S, const rt_camera &cam, int w, int h, uint32_t *prim_id, double *t,
                                uint64_t *counters, hipStream_t st);
// resident 256-thread blocks per CU of the path-tracing kernel (capped), for sizing its grid
int render_blocks_per_cu();
hipError_t launch_render(const DevScene &S, const RenderArgs &A, unsigned blocks, hipStream_t st);
hipError_t launch_resolve(const double *sum, uint64_t npix, int spp, uint8_t *bgr, hipStream_t st);

} // namespace rt
