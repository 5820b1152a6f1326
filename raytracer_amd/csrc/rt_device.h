// rt_device.h -- data the host hands to the gfx950 kernels (device-resident after upload).
#pragma once
#include <cstdint>

#include "../../include/rt_hip.h"
#include "rt_octree.h"

namespace rt {

// One Group child as the kernels see it (objects are walked in Group order by every lane, so
// this table is read through the scalar path).
struct DevObject {
    int32_t  type;       // RT_OBJ_*
    int32_t  material;
    uint32_t root;       // mesh: index of the octree root in DevScene::nodes
    uint32_t tri_base;   // mesh: index of its first triangle record
    uint32_t prim_base;  // global primitive id of the object's first primitive
    uint32_t pad;
    double   p[4];
};

struct DevScene {
    const DevObject   *objects;
    const rt_material *materials;
    const rt_light    *lights;
    const FlatNode    *nodes;      // all meshes' nodes, 64-byte records
    const uint32_t    *obj_index;  // per-node triangle lists (global triangle record indices)
    const double      *tris;       // triangle records: v0, e1, e2 (9 doubles)
    int32_t nobjects;
    int32_t nlights;
    double  background[3];
};

enum { RT_STACK_LEVELS = 16 };     // octree depth <= 15 -> at most 16 open nodes
enum { RT_BLOCK = 256 };

// Pending-ray record fields (one 8-byte word each), stored field-major and lane-interleaved:
// word(entry, field, lane) = ((entry * JOB_FIELDS + field) * nlanes + lane)
enum { JOB_EX = 0, JOB_EY, JOB_EZ, JOB_DX, JOB_DY, JOB_DZ, JOB_T1, JOB_WX, JOB_WY, JOB_WZ, JOB_KEY, JOB_META, JOB_FIELDS };
enum { JOB_RADIANCE = 0, JOB_SHADOW = 1 };

struct RenderArgs {
    rt_camera cam;
    int32_t  width, height;
    int32_t  n, m;
    int32_t  sample_begin, sample_end;
    int32_t  max_jump, rus_jump;
    uint64_t seed;
    int32_t  face_forward;
    int32_t  batch;            // 1: rays come from `rays` (getRayColor hook), 0: camera
    const rt_ray *rays;        // batch mode
    uint64_t nrays;
    double   t0, t1;           // batch mode window of the first ray
    int32_t  jumptime;         // batch mode initial depth
    int32_t  stack_cap;        // pending-ray entries per lane
    uint64_t *jobs;            // pending-ray storage
    uint32_t *next_work;       // global work counter
    double   *out;             // camera: radiance_sum [H][W][3] (+=) ; batch: out [n][3] (=)
    uint64_t *counters;        // nullable
    uint32_t *overflow;        // set to 1 if a lane ran out of pending-ray slots
};

} // namespace rt
