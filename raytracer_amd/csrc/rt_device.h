// rt_device.h -- data the host hands to the gfx950 kernels (device-resident after upload).
#pragma once
#include <cstdint>

#include "../../include/rt_hip.h"
#include "rt_octree.h"

namespace rt {

// Bounding boxes of the records of a node's run, for the kernels' single-precision pre-test.
// They live in the node's own grid, one byte a coordinate: the node's box starts at RT_GRID_ORG on
// every axis and its longest edge is RT_GRID_EDGE units, which leaves half an edge of room on either
// side.  A box is two words, lo.x | lo.y << 8 | lo.z << 16 | hi.x << 24 , hi.y | hi.z << 8
// (rt_pretest.h); two boxes to a 16-byte load.
#define RT_GRID_EDGE 127u
#define RT_GRID_ORG 64u
// The boxes are grown by RT_BOX_GROW grid units before they are rounded outwards to bytes: 2e-3 for the
// single-precision arithmetic of the box test itself, the rest for the double-precision primitive test, whose
// accepted points lie within 64 ulp * reach / 1e-5 of the primitive (see tri_hit) -- 2.8e-2 units for a ray
// whose reach is RT_REACH_GRID units of the node's grid.  A ray that reaches farther is not pre-tested in
// that node (node_step).
#define RT_BOX_GROW 3.2e-2
#define RT_REACH_GRID 4.0e7f
#define RT_BOX_SPARE 64          // spare boxes after the last record: a chunk of 64 is read from any record on without clamping
#ifndef RT_SEG_RECS
#define RT_SEG_RECS 16
#endif
// RT_SEG_RECS = records per segment of a run: every segment has a box of its own (the union of its records')

// One Group child as the kernels see it.  The object phase of the closest-hit engine reads
// this table at wave-uniform addresses (scalar loads); a record carries everything that phase
// needs for the object so that no dependent look-up follows: the first 16 bytes say what to do,
// the rest is fetched only by the branch that uses it.
struct DevObject {
    int32_t  type;       // RT_OBJ_*
    uint32_t ntri;       // mesh whose whole octree is a single leaf: its triangle count, else 0
    uint32_t root;       // mesh: index of the octree root in DevScene::nodes
    uint32_t leaf_begin; // mesh with ntri != 0: index of the leaf's first record in DevScene::leaf_tris
    int32_t  material;
    uint32_t tri_base;   // mesh: index of its first triangle record
    uint32_t prim_base;  // global primitive id of the object's first primitive
    uint32_t next_root;  // root of the next object's octree if that object is a mesh with inner nodes, else RT_NO_CHAIN:
                         // a lane that finishes this object's octree goes straight on to that root
    double   p[4];       // sphere: centre, radius; plane: coordinate
    double   lo[3], hi[3];   // mesh with ntri != 0: the root's box
    double   pad1[2];
};
static_assert(sizeof(DevObject) == 128, "DevObject must be 128 bytes");
#define RT_NO_CHAIN 0xFFFFFFFFu

struct DevScene {
    const DevObject   *objects;
    const rt_material *materials;
    const rt_light    *lights;
    const FlatNode    *nodes;      // all meshes' nodes, 64-byte records
    const LeafTri     *leaf_tris;  // per-node triangle runs (80-byte records, id = global triangle index)
    const uint32_t    *leaf_box;   // one packed bounding box per leaf_tris record (same index, two words each)
    const uint32_t    *run_box;    // one packed box per node (two words, by node index): the union of the boxes of its run
    const uint32_t    *seg_box;    // one packed box per segment of a run, at (obj_begin / RT_SEG_RECS) + node index + segment number
    const double      *tri_normals; // normalize(e1 ^ e2) (3 doubles) by global triangle index
    const int32_t     *perlin_p;   // Perlin permutation and unit vectors (perlin_n, 3 x perlin_n); null without Perlin materials
    const double      *perlin_g;
    int32_t nobjects;
    int32_t nlights;
    int32_t regular;               // every node coordinate is 0 or of magnitude 1e-100 .. 1e100
    int32_t nbig;                  // octrees with inner nodes whose root boxes are listed below (at most RT_MAX_BIG)
    int32_t perlin_n;
    int32_t obj_weight;            // weight of a lane that waits for an object phase when a wave chooses its next step kind (8 = a triangle-step lane)
    double  background[3];
    float   big_lo[4][3], big_hi[4][3];
    int32_t shade_kinds;           // material features the objects use: 1 mirror lobe, 2 dielectric lobe, 4 Phong / Perlin / check (any of these: 7)
    int32_t any_chain;             // some object has a next_root (see DevObject)
};

enum { RT_MAX_BIG = 4 };
enum { RT_STACK_LEVELS = 16 };     // octree depth <= 15 -> at most 16 open nodes
enum { RT_BLOCK = 256 };

// ---------------------------------------------------------------------------------------------
// Path tracing runs as a wavefront pipeline over pools of pending rays ("ray stacks"):
//   k_generate  appends camera rays for the next (pixel, sample) work items,
//   k_trace     finds the closest hit of every ray of the chunk on top of the source stack,
//   k_shade     consumes the hits: adds radiance to the pixel sums and appends the follow-up
//               rays (bounce rays and shadow rays) to the destination stacks,
//   k_trace_shadow  finds out whether a shadow ray is blocked and, if not, delivers its radiance.
// A pending radiance ray (camera ray or bounce ray) is 11 eight-byte words stored field-major (one
// array per field), so that the lanes of a wave read and write consecutive addresses.  Its window
// is not stored: a first ray uses the caller's [t0, t1], every other one [ZERO, INFINITE]
// (scene.cpp:247,265,283,301).
enum { RAY_EX = 0, RAY_EY, RAY_EZ, RAY_DX, RAY_DY, RAY_DZ, RAY_WX, RAY_WY, RAY_WZ, RAY_KEY, RAY_META, RAY_FIELDS };
// RAY_META: bits 0-31 pixel (or ray index in batch mode), 32-55 sample index, 56-61 depth
// (the jumpTime argument of rayColor), 62 E flag.  A ray whose depth is the call's initial depth
// and whose E flag is set is a first ray.

struct RayStack {
    uint64_t *f;        // RAY_FIELDS arrays of `cap` words each: field k at f + k*cap
    double   *hit_t;    // cap
    uint64_t *hit_id;   // cap: obj << 32 | prim, or ~0 for a miss
    uint64_t  cap;
};

// A shadow ray (scene.cpp:235-238) is terminal: the closest-hit kernel itself adds the radiance it
// carries to its pixel when nothing lies in its window, so it needs no key, no depth, no hit
// record and never reaches the shading kernel.  10 eight-byte words (origin, direction, window
// end, radiance) and the pixel index, field-major.
enum { SH_EX = 0, SH_EY, SH_EZ, SH_DX, SH_DY, SH_DZ, SH_T1, SH_CX, SH_CY, SH_CZ, SH_FIELDS };
struct ShadowStack {
    uint64_t *f;        // SH_FIELDS arrays of `cap` words
    uint32_t *pix;      // cap
    uint64_t  cap;
};

// Follow-up rays are appended by class: class 1 rays reach the root box of an octree with inner
// nodes (they will run node and triangle steps), class 0 rays do not (they finish within object
// phases).  Tracing the classes in separate launches keeps the lanes of a wave in step; the class
// never changes what a ray returns.
struct RayDest {
    RayStack    q[2];       // radiance rays by class
    ShadowStack sq[2];      // shadow rays by class
    uint64_t    base[2];    // first free slot of each radiance stack
    uint64_t    sbase[2];   // first free slot of each shadow stack
    uint32_t   *count;      // count[0..1] radiance, count[2..3] shadow rays appended by the running kernel (zeroed by the host)
};

// The workspace of the entry point that allocates nothing (rt_ray_color_noalloc): a file-scope
// device array of this many bytes, 256 bytes of control words followed by the four ray stacks.
#define RT_STATIC_SCRATCH_BYTES ((64u << 20) + 256u)
#define RT_NOALLOC_MAX_RAYS (1u << 20)

// Counters per stage of the closest-hit search, kept while profiling is on: the four work counters, then
// box tests and fp64 primitive tests executed, two spare words.  A sweep stage includes the engine's
// launch over its retry list.
enum { RT_KSTAT_ENGINE = 0, RT_KSTAT_ENGINE_SHADOW = 1, RT_KSTAT_SWEEP = 2, RT_KSTAT_SWEEP_SHADOW = 3, RT_KSTAT_WORDS = 8 };
static_assert(RT_KSTAT_STAGES == 4 && RT_KSTAT_ROW == 2 + RT_KSTAT_WORDS, "rt_scene_get_kernel_stats rows");

struct WaveArgs {
    rt_camera cam;
    int32_t  width, height;
    int32_t  n, m;
    int32_t  sample_begin, sample_end;
    int32_t  max_jump, rus_jump;
    uint64_t seed;
    int32_t  face_forward;
    int32_t  batch;            // 1: first rays come from `rays` (getRayColor hook), 0: camera
    const rt_ray *rays;        // batch mode
    double   t0, t1;           // window of first rays
    int32_t  jumptime;         // initial depth of first rays
    int32_t  pad;
    double   *out;             // camera: radiance_sum [H][W][3] ; batch: out [n][3]   (atomic +=)
    uint64_t *counters;        // nullable: {rays, inner visits, leaf visits, primitive tests}
    uint64_t *kstats;          // nullable (profiling): RT_KSTAT_STAGES blocks of RT_KSTAT_WORDS counters, one per stage of the closest-hit search
};

} // namespace rt
