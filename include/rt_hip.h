/*
 * rt_hip.h -- C ABI of librt_hip.so: octree ray intersection + Monte-Carlo path tracing on
 * MI355X (gfx950) behind the reference renderer's scene model.
 *
 * Conventions (the reference's own, scene.cpp:54,62,191; scene.h:74-79): every entry point
 * returns int, 0 = ok, negative = error; rt_last_error() gives the text; nothing throws
 * across the boundary.  All POD inputs are borrowed for the duration of the call and copied.
 * Entry points without a _host suffix take DEVICE pointers for bulk inputs/outputs and a HIP
 * stream (void*, NULL = default stream) and return after enqueueing; _host variants take
 * host pointers, allocate temporaries, synchronise and copy back (they are the drop-in for
 * code that reads Scene::colorArray_ on the host).
 *
 * A scene handle belongs to the device that was current when it was created and must not be
 * used from two threads at once (the reference clones its scene per thread for the same
 * reason, scene.cpp:150).  Upload is immutable: any number of trace/render calls may follow.
 */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material::Type bit flags -- material.h:10-16 */
enum { RT_MAT_NORMAL = 1, RT_MAT_SPECULAR = 2, RT_MAT_TRANSPARENT = 4, RT_MAT_MONTECARLO = 8 };
/* Node kinds -- Surface.h:50-56; RT_OBJ_MESH is the triangle mesh drawable this library adds;
 * RT_OBJ_HEART = HeartShape (Surface.h:151-174), RT_OBJ_GENUS2 = SurfaceOfGenus2
 * (ImplicitSurface<SurfaceOfGenus2Function>, Surface.h:121-147): implicit surfaces intersected
 * through an octree of voxels (ImplicitSurfaceOctree, octree.cpp:148-254), one tree per kind
 * shared by all objects of the kind (the static ptree of the reference classes). */
enum { RT_OBJ_SPHERE = 1, RT_OBJ_GROUND = 2, RT_OBJ_WALL_Z = 3, RT_OBJ_WALL_X = 4, RT_OBJ_MESH = 5,
       RT_OBJ_HEART = 6, RT_OBJ_GENUS2 = 7 };
/* NormalCheckMaterial / PositionCheckMaterial (material.h:70-92): getAmbient() returns |normal|,
 * or the hit's local position relative to check_box, instead of the stored ambient colour */
enum { RT_CHECK_NONE = 0, RT_CHECK_NORMAL = 1, RT_CHECK_POSITION = 2 };
/* PerlinNoiseNormalMaterial / PerlinNoiseNormalMaterial_Blood (material.h:95-164): Phong
 * materials whose colours follow Perlin noise at the hit position (the _Blood variant also mixes
 * in a vein colour and tilts the normal, material.cpp:67-87) */
enum { RT_PERLIN_NONE = 0, RT_PERLIN_NORMAL = 1, RT_PERLIN_BLOOD = 2 };
/* SphereLight / FaceLight / the point Light -- light.h:8-73 */
enum { RT_LIGHT_SPHERE = 1, RT_LIGHT_FACE = 2, RT_LIGHT_POINT = 3 };

#define RT_MISS 0xFFFFFFFFu

/* One child of the scene's Group, in Group::addChild order (Surface.cpp:154-158). */
typedef struct {
    int32_t type;      /* RT_OBJ_* */
    int32_t material;  /* Node::setMaterial, Surface.cpp:25-31 */
    int32_t mesh;      /* index into rt_scene_desc.meshes (RT_OBJ_MESH only) */
    int32_t reserved;
    double  p[4];      /* Sphere(center, radius): cx,cy,cz,r (Surface.h:98) ; Ground(y) / Wall_z(z) / Wall_x(x): p[0] ;
                          HeartShape / SurfaceOfGenus2 (center, scale): cx,cy,cz,scale */
} rt_object;

/* Plain Material -- material.h:55-66, ctors material.cpp:2-23.  With RT_MAT_MONTECARLO in
 * `type` the Monte-Carlo branch of Scene::rayColor shades the hit (scene.cpp:205-304), without
 * it the Phong / Whitted branch (scene.cpp:305-375), which reads the last four fields. */
typedef struct {
    int32_t type;
    int32_t check;       /* RT_CHECK_* */
    double  emission[3];
    double  color[3];    /* getColor() == getDiffuse() */
    double  blur;        /* getMirrorBlur() */
    double  atten[3];    /* getAttenuation() */
    double  nr;          /* getRefraCoef() */
    double  ambient[3];  /* getAmbient() */
    double  specular[3]; /* getSpecular() */
    double  mirror[3];   /* getMirror() */
    double  shine;       /* getShine() */
    double  check_box[6];/* PositionCheckMaterial(box): lower corner, upper corner */
    int32_t perlin;      /* RT_PERLIN_* */
    int32_t reserved;
    double  perlin_scale;/* scale_ (material.h:97-99) */
} rt_material;

/* SphereLight(pos, radius, emission) light.cpp:109-116 ; FaceLight(...) light.cpp:50-56 ;
 * Light(pos, ambient, diffuse, specular) light.cpp:8-11 (RT_LIGHT_POINT: no geometry, pos only).
 * ambient / specular are read by the Phong branch only (scene.cpp:308-320). */
typedef struct {
    int32_t type;
    int32_t reserved;
    double  pos[3];
    double  radius;
    double  normal[3];      /* normalised by the FaceLight ctor */
    double  up[3];
    double  right[3];       /* up ^ normal, not normalised */
    double  width, height;
    double  diffuse[3];     /* getDiffuse() */
    double  ambient[3];     /* getAmbient() */
    double  specular[3];    /* getSpecular() */
} rt_light;

typedef struct {
    const double *tris;     /* ntris x 9 doubles (v0 v1 v2), host memory */
    uint64_t ntris;
} rt_mesh;

/* What Scene::setRootNode / addLight / setClearColor assemble (scene.cpp:68-93). */
typedef struct {
    const rt_object   *objects;   uint64_t nobjects;
    const rt_material *materials; uint64_t nmaterials;
    const rt_light    *lights;    uint64_t nlights;
    const rt_mesh     *meshes;    uint64_t nmeshes;
    double  background[3];      /* backGround_ */
    double  tri_box_pad;        /* added on all six sides of a triangle's AABB (flat boxes are never inserted: algorithms.h:329-333) */
    int32_t octree_max_size;    /* Octree(maxSize = 128), octree.h:50 */
    int32_t octree_overlap_tol; /* OctreeNode::OverlapTolerance = 3, octree.h:26 */
    int32_t octree_max_depth;   /* depth guard (the reference has none), <= 15 */
    int32_t reserved;
    double  implicit_min_size;  /* half edge of an implicit surface's voxels; 0 = 3e-3, the value both reference
                                   surfaces use (octree.h:66, Surface.h:123) */
    const int32_t *perlin_p;    /* PerlinNoiseGenerator::P_ (permutation, perlin_n entries) and */
    const double  *perlin_g;    /* G_ (unit vectors, 3 x perlin_n); NULL = rt_perlin_tables(12325, 256, ..),
                                   the generator's own SEED and size (algorithms.h:91,99).  The animation of
                                   G_ by PerlinNoiseGeneratorWithTime::timeChanged is the caller's to apply. */
    int32_t perlin_n;
    int32_t reserved2;
} rt_scene_desc;

/* Camera members after Camera::init -- camera.cpp:37-48 */
typedef struct {
    double e[3];
    double llc[3];   /* leftLowerCorner */
    double u[3], v[3];
    double width, height;
    double look_at[3];  /* lookAt_ (ChangeFovCamera::timeChanged needs it, camera.cpp:88-94) */
    double aperture;    /* DepthCamera's lens radius factor (camera.cpp:50-72); 0 = pinhole Camera.
                           Set by the caller after rt_camera_init; used by rt_render only
                           (rt_trace_primary always takes the ray through the lens centre). */
} rt_camera;

/* Scene's private sampler / integrator constants, made settable -- scene.h:52-54 */
typedef struct {
    int32_t  width, height;            /* setPixels, scene.cpp:52-58 */
    int32_t  n, m;                     /* RTObject::setRandomSize(n, m): spp = n*m */
    int32_t  sample_begin, sample_end; /* slice of the n*m sample indices this call renders */
    int32_t  max_jump, rus_jump;       /* maxJump_ = 40, rusJump_ = 12 */
    uint64_t seed;                     /* replaces srand(1243), scene.cpp:22 */
    int32_t  face_forward;             /* 0 = raw normals as in scene.cpp:218,233 */
    int32_t  reserved;
} rt_render_params;

typedef struct { double e[3]; double d[3]; } rt_ray; /* types.h:180-184 */

typedef struct rt_scene rt_scene;

const char *rt_last_error(void);
/* Number of visible HIP devices, or a negative error. */
int rt_device_count(void);
/* Select the device used by this thread's later calls. */
int rt_set_device(int device);

/* Scene assembly + one-time upload.  Builds one octree per mesh on the host with the
 * reference policy (OctreeNode::insertObject / createChildren, octree.cpp:61-108), flattens
 * it into device-resident arrays and copies spheres, planes, materials and lights. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
int rt_scene_destroy(rt_scene *scene);
/* out[0..5] = nodes, leaves, object refs, refs at inner nodes, max depth, lost primitives */
int rt_scene_octree_stats(const rt_scene *scene, int mesh, uint64_t out[6]);
/* The voxel octree of the scene's implicit surfaces of one kind (RT_OBJ_HEART / RT_OBJ_GENUS2),
 * built by ImplicitSurfaceOctree::build's rule (octree.cpp:148-254): the same six numbers, and the
 * voxels in insertion order, 6 doubles each (lower, upper corner); returns the voxel count
 * (writing at most `cap` of them) or a negative error. */
int rt_scene_implicit_stats(const rt_scene *scene, int kind, uint64_t out[6]);
int64_t rt_scene_implicit_boxes(const rt_scene *scene, int kind, double *boxes, uint64_t cap);

/* PerlinNoiseGenerator::PerlinNoiseGenerator (algorithms.cpp:4-19) replayed on the host with the
 * generator it was written for (MSVC's rand(): s = s*214013 + 2531011, (s >> 16) & 0x7fff):
 * P[n], G[3n].  algorithm::perlinNoise (algorithms.h:56-87) over given tables, host arithmetic. */
int rt_perlin_tables(uint32_t seed, int n, int32_t *P, double *G);
double rt_perlin_noise(double x, double y, double z, double scale, const int32_t *P, const double *G, int n);
/* PerlinNoiseGeneratorWithTime::timeChanged (algorithms.cpp:57-77) under RTObject::setTime
 * (RTObject.h:110-117): the unit vectors G drift from frame to frame.  `clock` carries the
 * generator's static count / stage / stime between calls (start from zeros); prev_time is the time
 * of the previous call, as the generator's own getTime() would report it.  Host arithmetic. */
typedef struct { int32_t count, stage; double stime; } rt_perlin_clock;
int rt_perlin_time_changed(rt_perlin_clock *clock, double prev_time, double time, int n, double *G);
/* Replaces the Perlin tables of an uploaded scene (n must be the size it was created with). */
int rt_scene_set_perlin(rt_scene *scene, const int32_t *P, const double *G, int n, void *stream);

/* Camera::Camera -> init, camera.cpp:5-8,37-48 (host arithmetic) */
int rt_camera_init(const double look_from[3], const double look_at[3], const double vup[3],
                   double vfov, double aspect, rt_camera *out);
/* ChangeFovCamera::timeChanged(time), camera.cpp:88-94: the film of `origin` scaled by
 * sqrt(sin(2 pi time) * 1.5 + 1.6) + 0.3 around lookAt (host arithmetic; out may not alias origin) */
int rt_camera_change_fov(const rt_camera *origin, double time, rt_camera *out);
/* MovableFaceLight (light.cpp:87-107) at `time`: `origin` moved on its circle around the y axis */
int rt_light_move_face(const rt_light *origin, double time, rt_light *out);
/* Replaces light `index` of an uploaded scene (animated lights: RTObject::setTime cascade,
 * RTObject.h:110-117) without rebuilding anything else.  The light's geometry, if it has any
 * (SphereLight), is an object of the scene and does not move with it. */
int rt_scene_set_light(rt_scene *scene, uint64_t index, const rt_light *light, void *stream);

/* Closest hit for a batch of rays = group_->calHit(ray, t0, t1, &rec) (scene.cpp:202,
 * Surface.cpp:160-181, octree.cpp:285-376).  prim_id[i] = global primitive id (objects in
 * Group order; a mesh contributes one id per triangle) or RT_MISS; t, normal nullable.
 * counters (nullable, 4 x u64, device) += {rays, inner-node visits, leaf visits, primitive tests}. */
int rt_trace_rays(rt_scene *scene, const rt_ray *rays, uint64_t n, double t0, double t1,
                  uint32_t *prim_id, double *t, double *normal, uint64_t *counters, void *stream);
int rt_trace_rays_host(rt_scene *scene, const rt_ray *rays, uint64_t n, double t0, double t1,
                       uint32_t *prim_id, double *t, double *normal, uint64_t *counters);

/* Primary-ray hit / id buffer: Camera::getRay((x+.5)/W, (y+.5)/H) (camera.cpp:24-31, float
 * arguments) then the closest hit; row-major [y][x], y = 0 is the bottom row. */
int rt_trace_primary(rt_scene *scene, const rt_camera *cam, int width, int height,
                     uint32_t *prim_id, double *t, uint64_t *counters, void *stream);
int rt_trace_primary_host(rt_scene *scene, const rt_camera *cam, int width, int height,
                          uint32_t *prim_id, double *t, uint64_t *counters);

/* The batch hook the reference declares but never defines:
 *   void getRayColor(Scene*, std::vector<Ray>, double t0, double t1, int jumptime,
 *                    std::vector<Color>&)            (cudaRayTracer.h:4, scene.h:57)
 * out[i] = scene->rayColor(rays[i], t0, t1, jumptime) (scene.cpp:194-304); ray i draws the
 * sample stream of pixel i, sample 0 of a 1x1 table.  out = n x 3 doubles. */
int rt_ray_color(rt_scene *scene, const rt_ray *rays, uint64_t n, double t0, double t1,
                 int jumptime, const rt_render_params *params, double *out, uint64_t *counters,
                 void *stream);
int rt_ray_color_host(rt_scene *scene, const rt_ray *rays, uint64_t n, double t0, double t1,
                      int jumptime, const rt_render_params *params, double *out, uint64_t *counters);

/* rt_ray_color for callers that cannot allocate device memory (code that must not call hipMalloc:
 * tight serving loops, harnesses that forbid it).  It has no counterpart in the reference.  Same
 * arguments, results and counters as rt_ray_color (device pointers, returns when the work on
 * `stream` is complete), but the pending-ray stacks and control words live in a 64 MiB static
 * device array that every GPU the library runs on holds until the process ends, and the call makes
 * no hipMalloc, hipFree or allocating library call.  Limits: n <= 1 048 576 rays per call
 * (rejected beyond that).  All calls on one GPU share that GPU's workspace, whatever scene they
 * name: calls that overlap on one GPU race. */
int rt_ray_color_noalloc(rt_scene *scene, const rt_ray *rays, uint64_t n, double t0, double t1,
                         int jumptime, const rt_render_params *params, double *out, uint64_t *counters,
                         void *stream);

/* Scene::run / runParallelly (scene.cpp:95-172) for every pixel and the sample slice
 * [sample_begin, sample_end): radiance_sum[y][x][3] += sum of rayColor over the slice
 * (not divided, not clamped -- the clamp comes after averaging, scene.cpp:112-113, so
 * partial sums from several devices can be added first). */
int rt_render(rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
              double *radiance_sum, uint64_t *counters, void *stream);
/* rt_render and rt_ray_color run a multi-launch pipeline that reads one counter back per
 * round, so unlike the other device-pointer entry points they return when the work on
 * `stream` is complete. */
int rt_render_host(rt_scene *scene, const rt_camera *cam, const rt_render_params *params,
                   double *radiance_sum, uint64_t *counters);

/* Synchronises `stream` and reports a failure recorded by an earlier rt_render / rt_ray_color
 * launch on this scene (a lane ran out of pending-ray slots; the slot count is an upper bound
 * derived from the materials, so this is a should-not-happen check).  The _host variants call
 * it themselves.  No counterpart in the reference. */
int rt_scene_check(rt_scene *scene, void *stream);

/* Stage timing for measurement (no counterpart in the reference): when enabled, later
 * rt_render / rt_ray_color calls bracket every closest-hit and every shading launch with HIP
 * events on the caller's stream.  rt_scene_get_profile: out = {closest-hit launches, their
 * total ms, shading launches, their total ms, first-ray launches, rounds}; reset != 0 zeroes
 * the totals after reading. */
int rt_scene_set_profiling(rt_scene *scene, int enable);
int rt_scene_get_profile(rt_scene *scene, double out[6], int reset);
/* Per stage of the closest-hit search, while profiling is enabled (no counterpart in the reference).
 * Stages: 0 = rt::k_trace (bounce rays through the octree engine), 1 = rt::k_trace_shadow (shadow rays
 * through the engine), 2 = rt::k_sweep followed by rt::k_trace_retry (bounce rays expected to stay outside
 * the octrees), 3 = rt::k_sweep_shadow followed by rt::k_trace_shadow_retry.  out = RT_KSTAT_STAGES rows of
 * RT_KSTAT_ROW doubles: {launches, total ms, rays, inner-node visits, leaf visits, primitives covered (the
 * four work counters of rt_render), single-precision box tests executed, double-precision primitive tests
 * executed, 0, 0}.  "Covered" counts every primitive of every run the reference's walk opens; "executed"
 * counts the tests that were made for them (boxes cull the rest).  Synchronises the device. */
#define RT_KSTAT_STAGES 4
#define RT_KSTAT_ROW 10
int rt_scene_get_kernel_stats(rt_scene *scene, double *out, int reset);
/* Rays that went through the path-tracing pipeline since the last reset, counted by the host
 * driver (always, profiling enabled or not): out = {first rays generated, rays traced and handed
 * to shading, bounce rays appended that end inside object phases, bounce rays appended that reach
 * an octree with inner nodes, shadow rays appended of the one kind, of the other}.  No counterpart
 * in the reference. */
int rt_scene_get_ray_flow(rt_scene *scene, double out[6], int reset);
/* The device memory rt_render / rt_ray_color may take for their pending-ray stacks (one allocation
 * per scene handle, made by the first call that needs it, grown by later ones, freed by
 * rt_scene_destroy).  By default a call sizes its rounds to 16 Mi rays (RT_WAVEFRONT_RAYS in the
 * environment overrides) and never takes more than a quarter of the device memory that is free;
 * with a limit set, rounds shrink until the stacks fit `bytes` (0 = back to the default) and an
 * allocation made earlier that is larger than the limit is freed and made anew by the next call.
 * The smallest round is 65 536 rays: a limit below what that needs for the scene's materials and
 * lights makes the next rt_render / rt_ray_color fail (-1) instead of exceeding it.  The default
 * policy looks at the free memory of the moment, so several handles on one device that render at
 * the same time can each take a quarter of it: give them limits.  Smaller rounds change speed
 * only, never results.  rt_scene_workspace_bytes: what the handle holds now.
 * No counterpart in the reference. */
int rt_scene_set_workspace_limit(rt_scene *scene, uint64_t bytes);
int64_t rt_scene_workspace_bytes(const rt_scene *scene);

/* colorArray_ -> bytes: average, Vec4::regularize (types.h:167-177), int(c*255) stored B,G,R
 * (scene.cpp:183-185).  bgr = height x width x 3 bytes, bottom row first. */
int rt_resolve_bgr8(const double *radiance_sum, int width, int height, int spp, uint8_t *bgr,
                    void *stream);
int rt_resolve_bgr8_host(const double *radiance_sum, int width, int height, int spp, uint8_t *bgr);
/* The same with a tone reproduction operator applied to the averaged colour before the clamp:
 * SigmoidMethod / SqrtMethod::reproduction (tonereproductionmethod.h:11-35; the reference only
 * names them in a commented-out line, scene.cpp:109). */
enum { RT_TONE_NONE = 0, RT_TONE_SIGMOID = 1, RT_TONE_SQRT = 2 };
int rt_resolve_tone_bgr8(const double *radiance_sum, int width, int height, int spp, int tone, uint8_t *bgr, void *stream);
int rt_resolve_tone_bgr8_host(const double *radiance_sum, int width, int height, int spp, int tone, uint8_t *bgr);

/* ---- more than one GPU.  The reference's only parallelism is Scene::runParallelly(threadNum)
 * (scene.cpp:141-172: an OpenMP pixel queue over per-thread clones of the scene); here the sample
 * indices of every pixel are dealt to devices, each device renders its slice into a full-frame
 * buffer of un-normalised sums (rt_render), the buffers are added onto the root and the root
 * resolves the frame -- the clamp comes after the average (scene.cpp:112-113), so the sums must
 * meet unclamped.
 *
 * rt_comm: the RCCL communicator of that one exchange.  rt_comm_create: all members in this
 * process (ncclCommInitAll over `devices`, each listed once; member i = rank i).  For one process
 * per GPU: rank 0 calls rt_comm_unique_id, hands the 128 bytes to the others by any means, and
 * every rank calls rt_comm_init_rank (a communicator with one local member, member 0).
 * rt_reduce_and_resolve (every member must call it, each from its own thread or process):
 * radiance_sum (device, width x height x 3) is summed onto rank 0 in place with one reduce on
 * `stream`; on rank 0, if bgr (device) is given, the frame is then resolved as by
 * rt_resolve_tone_bgr8 with the frame's total sample count `spp`.  RCCL is loaded when first
 * needed; without it these calls fail with an error. */
typedef struct rt_comm rt_comm;
int rt_comm_create(int ndev, const int *devices, rt_comm **out);
int rt_comm_unique_id(uint8_t id[128]);
int rt_comm_init_rank(const uint8_t id[128], int nranks, int rank, int device, rt_comm **out);
int rt_comm_size(const rt_comm *comm);
int rt_comm_destroy(rt_comm *comm);
int rt_reduce_and_resolve(rt_comm *comm, int member, double *radiance_sum, int width, int height, int spp,
                          int tone, uint8_t *bgr, void *stream);

/* Scene::runParallelly for one process that drives several devices: a scene handle per device
 * (the reference clones its scene per worker, scene.cpp:150) and, per frame, one host thread per
 * device that renders a contiguous, balanced slice of [sample_begin, sample_end) and joins the
 * reduce.  reduce_mode RT_REDUCE_RCCL: rt_reduce_and_resolve over a communicator of the devices
 * (each listed once).  RT_REDUCE_HOST: the buffers are added in host memory in member order and the
 * sum resolved on member 0's device -- slower, needs no RCCL, and lets one device appear several
 * times (several members sharing a card, e.g. to rehearse the partition on a one-GPU machine).
 * rt_multi_render: tone as in rt_resolve_tone_bgr8; radiance_sum_host (width x height x 3 sums of
 * the whole range) and bgr_host (width x height x 3 bytes) are host buffers, each nullable;
 * counters (nullable, host, 4 x u64) += the members' work counters.  rt_multi_scene: member i's
 * handle, e.g. for rt_scene_set_light between frames (every member must be given the same changes). */
enum { RT_REDUCE_RCCL = 0, RT_REDUCE_HOST = 1 };
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, const int *devices, int ndev, int reduce_mode, rt_multi **out);
int rt_multi_size(const rt_multi *multi);
rt_scene *rt_multi_scene(rt_multi *multi, int member);
int rt_multi_render(rt_multi *multi, const rt_camera *cam, const rt_render_params *params, int tone,
                    double *radiance_sum_host, uint8_t *bgr_host, uint64_t counters[4]);
int rt_multi_destroy(rt_multi *multi);

/* Snapshot (bmpgenerator.cpp:3-45) with fixed-width header fields: 14 + 40 byte headers,
 * 24 bpp, bottom-up rows.  Rows are padded to a multiple of four bytes as the format requires; the
 * reference writes no padding, which is the same bytes whenever 3 * width is a multiple of four
 * (every size its driver uses, main.cpp:13-23) and an unreadable file otherwise. */
int rt_save_bmp(const uint8_t *bgr, int width, int height, const char *filename);

/* Video file for the animation loop -- AviSaver::Open / AddImg / Save (AviSaver.cpp:9-75, used by
 * main.cpp:76-101).  rt_avi_open: the RIFF/AVI container with uncompressed 24-bit frames
 * (bottom-up B,G,R rows, which is the layout rt_resolve_bgr8 produces and AddImg receives).  rt_avi_close writes the index and
 * the final sizes; a file that was not closed is not playable. */
typedef struct rt_avi rt_avi;
int rt_avi_open(const char *path, int width, int height, double fps, rt_avi **out);
/* The same with Motion-JPEG frames, the fourcc AviSaver::Open asks OpenCV for (AviSaver.cpp:60):
 * every frame a baseline JPEG (ITU-T T.81 Annex K tables, YCbCr 4:4:4) at `quality` 1..100.  Any
 * MJPG decoder reads the file; its bytes are not those of OpenCV's encoder. */
int rt_avi_open_mjpg(const char *path, int width, int height, double fps, int quality, rt_avi **out);
int rt_avi_add_frame(rt_avi *avi, const uint8_t *bgr);
int rt_avi_close(rt_avi *avi);

#ifdef __cplusplus
}
#endif
#endif
