/*
 * rt_oracle.h -- CPU oracle for the octree-intersection + Monte-Carlo path-tracing path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference renderer's
 * algorithm, used as the checker in tests/, in __graft_entry__.smoke() and as the
 * cpu_baseline leg of bench.py.  Nothing in the product (raytracer_amd/, include/) may
 * link, import or call it.
 *
 * Parity status: the intersection, octree, camera, light-sampling and jitter-table pieces
 * are pinned against the unmodified reference compiled from /root/reference (oracle/_ref,
 * see oracle/Makefile and tests/test_oracle_vs_ref.py) and against the golden vectors in
 * tests/golden/ captured from it; so are the implicit surfaces end to end (voxel set, voxel
 * octree, BoxNode hits, HeartShape / SurfaceOfGenus2 normals).  The integrator (ray_color) is a restatement of
 * scene.cpp:194-395; scene.cpp itself cannot be compiled here (it includes a Windows-only
 * header through bmpgenerator.h), so ray_color is pinned only through its pinned parts and
 * analytic known-answer scenes: "parity unpinned" for whole-path radiance.
 *
 * The POD structs below have the same layout as the ones in include/rt_hip.h (the product
 * boundary) so one scene description can be handed to both sides; the two headers are kept
 * separate on purpose so that the oracle stands alone.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material type bits -- material.h:10-16 */
enum { ORC_MAT_NORMAL = 1, ORC_MAT_SPECULAR = 2, ORC_MAT_TRANSPARENT = 4, ORC_MAT_MONTECARLO = 8 };
/* object kinds -- Surface.h:50-56 plus the new triangle mesh */
enum { ORC_OBJ_SPHERE = 1, ORC_OBJ_GROUND = 2, ORC_OBJ_WALL_Z = 3, ORC_OBJ_WALL_X = 4, ORC_OBJ_MESH = 5,
       ORC_OBJ_HEART = 6,    /* HeartShape (Surface.h:151-174): p = centre, scale */
       ORC_OBJ_GENUS2 = 7 }; /* ImplicitSurface<SurfaceOfGenus2Function> (Surface.h:121-147) */
/* NormalCheckMaterial / PositionCheckMaterial (material.h:70-92): what getAmbient returns */
enum { ORC_CHECK_NONE = 0, ORC_CHECK_NORMAL = 1, ORC_CHECK_POSITION = 2 };
/* PerlinNoiseNormalMaterial / PerlinNoiseNormalMaterial_Blood (material.h:95-164) */
enum { ORC_PERLIN_NONE = 0, ORC_PERLIN_NORMAL = 1, ORC_PERLIN_BLOOD = 2 };
enum { ORC_LIGHT_SPHERE = 1, ORC_LIGHT_FACE = 2, ORC_LIGHT_POINT = 3 };

#define ORC_MISS 0xFFFFFFFFu

typedef struct {
    int32_t type;      /* ORC_OBJ_* */
    int32_t material;  /* index into materials */
    int32_t mesh;      /* index into meshes (ORC_OBJ_MESH only) */
    int32_t reserved;
    double  p[4];      /* sphere: cx,cy,cz,r ; planes: p[0] = plane coordinate */
} orc_object;

typedef struct {
    int32_t type;      /* ORC_MAT_* bit flags */
    int32_t check;     /* ORC_CHECK_* */
    double  emission[3];
    double  color[3];  /* Material::diffuse == getColor() */
    double  blur;
    double  atten[3];  /* a_ */
    double  nr;        /* nr_ */
    double  ambient[3], specular[3], mirror[3];   /* Phong branch only (material.h:59-62) */
    double  shine;
    double  check_box[6];  /* PositionCheckMaterial's box: lo, hi */
    int32_t perlin;        /* ORC_PERLIN_* */
    int32_t reserved;
    double  perlin_scale;  /* scale_ of the Perlin materials (material.h:99) */
} orc_material;

typedef struct {
    int32_t type;      /* ORC_LIGHT_* */
    int32_t reserved;
    double  pos[3];
    double  radius;    /* sphere light */
    double  normal[3]; /* face light: already normalised (light.cpp:51) */
    double  up[3];
    double  right[3];  /* up ^ normal, NOT normalised (light.cpp:51) */
    double  width, height;
    double  diffuse[3];
    double  ambient[3], specular[3];              /* Phong branch only (light.h:27-29) */
} orc_light;

typedef struct {
    const double *tris; /* ntris x 9 doubles: v0 v1 v2 */
    uint64_t ntris;
} orc_mesh;

typedef struct {
    const orc_object   *objects;   uint64_t nobjects;
    const orc_material *materials; uint64_t nmaterials;
    const orc_light    *lights;    uint64_t nlights;
    const orc_mesh     *meshes;    uint64_t nmeshes;
    double  background[3];
    double  tri_box_pad;        /* added to every triangle AABB on all six sides */
    int32_t octree_max_size;    /* leaf capacity, octree.h:50 (128) */
    int32_t octree_overlap_tol; /* octree.h:26 (3) */
    int32_t octree_max_depth;   /* depth guard, absent in the reference */
    int32_t reserved;
    double  implicit_min_size;  /* half edge of the implicit surfaces' voxels; 0 = 3e-3 (octree.h:66, Surface.h:123) */
    const int32_t *perlin_p;    /* the Perlin generator's permutation P_ (perlin_n entries) ...     */
    const double  *perlin_g;    /* ... and unit vectors G_ (3 x perlin_n); NULL = orc_perlin_tables(12325, 256) */
    int32_t perlin_n;
    int32_t reserved2;
} orc_scene_desc;

typedef struct {
    double e[3];
    double llc[3];   /* leftLowerCorner */
    double u[3], v[3];
    double width, height;
    double look_at[3];   /* lookAt_, used by ChangeFovCamera::timeChanged (camera.cpp:88-94) */
    double aperture;     /* DepthCamera (camera.cpp:50-72); 0 = the plain pinhole Camera */
} orc_camera;

typedef struct {
    int32_t  width, height;
    int32_t  n, m;                     /* jitter table n x m, spp = n*m */
    int32_t  sample_begin, sample_end; /* this caller's slice of the n*m samples */
    int32_t  max_jump, rus_jump;       /* scene.h:53-54 (40, 12) */
    uint64_t seed;
    int32_t  face_forward;             /* 0 = reference-exact raw normals */
    int32_t  reserved;
} orc_render_params;

typedef struct { double e[3]; double d[3]; } orc_ray;

typedef struct orc_scene orc_scene;

/* scene life cycle: builds one pointer octree per mesh with the reference insert policy */
orc_scene *orc_scene_create(const orc_scene_desc *desc);
void       orc_scene_destroy(orc_scene *s);
const char *orc_last_error(void);

/* out[0..5] = nodes, leaves, object refs, refs held at inner nodes, max depth, lost primitives */
int orc_octree_stats(const orc_scene *s, int mesh, uint64_t out[6]);

/* The voxel octree shared by the implicit surfaces of one kind (ORC_OBJ_HEART / ORC_OBJ_GENUS2),
 * ImplicitSurfaceOctree::build (octree.cpp:148-254): stats as above; boxes = the voxels in the
 * order they are inserted (the std::set order), 6 doubles each (lo, hi); returns their number
 * (writes at most cap of them) or -1 */
int orc_implicit_stats(const orc_scene *s, int kind, uint64_t out[6]);
int64_t orc_implicit_boxes(const orc_scene *s, int kind, double *boxes, uint64_t cap);
/* the implicit function and the analytic normal of a kind at a point of its own frame */
double orc_implicit_func(int kind, const double pos[3]);
void orc_implicit_normal(int kind, const double pos[3], double out[3]);

/* algorithm::perlinNoise (algorithms.h:56-87) over given tables */
double orc_perlin_noise(double x, double y, double z, double scale, const int32_t *P, const double *G, int n);
/* PerlinNoiseGenerator's constructor (algorithms.cpp:4-19) replayed with the generator the reference
 * was written for (MSVC rand(): s = s*214013 + 2531011, (s >> 16) & 0x7fff, RAND_MAX 0x7fff) */
void orc_perlin_tables(uint32_t seed, int n, int32_t *P, double *G);
/* PerlinNoiseGeneratorWithTime::timeChanged, algorithms.cpp:57-77; clock = {count, stage} + stime */
typedef struct { int32_t count, stage; double stime; } orc_perlin_clock;
void orc_perlin_time_changed(orc_perlin_clock *clock, double prev_time, double time, int n, double *G);
/* What a Perlin material returns at position pos for a raw normal: out = ambient, diffuse, specular,
 * adjusted normal (12 doubles) -- material.h:104-108,128-157, material.cpp:67-87 */
void orc_perlin_material(const orc_material *m, const int32_t *P, const double *G, int n,
                         const double pos[3], const double normal[3], double out[12]);

/* Camera::init, camera.cpp:37-48 */
void orc_camera_init(const double from[3], const double at[3], const double vup[3],
                     double vfov, double aspect, orc_camera *out);
/* Camera::getRay, camera.cpp:24-31 (s,t rounded to float as in the reference signature) */
void orc_camera_get_ray(const orc_camera *c, double s, double t, orc_ray *out);
/* DepthCamera::getRay (camera.cpp:62-72) with the lens point (x, y) of the unit disk given */
void orc_camera_get_ray_disk(const orc_camera *c, double s, double t, double x, double y, orc_ray *out);
/* ChangeFovCamera::timeChanged (camera.cpp:88-94) applied to `origin`; MovableFaceLight::getPosition's
 * move of the light (light.cpp:101-107) applied to `origin` */
void orc_camera_change_fov(const orc_camera *origin, double time, orc_camera *out);
void orc_light_move_face(const orc_light *origin, double time, orc_light *out);

/* closest hit through Group::hit; prim_id = global primitive id or ORC_MISS.
 * counters (nullable) += {group rays, inner-node visits, leaf visits, primitive tests} */
int orc_trace_rays(const orc_scene *s, const orc_ray *rays, uint64_t n, double t0, double t1,
                   uint32_t *prim_id, double *t, double *normal /* n x 3, nullable */,
                   uint64_t counters[4]);
/* one ray through each pixel centre, s = (x+.5)/W, t = (y+.5)/H, row-major y*W+x, y=0 bottom */
int orc_trace_primary(const orc_scene *s, const orc_camera *c, int w, int h,
                      uint32_t *prim_id, double *t, uint64_t counters[4], int threads);

/* batch rayColor: ray i is treated as pixel i, sample 0 of a 1x1 table.  out = n x 3 */
int orc_ray_color(const orc_scene *s, const orc_ray *rays, uint64_t n, double t0, double t1,
                  int jumptime, const orc_render_params *p, double *out, uint64_t counters[4]);

/* Scene::run over the whole frame: radiance_sum[y][x][3] += sum over the sample slice
 * (unclamped, not divided).  Pixels are scheduled dynamically over `threads` threads. */
int orc_render(const orc_scene *s, const orc_camera *c, const orc_render_params *p,
               double *radiance_sum, uint64_t counters[4], int threads);
/* render only pixel indices [pix_begin, pix_end) (row-major); used for bounded CPU timing */
int orc_render_range(const orc_scene *s, const orc_camera *c, const orc_render_params *p,
                     uint64_t pix_begin, uint64_t pix_end, double *radiance_sum,
                     uint64_t counters[4], int threads);
/* average, regularize (types.h:167-177), int(c*255) as B,G,R bytes (scene.cpp:183-185) */
void orc_resolve_bgr8(const double *radiance_sum, int w, int h, int spp, uint8_t *bgr);
/* the same with a tone reproduction operator (tonereproductionmethod.h:11-35) applied to the
 * averaged colour before it is clamped */
enum { ORC_TONE_NONE = 0, ORC_TONE_SIGMOID = 1, ORC_TONE_SQRT = 2 };
double orc_tone(int tone, double x);
void orc_resolve_tone_bgr8(const double *sum, int w, int h, int spp, int tone, uint8_t *bgr);

/* unit-level entry points for known-answer tests */
int    orc_hit_box(const double lo[3], const double hi[3], const orc_ray *r, double t0, double t1,
                   double *r0, double *r1);                         /* algorithms.h:223-236 */
int    orc_sphere_hit(const double c[3], double radius, const orc_ray *r, double t0, double t1,
                      double *t);                                   /* Surface.cpp:73-111 */
int    orc_plane_hit(int type, double coord, const orc_ray *r, double t0, double t1, double *t);
int    orc_triangle_hit(const double tri[9], const orc_ray *r, double t0, double t1, double *t);
int    orc_refract(const double d[3], const double n[3], double eta, double out[3]); /* scene.cpp:384-395 */
void   orc_child_bound(const double lo[3], const double hi[3], int child, double clo[3], double chi[3]); /* octree.h:82-120 */
/* light.cpp:138-190 / 76-86: returns sampled point, its normal and pdf for shading point x */
void   orc_light_sample(const orc_light *l, const double x[3], double u, double v,
                        double pos[3], double nl[3], double *pdf);
void   orc_jitter_pair(uint64_t seed, uint64_t pixel, uint32_t object_id, uint32_t k,
                       uint32_t n, uint32_t m, double out[2]);
uint64_t orc_hash2(uint64_t a, uint64_t b);
double   orc_u01(uint64_t x);
uint32_t orc_permute(uint32_t i, uint32_t l, uint32_t p);
uint64_t orc_path_root_key(uint64_t seed, uint64_t pixel, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
