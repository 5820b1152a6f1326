// rt_api.cpp -- the C ABI of librt_hip.so (declared in include/rt_hip.h): scene upload,
// workspace management and kernel launches.  No CPU rendering path exists in this library:
// every trace / render entry point runs the gfx950 kernels or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_device.h"
#include "rt_internal.h"
#include "rt_kernels.h"
#include "rt_octree.h"
#include "rt_pretest.h"

using namespace rt;

namespace {

thread_local std::string g_err;

int fail(const std::string &m) { g_err = m; return -1; }
int fail_hip(const char *what, hipError_t e) { g_err = std::string(what) + ": " + hipGetErrorString(e); return -2; }

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail_hip(#expr, _e); } while (0)

struct MeshInfo { uint64_t stats[6]; };

template <class T>
hipError_t upload(const std::vector<T> &v, void **dptr)
{
    size_t bytes = v.size() * sizeof(T);
    hipError_t e = hipMalloc(dptr, bytes > 256 ? bytes : 256);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(*dptr, v.data(), bytes, hipMemcpyHostToDevice);
    return e;
}

} // namespace

int rt::set_error(const std::string &message, int code) { g_err = message; return code; }

struct rt_scene {
    int device = 0;
    int num_cu = 0;
    DevScene dev{};
    std::vector<void *> allocs;
    std::vector<MeshInfo> meshes;
    MeshInfo implicit_info[2];                 // voxel octree per implicit kind
    std::vector<double> implicit_boxes[2];
    bool has_implicit[2] = {false, false};
    std::vector<rt_material> materials;
    int nlights = 0;
    int max_push = 1;             // most radiance (bounce) rays one shaded hit can append
    int max_shadow = 0;           // most shadow rays one shaded hit can append
    // path-tracing workspace (grown on demand, reused by later calls): one allocation, carved into the ray stacks
    void *ws_mem = nullptr;
    uint64_t ws_bytes = 0;
    uint64_t ws_limit = 0;        // rt_scene_set_workspace_limit (0 = the default policy)
    uint32_t *d_ctrl = nullptr;   // [0..3] rays appended per destination by the running kernel; [4] the path tracer's ray hand-out counter; [5] the length of its retry list; [6] the closest-hit entry points' hand-out counter
    uint32_t *h_ctrl = nullptr;   // pinned mirror of d_ctrl[0..3]
    int failed = 0;               // a render ran out of ray-stack space
    // optional stage timing
    int profiling = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double prof[6] = {0, 0, 0, 0, 0, 0}; // trace launches, trace ms, shade launches, shade ms, generate launches, rounds
    double kprof[RT_KSTAT_STAGES][2] = {};   // per stage of the closest-hit search: launches, ms
    uint64_t *d_kstats = nullptr;        // RT_KSTAT_STAGES x RT_KSTAT_WORDS device counters (profiling)
    double flow[6] = {0, 0, 0, 0, 0, 0}; // first rays generated, rays traced and shaded, bounce rays appended (class 0, 1), shadow rays appended (class 0, 1)
};

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip("hipGetDeviceCount", e);
    return n;
}

int rt_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int rt_camera_init(const double from[3], const double at[3], const double vup[3], double vfov, double aspect, rt_camera *out)
{
    if (!from || !at || !vup || !out) return fail("rt_camera_init: null argument");
    // Camera::init, camera.cpp:37-48 (PI is the reference's 7-digit constant, definitions.h:7)
    const double PI = 3.1415926;
    auto sub = [](const double *a, const double *b, double *r) { for (int k = 0; k < 3; k++) r[k] = a[k] - b[k]; };
    auto length = [](const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
    auto normalize = [&](double *a) { double s = 1 / length(a); for (int k = 0; k < 3; k++) a[k] = a[k] * s; };
    auto cross = [](const double *l, const double *r, double *o) {
        o[0] = l[1] * r[2] - r[1] * l[2]; o[1] = l[2] * r[0] - l[0] * r[2]; o[2] = l[0] * r[1] - r[0] * l[1];
    };
    double tmp[3], w[3], u[3], v[3];
    sub(at, from, tmp);
    double d = length(tmp);
    sub(from, at, w); normalize(w);
    cross(vup, w, u); normalize(u);
    cross(w, u, v); normalize(v);
    double height = std::tan(vfov * PI / 180 / 2) * d * 2;
    double width = height * aspect;
    for (int k = 0; k < 3; k++) {
        out->e[k] = from[k];
        out->u[k] = u[k];
        out->v[k] = v[k];
        out->llc[k] = (at[k] - v[k] * (height / 2)) - u[k] * (width / 2);
    }
    out->width = width;
    out->height = height;
    for (int k = 0; k < 3; k++) out->look_at[k] = at[k];
    out->aperture = 0;
    return 0;
}

int rt_perlin_tables(uint32_t seed, int n, int32_t *P, double *G)
{
    if (n <= 0 || !P || !G) return fail("rt_perlin_tables: bad argument");
    uint32_t st = seed;                                                     // srand(SEED), algorithms.cpp:7
    auto next = [&]() { st = st * 214013u + 2531011u; return (int)((st >> 16) & 0x7fff); };
    for (int i = 0; i < n; i++) {
        // genRandomUnitVec, algorithms.h:37-46: draws until the point lies OUTSIDE the unit ball, then normalises
        double a[3], l;
        do {
            for (int k = 0; k < 3; k++) a[k] = next() * 1.0 / 0x7fff * 2 - 1;
            l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        } while (l < 1);
        const double inv = 1 / l;
        for (int k = 0; k < 3; k++) G[3 * i + k] = a[k] * inv;
    }
    for (int i = 0; i < n; i++) P[i] = i;
    for (int i = n - 1; i >= 1; i--) {                                      // suffle, algorithms.h:48-54
        int pick = (int)(next() * 1.0 / 0x7fff * (i + 1));
        if (pick > i) pick = i;                                             // rand() == RAND_MAX would index past the end
        std::swap(P[pick], P[i]);
    }
    return 0;
}

double rt_perlin_noise(double x, double y, double z, double scale, const int32_t *P, const double *G, int n)
{
    if (!P || !G || n <= 0) return 0.0;
    auto fai = [&](int t) { const int m = t % n; return P[m >= 0 ? m : n + m]; };
    auto weight = [](double t) { const double at = std::fabs(t); return at < 1 ? 2 * at * at * at - 3 * at * at + 1 : 0.0; };
    x /= scale; y /= scale; z /= scale;
    double result = 0;
    for (int c = 0; c < 8; c++) {
        const int i = (c & 1) ? (int)std::floor(x) : (int)(std::floor(x) + 1);
        const int j = (c & 2) ? (int)std::floor(y) : (int)(std::floor(y) + 1);
        const int k = (c & 4) ? (int)std::floor(z) : (int)(std::floor(z) + 1);
        const double u = x - i, v = y - j, w = z - k;
        const double *g = G + 3 * fai(i + fai(j + fai(k)));
        result += weight(u) * weight(v) * weight(w) * (g[0] * u + g[1] * v + g[2] * w);
    }
    return result;
}

int rt_perlin_time_changed(rt_perlin_clock *clk, double prev_time, double time, int n, double *G)
{
    if (!clk || !G || n <= 0) return fail("rt_perlin_time_changed: bad argument");
    if (time == clk->stime) return 0;
    const double dt = time - prev_time;
    // every sixth change starts a new stage (the random vectors the reference draws at that point
    // are never read, algorithms.cpp:62-70)
    if (clk->count % 6 == 0) { clk->count = 1; clk->stage++; }
    else clk->count++;
    for (int i = 0; i < n; i++) {                                           // algorithms.cpp:71-74, in place and in order
        const double *o = G + 3 * ((i + clk->stage) % n);
        double v[3];
        for (int k = 0; k < 3; k++) v[k] = G[3 * i + k] + o[k] * dt * 5;
        const double inv = 1 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int k = 0; k < 3; k++) G[3 * i + k] = v[k] * inv;
    }
    clk->stime = time;
    return 0;
}

int rt_scene_set_perlin(rt_scene *s, const int32_t *P, const double *G, int n, void *stream)
{
    if (!s || !P || !G) return fail("rt_scene_set_perlin: null argument");
    if (n <= 0 || n != s->dev.perlin_n) return fail("rt_scene_set_perlin: the scene was created with tables of another size (or without Perlin materials)");
    for (int i = 0; i < n; i++) if (P[i] < 0 || P[i] >= n) return fail("rt_scene_set_perlin: P entry out of range");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync((void *)s->dev.perlin_p, P, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync((void *)s->dev.perlin_g, G, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int rt_camera_change_fov(const rt_camera *origin, double time, rt_camera *out)
{
    if (!origin || !out || origin == out) return fail("rt_camera_change_fov: bad argument");
    *out = *origin;
    const double scale = std::sqrt(std::sin(time * 2 * 3.1415926) * 1.5 + 1.6) + 0.3;     // camera.cpp:90
    out->height = origin->height * scale;
    out->width = origin->width * scale;
    for (int k = 0; k < 3; k++)
        out->llc[k] = (origin->look_at[k] - origin->v[k] * (out->height / 2)) - origin->u[k] * (out->width / 2);
    return 0;
}

int rt_light_move_face(const rt_light *origin, double time, rt_light *out)
{
    if (!origin || !out) return fail("rt_light_move_face: null argument");
    const double radius = std::sqrt(origin->pos[0] * origin->pos[0] + origin->pos[2] * origin->pos[2]);   // light.cpp:90
    rt_light l = *origin;
    l.pos[0] = std::sin(time * 2 * 3.1415926) * radius;                                   // light.cpp:104-105
    l.pos[2] = std::cos(time * 2 * 3.1415926) * radius;
    *out = l;
    return 0;
}

int rt_scene_set_light(rt_scene *s, uint64_t index, const rt_light *light, void *stream)
{
    if (!s || !light) return fail("rt_scene_set_light: null argument");
    if (index >= (uint64_t)s->nlights) return fail("rt_scene_set_light: light index out of range");
    if (light->type != RT_LIGHT_SPHERE && light->type != RT_LIGHT_FACE && light->type != RT_LIGHT_POINT)
        return fail("rt_scene_set_light: unknown light type");
    HIP_TRY(hipSetDevice(s->device));
    // the record is copied from pageable memory: synchronous with respect to the host buffer
    HIP_TRY(hipMemcpyAsync((void *)(s->dev.lights + index), light, sizeof(rt_light), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int rt_scene_destroy(rt_scene *s)
{
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->ws_mem) (void)hipFree(s->ws_mem);
    if (s->d_ctrl) (void)hipFree(s->d_ctrl);
    if (s->d_kstats) (void)hipFree(s->d_kstats);
    if (s->h_ctrl) (void)hipHostFree(s->h_ctrl);
    for (int k = 0; k < 5; k++) if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
    delete s;
    return 0;
}

// Everything rt_scene_create derives from a scene description on the host, ready to be copied to a
// device: the octrees (built once, whatever the number of devices), their records and boxes, the
// object table.
struct HostScene {
    std::vector<DevObject> objs;
    std::vector<rt_material> mats;
    std::vector<rt_light> lights;
    std::vector<FlatNode> nodes;
    std::vector<LeafTri> leaf_tris;
    std::vector<double> tris, tri_normals;
    std::vector<uint32_t> leaf_box, seg_box, run_box;
    std::vector<int32_t> perlin_p;
    std::vector<double> perlin_g;
    std::vector<MeshInfo> infos;
    MeshInfo implicit_info[2] = {};
    std::vector<double> implicit_boxes[2];
    bool has_implicit[2] = {false, false};
    uint64_t nobjects = 0, nlights = 0;
    double background[3] = {0, 0, 0};
};

static int assemble_scene(const rt_scene_desc *d, HostScene &H)
{
    if (!d) return fail("rt_scene_create: null argument");
    auto &nodes = H.nodes; auto &leaf_tris = H.leaf_tris; auto &tris = H.tris; auto &infos = H.infos;
    auto &perlin_p = H.perlin_p; auto &perlin_g = H.perlin_g; auto &objs = H.objs;
    auto &implicit_info = H.implicit_info; auto &implicit_boxes = H.implicit_boxes; auto &has_implicit = H.has_implicit;
    auto &tri_normals = H.tri_normals; auto &leaf_box = H.leaf_box; auto &seg_box = H.seg_box; auto &run_box = H.run_box;
    H.nobjects = d->nobjects; H.nlights = d->nlights;
    for (int k = 0; k < 3; k++) H.background[k] = d->background[k];
    if ((d->nobjects && !d->objects) || (d->nmaterials && !d->materials) || (d->nlights && !d->lights) || (d->nmeshes && !d->meshes))
        return fail("rt_scene_create: null array with non-zero count");
    if (d->nobjects > 1u << 20 || d->nlights > 1024) return fail("rt_scene_create: too many objects or lights");
    const int max_size = d->octree_max_size > 0 ? d->octree_max_size : 128;
    const int tol = d->octree_overlap_tol > 0 ? d->octree_overlap_tol : 3;
    const int max_depth = d->octree_max_depth > 0 ? d->octree_max_depth : 12;
    if (max_depth > 15) return fail("rt_scene_create: octree_max_depth must be <= 15");

    for (uint64_t i = 0; i < d->nmaterials; i++) {
        const rt_material &m = d->materials[i];
        if ((m.type & RT_MAT_TRANSPARENT) && !(m.nr > 0)) return fail("rt_scene_create: transparent material needs nr > 0");
    }
    bool need_perlin = false;
    for (uint64_t i = 0; i < d->nmaterials; i++) {
        const rt_material &m = d->materials[i];
        if (m.perlin < RT_PERLIN_NONE || m.perlin > RT_PERLIN_BLOOD) return fail("rt_scene_create: unknown Perlin material kind");
        if (m.perlin != RT_PERLIN_NONE) {
            if (!(m.perlin_scale != 0.0) || !std::isfinite(m.perlin_scale)) return fail("rt_scene_create: Perlin material needs a finite, non-zero scale");
            need_perlin = true;
        }
        if (m.check < RT_CHECK_NONE || m.check > RT_CHECK_POSITION) return fail("rt_scene_create: unknown check material kind");
    }
    if (need_perlin) {
        const bool given = d->perlin_p && d->perlin_g && d->perlin_n > 0;
        const int n = given ? d->perlin_n : 256;
        perlin_p.resize(n); perlin_g.resize(3 * (size_t)n);
        if (given) {
            std::memcpy(perlin_p.data(), d->perlin_p, n * sizeof(int32_t));
            std::memcpy(perlin_g.data(), d->perlin_g, 3 * (size_t)n * sizeof(double));
            for (int i = 0; i < n; i++) if (perlin_p[i] < 0 || perlin_p[i] >= n) return fail("rt_scene_create: perlin_p entry out of range");
        } else {
            rt_perlin_tables(12325, n, perlin_p.data(), perlin_g.data());
        }
    }
    for (uint64_t i = 0; i < d->nlights; i++) {
        int t = d->lights[i].type;
        if (t != RT_LIGHT_SPHERE && t != RT_LIGHT_FACE && t != RT_LIGHT_POINT) return fail("rt_scene_create: unknown light type");
    }

    // one octree per mesh, concatenated into scene-wide arrays
    std::vector<uint32_t> mesh_root(d->nmeshes), mesh_tri_base(d->nmeshes);
    std::vector<uint64_t> mesh_ntris(d->nmeshes);
    infos.assign(d->nmeshes, MeshInfo{});
    for (uint64_t mi = 0; mi < d->nmeshes; mi++) {
        const rt_mesh &m = d->meshes[mi];
        if (m.ntris && !m.tris) return fail("rt_scene_create: mesh with null triangle array");
        FlatTree ft;
        std::string err;
        if (!build_octree(m.tris, m.ntris, d->tri_box_pad, max_size, tol, max_depth, ft, err))
            return fail("rt_scene_create: mesh " + std::to_string(mi) + ": " + err);
        const uint64_t node_base = nodes.size(), idx_base = leaf_tris.size(), tri_base = tris.size() / 9;
        if (node_base + ft.nodes.size() >= 0xFFFFFFF0ull || idx_base + ft.leaf_tris.size() >= 0xFFFFFFF0ull || tri_base + m.ntris >= 0xFFFFFFF0ull)
            return fail("rt_scene_create: scene too large for 32-bit indices");
        mesh_root[mi] = (uint32_t)node_base;
        mesh_tri_base[mi] = (uint32_t)tri_base;
        mesh_ntris[mi] = m.ntris;
        for (FlatNode f : ft.nodes) {
            if (f.child_base) f.child_base += (uint32_t)node_base;
            f.obj_begin += (uint32_t)idx_base;
            nodes.push_back(f);
        }
        for (LeafTri t : ft.leaf_tris) { t.id += (uint32_t)tri_base; leaf_tris.push_back(t); }
        tris.insert(tris.end(), ft.tri_rec.begin(), ft.tri_rec.end());
        std::memcpy(infos[mi].stats, ft.stats, sizeof ft.stats);
    }

    // one voxel octree per kind of implicit surface present
    uint32_t implicit_root[2] = {0, 0};
    uint64_t implicit_count[2] = {0, 0};
    for (uint64_t i = 0; i < d->nobjects; i++) {
        const int k = d->objects[i].type - RT_OBJ_HEART;
        if (k < 0 || k > 1 || has_implicit[k]) continue;
        FlatTree ft;
        std::string err;
        const double min_size = d->implicit_min_size > 0 ? d->implicit_min_size : 3e-3;
        if (!build_implicit_octree(k, min_size, max_size, tol, max_depth, ft, implicit_boxes[k], err))
            return fail("rt_scene_create: object " + std::to_string(i) + ": " + err);
        const uint64_t node_base = nodes.size(), idx_base = leaf_tris.size();
        if (node_base + ft.nodes.size() >= 0xFFFFFFF0ull || idx_base + ft.leaf_tris.size() >= 0xFFFFFFF0ull)
            return fail("rt_scene_create: scene too large for 32-bit indices");
        implicit_root[k] = (uint32_t)node_base;
        implicit_count[k] = implicit_boxes[k].size() / 6;
        for (FlatNode f : ft.nodes) {
            if (f.child_base) f.child_base += (uint32_t)node_base;
            f.obj_begin += (uint32_t)idx_base;
            nodes.push_back(f);
        }
        leaf_tris.insert(leaf_tris.end(), ft.leaf_tris.begin(), ft.leaf_tris.end());
        std::memcpy(implicit_info[k].stats, ft.stats, sizeof ft.stats);
        has_implicit[k] = true;
    }
    if (tris.size() / 9 >= (uint64_t)LEAF_VOXEL) return fail("rt_scene_create: too many triangles");

    objs.assign(d->nobjects, DevObject{});
    uint64_t prim = 0;
    for (uint64_t i = 0; i < d->nobjects; i++) {
        const rt_object &o = d->objects[i];
        DevObject &q = objs[i];
        std::memset(&q, 0, sizeof q);
        if (o.type < RT_OBJ_SPHERE || o.type > RT_OBJ_GENUS2) return fail("rt_scene_create: object " + std::to_string(i) + " has an unknown type");
        if (o.material < 0 || (uint64_t)o.material >= d->nmaterials) return fail("rt_scene_create: object " + std::to_string(i) + " has no valid material");
        q.type = o.type; q.material = o.material;
        for (int k = 0; k < 4; k++) q.p[k] = o.p[k];
        q.prim_base = (uint32_t)prim;
        if (o.type == RT_OBJ_MESH) {
            if (o.mesh < 0 || (uint64_t)o.mesh >= d->nmeshes) return fail("rt_scene_create: object " + std::to_string(i) + " has a bad mesh index");
            q.root = mesh_root[o.mesh];
            q.tri_base = mesh_tri_base[o.mesh];
            const FlatNode &rn = nodes[q.root];
            q.ntri = (rn.child_base == 0) ? rn.obj_count : 0;
            q.leaf_begin = rn.obj_begin;
            for (int k = 0; k < 3; k++) { q.lo[k] = rn.lo[k]; q.hi[k] = rn.hi[k]; }
            prim += mesh_ntris[o.mesh];
        } else if (o.type == RT_OBJ_HEART || o.type == RT_OBJ_GENUS2) {
            if (!(o.p[3] != 0.0) || !std::isfinite(o.p[3])) return fail("rt_scene_create: object " + std::to_string(i) + ": implicit surface needs a finite, non-zero scale");
            q.root = implicit_root[o.type - RT_OBJ_HEART];
            prim += implicit_count[o.type - RT_OBJ_HEART];
        } else {
            prim += 1;
        }
        if (prim >= 0xFFFFFFFFull) return fail("rt_scene_create: more than 2^32-1 primitives");
    }

    for (uint64_t i = 0; i < d->nobjects; i++) {
        const bool chain = i + 1 < d->nobjects && objs[i].type == RT_OBJ_MESH && objs[i].ntri == 0 && objs[i + 1].type == RT_OBJ_MESH && objs[i + 1].ntri == 0;
        objs[i].next_root = chain ? objs[i + 1].root : RT_NO_CHAIN;
    }

    H.mats.assign(d->materials, d->materials + d->nmaterials);
    H.lights.assign(d->lights, d->lights + d->nlights);
    // geometric normals: Vec3::operator^ and normalize as the kernels' cross / norm evaluate them
    tri_normals.assign(tris.size() / 3, 0.0);
    for (size_t i = 0; i < tris.size() / 9; i++) {
        const double *l = &tris[9 * i + 3], *r = &tris[9 * i + 6];
        const double n[3] = {l[1] * r[2] - r[1] * l[2], l[2] * r[0] - l[0] * r[2], l[0] * r[1] - r[0] * l[1]};
        const double inv = 1 / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        for (int k = 0; k < 3; k++) tri_normals[3 * i + k] = n[k] * inv;
    }
    // One box per record (same index) and one per segment of RT_SEG_RECS records of a node's run: the
    // union of its records' boxes, found by the kernels at (obj_begin / RT_SEG_RECS) + node index + segment
    // number -- runs lie in node order, so these places are distinct and in order (a run of c records has at
    // most c / RT_SEG_RECS + 1 segments).  Spare boxes behind the last record: the kernels read the boxes of
    // a whole segment from any record on.
    // The records of a run are put in the Morton order of their boxes' centres first, so that a segment's
    // records lie together and its box is small.  A node's list holds its primitives by increasing index
    // (they are inserted in index order, and a node that splits hands its list on in order,
    // octree.cpp:61-99), so "the later of two equal hits" (octree.cpp:31) is the one with the larger index
    // and the kernels can settle ties by index whatever the order of the records.  The run of a mesh whose
    // whole octree is one leaf keeps its order: the object sweep walks it record by record.
    leaf_box.assign((leaf_tris.size() + RT_BOX_SPARE) * 2, 0u);
    seg_box.assign((leaf_tris.size() / RT_SEG_RECS + nodes.size() + 2) * 2, 0u);
    run_box.assign(nodes.size() * 2 + 2, 0u);                     // per node: the union of its run's boxes
    for (size_t i = 0; i < leaf_box.size() / 2; i++) store_whole_node_box(&leaf_box[2 * i]);
    for (size_t i = 0; i < seg_box.size() / 2; i++) store_whole_node_box(&seg_box[2 * i]);
    std::vector<bool> keep_order(nodes.size(), false);
    for (uint64_t mi = 0; mi < d->nmeshes; mi++)
        if (nodes[mesh_root[mi]].child_base == 0) keep_order[mesh_root[mi]] = true;
    struct RecBox { uint32_t w[2]; int lo[3], hi[3]; uint32_t key, id, src; };
    std::vector<RecBox> rb;
    std::vector<LeafTri> moved;
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        FlatNode &n = nodes[ni];
        const float scale = node_grid_scale(n);
        uint32_t bits;
        std::memcpy(&bits, &scale, 4);
        n.pad = bits | (n.pad & 1u);                  // grid scale, voxel flag in the lowest bit
        rb.resize(n.obj_count);
        bool sortable = !keep_order[ni];
        for (uint32_t j = 0; j < n.obj_count; j++) {
            const LeafTri &t = leaf_tris[n.obj_begin + j];
            RecBox &r = rb[j];
            const BoxD b = (t.id & LEAF_VOXEL) ? clipped_voxel_box(t, n) : clipped_triangle_box(t, n);
            pack_box(r.w, b, n, scale, r.lo, r.hi);
            r.id = t.id; r.src = j;
            r.key = 0xFFFFFFFFu;                          // nothing of the primitive inside the node: to the end
            if (r.lo[0] <= r.hi[0]) {
                r.key = 0;
                for (int k = 0; k < 3; k++) {
                    const uint32_t c2 = (uint32_t)(r.lo[k] + r.hi[k]);       // twice the centre: 9 bits
                    for (int bit = 0; bit < 9; bit++) r.key |= ((c2 >> bit) & 1u) << (3 * bit + k);
                }
            }
            if (j && !(rb[j - 1].id < r.id)) sortable = false;
        }
        if (sortable && n.obj_count > 1) {
            std::sort(rb.begin(), rb.end(), [](const RecBox &a, const RecBox &b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });
            moved.assign(leaf_tris.begin() + n.obj_begin, leaf_tris.begin() + n.obj_begin + n.obj_count);
            for (uint32_t j = 0; j < n.obj_count; j++) leaf_tris[n.obj_begin + j] = moved[rb[j].src];
        }
        const size_t seg0 = (size_t)(n.obj_begin / RT_SEG_RECS) + ni;
        int rlo[3] = {256, 256, 256}, rhi[3] = {-1, -1, -1};
        for (uint32_t j = 0; j < n.obj_count; j += RT_SEG_RECS) {
            int ulo[3] = {256, 256, 256}, uhi[3] = {-1, -1, -1};
            for (uint32_t i = j; i < n.obj_count && i < j + RT_SEG_RECS; i++) {
                leaf_box[((size_t)n.obj_begin + i) * 2] = rb[i].w[0];
                leaf_box[((size_t)n.obj_begin + i) * 2 + 1] = rb[i].w[1];
                for (int k = 0; k < 3; k++) { if (rb[i].lo[k] < ulo[k]) ulo[k] = rb[i].lo[k]; if (rb[i].hi[k] > uhi[k]) uhi[k] = rb[i].hi[k]; }
            }
            store_grid_box(&seg_box[(seg0 + j / RT_SEG_RECS) * 2], ulo, uhi);
            for (int k = 0; k < 3; k++) { if (ulo[k] < rlo[k]) rlo[k] = ulo[k]; if (uhi[k] > rhi[k]) rhi[k] = uhi[k]; }
        }
        store_grid_box(&run_box[ni * 2], rlo, rhi);
    }
    return 0;
}

// The scene on the current device: one allocation per array, the handle owns them all.
static int upload_scene(const HostScene &H, rt_scene **out)
{
    *out = nullptr;
    const auto &objs = H.objs; const auto &mats = H.mats; const auto &lights = H.lights; const auto &nodes = H.nodes;
    const auto &leaf_tris = H.leaf_tris; const auto &tri_normals = H.tri_normals; const auto &leaf_box = H.leaf_box;
    const auto &seg_box = H.seg_box; const auto &run_box = H.run_box; const auto &perlin_p = H.perlin_p; const auto &perlin_g = H.perlin_g;
    rt_scene *s = new (std::nothrow) rt_scene;
    if (!s) return fail("rt_scene_create: out of memory");
    hipError_t e = hipGetDevice(&s->device);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, s->device);
        if (e == hipSuccess) s->num_cu = prop.multiProcessorCount;
    }
    void *p_obj = nullptr, *p_mat = nullptr, *p_light = nullptr, *p_nodes = nullptr, *p_idx = nullptr, *p_tris = nullptr, *p_sph = nullptr;
    void *p_pp = nullptr, *p_pg = nullptr, *p_seg = nullptr, *p_run = nullptr;
    auto up = [&](auto &vec, void **dst) {
        if (e != hipSuccess) return;
        e = upload(vec, dst);
        if (*dst) s->allocs.push_back(*dst);
    };
    up(objs, &p_obj); up(mats, &p_mat); up(lights, &p_light); up(nodes, &p_nodes); up(leaf_tris, &p_idx); up(tri_normals, &p_tris); up(leaf_box, &p_sph); up(seg_box, &p_seg); up(run_box, &p_run);
    up(perlin_p, &p_pp); up(perlin_g, &p_pg);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_ctrl, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_kstats, RT_KSTAT_STAGES * RT_KSTAT_WORDS * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(s->d_kstats, 0, RT_KSTAT_STAGES * RT_KSTAT_WORDS * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_ctrl, 64, hipHostMallocDefault);
    if (e != hipSuccess) { rt_scene_destroy(s); return fail_hip("scene upload", e); }
    s->dev.objects = (const DevObject *)p_obj;
    s->dev.materials = (const rt_material *)p_mat;
    s->dev.lights = (const rt_light *)p_light;
    s->dev.nodes = (const FlatNode *)p_nodes;
    s->dev.leaf_tris = (const LeafTri *)p_idx;
    s->dev.leaf_box = (const uint32_t *)p_sph;
    s->dev.seg_box = (const uint32_t *)p_seg;
    s->dev.run_box = (const uint32_t *)p_run;
    s->dev.tri_normals = (const double *)p_tris;
    s->dev.perlin_p = (const int32_t *)p_pp;
    s->dev.perlin_g = (const double *)p_pg;
    s->dev.perlin_n = (int32_t)perlin_p.size();
    s->dev.nobjects = (int32_t)H.nobjects;
    s->dev.nlights = (int32_t)H.nlights;
    s->dev.regular = 1;
    s->dev.any_chain = 0;
    for (const DevObject &q : objs) if (q.next_root != RT_NO_CHAIN) s->dev.any_chain = 1;
    s->dev.nbig = 0;
    {
        // what one sweep over the objects costs, in box-test units: one per object, two per triangle
        // of a mesh that is tested inside the sweep.  Timed on scenes with 3 (best weight >= 24) and
        // 25 units (best weight 8-10).
        uint64_t units = 0;
        for (const DevObject &q : objs) units += 1 + 2 * (uint64_t)q.ntri;
        s->dev.obj_weight = units <= 4 ? 24 : (units >= 20 ? 8 : (int32_t)(28 - units));
    }
    for (const DevObject &q : objs)
        if ((q.type == RT_OBJ_MESH && q.ntri == 0) || q.type == RT_OBJ_HEART || q.type == RT_OBJ_GENUS2) {
            // an octree with inner nodes; past RT_MAX_BIG of them the last box grows to hold the rest
            const FlatNode &rn = nodes[q.root];
            const bool fresh = s->dev.nbig < RT_MAX_BIG;
            const int b = fresh ? s->dev.nbig++ : RT_MAX_BIG - 1;
            for (int k = 0; k < 3; k++) {
                double wlo = rn.lo[k], whi = rn.hi[k];
                if (q.type != RT_OBJ_MESH) {
                    // the voxel tree lives in the surface's own frame: world = centre + scale * local, a little wider
                    const double a = q.p[k] + q.p[3] * rn.lo[k], c = q.p[k] + q.p[3] * rn.hi[k];
                    const double m = 1e-6 * (std::fabs(a) + std::fabs(c) + std::fabs(q.p[k]));
                    wlo = std::fmin(a, c) - m; whi = std::fmax(a, c) + m;
                }
                const float lo = std::nextafter((float)wlo, -INFINITY), hi = std::nextafter((float)whi, INFINITY);
                s->dev.big_lo[b][k] = fresh ? lo : std::fmin(s->dev.big_lo[b][k], lo);
                s->dev.big_hi[b][k] = fresh ? hi : std::fmax(s->dev.big_hi[b][k], hi);
            }
        }
    // RT_CLASS_GUESS=0 (a test switch): no boxes to guess with, every ray is class 0 and takes the
    // sweep kernel's way, the rays that enter an octree being handed on to the engine
    if (const char *g = std::getenv("RT_CLASS_GUESS")) if (g[0] == '0' && g[1] == 0) s->dev.nbig = 0;
    for (const FlatNode &f : nodes)
        for (int k = 0; k < 3; k++) {
            const double a = std::fabs(f.lo[k]), b = std::fabs(f.hi[k]);
            if (!((a == 0.0 || (a >= 1e-100 && a <= 1e100)) && (b == 0.0 || (b >= 1e-100 && b <= 1e100)))) s->dev.regular = 0;
        }
    for (int k = 0; k < 3; k++) s->dev.background[k] = H.background[k];
    s->meshes = H.infos;
    for (int k = 0; k < 2; k++) {
        s->implicit_info[k] = H.implicit_info[k];
        s->implicit_boxes[k] = H.implicit_boxes[k];
        s->has_implicit[k] = H.has_implicit[k];
    }
    s->materials = mats;
    s->nlights = (int)H.nlights;
    s->dev.shade_kinds = 0;                           // over the materials that objects use
    for (const DevObject &q : objs) {
        const rt_material &m = mats[q.material];
        if (!(m.type & RT_MAT_MONTECARLO) || m.perlin != RT_PERLIN_NONE || m.check != RT_CHECK_NONE) s->dev.shade_kinds |= 7;
        if (m.type & RT_MAT_SPECULAR) s->dev.shade_kinds |= 1;
        if (m.type & RT_MAT_TRANSPARENT) s->dev.shade_kinds |= 2;
    }
    for (const DevObject &q : objs) {                 // over the materials that objects use
        const rt_material &m = mats[q.material];
        const int push = ((m.type & RT_MAT_NORMAL) && (m.type & RT_MAT_MONTECARLO) ? 1 : 0) + ((m.type & RT_MAT_SPECULAR) ? 1 : 0) + ((m.type & RT_MAT_TRANSPARENT) ? 2 : 0);
        if (push > s->max_push) s->max_push = push;
        if ((m.type & RT_MAT_NORMAL) && s->nlights > s->max_shadow) s->max_shadow = s->nlights;
    }
    *out = s;
    return 0;
}

int rt_scene_create(const rt_scene_desc *d, rt_scene **out)
{
    if (!d || !out) return fail("rt_scene_create: null argument");
    *out = nullptr;
    HostScene H;
    if (int rc = assemble_scene(d, H)) return rc;
    return upload_scene(H, out);
}

} // extern "C"

int rt::create_scenes(const rt_scene_desc *d, const int *devices, int n, rt_scene **out)
{
    if (!d || !devices || n <= 0 || !out) return fail("create_scenes: bad argument");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    HostScene H;
    if (int rc = assemble_scene(d, H)) return rc;
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++)
        th.emplace_back([&, i]() {
            hipError_t e = hipSetDevice(devices[i]);
            rcs[i] = e == hipSuccess ? upload_scene(H, &out[i]) : fail_hip("hipSetDevice", e);
            if (rcs[i]) errs[i] = g_err;            // the error text is per thread
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n; i++)
        if (rcs[i]) {
            for (int k = 0; k < n; k++) if (out[k]) { rt_scene_destroy(out[k]); out[k] = nullptr; }
            g_err = errs[i];
            return rcs[i];
        }
    return 0;
}

extern "C" {

int rt_scene_octree_stats(const rt_scene *s, int mesh, uint64_t out[6])
{
    if (!s || !out) return fail("rt_scene_octree_stats: null argument");
    if (mesh < 0 || (size_t)mesh >= s->meshes.size()) return fail("rt_scene_octree_stats: bad mesh index");
    std::memcpy(out, s->meshes[mesh].stats, 6 * sizeof(uint64_t));
    return 0;
}

int rt_scene_implicit_stats(const rt_scene *s, int kind, uint64_t out[6])
{
    if (!s || !out) return fail("rt_scene_implicit_stats: null argument");
    const int k = kind - RT_OBJ_HEART;
    if (k < 0 || k > 1 || !s->has_implicit[k]) return fail("rt_scene_implicit_stats: the scene has no implicit surface of that kind");
    std::memcpy(out, s->implicit_info[k].stats, 6 * sizeof(uint64_t));
    return 0;
}

int64_t rt_scene_implicit_boxes(const rt_scene *s, int kind, double *boxes, uint64_t cap)
{
    if (!s) return fail("rt_scene_implicit_boxes: null argument");
    const int k = kind - RT_OBJ_HEART;
    if (k < 0 || k > 1 || !s->has_implicit[k]) return fail("rt_scene_implicit_boxes: the scene has no implicit surface of that kind");
    const uint64_t n = s->implicit_boxes[k].size() / 6;
    if (boxes && cap) std::memcpy(boxes, s->implicit_boxes[k].data(), (size_t)std::min(n, cap) * 6 * sizeof(double));
    return (int64_t)n;
}

int rt_trace_rays(rt_scene *s, const rt_ray *rays, uint64_t n, double t0, double t1, uint32_t *prim_id, double *t,
                  double *normal, uint64_t *counters, void *stream)
{
    if (!s) return fail("rt_trace_rays: null scene");
    if (n && (!rays || !prim_id)) return fail("rt_trace_rays: null buffer");
    if (n >= 0xFFFFFFFFull) return fail("rt_trace_rays: batch too large");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_trace_rays(s->dev, s->num_cu, s->d_ctrl + 6, rays, n, t0, t1, prim_id, t, normal, counters, (hipStream_t)stream));
    return 0;
}

int rt_trace_primary(rt_scene *s, const rt_camera *cam, int w, int h, uint32_t *prim_id, double *t, uint64_t *counters, void *stream)
{
    if (!s || !cam) return fail("rt_trace_primary: null argument");
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || (uint64_t)w * (uint64_t)h >= 0xFFFF0000ull) return fail("rt_trace_primary: bad image size");
    if (!prim_id) return fail("rt_trace_primary: null buffer");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_trace_primary(s->dev, s->num_cu, s->d_ctrl + 6, *cam, w, h, prim_id, t, counters, (hipStream_t)stream));
    return 0;
}

// ---------------------------------------------------------------- wavefront driver
// Pending rays live on stacks, two per ray class (see RayDest).  Each round pops a chunk of at
// most `target` rays from the top of one stack (the most recently appended, i.e. deepest, rays
// of the class that has more of them), traces and shades it, and appends the follow-up rays:
// those of the chunk's own class to that class's other stack, which becomes its source, those
// of the other class on top of that class's source.  New camera rays are added while the
// fuller class holds fewer than `target` rays, which keeps the trace kernel's grid full until
// every sample has been issued ("path regeneration").  One 8-byte read-back after each
// appending step tells the host how many rays each class received.
static uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 10);
    return (end && *end == 0 && x > 0) ? (uint64_t)x : dflt;
}

// The device memory one run of the pipeline works in: four radiance stacks of `cap` slots
// ([class][side]: a class's follow-up rays go to its other side), two shadow stacks of `scap` slots,
// and the control words (see rt_scene::d_ctrl); `target` = rays per round.  With P = most bounce
// rays and L = most shadow rays per shaded hit: cap = target (P + 3), scap = target (L + 1) (a
// shadow stack is emptied as soon as it holds `target` rays, and one round adds at most target L).
struct WaveSpace {
    RayStack stack[2][2];
    ShadowStack sstack[2];
    uint32_t *retry;            // scap ray indices: the retry list of one launch (see launch_trace)
    uint32_t *d_ctrl;
    uint64_t cap, scap, target;
};

static uint64_t round256(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }
static uint64_t radiance_stack_bytes(uint64_t cap) { return round256(cap * (RAY_FIELDS + 2) * sizeof(uint64_t)); }
static uint64_t shadow_stack_bytes(uint64_t scap) { return round256(scap * SH_FIELDS * sizeof(uint64_t)) + round256(scap * sizeof(uint32_t)); }
static uint64_t workspace_bytes(uint64_t target, uint64_t P, uint64_t L)
{
    return 4 * radiance_stack_bytes(target * (P + 3)) + 2 * shadow_stack_bytes(target * (L + 1)) + round256(target * (L + 1) * sizeof(uint32_t));
}
// Lays the six stacks for `target` rays per round out in memory at `base` (256-byte aligned).
static void carve_workspace(void *base, uint64_t target, uint64_t P, uint64_t L, WaveSpace &ws)
{
    unsigned char *p = (unsigned char *)base;
    ws.target = target; ws.cap = target * (P + 3); ws.scap = target * (L + 1);
    for (int c = 0; c < 2; c++)
        for (int k = 0; k < 2; k++) {
            uint64_t *m = (uint64_t *)p;
            ws.stack[c][k].f = m;
            ws.stack[c][k].hit_t = (double *)(m + ws.cap * RAY_FIELDS);
            ws.stack[c][k].hit_id = m + ws.cap * (RAY_FIELDS + 1);
            ws.stack[c][k].cap = ws.cap;
            p += radiance_stack_bytes(ws.cap);
        }
    for (int c = 0; c < 2; c++) {
        ws.sstack[c].f = (uint64_t *)p;
        ws.sstack[c].pix = (uint32_t *)(p + round256(ws.scap * SH_FIELDS * sizeof(uint64_t)));
        ws.sstack[c].cap = ws.scap;
        p += shadow_stack_bytes(ws.scap);
    }
    ws.retry = (uint32_t *)p;       // a launch traces at most target bounce rays or scap shadow rays
}

// One allocation of at least `bytes`, or none: a failed allocation leaves the scene without a workspace.
static int ensure_workspace(rt_scene *s, uint64_t bytes, hipStream_t st)
{
    // an allocation made before a (smaller) limit was set is given back, not kept
    const bool over_limit = s->ws_limit && s->ws_bytes > s->ws_limit;
    if (s->ws_mem && s->ws_bytes >= bytes && !over_limit) return 0;
    HIP_TRY(hipStreamSynchronize(st));
    if (s->ws_mem) { (void)hipFree(s->ws_mem); s->ws_mem = nullptr; s->ws_bytes = 0; }
    void *mem = nullptr;
    const hipError_t e = hipMalloc(&mem, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail_hip("hipMalloc (ray stacks)", e); }
    s->ws_mem = mem; s->ws_bytes = bytes;
    return 0;
}

static int wavefront_loop(rt_scene *s, const WaveSpace &ws, WaveArgs &A, uint64_t total_items, hipStream_t st)
{
    const uint64_t max_push = (uint64_t)s->max_push, max_shadow = (uint64_t)s->max_shadow, cap = ws.cap, target = ws.target;
    uint32_t *const d_ctrl = ws.d_ctrl;
    s->failed = 0;              // a failure of an earlier call has been reported by rt_scene_check or by that call's own status
    A.kstats = s->profiling ? s->d_kstats : nullptr;
    uint64_t count[2][2] = {{0, 0}, {0, 0}};      // radiance rays pending, [class][side]
    uint64_t scount[2] = {0, 0};                  // shadow rays pending, [class]
    int cur[2] = {0, 0};                          // side that is the class's source
    uint64_t issued = 0;
    // how many rays each destination received from the kernel that has just been enqueued
    auto read_counts = [&](uint64_t n[4]) -> int {
        HIP_TRY(hipMemcpyAsync(s->h_ctrl, d_ctrl, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int k = 0; k < 4; k++) n[k] = s->h_ctrl[k];
        return 0;
    };
    // the shadow tests of everything on a shadow stack (nothing is appended by them, so nothing is read back)
    auto drain = [&](int cls) -> int {
        if (!scount[cls]) return 0;
        if (s->profiling) HIP_TRY(hipEventRecord(s->ev[3], st));
        HIP_TRY(launch_trace_shadow(s->dev, s->num_cu, d_ctrl, ws.retry, A, ws.sstack[cls], 0, (uint32_t)scount[cls], cls, st));
        if (s->profiling) {
            float a = 0;
            HIP_TRY(hipEventRecord(s->ev[4], st));
            HIP_TRY(hipEventSynchronize(s->ev[4]));
            HIP_TRY(hipEventElapsedTime(&a, s->ev[3], s->ev[4]));
            s->prof[0] += 1; s->prof[1] += a;
            const int stage = cls ? RT_KSTAT_ENGINE_SHADOW : RT_KSTAT_SWEEP_SHADOW;
            s->kprof[stage][0] += 1; s->kprof[stage][1] += a;
        }
        scount[cls] = 0;
        return 0;
    };
    for (;;) {
        for (int c = 0; c < 2; c++) if (count[c][cur[c]] == 0 && count[c][cur[c] ^ 1] != 0) cur[c] ^= 1;
        uint64_t have = count[0][cur[0]] > count[1][cur[1]] ? count[0][cur[0]] : count[1][cur[1]];
        // top up with first rays
        if (issued < total_items && have < target) {
            uint64_t m = target - have;
            if (m > total_items - issued) m = total_items - issued;
            for (int c = 0; c < 2; c++) if (m > cap - count[c][cur[c]]) m = cap - count[c][cur[c]];
            if (m) {
                RayDest D;
                for (int c = 0; c < 2; c++) { D.q[c] = ws.stack[c][cur[c]]; D.base[c] = count[c][cur[c]]; D.sq[c] = ws.sstack[c]; D.sbase[c] = scount[c]; }
                D.count = d_ctrl;
                HIP_TRY(hipMemsetAsync(d_ctrl, 0, 16, st));
                HIP_TRY(launch_generate(s->dev, A, D, issued, (uint32_t)m, st));
                uint64_t n[4];
                if (int rc = read_counts(n)) return rc;
                count[0][cur[0]] += n[0]; count[1][cur[1]] += n[1];
                s->prof[4] += 1;
                s->flow[0] += (double)m;
                issued += m;
            }
        }
        const int X = count[0][cur[0]] > count[1][cur[1]] ? 0 : 1, Y = X ^ 1;
        if (count[X][cur[X]] == 0) {
            if (issued < total_items) { s->failed = 1; return fail("path tracing ran out of ray-stack space (raise the workspace limit)"); }
            break;
        }
        uint64_t c = count[X][cur[X]] < target ? count[X][cur[X]] : target;
        const uint64_t roomX = (cap - count[X][cur[X] ^ 1]) / max_push, roomY = (cap - count[Y][cur[Y]]) / max_push;
        if (c > roomX) c = roomX;
        if (c > roomY) c = roomY;
        if (c == 0) { s->failed = 1; return fail("path tracing ran out of ray-stack space (raise the workspace limit)"); }
        // room for the shadow rays of this round
        for (int k = 0; k < 2; k++) if (scount[k] + c * max_shadow > ws.scap) if (int rc = drain(k)) return rc;
        const uint64_t base = count[X][cur[X]] - c;
        RayDest D;
        D.q[X] = ws.stack[X][cur[X] ^ 1]; D.base[X] = count[X][cur[X] ^ 1];
        D.q[Y] = ws.stack[Y][cur[Y]];     D.base[Y] = count[Y][cur[Y]];
        for (int k = 0; k < 2; k++) { D.sq[k] = ws.sstack[k]; D.sbase[k] = scount[k]; }
        D.count = d_ctrl;
        HIP_TRY(hipMemsetAsync(d_ctrl, 0, 16, st));
        if (s->profiling) HIP_TRY(hipEventRecord(s->ev[0], st));
        HIP_TRY(launch_trace(s->dev, s->num_cu, d_ctrl, ws.retry, A, ws.stack[X][cur[X]], base, (uint32_t)c, X, st));
        if (s->profiling) HIP_TRY(hipEventRecord(s->ev[1], st));
        HIP_TRY(launch_shade(s->dev, A, ws.stack[X][cur[X]], base, (uint32_t)c, D, st));
        if (s->profiling) HIP_TRY(hipEventRecord(s->ev[2], st));
        uint64_t n[4];
        if (int rc = read_counts(n)) return rc;
        if (s->profiling) {
            float a = 0, b = 0;
            HIP_TRY(hipEventElapsedTime(&a, s->ev[0], s->ev[1]));
            HIP_TRY(hipEventElapsedTime(&b, s->ev[1], s->ev[2]));
            s->prof[0] += 1; s->prof[1] += a; s->prof[2] += 1; s->prof[3] += b;
            const int stage = X ? RT_KSTAT_ENGINE : RT_KSTAT_SWEEP;
            s->kprof[stage][0] += 1; s->kprof[stage][1] += a;
        }
        s->prof[5] += 1;
        s->flow[1] += (double)c;
        for (int k = 0; k < 4; k++) s->flow[2 + k] += (double)n[k];
        count[X][cur[X]] -= c;
        count[X][cur[X] ^ 1] += n[X];
        count[Y][cur[Y]] += n[Y];
        if (n[X]) cur[X] ^= 1;
        for (int k = 0; k < 2; k++) {
            scount[k] += n[2 + k];
            if (scount[k] >= target) if (int rc = drain(k)) return rc;
        }
    }
    for (int k = 0; k < 2; k++) if (int rc = drain(k)) return rc;
    // the last shadow tests still read the shadow stacks: the call returns when they are done, so that
    // the caller may hand the workspace (or the static one, which every scene of the device shares)
    // to the next call on any stream
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// Default size of a round (rays traced per launch).  Larger rounds mean fewer launches and fuller
// persistent grids but cost workspace: measured on the 1024 x 1024 benchmark frame, 16 Mi rays a
// round are within a few per cent of 64 Mi at a quarter of the memory.
#define RT_DEFAULT_ROUND_RAYS (16ull << 20)

static int run_wavefront(rt_scene *s, WaveArgs &A, uint64_t total_items, hipStream_t st)
{
    if (A.max_jump < 0 || A.max_jump > 60) return fail("render: max_jump must be in [0,60]");
    if (total_items == 0) return 0;
    const uint64_t P = (uint64_t)s->max_push, L = (uint64_t)s->max_shadow;
    // Rays per round: RT_WAVEFRONT_RAYS if set, else RT_DEFAULT_ROUND_RAYS; never more workspace
    // than the scene's limit (rt_scene_set_workspace_limit) or, without one, a quarter of the
    // device memory that is free.
    uint64_t target = env_u64("RT_WAVEFRONT_RAYS", RT_DEFAULT_ROUND_RAYS);
    if (target > total_items * (P ? P : 1)) target = total_items * (P ? P : 1);
    if (target < RT_BLOCK) target = RT_BLOCK;
    uint64_t limit = s->ws_limit;
    if (limit == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        limit = (uint64_t)(0.25 * ((double)free_b + (double)s->ws_bytes));
    }
    while (target > (1u << 16) && (workspace_bytes(target, P, L) > limit || target * (P + 3) >= 0xFFFFFFF0ull || target * (L + 1) >= 0xFFFFFFF0ull)) target >>= 1;
    if (target * (P + 3) >= 0xFFFFFFF0ull || target * (L + 1) >= 0xFFFFFFF0ull) return fail("render: ray stack too large");
    if (s->ws_limit && workspace_bytes(target, P, L) > s->ws_limit)
        return fail("render: the workspace limit of " + std::to_string(s->ws_limit) + " bytes is below the smallest round (65536 rays need " +
                    std::to_string(workspace_bytes(target, P, L)) + " bytes for this scene's materials and lights)");
    // on a device with less memory than that fall back to smaller rounds rather than fail
    for (;;) {
        const int rc = ensure_workspace(s, workspace_bytes(target, P, L), st);
        if (rc == 0) break;
        if (target <= (1u << 16)) return rc;
        target >>= 1;
    }
    WaveSpace ws;
    carve_workspace(s->ws_mem, target, P, L, ws);
    ws.d_ctrl = s->d_ctrl;
    return wavefront_loop(s, ws, A, total_items, st);
}

static int fill_args(const rt_render_params *p, WaveArgs &A, bool batch)
{
    if (!p) return fail("render: null params");
    std::memset(&A, 0, sizeof A);
    A.max_jump = p->max_jump; A.rus_jump = p->rus_jump; A.seed = p->seed; A.face_forward = p->face_forward;
    if (batch) { A.n = A.m = 1; A.sample_begin = 0; A.sample_end = 1; return 0; }
    if (p->width <= 0 || p->height <= 0 || p->width > 65536 || p->height > 65536) return fail("render: bad image size");
    if (p->n <= 0 || p->m <= 0 || (int64_t)p->n * p->m > (1 << 24)) return fail("render: bad jitter table size");
    if (p->sample_begin < 0 || p->sample_end > p->n * p->m || p->sample_begin > p->sample_end) return fail("render: bad sample range");
    A.width = p->width; A.height = p->height; A.n = p->n; A.m = p->m;
    A.sample_begin = p->sample_begin; A.sample_end = p->sample_end;
    return 0;
}

int rt_ray_color(rt_scene *s, const rt_ray *rays, uint64_t n, double t0, double t1, int jumptime, const rt_render_params *p,
                 double *out, uint64_t *counters, void *stream)
{
    if (!s) return fail("rt_ray_color: null scene");
    if (n && (!rays || !out)) return fail("rt_ray_color: null buffer");
    if (n >= 0xFFFFFFFFull) return fail("rt_ray_color: batch too large");
    if (jumptime < 0 || jumptime > 60) return fail("rt_ray_color: jumptime must be in [0,60]");
    WaveArgs A;
    if (int rc = fill_args(p, A, true)) return rc;
    HIP_TRY(hipSetDevice(s->device));
    A.batch = 1; A.rays = rays; A.t0 = t0; A.t1 = t1; A.jumptime = jumptime;
    A.out = out; A.counters = counters;
    if (n) HIP_TRY(hipMemsetAsync(out, 0, n * 3 * sizeof(double), (hipStream_t)stream));
    return run_wavefront(s, A, n, (hipStream_t)stream);
}

// rt_ray_color with its ray stacks and control words in the library's static device workspace
// instead of allocated memory: no hipMalloc / hipFree and no allocating library call is made.
// Every call on a device uses that device's one workspace (the stacks' contents need no clearing:
// the host tracks how many slots of each are live; the control words are zeroed before every
// kernel that counts into them), so calls that overlap on one device race.
int rt_ray_color_noalloc(rt_scene *s, const rt_ray *rays, uint64_t n, double t0, double t1, int jumptime, const rt_render_params *p,
                         double *out, uint64_t *counters, void *stream)
{
    if (!s) return fail("rt_ray_color_noalloc: null scene");
    if (n && (!rays || !out)) return fail("rt_ray_color_noalloc: null buffer");
    if (n > RT_NOALLOC_MAX_RAYS) return fail("rt_ray_color_noalloc: at most 1048576 rays per call (the static workspace is sized for that)");
    if (jumptime < 0 || jumptime > 60) return fail("rt_ray_color_noalloc: jumptime must be in [0,60]");
    WaveArgs A;
    if (int rc = fill_args(p, A, true)) return rc;
    if (A.max_jump < 0 || A.max_jump > 60) return fail("render: max_jump must be in [0,60]");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    void *base = nullptr;
    HIP_TRY(static_scratch(&base));
    // 256 bytes of control words, then the six stacks for the largest round that fits
    const uint64_t P = (uint64_t)s->max_push, L = (uint64_t)s->max_shadow;
    const uint64_t room = (uint64_t)RT_STATIC_SCRATCH_BYTES - 256u;
    uint64_t target = room / (4 * (P + 3) * (RAY_FIELDS + 2) * sizeof(uint64_t) + 2 * (L + 1) * (SH_FIELDS * sizeof(uint64_t) + sizeof(uint32_t)));
    target &= ~(uint64_t)63u;
    while (target >= RT_BLOCK && workspace_bytes(target, P, L) > room) target -= 64;
    if (target < RT_BLOCK) return fail("rt_ray_color_noalloc: the scene's materials and lights append more rays per hit than the static workspace holds");
    if (target > n * (P ? P : 1)) target = n * (P ? P : 1);
    if (target < RT_BLOCK) target = RT_BLOCK;
    WaveSpace ws;
    carve_workspace((unsigned char *)base + 256, target, P, L, ws);
    ws.d_ctrl = (uint32_t *)base;
    A.batch = 1; A.rays = rays; A.t0 = t0; A.t1 = t1; A.jumptime = jumptime;
    A.out = out; A.counters = counters;
    HIP_TRY(hipMemsetAsync(out, 0, n * 3 * sizeof(double), (hipStream_t)stream));
    return wavefront_loop(s, ws, A, n, (hipStream_t)stream);
}

int rt_render(rt_scene *s, const rt_camera *cam, const rt_render_params *p, double *radiance_sum, uint64_t *counters, void *stream)
{
    if (!s || !cam) return fail("rt_render: null argument");
    if (!radiance_sum) return fail("rt_render: null buffer");
    WaveArgs A;
    if (int rc = fill_args(p, A, false)) return rc;
    HIP_TRY(hipSetDevice(s->device));
    A.cam = *cam; A.out = radiance_sum; A.counters = counters;
    A.t0 = 0; A.t1 = 1e10; A.jumptime = 0;                       // rayColor(ray, 0, INFINITE), scene.cpp:110
    const uint64_t npix = (uint64_t)A.width * (uint64_t)A.height;
    if (npix >= 0xFFFFFFFFull) return fail("rt_render: image too large");
    return run_wavefront(s, A, npix * (uint64_t)(A.sample_end - A.sample_begin), (hipStream_t)stream);
}

/* Synchronises `stream` and reports whether an earlier path-tracing call on this scene
 * failed for lack of ray-stack space. */
int rt_scene_check(rt_scene *s, void *stream)
{
    if (!s) return fail("rt_scene_check: null scene");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (s->failed) return fail("an earlier path-tracing call ran out of ray-stack space");
    return 0;
}

int rt_scene_set_profiling(rt_scene *s, int enable)
{
    if (!s) return fail("rt_scene_set_profiling: null scene");
    HIP_TRY(hipSetDevice(s->device));
    if (enable) for (int k = 0; k < 5; k++) if (!s->ev[k]) HIP_TRY(hipEventCreate(&s->ev[k]));
    s->profiling = enable ? 1 : 0;
    return 0;
}

int rt_scene_get_profile(rt_scene *s, double out[6], int reset)
{
    if (!s || !out) return fail("rt_scene_get_profile: null argument");
    for (int k = 0; k < 6; k++) out[k] = s->prof[k];
    if (reset) for (int k = 0; k < 6; k++) s->prof[k] = 0;
    return 0;
}

int rt_scene_get_kernel_stats(rt_scene *s, double *out, int reset)
{
    if (!s || !out) return fail("rt_scene_get_kernel_stats: null argument");
    HIP_TRY(hipSetDevice(s->device));
    uint64_t h[RT_KSTAT_STAGES * RT_KSTAT_WORDS];
    HIP_TRY(hipMemcpy(h, s->d_kstats, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < RT_KSTAT_STAGES; k++) {
        double *o = out + k * (2 + RT_KSTAT_WORDS);
        o[0] = s->kprof[k][0]; o[1] = s->kprof[k][1];
        for (int j = 0; j < RT_KSTAT_WORDS; j++) o[2 + j] = (double)h[k * RT_KSTAT_WORDS + j];
    }
    if (reset) {
        HIP_TRY(hipMemset(s->d_kstats, 0, sizeof h));
        for (int k = 0; k < RT_KSTAT_STAGES; k++) s->kprof[k][0] = s->kprof[k][1] = 0;
    }
    return 0;
}

int rt_scene_get_ray_flow(rt_scene *s, double out[6], int reset)
{
    if (!s || !out) return fail("rt_scene_get_ray_flow: null argument");
    for (int k = 0; k < 6; k++) out[k] = s->flow[k];
    if (reset) for (int k = 0; k < 6; k++) s->flow[k] = 0;
    return 0;
}

int rt_scene_set_workspace_limit(rt_scene *s, uint64_t bytes)
{
    if (!s) return fail("rt_scene_set_workspace_limit: null scene");
    s->ws_limit = bytes;
    return 0;
}

int64_t rt_scene_workspace_bytes(const rt_scene *s)
{
    if (!s) { fail("rt_scene_workspace_bytes: null scene"); return -1; }
    return (int64_t)s->ws_bytes;
}

int rt_resolve_bgr8(const double *radiance_sum, int w, int h, int spp, uint8_t *bgr, void *stream)
{
    return rt_resolve_tone_bgr8(radiance_sum, w, h, spp, RT_TONE_NONE, bgr, stream);
}

int rt_resolve_tone_bgr8(const double *radiance_sum, int w, int h, int spp, int tone, uint8_t *bgr, void *stream)
{
    if (!radiance_sum || !bgr) return fail("rt_resolve_bgr8: null buffer");
    if (w <= 0 || h <= 0 || spp <= 0) return fail("rt_resolve_bgr8: bad size");
    if (tone < RT_TONE_NONE || tone > RT_TONE_SQRT) return fail("rt_resolve_bgr8: unknown tone operator");
    HIP_TRY(launch_resolve(radiance_sum, (uint64_t)w * h, spp, tone, bgr, (hipStream_t)stream));
    return 0;
}

// ---------------------------------------------------------------- host-buffer variants
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
}

int rt_trace_rays_host(rt_scene *s, const rt_ray *rays, uint64_t n, double t0, double t1, uint32_t *prim_id, double *t,
                       double *normal, uint64_t *counters)
{
    if (!s) return fail("rt_trace_rays_host: null scene");
    if (n && (!rays || !prim_id)) return fail("rt_trace_rays_host: null buffer");
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dr, dp, dt, dn, dc;
    HIP_TRY(dr.alloc(n * sizeof(rt_ray))); HIP_TRY(dp.alloc(n * 4)); HIP_TRY(dt.alloc(n * 8)); HIP_TRY(dn.alloc(n * 24)); HIP_TRY(dc.alloc(32));
    if (n) HIP_TRY(hipMemcpy(dr.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dc.p, 0, 32));
    if (int rc = rt_trace_rays(s, (const rt_ray *)dr.p, n, t0, t1, (uint32_t *)dp.p, (double *)dt.p, (double *)dn.p, (uint64_t *)dc.p, nullptr)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (n) {
        HIP_TRY(hipMemcpy(prim_id, dp.p, n * 4, hipMemcpyDeviceToHost));
        if (t) HIP_TRY(hipMemcpy(t, dt.p, n * 8, hipMemcpyDeviceToHost));
        if (normal) HIP_TRY(hipMemcpy(normal, dn.p, n * 24, hipMemcpyDeviceToHost));
    }
    if (counters) { uint64_t c[4]; HIP_TRY(hipMemcpy(c, dc.p, 32, hipMemcpyDeviceToHost)); for (int k = 0; k < 4; k++) counters[k] += c[k]; }
    return 0;
}

int rt_trace_primary_host(rt_scene *s, const rt_camera *cam, int w, int h, uint32_t *prim_id, double *t, uint64_t *counters)
{
    if (!s || !cam || !prim_id) return fail("rt_trace_primary_host: null argument");
    if (w <= 0 || h <= 0) return fail("rt_trace_primary_host: bad image size");
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = (size_t)w * h;
    DevBuf dp, dt, dc;
    HIP_TRY(dp.alloc(n * 4)); HIP_TRY(dt.alloc(n * 8)); HIP_TRY(dc.alloc(32));
    HIP_TRY(hipMemset(dc.p, 0, 32));
    if (int rc = rt_trace_primary(s, cam, w, h, (uint32_t *)dp.p, (double *)dt.p, (uint64_t *)dc.p, nullptr)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(prim_id, dp.p, n * 4, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, dt.p, n * 8, hipMemcpyDeviceToHost));
    if (counters) { uint64_t c[4]; HIP_TRY(hipMemcpy(c, dc.p, 32, hipMemcpyDeviceToHost)); for (int k = 0; k < 4; k++) counters[k] += c[k]; }
    return 0;
}

int rt_ray_color_host(rt_scene *s, const rt_ray *rays, uint64_t n, double t0, double t1, int jumptime, const rt_render_params *p,
                      double *out, uint64_t *counters)
{
    if (!s) return fail("rt_ray_color_host: null scene");
    if (n && (!rays || !out)) return fail("rt_ray_color_host: null buffer");
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dr, dout, dc;
    HIP_TRY(dr.alloc(n * sizeof(rt_ray))); HIP_TRY(dout.alloc(n * 24)); HIP_TRY(dc.alloc(32));
    if (n) HIP_TRY(hipMemcpy(dr.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dc.p, 0, 32));
    if (int rc = rt_ray_color(s, (const rt_ray *)dr.p, n, t0, t1, jumptime, p, (double *)dout.p, (uint64_t *)dc.p, nullptr)) return rc;
    if (int rc = rt_scene_check(s, nullptr)) return rc;
    if (n) HIP_TRY(hipMemcpy(out, dout.p, n * 24, hipMemcpyDeviceToHost));
    if (counters) { uint64_t c[4]; HIP_TRY(hipMemcpy(c, dc.p, 32, hipMemcpyDeviceToHost)); for (int k = 0; k < 4; k++) counters[k] += c[k]; }
    return 0;
}

int rt_render_host(rt_scene *s, const rt_camera *cam, const rt_render_params *p, double *radiance_sum, uint64_t *counters)
{
    if (!s || !cam || !p || !radiance_sum) return fail("rt_render_host: null argument");
    if (p->width <= 0 || p->height <= 0) return fail("rt_render_host: bad image size");
    HIP_TRY(hipSetDevice(s->device));
    const size_t bytes = (size_t)p->width * p->height * 24;
    DevBuf dout, dc;
    HIP_TRY(dout.alloc(bytes)); HIP_TRY(dc.alloc(32));
    HIP_TRY(hipMemcpy(dout.p, radiance_sum, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dc.p, 0, 32));
    if (int rc = rt_render(s, cam, p, (double *)dout.p, (uint64_t *)dc.p, nullptr)) return rc;
    if (int rc = rt_scene_check(s, nullptr)) return rc;
    HIP_TRY(hipMemcpy(radiance_sum, dout.p, bytes, hipMemcpyDeviceToHost));
    if (counters) { uint64_t c[4]; HIP_TRY(hipMemcpy(c, dc.p, 32, hipMemcpyDeviceToHost)); for (int k = 0; k < 4; k++) counters[k] += c[k]; }
    return 0;
}

int rt_resolve_bgr8_host(const double *radiance_sum, int w, int h, int spp, uint8_t *bgr)
{
    return rt_resolve_tone_bgr8_host(radiance_sum, w, h, spp, RT_TONE_NONE, bgr);
}

int rt_resolve_tone_bgr8_host(const double *radiance_sum, int w, int h, int spp, int tone, uint8_t *bgr)
{
    if (!radiance_sum || !bgr) return fail("rt_resolve_bgr8_host: null buffer");
    if (w <= 0 || h <= 0 || spp <= 0) return fail("rt_resolve_bgr8_host: bad size");
    const size_t n = (size_t)w * h;
    DevBuf ds, db;
    HIP_TRY(ds.alloc(n * 24)); HIP_TRY(db.alloc(n * 3));
    HIP_TRY(hipMemcpy(ds.p, radiance_sum, n * 24, hipMemcpyHostToDevice));
    if (int rc = rt_resolve_tone_bgr8((const double *)ds.p, w, h, spp, tone, (uint8_t *)db.p, nullptr)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bgr, db.p, n * 3, hipMemcpyDeviceToHost));
    return 0;
}

// Snapshot, bmpgenerator.cpp:3-45, with fixed-width fields (the reference's `unsigned long`
// header members are 8 bytes on LP64 and would corrupt the file).  The reference writes rows of
// 3*width bytes back to back, which is a valid file only when that is a multiple of four (true of
// every size its driver uses, main.cpp:13-23); here every row is padded to a multiple of four bytes
// as the format requires, so the two agree byte for byte on those sizes and any other width still
// gives a readable file.
int rt_save_bmp(const uint8_t *bgr, int width, int height, const char *filename)
{
    if (!bgr || !filename) return fail("rt_save_bmp: null argument");
    if (width <= 0 || height <= 0) return fail("rt_save_bmp: bad size");
    const uint32_t row = (uint32_t)width * 3u, stride = (row + 3u) & ~3u;
    if ((uint64_t)stride * (uint64_t)height > 0xFFFFFF00ull) return fail("rt_save_bmp: image too large for the format");
    const uint32_t size = stride * (uint32_t)height;
    unsigned char hdr[54];
    std::memset(hdr, 0, sizeof hdr);
    auto put16 = [&](int off, uint16_t v) { hdr[off] = v & 0xff; hdr[off + 1] = v >> 8; };
    auto put32 = [&](int off, uint32_t v) { for (int k = 0; k < 4; k++) hdr[off + k] = (v >> (8 * k)) & 0xff; };
    put16(0, 0x4d42);            // bfType "BM"
    put32(2, size + 54);         // bfSize
    put32(10, 54);               // bfOffBits
    put32(14, 40);               // biSize
    put32(18, (uint32_t)width);
    put32(22, (uint32_t)height); // positive: bottom-up rows
    put16(26, 1);                // biPlanes
    put16(28, 24);               // biBitCount
    put32(34, size);             // biSizeImage
    FILE *fp = std::fopen(filename, "wb");
    if (!fp) return fail(std::string("rt_save_bmp: cannot open ") + filename);
    bool ok = std::fwrite(hdr, 1, 54, fp) == 54;
    if (stride == row) {
        ok = ok && std::fwrite(bgr, 1, size, fp) == size;
    } else {
        const unsigned char zeros[4] = {0, 0, 0, 0};
        for (int y = 0; ok && y < height; y++)
            ok = std::fwrite(bgr + (size_t)y * row, 1, row, fp) == row && std::fwrite(zeros, 1, stride - row, fp) == stride - row;
    }
    ok = (std::fclose(fp) == 0) && ok;
    return ok ? 0 : fail("rt_save_bmp: short write");
}

// ---------------------------------------------------------------- AVI container (host only)
} // extern "C"
struct rt_avi {
    FILE *f = nullptr;
    int w = 0, h = 0;
    uint32_t stride = 0, frames = 0;
    double fps = 15;
    int quality = 0;                       // 0: uncompressed frames; 1..100: Motion-JPEG at that quality
    std::vector<uint32_t> sizes;           // payload bytes of every frame chunk
    uint64_t movi_bytes = 4;               // 'movi' + chunks so far
    uint32_t max_frame = 0;
};
namespace {
void le32(FILE *f, uint32_t v) { unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; std::fwrite(b, 1, 4, f); }
void le16(FILE *f, uint16_t v) { unsigned char b[2] = {(unsigned char)v, (unsigned char)(v >> 8)}; std::fwrite(b, 1, 2, f); }
void tag(FILE *f, const char *t) { std::fwrite(t, 1, 4, f); }
// header with the frame count and sizes known so far; rewritten by rt_avi_close
void avi_header(rt_avi *a)
{
    FILE *f = a->f;
    const uint32_t n = a->frames, frame_bytes = a->quality ? a->max_frame : a->stride * (uint32_t)a->h;
    const uint32_t movi_size = (uint32_t)a->movi_bytes, idx_size = n * 16;
    const char *codec = a->quality ? "MJPG" : "DIB ";
    std::fseek(f, 0, SEEK_SET);
    tag(f, "RIFF"); le32(f, 4 + (8 + 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40))) + (8 + movi_size) + (8 + idx_size)); tag(f, "AVI ");
    tag(f, "LIST"); le32(f, 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40))); tag(f, "hdrl");
    tag(f, "avih"); le32(f, 56);
    le32(f, (uint32_t)(1e6 / a->fps + 0.5));            // microseconds per frame
    le32(f, (uint32_t)(frame_bytes * a->fps + 0.5));     // max bytes per second
    le32(f, 0); le32(f, 0x10);                           // padding granularity, flags: AVIF_HASINDEX
    le32(f, n); le32(f, 0); le32(f, 1); le32(f, frame_bytes);
    le32(f, (uint32_t)a->w); le32(f, (uint32_t)a->h);
    le32(f, 0); le32(f, 0); le32(f, 0); le32(f, 0);
    tag(f, "LIST"); le32(f, 4 + (8 + 56) + (8 + 40)); tag(f, "strl");
    tag(f, "strh"); le32(f, 56);
    tag(f, "vids"); tag(f, codec); le32(f, 0); le16(f, 0); le16(f, 0); le32(f, 0);
    le32(f, 1000); le32(f, (uint32_t)(a->fps * 1000 + 0.5));   // scale, rate: rate / scale = frames per second
    le32(f, 0); le32(f, n); le32(f, frame_bytes); le32(f, 0xFFFFFFFFu); le32(f, 0);
    le16(f, 0); le16(f, 0); le16(f, (uint16_t)a->w); le16(f, (uint16_t)a->h);
    tag(f, "strf"); le32(f, 40);
    le32(f, 40); le32(f, (uint32_t)a->w); le32(f, (uint32_t)a->h); le16(f, 1); le16(f, 24);
    if (a->quality) tag(f, "MJPG"); else le32(f, 0);     // biCompression; BI_RGB with a positive height is bottom-up
    le32(f, a->quality ? (uint32_t)a->w * a->h * 3u : frame_bytes); le32(f, 0); le32(f, 0); le32(f, 0); le32(f, 0);
    tag(f, "LIST"); le32(f, movi_size); tag(f, "movi");
}

// ---- baseline JPEG (ITU-T T.81: 8-bit, Huffman, the tables of Annex K, YCbCr 4:4:4, JFIF header)
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQLum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                           18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffCode { uint16_t code[256]; uint8_t len[256]; };
HuffCode make_codes(const uint8_t bits[16], const uint8_t *vals)
{
    HuffCode h;
    std::memset(&h, 0, sizeof h);
    uint16_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++) { h.code[vals[k]] = code++; h.len[vals[k]] = (uint8_t)l; k++; }
        code <<= 1;
    }
    return h;
}
struct BitSink {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t code, int len)
    {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);           // byte stuffing
            n -= 8;
        }
    }
    void flush() { if (n) put(0x7F, 8 - n); }           // pad with ones
};
// forward DCT of one 8x8 block (separable, direct form), quantise, entropy-code
void encode_block(const float *px, const float *q, const double (*C)[8], int &dc_pred, const HuffCode &dc, const HuffCode &ac, BitSink &bs)
{
    double tmp[64], co[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) { double s = 0; for (int x = 0; x < 8; x++) s += px[8 * y + x] * C[u][x]; tmp[8 * y + u] = s; }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) { double s = 0; for (int y = 0; y < 8; y++) s += tmp[8 * y + u] * C[v][y]; co[8 * v + u] = s; }
    int zz[64];
    for (int i = 0; i < 64; i++) zz[i] = (int)std::lround(co[kZigzag[i]] / q[kZigzag[i]]);
    auto category = [](int v) { int a = v < 0 ? -v : v, c = 0; while (a) { c++; a >>= 1; } return c; };
    const int diff = zz[0] - dc_pred;
    dc_pred = zz[0];
    int c = category(diff);
    bs.put(dc.code[c], dc.len[c]);
    if (c) bs.put((uint32_t)(diff < 0 ? diff - 1 : diff), c);
    int run = 0;
    for (int i = 1; i < 64; i++) {
        if (zz[i] == 0) { run++; continue; }
        while (run > 15) { bs.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
        c = category(zz[i]);
        bs.put(ac.code[(run << 4) | c], ac.len[(run << 4) | c]);
        bs.put((uint32_t)(zz[i] < 0 ? zz[i] - 1 : zz[i]), c);
        run = 0;
    }
    if (run) bs.put(ac.code[0], ac.len[0]);            // end of block
}
// bgr: h rows of w pixels, bottom row first (the layout of rt_resolve_bgr8) -> a JFIF stream, top row first
void encode_jpeg(const uint8_t *bgr, int w, int h, int quality, std::vector<uint8_t> &out)
{
    static const HuffCode dcl = make_codes(kDcLumBits, kDcVals), dcc = make_codes(kDcChrBits, kDcVals);
    static const HuffCode acl = make_codes(kAcLumBits, kAcLumVals), acc = make_codes(kAcChrBits, kAcChrVals);
    double C[8][8];
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) C[u][x] = (u ? 0.5 : 0.35355339059327373) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;   // the usual quality scaling of the Annex K tables
    uint8_t qt[2][64];
    float qf[2][64];
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 64; i++) {
            int v = ((t ? kQChr[i] : kQLum[i]) * scale + 50) / 100;
            v = v < 1 ? 1 : (v > 255 ? 255 : v);
            qt[t][i] = (uint8_t)v; qf[t][i] = (float)v;
        }
    out.clear();
    auto be16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    auto marker = [&](int m) { out.push_back(0xFF); out.push_back((uint8_t)m); };
    marker(0xD8);
    marker(0xE0); be16(16); for (char ch : {'J', 'F', 'I', 'F', '\0'}) out.push_back((uint8_t)ch); out.push_back(1); out.push_back(1); out.push_back(0); be16(1); be16(1); out.push_back(0); out.push_back(0);
    for (int t = 0; t < 2; t++) { marker(0xDB); be16(67); out.push_back((uint8_t)t); for (int i = 0; i < 64; i++) out.push_back(qt[t][kZigzag[i]]); }
    marker(0xC0); be16(17); out.push_back(8); be16(h); be16(w); out.push_back(3);
    for (int c = 0; c < 3; c++) { out.push_back((uint8_t)(c + 1)); out.push_back(0x11); out.push_back(c ? 1 : 0); }
    auto dht = [&](int cls_id, const uint8_t bits[16], const uint8_t *vals, int nvals) {
        marker(0xC4); be16(19 + nvals); out.push_back((uint8_t)cls_id);
        for (int i = 0; i < 16; i++) out.push_back(bits[i]);
        for (int i = 0; i < nvals; i++) out.push_back(vals[i]);
    };
    dht(0x00, kDcLumBits, kDcVals, 12); dht(0x10, kAcLumBits, kAcLumVals, 162);
    dht(0x01, kDcChrBits, kDcVals, 12); dht(0x11, kAcChrBits, kAcChrVals, 162);
    marker(0xDA); be16(12); out.push_back(3);
    out.push_back(1); out.push_back(0x00); out.push_back(2); out.push_back(0x11); out.push_back(3); out.push_back(0x11);
    out.push_back(0); out.push_back(63); out.push_back(0);
    BitSink bs{out};
    int pred[3] = {0, 0, 0};
    float blk[3][64];
    for (int by = 0; by < h; by += 8)
        for (int bx = 0; bx < w; bx += 8) {
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    const int yy = by + y < h ? by + y : h - 1, xx = bx + x < w ? bx + x : w - 1;     // edge blocks repeat the last row / column
                    const uint8_t *p = bgr + ((size_t)(h - 1 - yy) * w + xx) * 3;
                    const float b = p[0], g = p[1], r = p[2];
                    blk[0][8 * y + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                    blk[1][8 * y + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                    blk[2][8 * y + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                }
            encode_block(blk[0], qf[0], C, pred[0], dcl, acl, bs);
            encode_block(blk[1], qf[1], C, pred[1], dcc, acc, bs);
            encode_block(blk[2], qf[1], C, pred[2], dcc, acc, bs);
        }
    bs.flush();
    marker(0xD9);
}

int avi_open(const char *path, int width, int height, double fps, int quality, rt_avi **out, const char *who)
{
    if (!path || !out) return fail(std::string(who) + ": null argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535 || !(fps > 0)) return fail(std::string(who) + ": bad size or frame rate");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(std::string(who) + ": cannot open " + path);
    rt_avi *a = new rt_avi;
    a->f = f; a->w = width; a->h = height; a->fps = fps; a->quality = quality;
    a->stride = ((uint32_t)width * 3u + 3u) & ~3u;
    avi_header(a);
    *out = a;
    return 0;
}
}
extern "C" {

int rt_avi_open(const char *path, int width, int height, double fps, rt_avi **out)
{
    return avi_open(path, width, height, fps, 0, out, "rt_avi_open");
}

int rt_avi_open_mjpg(const char *path, int width, int height, double fps, int quality, rt_avi **out)
{
    if (quality < 1 || quality > 100) return fail("rt_avi_open_mjpg: quality must be in [1,100]");
    return avi_open(path, width, height, fps, quality, out, "rt_avi_open_mjpg");
}

int rt_avi_add_frame(rt_avi *a, const uint8_t *bgr)
{
    if (!a || !a->f || !bgr) return fail("rt_avi_add_frame: null argument");
    std::vector<uint8_t> jpg;
    uint32_t bytes = a->stride * (uint32_t)a->h;
    if (a->quality) { encode_jpeg(bgr, a->w, a->h, a->quality, jpg); bytes = (uint32_t)jpg.size(); }
    const uint32_t padded = bytes + (bytes & 1u);
    if (a->movi_bytes + padded + 8 + 16ull * (a->frames + 1) > 0xF0000000ull) return fail("rt_avi_add_frame: file would exceed the 4 GB RIFF limit");
    tag(a->f, a->quality ? "00dc" : "00db"); le32(a->f, bytes);
    if (a->quality) {
        if (std::fwrite(jpg.data(), 1, bytes, a->f) != bytes) return fail("rt_avi_add_frame: write failed");
        if (bytes & 1u) std::fputc(0, a->f);
    } else {
        const unsigned char zero[4] = {0, 0, 0, 0};
        for (int y = 0; y < a->h; y++) {
            if (std::fwrite(bgr + (size_t)y * a->w * 3, 1, (size_t)a->w * 3, a->f) != (size_t)a->w * 3) return fail("rt_avi_add_frame: write failed");
            std::fwrite(zero, 1, a->stride - (uint32_t)a->w * 3u, a->f);
        }
    }
    a->sizes.push_back(bytes);
    a->movi_bytes += 8 + padded;
    if (bytes > a->max_frame) a->max_frame = bytes;
    a->frames++;
    return 0;
}

int rt_avi_close(rt_avi *a)
{
    if (!a) return 0;
    int rc = 0;
    if (a->f) {
        tag(a->f, "idx1"); le32(a->f, a->frames * 16);
        uint32_t off = 4;
        for (uint32_t i = 0; i < a->frames; i++) {
            tag(a->f, a->quality ? "00dc" : "00db"); le32(a->f, 0x10); le32(a->f, off); le32(a->f, a->sizes[i]);
            off += 8 + a->sizes[i] + (a->sizes[i] & 1u);
        }
        avi_header(a);
        if (std::fclose(a->f) != 0) rc = fail("rt_avi_close: close failed");
    }
    delete a;
    return rc;
}

} // extern "C"
