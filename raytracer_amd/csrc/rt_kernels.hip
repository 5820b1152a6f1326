// rt_kernels.hip -- gfx950 kernels: octree closest hit, primary-ray id buffer, and the
// Monte-Carlo path-tracing loop (persistent wave64 lanes).
//
// Everything is fp64 and must be built with -ffp-contract=off: hit primitive ids are
// compared bit for bit with a CPU evaluation of the same expressions, so the operation
// order of the reference's Vec3 helpers (types.h:19-74), of hitBox (algorithms.h:208-236)
// and of std::min / std::max is kept literally.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_device.h"
#include "rt_kernels.h"

namespace rt {

#define RT_INFINITE 1e10     // definitions.h:5
#define RT_ZERO 1e-9         // definitions.h:6
#define RT_PI 3.1415926      // definitions.h:7
#define TRI_DET_EPS 1e-200
#define TRI_GUARD2 1e-10      // (sine of the triangle's angle at v0 * cosine between ray and normal)^2 below which the determinant counts as zero

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 had(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }    // Vec4::operator&
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vec3::operator^, types.h:66-74
__device__ __forceinline__ V3 cross(V3 l, V3 r)
{
    return mk(l.y * r.z - r.y * l.z, l.z * r.x - l.x * r.z, l.x * r.y - r.x * l.y);
}
__device__ __forceinline__ double len(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// Vec3::normalize multiplies by the reciprocal, types.h:19-23
__device__ __forceinline__ V3 norm(V3 a) { return a * (1 / len(a)); }
__device__ __forceinline__ bool iszero(V3 a) { return fabs(a.x) < RT_ZERO && fabs(a.y) < RT_ZERO && fabs(a.z) < RT_ZERO; }
// Vec3::reflact, types.h:37-40
__device__ __forceinline__ V3 reflact(V3 d, V3 n)
{
    V3 proj = (-n) * dot(d, n);
    return norm(proj * 2 + d);
}
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }   // std::min
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }   // std::max

// ------------------------------------------------------------------ counter-based sampler
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ uint64_t hash2(uint64_t a, uint64_t b)
{
    return mix64((a ^ (b * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL);
}
__device__ __forceinline__ double u01(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }
// keyed permutation of [0,l) (Kensler 2013, "Correlated Multi-Jittered Sampling")
__device__ inline uint32_t permute(uint32_t i, uint32_t l, uint32_t p)
{
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p;              i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;         i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1;   i *= 1 | p >> 27;
                             i *= 0x6935fa69u;
        i ^= (i & w) >> 11;  i *= 0x74dcb303u;
        i ^= (i & w) >> 2;   i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;   i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
#define KEY_JITTER 0x4a495454ULL
#define KEY_PATH   0x50415448ULL
#define SLOT_BASE  16
enum { BR_DIFFUSE = 1, BR_SPECULAR = 2, BR_REFLECT = 3, BR_REFRACT = 4 };
enum { SLOT_RR = 0, SLOT_DIFF1 = 1, SLOT_DIFF2 = 2, SLOT_DISK = 3 };

// Entry k of object `object_id`'s n x m jitter table for this pixel: the table of
// RTObject::calRandom (RTObject.cpp:46-69) evaluated lazily -- stratum sigma(k) of a keyed
// permutation stands for the shuffled slot, one jitter pair per stratum.
__device__ inline void jitter_pair(uint64_t seed, uint64_t pixel, uint32_t object_id, uint32_t k,
                                   uint32_t n, uint32_t m, double &a, double &b)
{
    uint64_t kp = hash2(hash2(seed, pixel), KEY_JITTER + object_id);
    uint32_t sigma = permute(k, n * m, (uint32_t)(kp >> 32));
    uint32_t j = sigma % n, i = sigma / n;
    double nstep = 1.0 / n, mstep = 1.0 / m;
    double xi1 = u01(hash2(kp, 2ULL * sigma));
    double xi2 = u01(hash2(kp, 2ULL * sigma + 1));
    a = j * nstep + nstep * xi1;
    b = i * mstep + mstep * xi2;
}
__device__ __forceinline__ uint64_t path_root_key(uint64_t seed, uint64_t pixel, uint32_t k)
{
    return hash2(hash2(hash2(seed, pixel), KEY_PATH), k);
}
__device__ __forceinline__ double draw(uint64_t key, uint32_t slot) { return u01(hash2(key, SLOT_BASE + slot)); }

// ------------------------------------------------------------------ intersection
// algorithm::hitBox, algorithms.h:223-236
__device__ __forceinline__ bool hit_box(V3 lo, V3 hi, V3 e, V3 d, double t0, double t1, double &r0o, double &r1o)
{
    double ax = (lo.x - e.x) / d.x, ay = (lo.y - e.y) / d.y, az = (lo.z - e.z) / d.z;
    double bx = (hi.x - e.x) / d.x, by = (hi.y - e.y) / d.y, bz = (hi.z - e.z) / d.z;
    double r0 = smin(ax, bx), r1 = smax(ax, bx);
    double m;
    m = smin(ay, by); if (r0 < m) r0 = m;            // std::max_element over {x, y, z, t0}
    m = smin(az, bz); if (r0 < m) r0 = m;
    if (r0 < t0) r0 = t0;
    m = smax(ay, by); if (m < r1) r1 = m;            // std::min_element over {x, y, z, t1}
    m = smax(az, bz); if (m < r1) r1 = m;
    if (t1 < r1) r1 = t1;
    r0o = r0; r1o = r1;
    return r0 <= r1;
}

// rays traced, inner-node and leaf visits, primitives covered (what the reference's walk tests); then what
// was executed for them: single-precision box tests and double-precision primitive tests
struct Counters { uint32_t rays, inner, leaf, prim, box, fp64; };

// One 80-byte triangle record in registers.
struct TriReg { double v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z; uint32_t id; };
__device__ __forceinline__ TriReg tri_load(const LeafTri *p)
{
    // four 16-byte loads and one 12-byte load (the record's last word is padding and is not
    // loaded: a register written by a load in flight cannot be reused until the load lands)
    const double2 *q = reinterpret_cast<const double2 *>(p);
    const double2 a = q[0], b = q[1], c = q[2], d = q[3];
    TriReg r;
    r.v0x = a.x; r.v0y = a.y; r.v0z = b.x; r.e1x = b.y; r.e1y = c.x; r.e1z = c.y; r.e2x = d.x; r.e2y = d.y;
    r.e2z = p->e2[2];
    r.id = p->id;
    return r;
}

// Two-sided Moeller-Trumbore on a (v0, e1, e2) record; the octree payload test that stands
// where BoxNode::calHit does in the reference (Surface.cpp:43-59).  The barycentric tests
// compare the un-normalised numerators with |det| (sign flips are exact), so a division is
// only spent on rays that pass them.
__device__ __forceinline__ bool tri_hit(const TriReg &tr, V3 e, V3 d, double t0, double t1, double &tout)
{
    V3 v0 = mk(tr.v0x, tr.v0y, tr.v0z), e1 = mk(tr.e1x, tr.e1y, tr.e1z), e2 = mk(tr.e2x, tr.e2y, tr.e2z);
    V3 p = cross(d, e2);
    double det = dot(e1, p);
    double a = fabs(det);
    if (!(a >= TRI_DET_EPS)) return false;
    // A determinant within a few orders of its own rounding error is no hit: det = |d| |e1| |e2| * (sine of
    // the triangle's angle at v0) * (cosine between ray and normal), and the test wants that product of sine
    // and cosine to reach 1e-5.  Below it the quotients are noise over noise (a ray between lattice points in
    // the plane of a lattice triangle has numerators of exactly zero and "hits" at t = 0, wherever the
    // triangle is); above it the point the test accepts lies within 64 ulp * (distance + coordinates) / 1e-5
    // of the triangle, which is what the boxes of the pre-test are grown by (RT_BOX_GROW, RT_REACH_GRID).
    if (!(det * det >= TRI_GUARD2 * dot(e1, e1) * dot(e2, e2) * dot(d, d))) return false;
    V3 s = e - v0;
    double up = dot(s, p);
    double us = det < 0 ? -up : up;
    if (!(us >= 0.0 && us <= a)) return false;
    V3 q = cross(s, e1);
    double vp = dot(d, q);
    double vs = det < 0 ? -vp : vp;
    if (!(vs >= 0.0 && us + vs <= a)) return false;
    double t = dot(e2, q) / det;
    if (!(t >= t0 && t <= t1)) return false;
    tout = t;
    return true;
}

// Sphere::hit, Surface.cpp:73-111 (the tangent branch skips the [t0,t1] window)
__device__ __forceinline__ bool sphere_hit(V3 center, double radius, V3 e, V3 d, double t0, double t1, double &tout)
{
    V3 oc = e - center;
    double doc = dot(d, oc);
    double dieta = doc * doc - dot(d, d) * (dot(oc, oc) - radius * radius);
    double deno = dot(d, d);
    double pre = dot(-d, oc);
    double t;
    if (fabs(dieta) < RT_ZERO) {
        t = pre / deno;
    } else if (dieta > 0) {
        double sq = sqrt(dieta);
        double pt = (pre - sq) / deno;
        double nt = (pre + sq) / deno;
        bool f1 = (pt > t1 || pt < t0), f2 = (nt > t1 || nt < t0);
        if (f1 && f2) return false;
        t = !f1 ? pt : nt;
    } else {
        return false;
    }
    tout = t;
    return true;
}
// Ground / Wall_z / Wall_x ::hit, Surface.cpp:124-133, 194-204, 217-226
__device__ __forceinline__ bool plane_hit(int type, double coord, V3 e, V3 d, double t0, double t1, double &tout)
{
    double ee = (type == RT_OBJ_GROUND) ? e.y : (type == RT_OBJ_WALL_Z ? e.z : e.x);
    double dd = (type == RT_OBJ_GROUND) ? d.y : (type == RT_OBJ_WALL_Z ? d.z : d.x);
    if (ee > coord && dd < 0) {
        double t = (ee - coord) / -dd;
        if (t < t0 || t > t1) return false;
        tout = t;
        return true;
    }
    return false;
}

struct Hit { double t; int32_t obj; uint32_t prim; };

// ------------------------------------------------------------------ the closest-hit engine
// group_->calHit(ray, t0, t1, &rec) for a stream of rays: Group::hit (Surface.cpp:160-181)
// walks the scene's objects in order, every hit shrinking t1; a mesh object runs Octree::hit
// (octree.cpp:285-376), the ordered depth-first search, on its flattened tree.
//
// The engine is written for a 64-wide wave whose lanes follow DIFFERENT rays.  Each lane is a
// small state machine:
//     FETCH  take the next ray of the launch (persistent lanes: a lane that finishes its ray
//            takes another one; the lanes of a wave that need rays get consecutive indices
//            from one atomic, ballot + popcount)
//     OBJ    advance through the object list: spheres, planes and meshes whose octree is a
//            single leaf are tested on the spot, a larger mesh opens its octree root
//     NODE / POP   open an octree node / pop the next child from the LDS stack and open it
//     TRI    walk the triangle run of the opened node, two triangles per step
// and the wave executes one KIND of step at a time, the kind most lanes are waiting for (see
// the weights below).  Without this a wave would run at the pace of its longest ray and its
// longest leaf.
//
// Per open inner node the LDS stack keeps {child_base, visiting order (8 x 3 bits) |
// count << 24 | next << 28}.  The reference re-checks `minrec.t < next.t` before descending
// (octree.cpp:359); next.t is the child's own slab entry distance, which the child visit
// recomputes bit-identically, so the check is made there.  The child list of an inner node is
// computed before its own triangles are tested (the reference does it after, octree.cpp:320-344);
// the list does not depend on those tests, so the visit order is unchanged.
//
// FAST node steps (every participating ray has all direction components far from zero): a/d
// is formed from the ray's correctly rounded reciprocals and two fused multiply-adds
// (q0 = a*r; q = fma(fma(-q0,d,a),r,q0)), which reproduces the IEEE quotient, and no NaN can
// appear in the slab arithmetic, so fmin / fmax equal std::min / std::max.  Otherwise true
// divisions and the literal comparison forms are used (zero components give the +-inf / NaN
// cases of hitBox, algorithms.h:208-236).
// the closest-hit kernels are held to three waves per SIMD (168 VGPRs): left alone the register
// allocator takes a few more and drops to two
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 3
#endif
#define RT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(RT_TRACE_WAVES, RT_TRACE_WAVES)))
// largest batch of rays a wave reserves at a time, for rays that reach a large octree and for the rest
#ifndef RT_BATCH1
#define RT_BATCH1 128
#endif
#ifndef RT_BATCH0
#define RT_BATCH0 1024
#endif
#ifndef RT_WT
#define RT_WT 24
#endif
enum { TM_FETCH = 0, TM_OBJ = 1, TM_NODE = 2, TM_POP = 3, TM_TRI = 4, TM_EXIT = 5 };
// flag in TravState::oi while the lane's ray is expressed in an implicit surface's own frame
#define RT_OI_LOCAL 0x40000000
// Which kind of step a wave runs next: the one with the largest weighted count of waiting
// lanes.  Weights are in eighths of a triangle-step lane (triangle steps weigh 8); found by
// timing, a node step is preferred at 4/5 of the triangle lanes.  The weight of an object phase
// depends on what a sweep over the scene's objects costs and comes with the scene
// (DevScene::obj_weight): an object phase is also where lanes that have finished get their next
// ray, so the cheaper the sweep, the sooner it pays to go back to it.
#ifndef RT_WN
#define RT_WN 10
#endif
#ifdef RT_WO
#define RT_OBJ_WEIGHT(S) (RT_WO)
#else
#define RT_OBJ_WEIGHT(S) ((S).obj_weight)
#endif
template <bool FAST> __device__ __forceinline__ double tmin2(double a, double b) { return FAST ? fmin(a, b) : smin(a, b); }
template <bool FAST> __device__ __forceinline__ double tmax2(double a, double b) { return FAST ? fmax(a, b) : smax(a, b); }
template <bool FAST> __device__ __forceinline__ double qdiv(double a, double d, double r)
{
    if (FAST) {
        double q0 = a * r;
        double rem = __builtin_fma(-q0, d, a);
        return __builtin_fma(rem, r, q0);
    }
    return a / d;
}
__device__ __forceinline__ bool dir_is_regular(V3 d)
{
    const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    return ax >= 1e-150 && ax <= 1e150 && ay >= 1e-150 && ay <= 1e150 && az >= 1e-150 && az <= 1e150;
}
// coordinates that are zero or of magnitude 1e-100 .. 1e100: differences of two such numbers
// divided by a regular direction component neither overflow nor reach the subnormal range
__device__ __forceinline__ bool coord_is_regular(double x) { const double a = fabs(x); return a == 0.0 || (a >= 1e-100 && a <= 1e100); }
__device__ __forceinline__ bool org_is_regular(V3 e) { return coord_is_regular(e.x) && coord_is_regular(e.y) && coord_is_regular(e.z); }

// The box pre-test.  A single-precision test of the ray against a record's bounding box decides
// whether the 80-byte record is fetched and the fp64 test made at all; the same test against the
// box of a whole segment of a run decides whether the segment's records are looked at at all.  The
// boxes are bytes in the grid of the node that owns the run (rt_pretest.h), rounded outwards by at
// least RT_BOX_GROW grid units at build time, around the part of the primitive inside the node's box.
// The ray is taken into that grid when the run opens: g = the point where it enters the node's box
// (at most 256 units from the grid's origin), n = its unit direction; a point of the ray is
// g + tau * n, and the stretch inside the node's box is 0 <= tau <= tend.  Slab test per axis:
// tau = (b - g) / n = b * inv + c with inv = 1 / n and c = -g * inv, for b = the box's lower and
// upper plane; the ray crosses the box inside the window iff
//     max(min over the planes of every axis, 0) <= min(max over the planes of every axis, tend).
// Errors, each expressed as the shift of a plane that would have the same effect: g is rounded to
// single precision (3e-5 units, the arithmetic that forms it included), the direction too (1e-7 rad
// over at most 450 units: 5e-5), c and the fused multiply-add round once each (6e-8 of at most 450
// units: 3e-5 each) -- under 2e-4 units in all, a tenth of the 2e-3 units of the boxes' growth that are
// meant for it; the window's ends are off by less than 1e-4 units and RT_BOX_SLACK = 4e-3 units is given
// at either end.  The rest of the growth covers the fp64 test itself: what it accepts lies within
// 64 ulp * reach / 1e-5 of the triangle (tri_hit's guard bounds the conditioning), at most 2.8e-2 grid
// units while the ray's reach is under RT_REACH_GRID units of the node's grid, and a ray with a longer
// reach is not pre-tested in that node (node_step).  So the test only rejects rays that pass the
// primitive's part inside the node at a clear distance, which the fp64 test, hitting only inside the
// window, rejects too.  A direction component of zero gives inv = inf and tau = NaN for that axis;
// minima and maxima skip a NaN, so the axis does not constrain the result, and a test whose
// operands are all NaN passes.
// RayB is the lane's ray for that test: inv (written whenever the ray's direction changes), c and
// tend (written when a run opens).
#define RT_BOX_SLACK 4e-3f
struct RayB { float ix, iy, iz, tend, cx, cy, cz; };
struct RayInv { float ix, iy, iz, len; };
__device__ __forceinline__ RayInv ray_single(V3 d)
{
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float l2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    const float inv = __builtin_amdgcn_rsqf(l2);
    RayInv r;
    r.ix = __builtin_amdgcn_rcpf(dx * inv); r.iy = __builtin_amdgcn_rcpf(dy * inv); r.iz = __builtin_amdgcn_rcpf(dz * inv);
    r.len = l2 * inv;                                   // |d|
    return r;
}
__device__ __forceinline__ bool box_pass(uint32_t w0, uint32_t w1, const RayB &r)
{
    const float lx = (float)(w0 & 255u), ly = (float)((w0 >> 8) & 255u), lz = (float)((w0 >> 16) & 255u), hx = (float)(w0 >> 24);
    const float hy = (float)(w1 & 255u), hz = (float)((w1 >> 8) & 255u);
    const float ax = __builtin_fmaf(lx, r.ix, r.cx), bx = __builtin_fmaf(hx, r.ix, r.cx);
    const float ay = __builtin_fmaf(ly, r.iy, r.cy), by = __builtin_fmaf(hy, r.iy, r.cy);
    const float az = __builtin_fmaf(lz, r.iz, r.cz), bz = __builtin_fmaf(hz, r.iz, r.cz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), -RT_BOX_SLACK));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), r.tend));
    return !(tn > tf);
}

struct TravState {
    // ray
    V3 e, d, r;
    double t0, t1;          // t1 shrinks with every object hit (Surface.cpp:168,175)
    uint32_t item;          // index of the ray in its source
    int32_t hobj; uint32_t hprim;
    int32_t oi;             // object being processed
    bool fast;
    // octree
    double mint; uint32_t minprim;
    int sp; uint32_t cur, root;
    // triangle run: OctreeNode::hit (octree.cpp:19-38) hands every primitive the window
    // [bt0, bt1] and keeps a candidate only if t <= tempt; tempt starts at bt1 and only
    // shrinks, so the window's upper end is tempt itself and the best t so far is tempt too.
    uint32_t k, kend, bprim;
    double bt0, tempt;
    bool leaf;
};

// hitBox for a ray with regular direction: same values as hit_box (FAST node steps, above)
__device__ __forceinline__ bool hit_box_fast(V3 lo, V3 hi, V3 e, V3 d, V3 r, double t0, double t1, double &r0o, double &r1o)
{
    const double ax = qdiv<true>(lo.x - e.x, d.x, r.x), ay = qdiv<true>(lo.y - e.y, d.y, r.y), az = qdiv<true>(lo.z - e.z, d.z, r.z);
    const double bx = qdiv<true>(hi.x - e.x, d.x, r.x), by = qdiv<true>(hi.y - e.y, d.y, r.y), bz = qdiv<true>(hi.z - e.z, d.z, r.z);
    const double r0 = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmax(fmin(az, bz), t0));
    const double r1 = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmin(fmax(az, bz), t1));
    r0o = r0; r1o = r1;
    return r0 <= r1;
}

// Data that every lane of a wave reads at the same address (the object table, the root node and
// the triangles of a one-leaf mesh) is read through constant-address-space pointers: the scene
// is immutable while kernels run, and such reads go through the scalar path.
#define RT_UNIFORM __attribute__((address_space(4)))
typedef const DevObject RT_UNIFORM *u_object;
typedef const LeafTri RT_UNIFORM *u_leaftri;
__device__ __forceinline__ TriReg tri_load_uniform(u_leaftri p)
{
    TriReg r;
    r.v0x = p->v0[0]; r.v0y = p->v0[1]; r.v0z = p->v0[2];
    r.e1x = p->e1[0]; r.e1y = p->e1[1]; r.e1z = p->e1[2];
    r.e2x = p->e2[0]; r.e2y = p->e2[1]; r.e2z = p->e2[2];
    r.id = p->id;
    return r;
}

__device__ __forceinline__ double any_f64() { double x; asm volatile("" : "=v"(x)); return x; }
__device__ __forceinline__ uint32_t any_u32() { uint32_t x; asm volatile("" : "=v"(x)); return x; }

// One step of the walk for every lane that stands at a node (`in`): the node's slab test, Octree::hit's
// stop rule, the children the ray passes through pushed in order, and the node's own run opened if the
// box around it lets the ray through.  Written without branches on per-lane conditions: an instruction
// costs the wave the same whether one lane or all of them need it, and every join of divergent paths
// costs register copies, so the lanes' outcomes are selected at the end instead; the only branches are
// on conditions the whole wave shares.
template <bool FAST, int BLOCK>
__device__ __forceinline__ void node_step(const DevScene &S, TravState &T, int &mode, const bool in,
                                          uint2 (*stk)[BLOCK], double *run_bt1, float4 (*rayf)[BLOCK], Counters &c)
{
    const int tid = threadIdx.x;
    const V3 e = T.e, d = T.d, r = T.r;
    const double t0 = T.t0, t1 = T.t1;
    // lanes that take no part compute on whatever their registers hold (a value the compiler must treat
    // as defined, made by no instruction): every result of theirs is masked by `in`
    double lo0 = any_f64(), lo1 = any_f64(), lo2 = any_f64(), hi0 = any_f64(), hi1 = any_f64(), hi2 = any_f64();
    uint32_t child_base = any_u32(), obj_begin = any_u32(), obj_count = any_u32(), nd_pad = any_u32();
    uint2 run_box = make_uint2(any_u32(), any_u32());       // around everything in the node's run (asked for together with the node)
    if (in) {
        const FlatNode *nd = S.nodes + T.cur;
        lo0 = nd->lo[0]; lo1 = nd->lo[1]; lo2 = nd->lo[2];
        hi0 = nd->hi[0]; hi1 = nd->hi[1]; hi2 = nd->hi[2];
        child_base = nd->child_base; obj_begin = nd->obj_begin; obj_count = nd->obj_count; nd_pad = nd->pad;
        __builtin_memcpy(&run_box, S.run_box + 2ull * T.cur, 8);
    }
#if RT_STEP_STATS == 13
    // clock ticks (/16), lane 0 of the wave: inner = the wait for the node record, leaf = the rest of the step
    const bool first_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u;
    const long long tn0_ = clock64();
    __builtin_amdgcn_s_waitcnt(0);
    if (lo0 + hi0 + (double)child_base + (double)run_box.x == 1.2345e300) c.box++;
    const long long tn1_ = clock64();
    if (first_) c.inner += (uint32_t)((tn1_ - tn0_) >> 4);
    struct Tock { long long t; bool f; Counters &c; __device__ ~Tock() { if (f) c.leaf += (uint32_t)((clock64() - t) >> 4); } } tock_{tn1_, first_, c};
#endif
    // per-axis plane distances; the node's own slab test and all 8 child tests use them
    const double al0 = qdiv<FAST>(lo0 - e.x, d.x, r.x), al1 = qdiv<FAST>(lo1 - e.y, d.y, r.y), al2 = qdiv<FAST>(lo2 - e.z, d.z, r.z);
    const double ah0 = qdiv<FAST>(hi0 - e.x, d.x, r.x), ah1 = qdiv<FAST>(hi1 - e.y, d.y, r.y), ah2 = qdiv<FAST>(hi2 - e.z, d.z, r.z);
    double n0 = tmin2<FAST>(al0, ah0), n1 = tmax2<FAST>(al0, ah0);             // hitBox, algorithms.h:223-236
    if (FAST) {
        n0 = fmax(fmax(n0, fmin(al1, ah1)), fmax(fmin(al2, ah2), t0));
        n1 = fmin(fmin(n1, fmax(al1, ah1)), fmin(fmax(al2, ah2), t1));
    } else {
        double m;
        m = smin(al1, ah1); if (n0 < m) n0 = m;
        m = smin(al2, ah2); if (n0 < m) n0 = m;
        if (n0 < t0) n0 = t0;
        m = smax(al1, ah1); if (m < n1) n1 = m;
        m = smax(al2, ah2); if (m < n1) n1 = m;
        if (t1 < n1) n1 = t1;
    }
    const bool hb = n0 <= n1;
    // octree.cpp:291 / octree.cpp:359-361: a root the ray misses, or a node that starts behind the
    // closest hit so far, finishes this octree
    const bool alive = in & !(T.cur == T.root ? !hb : (T.mint < n0));
    const bool inner = alive & (child_base != 0);
#if !defined(RT_STEP_STATS)
    c.inner += inner ? 1u : 0u; c.leaf += (alive & !inner) ? 1u : 0u;
#else
    if (RT_STEP_STATS == 2) c.leaf += (alive & !inner) ? 1u : 0u;
#endif
    if (__any(inner)) {
        // 8 child slab tests (octree.cpp:337-343) from the {lo, mid, hi} plane distances
        const double am0 = qdiv<FAST>((lo0 + hi0) / 2 - e.x, d.x, r.x);
        const double am1 = qdiv<FAST>((lo1 + hi1) / 2 - e.y, d.y, r.y);
        const double am2 = qdiv<FAST>((lo2 + hi2) / 2 - e.z, d.z, r.z);
        uint32_t cnt = 0, order = 0;
        // The children the ray passes through and their order, the short way: in exact arithmetic the
        // outcome of the eight slab tests and the sort below is a function of the ORDER of the numbers
        // they compare -- the nine plane distances and the window's ends, no new arithmetic -- and
        // while no two of the numbers that matter are equal it is this: the ray enters the node at n0 in
        // the octant that axis k puts it in (the near half unless the middle plane lies behind the entry,
        // am_k < n0), and every middle plane it crosses inside the window, n0 < am_k < n1, takes it to
        // the next octant, in the order of the am_k; each octant is entered where the one before is
        // left, so the entry distances increase strictly and no tie rule comes into play.  Equal numbers
        // (a middle plane met exactly at the window's end, two middle planes met together) mean children
        // touched in a single point, which the reference visits and orders by index: then the whole
        // wave takes the long way below, literally.  (Checked against it for every order of the values.)
        bool long_way = true;
        if (FAST) {
            const bool g0 = am0 > n0, g1 = am1 > n0, g2 = am2 > n0;
            const bool i0 = g0 & (am0 < n1), i1 = g1 & (am1 < n1), i2 = g2 & (am2 < n1);
            const bool tie = (am0 == n0) | (am1 == n0) | (am2 == n0) | (am0 == n1) | (am1 == n1) | (am2 == n1) |
                             (i0 & i1 & (am0 == am1)) | (i0 & i2 & (am0 == am2)) | (i1 & i2 & (am1 == am2));
            long_way = __any(inner & tie);
#if RT_STEP_STATS == 12
            if (inner) { c.rays++; if (long_way) c.inner++; if (tie) c.leaf++; }    // lanes at inner nodes, lanes taken the long way, lanes with a tie of their own
#endif
            if (!long_way) {
                const bool y_lt_x = am1 < am0, z_lt_x = am2 < am0, z_lt_y = am2 < am1;
                // crossings that come before this axis's own
                const bool ax = i1 & y_lt_x, bx = i2 & z_lt_x, ay = i0 & !y_lt_x, by = i2 & z_lt_y, az = i0 & !z_lt_x, bz = i1 & !z_lt_y;
                // the half (0 low, 1 high) per axis in the first octant, and which axes have flipped by the
                // second, third, fourth octant: four 3-bit fields of one word, worked on together
                const uint32_t o0 = (uint32_t)((d.x < 0) != !g0) | (uint32_t)((d.y < 0) != !g1) << 1 | (uint32_t)((d.z < 0) != !g2) << 2;
                const uint32_t flips = (uint32_t)(i0 & !ax & !bx) << 3 | (uint32_t)(i1 & !ay & !by) << 4 | (uint32_t)(i2 & !az & !bz) << 5 |
                                       (uint32_t)(i0 & !(ax & bx)) << 6 | (uint32_t)(i1 & !(ay & by)) << 7 | (uint32_t)(i2 & !(az & bz)) << 8 |
                                       (uint32_t)i0 << 9 | (uint32_t)i1 << 10 | (uint32_t)i2 << 11;
                cnt = 1u + (uint32_t)__builtin_popcount(flips >> 9);
                const uint32_t octs = (o0 * 0x249u) ^ flips;                // o0 in all four fields, then the flips
                // x-half | y-half << 1 | z-half << 2  ->  child index in getChildBound's corner order
                // (0 1 3 2 4 5 7 6): the x bit flips where the y bit is set
                order = octs ^ ((octs >> 1) & 0x249u);
            }
        }
        if (long_way) {
            const double mnL[3] = { tmin2<FAST>(al0, am0), tmin2<FAST>(al1, am1), tmin2<FAST>(al2, am2) };
            const double mxL[3] = { tmax2<FAST>(al0, am0), tmax2<FAST>(al1, am1), tmax2<FAST>(al2, am2) };
            const double mnH[3] = { tmin2<FAST>(am0, ah0), tmin2<FAST>(am1, ah1), tmin2<FAST>(am2, ah2) };
            const double mxH[3] = { tmax2<FAST>(am0, ah0), tmax2<FAST>(am1, ah1), tmax2<FAST>(am2, ah2) };
            // entry distance per child, +inf for children the ray misses
            double ct[8];
            cnt = 0;
    #pragma unroll
            for (int ci = 0; ci < 8; ci++) {
                // corner order of getChildBound: x-half, y-half, z-half of child ci
                const int hx = (ci == 1 || ci == 2 || ci == 5 || ci == 6), hy = (ci == 2 || ci == 3 || ci == 6 || ci == 7), hz = ci >> 2;
                double r0, r1;
                if (FAST) {
                    r0 = fmax(fmax(hx ? mnH[0] : mnL[0], hy ? mnH[1] : mnL[1]), fmax(hz ? mnH[2] : mnL[2], t0));
                    r1 = fmin(fmin(hx ? mxH[0] : mxL[0], hy ? mxH[1] : mxL[1]), fmin(hz ? mxH[2] : mxL[2], t1));
                } else {
                    double q;
                    r0 = hx ? mnH[0] : mnL[0]; r1 = hx ? mxH[0] : mxL[0];
                    q = hy ? mnH[1] : mnL[1]; if (r0 < q) r0 = q;
                    q = hz ? mnH[2] : mnL[2]; if (r0 < q) r0 = q;
                    if (r0 < t0) r0 = t0;
                    q = hy ? mxH[1] : mxL[1]; if (q < r1) r1 = q;
                    q = hz ? mxH[2] : mxL[2]; if (q < r1) r1 = q;
                    if (t1 < r1) r1 = t1;
                }
                const bool h = r0 <= r1;
                ct[ci] = h ? r0 : __builtin_huge_val();
                cnt += h ? 1u : 0u;
            }
            // std::sort by entry distance (octree.cpp:344); equal keys keep child order.  rank[i] is
            // child i's place in that order: i, plus later children that go first (strictly smaller
            // key), minus earlier children it overtakes.  The hit children take ranks 0..cnt-1.
            // Child 0 contributes nothing to the packed order word, so its rank is not formed.
            uint32_t rank[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
    #pragma unroll
            for (int i = 0; i < 8; i++) {
    #pragma unroll
                for (int j = i + 1; j < 8; j++) {
                    const uint32_t jfirst = ct[j] < ct[i] ? 1u : 0u;
                    if (i) rank[i] += jfirst;
                    rank[j] -= jfirst;
                }
            }
            order = 0;
    #pragma unroll
            for (int i = 1; i < 8; i++) order |= (uint32_t)i << (3 * rank[i]);
        }
        const bool push = inner & (cnt != 0u);
        if (push) stk[T.sp][tid] = make_uint2(child_base, order | (cnt << 24));
        T.sp += push ? 1 : 0;
    }
    const bool has = alive & hb & (obj_count != 0u);        // OctreeNode::hit, octree.cpp:19-38
    bool open = false;
    if (__any(has)) {
        // the ray for the box pre-test: the point g where it enters this node's box, in the node's grid, as
        // c = -g / direction per axis, and the length of its stretch inside the box (see RayB)
        const float sc = __uint_as_float(nd_pad & ~1u);
        const float gx = (float)((e.x + d.x * n0) - lo0) * sc + (float)RT_GRID_ORG, gy = (float)((e.y + d.y * n0) - lo1) * sc + (float)RT_GRID_ORG;
        const float gz = (float)((e.z + d.z * n0) - lo2) * sc + (float)RT_GRID_ORG;
        const float4 fi = rayf[0][tid];
        const float dlen = rayf[1][tid].w;
        RayB rb;
        rb.ix = fi.x; rb.iy = fi.y; rb.iz = fi.z;
        rb.cx = -(gx * fi.x); rb.cy = -(gy * fi.y); rb.cz = -(gz * fi.z);
        rb.tend = (float)(n1 - n0) * dlen * sc + RT_BOX_SLACK;
        {
            // The boxes were grown for rays whose reach -- the largest coordinate of the origin plus the distance
            // to the far end of the node's window -- is at most RT_REACH_GRID units of this node's grid: so far
            // the double-precision test's own error stays inside the growth.  Beyond it (a node that is tiny
            // against its distance from the ray's origin or from the coordinate origin) the pre-test says yes to
            // everything in this run: a c of NaN makes every tau NaN, and a test of nothing but NaN passes.
            const float reach = __builtin_fmaf((float)n1, dlen, (float)fmax(fmax(fabs(e.x), fabs(e.y)), fabs(e.z))) * sc;
            const float off = __uint_as_float(0x7FC00000u);
            const bool far = !(reach <= RT_REACH_GRID);
            rb.cx = far ? off : rb.cx; rb.cy = far ? off : rb.cy; rb.cz = far ? off : rb.cz;
        }
        // a run whose box the ray passes at a distance is covered without a look at its records
        open = has & box_pass(run_box.x, run_box.y, rb);
        c.box += has ? 1u : 0u;
#if !defined(RT_STEP_STATS)
        c.prim += (has & !open) ? obj_count : 0u;
#endif
        T.k = open ? obj_begin : T.k; T.kend = open ? obj_begin + obj_count : T.kend;
        T.bt0 = open ? n0 : T.bt0; T.tempt = open ? n1 : T.tempt; T.bprim = open ? RT_MISS : T.bprim;
        T.leaf = open ? child_base == 0 : T.leaf;
        if (open) {
            if (nd_pad & 1u) run_bt1[tid] = n1;             // voxel records read it back
            rayf[1][tid] = make_float4(rb.cx, rb.cy, rb.cz, dlen);
            rayf[0][tid].w = rb.tend;
        }
    }
    mode = !in ? mode : !alive ? TM_OBJ : open ? TM_TRI : TM_POP;
}

// ------------------------------------------------------------------ the cooperative triangle phase
// OctreeNode::hit (octree.cpp:19-38) for every lane of the wave that has a run open (an "owner"),
// done by the whole wave at once instead of by each lane for itself:
//   1. segments: every owner cuts its run into segments of RT_SEG records and lists the next ones in
//      the wave's raw table in LDS (a prefix sum over the owners' segment counts gives each its
//      place), at most one segment per lane of the wave; a run with a whole chunk of records or more
//      left gets the round to itself, its segments named by all lanes at once.  Every lane then takes
//      one listed segment, fetches the segment's box and its owner's ray and makes the box pre-test:
//      segments that pass go to the wave's work table (ballot + count of the set bits below the
//      lane), the others are done with: none of their records is looked at.  Rounds
//      repeat until the work table holds four chunks' worth or the owners have nothing left;
//   2. pre-test: a chunk = 64 * RT_LANE_RECS / RT_SEG consecutive segments of the work table; every
//      lane takes RT_LANE_RECS consecutive records of its segment, tests their boxes against the
//      segment owner's ray (read from LDS, where the owner left it when the run opened) and appends
//      (record, owner) for those that pass to the wave's candidate list in LDS;
//   3. test: every lane takes one candidate, fetches its 80-byte record and the owner's fp64 ray
//      and window (cross-lane reads) and makes the fp64 primitive test -- up to 64 candidates of
//      different owners per pass;
//   4. per owner the pass keeps the smallest hit distance (an LDS minimum over order-preserving
//      integer keys) and, among the candidates that reached it, the one with the largest primitive
//      index (an LDS maximum): OctreeNode::hit accepts a hit with `t <= tempt`, so of equal distances
//      the primitive that comes later in the node's list wins (octree.cpp:31) -- the list is in index
//      order, so that is the larger index, whatever order the records of the run are stored in -- and
//      a hit farther than an earlier one never survives: the same outcome as testing the owner's
//      candidates one after the other in list order.
// The wave's cost follows the number of records near its rays, not its longest run, and the fp64
// tests of all owners share their instructions.
#ifndef RT_CAND_CAP
#define RT_CAND_CAP 128          // candidate slots per wave (a ring; a power of two, at least 64)
#endif
#define RT_SEG RT_SEG_RECS       // records per segment (a power of two, at most 64)
#define RT_LANE_RECS 4           // records a lane pre-tests per chunk (their boxes are two 16-byte loads)
#define RT_SEG_CAP 128           // work table entries per wave, one of them the empty entry: a round adds at most 64 to fewer than 64
#define RT_SEG_ROUND_MAX 15      // most segments one owner lists per round (4 bits in the prefix sum)
struct CoopLds {                 // per wave
    uint64_t seg[RT_SEG_CAP];    // work table: first record | owner lane << 32 | records in the segment << 40; behind the last one an empty entry
    uint32_t crec[RT_CAND_CAP];  // candidate's record index
    uint8_t  cown[RT_CAND_CAP];  // candidate's owner lane
    union {
        struct {
            unsigned long long okey[64]; // per owner lane, per pass: smallest hit distance as an ordered integer
            uint32_t owin[64];           //   primitive id << 1 | sign of a zero distance, the largest among the candidates that reached it
        };
        uint64_t raw[64];        // raw table of a round (entries as in seg); no fp64 pass runs while it is in use
    };
};
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
// the lanes for which p holds, as a mask (the predicate's own lane mask: no 0/1 value is formed)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// acc + the number of set bits of m below this lane
__device__ __forceinline__ uint32_t count_below(unsigned long long m, uint32_t acc)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc));
}
// doubles (no NaN) <-> unsigned integers in the same order
__device__ __forceinline__ unsigned long long ordered_key(double t)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(t);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_value(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

template <int BLOCK>
__device__ __forceinline__ void tri_phase(const DevScene &S, TravState &T, int &mode, const double *run_bt1,
                                          const float4 (*rayf)[BLOCK], CoopLds &L, Counters &c)
{
    static_assert((RT_CAND_CAP & (RT_CAND_CAP - 1)) == 0 && RT_CAND_CAP >= 64, "candidate ring");
    static_assert((RT_SEG & (RT_SEG - 1)) == 0 && RT_SEG <= 64 && RT_SEG <= RT_BOX_SPARE, "segment size");
    // a chunk = RT_LANE_RECS consecutive records per lane, RT_SEG / RT_LANE_RECS lanes to a segment
    constexpr uint32_t LANES_PER_SEG = RT_SEG / RT_LANE_RECS, SEGS_PER_CHUNK = 64u / LANES_PER_SEG;
    static_assert(RT_LANE_RECS == 4 && RT_SEG % RT_LANE_RECS == 0, "a lane's boxes are two 16-byte loads");
    const uint32_t lane = threadIdx.x & 63u, wave0 = threadIdx.x & ~63u;
    const uint32_t sub = (lane % LANES_PER_SEG) * RT_LANE_RECS, sseg = lane / LANES_PER_SEG;     // the lane's place in a chunk
    uint32_t produced = 0, tested = 0;                 // candidates appended / tested so far (wave-uniform)
    uint32_t nbox = 0;                                 // box tests made for segments and records (wave-uniform)
    const bool owner = mode == TM_TRI;
    uint32_t kcur = T.k;                               // first record of the owner's run that is not scheduled yet
    const uint32_t segbase = T.k / RT_SEG + T.cur;     // where the boxes of the owner's segments start (DevScene::seg_box)

#if RT_STEP_STATS == 8 || RT_STEP_STATS == 9
    long long tk8 = clock64();
#define RT_TICK8(field) { const long long t_ = clock64(); if (lane == 0) c.field += (uint32_t)((t_ - tk8) >> 4); tk8 = t_; }
#else
#define RT_TICK8(field)
#endif
#if RT_STEP_STATS == 9
#undef RT_TICK8
#define RT_TICK8(field)
#define RT_TICK9(field) { const long long t_ = clock64(); if (lane == 0) c.field += (uint32_t)((t_ - tk8) >> 4); tk8 = t_; }
#else
#define RT_TICK9(field)
#endif
    // steps 3 and 4 for everything appended so far
    auto test_all = [&]() {
        RT_TICK8(inner)
        while (tested != produced) {
#if RT_STEP_STATS == 8
            const long long tp0 = clock64();
#endif
            const uint32_t left = produced - tested, n = left < 64u ? left : 64u;
            const bool act = lane < n;
            const uint32_t slot = (tested + lane) & (RT_CAND_CAP - 1u);
            L.okey[lane] = ~0ull; L.owin[lane] = 0u;
            const uint32_t rec = act ? L.crec[slot] : 0u;
            const int o = act ? (int)L.cown[slot] : (int)lane;
            TriReg R = {};
            if (act) R = tri_load(S.leaf_tris + rec);
            // the owner's ray and window (all lanes take part in the cross-lane reads)
            const V3 e = mk(shfl_d(T.e.x, o), shfl_d(T.e.y, o), shfl_d(T.e.z, o));
            const V3 d = mk(shfl_d(T.d.x, o), shfl_d(T.d.y, o), shfl_d(T.d.z, o));
            const double bt0 = shfl_d(T.bt0, o), tempt = shfl_d(T.tempt, o);
            const int fast = __shfl((int)T.fast, o, 64);
            bool hit = false;
            double tt = 0;
            uint32_t id = R.id;
            if (act) {
                if (R.id & LEAF_VOXEL) {
                    // BoxNode::calHit, Surface.cpp:43-59: the slab test runs on the node's whole window
                    // [bt0, bt1], the hit is the entry point moved back along the ray by the diagonal
                    // and is kept if it is not beyond the best distance so far (octree.cpp:30-33)
                    const V3 lo = mk(R.v0x, R.v0y, R.v0z), hi = mk(R.e1x, R.e1y, R.e1z);
                    const double bt1 = run_bt1[wave0 + o];
                    double r0, r1;
                    const bool h = fast ? hit_box_fast(lo, hi, e, d, mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z), bt0, bt1, r0, r1)
                                        : hit_box(lo, hi, e, d, bt0, bt1, r0, r1);
                    if (h) { tt = r0 - R.e2x / len(d); hit = tt <= tempt; }
                    id = R.id & ~LEAF_VOXEL;
                } else {
                    hit = tri_hit(R, e, d, bt0, tempt, tt);
                }
#if RT_STEP_STATS == 5
                c.rays += 1;                             // records that reached the fp64 test
#endif
            }
            // LDS operations of one wave are carried out in the order they are issued, so each stage
            // below sees the stage before it; the wave barriers only keep the compiler from mixing them
            // -0.0 and +0.0 are one distance to `t <= tempt` (octree.cpp:31), so they share a key; the
            // winner's own sign of zero travels in the lowest bit of the id word.  Of equal distances the
            // primitive that comes later in the node's list wins, and that is the one with the larger
            // index (the list is in index order; the records of a run are not)
            const unsigned long long key = ordered_key(tt + 0.0);
            const uint32_t idw = (id << 1) | ((tt == 0.0 && __builtin_signbit(tt)) ? 1u : 0u);
#if RT_STEP_STATS == 8
            if (__builtin_amdgcn_readfirstlane((int)(key != 0ull)) >= 0 && lane == 0) c.prim += (uint32_t)((clock64() - tp0) >> 4);
#endif
            __builtin_amdgcn_wave_barrier();
            if (hit) atomicMin(&L.okey[o], key);
            __builtin_amdgcn_wave_barrier();
            const bool best = hit && L.okey[o] == key;
            if (best) atomicMax(&L.owin[o], idw);
            __builtin_amdgcn_wave_barrier();
            if (owner) {
                const unsigned long long k = L.okey[lane];
                if (k != ~0ull) {
                    const double t = ordered_value(k);
                    const uint32_t w = L.owin[lane];
                    if (t < T.tempt || (t == T.tempt && (T.bprim == RT_MISS || (w >> 1) > T.bprim))) {   // octree.cpp:30-33
                        T.tempt = (w & 1u) ? -0.0 : t; T.bprim = w >> 1;
                    }
                }
            }
            tested += n;
#if RT_STEP_STATS == 7
            if (lane == 0) { c.leaf += 1; c.prim += n; }    // fp64 passes and the candidates they test
#endif
        }
        RT_TICK8(leaf)
    };

    auto ray_of = [&](uint32_t o) -> RayB {
        const float4 a_ = rayf[0][wave0 + o], b_ = rayf[1][wave0 + o];
        RayB r;
        r.ix = a_.x; r.iy = a_.y; r.iz = a_.z; r.tend = a_.w; r.cx = b_.x; r.cy = b_.y; r.cz = b_.z;
        return r;
    };
    // ---- step 2: the work table's segments, pre-tested chunk by chunk.  A lane takes RT_LANE_RECS
    // consecutive records of its segment (two 16-byte loads bring their boxes, two LDS reads the
    // owner's ray).  A chunk's segment entry is read from LDS two chunks ahead, its boxes and its
    // owner's ray are requested one chunk ahead of their use, unconditionally (past the last segment:
    // the empty entry), so that nothing a chunk's tests wait for was requested in the same iteration.
    // Nothing else is loaded inside the loop: when the candidate ring cannot take a chunk's
    // candidates the loop is left, the ring emptied by an fp64 pass, and the loop entered again.
    auto run_chunks = [&](const uint32_t total) {
        if (lane == 0) L.seg[total] = 0ull;                                         // what a chunk reads past the last segment: no records
        const uint32_t nchunk = (total + SEGS_PER_CHUNK - 1u) / SEGS_PER_CHUNK;
        RT_TICK8(rays)
        auto entry = [&](uint32_t ch) -> unsigned long long {
            const uint32_t sidx = ch * SEGS_PER_CHUNK + sseg;
            return L.seg[sidx < total ? sidx : total];
        };
        struct Boxes { uint4 a, b; };
        auto boxes = [&](unsigned long long e) -> Boxes {
            Boxes v;
            __builtin_memcpy(&v, S.leaf_box + 2ull * ((uint32_t)e + sub), 32);       // aligned to a box's size only
            return v;
        };
        // Candidates in list order: lane by lane, a lane's records in order.  Of a chunk's candidates
        // those at places [from, from + take) are appended (all of them, unless the chunk has more
        // than the ring holds).
        auto append = [&](const unsigned long long (&m)[RT_LANE_RECS], const bool (&p)[RT_LANE_RECS], unsigned long long e0) {
            uint32_t place = produced;
#pragma unroll
            for (int r = 0; r < RT_LANE_RECS; r++) place = count_below(m[r], place);
#pragma unroll
            for (int r = 0; r < RT_LANE_RECS; r++) {
                if (p[r]) {
                    const uint32_t slot = place & (RT_CAND_CAP - 1u);
                    L.crec[slot] = (uint32_t)e0 + sub + (uint32_t)r; L.cown[slot] = (uint8_t)((uint32_t)(e0 >> 32) & 0xFFu);
                    place++;
                }
            }
        };
        // the same for the places [from, from + take) of a chunk that has more candidates than the ring takes at once
        auto append_part = [&](const unsigned long long (&m)[RT_LANE_RECS], unsigned long long e0, uint32_t from, uint32_t take) {
            uint32_t place = 0;
#pragma unroll
            for (int r = 0; r < RT_LANE_RECS; r++) place = count_below(m[r], place);
#pragma unroll
            for (int r = 0; r < RT_LANE_RECS; r++) {
                if ((m[r] >> lane) & 1ull) {
                    if (place - from < take) {
                        const uint32_t slot = (produced + place - from) & (RT_CAND_CAP - 1u);
                        L.crec[slot] = (uint32_t)e0 + sub + (uint32_t)r; L.cown[slot] = (uint8_t)((uint32_t)(e0 >> 32) & 0xFFu);
                    }
                    place++;
                }
            }
        };
        uint32_t ch = 0;
        while (ch < nchunk) {
            // the chunk in hand, and the next one: its entry has been read, its boxes are requested
            // at the top of the iteration before its own (only while there is a next chunk: a load
            // that nobody waits for would hold up whoever reuses its register)
            unsigned long long e0 = entry(ch), e1 = entry(ch + 1u), held_m[RT_LANE_RECS], held_e = 0ull;
            Boxes bx0 = boxes(e0), bx1 = bx0;
            RayB rf0 = ray_of((uint32_t)(e0 >> 32) & 0xFFu);
            bool held = false;
            uint32_t held_n = 0;
            for (; ch < nchunk; ch++) {
                if (ch + 1u < nchunk) bx1 = boxes(e1);
                const unsigned long long e2 = entry(ch + 2u);
                const uint32_t cnt = (uint32_t)(e0 >> 40);
                const RayB rf1 = ray_of((uint32_t)(e1 >> 32) & 0xFFu);           // the next chunk's ray
                const RayB rf = rf0;
                const uint32_t w[RT_LANE_RECS][2] = { { bx0.a.x, bx0.a.y }, { bx0.a.z, bx0.a.w }, { bx0.b.x, bx0.b.y }, { bx0.b.z, bx0.b.w } };
                unsigned long long m[RT_LANE_RECS];
                bool p[RT_LANE_RECS];
                uint32_t np = 0;
#pragma unroll
                for (int r = 0; r < RT_LANE_RECS; r++) {
                    const bool in_seg = sub + (uint32_t)r < cnt, near = box_pass(w[r][0], w[r][1], rf);
                    p[r] = in_seg & near;
                    const unsigned long long im = ballot64(in_seg);
                    m[r] = im & ballot64(near);                                   // two compare masks and a scalar AND
                    np += (uint32_t)__popcll(m[r]);
                    nbox += (uint32_t)__popcll(im);
                }
#if RT_STEP_STATS == 7
                if (lane == 0) c.inner += 1;                    // chunks
#endif
                if (produced - tested + np > (uint32_t)RT_CAND_CAP) {
                    held = true; held_e = e0; held_n = np;
#pragma unroll
                    for (int r = 0; r < RT_LANE_RECS; r++) held_m[r] = m[r];
                    ch++;
                    break;
                }
                append(m, p, e0);
                produced += np;
                e0 = e1; e1 = e2; bx0 = bx1; rf0 = rf1;
            }
            if (held) {
                // an fp64 pass empties the ring; a chunk with more candidates than the ring holds goes in pieces
                for (uint32_t from = 0; from < held_n; from += RT_CAND_CAP) {
                    test_all();
                    const uint32_t take = held_n - from < (uint32_t)RT_CAND_CAP ? held_n - from : (uint32_t)RT_CAND_CAP;
                    append_part(held_m, held_e, from, take);
                    produced += take;
                }
            }
        }
    };

    // ---- step 1: rounds.  An owner with at least a whole chunk of records left takes the round for
    // itself: its next whole segments are named by all lanes at once (lane i: base + i * RT_SEG).
    // Otherwise every owner lists the next segments of its run in the raw table, each at its place
    // in a prefix sum over the owners' segment counts, and every lane reads one entry back.  Then
    // the segments' own box test; what passes moves on to the work table.
    uint32_t work_n = 0;                               // entries in the work table (wave-uniform)
    for (;;) {
        constexpr uint32_t CHUNK_RECS = 64u * RT_LANE_RECS;
        const uint32_t left = owner ? T.kend - kcur : 0u;
        const unsigned long long big = ballot64(left >= CHUNK_RECS);
        uint32_t total = 0;
        unsigned long long rawe = 0ull;
        if (big != 0ull) {
            const uint32_t solo_o = (uint32_t)__ffsll((long long)big) - 1u;
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)kcur, (int)solo_o);
            total = ((uint32_t)__builtin_amdgcn_readlane((int)T.kend, (int)solo_o) - base) / RT_SEG;
            if (total > 64u) total = 64u;
            if (lane < total) rawe = (unsigned long long)(base + lane * RT_SEG) | ((unsigned long long)solo_o << 32) | ((unsigned long long)RT_SEG << 40);
            if (lane == solo_o) kcur += total * RT_SEG;
        } else {
            uint32_t nseg = (left + (RT_SEG - 1u)) / RT_SEG;
            if (nseg > (uint32_t)RT_SEG_ROUND_MAX) nseg = RT_SEG_ROUND_MAX;
            const unsigned long long b0 = ballot64((nseg & 1u) != 0u), b1 = ballot64((nseg & 2u) != 0u), b2 = ballot64((nseg & 4u) != 0u), b3 = ballot64((nseg & 8u) != 0u);
            uint32_t first = count_below(b0, 0u) + 2u * count_below(b1, 0u) + 4u * count_below(b2, 0u) + 8u * count_below(b3, 0u);
            total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2) + 8u * (uint32_t)__popcll(b3);
            if (total == 0u) break;
            RT_TICK9(rays)
            if (total > 64u) total = 64u;                                            // the rest waits for the next round
            if (first > total) first = total;
            if (nseg > total - first) nseg = total - first;
#pragma unroll 1
            for (uint32_t j = 0; j < (uint32_t)RT_SEG_ROUND_MAX; j++) {
                if (ballot64(j < nseg) == 0ull) break;
                if (j < nseg) {
                    const uint32_t r0 = kcur + j * RT_SEG, cnt = T.kend - r0 < (uint32_t)RT_SEG ? T.kend - r0 : (uint32_t)RT_SEG;
                    L.raw[first + j] = (unsigned long long)r0 | ((unsigned long long)lane << 32) | ((unsigned long long)cnt << 40);
                }
            }
            if (nseg) { const uint32_t adv = nseg * RT_SEG; kcur = T.kend - kcur < adv ? T.kend : kcur + adv; }
            __builtin_amdgcn_wave_barrier();
            if (lane < total) rawe = L.raw[lane];
            __builtin_amdgcn_wave_barrier();
            RT_TICK9(inner)
        }
        {
            // the segment's box against its owner's ray
            const bool have = lane < total;
            const uint32_t so = (uint32_t)(rawe >> 32) & 0xFFu;
            const uint32_t obase = (uint32_t)__shfl((int)segbase, (int)so, 64), ok = (uint32_t)__shfl((int)T.k, (int)so, 64);
            const uint32_t sidx = have ? obase + ((uint32_t)rawe - ok) / RT_SEG : 0u;
            uint2 sb;
            __builtin_memcpy(&sb, S.seg_box + 2ull * sidx, 8);
            const RayB rf = ray_of(so);
            const bool pass = have && box_pass(sb.x, sb.y, rf);
            const unsigned long long pm = ballot64(pass);
            if (pass) L.seg[count_below(pm, work_n)] = rawe;
            work_n += (uint32_t)__popcll(pm);
            nbox += total;
#if RT_STEP_STATS == 10
            if (lane == 0) { c.rays += 1; c.inner += total; c.leaf += (uint32_t)__popcll(pm); }      // rounds, segments listed, segments that pass
#endif
        }
        RT_TICK9(leaf)
        if (work_n >= 64u) {
#if RT_STEP_STATS == 10
            if (lane == 0) c.prim += 1;                         // passes over the work table
#endif
            run_chunks(work_n); work_n = 0;
            RT_TICK9(prim)
        }
    }
    if (work_n) {
#if RT_STEP_STATS == 10
        if (lane == 0) c.prim += 1;
#endif
        run_chunks(work_n);
    }
    RT_TICK9(prim)
    test_all();
    RT_TICK8(rays)
    RT_TICK9(prim)
#if !defined(RT_STEP_STATS)
    if (lane == 0) { c.box += nbox; c.fp64 += tested; }
#endif
    if (owner) {
#if RT_STEP_STATS != 1 && RT_STEP_STATS != 3 && RT_STEP_STATS != 4 && RT_STEP_STATS != 6 && RT_STEP_STATS != 7 && RT_STEP_STATS != 8 && RT_STEP_STATS != 9 && RT_STEP_STATS != 10 && RT_STEP_STATS != 12 && RT_STEP_STATS != 13
        c.prim += T.kend - T.k;
#endif
        bool improved = false;
        if (T.bprim != RT_MISS && T.tempt < T.mint) { T.mint = T.tempt; T.minprim = T.bprim; improved = true; }   // octree.cpp:320-325
        mode = (T.leaf && improved) ? TM_OBJ : TM_POP;                                                          // octree.cpp:326-329
    }
}
// OctreeNode::hit for a one-leaf mesh inside the object sweep (every lane walks the same list)
__device__ __forceinline__ void tri_test(const TriReg &tr, TravState &T)
{
    double tt;
    if (tri_hit(tr, T.e, T.d, T.bt0, T.tempt, tt)) { T.tempt = tt; T.bprim = tr.id; }   // last equal wins, octree.cpp:31
}

// Src supplies the rays of one launch:
//   uint32_t count() const;
//   bool load(uint32_t item, V3 &e, V3 &d, double &t0, double &t1) const;  // false: nothing to trace
//   void reload(uint32_t item, V3 &e, V3 &d) const;                        // the same e, d again
//   void store(uint32_t item, bool hit, const Hit &h) const;
// `next` is a zero-initialised device counter shared by the launch; batch_max (a multiple of 64)
// bounds how many consecutive rays a wave reserves at a time: large batches spare the counter
// when rays are cheap, small ones keep the rays in flight close together (cache footprint).
//
// The step kinds are nested loops (object phase outermost, triangle phase innermost) so that the
// state a phase does not touch stays put in its registers while that phase repeats.
// CHAIN: lanes go from one octree straight on to the next object's (see `chain` below; scenes with several
// octrees in a row); it never changes a result, and kernels compiled without it are leaner.
template <int BLOCK, bool CHAIN = false, class Src>
__device__ inline void trace_engine(const DevScene &S, const Src &src, uint32_t *next, uint32_t batch_max,
                                    uint2 (*stk)[BLOCK], Counters &c)
{
    const int tid = threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = src.count();
    __shared__ double run_bt1[BLOCK];        // upper end of the window of the lane's current primitive run
    // the lane's ray for the box pre-test (see RayB), read back by the triangle phase: 1 / direction and
    // |d| are formed whenever T.d changes, the entry point and the window's length when a run opens
    // (node_step).  rayf[0] = {inv.x, inv.y, inv.z, tend}, rayf[1] = {c.x, c.y, c.z, |d|}
    __shared__ float4 rayf[2][BLOCK];
#define RT_RAYF_PUT() { const RayInv q_ = ray_single(T.d); rayf[0][tid] = make_float4(q_.ix, q_.iy, q_.iz, 0.0f); rayf[1][tid] = make_float4(0.0f, 0.0f, 0.0f, q_.len); }
    __shared__ CoopLds coop[BLOCK / 64];      // the wave's candidate list of the cooperative triangle phase
    // Three blocks of 256 lanes share a CU's 160 KB only while a block asks for no more than 53760
    // bytes (the allocation is rounded up in steps that 53760 exactly fills): 32 bytes more and a
    // CU holds two blocks, measured as 9 % (benchmark frame) to 17 % (atrium) more time.
    static_assert(BLOCK != 256 || sizeof(CoopLds) * (BLOCK / 64) + (sizeof(double) + 2 * sizeof(float4) + 2 * RT_STACK_LEVELS * sizeof(uint32_t)) * BLOCK <= 53760,
                  "the closest-hit engine's LDS must leave room for three blocks per CU");
    TravState T = {};
    int mode = TM_FETCH;
    // rays are handed to a wave in batches of 64 .. batch_max: a quarter of an even share of what is left
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    uint32_t batch = total / (nwaves * 4u);
    batch = batch < 64u ? 64u : (batch > batch_max ? batch_max : (batch + 63u) & ~63u);
    uint32_t wave_base = 0, wave_end = 0;
    bool drained = false;
#if RT_STEP_STATS == 4
    // clock ticks (/16) per phase, lane 0 of every wave: rays = object phases, inner = node steps,
    // leaf = triangle steps, prim = the wave's whole life
    long long tk0 = clock64();
    const long long tk_start = tk0;
#define RT_TICK(field) { const long long tk1 = clock64(); if (lane == 0) c.field += (uint32_t)((tk1 - tk0) >> 4); tk0 = tk1; }
#elif RT_STEP_STATS == 13
    // rays = whole node steps (vote and pop included); node_step itself adds the wait for the node record and the rest
    long long tk0 = clock64();
#define RT_TICK(field) { const long long tk1 = clock64(); if (lane == 0 && #field[0] == 'i') c.rays += (uint32_t)((tk1 - tk0) >> 4); tk0 = tk1; }
#else
#define RT_TICK(field)
#endif
    // The next node of the lane's depth-first walk: the next child, in sorted order, of the deepest
    // open node that has one left (octree.cpp:349-368); with none left the octree is finished.
    auto pop_now = [&]() {
        // a level leaves the stack together with its last child, so the top level always has a child
        // left: one LDS read per pop, no unwinding through exhausted levels
        if (T.sp == 0) { mode = TM_OBJ; return; }
        const uint2 top = stk[T.sp - 1][tid];                      // {first child, order word}
        const uint32_t w = top.y;
        const uint32_t cnt = (w >> 24) & 15u, pos = w >> 28;
        T.cur = top.x + ((w >> (3 * pos)) & 7u);
        if (pos + 1u == cnt) T.sp--;
        else stk[T.sp - 1][tid].y = w + (1u << 28);
        mode = TM_NODE;
    };
    // A lane whose octree is finished and whose next object is an octree with inner nodes as well goes
    // straight on to that octree's root, without waiting for an object phase: Octree::hit's verdict
    // (octree.cpp:371-375), then the next child of the group (Surface.cpp:160-181).
    auto chain = [&]() {
        if (T.oi & RT_OI_LOCAL) return;
        const uint32_t nr = S.objects[T.oi].next_root;
        if (nr == RT_NO_CHAIN) return;
        if (T.minprim != RT_MISS && T.mint <= T.t1) { T.t1 = T.mint; T.hobj = T.oi; T.hprim = T.minprim; }
        T.oi++;
        T.root = nr; T.cur = nr; T.sp = 0; T.mint = T.t1 + 1; T.minprim = RT_MISS; mode = TM_NODE;
    };
    for (;;) {
        // ---------------------------------------------------------------- object phase
        // Group::hit, Surface.cpp:160-181.  Lanes in OBJ mode have just finished the octree of
        // object T.oi; lanes in FETCH mode need a ray.  The phase hands out rays and then sweeps the
        // object table once, object by object, all lanes that stand at that object taking part
        // (a new ray starts at object 0, so within a sweep every lane moves forward until it
        // reaches an octree or the end of the table).
#if RT_STEP_STATS == 1
        c.rays += 1;
#elif RT_STEP_STATS == 2
        if (mode == TM_FETCH || mode == TM_OBJ) c.rays += 1;
#elif RT_STEP_STATS == 6
        {   // object phases and the lanes that take part in them; node steps and their lanes below
            const uint32_t sO = (uint32_t)__popcll(__ballot(mode == TM_OBJ || mode == TM_FETCH));
            if (lane == 0) { c.leaf += 1; c.prim += sO; }
        }
#endif
        if (mode == TM_OBJ) {
            if (T.oi & RT_OI_LOCAL) {
                // back from an implicit surface's voxel tree: the ray returns to the scene's frame
                T.oi &= ~RT_OI_LOCAL;
                src.reload(T.item, T.e, T.d);
                T.fast = S.regular && dir_is_regular(T.d) && org_is_regular(T.e);
                T.r = T.fast ? mk(1.0 / T.d.x, 1.0 / T.d.y, 1.0 / T.d.z) : mk(0, 0, 0);
                RT_RAYF_PUT()
            }
            // Octree::hit's verdict (octree.cpp:371-375)
            if (T.minprim != RT_MISS && T.mint <= T.t1) { T.t1 = T.mint; T.hobj = T.oi; T.hprim = T.minprim; }
            T.oi++;
        }
        {
            const unsigned long long need = __ballot(mode == TM_FETCH);
            const uint32_t need_n = (uint32_t)__popcll(need), left = wave_end - wave_base;
            uint32_t nb_base = wave_end, nb_end = wave_end;
            if (need_n > left && !drained) {
                // the wave's batch cannot serve every free lane: reserve another one (one atomic per
                // batch, not per step: a counter that every wave of the launch increments
                // serialises them).  Batches shrink as the launch runs out of rays, so that the
                // waves finish together.
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(next, batch);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                nb_base = b < total ? b : total;
                nb_end = (total - nb_base) < batch ? total : nb_base + batch;
                drained = nb_end == total;
                batch = (total - nb_end) / (nwaves * 4u);
                batch = batch < 64u ? 64u : (batch > batch_max ? batch_max : (batch + 63u) & ~63u);
            }
            if (mode == TM_FETCH) {
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                const uint32_t item = rank < left ? wave_base + rank : nb_base + (rank - left);
                if (rank >= left && item >= nb_end) {
                    if (drained) mode = TM_EXIT;            // else: stays in FETCH until the next batch
                } else {
                    T.item = item;
                    T.hobj = -1; T.hprim = 0; T.oi = 0;
                    if (src.load(item, T.e, T.d, T.t0, T.t1)) {
#ifndef RT_STEP_STATS
                        c.rays++;
#endif
                        T.fast = S.regular && dir_is_regular(T.d) && org_is_regular(T.e);
                        T.r = T.fast ? mk(1.0 / T.d.x, 1.0 / T.d.y, 1.0 / T.d.z) : mk(0, 0, 0);
                        RT_RAYF_PUT()
                        mode = TM_OBJ;
                    } else {
                        Hit h; h.t = RT_INFINITE; h.obj = -1; h.prim = 0;
                        src.store(item, false, h);          // stays in FETCH
                    }
                }
            }
            if (need_n > left) {
                const uint32_t over = need_n - left, nb = nb_end - nb_base;
                wave_base = nb_base + (over < nb ? over : nb);
                wave_end = nb_end;
            } else {
                wave_base += need_n;
            }
            // Scalar loads are issued well before their values are needed: the 16-byte head of
            // the next object while this one is processed, a one-leaf mesh's first triangle
            // together with its box, every further triangle while the previous one is tested.
            const u_object objects = (u_object)(unsigned long long)S.objects;
            struct Head { uint32_t x, y, z, w; };                                   // {type, ntri, root, leaf_begin}
            Head head_next = { (uint32_t)objects[0].type, objects[0].ntri, objects[0].root, objects[0].leaf_begin };
            for (int o = 0; o < S.nobjects; o++) {
                const Head head = head_next;
                const int on = o + 1 < S.nobjects ? o + 1 : o;
                head_next.x = (uint32_t)objects[on].type; head_next.y = objects[on].ntri; head_next.z = objects[on].root; head_next.w = objects[on].leaf_begin;
                const bool mine = mode == TM_OBJ && T.oi == o;
                if (__ballot(mine) == 0ull) continue;
                const int otype = (int)head.x;
                if (otype == RT_OBJ_MESH) {
                    const uint32_t ntri = head.y, root = head.z;
                    if (!ntri) {
                        // an octree with inner nodes: leave the sweep here (octree.cpp:312-313)
                        if (mine) { T.root = root; T.cur = root; T.sp = 0; T.mint = T.t1 + 1; T.minprim = RT_MISS; mode = TM_NODE; }
                        continue;
                    }
                    // The whole octree is one leaf of ntri triangles: Octree::hit (octree.cpp:285-376)
                    // reduces to the root's slab test and OctreeNode::hit on it, and all lanes walk
                    // the same triangle list.
                    const u_leaftri P = (u_leaftri)(unsigned long long)(S.leaf_tris + head.w);
                    TriReg tr_next = tri_load_uniform(P);
                    const V3 lo = mk(objects[o].lo[0], objects[o].lo[1], objects[o].lo[2]), hi = mk(objects[o].hi[0], objects[o].hi[1], objects[o].hi[2]);
                    bool hb = false;
                    if (mine) {
                        double n0, n1;
                        hb = T.fast ? hit_box_fast(lo, hi, T.e, T.d, T.r, T.t0, T.t1, n0, n1) : hit_box(lo, hi, T.e, T.d, T.t0, T.t1, n0, n1);
                        if (hb) {
#if !defined(RT_STEP_STATS) || RT_STEP_STATS < 3
                            c.leaf++; c.prim += ntri; c.fp64 += ntri;
#endif
                            T.bt0 = n0; T.tempt = n1; T.bprim = RT_MISS;
                        }
                    }
                    if (__ballot(hb) != 0ull) {
#pragma unroll 1
                        for (uint32_t k = 0; k < ntri; k++) {
                            const TriReg tr = tr_next;
                            tr_next = tri_load_uniform(P + (k + 1 < ntri ? k + 1 : k));
                            if (hb) tri_test(tr, T);
                        }
                        if (hb) {
                            double mint = T.t1 + 1;                                         // octree.cpp:313
                            if (T.bprim != RT_MISS && T.tempt < mint) {                     // octree.cpp:320-325
                                mint = T.tempt;
                                if (mint <= T.t1) { T.t1 = mint; T.hobj = o; T.hprim = T.bprim; }   // octree.cpp:371-375
                            }
                        }
                    }
                } else if (otype == RT_OBJ_HEART || otype == RT_OBJ_GENUS2) {
                    // HeartShape::hit (Surface.cpp:268-276), ImplicitSurface::hit (Surface.h:308-316):
                    // the voxel tree is traversed with the ray taken into the surface's frame; t is
                    // the same parameter in both frames.  The lane leaves the sweep here.
                    if (mine) {
                        const double inv = 1.0 / objects[o].p[3];
                        T.e = (T.e - mk(objects[o].p[0], objects[o].p[1], objects[o].p[2])) * inv;
                        T.d = T.d * inv;
                        T.fast = S.regular && dir_is_regular(T.d) && org_is_regular(T.e);
                        T.r = T.fast ? mk(1.0 / T.d.x, 1.0 / T.d.y, 1.0 / T.d.z) : mk(0, 0, 0);
                        RT_RAYF_PUT()
                        T.oi = o | RT_OI_LOCAL;
                        T.root = head.z; T.cur = head.z; T.sp = 0; T.mint = T.t1 + 1; T.minprim = RT_MISS; mode = TM_NODE;
                    }
                    continue;
                } else if (mine) {
                    const double p0 = objects[o].p[0], p1 = objects[o].p[1], p2 = objects[o].p[2], p3 = objects[o].p[3];
                    double t;
                    const bool hit = (otype == RT_OBJ_SPHERE) ? sphere_hit(mk(p0, p1, p2), p3, T.e, T.d, T.t0, T.t1, t)
                                                              : plane_hit(otype, p0, T.e, T.d, T.t0, T.t1, t);
                    if (hit) { T.t1 = t; T.hobj = o; T.hprim = 0; }
                }
                if (mine) T.oi = o + 1;
            }
            {
                // past the last object: the ray is done
                const bool done = mode == TM_OBJ;
                Hit h; h.t = T.t1; h.obj = T.hobj; h.prim = T.hprim;
                src.store_wave(done, T.item, T.hobj >= 0, h);
                if (done) mode = TM_FETCH;
            }
        }
        RT_TICK(rays)
        // ------------------------------------------- node and triangle steps until the next object step
        bool finished = false;
        for (;;) {
            const unsigned long long mO = __ballot(mode == TM_FETCH || mode == TM_OBJ);
            const unsigned long long mN = __ballot(mode == TM_NODE || mode == TM_POP);
            const unsigned long long mT = __ballot(mode == TM_TRI);
            if ((mO | mN | mT) == 0ull) { finished = true; break; }
            const int nO = __popcll(mO) * RT_OBJ_WEIGHT(S), nN = __popcll(mN) * RT_WN, nT = __popcll(mT) * RT_WT;
            if (nO != 0 && nO >= nN && nO >= nT) break;
            if (nN != 0 && nN >= nT) {
                // -------------------------------------------------------- node step
#if RT_STEP_STATS == 1
                c.inner += 1;
#elif RT_STEP_STATS == 2
                if (mode == TM_NODE || mode == TM_POP) c.inner += 1;
#elif RT_STEP_STATS == 6
                if (lane == 0) { c.rays += 1; c.inner += (uint32_t)__popcll(mN); }
#endif
                if (mode == TM_POP) pop_now();
                const bool in = mode == TM_NODE;
#if RT_STEP_STATS == 13
                { const long long tv_ = clock64(); if (lane == 0) c.prim += (uint32_t)((tv_ - tk0) >> 4); }   // prim = the vote and the pop
#endif
                if (__all(!in || T.fast)) node_step<true, BLOCK>(S, T, mode, in, stk, run_bt1, rayf, c);
                else node_step<false, BLOCK>(S, T, mode, in, stk, run_bt1, rayf, c);
                if (CHAIN && ((mN >> lane) & 1ull) && mode == TM_OBJ) chain();
                RT_TICK(inner)
                continue;
            }
            // ------------------------------------------------------------ triangle phase
#if RT_STEP_STATS == 1
            c.prim += 1;
#elif RT_STEP_STATS == 7
            if (lane == 0) c.rays += 1;                         // triangle phases
#elif RT_STEP_STATS == 3
            {   // triangle phases, and the lanes in each state when they start
                const uint32_t sN = (uint32_t)__popcll(mN), sO = (uint32_t)__popcll(mO);
                if (lane == 0) { c.rays += 1; c.inner += (uint32_t)__popcll(mT); c.leaf += sN; c.prim += sO; }
            }
#endif
            tri_phase<BLOCK>(S, T, mode, run_bt1, rayf, coop[tid >> 6], c);
            if (CHAIN && ((mT >> lane) & 1ull) && mode == TM_OBJ) chain();
            RT_TICK(leaf)
        }
        if (finished) break;
    }
#if RT_STEP_STATS == 4
    if (lane == 0) c.prim += (uint32_t)((clock64() - tk_start) >> 4);
#endif
}

// Drawable::doAfterHit -> getRawNormal (Surface.cpp:228-231, 68-71; Surface.h:184,197,211);
// a mesh hit takes the hit triangle's geometric normal (normalize(e1 ^ e2), formed once at upload).
// ------------------------------------------------------------------ Perlin materials
// algorithm::perlinNoise, algorithms.h:56-87
// (a real call, not inlined: the Perlin materials are rare, and their code must not weigh on the
// register budget of the kernels that shade every other material)
__device__ __noinline__ double perlin_noise_call(const int32_t *perlin_p, const double *perlin_g, int n, double px, double py, double pz, double scale)
{
    const double x = px / scale, y = py / scale, z = pz / scale;
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    double result = 0;
#pragma unroll 1
    for (int c = 0; c < 8; c++) {
        const int i = (c & 1) ? (int)fx : (int)(fx + 1);
        const int j = (c & 2) ? (int)fy : (int)(fy + 1);
        const int k = (c & 4) ? (int)fz : (int)(fz + 1);
        const double u = x - i, v = y - j, w = z - k;
        int m = k % n;               m = perlin_p[m >= 0 ? m : n + m];
        m = (j + m) % n;             m = perlin_p[m >= 0 ? m : n + m];
        m = (i + m) % n;             m = perlin_p[m >= 0 ? m : n + m];
        const double *g = perlin_g + 3 * m;
        const double au = fabs(u), av = fabs(v), aw = fabs(w);
        const double wu = au < 1 ? 2 * au * au * au - 3 * au * au + 1 : 0.0;
        const double wv = av < 1 ? 2 * av * av * av - 3 * av * av + 1 : 0.0;
        const double ww = aw < 1 ? 2 * aw * aw * aw - 3 * aw * aw + 1 : 0.0;
        result += wu * wv * ww * (g[0] * u + g[1] * v + g[2] * w);
    }
    return result;
}
__device__ inline double perlin_noise(const DevScene &S, V3 p, double scale) { return perlin_noise_call(S.perlin_p, S.perlin_g, S.perlin_n, p.x, p.y, p.z, scale); }
__device__ inline double perlin_normalized(const DevScene &S, V3 p, double scale) { return perlin_noise(S, p, scale) / 2 + 0.5; }      // algorithms.cpp:37-40
__device__ inline double perlin_line(const DevScene &S, V3 p, double scale) { return smin(fabs(perlin_noise(S, p, scale)) * 11, 1.0); }  // algorithms.cpp:32-35
// PerlinNoiseNormalMaterial_Blood::getNormal (material.cpp:73-87) at the position the material
// was given by Node::calHit (Surface.h:277)
__device__ inline V3 blood_normal(const DevScene &S, const rt_material &m, V3 pos, V3 normal)
{
    const double linenoise = perlin_line(S, pos, m.perlin_scale);
    V3 vec = mk(0, 0, 1);
    if (iszero(cross(normal, vec))) vec = mk(1, 0, 0);
    const V3 b = norm(cross(normal, vec));
    const V3 t = norm(cross(b, normal));
    const double dpb = linenoise - perlin_line(S, pos + t * 0.001, m.perlin_scale);
    const double dpt = linenoise - perlin_line(S, pos + b * 0.001, m.perlin_scale);
    return norm((normal + (b * dpb) * 60) + (t * dpt) * 60);
}
// getAmbient / getDiffuse / getSpecular of the Perlin materials (material.h:104-108, 128-157)
__device__ inline void perlin_colours(const DevScene &S, const rt_material &m, V3 pos, V3 &amb, V3 &dif, V3 &spe)
{
    if (m.perlin == RT_PERLIN_NORMAL) {
        const double noise = perlin_normalized(S, pos, m.perlin_scale);
        amb = amb * noise; dif = dif * noise; spe = spe * noise;
    } else {
        const double noise = smin(perlin_normalized(S, pos, m.perlin_scale) * 1.5, 1.0);     // material.cpp:67-71
        const double linenoise = perlin_line(S, pos, m.perlin_scale);
        const double hf1 = smin(1.0, smax(0.0, smax(linenoise - 0.6, 0.6 - linenoise) * 2.5 - 0.05));
        const double hf2 = smin(1.0, smax(0.0, smax(noise - 0.4, 0.4 - noise) * 5 - 0.05));
        const double hf = hf1 * hf2;
        const double high = (1 - linenoise) * 1.7, nz = noise / 2.5;
        const V3 centre = mk(0.45, 0, 0), veins = centre * (high + nz);
        amb = ((amb * (1 - nz) + veins) * hf) * 0.35;
        dif = ((dif * (1 - nz) + veins) * hf) * 1.0;
        // getSpecular is written without parentheses around its last `&`: (S&n + C&(1-n)) & (hf*0.85)
        spe = (spe * noise + centre * (1 - noise)) * (hf * 0.85);
    }
}

// HeartShape::getRawNormal (Surface.cpp:239-265) and SurfaceOfGenus2Function::getNormal
// (Surface.cpp:333-343) at a point of the surface's own frame (y and z swapped as there)
__device__ inline V3 implicit_normal(int type, V3 pos)
{
    const double x = pos.x, y = pos.z, z = pos.y;
    V3 n;
    if (type == RT_OBJ_HEART) {
        double temp = x * x + y * y * 9.0 / 4 + z * z - 1;
        temp = 3 * temp * temp;
        temp = smax(temp, 0.5);
        n.x = -2 * x * z * z * z + temp * 2 * x;
        n.z = -9.0 / 40 * y * z * z * z + temp * 9.0 / 2 * y;
        n.y = -3 * x * x * z * z - 27.0 / 80 * y * y * z * z + 2 * temp * z;
    } else {
        n.x = -12 * x * y * (1 - z * z) + 4 * (x * x + y * y) * x;
        n.z = 6 * (1 - z * z) * (y * y - x * x) + 4 * (x * x + y * y) * y;
        n.y = 2 * y * (y * y - 3 * x * x) * -2 * z + 36 * z * z * z - 20 * z;
    }
    return norm(n);
}
__device__ inline V3 raw_normal(const DevScene &S, const Hit &h, V3 p)
{
    const DevObject &o = S.objects[h.obj];
    if (o.type == RT_OBJ_SPHERE) return norm(p - mk(o.p[0], o.p[1], o.p[2]));
    if (o.type == RT_OBJ_HEART || o.type == RT_OBJ_GENUS2) return implicit_normal(o.type, (p - mk(o.p[0], o.p[1], o.p[2])) * (1.0 / o.p[3]));
    if (o.type == RT_OBJ_GROUND) return mk(0, 1, 0);
    if (o.type == RT_OBJ_WALL_Z) return mk(0, 0, 1);
    if (o.type == RT_OBJ_WALL_X) return mk(1, 0, 0);
    return ld3(S.tri_normals + 3ull * h.prim);
}
// Node::getNormal, Surface.cpp:15-23: the material may adjust the raw normal
__device__ inline V3 hit_normal(const DevScene &S, const Hit &h, V3 p)
{
    const V3 n = raw_normal(S, h, p);
    const rt_material &m = S.materials[S.objects[h.obj].material];
    return m.perlin == RT_PERLIN_BLOOD ? blood_normal(S, m, p, n) : n;
}
__device__ __forceinline__ uint32_t global_prim(const DevScene &S, const Hit &h)
{
    const DevObject &o = S.objects[h.obj];
    if (o.type == RT_OBJ_HEART || o.type == RT_OBJ_GENUS2) return o.prim_base + h.prim;      // voxel index
    return o.type == RT_OBJ_MESH ? o.prim_base + (h.prim - o.tri_base) : o.prim_base;
}

// Camera::getRay(float s, float t), camera.cpp:24-31
__device__ __forceinline__ void camera_ray(const rt_camera &c, float s, float t, V3 &e, V3 &d)
{
    e = ld3(c.e);
    V3 lookat = ld3(c.llc) + (ld3(c.u) * (double)s) * c.width + (ld3(c.v) * (double)t) * c.height;
    d = norm(lookat - e);
}

// DepthCamera::getRay, camera.cpp:62-72: the origin moves on the lens disk (randFromDisk,
// algorithms.h:30-35, drawn from the sample's own key), the film point stays
#define KEY_LENS 0x4c454e53ULL
__device__ inline void camera_ray_lens(const rt_camera &c, float s, float t, uint64_t root_key, V3 &e, V3 &d)
{
    const uint64_t key = hash2(root_key, KEY_LENS);
    double x, y;
    uint32_t a = 0;
    do {
        x = (draw(key, SLOT_DISK + 2 * a) - 0.5) * 2;
        y = (draw(key, SLOT_DISK + 2 * a + 1) - 0.5) * 2;
        a++;
    } while (x * x + y * y >= 1.0);
    e = (ld3(c.e) + ld3(c.u) * (x * c.aperture)) + ld3(c.v) * (y * c.aperture);
    V3 lookat = ld3(c.llc) + (ld3(c.u) * (double)s) * c.width + (ld3(c.v) * (double)t) * c.height;
    d = norm(lookat - e);
}

__device__ inline void flush_counters(uint64_t *counters, uint64_t *kstats, const Counters &c)
{
    if (!counters && !kstats) return;
    // one atomic per wave and counter: wave-reduce with cross-lane adds first.  `counters` = the caller's
    // four work counters, `kstats` = the launch's own block of RT_KSTAT_WORDS (profiling only)
    uint64_t v[6] = { c.rays, c.inner, c.leaf, c.prim, c.box, c.fp64 };
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k >= 4 && !kstats) break;
        uint64_t x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) {
            if (counters && k < 4) atomicAdd((unsigned long long *)&counters[k], (unsigned long long)x);
            if (kstats) atomicAdd((unsigned long long *)&kstats[k], (unsigned long long)x);
        }
    }
}

// Pixel enumeration: 8 x 8 tiles over the part of the image that whole tiles cover, then the
// right and top remainders row by row (a bijection on [0, W*H) for any size).
__device__ __forceinline__ void work_pixel(uint32_t idx, int W, int H, int &x, int &y)
{
    const uint32_t Wt = (uint32_t)W & ~7u, Ht = (uint32_t)H & ~7u;
    const uint32_t tiled = Wt * Ht;
    if (idx < tiled) {
        const uint32_t tile = idx >> 6, l = idx & 63u, tx = Wt >> 3;
        x = (int)((tile % tx) * 8 + (l & 7u));
        y = (int)((tile / tx) * 8 + (l >> 3));
        return;
    }
    idx -= tiled;
    const uint32_t rw = (uint32_t)W - Wt;
    if (idx < rw * (uint32_t)H) { x = (int)(Wt + idx % rw); y = (int)(idx / rw); return; }
    idx -= rw * (uint32_t)H;
    x = (int)(idx % Wt); y = (int)(Ht + idx / Wt);
}

// ------------------------------------------------------------------ closest-hit kernels
struct RaySrc {             // rt_trace_rays: caller's ray array
    DevScene S; const rt_ray *rays; uint32_t n; double t0, t1; uint32_t *prim_id; double *tout; double *normal;
    __device__ uint32_t count() const { return n; }
    __device__ bool load(uint32_t i, V3 &e, V3 &d, double &a, double &b) const { e = ld3(rays[i].e); d = ld3(rays[i].d); a = t0; b = t1; return true; }
    __device__ void reload(uint32_t i, V3 &e, V3 &d) const { e = ld3(rays[i].e); d = ld3(rays[i].d); }
    __device__ void store_wave(bool on, uint32_t i, bool hit, const Hit &h) const { if (on) store(i, hit, h); }
    __device__ void store(uint32_t i, bool hit, const Hit &h) const
    {
        prim_id[i] = hit ? global_prim(S, h) : RT_MISS;
        if (tout) tout[i] = hit ? h.t : RT_INFINITE;
        if (normal) {
            // the hit is parked in the ray's normal slot; k_ray_normals turns it into the normal
            normal[3ull * i] = __longlong_as_double(hit ? (long long)(((uint64_t)(uint32_t)h.obj << 32) | h.prim) : -1ll);
            normal[3ull * i + 1] = h.t;
        }
    }
};
// The normals of rt_trace_rays' hits (Drawable::doAfterHit, Surface.cpp:228-231), from the hits
// that k_trace_rays left in the normal array.
__global__ __launch_bounds__(RT_BLOCK) void k_ray_normals(DevScene S, const rt_ray *rays, uint32_t n, double *normal)
{
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = (uint64_t)__double_as_longlong(normal[3ull * i]);
    V3 nn = mk(0, 0, 0);
    if (id != ~0ull) {
        Hit h; h.t = normal[3ull * i + 1]; h.obj = (int32_t)(id >> 32); h.prim = (uint32_t)id;
        nn = hit_normal(S, h, ld3(rays[i].e) + ld3(rays[i].d) * h.t);
    }
    normal[3ull * i] = nn.x; normal[3ull * i + 1] = nn.y; normal[3ull * i + 2] = nn.z;
}
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace_rays(RaySrc src, uint32_t *next, uint64_t *counters)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK>(src.S, src, next, 128u, stk, c);
    flush_counters(counters, nullptr, c);
}

struct PrimarySrc {         // rt_trace_primary: one ray through every pixel centre
    DevScene S; rt_camera cam; int w, h; uint32_t *prim_id; double *tout;
    __device__ uint32_t count() const { return (uint32_t)w * (uint32_t)h; }
    __device__ bool load(uint32_t i, V3 &e, V3 &d, double &a, double &b) const
    {
        int x, y;
        work_pixel(i, w, h, x, y);
        camera_ray(cam, (float)((0.5 + x * 1.0) / w), (float)((0.5 + y * 1.0) / h), e, d);
        a = 0.0; b = RT_INFINITE;
        return true;
    }
    __device__ void reload(uint32_t i, V3 &e, V3 &d) const { double a, b; load(i, e, d, a, b); }
    __device__ void store_wave(bool on, uint32_t i, bool hit, const Hit &h) const { if (on) store(i, hit, h); }
    __device__ void store(uint32_t i, bool hit, const Hit &hh) const
    {
        int x, y;
        work_pixel(i, w, h, x, y);
        const uint64_t pix = (uint64_t)y * w + x;
        prim_id[pix] = hit ? global_prim(S, hh) : RT_MISS;
        if (tout) tout[pix] = hit ? hh.t : RT_INFINITE;
    }
};
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace_primary(PrimarySrc src, uint32_t *next, uint64_t *counters)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK>(src.S, src, next, 128u, stk, c);
    flush_counters(counters, nullptr, c);
}

// ------------------------------------------------------------------ lights
// firstIntersection2Sphere, algorithms.h:15-24
__device__ inline V3 first_intersection_sphere(V3 re, V3 rd, V3 center, double radius, double &length)
{
    V3 dir = center - re;
    double dist = len(dir);
    double cosv = dot(rd, norm(dir));
    double coslen = dist * cosv;
    double sinlen = dist * sqrt(1 - cosv * cosv);
    double dlen = sqrt(radius * radius - sinlen * sinlen);
    length = coslen - dlen;
    return re + rd * length;
}
// SphereLight::getPosition (light.cpp:138-190) / FaceLight::getPosition (light.cpp:76-86)
__device__ inline V3 light_sample(const rt_light &l, V3 ppos, double ru, double rv, V3 &normal, double &p)
{
    V3 pos_ = ld3(l.pos);
    if (l.type == RT_LIGHT_POINT) {                        // Light::getPosition, light.cpp:37-41: the normal stays (0,0,0)
        p = 1;
        normal = mk(0, 0, 0);
        return pos_;
    }
    if (l.type == RT_LIGHT_FACE) {
        V3 pos = pos_ + ((ld3(l.right) * (ru - 0.5)) * l.width + (ld3(l.up) * (rv - 0.5)) * l.height);
        p = 1.0 / l.width / l.height;
        normal = ld3(l.normal);
        return pos;
    }
    double radius_ = l.radius;
    V3 dir = pos_ - ppos;
    double dl = len(dir);
    if (fabs(dl - radius_) < RT_ZERO) { p = RT_INFINITE * 1.1; normal = mk(0, 0, 0); return mk(0, 0, 0); }
    V3 w = dir;
    V3 u = norm(cross(fabs(w.x) > 0.1 ? mk(0, 1, 0) : mk(1, 0, 0), w)) * dl;
    V3 v = norm(cross(w, u)) * dl;
    double cosmax = sqrt(1 - radius_ * radius_ / dot(dir, dir));
    if (cosmax >= 1 - RT_ZERO) { p = RT_INFINITE * 1.1; normal = mk(0, 0, 0); return mk(0, 0, 0); }
    double cos_a = 1 - ru + ru * cosmax;
    double sin_a = sqrt(1 - cos_a * cos_a);
    double phi = 2 * RT_PI * rv;
    V3 lightDir = ((u * cos(phi)) * sin_a + (v * sin(phi)) * sin_a) + w * cos_a;
    lightDir = norm(lightDir);
    double length;
    V3 hitp = first_intersection_sphere(ppos, lightDir, pos_, radius_, length);
    V3 nl = norm(hitp - pos_);
    normal = nl;
    p = dot(nl, -lightDir) / (2 * RT_PI * (1 - cosmax)) / length / length;
    return hitp;
}
// refract, scene.cpp:384-395 (true = total internal reflection)
__device__ inline bool refract_dir(V3 d, V3 normal, double n, V3 &t)
{
    double dn = dot(d, normal);
    double inSqrt = 1 - (1 - dn * dn) / n / n;
    if (inSqrt < 0) { t = reflact(d, normal); return true; }
    t = (d - normal * dn) * (1 / n) - normal * sqrt(inSqrt);
    t = norm(t);
    return false;
}

// ------------------------------------------------------------------ path tracing (wavefront)
// Scene::rayColor (scene.cpp:194-304) as a pipeline over pools of pending rays.  Recursion
// becomes data: shading a vertex appends one record per follow-up ray (a bounce ray carries
// throughput x lobe colour, a shadow ray carries the radiance it delivers if unoccluded), and
// every contribution is added to its pixel's sum with a hardware fp64 atomic.  Random numbers
// are keyed by a ray's position in the path tree, so the order in which rays are served does
// not change any path; only the order of the fp64 additions into a pixel varies.
__device__ __forceinline__ uint64_t pack_meta(uint32_t pix, uint32_t k, int32_t jump, int32_t E)
{
    return (uint64_t)pix | ((uint64_t)(k & 0xFFFFFFu) << 32) | ((uint64_t)(jump & 63) << 56) | ((uint64_t)(E & 1) << 62);
}
__device__ __forceinline__ void shadow_store(const ShadowStack &q, uint64_t slot, V3 e, V3 d, double t1, V3 contrib, uint32_t pix)
{
    uint64_t *b = q.f + slot;
    const uint64_t c = q.cap;
    b[SH_EX * c] = __double_as_longlong(e.x); b[SH_EY * c] = __double_as_longlong(e.y); b[SH_EZ * c] = __double_as_longlong(e.z);
    b[SH_DX * c] = __double_as_longlong(d.x); b[SH_DY * c] = __double_as_longlong(d.y); b[SH_DZ * c] = __double_as_longlong(d.z);
    b[SH_T1 * c] = __double_as_longlong(t1);
    b[SH_CX * c] = __double_as_longlong(contrib.x); b[SH_CY * c] = __double_as_longlong(contrib.y); b[SH_CZ * c] = __double_as_longlong(contrib.z);
    q.pix[slot] = pix;
}
__device__ __forceinline__ void ray_store(const RayStack &q, uint64_t slot, V3 e, V3 d, V3 w, uint64_t key, uint64_t meta)
{
    uint64_t *b = q.f + slot;
    const uint64_t c = q.cap;
    b[RAY_EX * c] = __double_as_longlong(e.x); b[RAY_EY * c] = __double_as_longlong(e.y); b[RAY_EZ * c] = __double_as_longlong(e.z);
    b[RAY_DX * c] = __double_as_longlong(d.x); b[RAY_DY * c] = __double_as_longlong(d.y); b[RAY_DZ * c] = __double_as_longlong(d.z);
    b[RAY_WX * c] = __double_as_longlong(w.x); b[RAY_WY * c] = __double_as_longlong(w.y); b[RAY_WZ * c] = __double_as_longlong(w.z);
    b[RAY_KEY * c] = key;
    b[RAY_META * c] = meta;
}
__device__ __forceinline__ double ray_f(const RayStack &q, uint64_t slot, int field) { return __longlong_as_double(q.f[(uint64_t)field * q.cap + slot]); }

// Class of a ray (see RayDest): does it reach the root box of an octree with inner nodes within
// its window?  A plain single-precision slab test: the answer only decides which rays are traced
// together, never what they return.
__device__ __forceinline__ int ray_class(const DevScene &S, V3 e, V3 d, double t1)
{
    const float ex = (float)e.x, ey = (float)e.y, ez = (float)e.z;
    const float rx = __builtin_amdgcn_rcpf((float)d.x), ry = __builtin_amdgcn_rcpf((float)d.y), rz = __builtin_amdgcn_rcpf((float)d.z);
    const float tmax = (float)t1;
    int cls = 0;
    for (int b = 0; b < S.nbig; b++) {
        const float ax = (S.big_lo[b][0] - ex) * rx, bx = (S.big_hi[b][0] - ex) * rx;
        const float ay = (S.big_lo[b][1] - ey) * ry, by = (S.big_hi[b][1] - ey) * ry;
        const float az = (S.big_lo[b][2] - ez) * rz, bz = (S.big_hi[b][2] - ez) * rz;
        const float r0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
        const float r1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
        if (!(r0 > r1 * 1.0001f + 1e-6f)) cls = 1;
    }
    return cls;
}

// Slot reservation for a block that appends rays: every wave passes its totals per destination
// (radiance class 0 / 1, shadow class 0 / 1), one atomic per destination and block reserves the
// slots, and each wave gets the first slot of its share.  All threads of the block must call this.
struct SlotCursor { uint64_t s[4]; };
__device__ __forceinline__ SlotCursor block_reserve(const RayDest &D, uint32_t wave_r0, uint32_t wave_r1, uint32_t wave_s0, uint32_t wave_s1)
{
    __shared__ uint32_t wave_tot[4][RT_BLOCK / 64], block_base[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __syncthreads();                     // an earlier reservation of this block has been read by every wave
    if (lane == 0) { wave_tot[0][wv] = wave_r0; wave_tot[1][wv] = wave_r1; wave_tot[2][wv] = wave_s0; wave_tot[3][wv] = wave_s1; }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t tot = 0;
        for (int i = 0; i < RT_BLOCK / 64; i++) tot += wave_tot[threadIdx.x][i];
        block_base[threadIdx.x] = tot ? atomicAdd(D.count + threadIdx.x, tot) : 0u;
    }
    __syncthreads();
    SlotCursor c;
    c.s[0] = D.base[0] + block_base[0];
    c.s[1] = D.base[1] + block_base[1];
    c.s[2] = D.sbase[0] + block_base[2];
    c.s[3] = D.sbase[1] + block_base[3];
    for (uint32_t i = 0; i < wv; i++) { c.s[0] += wave_tot[0][i]; c.s[1] += wave_tot[1][i]; c.s[2] += wave_tot[2][i]; c.s[3] += wave_tot[3][i]; }
    return c;
}
// The lanes with `want` append one ray each (class `cls`) at the cursor pair `pair` (0 = radiance
// stacks, 1 = shadow stacks) and move it on.  m1 = ballot(want && cls), m0 = ballot(want && !cls),
// computed by the caller before reserving.
__device__ __forceinline__ uint64_t take_slot(SlotCursor &c, int pair, unsigned long long m0, unsigned long long m1, int cls)
{
    const unsigned long long lt = (1ull << (threadIdx.x & 63u)) - 1ull;
    uint64_t *s = c.s + 2 * pair;
    const uint64_t slot = cls ? s[1] + (uint32_t)__popcll(m1 & lt) : s[0] + (uint32_t)__popcll(m0 & lt);
    s[0] += (uint32_t)__popcll(m0);
    s[1] += (uint32_t)__popcll(m1);
    return slot;
}

// First rays of work items [first, first+count): item = pixel-major (all samples of a pixel, then
// the next pixel of the tile order), so that the lanes of a wave start out on one pixel or two:
// their rays stay neighbours on the stacks, and what they add to the pixel is summed inside the
// wave first (add_radiance_wave).
__global__ __launch_bounds__(RT_BLOCK) void k_generate(DevScene S, WaveArgs A, RayDest D, uint64_t first, uint32_t count)
{
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool live = i < count;
    const uint64_t item = first + (live ? i : 0);
    V3 e, d;
    double t1;
    uint32_t pix, k;
    if (A.batch) {
        pix = (uint32_t)item; k = 0;
        e = ld3(A.rays[item].e); d = ld3(A.rays[item].d);
        t1 = A.t1;
    } else {
        const uint32_t ns = (uint32_t)(A.sample_end - A.sample_begin);
        k = (uint32_t)A.sample_begin + (uint32_t)(item % ns);
        int x, y;
        work_pixel((uint32_t)(item / ns), A.width, A.height, x, y);
        pix = (uint32_t)y * (uint32_t)A.width + (uint32_t)x;
        double jx, jy;                                          // Scene::run, scene.cpp:105-108
        jitter_pair(A.seed, pix, 0, k, (uint32_t)A.n, (uint32_t)A.m, jx, jy);
        const float fs = (float)((jx + x * 1.0) / A.width), ft = (float)((jy + y * 1.0) / A.height);
        if (A.cam.aperture != 0) camera_ray_lens(A.cam, fs, ft, path_root_key(A.seed, pix, k), e, d);
        else camera_ray(A.cam, fs, ft, e, d);
        t1 = A.t1;
    }
    const int cls = live ? ray_class(S, e, d, t1) : 0;
    const unsigned long long m0 = __ballot(live && !cls), m1 = __ballot(live && cls);
    SlotCursor cur = block_reserve(D, (uint32_t)__popcll(m0), (uint32_t)__popcll(m1), 0u, 0u);
    const uint64_t slot = take_slot(cur, 0, m0, m1, cls);
    if (live) ray_store(D.q[cls], slot, e, d, mk(1, 1, 1), path_root_key(A.seed, pix, k), pack_meta(pix, k, A.jumptime, 1));
}

__device__ __forceinline__ void add_radiance(double *out, uint32_t pix, V3 v)
{
    if (v.x != 0.0) unsafeAtomicAdd(out + 3ull * pix, v.x);
    if (v.y != 0.0) unsafeAtomicAdd(out + 3ull * pix + 1, v.y);
    if (v.z != 0.0) unsafeAtomicAdd(out + 3ull * pix + 2, v.z);
}

// add_radiance for a whole wave (every lane calls it; `on` = this lane has something to add):
// neighbouring lanes that add to the same pixel are summed first (a segmented scan over the runs
// of equal pixels) and the last lane of a run adds the run's sum, so a pixel's 64 samples in one
// wave cost one addition per channel instead of 64 to the same address.  With few such
// neighbours the lanes add for themselves.
__device__ __forceinline__ void add_radiance_wave(double *out, bool on, uint32_t pix, V3 v)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long act = ballot64(on);
    if (act == 0ull) return;
    const uint32_t key = on ? pix : 0xFFFFFFFFu;                  // lanes with nothing to add end a run
    const uint32_t before = (uint32_t)__shfl_up((int)key, 1, 64);
    const unsigned long long heads = ballot64(lane == 0u || before != key);
    if (2u * (uint32_t)__popcll(heads & act) > (uint32_t)__popcll(act)) {
        if (on) add_radiance(out, pix, v);
        return;
    }
    if (!on) v = mk(0, 0, 0);
    const uint32_t start = 63u - (uint32_t)__clzll((long long)(heads & (~0ull >> (63u - lane))));   // first lane of this lane's run
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const V3 u = mk(__shfl_up(v.x, off, 64), __shfl_up(v.y, off, 64), __shfl_up(v.z, off, 64));
        if (lane >= start + off) v = v + u;
    }
    const unsigned long long tails = (heads >> 1) | (1ull << 63);
    if (on && ((tails >> lane) & 1ull)) add_radiance(out, pix, v);
}

// The shadow test for the chunk [base, base+count) of a shadow stack: group_->calHit(newray, ZERO,
// dist - ZERO, nullptr) (scene.cpp:236), the same closest-hit search as for any ray (the reference
// does not stop at the first blocker, and the work counters count that search); a ray that finds
// nothing adds its radiance to its pixel (scene.cpp:238) right here.
struct ShadowSrc {
    ShadowStack q; uint64_t base; uint32_t n; double *out;
    __device__ double f(uint64_t slot, int field) const { return __longlong_as_double(q.f[(uint64_t)field * q.cap + slot]); }
    __device__ uint32_t count() const { return n; }
    __device__ bool load(uint32_t i, V3 &e, V3 &d, double &t0, double &t1) const
    {
        const uint64_t slot = base + i;
        e = mk(f(slot, SH_EX), f(slot, SH_EY), f(slot, SH_EZ));
        d = mk(f(slot, SH_DX), f(slot, SH_DY), f(slot, SH_DZ));
        t0 = RT_ZERO;
        t1 = f(slot, SH_T1);
        return true;
    }
    __device__ void reload(uint32_t i, V3 &e, V3 &d) const
    {
        const uint64_t slot = base + i;
        e = mk(f(slot, SH_EX), f(slot, SH_EY), f(slot, SH_EZ));
        d = mk(f(slot, SH_DX), f(slot, SH_DY), f(slot, SH_DZ));
    }
    __device__ void store(uint32_t i, bool hit, const Hit &) const
    {
        if (hit) return;
        const uint64_t slot = base + i;
        add_radiance(out, q.pix[slot], mk(f(slot, SH_CX), f(slot, SH_CY), f(slot, SH_CZ)));
    }
    // store() for the lanes with `on`, called by the whole wave
    __device__ void store_wave(bool on, uint32_t i, bool hit, const Hit &) const
    {
        const bool add = on && !hit;
        const uint64_t slot = base + (add ? i : 0u);
        uint32_t pix = 0;
        V3 v = mk(0, 0, 0);
        if (add) { pix = q.pix[slot]; v = mk(f(slot, SH_CX), f(slot, SH_CY), f(slot, SH_CZ)); }
        add_radiance_wave(out, add, pix, v);
    }
};
template <bool CHAIN>
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace_shadow(DevScene S, ShadowSrc src, uint32_t *next, uint32_t batch_max, uint64_t *counters, uint64_t *kstats)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK, CHAIN>(S, src, next, batch_max, stk, c);
    flush_counters(counters, kstats, c);
}

// Closest hit for the chunk [base, base+count) of a ray stack.
struct StackSrc {
    WaveArgs A; RayStack q; uint64_t base; uint32_t n;
    __device__ uint32_t count() const { return n; }
    __device__ bool load(uint32_t i, V3 &e, V3 &d, double &t0, double &t1) const
    {
        const uint64_t slot = base + i;
        const uint64_t meta = q.f[(uint64_t)RAY_META * q.cap + slot];
        // origin and direction are asked for together with the depth word (one wait, not two)
        e = mk(ray_f(q, slot, RAY_EX), ray_f(q, slot, RAY_EY), ray_f(q, slot, RAY_EZ));
        d = mk(ray_f(q, slot, RAY_DX), ray_f(q, slot, RAY_DY), ray_f(q, slot, RAY_DZ));
        const int jump = (int)((meta >> 56) & 63);
        // depth cut-off (scene.cpp:196-198) happens before any intersection work
        if (jump + 1 > A.max_jump) return false;
        const bool first = jump == A.jumptime && ((meta >> 62) & 1);
        t0 = first ? A.t0 : RT_ZERO;
        t1 = first ? A.t1 : RT_INFINITE;
        return true;
    }
    __device__ void reload(uint32_t i, V3 &e, V3 &d) const
    {
        const uint64_t slot = base + i;
        e = mk(ray_f(q, slot, RAY_EX), ray_f(q, slot, RAY_EY), ray_f(q, slot, RAY_EZ));
        d = mk(ray_f(q, slot, RAY_DX), ray_f(q, slot, RAY_DY), ray_f(q, slot, RAY_DZ));
    }
    __device__ void store_wave(bool on, uint32_t i, bool hit, const Hit &h) const { if (on) store(i, hit, h); }
    __device__ void store(uint32_t i, bool hit, const Hit &h) const
    {
        const uint64_t slot = base + i;
        q.hit_t[slot] = hit ? h.t : RT_INFINITE;
        q.hit_id[slot] = hit ? (((uint64_t)(uint32_t)h.obj << 32) | h.prim) : ~0ull;
    }
};
template <bool CHAIN>
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace(DevScene S, StackSrc src, uint32_t *next, uint32_t batch_max)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK, CHAIN>(S, src, next, batch_max, stk, c);
    flush_counters(src.A.counters, src.A.kstats ? src.A.kstats + RT_KSTAT_WORDS * RT_KSTAT_ENGINE : nullptr, c);
}

// ------------------------------------------------------------------ rays that stay outside the octrees
// Rays of class 0 (see RayDest) are expected to miss the root box of every octree with inner
// nodes, so their whole search is one walk over the object table: Group::hit (Surface.cpp:160-181)
// with spheres, planes and one-leaf meshes answered on the spot.  Such rays need none of the
// closest-hit engine's machinery -- no node stack, no phases, no hand-out of rays to free lanes --
// and a plain kernel serves them: one ray per lane from start to end, every lane at the same
// object at the same time, registers and LDS for far more waves per SIMD.
// The class is a single-precision guess, though.  A ray that does reach an octree's root box here
// (the same double-precision slab test that the engine's first node step makes) is not answered:
// its index goes to the launch's retry list, nothing is counted or stored for it, and the engine
// traces the rays of that list from scratch right after this kernel (RetrySrc), so every ray gets
// the result and the work counters that the engine alone would have produced.
template <class Src>
__device__ __forceinline__ void sweep_rays(const DevScene &S, const Src &src, uint32_t *retry_count, uint32_t *retry_list, Counters &c)
{
    const uint32_t total = src.count();
    const uint32_t lane = threadIdx.x & 63u;
    const u_object objects = (u_object)(unsigned long long)S.objects;
    for (uint32_t first = blockIdx.x * RT_BLOCK; first < total; first += gridDim.x * RT_BLOCK) {
        const uint32_t item = first + threadIdx.x;
        V3 e = mk(0, 0, 0), d = mk(0, 0, 1);
        double t0 = 0, t1 = 0;
        bool live = false;
        if (item < total) {
            live = src.load(item, e, d, t0, t1);
            if (!live) { Hit h; h.t = RT_INFINITE; h.obj = -1; h.prim = 0; src.store(item, false, h); }
        }
        const bool fast = S.regular && dir_is_regular(d) && org_is_regular(e);
        const V3 r = fast ? mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z) : mk(0, 0, 0);
        int32_t hobj = -1;
        uint32_t hprim = 0, nleaf = 0, nprim = 0;
        bool defer = false;
        for (int o = 0; o < S.nobjects; o++) {
            const bool mine = live && !defer;
            if (ballot64(mine) == 0ull) break;
            const int otype = objects[o].type;
            if (otype == RT_OBJ_MESH) {
                const uint32_t ntri = objects[o].ntri;
                const V3 lo = mk(objects[o].lo[0], objects[o].lo[1], objects[o].lo[2]), hi = mk(objects[o].hi[0], objects[o].hi[1], objects[o].hi[2]);
                double n0 = 0, n1 = 0;
                const bool hb = mine && (fast ? hit_box_fast(lo, hi, e, d, r, t0, t1, n0, n1) : hit_box(lo, hi, e, d, t0, t1, n0, n1));
                if (!ntri) { if (hb) defer = true; continue; }          // an octree with inner nodes: Octree::hit's root test (octree.cpp:291)
                // a one-leaf octree: the root's slab test and OctreeNode::hit on it (octree.cpp:285-376, 19-38)
                double tempt = n1;
                uint32_t bprim = RT_MISS;
                if (hb) { nleaf++; nprim += ntri; }
                if (ballot64(hb) != 0ull) {
                    const u_leaftri P = (u_leaftri)(unsigned long long)(S.leaf_tris + objects[o].leaf_begin);
                    TriReg tr_next = tri_load_uniform(P);
#pragma unroll 1
                    for (uint32_t k = 0; k < ntri; k++) {
                        const TriReg tr = tr_next;
                        tr_next = tri_load_uniform(P + (k + 1 < ntri ? k + 1 : k));
                        double tt;
                        if (hb && tri_hit(tr, e, d, n0, tempt, tt)) { tempt = tt; bprim = tr.id; }      // last equal wins, octree.cpp:31
                    }
                    if (hb && bprim != RT_MISS && tempt < t1 + 1 && tempt <= t1) { t1 = tempt; hobj = o; hprim = bprim; }    // octree.cpp:313, 320-325, 371-375
                }
            } else if (otype == RT_OBJ_HEART || otype == RT_OBJ_GENUS2) {
                // HeartShape::hit (Surface.cpp:268-276), ImplicitSurface::hit (Surface.h:308-316): the
                // voxel tree's root test with the ray in the surface's frame
                if (mine) {
                    const double inv = 1.0 / objects[o].p[3];
                    const V3 le = (e - mk(objects[o].p[0], objects[o].p[1], objects[o].p[2])) * inv, ld = d * inv;
                    const FlatNode *rn = S.nodes + objects[o].root;
                    const V3 lo = mk(rn->lo[0], rn->lo[1], rn->lo[2]), hi = mk(rn->hi[0], rn->hi[1], rn->hi[2]);
                    const bool lfast = S.regular && dir_is_regular(ld) && org_is_regular(le);
                    double n0, n1;
                    if (lfast ? hit_box_fast(lo, hi, le, ld, mk(1.0 / ld.x, 1.0 / ld.y, 1.0 / ld.z), t0, t1, n0, n1) : hit_box(lo, hi, le, ld, t0, t1, n0, n1)) defer = true;
                }
            } else if (mine) {
                const double p0 = objects[o].p[0], p1 = objects[o].p[1], p2 = objects[o].p[2], p3 = objects[o].p[3];
                double t;
                const bool hit = (otype == RT_OBJ_SPHERE) ? sphere_hit(mk(p0, p1, p2), p3, e, d, t0, t1, t) : plane_hit(otype, p0, e, d, t0, t1, t);
                if (hit) { t1 = t; hobj = o; hprim = 0; }
            }
        }
        {
            const bool done = live && !defer;
            Hit h; h.t = t1; h.obj = hobj; h.prim = hprim;
            src.store_wave(done, item, hobj >= 0, h);
            if (done) { c.rays++; c.leaf += nleaf; c.prim += nprim; c.fp64 += nprim; }
        }
        const unsigned long long dm = ballot64(defer);
        if (dm != 0ull) {
            uint32_t at = 0;
            if (lane == (uint32_t)__ffsll((long long)dm) - 1u) at = atomicAdd(retry_count, (uint32_t)__popcll(dm));
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, __ffsll((long long)dm) - 1);
            if (defer) retry_list[at + count_below(dm, 0u)] = item;
        }
    }
}
// The rays of a retry list through the closest-hit engine: item i of the launch is ray list[i] of `b`.
template <class Base>
struct RetrySrc {
    Base b; const uint32_t *list; const uint32_t *n;
    __device__ uint32_t count() const { return *n; }
    __device__ bool load(uint32_t i, V3 &e, V3 &d, double &t0, double &t1) const { return b.load(list[i], e, d, t0, t1); }
    __device__ void reload(uint32_t i, V3 &e, V3 &d) const { b.reload(list[i], e, d); }
    __device__ void store(uint32_t i, bool hit, const Hit &h) const { b.store(list[i], hit, h); }
    __device__ void store_wave(bool on, uint32_t i, bool hit, const Hit &h) const { b.store_wave(on, on ? list[i] : 0u, hit, h); }
};
__global__ __launch_bounds__(RT_BLOCK) void k_sweep(DevScene S, StackSrc src, uint32_t *retry_count, uint32_t *retry_list)
{
    Counters c = { 0, 0, 0, 0, 0, 0 };
    sweep_rays(S, src, retry_count, retry_list, c);
    flush_counters(src.A.counters, src.A.kstats ? src.A.kstats + RT_KSTAT_WORDS * RT_KSTAT_SWEEP : nullptr, c);
}
__global__ __launch_bounds__(RT_BLOCK) void k_sweep_shadow(DevScene S, ShadowSrc src, uint32_t *retry_count, uint32_t *retry_list, uint64_t *counters, uint64_t *kstats)
{
    Counters c = { 0, 0, 0, 0, 0, 0 };
    sweep_rays(S, src, retry_count, retry_list, c);
    flush_counters(counters, kstats, c);
}
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace_retry(DevScene S, RetrySrc<StackSrc> src, uint32_t *next, uint32_t batch_max)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK>(S, src, next, batch_max, stk, c);
    flush_counters(src.b.A.counters, src.b.A.kstats ? src.b.A.kstats + RT_KSTAT_WORDS * RT_KSTAT_SWEEP : nullptr, c);
}
RT_TRACE_ATTR
__global__ __launch_bounds__(RT_BLOCK) void k_trace_shadow_retry(DevScene S, RetrySrc<ShadowSrc> src, uint32_t *next, uint32_t batch_max, uint64_t *counters, uint64_t *kstats)
{
    __shared__ uint2 stk[RT_STACK_LEVELS][RT_BLOCK];
    Counters c = { 0, 0, 0, 0, 0, 0 };
    trace_engine<RT_BLOCK>(S, src, next, batch_max, stk, c);
    flush_counters(counters, kstats, c);
}

// Consume the hits of chunk [base, base+count) of `src`; follow-up rays are appended to D.
// KINDS says which material features the scene's objects use (DevScene::shade_kinds): 1 = a mirror
// lobe, 2 = a dielectric lobe, 4 = the Phong / Whitted branch (materials without MONTECARLO, Perlin
// and check variants).  What a scene does not use is compiled out, which is all that differs: the
// kernel for plain Monte-Carlo diffuse scenes needs 163 registers a lane instead of 256 and runs
// three waves per SIMD instead of two.
template <int KINDS>
#ifdef RT_SHADE_WAVES
__attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, RT_SHADE_WAVES)))
#endif
__global__ __launch_bounds__(RT_BLOCK) void k_shade(DevScene S, WaveArgs A, RayStack src, uint64_t base, uint32_t count, RayDest D)
{
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool live = i < count;
    const uint64_t slot = base + (live ? i : 0);
    const uint64_t meta = src.f[(uint64_t)RAY_META * src.cap + slot];
    const uint32_t pix = (uint32_t)meta, k = (uint32_t)(meta >> 32) & 0xFFFFFFu;
    const int jump = (int)((meta >> 56) & 63), E = (int)((meta >> 62) & 1);
    const uint64_t id = src.hit_id[slot];
    const bool hit = id != ~0ull;
    const V3 W = mk(ray_f(src, slot, RAY_WX), ray_f(src, slot, RAY_WY), ray_f(src, slot, RAY_WZ));
    const uint32_t tab_n = A.batch ? 1u : (uint32_t)A.n, tab_m = A.batch ? 1u : (uint32_t)A.m;

    V3 sum = mk(0, 0, 0);                 // everything this ray adds to its pixel
    bool shade = false;
    if (live) {
        if (!hit) sum = had(W, ld3(S.background));                           // scene.cpp:196-198, 378-380
        else shade = true;
    }
    // values every push site needs, asked for together with the fields above (a miss reads them for
    // nothing, but a hit does not wait a second time)
    V3 re = mk(ray_f(src, slot, RAY_EX), ray_f(src, slot, RAY_EY), ray_f(src, slot, RAY_EZ));
    V3 rd = mk(ray_f(src, slot, RAY_DX), ray_f(src, slot, RAY_DY), ray_f(src, slot, RAY_DZ));
    uint64_t key = src.f[(uint64_t)RAY_KEY * src.cap + slot];
    double ht = src.hit_t[slot];
    if (!shade) { re = mk(0, 0, 0); rd = mk(0, 0, 1); key = 0; ht = 0; }
    const int jt = jump + 1;
    Hit h; h.t = ht; h.obj = (int32_t)(id >> 32); h.prim = (uint32_t)id;
    if (!shade) { h.obj = 0; h.prim = 0; }
    const rt_material &mat = S.materials[shade ? S.objects[h.obj].material : 0];
    const int mtype = shade ? mat.type : 0;
    const V3 p = re + rd * ht;
    V3 normal = mk(0, 1, 0);
    if (shade) normal = hit_normal(S, h, p);
    const V3 emission = ld3(mat.emission), mcolor = ld3(mat.color);
    const double Ed = (double)E;
    // mc: the Monte-Carlo branch (scene.cpp:205-304); otherwise the Phong / Whitted branch
    // (scene.cpp:305-375): no roulette, no emission, ambient + shadowed diffuse / specular terms
    // per light instead of the diffuse bounce, the mirror colour instead of getColor()
    const bool mc = (KINDS & 4) ? (mtype & RT_MAT_MONTECARLO) != 0 : true;
    V3 cc = mcolor;
    bool alive = shade;
    if (shade && mc && jt > A.rus_jump) {                                    // Russian roulette, scene.cpp:208-212
        double pr = (cc.x > cc.y && cc.x > cc.z) ? cc.x : (cc.y > cc.z ? cc.y : cc.z);
        if (draw(key, SLOT_RR) < pr) cc = cc * (1 / pr);
        else { sum = sum + had(W, emission * Ed); alive = false; }
    }

    // ---- diffuse lobe, scene.cpp:213-248
    const bool lobeN = alive && mc && (mtype & RT_MAT_NORMAL);
    const bool phongN = (KINDS & 4) && alive && !mc && (mtype & RT_MAT_NORMAL);
    V3 w = normal, dn = mk(0, 0, 0), wn = normal;
    if (lobeN) {
        double r1 = RT_PI * 2 * draw(key, SLOT_DIFF1);
        double r2 = draw(key, SLOT_DIFF2), r2s = sqrt(r2);
        if (A.face_forward && dot(w, rd) > 0) w = -w;
        V3 u = norm(cross(fabs(w.x) > 0.1 ? mk(0, 1, 0) : mk(1, 0, 0), w));
        V3 v = cross(w, u);
        double sn, cs;
        sincos(r1, &sn, &cs);                                               // one argument reduction for both
        dn = norm(((u * cs) * r2s + (v * sn) * r2s) + w * sqrt(1 - r2));
        wn = norm(w);
        sum = sum + had(W, emission * Ed);
    }
    // next-event estimation, scene.cpp:223-246; the shadow ray of light 0 is appended together
    // with the lobe rays below, further lights append theirs one by one
    // the colours the Phong terms use: the material's own, or what a Perlin material makes of them here
    V3 pamb = ld3(mat.ambient), pdif = mcolor, pspe = ld3(mat.specular);
    if (phongN && mat.perlin != RT_PERLIN_NONE) perlin_colours(S, mat, p, pamb, pdif, pspe);
    bool want0 = false;
    V3 ldir0 = mk(0, 0, 0), contrib0 = mk(0, 0, 0);
    double dist0 = 0;
    for (int li = 0; li < S.nlights; li++) {
        bool want = false;
        V3 ldir = mk(0, 0, 0), contrib = mk(0, 0, 0);
        double dist = 0;
        if (lobeN) {
            const rt_light &L = S.lights[li];
            double lu, lv;
            jitter_pair(A.seed, pix, 1u + (uint32_t)li, k, tab_n, tab_m, lu, lv);
            V3 nl; double pr;
            V3 ppos = rd * ht + re;                                          // light.cpp:143
            V3 lightpos = light_sample(L, ppos, lu, lv, nl, pr);
            if (!(pr > RT_INFINITE)) {
                dist = len(lightpos - p);
                ldir = norm(lightpos - p);
                double dt = dot(nl, -ldir);
                double dt2 = dot(wn, ldir);
                if (!(dt < 0 || dt2 < 0)) {
                    V3 ld = ((ld3(L.diffuse) * dt) * dt2) * (1 / pr / dist / dist);
                    contrib = had(W, had(cc, ld));
                    want = true;
                }
            }
        } else if (phongN) {                                                 // scene.cpp:307-322
            const rt_light &L = S.lights[li];
            V3 amb = pamb;
            if (mat.check == RT_CHECK_NORMAL) amb = mk(fabs(normal.x), fabs(normal.y), fabs(normal.z));              // material.h:78
            else if (mat.check == RT_CHECK_POSITION) {                                                                // material.h:88-89
                // HitRecord::localp: the hit in a HeartShape's own frame (Surface.cpp:268-275), zero for any other drawable
                const DevObject &ho = S.objects[h.obj];
                V3 lp = mk(0, 0, 0);
                if (ho.type == RT_OBJ_HEART) {
                    const double inv = 1.0 / ho.p[3];
                    lp = (re - mk(ho.p[0], ho.p[1], ho.p[2])) * inv + (rd * inv) * ht;
                }
                const V3 blo = ld3(mat.check_box), bhi = ld3(mat.check_box + 3);
                amb = mk((lp.x - blo.x) / (bhi.x - blo.x), (lp.y - blo.y) / (bhi.y - blo.y), (lp.z - blo.z) / (bhi.z - blo.z));
            }
            sum = sum + had(W, had(ld3(L.ambient), amb));
            // light->getPosition() without a hit record: the point itself, a jittered point of a
            // FaceLight; a SphereLight (which would dereference the null record, light.cpp:143) is
            // taken at its centre
            V3 lightpos = ld3(L.pos);
            if (L.type == RT_LIGHT_FACE) {
                double lu, lv;
                jitter_pair(A.seed, pix, 1u + (uint32_t)li, k, tab_n, tab_m, lu, lv);
                V3 nl; double pr;
                lightpos = light_sample(L, p, lu, lv, nl, pr);
            }
            ldir = norm(lightpos - p);
            dist = len(lightpos - p);
            double kd = dot(normal, ldir);
            if (!(0.0 < kd)) kd = 0.0;                                       // max(0.0, n.d)
            const double ks = pow(dot(norm(ldir - rd), normal), mat.shine);
            contrib = had(W, had(pdif, ld3(L.diffuse) * kd) + had(pspe, ld3(L.specular) * ks));
            want = true;                                                     // the shadow test is made whatever it will deliver
        }
        if (li == 0) { want0 = want; ldir0 = ldir; contrib0 = contrib; dist0 = dist; continue; }
        const int cls = want ? ray_class(S, p, ldir, dist - RT_ZERO) : 0;
        const unsigned long long m0 = __ballot(want && !cls), m1 = __ballot(want && cls);
        SlotCursor cur = block_reserve(D, 0u, 0u, (uint32_t)__popcll(m0), (uint32_t)__popcll(m1));
        const uint64_t slot = take_slot(cur, 1, m0, m1, cls);
        if (want) shadow_store(D.sq[cls], slot, p, ldir, dist - RT_ZERO, contrib, pix);
    }

    // ---- mirror lobe, scene.cpp:249-266
    const bool lobeS = (KINDS & 1) && alive && (mtype & RT_MAT_SPECULAR);
    V3 md = mk(0, 0, 0);
    if (lobeS) {
        md = rd - (normal * dot(normal, rd)) * 2;
        if (mat.blur >= RT_ZERO) {
            V3 vec = mk(0, 0, 1);
            if (iszero(cross(vec, md))) vec = mk(0, 1, 0);
            V3 u = norm(cross(vec, md));
            V3 v = norm(cross(md, u));
            double x, y;
            uint32_t a = 0;
            do {                                                             // randFromDisk, algorithms.h:30-35
                x = (draw(key, SLOT_DISK + 2 * a) - 0.5) * 2;
                y = (draw(key, SLOT_DISK + 2 * a + 1) - 0.5) * 2;
                a++;
            } while (x * x + y * y >= 1.0);
            x *= mat.blur; y *= mat.blur;
            md = (md + u * x) + v * y;
        }
        if (mc) sum = sum + had(W, emission * Ed);
    }
    const V3 mweight = mc ? mcolor : ld3(mat.mirror);                        // scene.cpp:265 / :341

    // ---- dielectric lobe, scene.cpp:267-303 (and :343-373)
    const bool lobeT = (KINDS & 2) && alive && (mtype & RT_MAT_TRANSPARENT);
    V3 kk = mk(1, 1, 1), r = mk(0, 0, 0), rfr = mk(0, 0, 0);
    double R = 0;
    bool tir = false;
    if (lobeT) {
        double cosv = 0;
        r = reflact(rd, normal);
        if (dot(normal, rd) > 0) {
            kk.x = exp(-mat.atten[0] * ht);
            kk.y = exp(-mat.atten[1] * ht);
            kk.z = exp(-mat.atten[2] * ht);
            if (refract_dir(rd, -normal, 1 / mat.nr, rfr)) { tir = true; if (mc) sum = sum + had(W, emission * Ed); }
            else cosv = dot(rfr, normal);
        } else {
            cosv = dot(-rd, normal);
            refract_dir(rd, normal, mat.nr, rfr);
        }
        if (!tir) {
            double n = mat.nr;
            double R0 = (n - 1) * (n - 1) / (n + 1) / (n + 1);
            R = R0 + (1 - R0) * pow(1 - cosv, 5.0);
        }
    }
    const bool wantT = lobeT && !tir;

    // ---- append the follow-up rays: one slot reservation per block and class for the five kinds
    // (diffuse bounce, shadow ray of light 0, mirror, reflected, refracted); a block's rays of one
    // kind and class end up next to each other
    {
        const int k0 = lobeN ? ray_class(S, p, dn, RT_INFINITE) : 0;
        const int k1 = want0 ? ray_class(S, p, ldir0, dist0 - RT_ZERO) : 0;
        const int k2 = lobeS ? ray_class(S, p, md, RT_INFINITE) : 0;
        const int k3 = lobeT ? ray_class(S, p, r, RT_INFINITE) : 0;
        const int k4 = wantT ? ray_class(S, p, rfr, RT_INFINITE) : 0;
        const unsigned long long a0 = __ballot(lobeN && !k0), b0 = __ballot(lobeN && k0);
        const unsigned long long a1 = __ballot(want0 && !k1), b1 = __ballot(want0 && k1);
        const unsigned long long a2 = __ballot(lobeS && !k2), b2 = __ballot(lobeS && k2);
        const unsigned long long a3 = __ballot(lobeT && !k3), b3 = __ballot(lobeT && k3);
        const unsigned long long a4 = __ballot(wantT && !k4), b4 = __ballot(wantT && k4);
        SlotCursor cur = block_reserve(D, (uint32_t)(__popcll(a0) + __popcll(a2) + __popcll(a3) + __popcll(a4)),
                                       (uint32_t)(__popcll(b0) + __popcll(b2) + __popcll(b3) + __popcll(b4)),
                                       (uint32_t)__popcll(a1), (uint32_t)__popcll(b1));
        uint64_t slot;
        slot = take_slot(cur, 0, a0, b0, k0);
        if (lobeN) ray_store(D.q[k0], slot, p, dn, had(W, mcolor), hash2(key, BR_DIFFUSE), pack_meta(pix, k, jt + 1, 0));
        slot = take_slot(cur, 1, a1, b1, k1);
        if (want0) shadow_store(D.sq[k1], slot, p, ldir0, dist0 - RT_ZERO, contrib0, pix);
        slot = take_slot(cur, 0, a2, b2, k2);
        if (lobeS) ray_store(D.q[k2], slot, p, md, had(W, mweight), hash2(key, BR_SPECULAR), pack_meta(pix, k, jt, 1));
        // reflected ray: weight k (total internal reflection) or k*R
        slot = take_slot(cur, 0, a3, b3, k3);
        if (lobeT) ray_store(D.q[k3], slot, p, r, had(W, tir ? kk : kk * R), hash2(key, BR_REFLECT), pack_meta(pix, k, jt, 1));
        slot = take_slot(cur, 0, a4, b4, k4);
        if (wantT) ray_store(D.q[k4], slot, p, rfr, had(W, kk * (1 - R)), hash2(key, BR_REFRACT), pack_meta(pix, k, jt, 1));
    }

    if (live) add_radiance(A.out, pix, sum);
}

// colorArray_ -> bytes: average, regularize (types.h:167-177), int(c*255) as B,G,R (scene.cpp:183-185)
// with tone != 0 a tone reproduction operator acts on the average first: SigmoidMethod (1) /
// SqrtMethod (2)::reproduction, tonereproductionmethod.h:11-35
__global__ void k_resolve_bgr8(const double *sum, uint64_t npix, double inv, int tone, uint8_t *bgr)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = sum[3 * i + k] * inv;
        if (tone == RT_TONE_SIGMOID) v = 1.0 / (1 + exp(-(2 * v - 1))) * 2.164 - 0.582;
        else if (tone == RT_TONE_SQRT) v = v > 1 ? pow(v, 1.0 / 2) : v;
        if (v > 1.0) v = 1.0; else if (v < 0) v = 0;
        c[k] = v;
    }
    bgr[3 * i + 2] = (uint8_t)(int)(c[0] * 255);
    bgr[3 * i + 1] = (uint8_t)(int)(c[1] * 255);
    bgr[3 * i + 0] = (uint8_t)(int)(c[2] * 255);
}

// ------------------------------------------------------------------ launchers
// Persistent grids: no more blocks than stay resident, no more than the rays need.
static unsigned engine_blocks(uint64_t items, int num_cu, const void *kernel)
{
    static thread_local const void *cached_k[8] = {};
    static thread_local int cached_v[8] = {};
    int per_cu = 0;
    for (int i = 0; i < 8; i++) if (cached_k[i] == kernel) per_cu = cached_v[i];
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        for (int i = 0; i < 8; i++) if (!cached_k[i]) { cached_k[i] = kernel; cached_v[i] = per_cu; break; }
    }
    uint64_t resident = (uint64_t)(num_cu > 0 ? num_cu : 256) * (uint64_t)per_cu;
    uint64_t need = (items + RT_BLOCK - 1) / RT_BLOCK;
    return (unsigned)(need < resident ? need : resident);
}
hipError_t launch_trace_rays(const DevScene &S, int num_cu, uint32_t *next, const rt_ray *rays, uint64_t n, double t0, double t1,
                             uint32_t *prim_id, double *t, double *normal, uint64_t *counters, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    RaySrc src = { S, rays, (uint32_t)n, t0, t1, prim_id, t, normal };
    hipError_t e = hipMemsetAsync(next, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_trace_rays, dim3(engine_blocks(n, num_cu, (const void *)k_trace_rays)), dim3(RT_BLOCK), 0, st, src, next, counters);
    if (normal) hipLaunchKernelGGL(k_ray_normals, dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, st, S, rays, (uint32_t)n, normal);
    return hipGetLastError();
}
hipError_t launch_trace_primary(const DevScene &S, int num_cu, uint32_t *next, const rt_camera &cam, int w, int h, uint32_t *prim_id,
                                double *t, uint64_t *counters, hipStream_t st)
{
    PrimarySrc src = { S, cam, w, h, prim_id, t };
    hipError_t e = hipMemsetAsync(next, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_trace_primary, dim3(engine_blocks((uint64_t)w * h, num_cu, (const void *)k_trace_primary)), dim3(RT_BLOCK), 0, st, src, next, counters);
    return hipGetLastError();
}
hipError_t launch_generate(const DevScene &S, const WaveArgs &A, const RayDest &dst, uint64_t first, uint32_t count, hipStream_t st)
{
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_generate, dim3((count + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, S, A, dst, first, count);
    return hipGetLastError();
}
// grid of a sweep kernel: enough blocks to fill the device a few times over, each striding over the rays
static unsigned sweep_blocks(uint64_t items, int num_cu)
{
    const uint64_t need = (items + RT_BLOCK - 1) / RT_BLOCK, most = (uint64_t)(num_cu > 0 ? num_cu : 256) * 16u;
    return (unsigned)(need < most ? need : most);
}
// `ctrl` = the scene's control words: [4] the engine's ray hand-out counter, [5] the length of the
// retry list (both zeroed here); `retry` = room for `count` ray indices.
hipError_t launch_trace(const DevScene &S, int num_cu, uint32_t *ctrl, uint32_t *retry, const WaveArgs &A, const RayStack &q, uint64_t base, uint32_t count,
                        int ray_class, hipStream_t st)
{
    if (!count) return hipSuccess;
    StackSrc src = { A, q, base, count };
    hipError_t e = hipMemsetAsync(ctrl + 4, 0, 8, st);
    if (e != hipSuccess) return e;
    if (ray_class == 0) {
        // the object walk for all, then the engine for the few that reached an octree after all
        RetrySrc<StackSrc> again = { src, retry, ctrl + 5 };
        hipLaunchKernelGGL(k_sweep, dim3(sweep_blocks(count, num_cu)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 5, retry);
        hipLaunchKernelGGL(k_trace_retry, dim3(engine_blocks(count, num_cu, (const void *)k_trace_retry)), dim3(RT_BLOCK), 0, st, S, again, ctrl + 4, (uint32_t)RT_BATCH1);
        return hipGetLastError();
    }
    if (S.any_chain) hipLaunchKernelGGL(k_trace<true>, dim3(engine_blocks(count, num_cu, (const void *)k_trace<true>)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 4, (uint32_t)RT_BATCH1);
    else hipLaunchKernelGGL(k_trace<false>, dim3(engine_blocks(count, num_cu, (const void *)k_trace<false>)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 4, (uint32_t)RT_BATCH1);
    return hipGetLastError();
}
hipError_t launch_trace_shadow(const DevScene &S, int num_cu, uint32_t *ctrl, uint32_t *retry, const WaveArgs &A, const ShadowStack &q, uint64_t base,
                               uint32_t count, int ray_class, hipStream_t st)
{
    if (!count) return hipSuccess;
    ShadowSrc src = { q, base, count, A.out };
    hipError_t e = hipMemsetAsync(ctrl + 4, 0, 8, st);
    if (e != hipSuccess) return e;
    if (ray_class == 0) {
        RetrySrc<ShadowSrc> again = { src, retry, ctrl + 5 };
        hipLaunchKernelGGL(k_sweep_shadow, dim3(sweep_blocks(count, num_cu)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 5, retry, A.counters,
                           A.kstats ? A.kstats + RT_KSTAT_WORDS * RT_KSTAT_SWEEP_SHADOW : nullptr);
        hipLaunchKernelGGL(k_trace_shadow_retry, dim3(engine_blocks(count, num_cu, (const void *)k_trace_shadow_retry)), dim3(RT_BLOCK), 0, st, S, again, ctrl + 4,
                           (uint32_t)RT_BATCH1, A.counters, A.kstats ? A.kstats + RT_KSTAT_WORDS * RT_KSTAT_SWEEP_SHADOW : nullptr);
        return hipGetLastError();
    }
    uint64_t *const ks = A.kstats ? A.kstats + RT_KSTAT_WORDS * RT_KSTAT_ENGINE_SHADOW : nullptr;
    if (S.any_chain) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(engine_blocks(count, num_cu, (const void *)k_trace_shadow<true>)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 4,
                                        (uint32_t)RT_BATCH1, A.counters, ks);
    else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(engine_blocks(count, num_cu, (const void *)k_trace_shadow<false>)), dim3(RT_BLOCK), 0, st, S, src, ctrl + 4,
                            (uint32_t)RT_BATCH1, A.counters, ks);
    return hipGetLastError();
}
hipError_t launch_shade(const DevScene &S, const WaveArgs &A, const RayStack &src, uint64_t base, uint32_t count,
                        const RayDest &dst, hipStream_t st)
{
    if (!count) return hipSuccess;
    const dim3 grid((count + RT_BLOCK - 1) / RT_BLOCK), block(RT_BLOCK);
    switch (S.shade_kinds) {
    case 0: hipLaunchKernelGGL(k_shade<0>, grid, block, 0, st, S, A, src, base, count, dst); break;
    case 1: hipLaunchKernelGGL(k_shade<1>, grid, block, 0, st, S, A, src, base, count, dst); break;
    case 2: hipLaunchKernelGGL(k_shade<2>, grid, block, 0, st, S, A, src, base, count, dst); break;
    case 3: hipLaunchKernelGGL(k_shade<3>, grid, block, 0, st, S, A, src, base, count, dst); break;
    default: hipLaunchKernelGGL(k_shade<7>, grid, block, 0, st, S, A, src, base, count, dst); break;
    }
    return hipGetLastError();
}
// ------------------------------------------------------------------ static workspace
// Device memory that exists without an allocation call, for callers that may not make one: each
// device that loads this code object holds its own copy of the array, at its own address.  In host
// code the array's name only stands for the symbol; the address on a device comes from
// hipGetSymbolAddress with that device current, and that address (never the name) is what kernels,
// memsets and copies are given.
static __device__ __attribute__((used)) __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

hipError_t static_scratch(void **ptr)
{
    enum { MAX_DEVICES = 64 };
    static std::mutex mu;
    static void *addr[MAX_DEVICES] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!addr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        addr[dev] = p;
    }
    *ptr = addr[dev];
    return hipSuccess;
}

hipError_t launch_resolve(const double *sum, uint64_t npix, int spp, int tone, uint8_t *bgr, hipStream_t st)
{
    if (npix == 0) return hipSuccess;
    double inv = 1.0 / spp;                                   // scene.cpp:112
    hipLaunchKernelGGL(k_resolve_bgr8, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, sum, npix, inv, tone, bgr);
    return hipGetLastError();
}

} // namespace rt
