/*
 * rt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see rt_oracle.h).
 *
 * A plain-C restatement of the reference path tracer's hot path.  Every function names the
 * reference lines it follows.  Arithmetic order is kept operation for operation (fp64, no
 * FMA contraction: build with -ffp-contract=off) because primary-hit primitive ids are
 * compared bit-exactly with the HIP kernels.
 *
 * Deliberate, documented deviations from the reference:
 *   - libc rand() is replaced by a counter-based generator keyed on (seed, pixel, sample,
 *     path node, slot); uniforms are 53-bit on [0,1) instead of 15-bit on [0,1].
 *   - the per-pixel shuffled jitter table (RTObject.cpp:46-69) is evaluated lazily through a
 *     keyed permutation: same structure (every stratum used exactly once per pixel and
 *     object, one jitter pair per stratum), different random numbers.
 *   - a triangle primitive and a Mesh drawable exist (the reference has none); they plug in
 *     where BoxNode / HeartShape do (Surface.cpp:43-59, 268-277).  Mesh::hit keeps rec->ray
 *     intact (HeartShape's path zeroes it, octree.cpp:312,372).
 *   - octree depth guard, "no hit child -> pop" instead of spinning (octree.cpp:316,337-347).
 *   - optional face-forward normal for the diffuse lobe (off = reference behaviour).
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* definitions.h:5-8 */
#define RT_INFINITE 1e10
#define RT_ZERO 1e-9
#define RT_PI 3.1415926

static __thread char g_err[256];
const char *orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ------------------------------------------------------------------ Vec3, types.h:8-93 */
typedef struct { double x, y, z; } v3;
static inline v3 V(double x, double y, double z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vdivv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* operator^, types.h:66-74 */
static inline v3 vcross(v3 l, v3 r)
{
    double a = l.x, b = l.y, c = l.z, d = r.x, e = r.y, f = r.z;
    return V(b * f - e * c, c * d - a * f, a * e - d * b);
}
static inline double vlen(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
/* types.h:19-23: multiply by the reciprocal, not divide */
static inline v3 vnorm(v3 a) { return vmul(a, 1 / vlen(a)); }
/* types.h:29-31 */
static inline int viszero(v3 a) { return fabs(a.x) < RT_ZERO && fabs(a.y) < RT_ZERO && fabs(a.z) < RT_ZERO; }
/* Vec3::reflact, types.h:37-40 */
static inline v3 vreflact(v3 d, v3 n)
{
    v3 proj = vmul(vneg(n), vdot(d, n));
    return vnorm(vadd(vmul(proj, 2), d));
}
static inline v3 v3p(const double *p) { return V(p[0], p[1], p[2]); }
static inline void v3st(double *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

typedef struct { v3 e, d; } ray_t;
static inline ray_t rayp(const orc_ray *r) { ray_t o; o.e = v3p(r->e); o.d = v3p(r->d); return o; }

/* std::min / std::max semantics (they matter when a NaN or a tie is involved) */
static inline double stdmin(double a, double b) { return (b < a) ? b : a; }
static inline double stdmax(double a, double b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------ counter-based RNG */
static inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
uint64_t orc_hash2(uint64_t a, uint64_t b)
{
    return mix64((a ^ (b * 0xD6E8FEB86659FD93ULL)) + 0x9E3779B97F4A7C15ULL);
}
double orc_u01(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }

/* keyed permutation of [0,l) (Kensler, "Correlated Multi-Jittered Sampling", 2013) */
uint32_t orc_permute(uint32_t i, uint32_t l, uint32_t p)
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

#define KEY_JITTER 0x4a495454ULL /* + object id */
#define KEY_PATH   0x50415448ULL
#define SLOT_BASE  16            /* draw slots live above the branch codes 1..4 */
enum { BR_DIFFUSE = 1, BR_SPECULAR = 2, BR_REFLECT = 3, BR_REFRACT = 4 };
enum { SLOT_RR = 0, SLOT_DIFF1 = 1, SLOT_DIFF2 = 2, SLOT_DISK = 3 };

/* Entry `k` of the n x m jitter table of object `object_id` for this pixel.
 * RTObject.cpp:53-58 builds pair (j*nstep + nstep*xi, i*mstep + mstep*xi') row-major over
 * i<m, j<n, then shuffles (:61-65); here stratum sigma(k) plays the shuffled slot. */
void orc_jitter_pair(uint64_t seed, uint64_t pixel, uint32_t object_id, uint32_t k,
                     uint32_t n, uint32_t m, double out[2])
{
    uint64_t kp = orc_hash2(orc_hash2(seed, pixel), KEY_JITTER + object_id);
    uint32_t sigma = orc_permute(k, n * m, (uint32_t)(kp >> 32));
    uint32_t j = sigma % n, i = sigma / n;
    double nstep = 1.0 / n, mstep = 1.0 / m;
    double xi1 = orc_u01(orc_hash2(kp, 2ULL * sigma));
    double xi2 = orc_u01(orc_hash2(kp, 2ULL * sigma + 1));
    out[0] = j * nstep + nstep * xi1;
    out[1] = i * mstep + mstep * xi2;
}
uint64_t orc_path_root_key(uint64_t seed, uint64_t pixel, uint32_t k)
{
    return orc_hash2(orc_hash2(orc_hash2(seed, pixel), KEY_PATH), k);
}
static inline double draw(uint64_t key, uint32_t slot) { return orc_u01(orc_hash2(key, SLOT_BASE + slot)); }

/* ------------------------------------------------------------------ hitBox, algorithms.h:208-236 */
static int hit_box(v3 lo, v3 hi, ray_t r, double t0, double t1, double *rt0, double *rt1)
{
    v3 vt0 = vdivv(vsub(lo, r.e), r.d);
    v3 vt1 = vdivv(vsub(hi, r.e), r.d);
    double tmin[4] = { stdmin(vt0.x, vt1.x), stdmin(vt0.y, vt1.y), stdmin(vt0.z, vt1.z), t0 };
    double tmax[4] = { stdmax(vt0.x, vt1.x), stdmax(vt0.y, vt1.y), stdmax(vt0.z, vt1.z), t1 };
    double r0 = tmin[0], r1 = tmax[0];
    for (int i = 1; i < 4; i++) {          /* std::max_element / std::min_element */
        if (r0 < tmin[i]) r0 = tmin[i];
        if (tmax[i] < r1) r1 = tmax[i];
    }
    if (r0 <= r1) { *rt0 = r0; *rt1 = r1; return 1; }
    return 0;
}
int orc_hit_box(const double lo[3], const double hi[3], const orc_ray *r, double t0, double t1,
                double *r0, double *r1)
{
    double a = 0, b = 0;
    int h = hit_box(v3p(lo), v3p(hi), rayp(r), t0, t1, &a, &b);
    if (r0) *r0 = a;
    if (r1) *r1 = b;
    return h;
}

/* getBoundPoint, algorithms.h:299-321: corner order, not Morton */
static v3 bound_point(int index, v3 lo, v3 hi)
{
    int ix, iy;
    switch (index % 4) {
    case 0: ix = 0; iy = 0; break;
    case 1: ix = 1; iy = 0; break;
    case 2: ix = 1; iy = 1; break;
    default: ix = 0; iy = 1; break;
    }
    int iz = (index >> 2) & 1;
    return V(ix ? hi.x : lo.x, iy ? hi.y : lo.y, iz ? hi.z : lo.z);
}
/* OctreeNode::getChildBound, octree.h:82-120 */
static void child_bound(v3 lo, v3 hi, int c, v3 *clo, v3 *chi)
{
#define P(i) bound_point(i, lo, hi)
#define MID(a, b) vdivs(vadd(P(a), P(b)), 2)
    switch (c % 8) {
    case 0: *clo = P(0);      *chi = MID(0, 6); break;
    case 1: *clo = MID(0, 1); *chi = MID(1, 6); break;
    case 2: *clo = MID(0, 2); *chi = MID(2, 6); break;
    case 3: *clo = MID(0, 3); *chi = MID(3, 6); break;
    case 4: *clo = MID(0, 4); *chi = MID(5, 7); break;
    case 5: *clo = MID(1, 4); *chi = MID(5, 6); break;
    case 6: *clo = MID(1, 7); *chi = P(6);      break;
    default: *clo = MID(0, 7); *chi = MID(6, 7); break;
    }
#undef P
#undef MID
}
void orc_child_bound(const double lo[3], const double hi[3], int child, double clo[3], double chi[3])
{
    v3 a, b;
    child_bound(v3p(lo), v3p(hi), child, &a, &b);
    v3st(clo, a); v3st(chi, b);
}
/* isBoundBoxOverLap, algorithms.h:329-333 (strict) */
static int box_overlap(v3 alo, v3 ahi, v3 blo, v3 bhi)
{
    return (stdmax(alo.x, blo.x) < stdmin(ahi.x, bhi.x)) &&
           (stdmax(alo.y, blo.y) < stdmin(ahi.y, bhi.y)) &&
           (stdmax(alo.z, blo.z) < stdmin(ahi.z, bhi.z));
}

/* ------------------------------------------------------------------ primitives */
/* Sphere::hit, Surface.cpp:73-111.  The tangent branch bypasses the [t0,t1] window. */
static int sphere_hit(v3 c, double radius, ray_t r, double t0, double t1, double *tout)
{
    v3 oc = vsub(r.e, c);
    double doc = vdot(r.d, oc);
    double dieta = doc * doc - vdot(r.d, r.d) * (vdot(oc, oc) - radius * radius);
    double deno = vdot(r.d, r.d);
    double pre = vdot(vneg(r.d), oc);
    double t;
    if (fabs(dieta) < RT_ZERO) {
        t = pre / deno;
    } else if (dieta > 0) {
        double pt = (pre - sqrt(dieta)) / deno;
        double nt = (pre + sqrt(dieta)) / deno;
        int f1 = 0, f2 = 0;
        if (pt > t1 || pt < t0) f1 = 1;
        if (nt > t1 || nt < t0) f2 = 1;
        if (f1 && f2) return 0;
        t = !f1 ? pt : nt;
    } else {
        return 0; /* dieta < 0; a NaN discriminant is treated as a miss (UB in the reference) */
    }
    *tout = t;
    return 1;
}
int orc_sphere_hit(const double c[3], double radius, const orc_ray *r, double t0, double t1, double *t)
{
    return sphere_hit(v3p(c), radius, rayp(r), t0, t1, t);
}
/* Ground / Wall_z / Wall_x ::hit, Surface.cpp:124-133, 194-204, 217-226 */
static int plane_hit(int type, double coord, ray_t r, double t0, double t1, double *tout)
{
    double e, d;
    if (type == ORC_OBJ_GROUND) { e = r.e.y; d = r.d.y; }
    else if (type == ORC_OBJ_WALL_Z) { e = r.e.z; d = r.d.z; }
    else { e = r.e.x; d = r.d.x; }
    if (e > coord && d < 0) {
        double t = (e - coord) / -d;
        if (t < t0 || t > t1) return 0;
        *tout = t;
        return 1;
    }
    return 0;
}
int orc_plane_hit(int type, double coord, const orc_ray *r, double t0, double t1, double *t)
{
    return plane_hit(type, coord, rayp(r), t0, t1, t);
}
/* New primitive (no reference counterpart): two-sided Moeller-Trumbore.  The barycentric
 * tests are made on the un-normalised numerators against |det| (exact sign flips instead of
 * a multiplication by 1/det), so the only division is the one that produces t for a ray that
 * passed them.  Takes the place of BoxNode::calHit (Surface.cpp:43-59) as the octree payload. */
#define TRI_DET_EPS 1e-200
#define TRI_GUARD2 1e-10
static int triangle_hit(v3 v0, v3 e1, v3 e2, ray_t r, double t0, double t1, double *tout)
{
    v3 p = vcross(r.d, e2);
    double det = vdot(e1, p);
    double a = fabs(det);
    if (!(a >= TRI_DET_EPS)) return 0;
    /* a determinant within a few orders of its rounding error (the ray runs in the triangle's plane, or the
     * triangle has no area: sine of its angle at v0 times cosine between ray and normal under 1e-5) is no hit:
     * the quotients below would be noise over noise */
    if (!(det * det >= TRI_GUARD2 * vdot(e1, e1) * vdot(e2, e2) * vdot(r.d, r.d))) return 0;
    v3 s = vsub(r.e, v0);
    double up = vdot(s, p);
    double us = det < 0 ? -up : up;
    if (!(us >= 0.0 && us <= a)) return 0;
    v3 q = vcross(s, e1);
    double vp = vdot(r.d, q);
    double vs = det < 0 ? -vp : vp;
    if (!(vs >= 0.0 && us + vs <= a)) return 0;
    double t = vdot(e2, q) / det;
    if (!(t >= t0 && t <= t1)) return 0;
    *tout = t;
    return 1;
}
int orc_triangle_hit(const double tri[9], const orc_ray *r, double t0, double t1, double *t)
{
    v3 v0 = v3p(tri), v1 = v3p(tri + 3), v2 = v3p(tri + 6);
    return triangle_hit(v0, vsub(v1, v0), vsub(v2, v0), rayp(r), t0, t1, t);
}

/* ------------------------------------------------------------------ octree, octree.h/.cpp */
typedef struct onode {
    v3 lo, hi;
    struct onode *child[8];
    int full;
    int depth;
    uint32_t *objs; /* std::list order */
    uint32_t nobj, cap;
} onode;

typedef struct {
    const double *tris;   /* borrowed copy owned by scene */
    uint64_t ntris;
    v3 *blo, *bhi;        /* padded triangle AABBs */
    onode *root;
    int max_size, tol, max_depth;
    uint64_t lost;
    int boxes;            /* the primitives are the boxes blo/bhi themselves (BoxNode) */
} otree;

static onode *onode_new(v3 lo, v3 hi, int depth)
{
    onode *n = (onode *)calloc(1, sizeof *n);
    n->lo = lo; n->hi = hi; n->depth = depth;
    return n;
}
static void onode_free(onode *n)
{
    if (!n) return;
    for (int i = 0; i < 8; i++) onode_free(n->child[i]);
    free(n->objs);
    free(n);
}
static void onode_push(onode *n, uint32_t o)
{
    if (n->nobj == n->cap) {
        n->cap = n->cap ? n->cap * 2 : 16;
        n->objs = (uint32_t *)realloc(n->objs, n->cap * sizeof(uint32_t));
    }
    n->objs[n->nobj++] = o;
}
/* OctreeNode::insertObject, octree.cpp:61-99 */
static void onode_insert(otree *t, onode *n, uint32_t o)
{
    if (!n->full) {
        onode_push(n, o);
        if ((int)n->nobj == t->max_size && n->depth < t->max_depth) {
            n->full = 1;
            for (int i = 0; i < 8; i++) {           /* createChildren, octree.cpp:101-108 */
                v3 clo, chi;
                child_bound(n->lo, n->hi, i, &clo, &chi);
                n->child[i] = onode_new(clo, chi, n->depth + 1);
            }
            /* re-insert every element in list order; kept ones are appended again */
            uint32_t tot = n->nobj;
            uint32_t *old = n->objs;
            n->objs = NULL; n->nobj = 0; n->cap = 0;
            for (uint32_t k = 0; k < tot; k++) onode_insert(t, n, old[k]);
            free(old);
        }
    } else {
        int a[8], count = 0;
        for (int i = 0; i < 8; i++) {
            v3 clo, chi;
            child_bound(n->lo, n->hi, i, &clo, &chi);
            a[i] = box_overlap(t->blo[o], t->bhi[o], clo, chi);
            count += a[i];
        }
        if (count > t->tol) {
            onode_push(n, o);
        } else {
            if (count == 0) t->lost++;              /* strict overlap: silently dropped */
            for (int i = 0; i < 8; i++)
                if (a[i]) onode_insert(t, n->child[i], o);
        }
    }
}

typedef struct { uint64_t rays, inner, leaf, prim; } counters_t;

/* BoxNode::calHit, Surface.cpp:43-59: a voxel is hit where the ray enters it, moved back towards
 * the ray origin by the voxel's diagonal (the second assignment of rec->t is the one that counts) */
static int box_node_hit(v3 lo, v3 hi, ray_t r, double t0, double t1, double *tout)
{
    double rt0, rt1;
    if (!hit_box(lo, hi, r, t0, t1, &rt0, &rt1)) return 0;
    *tout = rt0 - vlen(vsub(lo, hi)) / vlen(r.d);
    return 1;
}

typedef struct {
    double t;
    v3 normal;
    int mat;
    int obj;        /* index in group order */
    uint32_t prim;  /* triangle index inside a mesh */
    ray_t ray;
    v3 localp;      /* written by a HeartShape hit only (Surface.cpp:275) */
} hitrec;

/* OctreeNode::hit, octree.cpp:19-38.  Returns 1 when some object was hit; best and bprim are
 * written only by an object whose t <= the running minimum (last equal wins). */
static int onode_hit(const otree *t, const onode *n, ray_t r, double t0, double t1,
                     double *best, uint32_t *bprim, counters_t *c)
{
    double bt0, bt1;
    if (!hit_box(n->lo, n->hi, r, t0, t1, &bt0, &bt1)) return 0;
    double tempt = bt1;
    int flag = 0;
    for (uint32_t k = 0; k < n->nobj; k++) {
        uint32_t o = n->objs[k];
        double tt;
        if (c) c->prim++;
        int h;
        if (t->boxes) {
            h = box_node_hit(t->blo[o], t->bhi[o], r, bt0, bt1, &tt);
        } else {
            const double *tr = t->tris + 9 * (uint64_t)o;
            v3 v0 = v3p(tr), v1 = v3p(tr + 3), v2 = v3p(tr + 6);
            h = triangle_hit(v0, vsub(v1, v0), vsub(v2, v0), r, bt0, bt1, &tt);
        }
        if (h) {
            flag = 1;
            if (tt <= tempt) { tempt = tt; *best = tt; *bprim = o; }
        }
    }
    return flag;
}

#define OSTACK 64
/* Octree::hit, octree.cpp:285-376: ordered DFS with an explicit stack */
static int otree_hit(const otree *t, ray_t r, double t0, double t1, double *tout, uint32_t *pout,
                     counters_t *c)
{
    double d0, d1;
    if (!hit_box(t->root->lo, t->root->hi, r, t0, t1, &d0, &d1)) return 0;
    const onode *stack[OSTACK];
    int nchild[OSTACK], index[OSTACK];
    int cidx[OSTACK][8];
    double ct[OSTACK][8];
    int sp = 0;
    stack[0] = t->root; nchild[0] = 0; index[0] = -1; sp = 1;
    double mint = t1 + 1;
    uint32_t minprim = ORC_MISS;
    while (sp > 0) {
        int top = sp - 1;
        if (nchild[top] == 0) {                    /* node not processed yet */
            const onode *n = stack[top];
            double tt = 0; uint32_t pp = ORC_MISS;
            int flag = 0;
            if (c) { if (n->full) c->inner++; else c->leaf++; }
            if (onode_hit(t, n, r, t0, t1, &tt, &pp, c)) {
                if (pp != ORC_MISS && tt < mint) { mint = tt; minprim = pp; flag = 1; }
            }
            if (!n->full) {
                if (flag) break;                   /* leaf improved the minimum: done */
                sp--;
            } else {
                int k = 0;
                for (int i = 0; i < 8; i++) {
                    double e0, e1;
                    if (hit_box(n->child[i]->lo, n->child[i]->hi, r, t0, t1, &e0, &e1)) {
                        cidx[top][k] = i; ct[top][k] = e0; k++;
                    }
                }
                /* std::sort by entry t (octree.cpp:344); ties are resolved by child index
                 * (insertion sort is stable), the documented rule for exact ties */
                for (int a = 1; a < k; a++) {
                    int ci = cidx[top][a]; double cv = ct[top][a]; int b = a - 1;
                    while (b >= 0 && cv < ct[top][b]) { cidx[top][b + 1] = cidx[top][b]; ct[top][b + 1] = ct[top][b]; b--; }
                    cidx[top][b + 1] = ci; ct[top][b + 1] = cv;
                }
                nchild[top] = k;
                if (k == 0) sp--;                  /* the reference would spin here */
            }
        } else {
            index[top]++;
            if (index[top] == nchild[top]) {
                sp--;
            } else {
                if (mint < ct[top][index[top]]) break;
                if (sp >= OSTACK) break;
                stack[sp] = stack[top]->child[cidx[top][index[top]]];
                nchild[sp] = 0; index[sp] = -1; sp++;
            }
        }
    }
    if (mint <= t1 && minprim != ORC_MISS) { *tout = mint; *pout = minprim; return 1; }
    return 0;
}

/* ------------------------------------------------------------------ implicit surfaces */
/* HeartShape::heartImplicitFunc (Surface.h:218-224), SurfaceOfGenus2Function::ImplicitFunc
 * (Surface.cpp:318-325); both swap y and z */
static double implicit_func(int kind, v3 pos)
{
    double x = pos.x, y = pos.z, z = pos.y;
    if (kind == ORC_OBJ_HEART) {
        double temp = x * x + y * y * 9.0 / 4 + z * z - 1;
        return -(x * x * z * z * z) - y * y * z * z * z * 9.0 / 80 + temp * temp * temp;
    }
    double temp = x * x + y * y;
    return 2 * y * (y * y - 3 * x * x) * (1 - z * z) + temp * temp - (9 * z * z - 1) * (1 - z * z);
}
double orc_implicit_func(int kind, const double pos[3]) { return implicit_func(kind, v3p(pos)); }
/* HeartShape::getRawNormal (Surface.cpp:239-265) and SurfaceOfGenus2Function::getNormal
 * (Surface.cpp:333-343) on a point of the surface's own frame */
static v3 implicit_normal(int kind, v3 pos)
{
    double x = pos.x, y = pos.z, z = pos.y;
    v3 n;
    if (kind == ORC_OBJ_HEART) {
        double temp = x * x + y * y * 9.0 / 4 + z * z - 1;
        temp = 3 * temp * temp;
        temp = stdmax(temp, 0.5);
        n.x = -2 * x * z * z * z + temp * 2 * x;
        n.z = -9.0 / 40 * y * z * z * z + temp * 9.0 / 2 * y;
        n.y = -3 * x * x * z * z - 27.0 / 80 * y * y * z * z + 2 * temp * z;
    } else {
        n.x = -12 * x * y * (1 - z * z) + 4 * (x * x + y * y) * x;
        n.z = 6 * (1 - z * z) * (y * y - x * x) + 4 * (x * x + y * y) * y;
        n.y = 2 * y * (y * y - 3 * x * x) * -2 * z + 36 * z * z * z - 20 * z;
    }
    return vnorm(n);
}
void orc_implicit_normal(int kind, const double pos[3], double out[3]) { v3st(out, implicit_normal(kind, v3p(pos))); }

/* algorithm::calSingleRoot, algorithms.h:143-172: bisection down to |f| <= 10e-8 */
static int single_root(int kind, v3 sp, v3 unit, double t1, double t2, double *root)
{
#define F(t) implicit_func(kind, vadd(vmul(unit, (t)), sp))
    double f1 = F(t1), f2 = F(t2);
    if (f1 == 0.0) { *root = t1; return 0; }
    if (f2 == 0.0) { *root = t2; return 0; }
    if (f1 * f2 > 0.0) return -1;
    double tm = (t1 + t2) / 2, fm = F(tm);
    while (fabs(fm) > 10e-8) {
        if (f1 * fm < 0) { t2 = tm; f2 = fm; } else { t1 = tm; f1 = fm; }
        tm = (t1 + t2) / 2;
        fm = F(tm);
    }
#undef F
    *root = tm;
    return 0;
}

/* algorithm::getAdjBox, algorithms.h:251-278 (faces FRONT, TOP, LEFT, RIGHT, DOWN, BACK) */
static void adj_box(v3 lo, v3 hi, int face, v3 *alo, v3 *ahi)
{
    switch (face) {
    case 0: *alo = vsub(lo, V(0, hi.y - lo.y, 0)); *ahi = bound_point(5, lo, hi); break;
    case 1: *alo = bound_point(4, lo, hi); *ahi = vadd(hi, V(0, 0, hi.z - lo.z)); break;
    case 2: *alo = vsub(lo, V(hi.x - lo.x, 0, 0)); *ahi = bound_point(7, lo, hi); break;
    case 3: *alo = bound_point(1, lo, hi); *ahi = vadd(hi, V(hi.x - lo.x, 0, 0)); break;
    case 4: *alo = vsub(lo, V(0, 0, hi.z - lo.z)); *ahi = bound_point(2, lo, hi); break;
    default: *alo = bound_point(3, lo, hi); *ahi = vadd(hi, V(0, hi.y - lo.y, 0)); break;
    }
}

/* The set of voxels, keyed the way BoxNode::operator< (Surface.h:30-44) sees them: two boxes whose
 * lower corners are closer than 1e-9 are the same element, otherwise they order by x, then y,
 * then z.  Voxel corners reached along different paths differ by rounding only (~1e-13), distinct
 * voxels by a whole edge, so the integer cell of the lower corner is an exact key. */
typedef struct { int64_t ix, iy, iz; uint32_t box; } vcell;
typedef struct { vcell *slot; uint64_t cap, n; } vset;
static uint64_t vcell_hash(int64_t x, int64_t y, int64_t z)
{
    return mix64((uint64_t)x * 0x9E3779B97F4A7C15ULL ^ mix64((uint64_t)y * 0xC2B2AE3D27D4EB4FULL ^ mix64((uint64_t)z)));
}
static int vset_find_or_add(vset *s, int64_t x, int64_t y, int64_t z, uint32_t box, int add)
{
    if ((s->n + 1) * 2 > s->cap) {
        uint64_t ncap = s->cap ? s->cap * 2 : 1024;
        vcell *ns = (vcell *)malloc(ncap * sizeof(vcell));
        for (uint64_t i = 0; i < ncap; i++) ns[i].box = ORC_MISS;
        for (uint64_t i = 0; i < s->cap; i++) {
            if (s->slot[i].box == ORC_MISS) continue;
            uint64_t h = vcell_hash(s->slot[i].ix, s->slot[i].iy, s->slot[i].iz) & (ncap - 1);
            while (ns[h].box != ORC_MISS) h = (h + 1) & (ncap - 1);
            ns[h] = s->slot[i];
        }
        free(s->slot); s->slot = ns; s->cap = ncap;
    }
    uint64_t h = vcell_hash(x, y, z) & (s->cap - 1);
    while (s->slot[h].box != ORC_MISS) {
        if (s->slot[h].ix == x && s->slot[h].iy == y && s->slot[h].iz == z) return 1;
        h = (h + 1) & (s->cap - 1);
    }
    if (add) { s->slot[h].ix = x; s->slot[h].iy = y; s->slot[h].iz = z; s->slot[h].box = box; s->n++; }
    return 0;
}
static int vcell_cmp(const void *a, const void *b)
{
    const vcell *p = (const vcell *)a, *q = (const vcell *)b;
    if (p->ix != q->ix) return p->ix < q->ix ? -1 : 1;
    if (p->iy != q->iy) return p->iy < q->iy ? -1 : 1;
    if (p->iz != q->iz) return p->iz < q->iz ? -1 : 1;
    return 0;
}
static int same_sign(const double a[8])
{
    int pflag = 1, nflag = 1;
    for (int i = 0; i < 8; i++) { if (a[i] > 0) nflag = 0; if (a[i] < 0) pflag = 0; }
    return pflag || nflag;
}
#define ORC_MAX_VOXELS (1u << 24)
/* ImplicitSurfaceOctree::build, octree.cpp:148-254: find the surface between the two seed points,
 * put a voxel of half edge min_size around that point and flood-fill over face neighbours whose
 * eight corner values change sign; every voxel goes into the octree in the set's order. */
static int implicit_tree_build(otree *t, int kind, double min_size)
{
    v3 s0 = V(0, 0, 0), s1 = kind == ORC_OBJ_HEART ? V(2, 0, 0) : V(0, 0.9, 0);   /* Surface.cpp:302, 327-331 */
    v3 unit = vnorm(vsub(s1, s0));
    double root;
    if (single_root(kind, s0, unit, 0, vlen(vsub(s1, s0)), &root)) FAIL("implicit surface: no root between the seed points");
    v3 midp = vadd(vmul(unit, root), s0);
    v3 olo = vsub(midp, V(min_size, min_size, min_size)), ohi = vadd(midp, V(min_size, min_size, min_size));
    const double edge = 2 * min_size;
    uint64_t cap = 1024, n = 0, head = 0;
    v3 *lo = (v3 *)malloc(cap * sizeof(v3)), *hi = (v3 *)malloc(cap * sizeof(v3));
    vset set = { NULL, 0, 0 };
    lo[0] = olo; hi[0] = ohi; n = 1;
    vset_find_or_add(&set, 0, 0, 0, 0, 1);
    v3 blo = V(RT_INFINITE, RT_INFINITE, RT_INFINITE), bhi = V(-RT_INFINITE, -RT_INFINITE, -RT_INFINITE);
    int rc = 0;
    while (head < n) {                                  /* boxq is a FIFO: boxes are visited in the order they were found */
        v3 nlo = lo[head], nhi = hi[head];
        head++;
        blo.x = stdmin(blo.x, nlo.x); blo.y = stdmin(blo.y, nlo.y); blo.z = stdmin(blo.z, nlo.z);
        bhi.x = stdmax(bhi.x, nhi.x); bhi.y = stdmax(bhi.y, nhi.y); bhi.z = stdmax(bhi.z, nhi.z);
        for (int f = 0; f < 6; f++) {
            v3 alo, ahi;
            adj_box(nlo, nhi, f, &alo, &ahi);
            int64_t ix = llround((alo.x - olo.x) / edge), iy = llround((alo.y - olo.y) / edge), iz = llround((alo.z - olo.z) / edge);
            if (vset_find_or_add(&set, ix, iy, iz, 0, 0)) continue;
            double pv[8];
            for (int i = 0; i < 8; i++) pv[i] = implicit_func(kind, bound_point(i, alo, ahi));
            if (same_sign(pv)) continue;
            if (n >= ORC_MAX_VOXELS) { rc = -1; break; }
            if (n == cap) { cap *= 2; lo = (v3 *)realloc(lo, cap * sizeof(v3)); hi = (v3 *)realloc(hi, cap * sizeof(v3)); }
            lo[n] = alo; hi[n] = ahi;
            vset_find_or_add(&set, ix, iy, iz, (uint32_t)n, 1);
            n++;
        }
        if (rc) break;
    }
    if (rc) { free(lo); free(hi); free(set.slot); FAIL("implicit surface: too many voxels (min size too small)"); }
    /* the set's iteration order */
    vcell *order = (vcell *)malloc(n * sizeof(vcell));
    uint64_t k = 0;
    for (uint64_t i = 0; i < set.cap; i++) if (set.slot[i].box != ORC_MISS) order[k++] = set.slot[i];
    qsort(order, n, sizeof(vcell), vcell_cmp);
    t->blo = (v3 *)malloc(n * sizeof(v3)); t->bhi = (v3 *)malloc(n * sizeof(v3));
    for (uint64_t i = 0; i < n; i++) { t->blo[i] = lo[order[i].box]; t->bhi[i] = hi[order[i].box]; }
    free(order); free(lo); free(hi); free(set.slot);
    t->boxes = 1; t->ntris = n; t->tris = NULL;
    t->root = onode_new(blo, bhi, 0);                   /* setBoundBox(box); Octree::build() */
    for (uint64_t i = 0; i < n; i++) onode_insert(t, t->root, (uint32_t)i);
    return 0;
}

/* ------------------------------------------------------------------ Perlin materials */
typedef struct { const int32_t *P; const double *G; int n; } perlin_t;
static inline int perlin_fai(const perlin_t *t, int v)
{
    int m = v % t->n;
    return t->P[m >= 0 ? m : t->n + m];
}
static inline double perlin_weight(double t)
{
    double at = fabs(t);
    if (at < 1) return 2 * at * at * at - 3 * at * at + 1;
    return 0.0;
}
/* algorithm::perlinNoise, algorithms.h:56-87 */
static double perlin_noise(const perlin_t *t, double x, double y, double z, double scale)
{
    x /= scale; y /= scale; z /= scale;
    double result = 0;
    for (int n = 0; n < 8; n++) {
        int i = (n & 0x01) ? floor(x) : floor(x) + 1;
        int j = (n & 0x02) ? floor(y) : floor(y) + 1;
        int k = (n & 0x04) ? floor(z) : floor(z) + 1;
        double u = x - i, v = y - j, w = z - k;
        const double *g = t->G + 3 * perlin_fai(t, i + perlin_fai(t, j + perlin_fai(t, k)));
        result += perlin_weight(u) * perlin_weight(v) * perlin_weight(w) * (g[0] * u + g[1] * v + g[2] * w);
    }
    return result;
}
double orc_perlin_noise(double x, double y, double z, double scale, const int32_t *P, const double *G, int n)
{
    perlin_t t = { P, G, n };
    return perlin_noise(&t, x, y, z, scale);
}
static inline double perlin_normalized(const perlin_t *t, v3 p, double scale) { return perlin_noise(t, p.x, p.y, p.z, scale) / 2 + 0.5; }          /* algorithms.cpp:37-40 */
static inline double perlin_line(const perlin_t *t, v3 p, double scale) { return stdmin(fabs(perlin_noise(t, p.x, p.y, p.z, scale)) * 11, 1.0); }   /* algorithms.cpp:32-35 */

void orc_perlin_tables(uint32_t seed, int n, int32_t *P, double *G)
{
    uint32_t st = seed;                                    /* srand(SEED), algorithms.cpp:7 */
#define MSRAND() (st = st * 214013u + 2531011u, (int)((st >> 16) & 0x7fff))
    for (int i = 0; i < n; i++) {                          /* genRandomUnitVec, algorithms.h:37-46: keeps points OUTSIDE the unit ball */
        v3 a;
        do {
            a.x = MSRAND() * 1.0 / 0x7fff * 2 - 1;
            a.y = MSRAND() * 1.0 / 0x7fff * 2 - 1;
            a.z = MSRAND() * 1.0 / 0x7fff * 2 - 1;
        } while (vlen(a) < 1);
        a = vnorm(a);
        G[3 * i] = a.x; G[3 * i + 1] = a.y; G[3 * i + 2] = a.z;
    }
    for (int i = 0; i < n; i++) P[i] = i;
    for (int i = n - 1; i >= 1; i--) {                     /* suffle, algorithms.h:48-54 */
        int slct = (int)(MSRAND() * 1.0 / 0x7fff * (i + 1));
        if (slct > i) slct = i;                            /* rand() == RAND_MAX indexes one past the end in the reference */
        int32_t tmp = P[slct]; P[slct] = P[i]; P[i] = tmp;
    }
#undef MSRAND
}

void orc_perlin_time_changed(orc_perlin_clock *clk, double prev_time, double time, int n, double *G)
{
    if (time == clk->stime) return;
    double dt = time - prev_time;
    if (clk->count % 6 == 0) { clk->count = 1; clk->stage++; }   /* the vectors drawn here are never used */
    else clk->count++;
    for (int i = 0; i < n; i++) {
        v3 g = vadd(v3p(G + 3 * i), vmul(vmul(v3p(G + 3 * ((i + clk->stage) % n)), dt), 5));
        v3st(G + 3 * i, vnorm(g));
    }
    clk->stime = time;
}

/* The colours and the normal a material presents at a hit: plain materials return their fields;
 * PerlinNoiseNormalMaterial scales them by the noise at the position (material.h:104-108); the
 * _Blood variant mixes in a vein colour and tilts the normal along the noise gradient
 * (material.h:128-157, material.cpp:67-87).  setPosition runs in Node::calHit before the normal
 * is taken (Surface.h:272-283). */
typedef struct { v3 ambient, diffuse, specular, normal; } surf_t;
static surf_t material_at(const orc_material *m, const perlin_t *t, v3 pos, v3 normal)
{
    surf_t o = { v3p(m->ambient), v3p(m->color), v3p(m->specular), normal };
    if (m->perlin == ORC_PERLIN_NORMAL) {
        double noise = perlin_normalized(t, pos, m->perlin_scale);
        o.ambient = vmul(o.ambient, noise); o.diffuse = vmul(o.diffuse, noise); o.specular = vmul(o.specular, noise);
    } else if (m->perlin == ORC_PERLIN_BLOOD) {
        double noise = stdmin(perlin_normalized(t, pos, m->perlin_scale) * 1.5, 1.0);
        double linenoise = perlin_line(t, pos, m->perlin_scale);
        /* getNormal, material.cpp:73-87 */
        v3 vec = V(0, 0, 1);
        if (viszero(vcross(normal, vec))) vec = V(1, 0, 0);
        v3 b = vnorm(vcross(normal, vec));
        v3 tg = vnorm(vcross(b, normal));
        double dpb = linenoise - perlin_line(t, vadd(pos, vmul(tg, 0.001)), m->perlin_scale);
        double dpt = linenoise - perlin_line(t, vadd(pos, vmul(b, 0.001)), m->perlin_scale);
        o.normal = vnorm(vadd(vadd(normal, vmul(vmul(b, dpb), 60)), vmul(vmul(tg, dpt), 60)));
        double hf1 = stdmin(1.0, stdmax(0.0, stdmax(linenoise - 0.6, 0.6 - linenoise) * 2.5 - 0.05));
        double hf2 = stdmin(1.0, stdmax(0.0, stdmax(noise - 0.4, 0.4 - noise) * 5 - 0.05));
        double hf = hf1 * hf2;
        double ln = 1 - linenoise, high = ln * 1.7, nz = noise / 2.5;
        v3 centre = V(0.45, 0, 0);
        v3 veins = vmul(centre, high + nz);
        o.ambient = vmul(vmul(vadd(vmul(v3p(m->ambient), 1 - nz), veins), hf), 0.35);
        o.diffuse = vmul(vmul(vadd(vmul(v3p(m->color), 1 - nz), veins), hf), 1.0);
        /* getSpecular (material.h:150-157) is written without parentheses around its last `&`:
         * `+` binds tighter than `&` and `*` tighter still, so it is (S&n + C&(1-n)) & (hf*0.85) */
        o.specular = vmul(vadd(vmul(v3p(m->specular), noise), vmul(centre, 1 - noise)), hf * 0.85);
    }
    return o;
}
void orc_perlin_material(const orc_material *m, const int32_t *P, const double *G, int n,
                         const double pos[3], const double normal[3], double out[12])
{
    perlin_t t = { P, G, n };
    surf_t o = material_at(m, &t, v3p(pos), v3p(normal));
    v3st(out, o.ambient); v3st(out + 3, o.diffuse); v3st(out + 6, o.specular); v3st(out + 9, o.normal);
}

/* ------------------------------------------------------------------ scene */
struct orc_scene {
    orc_object *objects;     uint64_t nobjects;
    orc_material *materials; uint64_t nmaterials;
    orc_light *lights;       uint64_t nlights;
    otree *trees;            uint64_t nmeshes;
    double **tri_store;
    uint32_t *prim_base;     /* global primitive id of each object's first primitive */
    v3 background;
    perlin_t perlin;         /* tables of the Perlin materials (owned copies) */
    otree itree[2];          /* the voxel octree of each implicit kind (HeartShape::ptree is one static tree per class) */
    int has_itree[2];
};
static inline int is_implicit(int type) { return type == ORC_OBJ_HEART || type == ORC_OBJ_GENUS2; }

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    for (uint64_t i = 0; i < s->nmeshes; i++) {
        onode_free(s->trees[i].root);
        free(s->trees[i].blo); free(s->trees[i].bhi);
        free(s->tri_store[i]);
    }
    for (int k = 0; k < 2; k++) {
        if (!s->has_itree[k]) continue;
        onode_free(s->itree[k].root);
        free(s->itree[k].blo); free(s->itree[k].bhi);
    }
    free((void *)s->perlin.P); free((void *)s->perlin.G);
    free(s->trees); free(s->tri_store); free(s->objects); free(s->materials); free(s->lights);
    free(s->prim_base); free(s);
}

static void *dupmem(const void *p, size_t n)
{
    void *q = malloc(n ? n : 1);
    if (n) memcpy(q, p, n);
    return q;
}

orc_scene *orc_scene_create(const orc_scene_desc *d)
{
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->nobjects = d->nobjects; s->nmaterials = d->nmaterials; s->nlights = d->nlights; s->nmeshes = d->nmeshes;
    s->objects = (orc_object *)dupmem(d->objects, d->nobjects * sizeof(orc_object));
    s->materials = (orc_material *)dupmem(d->materials, d->nmaterials * sizeof(orc_material));
    s->lights = (orc_light *)dupmem(d->lights, d->nlights * sizeof(orc_light));
    s->trees = (otree *)calloc(d->nmeshes ? d->nmeshes : 1, sizeof(otree));
    s->tri_store = (double **)calloc(d->nmeshes ? d->nmeshes : 1, sizeof(double *));
    s->background = v3p(d->background);
    {
        int need = 0;
        for (uint64_t i = 0; i < d->nmaterials; i++) need |= d->materials[i].perlin != ORC_PERLIN_NONE;
        if (need) {
            int n = d->perlin_p && d->perlin_g && d->perlin_n > 0 ? d->perlin_n : 256;
            int32_t *P = (int32_t *)malloc(n * sizeof(int32_t));
            double *G = (double *)malloc(3 * n * sizeof(double));
            if (d->perlin_p && d->perlin_g && d->perlin_n > 0) { memcpy(P, d->perlin_p, n * sizeof(int32_t)); memcpy(G, d->perlin_g, 3 * n * sizeof(double)); }
            else orc_perlin_tables(12325, n, P, G);        /* SEED, algorithms.h:99 */
            s->perlin.P = P; s->perlin.G = G; s->perlin.n = n;
        }
    }
    for (uint64_t mi = 0; mi < d->nmeshes; mi++) {
        otree *t = &s->trees[mi];
        uint64_t n = d->meshes[mi].ntris;
        s->tri_store[mi] = (double *)dupmem(d->meshes[mi].tris, n * 9 * sizeof(double));
        t->tris = s->tri_store[mi]; t->ntris = n;
        t->max_size = d->octree_max_size > 0 ? d->octree_max_size : 128;
        t->tol = d->octree_overlap_tol > 0 ? d->octree_overlap_tol : 3;
        t->max_depth = d->octree_max_depth > 0 ? d->octree_max_depth : 12;
        if (t->max_depth > OSTACK - 2) t->max_depth = OSTACK - 2;
        t->blo = (v3 *)malloc((n ? n : 1) * sizeof(v3));
        t->bhi = (v3 *)malloc((n ? n : 1) * sizeof(v3));
        v3 lo = V(RT_INFINITE, RT_INFINITE, RT_INFINITE), hi = V(-RT_INFINITE, -RT_INFINITE, -RT_INFINITE);
        double pad = d->tri_box_pad;
        for (uint64_t k = 0; k < n; k++) {
            const double *tr = t->tris + 9 * k;
            v3 a = v3p(tr), b = v3p(tr + 3), c = v3p(tr + 6);
            v3 l = V(stdmin(stdmin(a.x, b.x), c.x) - pad, stdmin(stdmin(a.y, b.y), c.y) - pad, stdmin(stdmin(a.z, b.z), c.z) - pad);
            v3 h = V(stdmax(stdmax(a.x, b.x), c.x) + pad, stdmax(stdmax(a.y, b.y), c.y) + pad, stdmax(stdmax(a.z, b.z), c.z) + pad);
            t->blo[k] = l; t->bhi[k] = h;
            /* global box accumulation, octree.cpp:207-212 */
            lo.x = stdmin(lo.x, l.x); lo.y = stdmin(lo.y, l.y); lo.z = stdmin(lo.z, l.z);
            hi.x = stdmax(hi.x, h.x); hi.y = stdmax(hi.y, h.y); hi.z = stdmax(hi.z, h.z);
        }
        t->root = onode_new(lo, hi, 0);            /* Octree::build, octree.cpp:265-269 */
        for (uint64_t k = 0; k < n; k++) onode_insert(t, t->root, (uint32_t)k);
    }
    s->prim_base = (uint32_t *)calloc(d->nobjects + 1, sizeof(uint32_t));
    uint32_t base = 0;
    for (uint64_t i = 0; i < d->nobjects; i++) {
        s->prim_base[i] = base;
        const orc_object *o = &s->objects[i];
        if (o->type == ORC_OBJ_MESH) {
            if (o->mesh < 0 || (uint64_t)o->mesh >= d->nmeshes) { snprintf(g_err, sizeof g_err, "object %d: bad mesh index", (int)i); orc_scene_destroy(s); return NULL; }
            base += (uint32_t)s->trees[o->mesh].ntris;
        } else if (is_implicit(o->type)) {
            int k = o->type - ORC_OBJ_HEART;
            if (!s->has_itree[k]) {
                otree *t = &s->itree[k];
                t->max_size = d->octree_max_size > 0 ? d->octree_max_size : 128;
                t->tol = d->octree_overlap_tol > 0 ? d->octree_overlap_tol : 3;
                t->max_depth = d->octree_max_depth > 0 ? d->octree_max_depth : 12;
                if (t->max_depth > OSTACK - 2) t->max_depth = OSTACK - 2;
                if (implicit_tree_build(t, o->type, d->implicit_min_size > 0 ? d->implicit_min_size : 3e-3)) { orc_scene_destroy(s); return NULL; }
                s->has_itree[k] = 1;
            }
            base += (uint32_t)s->itree[k].ntris;
        } else {
            base += 1;
        }
        if (o->material < 0 || (uint64_t)o->material >= d->nmaterials) { snprintf(g_err, sizeof g_err, "object %d: bad material index", (int)i); orc_scene_destroy(s); return NULL; }
    }
    s->prim_base[d->nobjects] = base;
    return s;
}

static void stats_walk(const onode *n, uint64_t out[6])
{
    out[0]++;
    out[2] += n->nobj;
    if ((uint64_t)n->depth > out[4]) out[4] = n->depth;
    if (!n->full) { out[1]++; return; }
    out[3] += n->nobj;
    for (int i = 0; i < 8; i++) stats_walk(n->child[i], out);
}
int orc_octree_stats(const orc_scene *s, int mesh, uint64_t out[6])
{
    if (mesh < 0 || (uint64_t)mesh >= s->nmeshes) FAIL("bad mesh index");
    memset(out, 0, 6 * sizeof(uint64_t));
    stats_walk(s->trees[mesh].root, out);
    out[5] = s->trees[mesh].lost;
    return 0;
}

int orc_implicit_stats(const orc_scene *s, int kind, uint64_t out[6])
{
    if (!is_implicit(kind) || !s->has_itree[kind - ORC_OBJ_HEART]) FAIL("no implicit surface of that kind in the scene");
    memset(out, 0, 6 * sizeof(uint64_t));
    stats_walk(s->itree[kind - ORC_OBJ_HEART].root, out);
    out[5] = s->itree[kind - ORC_OBJ_HEART].lost;
    return 0;
}
int64_t orc_implicit_boxes(const orc_scene *s, int kind, double *boxes, uint64_t cap)
{
    if (!is_implicit(kind) || !s->has_itree[kind - ORC_OBJ_HEART]) FAIL("no implicit surface of that kind in the scene");
    const otree *t = &s->itree[kind - ORC_OBJ_HEART];
    for (uint64_t i = 0; i < t->ntris && i < cap; i++) { v3st(boxes + 6 * i, t->blo[i]); v3st(boxes + 6 * i + 3, t->bhi[i]); }
    return (int64_t)t->ntris;
}

/* Node::calHit (Surface.h:272-283) for one top-level object, including the normal that
 * Drawable::doAfterHit (Surface.cpp:228-231) stores for every accepted candidate. */
static int object_calhit(const orc_scene *s, int oi, ray_t r, double t0, double t1, hitrec *rec, counters_t *c)
{
    const orc_object *o = &s->objects[oi];
    double t; uint32_t prim = 0;
    int hit = 0;
    v3 localp = V(0, 0, 0);
    switch (o->type) {
    case ORC_OBJ_SPHERE: hit = sphere_hit(V(o->p[0], o->p[1], o->p[2]), o->p[3], r, t0, t1, &t); break;
    case ORC_OBJ_GROUND: case ORC_OBJ_WALL_Z: case ORC_OBJ_WALL_X:
        hit = plane_hit(o->type, o->p[0], r, t0, t1, &t); break;
    case ORC_OBJ_MESH: hit = otree_hit(&s->trees[o->mesh], r, t0, t1, &t, &prim, c); break;
    case ORC_OBJ_HEART: case ORC_OBJ_GENUS2: {
        /* HeartShape::hit (Surface.cpp:268-276), ImplicitSurface::hit (Surface.h:308-316): the
         * shared voxel tree is traversed with the ray taken into the surface's frame */
        ray_t nr;
        nr.d = vmul(r.d, 1.0 / o->p[3]);
        nr.e = vmul(vsub(r.e, V(o->p[0], o->p[1], o->p[2])), 1.0 / o->p[3]);
        hit = otree_hit(&s->itree[o->type - ORC_OBJ_HEART], nr, t0, t1, &t, &prim, c);
        if (hit) localp = o->type == ORC_OBJ_HEART ? vadd(nr.e, vmul(nr.d, t)) : V(0, 0, 0);   /* Surface.cpp:275 */
    } break;
    default: break;
    }
    if (!hit) return 0;
    /* The reference leaves localp untouched for every other drawable, so a record keeps the value
     * of an earlier, farther HeartShape hit (the record is shared by the Group's children);
     * here localp always belongs to the accepted hit: zero unless that hit is a HeartShape. */
    rec->localp = localp;
    rec->t = t; rec->mat = o->material; rec->obj = oi; rec->prim = prim;
    v3 p = vadd(r.e, vmul(r.d, t));
    switch (o->type) {
    case ORC_OBJ_HEART: case ORC_OBJ_GENUS2:                                                /* Surface.cpp:239-265, Surface.h:301-305 */
        rec->normal = implicit_normal(o->type, vmul(vsub(p, V(o->p[0], o->p[1], o->p[2])), 1.0 / o->p[3])); break;
    case ORC_OBJ_SPHERE: rec->normal = vnorm(vsub(p, V(o->p[0], o->p[1], o->p[2]))); break; /* Surface.cpp:68-71 */
    case ORC_OBJ_GROUND: rec->normal = V(0, 1, 0); break;                                   /* Surface.h:184 */
    case ORC_OBJ_WALL_Z: rec->normal = V(0, 0, 1); break;                                   /* Surface.h:197 */
    case ORC_OBJ_WALL_X: rec->normal = V(1, 0, 0); break;                                   /* Surface.h:211 */
    default: {
        const double *tr = s->trees[o->mesh].tris + 9 * (uint64_t)prim;
        v3 v0 = v3p(tr), v1 = v3p(tr + 3), v2 = v3p(tr + 6);
        rec->normal = vnorm(vcross(vsub(v1, v0), vsub(v2, v0)));
    } break;
    }
    /* Node::getNormal hands the raw normal to the material (Surface.cpp:15-23), which has just
     * been given the hit position (Surface.h:277) */
    if (s->materials[o->material].perlin == ORC_PERLIN_BLOOD)
        rec->normal = material_at(&s->materials[o->material], &s->perlin, vadd(vmul(r.d, t), r.e), rec->normal).normal;
    return 1;
}

/* Group::hit, Surface.cpp:160-181: linear closest hit, every hit shrinks t1 */
static int group_hit(const orc_scene *s, ray_t r, double t0, double t1, hitrec *rec, counters_t *c)
{
    hitrec tmp;
    hitrec *R = rec ? rec : &tmp;
    int flag = 0;
    R->localp = V(0, 0, 0);
    if (c) c->rays++;
    for (uint64_t i = 0; i < s->nobjects; i++) {
        if (object_calhit(s, (int)i, r, t0, t1, R, c)) { flag = 1; t1 = R->t; }
    }
    return flag;
}

/* ------------------------------------------------------------------ camera, camera.cpp */
void orc_camera_init(const double from[3], const double at[3], const double vup[3],
                     double vfov, double aspect, orc_camera *out)
{
    v3 lf = v3p(from), la = v3p(at), up = v3p(vup);
    double d = vlen(vsub(la, lf));
    v3 w = vnorm(vsub(lf, la));
    v3 u = vnorm(vcross(up, w));
    v3 v = vnorm(vcross(w, u));
    double height = tan(vfov * RT_PI / 180 / 2) * d * 2;
    double width = height * aspect;
    v3 llc = vsub(vsub(la, vmul(v, height / 2)), vmul(u, width / 2));
    v3st(out->e, lf); v3st(out->llc, llc); v3st(out->u, u); v3st(out->v, v);
    out->width = width; out->height = height;
    v3st(out->look_at, la);
    out->aperture = 0;
}
/* ChangeFovCamera::timeChanged, camera.cpp:88-94 */
void orc_camera_change_fov(const orc_camera *origin, double time, orc_camera *out)
{
    *out = *origin;
    double scale = sqrt(sin(time * 2 * 3.1415926) * 1.5 + 1.6) + 0.3;
    out->height = origin->height * scale;
    out->width = origin->width * scale;
    v3 llc = vsub(vsub(v3p(origin->look_at), vmul(v3p(origin->v), out->height / 2)), vmul(v3p(origin->u), out->width / 2));
    v3st(out->llc, llc);
}
/* MovableFaceLight, light.cpp:87-107: the light circles the y axis at its original distance */
void orc_light_move_face(const orc_light *origin, double time, orc_light *out)
{
    *out = *origin;
    double radius = sqrt(origin->pos[0] * origin->pos[0] + origin->pos[2] * origin->pos[2]);
    out->pos[0] = sin(time * 2 * 3.1415926) * radius;
    out->pos[2] = cos(time * 2 * 3.1415926) * radius;
}
/* DepthCamera::getRay, camera.cpp:62-72: the origin moves on the lens disk (randFromDisk,
 * algorithms.h:30-35, drawn from the sample's own key), the film point stays */
#define KEY_LENS 0x4c454e53ULL
static ray_t camera_ray_disk(const orc_camera *c, float s, float t, double x, double y)
{
    ray_t r;
    r.e = vadd(vadd(v3p(c->e), vmul(v3p(c->u), x * c->aperture)), vmul(v3p(c->v), y * c->aperture));
    v3 lookat = vadd(vadd(v3p(c->llc), vmul(vmul(v3p(c->u), s), c->width)), vmul(vmul(v3p(c->v), t), c->height));
    r.d = vnorm(vsub(lookat, r.e));
    return r;
}
static ray_t camera_ray_lens(const orc_camera *c, float s, float t, uint64_t root_key)
{
    uint64_t key = orc_hash2(root_key, KEY_LENS);
    double x, y;
    uint32_t a = 0;
    do {
        x = (draw(key, SLOT_DISK + 2 * a) - 0.5) * 2;
        y = (draw(key, SLOT_DISK + 2 * a + 1) - 0.5) * 2;
        a++;
    } while (x * x + y * y >= 1.0);
    return camera_ray_disk(c, s, t, x, y);
}
/* DepthCamera::getRay (camera.cpp:62-72) for a given point (x, y) of the unit disk */
void orc_camera_get_ray_disk(const orc_camera *c, double s, double t, double x, double y, orc_ray *out)
{
    ray_t r = camera_ray_disk(c, (float)s, (float)t, x, y);
    v3st(out->e, r.e); v3st(out->d, r.d);
}
static ray_t camera_ray(const orc_camera *c, float s, float t)
{
    ray_t r;
    r.e = v3p(c->e);
    v3 lookat = vadd(vadd(v3p(c->llc), vmul(vmul(v3p(c->u), s), c->width)), vmul(vmul(v3p(c->v), t), c->height));
    r.d = vnorm(vsub(lookat, r.e));
    return r;
}
void orc_camera_get_ray(const orc_camera *c, double s, double t, orc_ray *out)
{
    ray_t r = camera_ray(c, (float)s, (float)t);
    v3st(out->e, r.e); v3st(out->d, r.d);
}

/* ------------------------------------------------------------------ lights, light.cpp */
/* firstIntersection2Sphere, algorithms.h:15-24 */
static v3 first_intersection_sphere(ray_t ray, v3 center, double radius, double *length)
{
    v3 dir = vsub(center, ray.e);
    double dist = vlen(dir);
    double cosv = vdot(ray.d, vnorm(dir));
    double coslen = dist * cosv;
    double sinlen = dist * sqrt(1 - cosv * cosv);
    double dlen = sqrt(radius * radius - sinlen * sinlen);
    *length = coslen - dlen;
    return vadd(ray.e, vmul(ray.d, *length));
}
static v3 light_sample(const orc_light *l, v3 ppos, double ru, double rv, v3 *normal, double *p)
{
    v3 pos_ = v3p(l->pos);
    if (l->type == ORC_LIGHT_POINT) {                      /* Light::getPosition, light.cpp:37-41: the normal stays (0,0,0) */
        *p = 1;
        *normal = V(0, 0, 0);
        return pos_;
    }
    if (l->type == ORC_LIGHT_FACE) {                       /* FaceLight::getPosition, light.cpp:76-86 */
        v3 pos = pos_;
        pos = vadd(pos, vadd(vmul(vmul(v3p(l->right), ru - 0.5), l->width), vmul(vmul(v3p(l->up), rv - 0.5), l->height)));
        *p = 1.0 / l->width / l->height;
        *normal = v3p(l->normal);
        return pos;
    }
    /* SphereLight::getPosition, light.cpp:138-190 */
    double radius_ = l->radius;
    v3 dir = vsub(pos_, ppos);
    double dl = vlen(dir);
    if (fabs(dl - radius_) < RT_ZERO) { *p = RT_INFINITE * 1.1; return V(0, 0, 0); }
    v3 w = dir;
    v3 u = vmul(vnorm(vcross(fabs(w.x) > 0.1 ? V(0, 1, 0) : V(1, 0, 0), w)), dl);
    v3 v = vmul(vnorm(vcross(w, u)), dl);
    double cosmax = sqrt(1 - radius_ * radius_ / vdot(dir, dir));
    if (cosmax >= 1 - RT_ZERO) { *p = RT_INFINITE * 1.1; return V(0, 0, 0); }
    double cos_a = 1 - ru + ru * cosmax;
    double sin_a = sqrt(1 - cos_a * cos_a);
    double phi = 2 * RT_PI * rv;
    v3 lightDir = vadd(vadd(vmul(vmul(u, cos(phi)), sin_a), vmul(vmul(v, sin(phi)), sin_a)), vmul(w, cos_a));
    lightDir = vnorm(lightDir);
    double length;
    ray_t lr; lr.e = ppos; lr.d = lightDir;
    v3 hitp = first_intersection_sphere(lr, pos_, radius_, &length);
    v3 nl = vnorm(vsub(hitp, pos_));
    *normal = nl;
    *p = vdot(nl, vneg(lightDir)) / (2 * RT_PI * (1 - cosmax)) / length / length;
    return hitp;
}
void orc_light_sample(const orc_light *l, const double x[3], double u, double v,
                      double pos[3], double nl[3], double *pdf)
{
    v3 n = V(0, 0, 0);
    double p = 0;
    v3 r = light_sample(l, v3p(x), u, v, &n, &p);
    v3st(pos, r); v3st(nl, n); *pdf = p;
}

/* refract, scene.cpp:384-395: returns -1 on total internal reflection */
static int refract_dir(v3 d, v3 normal, double n, v3 *t)
{
    double dn = vdot(d, normal);
    double inSqrt = 1 - (1 - dn * dn) / n / n;
    if (inSqrt < 0) { *t = vreflact(d, normal); return -1; }
    *t = vsub(vmul(vsub(d, vmul(normal, dn)), 1 / n), vmul(normal, sqrt(inSqrt)));
    *t = vnorm(*t);
    return 0;
}
int orc_refract(const double d[3], const double n[3], double eta, double out[3])
{
    v3 t;
    int r = refract_dir(v3p(d), v3p(n), eta, &t);
    v3st(out, t);
    return r;
}

/* ------------------------------------------------------------------ integrator */
typedef struct {
    const orc_scene *s;
    int max_jump, rus_jump, face_forward;
    const double *light_uv;  /* nlights x 2: this sample's entry of every light's table */
    counters_t *c;
} pctx;

/* Scene::rayColor, scene.cpp:194-381 */
static v3 ray_color(const pctx *cx, ray_t ray, double t0, double t1, int jumpTime, int E, uint64_t key)
{
    const orc_scene *s = cx->s;
    if (++jumpTime > cx->max_jump) return s->background;
    hitrec rec;
    rec.ray = ray;
    if (!group_hit(s, ray, t0, t1, &rec, cx->c)) return s->background;
    const orc_material *mat = &s->materials[rec.mat];
    v3 p = vadd(ray.e, vmul(ray.d, rec.t));
    v3 color = V(0, 0, 0);
    v3 emission = v3p(mat->emission), mcolor = v3p(mat->color);
    double Ed = (double)E;
    if (!(mat->type & ORC_MAT_MONTECARLO)) {
        /* Phong / Whitted branch, scene.cpp:305-375 */
        if (mat->type & ORC_MAT_NORMAL) {                                /* :307-322 */
            surf_t sf = { v3p(mat->ambient), v3p(mat->color), v3p(mat->specular), rec.normal };
            if (mat->perlin != ORC_PERLIN_NONE) sf = material_at(mat, &s->perlin, vadd(vmul(ray.d, rec.t), ray.e), rec.normal);
            for (uint64_t li = 0; li < s->nlights; li++) {
                const orc_light *L = &s->lights[li];
                v3 amb = sf.ambient;
                if (mat->check == ORC_CHECK_NORMAL) amb = V(fabs(rec.normal.x), fabs(rec.normal.y), fabs(rec.normal.z));           /* material.h:78 */
                else if (mat->check == ORC_CHECK_POSITION)                                                                        /* material.h:88-89 */
                    amb = vdivv(vsub(rec.localp, v3p(mat->check_box)), vsub(v3p(mat->check_box + 3), v3p(mat->check_box)));
                color = vadd(color, V(L->ambient[0] * amb.x, L->ambient[1] * amb.y, L->ambient[2] * amb.z));
                /* light->getPosition() without a hit record: the point itself, a jittered point of
                 * a FaceLight; a SphereLight would dereference the null record (light.cpp:143) and
                 * is taken at its centre here */
                v3 lightpos = v3p(L->pos);
                if (L->type == ORC_LIGHT_FACE) { v3 nn; double pp; lightpos = light_sample(L, p, cx->light_uv[2 * li], cx->light_uv[2 * li + 1], &nn, &pp); }
                ray_t nr;
                nr.e = p;
                nr.d = vnorm(vsub(lightpos, p));
                if (!group_hit(s, nr, RT_ZERO, vlen(vsub(lightpos, p)) - RT_ZERO, NULL, cx->c)) {
                    double kd = vdot(rec.normal, nr.d);
                    if (!(0.0 < kd)) kd = 0.0;                           /* max(0.0, n.d) */
                    color = vadd(color, V(sf.diffuse.x * (L->diffuse[0] * kd), sf.diffuse.y * (L->diffuse[1] * kd), sf.diffuse.z * (L->diffuse[2] * kd)));
                    double ks = pow(vdot(vnorm(vsub(nr.d, ray.d)), rec.normal), mat->shine);
                    color = vadd(color, V(sf.specular.x * (L->specular[0] * ks), sf.specular.y * (L->specular[1] * ks), sf.specular.z * (L->specular[2] * ks)));
                }
            }
        }
        if (mat->type & ORC_MAT_SPECULAR) {                              /* :324-341 */
            ray_t mr;
            mr.e = p;
            mr.d = vsub(ray.d, vmul(vmul(rec.normal, vdot(rec.normal, ray.d)), 2));
            if (mat->blur >= RT_ZERO) {
                v3 vec = V(0, 0, 1);
                if (viszero(vcross(vec, mr.d))) vec = V(0, 1, 0);
                v3 u = vnorm(vcross(vec, mr.d));
                v3 v = vnorm(vcross(mr.d, u));
                double x, y;
                uint32_t a = 0;
                do {
                    x = (draw(key, SLOT_DISK + 2 * a) - 0.5) * 2;
                    y = (draw(key, SLOT_DISK + 2 * a + 1) - 0.5) * 2;
                    a++;
                } while (x * x + y * y >= 1.0);
                x *= mat->blur; y *= mat->blur;
                mr.d = vadd(vadd(mr.d, vmul(u, x)), vmul(v, y));
            }
            v3 in = ray_color(cx, mr, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_SPECULAR));
            color = vadd(color, V(mat->mirror[0] * in.x, mat->mirror[1] * in.y, mat->mirror[2] * in.z));
        }
        if (mat->type & ORC_MAT_TRANSPARENT) {                           /* :343-373, as :267-303 without emission */
            ray_t rr;
            rr.e = p;
            v3 k = V(1, 1, 1);
            double cc = 0;
            v3 r = vreflact(ray.d, rec.normal);
            int flag = 0;
            if (vdot(rec.normal, ray.d) > 0) {
                k.x = exp(-mat->atten[0] * rec.t);
                k.y = exp(-mat->atten[1] * rec.t);
                k.z = exp(-mat->atten[2] * rec.t);
                if (refract_dir(ray.d, vneg(rec.normal), 1 / mat->nr, &rr.d)) {
                    flag = 1;
                    ray_t q; q.e = p; q.d = r;
                    v3 in = ray_color(cx, q, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFLECT));
                    color = vadd(color, V(k.x * in.x, k.y * in.y, k.z * in.z));
                } else {
                    cc = vdot(rr.d, rec.normal);
                }
            } else {
                cc = vdot(vneg(ray.d), rec.normal);
                refract_dir(ray.d, rec.normal, mat->nr, &rr.d);
            }
            if (!flag) {
                double n = mat->nr;
                double R0 = (n - 1) * (n - 1) / (n + 1) / (n + 1);
                double R = R0 + (1 - R0) * pow(1 - cc, 5);
                ray_t q; q.e = p; q.d = r;
                v3 a = ray_color(cx, q, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFLECT));
                v3 b = ray_color(cx, rr, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFRACT));
                v3 sum = vadd(vmul(a, R), vmul(b, 1 - R));
                color = vadd(color, V(k.x * sum.x, k.y * sum.y, k.z * sum.z));
            }
        }
        return color;
    }
    v3 c = mcolor;
    if (jumpTime > cx->rus_jump) {                                       /* :208-212 */
        double pr = (c.x > c.y && c.x > c.z) ? c.x : (c.y > c.z ? c.y : c.z);
        if (draw(key, SLOT_RR) < pr) c = vmul(c, 1 / pr);
        else return vmul(emission, Ed);
    }
    if (mat->type & ORC_MAT_NORMAL) {                                    /* :213-248 */
        double r1 = RT_PI * 2 * draw(key, SLOT_DIFF1);
        double r2 = draw(key, SLOT_DIFF2), r2s = sqrt(r2);
        v3 w = rec.normal;
        if (cx->face_forward && vdot(w, ray.d) > 0) w = vneg(w);
        v3 u = vnorm(vcross(fabs(w.x) > 0.1 ? V(0, 1, 0) : V(1, 0, 0), w));
        v3 v = vcross(w, u);
        v3 d = vnorm(vadd(vadd(vmul(vmul(u, cos(r1)), r2s), vmul(vmul(v, sin(r1)), r2s)), vmul(w, sqrt(1 - r2))));
        for (uint64_t li = 0; li < s->nlights; li++) {
            const orc_light *L = &s->lights[li];
            double pr; v3 normal;
            v3 ppos = vadd(vmul(rec.ray.d, rec.t), rec.ray.e);          /* light.cpp:143 */
            v3 lightpos = light_sample(L, ppos, cx->light_uv[2 * li], cx->light_uv[2 * li + 1], &normal, &pr);
            if (pr > RT_INFINITE) continue;
            ray_t nr;
            double dist = vlen(vsub(lightpos, p));
            nr.e = p;
            nr.d = vnorm(vsub(lightpos, p));
            double dot = vdot(normal, vneg(nr.d));
            double dot2 = vdot(vnorm(w), nr.d);
            if (dot < 0 || dot2 < 0) continue;
            if (!group_hit(s, nr, RT_ZERO, dist - RT_ZERO, NULL, cx->c)) {
                v3 ld = vmul(vmul(vmul(v3p(L->diffuse), dot), dot2), 1 / pr / dist / dist);
                color = vadd(color, V(c.x * ld.x, c.y * ld.y, c.z * ld.z));
            }
        }
        ray_t nr2; nr2.e = p; nr2.d = d;
        v3 in = ray_color(cx, nr2, RT_ZERO, RT_INFINITE, jumpTime + 1, 0, orc_hash2(key, BR_DIFFUSE));
        color = vadd(vadd(color, vmul(emission, Ed)), V(mcolor.x * in.x, mcolor.y * in.y, mcolor.z * in.z));
    }
    if (mat->type & ORC_MAT_SPECULAR) {                                  /* :249-266 */
        ray_t mr;
        mr.e = p;
        mr.d = vsub(ray.d, vmul(vmul(rec.normal, vdot(rec.normal, ray.d)), 2));
        if (mat->blur >= RT_ZERO) {
            v3 vec = V(0, 0, 1);
            if (viszero(vcross(vec, mr.d))) vec = V(0, 1, 0);
            v3 u = vnorm(vcross(vec, mr.d));
            v3 v = vnorm(vcross(mr.d, u));
            double x, y;
            uint32_t a = 0;
            do {                                                         /* randFromDisk, algorithms.h:30-35 */
                x = (draw(key, SLOT_DISK + 2 * a) - 0.5) * 2;
                y = (draw(key, SLOT_DISK + 2 * a + 1) - 0.5) * 2;
                a++;
            } while (x * x + y * y >= 1.0);
            x *= mat->blur; y *= mat->blur;
            mr.d = vadd(vadd(mr.d, vmul(u, x)), vmul(v, y));
        }
        v3 in = ray_color(cx, mr, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_SPECULAR));
        color = vadd(color, vadd(vmul(emission, Ed), V(mcolor.x * in.x, mcolor.y * in.y, mcolor.z * in.z)));
    }
    if (mat->type & ORC_MAT_TRANSPARENT) {                               /* :267-303 */
        ray_t rr;
        rr.e = p;
        v3 k = V(1, 1, 1);
        double cc = 0;
        v3 r = vreflact(ray.d, rec.normal);
        int flag = 0;
        if (vdot(rec.normal, ray.d) > 0) {
            k.x = exp(-mat->atten[0] * rec.t);
            k.y = exp(-mat->atten[1] * rec.t);
            k.z = exp(-mat->atten[2] * rec.t);
            if (refract_dir(ray.d, vneg(rec.normal), 1 / mat->nr, &rr.d)) {
                flag = 1;
                ray_t q; q.e = p; q.d = r;
                v3 in = ray_color(cx, q, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFLECT));
                color = vadd(color, vadd(vmul(emission, Ed), V(k.x * in.x, k.y * in.y, k.z * in.z)));
            } else {
                cc = vdot(rr.d, rec.normal);
            }
        } else {
            cc = vdot(vneg(ray.d), rec.normal);
            k.x = k.y = k.z = 1.0;
            refract_dir(ray.d, rec.normal, mat->nr, &rr.d);
        }
        if (!flag) {
            double n = mat->nr;
            double R0 = (n - 1) * (n - 1) / (n + 1) / (n + 1);
            double R = R0 + (1 - R0) * pow(1 - cc, 5);
            ray_t q; q.e = p; q.d = r;
            v3 a = ray_color(cx, q, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFLECT));
            v3 b = ray_color(cx, rr, RT_ZERO, RT_INFINITE, jumpTime, 1, orc_hash2(key, BR_REFRACT));
            v3 sum = vadd(vmul(a, R), vmul(b, 1 - R));                   /* k multiplies both terms (:301) */
            color = vadd(color, V(k.x * sum.x, k.y * sum.y, k.z * sum.z));
        }
    }
    return color;
}

static int check_params(const orc_scene *s, const orc_render_params *p)
{
    if (!s || !p) FAIL("null argument");
    if (p->n <= 0 || p->m <= 0) FAIL("bad jitter table size");
    if (p->sample_begin < 0 || p->sample_end > p->n * p->m || p->sample_begin > p->sample_end) FAIL("bad sample range");
    if (p->max_jump < 0 || p->max_jump > 60) FAIL("bad max_jump");
    return 0;
}

int orc_ray_color(const orc_scene *s, const orc_ray *rays, uint64_t n, double t0, double t1,
                  int jumptime, const orc_render_params *p, double *out, uint64_t counters[4])
{
    if (check_params(s, p)) return -1;
    counters_t c = { 0, 0, 0, 0 };
    double *uv = (double *)malloc((s->nlights ? s->nlights : 1) * 2 * sizeof(double));
    for (uint64_t i = 0; i < n; i++) {
        for (uint64_t li = 0; li < s->nlights; li++) orc_jitter_pair(p->seed, i, 1 + (uint32_t)li, 0, 1, 1, uv + 2 * li);
        pctx cx = { s, p->max_jump, p->rus_jump, p->face_forward, uv, &c };
        v3 col = ray_color(&cx, rayp(&rays[i]), t0, t1, jumptime, 1, orc_path_root_key(p->seed, i, 0));
        v3st(out + 3 * i, col);
    }
    free(uv);
    if (counters) { counters[0] += c.rays; counters[1] += c.inner; counters[2] += c.leaf; counters[3] += c.prim; }
    return 0;
}

/* Scene::run(x,y), scene.cpp:97-114, for the caller's slice of the sample index range */
static void render_pixel(const orc_scene *s, const orc_camera *cam, const orc_render_params *p,
                         uint64_t pix, double *uv, double *out, counters_t *c)
{
    int x = (int)(pix % (uint64_t)p->width), y = (int)(pix / (uint64_t)p->width);
    v3 color = V(0, 0, 0);
    for (int k = p->sample_begin; k < p->sample_end; k++) {
        double pr[2];
        orc_jitter_pair(p->seed, pix, 0, (uint32_t)k, (uint32_t)p->n, (uint32_t)p->m, pr);
        for (uint64_t li = 0; li < s->nlights; li++)
            orc_jitter_pair(p->seed, pix, 1 + (uint32_t)li, (uint32_t)k, (uint32_t)p->n, (uint32_t)p->m, uv + 2 * li);
        const uint64_t root = orc_path_root_key(p->seed, pix, (uint32_t)k);
        const float fs = (float)((pr[0] + x * 1.0) / p->width), ft = (float)((pr[1] + y * 1.0) / p->height);
        ray_t ray = cam->aperture != 0 ? camera_ray_lens(cam, fs, ft, root) : camera_ray(cam, fs, ft);
        pctx cx = { s, p->max_jump, p->rus_jump, p->face_forward, uv, c };
        color = vadd(color, ray_color(&cx, ray, 0, RT_INFINITE, 0, 1, root));
    }
    out[0] += color.x; out[1] += color.y; out[2] += color.z;
}

int orc_render_range(const orc_scene *s, const orc_camera *cam, const orc_render_params *p,
                     uint64_t pix_begin, uint64_t pix_end, double *radiance_sum,
                     uint64_t counters[4], int threads)
{
    if (check_params(s, p)) return -1;
    if (p->width <= 0 || p->height <= 0) FAIL("bad image size");
    uint64_t tot = (uint64_t)p->width * (uint64_t)p->height;
    if (pix_end > tot || pix_begin > pix_end) FAIL("bad pixel range");
    if (threads < 1) threads = 1;
    uint64_t cr = 0, ci = 0, cl = 0, cp = 0;
    /* runParallelly (scene.cpp:141-172): a dynamic queue of single pixels */
#pragma omp parallel num_threads(threads) reduction(+ : cr, ci, cl, cp)
    {
        double *uv = (double *)malloc((s->nlights ? s->nlights : 1) * 2 * sizeof(double));
        counters_t c = { 0, 0, 0, 0 };
#pragma omp for schedule(dynamic, 16)
        for (uint64_t pix = pix_begin; pix < pix_end; pix++)
            render_pixel(s, cam, p, pix, uv, radiance_sum + 3 * pix, &c);
        cr += c.rays; ci += c.inner; cl += c.leaf; cp += c.prim;
        free(uv);
    }
    if (counters) { counters[0] += cr; counters[1] += ci; counters[2] += cl; counters[3] += cp; }
    return 0;
}
int orc_render(const orc_scene *s, const orc_camera *cam, const orc_render_params *p,
               double *radiance_sum, uint64_t counters[4], int threads)
{
    if (!p) FAIL("null argument");
    return orc_render_range(s, cam, p, 0, (uint64_t)p->width * (uint64_t)p->height, radiance_sum, counters, threads);
}

/* SigmoidMethod / SqrtMethod::reproduction, tonereproductionmethod.h:11-35, per channel */
double orc_tone(int tone, double x)
{
    if (tone == ORC_TONE_SIGMOID) return 1.0 / (1 + exp(-(2 * x - 1))) * 2.164 - 0.582;
    if (tone == ORC_TONE_SQRT) return x > 1 ? pow(x, 1.0 / 2) : x;
    return x;
}
void orc_resolve_bgr8(const double *sum, int w, int h, int spp, uint8_t *bgr) { orc_resolve_tone_bgr8(sum, w, h, spp, ORC_TONE_NONE, bgr); }
void orc_resolve_tone_bgr8(const double *sum, int w, int h, int spp, int tone, uint8_t *bgr)
{
    double inv = 1.0 / spp;                       /* scene.cpp:112: color * (1.0 / (n * n)) */
    for (uint64_t i = 0; i < (uint64_t)w * (uint64_t)h; i++) {
        double c[3];
        for (int k = 0; k < 3; k++) {
            double v = orc_tone(tone, sum[3 * i + k] * inv);
            if (v > 1.0) v = 1.0; else if (v < 0) v = 0;   /* regularize, types.h:167-177 */
            c[k] = v;
        }
        bgr[3 * i + 2] = (uint8_t)(int)(c[0] * 255);       /* scene.cpp:183-185 */
        bgr[3 * i + 1] = (uint8_t)(int)(c[1] * 255);
        bgr[3 * i + 0] = (uint8_t)(int)(c[2] * 255);
    }
}

int orc_trace_rays(const orc_scene *s, const orc_ray *rays, uint64_t n, double t0, double t1,
                   uint32_t *prim_id, double *t, double *normal, uint64_t counters[4])
{
    counters_t c = { 0, 0, 0, 0 };
    for (uint64_t i = 0; i < n; i++) {
        hitrec rec;
        if (group_hit(s, rayp(&rays[i]), t0, t1, &rec, &c)) {
            prim_id[i] = s->prim_base[rec.obj] + rec.prim;
            if (t) t[i] = rec.t;
            if (normal) v3st(normal + 3 * i, rec.normal);
        } else {
            prim_id[i] = ORC_MISS;
            if (t) t[i] = RT_INFINITE;
            if (normal) v3st(normal + 3 * i, V(0, 0, 0));
        }
    }
    if (counters) { counters[0] += c.rays; counters[1] += c.inner; counters[2] += c.leaf; counters[3] += c.prim; }
    return 0;
}

int orc_trace_primary(const orc_scene *s, const orc_camera *cam, int w, int h,
                      uint32_t *prim_id, double *t, uint64_t counters[4], int threads)
{
    if (w <= 0 || h <= 0) FAIL("bad image size");
    if (threads < 1) threads = 1;
    uint64_t cr = 0, ci = 0, cl = 0, cp = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : cr, ci, cl, cp)
    for (int64_t pix = 0; pix < (int64_t)w * h; pix++) {
        int x = (int)(pix % w), y = (int)(pix / w);
        counters_t c = { 0, 0, 0, 0 };
        ray_t ray = camera_ray(cam, (float)((0.5 + x * 1.0) / w), (float)((0.5 + y * 1.0) / h));
        hitrec rec;
        if (group_hit(s, ray, 0, RT_INFINITE, &rec, &c)) {
            prim_id[pix] = s->prim_base[rec.obj] + rec.prim;
            if (t) t[pix] = rec.t;
        } else {
            prim_id[pix] = ORC_MISS;
            if (t) t[pix] = RT_INFINITE;
        }
        cr += c.rays; ci += c.inner; cl += c.leaf; cp += c.prim;
    }
    if (counters) { counters[0] += cr; counters[1] += ci; counters[2] += cl; counters[3] += cp; }
    return 0;
}
