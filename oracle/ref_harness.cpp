// ref_harness.cpp -- C-ABI window onto the UNMODIFIED reference objects (test infrastructure).
//
// Linked with the reference's own translation units (compiled where they lie, see Makefile)
// into oracle/_ref/libref_harness.so.  It is used to pin the CPU oracle (rt_oracle.c) and to
// generate the golden vectors under tests/golden/.  It never ships to, or runs in, the
// product path.
//
// The reference has no triangle primitive, so a harness-side AbstructNode subclass
// (TriangleNode below, our code) is inserted into the reference's own Octree, exactly the way
// the reference inserts BoxNode voxels (octree.cpp:250-252); the tree build
// (octree.cpp:61-108) and the ordered traversal (octree.cpp:285-376) that run are the
// reference's.
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <assert.h>
#include <vector>
#include <list>
#include <set>
#include <map>
#include <array>
#include <queue>
#include <memory>
#include <utility>
#include <functional>
#include <algorithm>
#include <iostream>

// open the reference classes up for inspection (tree shape, jitter tables, camera frame)
#define private public
#define protected public
#include "types.h"
#include "definitions.h"
#include "RTObject.h"
#include "algorithms.h"
#include "octree.h"
#include "Surface.h"
#include "material.h"
#include "light.h"
#include "camera.h"
#include "tonereproductionmethod.h"
#undef private
#undef protected

namespace {

Vec3 v3(const double *p) { return Vec3(p[0], p[1], p[2]); }
void st(double *p, const Vec3 &v) { p[0] = v.x_; p[1] = v.y_; p[2] = v.z_; }
Ray mkray(const double *r) { Ray x; x.e = v3(r); x.d = v3(r + 3); return x; }

// Triangle payload for the reference Octree.  Same arithmetic as the oracle's
// triangle_hit(); the primitive index travels in HitRecord::localp.x_.
class TriangleNode : public AbstructNode {
public:
    TriangleNode(const double *t, double pad, uint32_t id) : id_(id)
    {
        v0_ = v3(t); Vec3 v1 = v3(t + 3), v2 = v3(t + 6);
        e1_ = v1 - v0_; e2_ = v2 - v0_;
        box_[0] = Vec3(std::min(std::min(v0_.x_, v1.x_), v2.x_) - pad, std::min(std::min(v0_.y_, v1.y_), v2.y_) - pad, std::min(std::min(v0_.z_, v1.z_), v2.z_) - pad);
        box_[1] = Vec3(std::max(std::max(v0_.x_, v1.x_), v2.x_) + pad, std::max(std::max(v0_.y_, v1.y_), v2.y_) + pad, std::max(std::max(v0_.z_, v1.z_), v2.z_) + pad);
        dirtyBound_ = false;
    }
    bool calHit(const Ray &ray, double t0, double t1, HitRecord *rec) override
    {
        ++tests;
        Vec3 p = ray.d ^ e2_;
        double det = e1_ * p;
        double a = fabs(det);
        if (!(a >= 1e-200)) return false;
        if (!(det * det >= 1e-10 * (e1_ * e1_) * (e2_ * e2_) * (ray.d * ray.d))) return false;      // a determinant near its rounding error is no hit
        Vec3 s = ray.e - v0_;
        double up = s * p;
        double us = det < 0 ? -up : up;
        if (!(us >= 0.0 && us <= a)) return false;
        Vec3 q = s ^ e1_;
        double vp = ray.d * q;
        double vs = det < 0 ? -vp : vp;
        if (!(vs >= 0.0 && us + vs <= a)) return false;
        double t = (e2_ * q) / det;
        if (!(t >= t0 && t <= t1)) return false;
        rec->t = t;
        rec->localp = Vec3((double)id_, 0, 0);
        return true;
    }
    static uint64_t tests;
private:
    Vec3 v0_, e1_, e2_;
    uint32_t id_;
};
uint64_t TriangleNode::tests = 0;

struct RefTree {
    Octree *tree;
    std::vector<std::shared_ptr<TriangleNode>> prims;
};

void walk(const OctreeNode *n, int depth, uint64_t out[6])
{
    out[0]++;
    out[2] += n->objects_.size();
    if ((uint64_t)depth > out[4]) out[4] = depth;
    if (n->isLeaf()) { out[1]++; return; }
    out[3] += n->objects_.size();
    for (int i = 0; i < 8; i++) walk(n->children_[i].get(), depth + 1, out);
}

} // namespace

extern "C" {

// algorithm::hitBox, algorithms.h:223-236
int ref_hit_box(const double lo[3], const double hi[3], const double ray[6], double t0, double t1, double *r0, double *r1)
{
    BoundBox_t b; b[0] = v3(lo); b[1] = v3(hi);
    double a = 0, c = 0;
    bool h = algorithm::hitBox(b, mkray(ray), t0, t1, &a, &c);
    *r0 = a; *r1 = c;
    return h ? 1 : 0;
}

// Sphere through Node::calHit (Surface.h:272-283): hit, t and the stored normal
int ref_sphere_hit(const double c[3], double radius, const double ray[6], double t0, double t1, double *t, double normal[3])
{
    Sphere s(v3(c), radius);
    HitRecord rec;
    Ray r = mkray(ray);
    rec.ray = r;
    if (!s.calHit(r, t0, t1, &rec)) return 0;
    *t = rec.t; st(normal, rec.normal);
    return 1;
}

// type: 2 Ground (y), 3 Wall_z, 4 Wall_x
int ref_plane_hit(int type, double coord, const double ray[6], double t0, double t1, double *t, double normal[3])
{
    HitRecord rec;
    Ray r = mkray(ray);
    bool h = false;
    if (type == 2) { Ground g(coord); h = g.calHit(r, t0, t1, &rec); }
    else if (type == 3) { Wall_z g(coord); h = g.calHit(r, t0, t1, &rec); }
    else { Wall_x g(coord); h = g.calHit(r, t0, t1, &rec); }
    if (!h) return 0;
    *t = rec.t; st(normal, rec.normal);
    return 1;
}

// OctreeNode::getChildBound, octree.h:82-120
void ref_child_bound(const double lo[3], const double hi[3], int child, double clo[3], double chi[3])
{
    Vec3 b[2] = { v3(lo), v3(hi) };
    OctreeNode n(b, 128);
    Vec3 c[2];
    n.getChildBound(child, c);
    st(clo, c[0]); st(chi, c[1]);
}

// isBoundBoxOverLap, algorithms.h:329-333
int ref_box_overlap(const double alo[3], const double ahi[3], const double blo[3], const double bhi[3])
{
    BoundBox_t a, b; a[0] = v3(alo); a[1] = v3(ahi); b[0] = v3(blo); b[1] = v3(bhi);
    return algorithm::isBoundBoxOverLap(a, b) ? 1 : 0;
}

// Camera::init + members, camera.cpp:37-48.  out = e, llc, u, v, width, height (14 doubles)
void ref_camera_init(const double from[3], const double at[3], const double vup[3], double vfov, double aspect, double out[14])
{
    Camera c(v3(from), v3(at), v3(vup), vfov, aspect);
    st(out, c.e_); st(out + 3, c.leftLowerCorner); st(out + 6, c.u_); st(out + 9, c.v_);
    out[12] = c.width; out[13] = c.height;
}
// Camera::getRay(float,float), camera.cpp:24-31
void ref_camera_get_ray(const double from[3], const double at[3], const double vup[3], double vfov, double aspect,
                        double s, double t, double ray[6])
{
    Camera c(v3(from), v3(at), v3(vup), vfov, aspect);
    Ray r = c.getRay(s, t);
    st(ray, r.e); st(ray + 3, r.d);
}

// SphereLight::getPosition (light.cpp:138-190) for shading point x with table entry (u,v)
void ref_sphere_light_sample(const double pos[3], double radius, const double x[3], double u, double v,
                             double out_pos[3], double out_n[3], double *pdf)
{
    auto l = std::make_shared<SphereLight>(v3(pos), radius, Color(1, 1, 1, 1));
    l->randomPairs_.assign(1, std::make_pair(u, v));
    l->itIndex_ = 0;
    HitRecord rec;
    rec.t = 0;
    rec.ray.e = v3(x);
    rec.ray.d = Vec3(0, 0, 0);
    Vec3 n; double p = 0;
    Vec3 r = l->getPosition(&rec, &n, &p);
    st(out_pos, r); st(out_n, n); *pdf = p;
}
// FaceLight::getPosition (light.cpp:50-56, 76-86); also returns right_ and normal_ as built
void ref_face_light_sample(const double pos[3], const double normal[3], const double up[3], double w, double h,
                           double u, double v, double out_pos[3], double out_n[3], double *pdf,
                           double out_right[3])
{
    auto l = std::make_shared<FaceLight>(v3(pos), v3(normal), v3(up), w, h, Color(1, 1, 1, 1), Color(1, 1, 1, 1), Color(1, 1, 1, 1));
    l->randomPairs_.assign(1, std::make_pair(u, v));
    l->itIndex_ = 0;
    Vec3 n; double p = 0;
    Vec3 r = l->getPosition(nullptr, &n, &p);
    st(out_pos, r); st(out_n, n); *pdf = p; st(out_right, l->right_);
}

// Vec3::reflact (types.h:37-40) and normalize (types.h:19-23)
void ref_reflact(const double d[3], const double n[3], double out[3]) { st(out, v3(d).reflact(v3(n))); }
void ref_normalize(const double d[3], double out[3]) { st(out, v3(d).normalize()); }
void ref_cross(const double a[3], const double b[3], double out[3]) { st(out, v3(a) ^ v3(b)); }
// algorithm::firstIntersection2Sphere, algorithms.h:15-24
void ref_first_intersection_sphere(const double ray[6], const double c[3], double radius, double out[3], double *length)
{
    double len = 0;
    Vec3 r = algorithm::firstIntersection2Sphere(mkray(ray), v3(c), radius, len);
    st(out, r); *length = len;
}

// reference Octree over harness triangles: build, shape, closest hit
void *ref_tree_create(const double *tris, uint64_t n, double pad, int max_size)
{
    RefTree *t = new RefTree;
    Vec3 box[2] = { Vec3(INFINITE, INFINITE, INFINITE), Vec3(-INFINITE, -INFINITE, -INFINITE) };
    t->prims.reserve(n);
    for (uint64_t i = 0; i < n; i++) {
        auto p = std::make_shared<TriangleNode>(tris + 9 * i, pad, (uint32_t)i);
        BoundBox_t nb; p->getBoundBox(nb);
        box[0].x_ = std::min(box[0].x_, nb[0].x_); box[0].y_ = std::min(box[0].y_, nb[0].y_); box[0].z_ = std::min(box[0].z_, nb[0].z_);
        box[1].x_ = std::max(box[1].x_, nb[1].x_); box[1].y_ = std::max(box[1].y_, nb[1].y_); box[1].z_ = std::max(box[1].z_, nb[1].z_);
        t->prims.push_back(p);
    }
    t->tree = new Octree(max_size, box);
    t->tree->build();
    for (auto &p : t->prims) t->tree->insertObject(p);
    return t;
}
void ref_tree_destroy(void *h) { RefTree *t = (RefTree *)h; delete t->tree; delete t; }
void ref_tree_stats(void *h, uint64_t out[6])
{
    RefTree *t = (RefTree *)h;
    memset(out, 0, 6 * sizeof(uint64_t));
    walk(t->tree->root_.get(), 0, out);
}
// Octree::hit for n rays; prim = 0xFFFFFFFF on a miss.  tests (nullable) += primitive tests.
void ref_tree_hit(void *h, const double *rays, uint64_t n, double t0, double t1, uint32_t *prim, double *tout, uint64_t *tests)
{
    RefTree *t = (RefTree *)h;
    uint64_t before = TriangleNode::tests;
    for (uint64_t i = 0; i < n; i++) {
        HitRecord rec;
        if (t->tree->hit(mkray(rays + 6 * i), t0, t1, &rec)) { prim[i] = (uint32_t)rec.localp.x_; tout[i] = rec.t; }
        else { prim[i] = 0xFFFFFFFFu; tout[i] = INFINITE; }
    }
    if (tests) *tests += TriangleNode::tests - before;
}

// Group::hit (Surface.cpp:160-181) over spheres and planes.  objs: n x 6 doubles
// {type, p0, p1, p2, p3, unused}; returns index of the closest child or -1, its t and normal.
void ref_group_hit(const double *objs, int nobj, const double *rays, uint64_t n, double t0, double t1,
                   int32_t *which, double *tout, double *normal)
{
    auto g = std::make_shared<Group>();
    std::vector<std::shared_ptr<Material>> mats;
    for (int i = 0; i < nobj; i++) {
        const double *o = objs + 6 * i;
        std::shared_ptr<Node> nd;
        switch ((int)o[0]) {
        case 1: nd = std::make_shared<Sphere>(Vec3(o[1], o[2], o[3]), o[4]); break;
        case 2: nd = std::make_shared<Ground>(o[1]); break;
        case 3: nd = std::make_shared<Wall_z>(o[1]); break;
        default: nd = std::make_shared<Wall_x>(o[1]); break;
        }
        // the object index travels in the material's shine field
        auto m = std::make_shared<Material>(Color(0, 0, 0, 0), Color(0, 0, 0, 0), Color(0, 0, 0, 0), (double)i);
        nd->setMaterial(m);
        mats.push_back(m);
        g->addChild(nd);
    }
    for (uint64_t i = 0; i < n; i++) {
        HitRecord rec;
        Ray r = mkray(rays + 6 * i);
        rec.ray = r;
        if (g->calHit(r, t0, t1, &rec)) {
            which[i] = (int32_t)rec.mat->getShine(); tout[i] = rec.t; st(normal + 3 * i, rec.normal);
        } else {
            which[i] = -1; tout[i] = INFINITE; normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0;
        }
    }
}

// ---- implicit surfaces: the reference's own voxel octree (ImplicitSurfaceOctree::build,
// octree.cpp:148-254) over its own implicit functions; kind 6 = HeartShape, 7 = SurfaceOfGenus2
struct RefImplicit {
    ImplicitSurfaceOctree *tree;
    SurfaceOfGenus2Function g2;
};
void *ref_implicit_create(int kind, double min_size)
{
    RefImplicit *r = new RefImplicit;
    if (kind == 6) {
        Point_t seedp[2]{ Vec3(0, 0, 0), Vec3(2, 0, 0) };                      // Surface.cpp:302
        r->tree = new ImplicitSurfaceOctree(HeartShape::heartImplicitFunc, seedp, min_size);
    } else {
        Point_t seedp[2]{ r->g2.getSeedPoint(0), r->g2.getSeedPoint(1) };       // Surface.h:321
        r->tree = new ImplicitSurfaceOctree(std::bind(&SurfaceOfGenus2Function::ImplicitFunc, &r->g2, std::placeholders::_1), seedp, min_size);
    }
    if (r->tree->build()) { delete r->tree; delete r; return nullptr; }
    return r;
}
void ref_implicit_destroy(void *h) { RefImplicit *r = (RefImplicit *)h; delete r->tree; delete r; }
void ref_implicit_stats(void *h, uint64_t out[6])
{
    memset(out, 0, 6 * sizeof(uint64_t));
    walk(((RefImplicit *)h)->tree->root_.get(), 0, out);
}
namespace {
struct BoxLess { bool operator()(const BoxNode *a, const BoxNode *b) const { return *a < *b; } };
void collect(const OctreeNode *n, std::set<const BoxNode *, BoxLess> &all)
{
    for (const auto &o : n->objects_) all.insert(static_cast<const BoxNode *>(o.get()));
    if (n->isLeaf()) return;
    for (int i = 0; i < 8; i++) collect(n->children_[i].get(), all);
}
}
// the voxels in the order of the reference's std::set (BoxNode::operator<): 6 doubles each
int64_t ref_implicit_boxes(void *h, double *boxes, uint64_t cap)
{
    std::set<const BoxNode *, BoxLess> all;
    collect(((RefImplicit *)h)->tree->root_.get(), all);
    uint64_t i = 0;
    for (const BoxNode *b : all) {
        if (i < cap) { st(boxes + 6 * i, b->box_[0]); st(boxes + 6 * i + 3, b->box_[1]); }
        i++;
    }
    return (int64_t)i;
}
// Octree::hit on the voxel tree (BoxNode::calHit payload, Surface.cpp:43-59)
void ref_implicit_hit(void *h, const double *rays, uint64_t n, double t0, double t1, int32_t *hit, double *tout)
{
    RefImplicit *r = (RefImplicit *)h;
    for (uint64_t i = 0; i < n; i++) {
        HitRecord rec;
        hit[i] = r->tree->hit(mkray(rays + 6 * i), t0, t1, &rec) ? 1 : 0;
        tout[i] = hit[i] ? rec.t : INFINITE;
    }
}
double ref_implicit_func(int kind, const double pos[3])
{
    if (kind == 6) return HeartShape::heartImplicitFunc(v3(pos));
    SurfaceOfGenus2Function f;
    return f.ImplicitFunc(v3(pos));
}
// The drawables themselves through Node::calHit (Surface.h:272-283): ray taken into the
// surface's frame, shared tree (built on first use with the reference's default voxel size),
// analytic normal, localp (HeartShape only).  The reference prints diagnostics to stdout from
// HeartShape::getRawNormal for some points; they are left alone.
void ref_implicit_object_hit(int kind, const double center[3], double scale, const double *rays, uint64_t n,
                             double t0, double t1, int32_t *hit, double *tout, double *normal, double *localp)
{
    std::shared_ptr<Drawable> d;
    if (kind == 6) d = std::make_shared<HeartShape>(v3(center), scale);
    else d = std::make_shared<SurfaceOfGenus2>(v3(center), scale);
    for (uint64_t i = 0; i < n; i++) {
        HitRecord rec;
        Ray r = mkray(rays + 6 * i);
        rec.ray = r;
        hit[i] = d->calHit(r, t0, t1, &rec) ? 1 : 0;
        tout[i] = hit[i] ? rec.t : INFINITE;
        st(normal + 3 * i, hit[i] ? rec.normal : Vec3(0, 0, 0));
        st(localp + 3 * i, hit[i] ? rec.localp : Vec3(0, 0, 0));
    }
}

// ---- Perlin noise and the Perlin materials (algorithms.h:56-111, algorithms.cpp:2-40,
// material.h:95-164, material.cpp:67-87)
double ref_perlin_noise(double x, double y, double z, double scale, const int32_t *P, const double *G, int n)
{
    std::vector<int> p(P, P + n);
    std::vector<Vec3> g;
    for (int i = 0; i < n; i++) g.push_back(Vec3(G[3 * i], G[3 * i + 1], G[3 * i + 2]));
    return algorithm::perlinNoise(x, y, z, scale, p, g, n);
}
// The generator's tables as its constructor fills them HERE (glibc rand(); the reference was
// written for MSVC's generator, so these are test data, not the Windows tables)
void ref_perlin_tables(int n, int32_t *P, double *G)
{
    algorithm::PerlinNoiseGenerator::P_.clear();
    algorithm::PerlinNoiseGenerator::G_.clear();
    algorithm::PerlinNoiseGenerator gen(n);
    for (int i = 0; i < n; i++) {
        P[i] = algorithm::PerlinNoiseGenerator::P_[i];
        st(G + 3 * i, algorithm::PerlinNoiseGenerator::G_[i]);
    }
}
// kind 1: PerlinNoiseNormalMaterial, 2: PerlinNoiseNormalMaterial_Blood, with the generator's
// static tables set to P, G.  out = getAmbient, getDiffuse, getSpecular, getNormal(normal)
// after setPosition(pos) (12 doubles).
void ref_perlin_material(int kind, const double amb[3], const double dif[3], const double spe[3], double shine, double scale,
                         const int32_t *P, const double *G, int n, const double pos[3], const double normal[3], double out[12])
{
    std::shared_ptr<PerlinNoiseNormalMaterial> m;
    Color a(amb[0], amb[1], amb[2], 1), d(dif[0], dif[1], dif[2], 1), sp(spe[0], spe[1], spe[2], 1);
    if (kind == 1) m = std::make_shared<PerlinNoiseNormalMaterial>(a, d, sp, shine, scale);
    else m = std::make_shared<PerlinNoiseNormalMaterial_Blood>(a, d, sp, shine, scale);
    algorithm::PerlinNoiseGenerator::P_.assign(P, P + n);
    algorithm::PerlinNoiseGenerator::G_.clear();
    for (int i = 0; i < n; i++) algorithm::PerlinNoiseGenerator::G_.push_back(Vec3(G[3 * i], G[3 * i + 1], G[3 * i + 2]));
    m->setPosition(v3(pos));
    HitRecord rec;
    Color ca = m->getAmbient(&rec), cd = m->getDiffuse(&rec), cs = m->getSpecular(&rec);
    Vec3 nn = m->getNormal(v3(normal));
    out[0] = ca.x_; out[1] = ca.y_; out[2] = ca.z_;
    out[3] = cd.x_; out[4] = cd.y_; out[5] = cd.z_;
    out[6] = cs.x_; out[7] = cs.y_; out[8] = cs.z_;
    st(out + 9, nn);
}

// SigmoidMethod (1) / SqrtMethod (2)::reproduction, tonereproductionmethod.h:11-35
void ref_tone(int method, const double in[3], double out[3])
{
    Color c(in[0], in[1], in[2], 1);
    Color r = method == 1 ? SigmoidMethod().reproduction(c) : SqrtMethod().reproduction(c);
    out[0] = r.x_; out[1] = r.y_; out[2] = r.z_;
}

// PerlinNoiseGeneratorWithTime::timeChanged (algorithms.cpp:57-77) on given tables and static state.
// clock = {count, stage}; G is updated in place.
void ref_perlin_time_changed(int32_t clock[2], double *stime, double prev_time, double time, int n, double *G)
{
    using Gen = algorithm::PerlinNoiseGeneratorWithTime;
    algorithm::PerlinNoiseGenerator::P_.assign(n, 0);
    for (int i = 0; i < n; i++) algorithm::PerlinNoiseGenerator::P_[i] = i;
    algorithm::PerlinNoiseGenerator::G_.clear();
    for (int i = 0; i < n; i++) algorithm::PerlinNoiseGenerator::G_.push_back(Vec3(G[3 * i], G[3 * i + 1], G[3 * i + 2]));
    Gen gen(n);
    Gen::count = clock[0]; Gen::stage = clock[1]; Gen::stime = *stime;
    gen.time_ = prev_time;
    gen.timeChanged(time);
    clock[0] = Gen::count; clock[1] = Gen::stage; *stime = Gen::stime;
    for (int i = 0; i < n; i++) st(G + 3 * i, algorithm::PerlinNoiseGenerator::G_[i]);
}

// ChangeFovCamera under RTObject::setTime (camera.cpp:74-94): out = e, llc, u, v, width, height (14 doubles)
void ref_change_fov_camera(const double from[3], const double at[3], const double vup[3], double vfov, double aspect,
                           double time, double out[14])
{
    ChangeFovCamera c(v3(from), v3(at), v3(vup), vfov, aspect);
    c.setTime(time);
    st(out, c.e_); st(out + 3, c.leftLowerCorner); st(out + 6, c.u_); st(out + 9, c.v_);
    out[12] = c.width; out[13] = c.height;
}
// MovableFaceLight::getPosition at a time (light.cpp:87-107) with table entry (u, v): the sampled
// point, and the light's own position after the move
void ref_movable_face_light(const double pos[3], const double normal[3], const double up[3], double w, double h,
                            double time, double u, double v, double out_sample[3], double out_pos[3])
{
    auto l = std::make_shared<MovableFaceLight>(v3(pos), v3(normal), v3(up), w, h, Color(1, 1, 1, 1), Color(1, 1, 1, 1), Color(1, 1, 1, 1));
    l->setTime(time);
    l->randomPairs_.assign(1, std::make_pair(u, v));
    l->itIndex_ = 0;
    Vec3 n; double p = 0;
    Vec3 r = l->getPosition(nullptr, &n, &p);
    st(out_sample, r); st(out_pos, l->pos_);
}

// DepthCamera::getRay (camera.cpp:62-72).  Its lens point comes from algorithm::randFromDisk
// (rand()); the generator is seeded, the point drawn here, then seeded again so that getRay draws
// the same one.  out = e, d; xy = the lens point.
void ref_depth_camera_ray(const double from[3], const double at[3], const double vup[3], double vfov, double aspect, double aperture,
                          unsigned seed, double s, double t, double ray[6], double xy[2])
{
    DepthCamera c(v3(from), v3(at), v3(vup), vfov, aspect, aperture);
    srand(seed);
    algorithm::randFromDisk(&xy[0], &xy[1]);
    srand(seed);
    Ray r = c.getRay((float)s, (float)t);
    st(ray, r.e); st(ray + 3, r.d);
}

// RTObject::calRandom structure check (RTObject.cpp:46-69) cannot be run on glibc
// (RAND_MAX + 1 overflows); only the table layout constants are exposed.
double ref_const_zero(void) { return ZERO; }
double ref_const_infinite(void) { return INFINITE; }
double ref_const_pi(void) { return PI; }

} // extern "C"
