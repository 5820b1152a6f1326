// rt_octree.cpp -- host-side octree construction + flattening (runs once per scene).
//
// Insert policy follows OctreeNode::insertObject (octree.cpp:61-99): a node collects
// primitives until it holds max_size of them, then creates its 8 children and re-inserts
// everything; in a split node a primitive whose box strictly overlaps more than
// overlap_tol children stays at the node, otherwise it is inserted into every overlapped
// child (duplication).  Only hit results are observable, so the in-memory form here (index
// vectors, then a breadth-first array) is free to differ from the reference's pointer tree;
// the per-node primitive ORDER is kept because ties inside a node go to the later entry
// (octree.cpp:31).
#include "rt_octree.h"

#include <algorithm>
#include <cmath>
#include <deque>

namespace rt {
namespace {

struct Box { double lo[3], hi[3]; };

// std::min / std::max argument order of the reference expressions is kept
inline double smin(double a, double b) { return (b < a) ? b : a; }
inline double smax(double a, double b) { return (a < b) ? b : a; }

// Corner `index` of a box: algorithm::getBoundPoint, algorithms.h:299-321
inline void bound_point(int index, const Box &b, double out[3])
{
    static const int ixs[4] = {0, 1, 1, 0}, iys[4] = {0, 0, 1, 1};
    int ix = ixs[index % 4], iy = iys[index % 4], iz = (index >> 2) & 1;
    out[0] = ix ? b.hi[0] : b.lo[0];
    out[1] = iy ? b.hi[1] : b.lo[1];
    out[2] = iz ? b.hi[2] : b.lo[2];
}
inline void mid(int a, int b, const Box &bx, double out[3])
{
    double p[3], q[3];
    bound_point(a, bx, p); bound_point(b, bx, q);
    for (int k = 0; k < 3; k++) out[k] = (p[k] + q[k]) / 2;
}
// OctreeNode::getChildBound, octree.h:82-120
void child_bound(const Box &b, int c, Box &out)
{
    switch (c) {
    case 0: bound_point(0, b, out.lo); mid(0, 6, b, out.hi); break;
    case 1: mid(0, 1, b, out.lo); mid(1, 6, b, out.hi); break;
    case 2: mid(0, 2, b, out.lo); mid(2, 6, b, out.hi); break;
    case 3: mid(0, 3, b, out.lo); mid(3, 6, b, out.hi); break;
    case 4: mid(0, 4, b, out.lo); mid(5, 7, b, out.hi); break;
    case 5: mid(1, 4, b, out.lo); mid(5, 6, b, out.hi); break;
    case 6: mid(1, 7, b, out.lo); bound_point(6, b, out.hi); break;
    default: mid(0, 7, b, out.lo); mid(6, 7, b, out.hi); break;
    }
}
// algorithm::isBoundBoxOverLap, algorithms.h:329-333 (strict)
inline bool overlap(const Box &a, const Box &b)
{
    return smax(a.lo[0], b.lo[0]) < smin(a.hi[0], b.hi[0]) &&
           smax(a.lo[1], b.lo[1]) < smin(a.hi[1], b.hi[1]) &&
           smax(a.lo[2], b.lo[2]) < smin(a.hi[2], b.hi[2]);
}

struct BNode {
    Box box;
    int depth = 0;
    bool full = false;
    int child[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    std::vector<uint32_t> objs;
};

struct Builder {
    std::deque<BNode> nodes;      // deque: references stay valid while children are appended
    const std::vector<Box> &pbox;
    int max_size, tol, max_depth;
    uint64_t lost = 0;

    Builder(const std::vector<Box> &pb, int ms, int t, int md) : pbox(pb), max_size(ms), tol(t), max_depth(md) {}

    void insert(int ni, uint32_t o)
    {
        BNode &n = nodes[ni];
        if (!n.full) {
            n.objs.push_back(o);
            if ((int)n.objs.size() == max_size && n.depth < max_depth) {
                n.full = true;
                for (int i = 0; i < 8; i++) {
                    BNode c;
                    child_bound(n.box, i, c.box);
                    c.depth = n.depth + 1;
                    nodes.push_back(std::move(c));
                    nodes[ni].child[i] = (int)nodes.size() - 1;
                }
                std::vector<uint32_t> old;
                old.swap(nodes[ni].objs);
                for (uint32_t q : old) insert(ni, q);
            }
            return;
        }
        bool a[8];
        int count = 0;
        for (int i = 0; i < 8; i++) {
            a[i] = overlap(pbox[o], nodes[n.child[i]].box);
            count += a[i] ? 1 : 0;
        }
        if (count > tol) { n.objs.push_back(o); return; }
        if (count == 0) lost++;
        for (int i = 0; i < 8; i++)
            if (a[i]) insert(nodes[ni].child[i], o);
    }
};

} // namespace

bool build_octree(const double *tris, uint64_t ntris, double pad, int max_size, int overlap_tol,
                  int max_depth, FlatTree &out, std::string &err)
{
    if (ntris >= 0x7FFFFFFFull) { err = "mesh has too many triangles"; return false; }
    if (max_size < 1) { err = "octree_max_size must be >= 1"; return false; }
    if (max_depth < 0 || max_depth > 15) { err = "octree_max_depth must be in [0,15]"; return false; }
    if (!(pad >= 0.0)) { err = "tri_box_pad must be >= 0"; return false; }

    std::vector<Box> pbox(ntris);
    Box root;
    for (int k = 0; k < 3; k++) { root.lo[k] = 1e10; root.hi[k] = -1e10; }
    out.tri_rec.resize(ntris * 9);
    for (uint64_t i = 0; i < ntris; i++) {
        const double *t = tris + 9 * i;
        for (int k = 0; k < 9; k++)
            if (!std::isfinite(t[k])) { err = "triangle vertex is not finite"; return false; }
        for (int k = 0; k < 3; k++) {
            pbox[i].lo[k] = smin(smin(t[k], t[3 + k]), t[6 + k]) - pad;
            pbox[i].hi[k] = smax(smax(t[k], t[3 + k]), t[6 + k]) + pad;
            root.lo[k] = smin(root.lo[k], pbox[i].lo[k]);
            root.hi[k] = smax(root.hi[k], pbox[i].hi[k]);
            out.tri_rec[9 * i + k] = t[k];
            out.tri_rec[9 * i + 3 + k] = t[3 + k] - t[k];
            out.tri_rec[9 * i + 6 + k] = t[6 + k] - t[k];
        }
    }

    Builder b(pbox, max_size, overlap_tol, max_depth);
    BNode r;
    r.box = root;
    b.nodes.push_back(std::move(r));
    for (uint64_t i = 0; i < ntris; i++) b.insert(0, (uint32_t)i);

    // breadth-first flattening: the 8 children of a node become 8 consecutive records
    std::vector<int> order;          // flat index -> builder index
    order.reserve(b.nodes.size());
    order.push_back(0);
    out.nodes.clear();
    out.leaf_tris.clear();
    for (size_t fi = 0; fi < order.size(); fi++) {
        const BNode &n = b.nodes[order[fi]];
        FlatNode f;
        for (int k = 0; k < 3; k++) { f.lo[k] = n.box.lo[k]; f.hi[k] = n.box.hi[k]; }
        f.pad = 0;
        f.obj_begin = (uint32_t)out.leaf_tris.size();
        f.obj_count = (uint32_t)n.objs.size();
        for (uint32_t o : n.objs) {
            LeafTri lt;
            for (int k = 0; k < 3; k++) { lt.v0[k] = out.tri_rec[9ull * o + k]; lt.e1[k] = out.tri_rec[9ull * o + 3 + k]; lt.e2[k] = out.tri_rec[9ull * o + 6 + k]; }
            lt.id = o; lt.pad = 0;
            out.leaf_tris.push_back(lt);
        }
        if (n.full) {
            f.child_base = (uint32_t)order.size();
            for (int i = 0; i < 8; i++) order.push_back(n.child[i]);
        } else {
            f.child_base = 0;
        }
        out.nodes.push_back(f);
        out.stats[0]++;
        out.stats[2] += n.objs.size();
        if (!n.full) out.stats[1]++; else out.stats[3] += n.objs.size();
        if ((uint64_t)n.depth > out.stats[4]) out.stats[4] = n.depth;
    }
    out.stats[5] = b.lost;

    // The kernels derive a child's box from its parent's as {lo, (lo+hi)/2, hi} per axis; check
    // that this closed form reproduces the stored corner-midpoint boxes bit for bit.
    for (const FlatNode &f : out.nodes) {
        if (!f.child_base) continue;
        for (int c = 0; c < 8; c++) {
            static const int hx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, hy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, hz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
            const int h[3] = {hx[c], hy[c], hz[c]};
            const FlatNode &ch = out.nodes[f.child_base + c];
            for (int k = 0; k < 3; k++) {
                double m = (f.lo[k] + f.hi[k]) / 2;
                double elo = h[k] ? m : f.lo[k], ehi = h[k] ? f.hi[k] : m;
                if (ch.lo[k] != elo || ch.hi[k] != ehi) { err = "internal: child box closed form mismatch"; return false; }
            }
        }
    }
    return true;
}

} // namespace rt
