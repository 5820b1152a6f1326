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
#include <tuple>
#include <unordered_set>

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

// Inserts primitives 0..n-1 (boxes pbox) in index order into a tree rooted at `root` and flattens
// it breadth first; record(o) yields primitive o's LeafTri.
template <class MakeRecord>
bool build_flat(const std::vector<Box> &pbox, const Box &root, int max_size, int overlap_tol, int max_depth,
                MakeRecord record, FlatTree &out, std::string &err)
{
    const uint64_t ntris = pbox.size();
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
            out.leaf_tris.push_back(record(o));
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

    return build_flat(pbox, root, max_size, overlap_tol, max_depth, [&](uint32_t o) {
        LeafTri lt;
        for (int k = 0; k < 3; k++) { lt.v0[k] = out.tri_rec[9ull * o + k]; lt.e1[k] = out.tri_rec[9ull * o + 3 + k]; lt.e2[k] = out.tri_rec[9ull * o + 6 + k]; }
        lt.id = o; lt.pad = 0;
        return lt;
    }, out, err);
}

// ---------------------------------------------------------------------------------------------
// Implicit surfaces.  kind 0: HeartShape::heartImplicitFunc (Surface.h:218-224); kind 1:
// SurfaceOfGenus2Function::ImplicitFunc (Surface.cpp:318-325).  Both take y and z swapped.
double implicit_value(int kind, const double pos[3])
{
    const double x = pos[0], y = pos[2], z = pos[1];
    if (kind == 0) {
        const double temp = x * x + y * y * 9.0 / 4 + z * z - 1;
        return -(x * x * z * z * z) - y * y * z * z * z * 9.0 / 80 + temp * temp * temp;
    }
    const double temp = x * x + y * y;
    return 2 * y * (y * y - 3 * x * x) * (1 - z * z) + temp * temp - (9 * z * z - 1) * (1 - z * z);
}

namespace {

struct Cell {
    int64_t x, y, z;
    bool operator==(const Cell &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellHash {
    size_t operator()(const Cell &c) const
    {
        uint64_t h = (uint64_t)c.x * 0x9E3779B97F4A7C15ull;
        h = (h ^ (h >> 29)) + (uint64_t)c.y * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 31)) + (uint64_t)c.z * 0x94D049BB133111EBull;
        return (size_t)(h ^ (h >> 32));
    }
};

// algorithm::getAdjBox, algorithms.h:251-278: the box sharing face f (FRONT, TOP, LEFT, RIGHT,
// DOWN, BACK) with b, built from b's corners and one edge length
Box neighbour(const Box &b, int f)
{
    Box a;
    auto corner = [&](int i, double out[3]) { bound_point(i, b, out); };
    for (int k = 0; k < 3; k++) { a.lo[k] = b.lo[k]; a.hi[k] = b.hi[k]; }
    switch (f) {
    case 0: a.lo[0] = b.lo[0] - 0; a.lo[1] = b.lo[1] - (b.hi[1] - b.lo[1]); a.lo[2] = b.lo[2] - 0; corner(5, a.hi); break;
    case 1: corner(4, a.lo); a.hi[0] = b.hi[0] + 0; a.hi[1] = b.hi[1] + 0; a.hi[2] = b.hi[2] + (b.hi[2] - b.lo[2]); break;
    case 2: a.lo[0] = b.lo[0] - (b.hi[0] - b.lo[0]); a.lo[1] = b.lo[1] - 0; a.lo[2] = b.lo[2] - 0; corner(7, a.hi); break;
    case 3: corner(1, a.lo); a.hi[0] = b.hi[0] + (b.hi[0] - b.lo[0]); a.hi[1] = b.hi[1] + 0; a.hi[2] = b.hi[2] + 0; break;
    case 4: a.lo[0] = b.lo[0] - 0; a.lo[1] = b.lo[1] - 0; a.lo[2] = b.lo[2] - (b.hi[2] - b.lo[2]); corner(2, a.hi); break;
    default: corner(3, a.lo); a.hi[0] = b.hi[0] + 0; a.hi[1] = b.hi[1] + (b.hi[1] - b.lo[1]); a.hi[2] = b.hi[2] + 0; break;
    }
    return a;
}

bool crosses_surface(int kind, const Box &b)
{
    bool pos = false, neg = false;
    for (int i = 0; i < 8; i++) {
        double p[3];
        bound_point(i, b, p);
        const double v = implicit_value(kind, p);
        pos |= v > 0; neg |= v < 0;
    }
    return pos && neg;          // not all of one sign (zeros count for both, octree.cpp:184-192)
}

} // namespace

// ImplicitSurfaceOctree::build, octree.cpp:148-254.  The surface point between the two seed
// points is found by bisection (algorithm::calSingleRoot, algorithms.h:143-172), a voxel of half
// edge min_size is centred on it, and the voxel set grows breadth first over face neighbours that
// the surface crosses.  The reference keeps the voxels in a std::set ordered by lower corner (x,
// then y, then z, coordinates closer than 1e-9 being equal -- BoxNode::operator<, Surface.h:30-44)
// and inserts them into the octree in that order; here a voxel is identified by the integer cell
// of its lower corner relative to the first voxel (corners of one cell reached along different
// paths differ by rounding only), the first box found for a cell is the one kept, as in the set,
// and the cells are sorted lexicographically.
bool build_implicit_octree(int kind, double min_size, int max_size, int overlap_tol, int max_depth,
                           FlatTree &out, std::vector<double> &boxes, std::string &err)
{
    if (kind != 0 && kind != 1) { err = "unknown implicit surface"; return false; }
    if (!(min_size > 0.0) || !std::isfinite(min_size)) { err = "implicit_min_size must be positive"; return false; }
    if (max_size < 1) { err = "octree_max_size must be >= 1"; return false; }
    if (max_depth < 0 || max_depth > 15) { err = "octree_max_depth must be in [0,15]"; return false; }
    const double s0[3] = {0, 0, 0};
    const double s1[3] = {kind == 0 ? 2.0 : 0.0, kind == 0 ? 0.0 : 0.9, 0.0};           // Surface.cpp:302, 327-331
    double unit[3], dist;
    {
        const double v[3] = {s1[0] - s0[0], s1[1] - s0[1], s1[2] - s0[2]};
        dist = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double inv = 1 / dist;
        for (int k = 0; k < 3; k++) unit[k] = v[k] * inv;
    }
    auto along = [&](double t) { const double p[3] = {unit[0] * t + s0[0], unit[1] * t + s0[1], unit[2] * t + s0[2]}; return implicit_value(kind, p); };
    double ta = 0, tb = dist, fa = along(ta), fb = along(tb), root;
    if (fa == 0.0) root = ta;
    else if (fb == 0.0) root = tb;
    else if (fa * fb > 0.0) { err = "implicit surface: no sign change between the seed points"; return false; }
    else {
        double tm = (ta + tb) / 2, fm = along(tm);
        while (std::fabs(fm) > 10e-8) {
            if (fa * fm < 0) { tb = tm; fb = fm; } else { ta = tm; fa = fm; }
            tm = (ta + tb) / 2;
            fm = along(tm);
        }
        root = tm;
    }
    Box first;
    for (int k = 0; k < 3; k++) {
        const double m = unit[k] * root + s0[k];
        first.lo[k] = m - min_size; first.hi[k] = m + min_size;
    }
    const double edge = 2 * min_size;
    const size_t limit = 1u << 24;
    std::vector<Box> found;              // in the order they were discovered = the order they are visited
    std::vector<Cell> cell_of;
    std::unordered_set<Cell, CellHash> seen;
    found.push_back(first); cell_of.push_back(Cell{0, 0, 0}); seen.insert(Cell{0, 0, 0});
    Box all;
    for (int k = 0; k < 3; k++) { all.lo[k] = 1e10; all.hi[k] = -1e10; }
    for (size_t head = 0; head < found.size(); head++) {
        const Box b = found[head];
        for (int k = 0; k < 3; k++) { all.lo[k] = smin(all.lo[k], b.lo[k]); all.hi[k] = smax(all.hi[k], b.hi[k]); }
        for (int f = 0; f < 6; f++) {
            const Box a = neighbour(b, f);
            const Cell c{std::llround((a.lo[0] - first.lo[0]) / edge), std::llround((a.lo[1] - first.lo[1]) / edge), std::llround((a.lo[2] - first.lo[2]) / edge)};
            if (seen.count(c)) continue;
            if (!crosses_surface(kind, a)) continue;
            if (found.size() >= limit) { err = "implicit surface: too many voxels (implicit_min_size too small)"; return false; }
            seen.insert(c);
            found.push_back(a); cell_of.push_back(c);
        }
    }
    std::vector<uint32_t> order(found.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) {
        return std::tie(cell_of[p].x, cell_of[p].y, cell_of[p].z) < std::tie(cell_of[q].x, cell_of[q].y, cell_of[q].z);
    });
    std::vector<Box> pbox(found.size());
    boxes.resize(found.size() * 6);
    for (size_t i = 0; i < order.size(); i++) {
        pbox[i] = found[order[i]];
        for (int k = 0; k < 3; k++) { boxes[6 * i + k] = pbox[i].lo[k]; boxes[6 * i + 3 + k] = pbox[i].hi[k]; }
    }
    out.tri_rec.clear();
    // a voxel travels in a triangle record: v0 = lower corner, e1 = upper corner, e2[0] = length of
    // the diagonal (what BoxNode::calHit moves the entry point back by, Surface.cpp:55), and the
    // top bit of the id says that the record is a voxel
    const bool ok = build_flat(pbox, all, max_size, overlap_tol, max_depth, [&](uint32_t o) {
        LeafTri lt;
        const Box &b = pbox[o];
        const double dx = b.lo[0] - b.hi[0], dy = b.lo[1] - b.hi[1], dz = b.lo[2] - b.hi[2];
        for (int k = 0; k < 3; k++) { lt.v0[k] = b.lo[k]; lt.e1[k] = b.hi[k]; lt.e2[k] = 0; }
        lt.e2[0] = std::sqrt(dx * dx + dy * dy + dz * dz);
        lt.id = o | LEAF_VOXEL; lt.pad = 0;
        return lt;
    }, out, err);
    for (FlatNode &f : out.nodes) f.pad = 1;          // marks a node whose primitives are voxels
    return ok;
}

} // namespace rt
