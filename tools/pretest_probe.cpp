// pretest_probe.cpp -- development aid (host only): for one ray and one mesh, every record on the ray's walk
// through the mesh's octree that the double-precision test accepts, with what the single-precision box test
// (replayed here in the kernels' own operation order) makes of the record's box, its segment's and its run's.
//   g++ -O2 -std=c++17 -I raytracer_amd/csrc tools/pretest_probe.cpp raytracer_amd/csrc/rt_octree.cpp -o /tmp/pretest_probe
//   /tmp/pretest_probe mesh.bin max_size max_depth  ex ey ez dx dy dz  t0 t1
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_octree.h"
#include "rt_pretest.h"

using namespace rt;

struct V { double x, y, z; };
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V l, V r) { return {l.y * r.z - r.y * l.z, l.z * r.x - l.x * r.z, l.x * r.y - r.x * l.y}; }

static bool hit_box(const double *lo, const double *hi, V e, V d, double t0, double t1, double &r0, double &r1)
{
    double a[3] = {(lo[0] - e.x) / d.x, (lo[1] - e.y) / d.y, (lo[2] - e.z) / d.z};
    double b[3] = {(hi[0] - e.x) / d.x, (hi[1] - e.y) / d.y, (hi[2] - e.z) / d.z};
    r0 = std::min(a[0], b[0]); r1 = std::max(a[0], b[0]);
    for (int k = 1; k < 3; k++) { double m = std::min(a[k], b[k]); if (r0 < m) r0 = m; m = std::max(a[k], b[k]); if (m < r1) r1 = m; }
    if (r0 < t0) r0 = t0;
    if (t1 < r1) r1 = t1;
    return r0 <= r1;
}
static bool tri_hit(const LeafTri &tr, V e, V d, double t0, double t1, double &tout)
{
    V v0{tr.v0[0], tr.v0[1], tr.v0[2]}, e1{tr.e1[0], tr.e1[1], tr.e1[2]}, e2{tr.e2[0], tr.e2[1], tr.e2[2]};
    V p = cross(d, e2);
    double det = dot(e1, p), a = std::fabs(det);
    if (!(a >= 1e-200)) return false;
    if (!(det * det >= 1e-10 * dot(e1, e1) * dot(e2, e2) * dot(d, d))) return false;
    V s = e - v0;
    double up = dot(s, p), us = det < 0 ? -up : up;
    if (!(us >= 0.0 && us <= a)) return false;
    V q = cross(s, e1);
    double vp = dot(d, q), vs = det < 0 ? -vp : vp;
    if (!(vs >= 0.0 && us + vs <= a)) return false;
    double t = dot(e2, q) / det;
    if (!(t >= t0 && t <= t1)) return false;
    tout = t;
    return true;
}
struct RayB { float ix, iy, iz, tend, cx, cy, cz; };
static bool box_pass(uint32_t w0, uint32_t w1, const RayB &r, float *tn_out, float *tf_out)
{
    const float SL = 4e-3f;
    const float lx = (float)(w0 & 255u), ly = (float)((w0 >> 8) & 255u), lz = (float)((w0 >> 16) & 255u), hx = (float)(w0 >> 24);
    const float hy = (float)(w1 & 255u), hz = (float)((w1 >> 8) & 255u);
    const float ax = std::fmaf(lx, r.ix, r.cx), bx = std::fmaf(hx, r.ix, r.cx);
    const float ay = std::fmaf(ly, r.iy, r.cy), by = std::fmaf(hy, r.iy, r.cy);
    const float az = std::fmaf(lz, r.iz, r.cz), bz = std::fmaf(hz, r.iz, r.cz);
    const float tn = std::fmax(std::fmax(std::fmin(ax, bx), std::fmin(ay, by)), std::fmax(std::fmin(az, bz), -SL));
    const float tf = std::fmin(std::fmin(std::fmax(ax, bx), std::fmax(ay, by)), std::fmin(std::fmax(az, bz), r.tend));
    *tn_out = tn; *tf_out = tf;
    return !(tn > tf);
}

int main(int argc, char **argv)
{
    if (argc < 12) return 1;
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<double> tris(sz / 8);
    if (fread(tris.data(), 8, tris.size(), f) != tris.size()) return 1;
    fclose(f);
    FlatTree ft;
    std::string err;
    const double pad = getenv("RT_PAD") ? atof(getenv("RT_PAD")) : 1e-7;
    const int tol = getenv("RT_TOL") ? atoi(getenv("RT_TOL")) : 3;
    if (!build_octree(tris.data(), tris.size() / 9, pad, atoi(argv[2]), tol, atoi(argv[3]), ft, err)) { printf("%s\n", err.c_str()); return 1; }
    V e{atof(argv[4]), atof(argv[5]), atof(argv[6])}, d{atof(argv[7]), atof(argv[8]), atof(argv[9])};
    const double t0 = atof(argv[10]), t1 = atof(argv[11]);
    printf("%zu nodes, %zu records\n", ft.nodes.size(), ft.leaf_tris.size());
    const float fdx = (float)d.x, fdy = (float)d.y, fdz = (float)d.z;
    const float l2 = std::fmaf(fdx, fdx, std::fmaf(fdy, fdy, fdz * fdz)), inv = 1.0f / std::sqrt(l2);
    const float ix = 1.0f / (fdx * inv), iy = 1.0f / (fdy * inv), iz = 1.0f / (fdz * inv), dlen = l2 * inv;
    for (size_t ni = 0; ni < ft.nodes.size(); ni++) {
        const FlatNode &n = ft.nodes[ni];
        if (!n.obj_count) continue;
        double n0, n1;
        if (!hit_box(n.lo, n.hi, e, d, t0, t1, n0, n1)) continue;
        const float sc = node_grid_scale(n);
        const float gx = (float)((e.x + d.x * n0) - n.lo[0]) * sc + (float)RT_GRID_ORG, gy = (float)((e.y + d.y * n0) - n.lo[1]) * sc + (float)RT_GRID_ORG;
        const float gz = (float)((e.z + d.z * n0) - n.lo[2]) * sc + (float)RT_GRID_ORG;
        RayB rb{ix, iy, iz, (float)(n1 - n0) * dlen * sc + 4e-3f, -(gx * ix), -(gy * iy), -(gz * iz)};
        int rlo[3] = {256, 256, 256}, rhi[3] = {-1, -1, -1};
        std::vector<uint32_t> w(2 * n.obj_count);
        std::vector<int> lo(3 * n.obj_count), hi(3 * n.obj_count);
        for (uint32_t j = 0; j < n.obj_count; j++) {
            const LeafTri &t = ft.leaf_tris[n.obj_begin + j];
            const BoxD b = clipped_triangle_box(t, n);
            pack_box(&w[2 * j], b, n, sc, &lo[3 * j], &hi[3 * j]);
            for (int k = 0; k < 3; k++) { rlo[k] = std::min(rlo[k], lo[3 * j + k]); rhi[k] = std::max(rhi[k], hi[3 * j + k]); }
        }
        uint32_t rw[2];
        store_grid_box(rw, rlo, rhi);
        for (uint32_t j = 0; j < n.obj_count; j++) {
            const LeafTri &t = ft.leaf_tris[n.obj_begin + j];
            double tt;
            if (!tri_hit(t, e, d, n0, n1, tt)) continue;
            float tn, tf, rn, rf;
            const bool pass = box_pass(w[2 * j], w[2 * j + 1], rb, &tn, &tf), rpass = box_pass(rw[0], rw[1], rb, &rn, &rf);
            const BoxD b = clipped_triangle_box(t, n);
            printf("node %zu %s box [%g %g %g]..[%g %g %g] window [%.17g, %.17g] scale %g entry grid (%.6f %.6f %.6f) tend %g\n", ni, n.child_base ? "inner" : "leaf",
                   n.lo[0], n.lo[1], n.lo[2], n.hi[0], n.hi[1], n.hi[2], n0, n1, sc, gx, gy, gz, rb.tend);
            printf("  record %u id %u hit t %.17g: clipped ok %d empty %d [%g %g %g]..[%g %g %g] bytes lo (%d %d %d) hi (%d %d %d) -> record box %s (tn %g tf %g), run box %s (tn %g tf %g)\n",
                   j, t.id, tt, (int)b.ok, (int)b.empty, b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], lo[3 * j], lo[3 * j + 1], lo[3 * j + 2], hi[3 * j], hi[3 * j + 1], hi[3 * j + 2],
                   pass ? "PASS" : "REJECT", tn, tf, rpass ? "PASS" : "REJECT", rn, rf);
        }
    }
    return 0;
}
